"""The oracle (oracle/hrt_oracle.c) against the committed golden vectors recorded from the
REAL reference (tests/golden/make_golden.py): every output array, bit for bit -- written
slots, untouched sentinels, RaysInfo snapshots and active masks included."""
import os

import numpy as np
import pytest

from oracle import oracle

from . import configs as K
from .golden.make_golden import SMALL, flat

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", list(SMALL))
def test_oracle_matches_reference_golden(name):
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    res = oracle.compute_paths(*K.args(SMALL[name]))
    assert not (res["scat"]["directions_tx"].view(np.uint32) != oracle.SENTINEL_U32).any()
    got = flat(res)
    assert set(got) == set(gold.files)
    for k in gold.files:
        assert np.array_equal(got[k], gold[k]), "%s: %s differs from the reference" % (name, k)


def test_known_answers_box():
    """SURVEY.md 8(c) anchors measured on the reference (box.hrt, C1)."""
    r = oracle.compute_paths(*K.args(K.C1))
    los, s, ex = r["los"], r["scat"], r["extras"]
    assert np.float32(los["a_te_re"][0, 0]) == np.float32(0.00324648875)
    assert np.float32(los["tau"][0, 0]) == np.float32(8.17061885e-09)
    assert np.array_equal(np.asarray(los["directions_tx"][0, 0], np.float32), np.asarray([0.816496551, 0.408248276, -0.408248276], np.float32))
    assert np.float32(s["tau"][0, 0, 0, 0]) == np.float32(2.22000001e-08)
    assert np.float32(s["a_te_re"][0, 0, 0, 0]) == np.float32(-6.98922994e-12)
    assert np.float32(s["a_te_im"][0, 0, 0, 0]) == np.float32(-5.67484818e-12)
    assert np.float32(s["tau"][0, 0, 0, 4999]) == np.float32(4.65228744e-08)
    assert np.array_equal(np.asarray(ex["launch_dirs"][0], np.float32), np.asarray([0.00512505323, -0.0131816929, 0.999899983], np.float32))
    assert list(ex["live"]) == [10000, 10000] and ex["tests"] == 240012
    eta = ex["eta_table"][0]   # box is "air"? no: check the row the scene uses
    used = [i for i in range(17) if ex["eta_table"][i].any()]
    assert len(used) == 1
    # eta of concrete at 3 GHz
    if used[0] == 1:
        e = ex["eta_table"][1]
        assert np.float32(e[0]) == np.float32(5.23999977) and np.float32(e[4]) == np.float32(0.653726876)
        assert np.float32(e[1]) == np.float32(2.29353666) and np.float32(e[11]) == np.float32(0.5)
