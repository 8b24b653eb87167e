"""Parity on generated scenes (tests/scenes_gen.py): code paths the bundled scenes do not reach.
CPU part: the oracle against the LIVE reference on every generated scene (skipped where
oracle/_ref is absent).  GPU part: the product against the oracle, all intersection variants."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.tune import tuned

from hermespy_rt_amd import abi
from oracle import oracle

from . import configs as K
from . import scenes_gen as G
from .parity import assert_bit_equal, compare_dense

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RX = [[5, 3, 1.5], [-8, -4, 2.0], [12, 9, 8.0]]
TX = [[-10, 5, 6.0]]
RXV = [[1, 2, 0], [0, -3, 1], [2, 0, 0]]


def make(tmp, name):
    p = os.path.join(str(tmp), name + ".hrt")
    if name == "nasty":
        G.nasty(p)
        return G.cfg(p, [[1, 1, 2.5], [-3, 2, 1.0]], [[0.5, 0.7, 3.0]], 6000, 3, rx_vel=RXV[:2], tx_vel=[[4, 4, 0]])
    n_boxes, tilt, npth, nb = {"t300": (24, False, 6000, 3), "t300_tilted": (24, True, 6000, 3),
                               "t1104": (91, True, 3000, 2), "t2004_global_table": (166, True, 2000, 2),
                               "t6012_six_blocks": (500, True, 1500, 2)}[name]
    G.room_with_clutter(p, n_boxes, seed=len(name), tilt=tilt)
    return G.cfg(p, RX, TX, npth, nb, rx_vel=RXV, tx_vel=[[10, 0, 0]])


NAMES = ["t300", "t300_tilted", "t1104", "t2004_global_table", "t6012_six_blocks", "nasty"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_reference_on_generated_scene(ref_lib, name, tmp_path):
    c = make(tmp_path, name)
    ref = abi.run_compute_paths(ref_lib, *K.args(c))
    got = oracle.compute_paths(*K.args(c))
    for blk in ("los", "scat"):
        for k in ref[blk]:
            assert_bit_equal(got[blk][k], ref[blk][k], "%s.%s" % (blk, k))
    for k in ("los_rays", "los_active", "scat_rays", "scat_active"):
        assert_bit_equal(got[k], ref[k], k)
    hits = got["extras"]["live"]
    assert hits[1] > 0


CODE = r"""
import sys
sys.path.insert(0, %(repo)r)
from hermespy_rt_amd import abi, lib
from oracle import oracle
from tests import configs as K
from tests.parity import compare_dense
from tests.test_generated_scenes import make, NAMES
import tempfile
tmp = tempfile.mkdtemp()
for name in NAMES:
    c = make(tmp, name)
    got = abi.run_compute_paths(lib.load(), *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), (name, st)
    print(name, "ok", [int(x) for x in ref["extras"]["live"]])
print("GENERATED_OK")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("variant,lds_max", [(0, None), (1, None), (2, None), (2, 147456), (2, 0)])
def test_product_equals_oracle_on_generated_scenes(variant, lds_max):
    # lds_max: the triangle table staged in LDS up to the hardware limit (the > 64 KiB dynamic-LDS
    # path, one workgroup per CU) or never (every scene through the global-memory path); default:
    # staged up to 40 KiB
    env = tuned(variant=variant) if lds_max is None else tuned(variant=variant, lds_tri_bytes=lds_max)
    p = subprocess.run([sys.executable, "-c", CODE % dict(repo=REPO)], env=env, capture_output=True, text=True)
    assert p.returncode == 0 and "GENERATED_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
