"""HRT_FULL_RECORDS=1 (all nine record fields copied over PCIe) against the default slim path (directions_rx and
tau formed on the host from per-hit origin and delay with the reference's float sequence): the same bits in every
dense array -- the slim path's claim (csrc/host/compute_paths.c, scatter_ctx)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys
sys.path.insert(0, %(repo)r)
import numpy as np
from hermespy_rt_amd import abi, lib
from tests import configs as K
L = lib.load()
out = {}
for name, c in (("c3", K.small(K.C3_DOPPLER, 150000)), ("c4", K.small(K.C4_DOPPLER, 100000)), ("c2", K.small(K.C2, 50000))):
    def flat(prefix, d):
        for k, v in d.items():
            if isinstance(v, dict):
                flat(prefix + "_" + k, v)
            elif isinstance(v, np.ndarray):
                out[prefix + "_" + k] = v
    flat(name, abi.run_compute_paths(L, *K.args(c)))
np.savez(sys.argv[1], **out)
"""


def test_full_and_slim_records_give_the_same_dense_arrays(tmp_path):
    import numpy as np
    files = []
    for name, env in (("slim", dict(os.environ)), ("full", dict(os.environ, HRT_FULL_RECORDS="1"))):
        f = str(tmp_path / (name + ".npz"))
        p = subprocess.run([sys.executable, "-c", CODE % dict(repo=REPO), f], env=env, capture_output=True, text=True)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        files.append(np.load(f))
    a, b = files
    assert sorted(a.files) == sorted(b.files) and len(a.files) > 10
    for k in a.files:
        x, y = a[k], b[k]
        assert x.shape == y.shape and x.dtype == y.dtype, k
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), k
