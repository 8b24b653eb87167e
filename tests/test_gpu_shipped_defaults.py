"""The drop-in exactly as a user gets it.  tests/conftest.py sets HRT_TUNE=rxt_min_rays=0 for the whole
suite so that the direction tables / candidate masks are exercised at every size; a one-shot
compute_paths() call builds them only from 2^26 rays on (2^18 on tables of <= 64 triangles).  Here
the dense-parity cases run in a child process WITHOUT that variable -- the shipped default -- plus
two launch sets that cross the 2^18 gate on tiny tables, so both sides of the default are in the
driver's suite."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import os, sys
sys.path.insert(0, %(repo)r)
assert "HRT_TUNE" not in os.environ and "HRT_FUSE" not in os.environ
from hermespy_rt_amd import abi, lib
from oracle import oracle
from tests import configs as K
from tests.parity import compare_dense
from tests.test_gpu_dense_parity import CASES
L = lib.load()
cases = dict(CASES)
cases["C2_reflector_300k_masks_on"] = K.small(K.C2, 300000)          # >= 2^18 rays, 2 triangles
cases["C4_2cars_2x140k_masks_on"] = K.small(K.C4_DOPPLER, 140000)    # 2 TX x 140k >= 2^18, 26 triangles
for name, c in cases.items():
    got = abi.run_compute_paths(L, *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    s = compare_dense(got, ref)
    assert all(v == 0 for v in s.values()), (name, s)
print("DEFAULTS_OK", len(cases))
"""


def test_dense_parity_with_the_shipped_defaults():
    env = {k: v for k, v in os.environ.items() if k not in ("HRT_TUNE", "HRT_FUSE")}
    p = subprocess.run([sys.executable, "-c", CODE % dict(repo=REPO)], env=env, capture_output=True, text=True)
    assert p.returncode == 0 and "DEFAULTS_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
