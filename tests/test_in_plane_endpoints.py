"""Endpoints exactly in the planes of scene triangles (tests/configs.py IN_PLANE): the reference's
hit decisions there are rounding noise, and parity means reproducing that noise.
CPU part: the oracle against the LIVE reference (skipped where oracle/_ref is absent).
GPU part: the product against the oracle, all three intersection variants."""
import os
import subprocess
import sys

import pytest

from tests.tune import tuned

from hermespy_rt_amd import abi
from oracle import oracle

from . import configs as K
from .parity import compare_dense

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", list(K.IN_PLANE))
def test_oracle_equals_reference_in_plane(ref_lib, name):
    c = K.IN_PLANE[name]
    ref = abi.run_compute_paths(ref_lib, *K.args(c))
    got = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), st


CODE = r"""
import sys
sys.path.insert(0, %(repo)r)
from hermespy_rt_amd import abi, lib
from oracle import oracle
from tests import configs as K
from tests.parity import compare_dense
for name, c in K.IN_PLANE.items():
    got = abi.run_compute_paths(lib.load(), *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), (name, st)
print("IN_PLANE_OK")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_product_equals_oracle_in_plane(variant):
    env = tuned(variant=variant)
    p = subprocess.run([sys.executable, "-c", CODE % dict(repo=REPO)], env=env, capture_output=True, text=True)
    assert p.returncode == 0 and "IN_PLANE_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
