import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# A one-shot drop-in call builds the per-RX / per-TX direction tables only for launch sets of 2^26 rays
# or more (they cost more than they save below that: csrc/host/problem.c).  The parity tests want the
# table path exercised at every size: subprocesses inherit this too.
os.environ.setdefault("HRT_TUNE", "rxt_min_rays=0")   # (csrc/host/tune.c: the test switches live in HRT_TUNE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # a -m gpu run on a box without a GPU must fail loudly, not skip: nothing to do here.
    pass


@pytest.fixture(scope="session")
def product_lib():
    """The product's C-ABI library (HIP path).  Built by __graft_entry__.build()."""
    import hermespy_rt_amd.lib as L
    return L.load()


@pytest.fixture(scope="session")
def ref_lib():
    """The REAL reference, built in place by `make -C oracle ref` (absent -> skip)."""
    import ctypes
    from hermespy_rt_amd import abi
    p = os.path.join(REPO, "oracle", "_ref", "libhrt_ref.so")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/libhrt_ref.so not built (needs /root/reference)")
    from . import refabi
    return refabi.load()
