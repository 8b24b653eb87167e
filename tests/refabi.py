"""Checker plumbing: the REAL reference (oracle/_ref/libhrt_ref.so, built in place from
/root/reference by `make -C oracle ref`) behind the same ctypes declarations as the product.
Only tests/, tests/golden/make_golden.py and bench.py's cpu_baseline leg use this."""
import ctypes
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(REPO, "oracle", "_ref", "libhrt_ref.so")


def available():
    return os.path.exists(REF_SO)


def load():
    from hermespy_rt_amd import abi
    return abi.bind_c_abi(ctypes.CDLL(REF_SO))
