"""The device's float libm restatements (csrc/hrt_libm.h) evaluated ON THE GPU against the
host libm the reference calls: bit-exact on a dense sample of the domain the tracer produces
(the exhaustive host-side pin is oracle/libm_probe --full), and the double-precision incidence
angle on EVERY float input."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


def _device_eval(L, fn, x):
    from hermespy_rt_amd import lib
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    f32p = C.POINTER(C.c_float)
    lib.check(L.hrt_selftest_math(0, fn, x.ctypes.data_as(f32p), out.ctypes.data_as(f32p), x.size),
              "hrt_selftest_math")
    return out


def _floats(lo, hi, stride, both_signs=True):
    a, b = np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32)
    u = np.arange(int(a), int(b), stride, dtype=np.uint64).astype(np.uint32)
    x = u.view(np.float32)
    return np.concatenate([x, -x]) if both_signs else x


DOMAINS = {
    "sinf": (0.0, 120.0, 257),
    "cosf": (0.0, 120.0, 257),
    "expf": (0.0, 88.0, 257),
    "acosf": (0.0, 1.0000001, 251),
}


# the branch-free pair / cosine the shading kernel actually calls: (selftest code, host function)
FUSED = {"sincosf.sin": (5, "sinf"), "sincosf.cos": (6, "cosf"), "cosf_nb": (7, "cosf")}


@pytest.mark.parametrize("name", list(FUSED))
def test_device_fused_sincos_bit_exact(product_lib, name):
    code, host = FUSED[name]
    x = _floats(0.0, 120.0, 131)
    got = _device_eval(product_lib, code, x)
    ref = oracle.host_libm(host, x)
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), "%s: %d of %d differ, e.g. x=%r" % (name, (~same).sum(), x.size, x[~same][:3])


@pytest.mark.parametrize("name", list(DOMAINS))
def test_device_libm_bit_exact(product_lib, name):
    lo, hi, stride = DOMAINS[name]
    x = _floats(lo, hi, stride)
    got = _device_eval(product_lib, oracle.LIBM_FN[name], x)
    ref = oracle.host_libm(name, x)
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), "%s: %d of %d differ, e.g. x=%r" % (name, (~same).sum(), x.size, x[~same][:3])


def test_incidence_angle_exhaustive(product_lib):
    """acos in double (device library vs glibc) rounded to float and folded (src/compute_paths.c:
    281-283): a function of ONE float, so compared on EVERY float with |x| <= 1 -- 2.13e9 inputs,
    bit for bit (tests/exhaustive_incidence.py) -- and on a sample beyond 1 (NaN on both sides)."""
    from tests.exhaustive_incidence import mismatches, ONE
    bad = mismatches(product_lib)
    assert not bad, "%d inputs differ, e.g. %s" % (len(bad), ["x=0x%08x dev=0x%08x host=0x%08x" % t for t in bad[:5]])
    x = _floats(1.0000001, 3.0e38, 100003)
    got = _device_eval(product_lib, 4, x)
    ref = oracle.host_libm("incidence_angle", x)
    assert np.isnan(got).all() and np.isnan(ref).all()
    assert ONE == 0x3F800000
