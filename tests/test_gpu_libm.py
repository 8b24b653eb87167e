"""The device's float libm restatements (csrc/hrt_libm.h) evaluated ON THE GPU against the
host libm the reference calls: bit-exact on a dense sample of the domain the tracer produces
(the exhaustive host-side pin is oracle/libm_probe --full)."""
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


def test_incidence_angle_double_acos(product_lib):
    """acos in double (device library vs glibc) rounded to float: equal except, rarely, by one
    float ulp (documented residual)."""
    x = _floats(0.0, 1.0000001, 101)
    got = _device_eval(product_lib, 4, x)
    ref = oracle.host_libm("incidence_angle", x)
    diff = got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got), nan)
    assert np.abs(diff[~nan]).max() <= 1
    frac = float((diff[~nan] != 0).mean())
    print("incidence angle: %d inputs, %.3g differ by 1 ulp" % (x.size, frac))
    assert frac < 1e-6
