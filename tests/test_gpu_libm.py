"""The device's float libm restatements (csrc/hrt_libm.h) evaluated ON THE GPU against the
host libm the reference calls, on EVERY float of the domains the tracer can reach -- 1.56e10
comparisons for the float functions (what oracle/libm_probe --full pins for the same header compiled
for the host), 2.13e9 for the double-precision incidence angle.  About a minute on the GPU box."""
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


def test_device_libm_sample_beyond_the_domains(product_lib):
    """Outside the exhaustively compared domains the tracer never evaluates these functions; a
    sparse sample of the rest of the float line still has to agree on NaNs and signs."""
    x = np.array([np.inf, -np.inf, np.nan, 1.5, -1.5], np.float32)
    got = _device_eval(product_lib, oracle.LIBM_FN["acosf"], x)
    ref = oracle.host_libm("acosf", x)
    assert np.array_equal(np.isnan(got), np.isnan(ref))


@pytest.mark.parametrize("name", ["sinf", "cosf", "sincosf.sin", "sincosf.cos", "cosf_nb", "expf", "acosf"])
def test_device_libm_exhaustive(product_lib, name):
    """The float libm restatements (csrc/hrt_libm.h: glibc 2.35 / Arm Optimized Routines sinf, cosf,
    expf, fdlibm acosf, and the fused forms the shade kernel calls) evaluated ON THE DEVICE against
    the host libm the reference calls (src/compute_paths.c:310, 325, 365-383), on EVERY float of the
    domain the tracer can reach: |x| < 120 (2.24e9 inputs per function), |x| < 88 for expf,
    |x| <= 1 for acosf -- 1.56e10 comparisons in all, bit for bit (tests/exhaustive_incidence.py)."""
    from tests.exhaustive_incidence import mismatches, LIBM
    code, host, end = LIBM[name]
    bad = mismatches(product_lib, hi=end, code=code, host=host)
    assert not bad, "%s: %d inputs differ, e.g. %s" % (
        name, len(bad), ["x=0x%08x dev=0x%08x host=0x%08x" % t for t in bad[:5]])


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
