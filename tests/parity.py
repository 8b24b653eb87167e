"""Comparison helpers shared by the parity tests.

Tolerances (north_star): bit-exact for everything that is integer/index/geometry work --
written-slot sets, hit indices, counts, rays, active masks, delays (tau), directions -- and
(freq_shift included, down to the sign of zero: the reference's `+= 0` quirk Q10 is replayed) and
1e-5 relative for the complex amplitudes, which go through sin/cos/exp/acos (device libm vs
glibc).  'Relative' for a complex amplitude is taken on the complex value: |a - a_ref| <=
1e-5 * |a_ref| (+ a tiny absolute floor for exact zeros), the standard for phasors; the
real/imaginary parts alone can be arbitrarily close to zero.
"""
import numpy as np

AMP_RTOL = 1e-5


def bits(a):
    return a.view(np.uint8 if a.dtype == np.uint8 else np.uint32)


def assert_bit_equal(a, b, name):
    x, y = bits(np.ascontiguousarray(a)), bits(np.ascontiguousarray(b))
    assert x.shape == y.shape, "%s: shape %s vs %s" % (name, x.shape, y.shape)
    if not np.array_equal(x, y):
        bad = np.flatnonzero(x.ravel() != y.ravel())
        raise AssertionError("%s: %d of %d elements differ, first at %s: %r vs %r" % (
            name, bad.size, x.size, bad[:4], a.ravel()[bad[:4]], b.ravel()[bad[:4]]))


def assert_same_zero_aware(a, b, name):
    """bit-equal except that -0.0 == +0.0 (freq_shift of zero-velocity scenes)."""
    x, y = bits(np.ascontiguousarray(a)).ravel(), bits(np.ascontiguousarray(b)).ravel()
    ne = x != y
    if ne.any():
        both_zero = ((x | y) & 0x7FFFFFFF) == 0
        bad = np.flatnonzero(ne & ~both_zero)
        assert bad.size == 0, "%s: %d elements differ, first at %s: %r vs %r" % (
            name, bad.size, bad[:4], a.ravel()[bad[:4]], b.ravel()[bad[:4]])


def amp_error(re, im, re_ref, im_ref, sentinel_mask=None):
    """max over written slots of |a - a_ref| / |a_ref|, NaNs must coincide."""
    a = re.astype(np.float64) + 1j * im.astype(np.float64)
    r = re_ref.astype(np.float64) + 1j * im_ref.astype(np.float64)
    if sentinel_mask is not None:
        a, r = a[sentinel_mask], r[sentinel_mask]
    nan_a, nan_r = np.isnan(a), np.isnan(r)
    assert np.array_equal(nan_a, nan_r), "NaN pattern differs"
    ok = ~nan_r
    a, r = a[ok], r[ok]
    if a.size == 0:
        return 0.0
    mag = np.abs(r)
    err = np.abs(a - r)
    zero = mag == 0
    assert np.all(err[zero] == 0), "nonzero amplitude where the reference has an exact zero"
    return float(np.max(err[~zero] / mag[~zero])) if (~zero).any() else 0.0


def compare_dense(got, ref, amp_rtol=AMP_RTOL, check_rays=True, exact_freq_shift=True):
    """got/ref: dicts as returned by abi.run_compute_paths / oracle.compute_paths."""
    from hermespy_rt_amd.abi import written
    stats = {}
    for blk in ("los", "scat"):
        g, r = got[blk], ref[blk]
        # same written-slot sets in every array
        for k in r:
            assert np.array_equal(written(g[k]), written(r[k])), "%s.%s: written-slot set differs" % (blk, k)
        for k in ("tau", "directions_rx", "directions_tx"):
            assert_bit_equal(g[k], r[k], "%s.%s" % (blk, k))
        if exact_freq_shift:
            assert_bit_equal(g["freq_shift"], r["freq_shift"], blk + ".freq_shift")
        else:
            assert_same_zero_aware(g["freq_shift"], r["freq_shift"], blk + ".freq_shift")
        w = written(r["a_te_re"])
        for pol in ("te", "tm"):
            e = amp_error(g["a_%s_re" % pol], g["a_%s_im" % pol], r["a_%s_re" % pol], r["a_%s_im" % pol], w)
            stats["%s.a_%s" % (blk, pol)] = e
            # with the bit-exact device libm, amplitudes are normally bit-identical as well
            nbits = 0
            for part in ("re", "im"):
                x, y = bits(g["a_%s_%s" % (pol, part)]), bits(r["a_%s_%s" % (pol, part)])
                nbits += int(((x != y) & ~(np.isnan(g["a_%s_%s" % (pol, part)]) & np.isnan(r["a_%s_%s" % (pol, part)]))).sum())
            stats["%s.a_%s.not_bit_equal" % (blk, pol)] = nbits
            assert e <= amp_rtol, "%s.a_%s: relative error %.3g > %.1g" % (blk, pol, e, amp_rtol)
    if check_rays:
        for k in ("los_rays", "los_active", "scat_rays", "scat_active"):
            assert_bit_equal(got[k], ref[k], k)
    return stats
