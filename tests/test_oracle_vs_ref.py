"""The oracle against the real reference run LIVE (oracle/_ref/libhrt_ref.so, built in place
from /root/reference; skipped where that build is absent), on inputs beyond the committed
fixtures: odd ray counts, several TX with velocities (the reference's multi-TX layout quirks
Q9/Q11/Q12), coincident TX/RX, sub-sampled runs."""
import numpy as np
import pytest

from hermespy_rt_amd import abi
from oracle import oracle

from . import configs as K
from .parity import assert_bit_equal

CASES = {
    "c3_777": K.small(K.C3, 777),
    "c3_doppler_3001": K.small(K.C3_DOPPLER, 3001),
    "c4_3tx": dict(K.small(K.C4_DOPPLER, 1234), tx_pos=[[0, -20, 3], [0, 20, 3], [5, 0, 2]],
                   tx_vel=[[3, 1, 0], [0, -2, 1], [1, 1, 1]]),
    "c5_small": dict(K.small(K.C5, 130), num_bounces=2),
    "np_1": K.small(K.C1, 1),
    "np_8": K.small(K.C3, 8),
    "np_9_2tx": K.small(K.C4, 9),
    "coincident": K.small(K.COINCIDENT, 500),
}


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_equals_reference(ref_lib, name):
    c = CASES[name]
    ref = abi.run_compute_paths(ref_lib, *K.args(c))
    got = oracle.compute_paths(*K.args(c))
    for blk in ("los", "scat"):
        for k in ref[blk]:
            assert_bit_equal(got[blk][k], ref[blk][k], "%s.%s" % (blk, k))
    for k in ("los_rays", "los_active", "scat_rays", "scat_active"):
        assert_bit_equal(got[k], ref[k], k)
    # normals the reference left in the scene
    n_ref = np.concatenate(ref["normals"])
    assert_bit_equal(got["extras"]["normals"], n_ref, "normals")


def test_oracle_threads_do_not_change_bits():
    c = K.small(K.C3, 3000)
    a = oracle.compute_paths(*K.args(c), num_threads=1)
    b = oracle.compute_paths(*K.args(c), num_threads=4)
    for blk in ("los", "scat"):
        for k in a[blk]:
            assert_bit_equal(a[blk][k], b[blk][k], k)
    assert_bit_equal(a["scat_rays"], b["scat_rays"], "rays")


def test_oracle_subset_equals_full_restricted():
    """A strided subset run (used for the bounded CPU baseline) writes exactly the full run's
    values on the subset's paths."""
    c = K.small(K.C3, 4000)
    full = oracle.compute_paths(*K.args(c))
    sub = oracle.compute_paths(*K.args(c), subset=(1, 4000, 7))
    sel = np.arange(1, 4000, 7)
    for k in ("a_te_re", "a_tm_im", "tau"):
        assert_bit_equal(sub["scat"][k][..., sel], full["scat"][k][..., sel], k)
        rest = np.setdiff1d(np.arange(4000), sel)
        assert not abi.written(sub["scat"][k][..., rest]).any()
