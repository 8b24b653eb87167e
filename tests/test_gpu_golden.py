"""GPU parity against the committed golden vectors of the REAL reference (no oracle in the
loop): libhermespy_rt_amd.so's compute_paths() through the reference callers' harness,
every output array against tests/golden/*.npz."""
import os

import numpy as np
import pytest

from hermespy_rt_amd import abi

from . import configs as K
from .golden.make_golden import SMALL
from .parity import AMP_RTOL, amp_error, assert_same_zero_aware

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", list(SMALL))
def test_product_matches_reference_golden(product_lib, name):
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    r = abi.run_compute_paths(product_lib, *K.args(SMALL[name]))
    assert not abi.written(r["scat"]["directions_tx"]).any()          # quirk Q1
    f32 = lambda k: gold[k].view(np.float32)
    for blk in ("los", "scat"):
        for k in ("tau", "directions_rx") + (("directions_tx",) if blk == "los" else ()):
            assert np.array_equal(r[blk][k].view(np.uint32), gold["%s.%s" % (blk, k)]), (blk, k)
        # exact, down to the sign of zero (the reference's `+= 0` quirk Q10 is replayed in its order)
        assert np.array_equal(r[blk]["freq_shift"].view(np.uint32), gold[blk + ".freq_shift"]), blk + ".freq_shift"
        w = gold[blk + ".a_te_re"] != abi.SENTINEL_U32
        for pol in ("te", "tm"):
            for part in ("re", "im"):
                key = "%s.a_%s_%s" % (blk, pol, part)
                assert np.array_equal(abi.written(r[blk]["a_%s_%s" % (pol, part)]), w), key
            e = amp_error(r[blk]["a_%s_re" % pol], r[blk]["a_%s_im" % pol],
                          f32("%s.a_%s_re" % (blk, pol)), f32("%s.a_%s_im" % (blk, pol)), w)
            assert e <= AMP_RTOL, (blk, pol, e)
    for k in ("los_rays", "scat_rays"):
        assert np.array_equal(r[k].view(np.uint32), gold[k]), k
    for k in ("los_active", "scat_active"):
        assert np.array_equal(r[k], gold[k]), k
