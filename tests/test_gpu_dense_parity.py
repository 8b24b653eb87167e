"""GPU parity through the drop-in C ABI: libhermespy_rt_amd.so's compute_paths() (HIP path)
against the oracle on the same inputs, with the harness the reference's own callers use
(sentinel-prefilled caller buffers, so "not written" is part of the comparison)."""
import numpy as np
import pytest

from hermespy_rt_amd import abi
from oracle import oracle

from . import configs as K
from .parity import compare_dense

pytestmark = pytest.mark.gpu

CASES = {
    "C1_box_10k": K.C1,
    "C2_reflector_20k": K.small(K.C2, 20000),
    "C3_canyon_20k": K.small(K.C3, 20000),
    "C3_doppler_5k": K.small(K.C3_DOPPLER, 5000),
    "C4_2cars_2tx_20k": K.small(K.C4, 20000),
    "C4_doppler_odd_np": dict(K.small(K.C4_DOPPLER, 5001), num_bounces=3),
    "C5_8x8_4k": K.small(K.C5, 4096),
    "test_py": K.TEST_PY,
    "coincident_tx_rx": K.COINCIDENT,
    "tiny_np_7": K.small(K.C3, 7),
    "np_64": K.small(K.C1, 64),
}


@pytest.mark.parametrize("name", list(CASES))
def test_dense_parity(product_lib, name):
    c = CASES[name]
    got = abi.run_compute_paths(product_lib, *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    stats = compare_dense(got, ref)
    print(name, stats)


@pytest.mark.parametrize("name", ["C3_DOPPLER_8k_rays", "C5_600", "C4_DOPPLER_3001"])
def test_interleaved_amplitudes_entry_point(product_lib, name):
    """hrt_compute_paths_interleaved (amplitudes as re/im pairs, what a complex64 array is; the
    reference's ChannelInfo has four planes, inc/compute_paths.h:13-23): every output equal to
    hrt_compute_paths_ex, bit for bit, the untouched slots included -- with RaysInfo too."""
    from hermespy_rt_amd import abi
    from .parity import compare_dense
    if name == "C3_DOPPLER_8k_rays":
        c, rays = K.small(K.C3_DOPPLER, 8000), True
    elif name == "C5_600":
        c, rays = K.small(K.C5, 600), False
    else:
        c, rays = K.small(K.C4_DOPPLER, 3001), False
    a = abi.run_compute_paths(product_lib, *K.args(c), with_rays=rays, interleaved=True)
    b = abi.run_compute_paths(product_lib, *K.args(c), with_rays=rays)
    for blk in ("los", "scat"):
        for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im"):
            a[blk][k] = np.ascontiguousarray(a[blk][k])
    st = compare_dense(a, b)
    assert all(v == 0 for v in st.values()), st
