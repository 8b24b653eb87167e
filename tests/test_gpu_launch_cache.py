"""With HRT_HOST_LAUNCH=1 (launch tables from the host generators instead of the device, which is
the default) compute_paths keeps the launch-direction table and launch order of the last num_rays
between calls (csrc/host/compute_paths.c): a call served from the cache, a call after the cache
was replaced by another ray count, and a call after hrt_cache_clear() must all equal the oracle --
and so must the default, device-generated tables on the same sequence of calls."""
import pytest

from hermespy_rt_amd import abi
from oracle import oracle

from . import configs as K
from .parity import compare_dense

pytestmark = pytest.mark.gpu


def _check(product_lib, c):
    got = abi.run_compute_paths(product_lib, *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), st


@pytest.mark.parametrize("host_launch", ["1", "0"])
def test_repeated_calls_through_the_launch_cache(product_lib, monkeypatch, host_launch):
    monkeypatch.setenv("HRT_HOST_LAUNCH", host_launch)
    a = K.small(K.C3, 20000)
    b = K.small(K.C4_DOPPLER, 7001)      # other scene, other endpoints, other ray count
    a2 = dict(a, rx_pos=[[-12, 1.0, 1.5], [8, -1.5, 2.0], [30, 0, 1.5], [45, 2, 3]], f_ghz=28.0)
    product_lib.hrt_cache_clear()
    _check(product_lib, a)      # miss: fills the cache
    _check(product_lib, a)      # hit
    _check(product_lib, a2)     # hit with other endpoints / frequency: only num_rays is the key
    _check(product_lib, b)      # other num_rays: entry replaced
    _check(product_lib, a)      # miss again
    product_lib.hrt_cache_clear()
    _check(product_lib, a)
