"""Multi-GPU inside the C ABI (the reference's contract is one blocking call with
caller-owned outputs, inc/compute_paths.h:59-74): HRT_DEVICES deals the round-robin batches of the
launch set to one host thread per device, each copying its records into the caller's dense arrays.
On the one-GPU test box the devices are LOGICAL (the same HIP device several times): the code path
-- threads, per-device problem copies, workspaces, streams, staging, disjoint dense writes -- is the
multi-GPU one.  Every output array must equal the single-device result and the oracle, bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys
sys.path.insert(0, %(repo)r)
import numpy as np
from hermespy_rt_amd import abi, lib
from oracle import oracle
from tests import configs as K
from tests.parity import compare_dense
L = lib.load()
# (with RaysInfo too: every batch writes the snapshots of its own paths, on whatever device it ran; odd ray
# counts with two TXs put the Q12 bit quirk on a byte that two batches share)
for c, rays in ((K.small(K.C3, 70000), False), (K.small(K.C4_DOPPLER, 30001), False), (K.small(K.C5, 9000), False),
                (K.small(K.C3_DOPPLER, 20000), True), (K.small(K.C4_DOPPLER, 20483), True), (K.small(K.C5, 8195), True)):
    st = lib.Stats()
    got = abi.run_compute_paths(L, *K.args(c), with_rays=rays, stats=st)
    ref = oracle.compute_paths(*K.args(c))
    if not rays:   # without RaysInfo the ray arrays keep their sentinels in `got`
        for k in ("los_rays", "los_active", "scat_rays", "scat_active"):
            ref[k] = got[k]
    s = compare_dense(got, ref)
    assert all(v == 0 for v in s.values()), s
    want = min(%(want)d, (c["num_paths"] + 4095) // 4096)   # a batch is at least one 4096-path granule
    assert st.num_devices == want, (st.num_devices, want, st.num_batches)
    # (batches are a multiple of the devices unless the launch set has too few 4096-path granules for that)
    assert st.num_batches %% st.num_devices == 0 or st.num_batches == (c["num_paths"] + 4095) // 4096
    # per-device phase times and batch counts (what shows the balance on a node with several GPUs)
    nd = st.num_devices
    assert sum(int(st.dev_batches[d]) for d in range(nd)) == st.num_batches
    assert all(int(st.dev_id[d]) == 0 and st.dev_t_device_s[d] > 0 and st.dev_t_readback_s[d] > 0 for d in range(nd))
    assert abs(max(st.dev_t_device_s[d] for d in range(nd)) - st.t_device_s) < 1e-12
    assert [int(st.live[i]) for i in range(c["num_bounces"] + 1)] == [int(x) for x in ref["extras"]["live"]]
print("MULTI_OK")
"""


@pytest.mark.parametrize("devices,want", [("0", 1), ("0,0", 2), ("0,0,0,0", 4), ("0,0,0", 3)])
def test_logical_devices_equal_the_oracle(devices, want):
    env = dict(os.environ, HRT_DEVICES=devices)
    p = subprocess.run([sys.executable, "-c", CODE % dict(repo=REPO, want=want)], env=env, capture_output=True, text=True)
    assert p.returncode == 0 and "MULTI_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


def test_warm_calls_reuse_pooled_buffers(product_lib):
    """second and third call (other endpoints, then a smaller launch set) through the buffer pool;
    then hrt_cache_clear() and once more"""
    from hermespy_rt_amd import abi
    from oracle import oracle
    from . import configs as K
    from .parity import compare_dense
    a = K.small(K.C3, 40000)
    a2 = dict(a, rx_pos=[[-12, 1.0, 1.5], [8, -1.5, 2.0], [30, 0, 1.5], [45, 2, 3]])
    b = K.small(K.C4_DOPPLER, 9001)
    for c in (a, a2, b, a):
        got = abi.run_compute_paths(product_lib, *K.args(c))
        st = compare_dense(got, oracle.compute_paths(*K.args(c)))
        assert all(v == 0 for v in st.values()), st
    product_lib.hrt_cache_clear()
    got = abi.run_compute_paths(product_lib, *K.args(b))
    st = compare_dense(got, oracle.compute_paths(*K.args(b)))
    assert all(v == 0 for v in st.values()), st


def test_many_calls_reuse_the_parked_helper_threads(product_lib):
    """The dense writer's helper threads are parked between its loops and between calls (per calling
    thread): many calls of changing size and thread count, dense and list, must neither hang nor drift
    from the oracle; hrt_cache_clear() in between ends the helpers, and the next call starts new ones."""
    import os
    from hermespy_rt_amd import abi
    from oracle import oracle
    from . import configs as K
    from .parity import compare_dense
    big = K.small(K.C3, 300000)      # several 65 536-record ranges per block: the helpers are used
    small = K.small(K.C1, 500)       # one range: the calling thread alone
    ref_big = oracle.compute_paths(*K.args(big))
    ref_small = oracle.compute_paths(*K.args(small))
    old = os.environ.get("HRT_HOST_THREADS")
    try:
        for k in range(12):
            os.environ["HRT_HOST_THREADS"] = str((4, 16, 7, 32)[k % 4])
            c, ref = (big, ref_big) if k % 3 != 2 else (small, ref_small)
            st = compare_dense(abi.run_compute_paths(product_lib, *K.args(c)), ref)
            assert all(v == 0 for v in st.values()), (k, st)
            if k % 4 == 1:
                pl = abi.run_compute_paths_list(product_lib, *K.args(big))
                assert pl["rx"].size > 0
            if k % 5 == 4:
                product_lib.hrt_cache_clear()
    finally:
        if old is None:
            os.environ.pop("HRT_HOST_THREADS", None)
        else:
            os.environ["HRT_HOST_THREADS"] = old


def test_several_batches_on_one_device(product_lib, monkeypatch):
    """A workspace budget that the launch set does not fit (HRT_WORKSPACE_BYTES): ONE worker runs
    several round-robin batches one after the other through the same buffers -- dense and list -- and
    the result is still the oracle's, bit for bit."""
    from hermespy_rt_amd import abi, lib
    from oracle import oracle
    from . import configs as K
    from .parity import compare_dense
    monkeypatch.setenv("HRT_WORKSPACE_BYTES", str(40 << 20))
    for c, rays in ((K.small(K.C3, 150000), False), (K.small(K.C4_DOPPLER, 60001), False), (K.small(K.C3_DOPPLER, 150000), True),
                    (K.small(K.C4_DOPPLER, 60001), True)):
        st = lib.Stats()
        got = abi.run_compute_paths(product_lib, *K.args(c), with_rays=rays, stats=st)
        ref = oracle.compute_paths(*K.args(c))
        for k in ("los_rays", "los_active", "scat_rays", "scat_active"):
            if not rays:
                ref[k] = got[k]
        s = compare_dense(got, ref)
        assert all(v == 0 for v in s.values()), s
        assert st.num_batches > 1 and st.num_devices == 1, (st.num_batches, st.num_devices)
        pl = abi.run_compute_paths_list(product_lib, *K.args(c))
        w = abi.written(ref["scat"]["directions_rx"][..., 0])
        assert pl["rx"].size == int(w.sum())
    product_lib.hrt_cache_clear()
