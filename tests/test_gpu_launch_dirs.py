"""Launch directions generated on the DEVICE (hrt_launch_dirs_device: device double math + host
patch of the values whose float rounding could depend on the math library) are bit-identical to
the host-libm directions -- the reference's Fibonacci sphere -- for the ray counts of every
BASELINE config, sharded or not; and a trace from them equals a trace from host directions."""
import ctypes as C

import numpy as np
import pytest

from . import configs as K

pytestmark = pytest.mark.gpu


def _host_dirs(L, s):
    from hermespy_rt_amd import lib
    n = int(L.hrt_shard_num_local(C.byref(s)))
    out = np.empty((n, 3), np.float32)
    lib.check(L.hrt_launch_dirs_host(C.byref(s), out.ctypes.data_as(C.POINTER(C.c_float)), 0))
    return out


@pytest.mark.parametrize("n,rank,count", [(10000, 0, 1), (1000000, 0, 1), (4000000, 0, 1), (8000000, 0, 1),
                                         (64000000, 3, 8), (16000000, 1, 4), (12345, 2, 3)])
def test_device_dirs_equal_host_dirs(product_lib, n, rank, count):
    import torch
    from hermespy_rt_amd import lib
    s = lib.Shard(n, rank, count, 0, 1)
    host = _host_dirs(product_lib, s)
    d = torch.empty((len(host), 3), dtype=torch.float32, device="cuda:0")
    n_p = C.c_uint64(0)
    lib.check(product_lib.hrt_launch_dirs_device(C.byref(s), C.c_void_p(d.data_ptr()), 0, None, C.byref(n_p)))
    dev = d.cpu().numpy()
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32)), \
        "%d of %d values differ" % ((dev.view(np.uint32) != host.view(np.uint32)).sum(), dev.size)
    print("N=%d shard %d/%d: %d rays, %d patched by the host" % (n, rank, count, len(host), n_p.value))
    assert n_p.value < 1e-4 * len(host) + 16


def test_trace_from_device_dirs_equals_trace_from_host_dirs():
    from hermespy_rt_amd.device import Tracer
    c = K.small(K.C3, 200000)
    outs = []
    for dd in (False, True):
        tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                    c["num_paths"], c["num_bounces"], device_dirs=dd)
        tr.trace()
        outs.append(tr.to_dense())
        tr.close()
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                              b.view(np.uint32) if b.dtype == np.float32 else b), k


@pytest.mark.parametrize("n,rank,count", [(10000, 0, 1), (4000000, 0, 1), (1000003, 1, 3), (70000, 2, 4)])
def test_device_launch_order_is_a_coherent_permutation(n, rank, count):
    """hrt_launch_order_device: a permutation of the shard's local rays whose consecutive runs of 64
    are narrow packets (mean half-angle within 1.6x of the host order's), and a trace with it gives
    the same result as with the host order."""
    import ctypes as C
    import torch
    from hermespy_rt_amd import lib as L_
    L = L_.load()
    sh = L_.Shard(n, rank, count, 0, 2)
    nl = int(L.hrt_shard_num_local(C.byref(sh)))
    order = torch.empty(nl, dtype=torch.int32, device="cuda")
    L_.check(L.hrt_launch_order_device(C.byref(sh), C.c_void_p(order.data_ptr()), 0, None))
    o = order.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(o), np.arange(nl))
    dirs = np.empty((nl, 3), np.float32)
    L_.check(L.hrt_launch_dirs_host(C.byref(sh), dirs.ctypes.data_as(C.POINTER(C.c_float)), 0))
    oh = np.empty(nl, np.uint32)
    L_.check(L.hrt_launch_order_host(C.byref(sh), dirs.ctypes.data_as(C.POINTER(C.c_float)), oh.ctypes.data_as(C.POINTER(C.c_uint32))))

    def mean_half_angle(perm):
        m = (nl // 64) * 64
        d = dirs[perm[:m]].reshape(-1, 64, 3).astype(np.float64)
        ax = d.sum(1)
        ax /= np.linalg.norm(ax, axis=1, keepdims=True)
        c = np.einsum("pkc,pc->pk", d, ax).min(1)
        return float(np.degrees(np.arccos(np.clip(c, -1, 1))).mean())

    a_dev, a_host = mean_half_angle(o), mean_half_angle(oh.astype(np.int64))
    assert a_dev < 1.6 * a_host + 0.05, (a_dev, a_host)


def test_trace_with_device_tables_equals_host_tables():
    from hermespy_rt_amd.device import Tracer
    c = K.small(K.C3, 300000)
    outs = []
    for dev in (False, True):
        tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                    c["num_paths"], c["num_bounces"], device_dirs=dev)
        tr.trace()
        outs.append(tr.to_dense())
        tr.close()
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                              b.view(np.uint32) if b.dtype == np.float32 else b), k
