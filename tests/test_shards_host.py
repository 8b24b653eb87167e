"""Host-side shard arithmetic and launch directions of the product library (no GPU needed):
round-robin shards partition the path range exactly, and hrt_launch_dirs_host reproduces the
reference's Fibonacci sphere (src/compute_paths.c:443-451) bit for bit, sharded or not,
with any thread count."""
import ctypes as C

import numpy as np
import pytest

from hermespy_rt_amd import lib
from oracle import oracle

from . import configs as K


def _shard(n, r, g, chunk=0, nb=1):
    return lib.Shard(n, r, g, chunk, nb)


def _dirs(L, s, threads=0):
    n = int(L.hrt_shard_num_local(C.byref(s)))
    out = np.empty((n, 3), np.float32)
    lib.check(L.hrt_launch_dirs_host(C.byref(s), out.ctypes.data_as(C.POINTER(C.c_float)), threads))
    return out


@pytest.mark.parametrize("n,g,chunk", [(1, 1, 0), (10000, 1, 0), (10000, 3, 64), (4096, 4, 0),
                                       (4097, 4, 0), (100000, 8, 4096), (12345, 7, 128), (63, 2, 64)])
def test_shards_partition_the_path_range(product_lib, n, g, chunk):
    seen = np.zeros(n, np.int32)
    total = 0
    for r in range(g):
        s = _shard(n, r, g, chunk)
        nl = int(product_lib.hrt_shard_num_local(C.byref(s)))
        total += nl
        p = np.array([product_lib.hrt_shard_global_path(C.byref(s), i) for i in range(0, nl, max(1, nl // 500))] +
                     ([product_lib.hrt_shard_global_path(C.byref(s), nl - 1)] if nl else []), dtype=np.int64)
        assert (p < n).all() and (np.diff(p[:-1]) > 0).all()
        # exhaustive for the small cases
        if n <= 20000:
            allp = np.array([product_lib.hrt_shard_global_path(C.byref(s), i) for i in range(nl)], dtype=np.int64)
            seen[allp] += 1
    assert total == n
    if n <= 20000:
        assert (seen == 1).all()


@pytest.mark.parametrize("n", [1, 7, 10000, 100003])
def test_launch_dirs_equal_reference_formula(product_lib, n):
    ref = oracle.compute_paths(*K.args(K.small(K.C1, n)))["extras"]["launch_dirs"]
    for threads in (1, 3, 0):
        got = _dirs(product_lib, _shard(n, 0, 1), threads)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_sharded_launch_dirs_use_global_index(product_lib):
    n, g = 50000, 3
    ref = oracle.compute_paths(*K.args(K.small(K.C1, n)))["extras"]["launch_dirs"]
    for r in range(g):
        s = _shard(n, r, g, 256)
        got = _dirs(product_lib, s)
        p = np.array([product_lib.hrt_shard_global_path(C.byref(s), i) for i in range(len(got))])
        assert np.array_equal(got.view(np.uint32), ref[p].view(np.uint32))


def test_golden_launch_anchor(product_lib):
    """N = 10000, path 0 and path 4999."""
    d = _dirs(product_lib, _shard(10000, 0, 1))
    assert np.array_equal(np.asarray(d[0], np.float32), np.asarray([0.00512505323, -0.0131816929, 0.999899983], np.float32))
    assert np.array_equal(np.asarray(d[4999], np.float32), np.asarray([-0.641443908, 0.767169952, 9.997288e-05], np.float32))


def test_bad_shards_are_rejected(product_lib):
    out = np.empty((4, 3), np.float32)
    bad = lib.Shard(0, 0, 1, 0, 1)
    assert product_lib.hrt_launch_dirs_host(C.byref(bad), out.ctypes.data_as(C.POINTER(C.c_float)), 1) != 0
    bad = lib.Shard(10, 2, 2, 0, 1)
    assert product_lib.hrt_shard_num_local(C.byref(bad)) == 0
