"""GPU parity of the device-resident API (hrt_trace on torch-owned buffers): compact hit
blocks / record blocks against the oracle's per-(bounce, ray) extras, and ray sharding:
the union of G shards equals the unsharded run bit for bit (rays are independent)."""
import numpy as np
import pytest

from oracle import oracle

from . import configs as K
from .parity import AMP_RTOL, amp_error, assert_bit_equal

pytestmark = pytest.mark.gpu


def _tracer(c, **kw):
    from hermespy_rt_amd.device import Tracer
    return Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"],
                  c["f_ghz"], c["num_paths"], c["num_bounces"], **kw)


CASES = {
    "C1": K.C1,
    "C3_20k": K.small(K.C3, 20000),
    "C4_2tx_9k": K.small(K.C4_DOPPLER, 9000),
    "C5_8x8_4k": K.small(K.C5, 4096),
}


@pytest.mark.parametrize("name", list(CASES))
def test_compact_blocks_vs_oracle(name):
    c = CASES[name]
    tr = _tracer(c)
    tr.trace()
    d = tr.to_dense()
    ref = oracle.compute_paths(*K.args(c))
    ex = ref["extras"]
    nb = c["num_bounces"]
    # live counts per bounce and the algorithmic test count
    assert list(d["counts"][:nb + 1]) == [int(x) for x in ex["live"]]
    assert tr.work()["tests"] == ex["tests"]
    # hit indices: bit-exact
    assert np.array_equal(d["hit_tri"], ex["hit_tri"])
    # incidence angle: double acos rounded to float on both sides -- the device library's and glibc's
    # agree on every float input (tests/exhaustive_incidence.py), so bit for bit
    hit = ex["hit_tri"] != 0xFFFFFFFF
    assert np.array_equal(d["hit_theta"][hit].view(np.uint32), ex["hit_theta"][hit].view(np.uint32))
    # launch directions are the host libm's on both sides
    assert_bit_equal(tr.dirs_host, ex["launch_dirs"], "launch_dirs")
    # records
    from hermespy_rt_amd.abi import written
    s = ref["scat"]
    w = written(s["a_te_re"])
    assert np.array_equal(written(d["a_te_re"]), w)
    assert_bit_equal(d["tau"], s["tau"], "tau")
    assert_bit_equal(d["directions_rx"], s["directions_rx"], "directions_rx")
    for pol in ("te", "tm"):
        e = amp_error(d["a_%s_re" % pol], d["a_%s_im" % pol], s["a_%s_re" % pol], s["a_%s_im" % pol], w)
        assert e <= AMP_RTOL, (pol, e)
    # post-bounce ray state of every hit == the reference's RaysInfo snapshot rows (ntx == 1)
    if len(c["tx_pos"]) == 1:
        npth = c["num_paths"]
        for b in range(nb):
            snap = ref["scat_rays"][(b + 1) * npth:(b + 2) * npth]
            h = hit[b, 0]
            assert_bit_equal(d["state"][b, 0][h][:, :6], snap[h], "rays after bounce %d" % b)
    tr.close()


@pytest.mark.parametrize("world,chunk", [(2, 0), (3, 64), (4, 256)])
def test_shards_union_equals_unsharded(world, chunk):
    c = K.small(K.C3, 10000)
    full = _tracer(c)
    full.trace()
    dfull = full.to_dense()
    merged = None
    total = 0
    for r in range(world):
        tr = _tracer(c, rank=r, world=world, chunk=chunk)
        total += tr.num_local
        tr.trace()
        d = tr.to_dense()
        if merged is None:
            merged = d
        else:
            for k, v in d.items():
                if k == "counts":
                    merged[k] = merged[k] + v
                    continue
                sent = v.view(np.uint32) == (0xFFFFFFFF if k == "hit_tri" else 0x7FC0DEAD)
                mv = merged[k].view(np.uint32)
                # shards are disjoint: nobody overwrites a slot someone else wrote
                msent = mv == (0xFFFFFFFF if k == "hit_tri" else 0x7FC0DEAD)
                assert not np.any(~sent & ~msent)
                mv[~sent] = v.view(np.uint32)[~sent]
        tr.close()
    assert total == c["num_paths"]
    for k in dfull:
        if k == "counts":
            assert list(merged[k][:5]) == list(dfull[k][:5])
        else:
            assert_bit_equal(merged[k], dfull[k], k)
    full.close()


def test_rays_per_shard_limit():
    """A shard may hold at most 2^32 / (HRT_HIT_FIELDS * 4) - 512 rays (32-bit offsets inside a
    block of field arrays): hrt_layout_query refuses more with HRT_E_CAPACITY, and the same
    total split over two shards is accepted."""
    import ctypes as C
    from hermespy_rt_amd import lib as _lib
    from hermespy_rt_amd.device import Tracer
    c = K.small(K.C1, 1000)
    tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                c["num_paths"], c["num_bounces"])
    L = _lib.load()
    lay = _lib.Layout()
    limit = (1 << 32) // (15 * 4) - 512
    ok = _lib.Shard(limit, 0, 1, 0, 1)
    assert L.hrt_layout_query(tr.problem, C.byref(ok), C.byref(lay)) == 0
    assert 15 * lay.cap * 4 < (1 << 32)
    too_big = _lib.Shard(limit + 4096, 0, 1, 0, 1)
    assert L.hrt_layout_query(tr.problem, C.byref(too_big), C.byref(lay)) == -4      # HRT_E_CAPACITY
    assert b"use more shards" in L.hrt_last_error()
    halves = _lib.Shard(limit + 4096, 1, 2, 0, 1)
    assert L.hrt_layout_query(tr.problem, C.byref(halves), C.byref(lay)) == 0
    tr.close()


def test_layout_has_the_wide_packet_queue_only_beyond_1024_triangles(tmp_path):
    """hrt_layout (include/hrt_device.h): tables with fine leaves (more than 1 024 triangles) carry the queue
    of wide packets -- entries and 64 keys each, between the survivor counts (four words per chunk) and the
    trace results --; small tables carry no queue."""
    import ctypes as C
    from hermespy_rt_amd import lib as _lib
    from hermespy_rt_amd.device import Tracer
    from . import scenes_gen as G
    L = _lib.load()
    p = str(tmp_path / "room.hrt")
    T = G.room_with_clutter(p, 100, seed=3, tilt=True)
    assert T > 1024
    big = Tracer(p, [[5, 3, 1.5], [-8, -4, 2.0]], [[-10, 5, 6.0]], [[0, 0, 0]] * 2, [[0, 0, 0]], 3.5, 100000, 2)
    c = K.small(K.C3, 100000)
    small = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"], c["num_paths"],
                   c["num_bounces"])
    lay = _lib.Layout()
    sh = _lib.Shard(100000, 0, 1, 0, 2)
    assert L.hrt_layout_query(big.problem, C.byref(sh), C.byref(lay)) == 0
    traces = (lay.cap // 64) * (2 + 1)
    assert lay.wide_cap == max(1024, traces // 2)
    assert lay.off_chunk_cnt < lay.off_wide_q < lay.off_wide_key < lay.off_res < lay.total_bytes
    assert lay.off_wide_q - lay.off_chunk_cnt >= (lay.cap // 256) * 16
    assert lay.off_wide_q + lay.wide_cap * 8 <= lay.off_wide_key
    assert lay.off_wide_key + lay.wide_cap * 64 * 8 <= lay.off_res
    sh = _lib.Shard(100000, 0, 1, 0, 4)
    assert L.hrt_layout_query(small.problem, C.byref(sh), C.byref(lay)) == 0
    assert lay.wide_cap == 0 and lay.off_wide_q == 0 and lay.off_wide_key == 0
    big.close()
    small.close()
