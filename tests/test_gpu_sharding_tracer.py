"""The N > 1 path with the REAL tracer (not the FakeResult of tests/test_sharding_gloo.py): two
ranks share the one GPU of the test box and talk over gloo -- the rehearsal mode of bench.py --
each traces its round-robin ray shard (hermespy_rt_amd.device.Tracer(rank, world)), the packed
records are gathered to rank 0 (hermespy_rt_amd.sharding.RecordGather: all_gather of counts,
pack, variable-size send/recv), and rank 0 rebuilds the dense arrays from the gathered exports
ALONE and compares every written slot with the oracle.  Over RCCL the only difference is the
transport (device tensors instead of host staging)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hermespy_rt_amd import sharding
        from hermespy_rt_amd.abi import written
        from hermespy_rt_amd.device import Tracer
        from oracle import oracle
        from tests import configs as K
        c = K.small(K.C3_DOPPLER, 50000)
        tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                    c["num_paths"], c["num_bounces"], rank=rank, world=world)
        tr.trace()
        g = sharding.RecordGather(tr, dst=0)
        for _ in range(2):                       # buffers are reused across steps
            exports = g.run()
        g2 = sharding.RecordGather(tr, dst=0, unblocked_only=True)   # (checked below)
        exports2 = g2.run()
        if rank != 0:
            assert exports is None and exports2 is None
            return
        ref = oracle.compute_paths(*K.args(c))
        nrx, ntx, nb, npth = tr.nrx, tr.ntx, tr.nb, tr.num_paths
        ch = 4096
        seen = np.zeros((nrx, ntx, nb, npth), bool)
        ok = True
        for r in range(world):
            cnt = g.counts_all[r]
            n_loc = int(tr.L.hrt_shard_num_local(__import__("ctypes").byref(
                __import__("hermespy_rt_amd.lib", fromlist=["Shard"]).Shard(npth, r, world, 0, nb))))
            for b, v in enumerate(sharding.unpack_export(exports[r].cpu(), cnt, nb, nrx)):
                h = int(cnt[b + 1])
                if not h:
                    continue
                ray = v["hit"][0].numpy().astype(np.int64) & 0xFFFFFFFF
                tx = ray // n_loc
                i = ray - tx * n_loc
                p = ((i // ch) * world + r) * ch + i % ch           # global path of the sender's local ray
                fs0 = v["hit"][3].numpy().view(np.float32)
                rec = v["rec"].numpy().view(np.float32)             # [nrx, 9, h]
                bits = v["mask"].numpy().view(np.uint64)
                for rx in range(nrx):
                    ub = ((bits[rx][np.arange(h) // 64] >> (np.arange(h) % 64).astype(np.uint64)) & np.uint64(1)).astype(bool)
                    seen[rx, tx, b, p] = True
                    for k, name in enumerate(("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau")):
                        ok &= np.array_equal(rec[rx, k].view(np.uint32), ref["scat"][name][rx, tx, b, p].view(np.uint32))
                    d = ref["scat"]["directions_rx"][rx, tx[ub], b, p[ub]]
                    ok &= np.array_equal(rec[rx, 5:8][:, ub].T.view(np.uint32), d.view(np.uint32))
                    fs = (fs0 - rec[rx, 8])[ub]                     # one TX: the dense array's value
                    e = ref["scat"]["freq_shift"][rx, tx[ub], b, p[ub]]
                    same = (fs.view(np.uint32) == e.view(np.uint32)) | ((fs == 0) & (e == 0))   # up to the sign of zero (Q10)
                    ok &= bool(same.all())
        ok &= np.array_equal(seen, written(ref["scat"]["a_te_re"]))   # every record of the reference, once
        q.put(bool(ok))
        # ---- the same gather with the unblocked records only (HRT_EXPORT_UNBLOCKED: compacted on the device by
        # the C entry): every non-zero record of the reference, once, and nothing else ----
        exports = exports2
        ok = True
        seen = np.zeros((nrx, ntx, nb, npth), bool)
        for r in range(world):
            cnt, ub_cnt = g2.counts_all[r], g2.meta_all[r][nb + 2:]
            n_loc = int(tr.L.hrt_shard_num_local(__import__("ctypes").byref(
                __import__("hermespy_rt_amd.lib", fromlist=["Shard"]).Shard(npth, r, world, 0, nb))))
            full = sharding.export_words(g.counts_all[r], nb, nrx)
            ok &= exports[r].numel() < full                          # (C3: ~14 % of the records are blocked)
            for b, v in enumerate(sharding.unpack_export(exports[r].cpu(), cnt, nb, nrx, ub_cnt, sharding.UNBLOCKED)):
                h = int(cnt[b + 1])
                if not h:
                    continue
                ray = v["hit"][0].numpy().astype(np.int64) & 0xFFFFFFFF
                tx = ray // n_loc
                i = ray - tx * n_loc
                p = ((i // ch) * world + r) * ch + i % ch
                for rx in range(nrx):
                    idx = v["index"][rx].numpy().astype(np.int64)
                    rec = v["rec"][rx].numpy().view(np.float32)     # [9, U]
                    ok &= bool((np.diff(idx) > 0).all())            # hit order
                    seen[rx, tx[idx], b, p[idx]] = True
                    for k, name in enumerate(("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau")):
                        ok &= np.array_equal(rec[k].view(np.uint32), ref["scat"][name][rx, tx[idx], b, p[idx]].view(np.uint32))
                    d = ref["scat"]["directions_rx"][rx, tx[idx], b, p[idx]]
                    ok &= np.array_equal(rec[5:8].T.view(np.uint32), d.view(np.uint32))
        ok &= np.array_equal(seen, written(ref["scat"]["directions_rx"][..., 0]))   # dir_rx is written for unblocked records only
        q.put(bool(ok))
        g2.close()
        g.close()
        tr.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_real_tracer_shards_gathered_over_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get() is True      # full export
    assert q.get() is True      # unblocked records only
