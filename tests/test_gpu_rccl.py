"""RCCL itself (torch.distributed backend "nccl" IS RCCL on ROCm): the device-tensor branch of the
N > 1 path -- sharding.RecordGather with HBM tensors (all_gather_into_tensor of the counts on the
device, the packed export left in HBM), the barrier / all_reduce bench.py issues around its timed
region, and a send/recv of a packed export to the rank itself -- on a process group of ONE rank: the
test box has one GPU and RCCL refuses two ranks on one device.  What this proves: the RCCL library
loads and initialises next to the tracer's HIP runtime, and every call the multi-GPU path makes is
accepted with the tensors it passes (dtype, device, contiguity); the transport between ranks is RCCL's.
Runs in a child process (a process group is process-global), under a timeout."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        from hermespy_rt_amd import sharding
        from hermespy_rt_amd.device import Tracer
        from tests import configs as K
        c = K.small(K.C3_DOPPLER, 50000)
        tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                    c["num_paths"], c["num_bounces"], rank=0, world=1)
        tr.trace()
        # the collectives of bench.py's timed region, on device tensors
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(t.item()) == 1.5
        g = sharding.RecordGather(tr, dst=0)
        assert g.via_host is False                      # the RCCL branch: HBM tensors, no host staging
        for _ in range(2):                              # buffers are reused across steps
            exports = g.run()
        assert exports is not None and exports[0].is_cuda
        counts = tr.counts()
        assert [int(x) for x in g.counts_all[0][1:tr.nb + 1]] == [int(x) for x in counts[1:tr.nb + 1]]
        ok = True
        for b, v in enumerate(sharding.unpack_export(exports[0], g.counts_all[0], tr.nb, tr.nrx)):
            h = int(counts[b + 1])
            ok &= bool(torch.equal(v["hit"], tr.hit_block(b)[:sharding.N_HIT_ROWS, :h]))
            ok &= bool(torch.equal(v["rec"], tr.rec_block(b)[:, :, :h]))
            ok &= bool(torch.equal(v["mask"], tr.mask_block(b)[:, :2 * ((h + 63) // 64)]))
        # a point-to-point transfer through RCCL: the packed export to this very rank and back
        src = exports[0]
        dst = torch.empty_like(src)
        ops = [dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        ok &= bool(torch.equal(src, dst))
        # ---- the C entry (include/hrt_device.h: hrt_gather_rccl): what a C / C++ consumer with one process per
        # GPU calls -- its own communicator from hrt_rccl_unique_id / hrt_rccl_comm_create, ncclAllGather of the
        # meta blocks, one group of ncclSend / ncclRecv (world 1: the root sends its export to itself) ----
        import ctypes as C
        from hermespy_rt_amd import lib
        L = lib.load()
        uid = lib.RcclId()
        lib.check(L.hrt_rccl_unique_id(C.byref(uid)), "hrt_rccl_unique_id")
        comm = C.c_void_p()
        lib.check(L.hrt_rccl_comm_create(C.byref(uid), 1, 0, 0, C.byref(comm)), "hrt_rccl_comm_create")
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for flags in (lib.EXPORT_FULL, lib.EXPORT_UNBLOCKED):
            gh = C.c_void_p()
            lib.check(L.hrt_gather_create(tr.problem, C.byref(tr.shard), 0, flags, C.byref(gh)), "hrt_gather_create")
            for _ in range(2):
                lib.check(L.hrt_gather_rccl(gh, comm, C.c_void_p(tr.ws.data_ptr()), st, 1), "hrt_gather_rccl")
            ptr, words = C.c_void_p(), C.c_uint64()
            lib.check(L.hrt_gather_export(gh, 0, C.byref(ptr), C.byref(words)), "hrt_gather_export")
            n = int(words.value)
            mine = torch.as_tensor(sharding._DevPtr(ptr.value, n), device=dev)
            got = torch.as_tensor(sharding._DevPtr(L.hrt_gather_received(gh, 0), n), device=dev)
            ok &= bool(torch.equal(mine, got)) and n > 0
            mw = int(L.hrt_gather_meta_words(gh))
            meta = np.ctypeslib.as_array(L.hrt_gather_meta(gh, 0), shape=(mw,)).copy()
            ok &= [int(x) for x in meta[1:tr.nb + 1]] == [int(x) for x in counts[1:tr.nb + 1]]
            if flags == lib.EXPORT_FULL:
                ok &= bool(torch.equal(got, exports[0]))          # the torch binding packs the same run
            else:
                ub = meta[tr.nb + 2:]
                views = sharding.unpack_export(got, meta[:tr.nb + 2], tr.nb, tr.nrx, ub, sharding.UNBLOCKED)
                for b, v in enumerate(views):
                    h = int(counts[b + 1])
                    rec = tr.rec_block(b)
                    bits = tr.mask_block(b).cpu().numpy().view(np.uint64)
                    for rx in range(tr.nrx):
                        m = ((bits[rx][np.arange(h) // 64] >> (np.arange(h) % 64).astype(np.uint64)) & np.uint64(1)).astype(bool)
                        idx = torch.from_numpy(np.nonzero(m)[0].astype(np.int64)).to(dev)
                        ok &= int(ub[b * tr.nrx + rx]) == int(m.sum())
                        ok &= bool(torch.equal(v["index"][rx].to(torch.int64), idx))
                        ok &= bool(torch.equal(v["rec"][rx], rec[rx][:, idx]))
            L.hrt_gather_destroy(gh)
        lib.check(L.hrt_rccl_comm_destroy(comm), "hrt_rccl_comm_destroy")
        q.put(bool(ok))
        g.close()
        tr.close()
    finally:
        dist.destroy_process_group()


def test_record_gather_over_rccl_world_1():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = ctx.Process(target=_worker, args=(port, q))
    p.start()
    p.join(240)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("RCCL worker did not finish in 240 s")
    assert p.exitcode == 0
    assert q.get() is True
