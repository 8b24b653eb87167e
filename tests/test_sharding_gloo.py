"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups run the same pack ->
variable-size gather -> unpack code the GPUs run over RCCL (hermespy_rt_amd.sharding), on
synthetic compact results, and the root checks every received word."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hermespy_rt_amd import sharding


class FakeResult:
    """Stands in for device.Tracer: same block accessors, CPU tensors, seeded contents."""

    def __init__(self, rank, nb=3, nrx=2, cap=1024):
        g = torch.Generator().manual_seed(1234 + rank)
        self.nb, self.nrx, self.cap = nb, nrx, cap
        self.hb = [torch.randint(-2**31, 2**31 - 1, (15, cap), dtype=torch.int32, generator=g) for _ in range(nb)]
        self.rb = [torch.randint(-2**31, 2**31 - 1, (nrx, 9, cap), dtype=torch.int32, generator=g) for _ in range(nb)]
        self.mb = [torch.randint(-2**31, 2**31 - 1, (nrx, 2 * cap // 64), dtype=torch.int32, generator=g) for _ in range(nb)]
        live = [cap - 7 * rank]
        for b in range(nb):
            live.append(max(0, live[-1] // (2 + rank) - (1 if b else 0)))
        live[-1] = 0 if rank == 1 else live[-1]          # an empty bounce on one rank
        self.counts = torch.tensor([0] + live[1:] + [0], dtype=torch.int32)

    def counts_tensor(self):
        return self.counts

    def hit_block(self, b):
        return self.hb[b]

    def rec_block(self, b):
        return self.rb[b]

    def mask_block(self, b):
        return self.mb[b]


def _expect(src):
    c = src.counts.numpy().astype(np.int64)
    return sharding.pack_export(src, c).clone(), c


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src = FakeResult(rank)
        g = sharding.RecordGather(src, dst=0)
        for _ in range(2):                      # buffers are reused across steps
            exports = g.run()
        if rank == 0:
            ok = True
            for r in range(world):
                want, c = _expect(FakeResult(r))
                ok &= bool(torch.equal(exports[r], want))
                ok &= list(g.counts_all[r]) == list(c)
                views = sharding.unpack_export(exports[r], c, src.nb, src.nrx)
                for b, v in enumerate(views):
                    h = int(c[b + 1])
                    ok &= bool(torch.equal(v["hit"], FakeResult(r).hb[b][:4, :h]))
                    ok &= bool(torch.equal(v["rec"], FakeResult(r).rb[b][:, :, :h]))
                    ok &= bool(torch.equal(v["mask"], FakeResult(r).mb[b][:, :2 * ((h + 63) // 64)]))
            q.put(ok)
        else:
            assert exports is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gather_of_packed_records_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True


def test_pack_unpack_roundtrip_single_process():
    src = FakeResult(0, nb=4, nrx=3, cap=512)
    c = src.counts.numpy().astype(np.int64)
    buf = sharding.pack_export(src, c)
    assert buf.numel() == sharding.export_words(c, src.nb, src.nrx)
    for b, v in enumerate(sharding.unpack_export(buf, c, src.nb, src.nrx)):
        h = int(c[b + 1])
        assert torch.equal(v["hit"], src.hb[b][:4, :h])
        assert torch.equal(v["rec"], src.rb[b][:, :, :h])
