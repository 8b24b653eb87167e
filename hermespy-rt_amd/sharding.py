"""Ray sharding across GPUs: one process per GPU, RCCL gather of the compact path records.

Rays are independent, so the launch set is cut into round-robin shards (hrt_device.h,
hrt_shard) and every rank traces its shard with a full copy of the (tiny) scene.  The only
exchange step of the path is the gather of each rank's path records to rank 0, over xGMI:
every peer -> root transfer has its own link, so it is one variable-size send per peer
(grouped isend/irecv = ncclGroup of ncclSend/ncclRecv), not a ring collective.

The export format is flat int32 words (floats are bit-cast), per bounce b with H hits:
    hit rows   [4, H]        ray (tx*num_local + local_i), tri, theta, fs0
    records    [nrx, 9, H]   HRT_REC_* fields
    masks      [nrx, 2*ceil(H/64)]  "unblocked" bit words
`unpack_export` returns views into a received buffer; `Shard mapping` of the sender (rank,
world, chunk, num_local) turns local ray ids into global path indices.
"""
import torch
import torch.distributed as dist

N_HIT_ROWS = 4   # ray, tri, theta, fs0 -- the per-hit data a consumer of records needs
N_REC = 9


def export_words(counts, nb, nrx):
    """int32 words of the export of a rank with live counts `counts` (counts[b+1] = H_b)."""
    n = 0
    for b in range(nb):
        h = int(counts[b + 1])
        n += N_HIT_ROWS * h + nrx * N_REC * h + nrx * 2 * ((h + 63) // 64)
    return n


def pack_export(src, counts, out=None):
    """Pack the compact result of `src` (a device.Tracer or anything with nb, nrx,
    hit_block(b), rec_block(b), mask_block(b)) into one contiguous int32 tensor: three strided
    copies per bounce.  `counts` is the host copy of the live counts."""
    nb, nrx = src.nb, src.nrx
    n = export_words(counts, nb, nrx)
    ref = src.hit_block(0)
    if out is None or out.numel() < n:
        out = torch.empty(max(n, 1), dtype=torch.int32, device=ref.device)
    pos = 0
    for b in range(nb):
        h = int(counts[b + 1])
        if h == 0:
            continue
        k = N_HIT_ROWS * h
        out[pos:pos + k].view(N_HIT_ROWS, h).copy_(src.hit_block(b)[:N_HIT_ROWS, :h])
        pos += k
        k = nrx * N_REC * h
        out[pos:pos + k].view(nrx, N_REC, h).copy_(src.rec_block(b)[:, :, :h])
        pos += k
        nw = 2 * ((h + 63) // 64)
        k = nrx * nw
        out[pos:pos + k].view(nrx, nw).copy_(src.mask_block(b)[:, :nw])
        pos += k
    assert pos == n
    return out[:n]


def unpack_export(buf, counts, nb, nrx):
    """Views into a packed export: list over bounces of dict(hit=[4,H], rec=[nrx,9,H],
    mask=[nrx, 2*ceil(H/64)])."""
    out, pos = [], 0
    for b in range(nb):
        h = int(counts[b + 1])
        nw = 2 * ((h + 63) // 64)
        hit = buf[pos:pos + N_HIT_ROWS * h].view(N_HIT_ROWS, h)
        pos += N_HIT_ROWS * h
        rec = buf[pos:pos + nrx * N_REC * h].view(nrx, N_REC, h)
        pos += nrx * N_REC * h
        mask = buf[pos:pos + nrx * nw].view(nrx, nw)
        pos += nrx * nw
        out.append(dict(hit=hit, rec=rec, mask=mask))
    return out


class RecordGather:
    """Gather of every rank's packed export to `dst`.  Reusable across steps (buffers are
    kept).  After run(): on dst, `self.counts_all` [world, nb+2] and `self.exports[r]` (views;
    exports[dst] is the local pack)."""

    def __init__(self, src, dst=0, group=None):
        self.src, self.dst, self.group = src, dst, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # gloo cannot move device tensors point-to-point: stage through the host there (CPU
        # tests and single-GPU rehearsals); RCCL ("nccl") sends HBM to HBM over xGMI
        self.via_host = dist.get_backend(group) == "gloo"
        self.pack_buf = None
        self.recv_bufs = [None] * self.world
        self.exports = [None] * self.world
        self.counts_all = None

    def run(self):
        src = self.src
        nb, nrx = src.nb, src.nrx
        c_dev = src.counts_tensor().to(torch.int64)
        if self.via_host:
            c_dev = c_dev.cpu()
        all_c = torch.empty(self.world * (nb + 2), dtype=torch.int64, device=c_dev.device)
        dist.all_gather_into_tensor(all_c, c_dev, group=self.group)
        counts_all = all_c.view(self.world, nb + 2).cpu().numpy()   # sizes are needed on the host
        self.counts_all = counts_all
        mine = pack_export(src, counts_all[self.rank], self.pack_buf)
        if self.pack_buf is None or self.pack_buf.numel() < mine.numel():
            self.pack_buf = mine
        if self.via_host:
            mine = mine.cpu()
        ops = []
        if self.rank == self.dst:
            for r in range(self.world):
                if r == self.rank:
                    self.exports[r] = mine
                    continue
                n = export_words(counts_all[r], nb, nrx)
                if self.recv_bufs[r] is None or self.recv_bufs[r].numel() < n:
                    self.recv_bufs[r] = torch.empty(max(n, 1), dtype=torch.int32, device=mine.device)
                self.exports[r] = self.recv_bufs[r][:n]
                if n:
                    ops.append(dist.P2POp(dist.irecv, self.exports[r], r, self.group))
        elif mine.numel():
            ops.append(dist.P2POp(dist.isend, mine, self.dst, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return self.exports if self.rank == self.dst else None
