"""Ray sharding across GPUs: one process per GPU, gather of the compact path records to one rank.

Rays are independent, so the launch set is cut into round-robin shards (hrt_device.h,
hrt_shard) and every rank traces its shard with a full copy of the (tiny) scene.  The only
exchange step of the path is the gather of each rank's packed export to rank 0, over xGMI:
every peer -> root transfer has its own link, so it is one variable-size send per peer
(grouped isend/irecv = ncclGroup of ncclSend/ncclRecv), not a ring collective.

This module is the torch.distributed binding of the C ABI's export (include/hrt_device.h, "packed
export and gather"; csrc/host/gather.c): the layout comes from hrt_export_words / hrt_export_locate,
a Tracer's export is packed on the device by hrt_gather_prepare / hrt_gather_pack, and only the
transport is torch's (RCCL for HBM tensors; gloo, staged through the host, in CPU tests and one-GPU
rehearsals).  A C / C++ consumer uses hrt_gather_rccl for the same exchange without Python.

A rank's META block = counts[nb + 2] | unblocked[nb][nrx]; the run, per bounce b with H hits:
    hit rows   [4, H]        ray (tx*num_local + local_i), tri, theta, fs0
    FULL:      per rx  records [9, H], mask [2*ceil(H/64)] ("unblocked" bit words)
    UNBLOCKED: per rx  index [U] (hit of each kept record), records [9, U]
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import lib as _lib

N_HIT_ROWS = 4   # ray, tri, theta, fs0 -- the per-hit data a consumer of records needs
N_REC = 9
FULL, UNBLOCKED = _lib.EXPORT_FULL, _lib.EXPORT_UNBLOCKED


def _u32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64) & 0xFFFFFFFF, dtype=np.uint32)


def make_meta(counts, nb, nrx, unblocked=None):
    """meta block from the counts (and, for UNBLOCKED exports, the per-(bounce, rx) unblocked counts)"""
    m = np.zeros(nb + 2 + nb * nrx, np.uint32)
    m[:nb + 2] = _u32(counts)[:nb + 2]
    if unblocked is not None:
        m[nb + 2:] = _u32(unblocked).reshape(-1)
    return m


def export_words(counts, nb, nrx, unblocked=None, flags=FULL):
    """int32 words of the export of a rank with live counts `counts` (counts[b+1] = H_b)."""
    m = make_meta(counts, nb, nrx, unblocked)
    return int(_lib.load().hrt_export_words(m.ctypes.data_as(C.POINTER(C.c_uint32)), nb, nrx, flags))


def locate(meta, nb, nrx, flags, b, rx):
    part = _lib.ExportPart()
    _lib.check(_lib.load().hrt_export_locate(meta.ctypes.data_as(C.POINTER(C.c_uint32)), nb, nrx, flags, b, rx, C.byref(part)),
               "hrt_export_locate")
    return part


def pack_export(src, counts, out=None):
    """FULL export of `src` (anything with nb, nrx, hit_block(b), rec_block(b), mask_block(b): a
    device.Tracer, or the CPU stand-in of the gloo tests) by strided torch copies at the offsets the C
    layout functions give.  (A Tracer's export is normally packed by the library: RecordGather.)"""
    nb, nrx = src.nb, src.nrx
    meta = make_meta(counts, nb, nrx)
    n = export_words(counts, nb, nrx)
    ref = src.hit_block(0)
    if out is None or out.numel() < n:
        out = torch.empty(max(n, 1), dtype=torch.int32, device=ref.device)
    for b in range(nb):
        h = int(counts[b + 1])
        if h == 0:
            continue
        for rx in range(nrx):
            p = locate(meta, nb, nrx, FULL, b, rx)
            if rx == 0:
                out[p.off_hit:p.off_hit + N_HIT_ROWS * h].view(N_HIT_ROWS, h).copy_(src.hit_block(b)[:N_HIT_ROWS, :h])
            out[p.off_rec:p.off_rec + N_REC * h].view(N_REC, h).copy_(src.rec_block(b)[rx, :, :h])
            nw = 2 * ((h + 63) // 64)
            out[p.off_mask:p.off_mask + nw].copy_(src.mask_block(b)[rx, :nw])
    return out[:n]


def unpack_export(buf, counts, nb, nrx, unblocked=None, flags=FULL):
    """Views into a packed export: list over bounces of dict(hit=[4,H], and FULL: rec=[nrx,9,H],
    mask=[nrx, 2*ceil(H/64)]; UNBLOCKED: index=[U_rx] and rec=[9,U_rx] per rx, as lists)."""
    meta = make_meta(counts, nb, nrx, unblocked)
    out = []
    for b in range(nb):
        h = int(counts[b + 1])
        p0 = locate(meta, nb, nrx, flags, b, 0)
        hit = buf[p0.off_hit:p0.off_hit + N_HIT_ROWS * h].view(N_HIT_ROWS, h)
        if flags == FULL:
            nw = 2 * ((h + 63) // 64)
            # (per rx: records then mask, rx after rx -- contiguous, so the old [nrx, ...] views still hold
            # as strided views of one block)
            stride = N_REC * h + nw
            block = buf[p0.off_rec:p0.off_rec + nrx * stride].view(nrx, stride) if h else buf[:0].view(nrx, 0)
            rec = block[:, :N_REC * h].reshape(nrx, N_REC, h) if h else buf[:0].view(nrx, N_REC, 0)
            mask = block[:, N_REC * h:] if h else buf[:0].view(nrx, 0)
            out.append(dict(hit=hit, rec=rec, mask=mask))
        else:
            idx, rec = [], []
            for rx in range(nrx):
                p = locate(meta, nb, nrx, flags, b, rx)
                u = int(p.records)
                idx.append(buf[p.off_index:p.off_index + u])
                rec.append(buf[p.off_rec:p.off_rec + N_REC * u].view(N_REC, u))
            out.append(dict(hit=hit, index=idx, rec=rec))
    return out


class _DevPtr:
    """a device allocation owned by the library, as something torch.as_tensor can alias"""

    def __init__(self, ptr, words):
        self.__cuda_array_interface__ = dict(shape=(int(words),), typestr="<i4", data=(int(ptr), False), version=2)


class RecordGather:
    """Gather of every rank's packed export to `dst`.  Reusable across steps (buffers are kept).
    After run(): on dst, `self.counts_all` [world, nb+2], `self.meta_all` [world, meta words] and
    `self.exports[r]` (views; exports[dst] is the local pack).  `unblocked_only`: blocked records (all
    zero) do not travel (HRT_EXPORT_UNBLOCKED)."""

    def __init__(self, src, dst=0, group=None, unblocked_only=False):
        self.src, self.dst, self.group = src, dst, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # gloo cannot move device tensors point-to-point: stage through the host there (CPU
        # tests and single-GPU rehearsals); RCCL ("nccl") sends HBM to HBM over xGMI
        self.via_host = dist.get_backend(group) == "gloo"
        self.flags = UNBLOCKED if unblocked_only else FULL
        self.pack_buf = None
        self.recv_bufs = [None] * self.world
        self.exports = [None] * self.world
        self.counts_all = self.meta_all = None
        # a Tracer: the library packs on the device (hrt_gather_*); anything else (the CPU stand-in of the
        # gloo tests): torch copies at the same offsets
        self.g = None
        if hasattr(src, "problem"):
            self.L = _lib.load()
            h = C.c_void_p()
            _lib.check(self.L.hrt_gather_create(src.problem, C.byref(src.shard), dst, self.flags, C.byref(h)), "hrt_gather_create")
            self.g = h
        elif unblocked_only:
            raise ValueError("unblocked_only needs a device.Tracer (the library compacts the records)")

    def close(self):
        if self.g is not None:
            self.L.hrt_gather_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.src.device).cuda_stream)

    def run(self):
        src = self.src
        nb, nrx = src.nb, src.nrx
        mw = nb + 2 + nb * nrx
        if self.g is not None:
            _lib.check(self.L.hrt_gather_prepare(self.g, C.c_void_p(src.ws.data_ptr()), self._stream()), "hrt_gather_prepare")
            m_dev = torch.as_tensor(_DevPtr(self.L.hrt_gather_meta_device(self.g), mw), device=src.device).to(torch.int64) & 0xFFFFFFFF
        else:
            c = src.counts_tensor().to(torch.int64) & 0xFFFFFFFF
            m_dev = torch.cat([c, torch.zeros(nb * nrx, dtype=torch.int64, device=c.device)])
        if self.via_host:
            m_dev = m_dev.cpu()
        all_m = torch.empty(self.world * mw, dtype=torch.int64, device=m_dev.device)
        dist.all_gather_into_tensor(all_m, m_dev, group=self.group)
        meta_all = all_m.view(self.world, mw).cpu().numpy()   # sizes are needed on the host
        self.meta_all = meta_all
        self.counts_all = meta_all[:, :nb + 2]
        if self.g is not None:
            for r in range(self.world):
                m = _u32(meta_all[r])
                _lib.check(self.L.hrt_gather_set_meta(self.g, r, m.ctypes.data_as(C.POINTER(C.c_uint32))), "hrt_gather_set_meta")
            ptr, words = C.c_void_p(), C.c_uint64()
            _lib.check(self.L.hrt_gather_pack(self.g, C.c_void_p(src.ws.data_ptr()), self._stream(), C.byref(ptr), C.byref(words)),
                       "hrt_gather_pack")
            n = int(words.value)
            mine = (torch.as_tensor(_DevPtr(ptr.value, n), device=src.device) if n
                    else torch.empty(0, dtype=torch.int32, device=src.device))
        else:
            mine = pack_export(src, self.counts_all[self.rank], self.pack_buf)
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
                n = export_words(self.counts_all[r], nb, nrx, meta_all[r][nb + 2:], self.flags)
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
