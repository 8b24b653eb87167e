"""Device-resident tracer: the hot path on torch-owned HBM buffers and streams.

torch is plumbing here (device memory, streams, torch.distributed); the work is done by
hrt_trace() of libhermespy_rt_amd.so on the CURRENT torch stream, called through the C ABI
with raw device pointers.

    tr = Tracer(scene_path, rx_pos, tx_pos, rx_vel, tx_vel, f_ghz, num_paths, num_bounces,
                rank=0, world=1)
    tr.trace()                 # asynchronous on torch's current stream
    counts = tr.counts()       # [nb+2] numpy (syncs)
    hits = tr.hits(b)          # dict name -> torch view of the first H_b elements
    recs = tr.records(b)       # dict name -> [nrx, H_b] torch views, + 'unblocked' bool
"""
import ctypes as C

import numpy as np

from . import abi
from . import lib as _lib


class Tracer:
    def __init__(self, scene_path, rx_pos, tx_pos, rx_vel, tx_vel, f_ghz, num_paths,
                 num_bounces, rank=0, world=1, chunk=0, device=None, host_threads=0, coherent=True,
                 device_dirs=False):
        import torch  # torch first: its HIP runtime is the one the library binds to

        if not torch.cuda.is_available():
            raise _lib.HrtError("no HIP device visible to torch; hermespy-rt_amd has no CPU path")
        self.torch = torch
        self.L = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        rx_pos = np.ascontiguousarray(np.asarray(rx_pos, np.float32).reshape(-1, 3))
        tx_pos = np.ascontiguousarray(np.asarray(tx_pos, np.float32).reshape(-1, 3))
        self.nrx, self.ntx = rx_pos.shape[0], tx_pos.shape[0]
        rx_vel = np.ascontiguousarray(np.asarray(rx_vel, np.float32).reshape(self.nrx, 3))
        tx_vel = np.ascontiguousarray(np.asarray(tx_vel, np.float32).reshape(self.ntx, 3))
        self.num_paths, self.nb = int(num_paths), int(num_bounces)
        self.f_ghz = float(f_ghz)
        V3 = C.POINTER(abi.Vec3)

        scene = self.L.scene_load(str(scene_path).encode())
        try:
            h = C.c_void_p()
            _lib.check(self.L.hrt_problem_create(
                C.byref(scene), rx_pos.ctypes.data_as(V3), tx_pos.ctypes.data_as(V3),
                rx_vel.ctypes.data_as(V3), tx_vel.ctypes.data_as(V3), C.c_float(self.f_ghz),
                self.nrx, self.ntx, self.device.index, C.byref(h)), "hrt_problem_create")
            self.problem = h
        finally:
            abi.free_scene(scene)
        self.num_tri = int(self.L.hrt_problem_num_triangles(self.problem))
        # rows of the device table are in a spatial order (acceleration structure); tri_order[row]
        # = the flat index the reference's (mesh, face) scan gives that triangle
        self.tri_order = np.empty(max(self.num_tri, 1), np.uint32)
        _lib.check(self.L.hrt_problem_tri_order(self.problem,
                                                self.tri_order.ctypes.data_as(C.POINTER(C.c_uint32))),
                   "hrt_problem_tri_order")

        self.shard = _lib.Shard(self.num_paths, rank, world, chunk, self.nb)
        self.num_local = int(self.L.hrt_shard_num_local(C.byref(self.shard)))
        self.layout = _lib.Layout()
        _lib.check(self.L.hrt_layout_query(self.problem, C.byref(self.shard), C.byref(self.layout)),
                   "hrt_layout_query")
        self.cap = int(self.layout.cap)

        # launch directions of this shard, once: host libm (bit-identical to the reference by
        # construction), or generated on the device with the host patching the few values whose
        # float rounding could depend on the math library (bit-identical too; see hrt_device.h)
        self.dirs_patched = None
        if device_dirs:
            with torch.cuda.device(self.device):
                self.dirs = torch.empty((self.num_local, 3), dtype=torch.float32, device=self.device)
            n_p = C.c_uint64(0)
            _lib.check(self.L.hrt_launch_dirs_device(
                C.byref(self.shard), C.c_void_p(self.dirs.data_ptr()), self.device.index,
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), C.byref(n_p)),
                "hrt_launch_dirs_device")
            self.dirs_patched = int(n_p.value)
            dirs = None
        else:
            dirs = np.empty((self.num_local, 3), np.float32)
            _lib.check(self.L.hrt_launch_dirs_host(
                C.byref(self.shard), dirs.ctypes.data_as(C.POINTER(C.c_float)), host_threads),
                "hrt_launch_dirs_host")
        self.dirs_host = dirs
        # coherent launch order: a wave = a narrow ray packet (speed only; results are keyed
        # by ray id)
        self.order_host = None
        order_dev = None
        if coherent and device_dirs:
            # ... and the coherent order on the device too (hrt_launch_order_device)
            with torch.cuda.device(self.device):
                order_dev = torch.empty(self.num_local, dtype=torch.int32, device=self.device)
            _lib.check(self.L.hrt_launch_order_device(
                C.byref(self.shard), C.c_void_p(order_dev.data_ptr()), self.device.index,
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "hrt_launch_order_device")
        elif coherent:
            order = np.empty(self.num_local, np.uint32)
            _lib.check(self.L.hrt_launch_order_host(
                C.byref(self.shard), dirs.ctypes.data_as(C.POINTER(C.c_float)) if dirs is not None else None,
                order.ctypes.data_as(C.POINTER(C.c_uint32))), "hrt_launch_order_host")
            self.order_host = order
        with torch.cuda.device(self.device):
            if dirs is not None:
                self.dirs = torch.from_numpy(dirs).to(self.device)
            self.order = order_dev if order_dev is not None else (
                torch.from_numpy(self.order_host.view(np.int32)).to(self.device) if coherent else None)
            # the launch set is traced again and again: permute the direction table into launch
            # order ONCE, so the launch kernels read contiguous runs (HRT_DIRS_IN_LAUNCH_ORDER)
            self.flags = 0
            if coherent:
                self.dirs_launch = self.dirs[self.order.to(torch.int64) & 0xFFFFFFFF].contiguous()
                self.flags = _lib.DIRS_IN_LAUNCH_ORDER
            self.ws = torch.empty(int(self.layout.total_bytes), dtype=torch.uint8, device=self.device)
        assert self.ws.data_ptr() % 256 == 0
        self.last_times = None

    def close(self):
        if getattr(self, "problem", None):
            self.L.hrt_problem_destroy(self.problem)
            self.problem = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ run
    def trace(self, timed=False):
        """Enqueue the whole path on torch's current stream.  timed=True records HIP events
        around every launch on that stream, waits, and returns (los_ms, [trace_ms per launch]);
        shade / compaction times are in last_shade_ms / last_compact_ms."""
        torch = self.torch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        times = _lib.KernelTimes() if timed else None
        if timed:
            t = self.new_timer()
            self.trace_with_timer(t)
            _lib.check(self.L.hrt_timer_read(t, C.byref(times)), "hrt_timer_read")
            self.L.hrt_timer_destroy(t)
        else:
            _lib.check(self.L.hrt_trace_flags(
                self.problem, C.byref(self.shard), C.c_void_p(self._dirs_ptr()),
                C.c_void_p(self.order.data_ptr()) if self.order is not None else None,
                C.c_void_p(self.ws.data_ptr()), C.c_uint64(self.ws.numel()), C.c_void_p(stream),
                None, C.c_uint32(self.flags)), "hrt_trace_flags")
        if timed:
            n = int(times.num_bounce_launches)
            self.last_times = (float(times.los_ms), [float(times.trace_ms[i]) for i in range(n)])
            self.last_shade_ms = [float(times.shade_ms[i]) for i in range(n)]
            self.last_compact_ms = [0.0] * n
            self.last_records_ms = [float(times.records_ms[i]) for i in range(n)]
            return self.last_times
        return None

    def regen_launch_tables(self):
        """Generate this shard's launch directions and coherent launch order again, ON THE DEVICE
        (hrt_launch_dirs_device, hrt_launch_order_device), and permute the direction table into launch
        order -- what a caller pays per call when the launch set is not kept (bench.py:
        step_incl_launch_ms).  The directions are bit-identical to the host's; results do not depend
        on the order."""
        torch = self.torch
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if self.order is None:
            raise _lib.HrtError("regen_launch_tables needs a coherent Tracer")
        _lib.check(self.L.hrt_launch_dirs_device(C.byref(self.shard), C.c_void_p(self.dirs.data_ptr()),
                                                 self.device.index, stream, None), "hrt_launch_dirs_device")
        _lib.check(self.L.hrt_launch_order_device(C.byref(self.shard), C.c_void_p(self.order.data_ptr()),
                                                  self.device.index, stream), "hrt_launch_order_device")
        self.dirs_launch = self.dirs[self.order.to(torch.int64) & 0xFFFFFFFF].contiguous()
        self.flags = _lib.DIRS_IN_LAUNCH_ORDER

    # ------------------------------------------------------------------ deferred timing
    def new_timer(self):
        t = C.c_void_p()
        _lib.check(self.L.hrt_timer_create(self.nb, C.byref(t)), "hrt_timer_create")
        return t

    def trace_with_timer(self, timer):
        """Like trace(): asynchronous; HIP events around every kernel go into `timer`."""
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.L.hrt_trace_flags(
            self.problem, C.byref(self.shard), C.c_void_p(self._dirs_ptr()),
            C.c_void_p(self.order.data_ptr()) if self.order is not None else None,
            C.c_void_p(self.ws.data_ptr()), C.c_uint64(self.ws.numel()), C.c_void_p(stream), timer,
            C.c_uint32(self.flags)), "hrt_trace_flags")

    def _dirs_ptr(self):
        return (self.dirs_launch if self.flags & _lib.DIRS_IN_LAUNCH_ORDER else self.dirs).data_ptr()

    def read_timer(self, timer, destroy=True):
        """-> dict(los_ms, trace_ms[], shade_ms[], records_ms[], scan_ms[]); waits for the timer's last event."""
        times = _lib.KernelTimes()
        _lib.check(self.L.hrt_timer_read(timer, C.byref(times)), "hrt_timer_read")
        n = int(times.num_bounce_launches)
        out = dict(los_ms=float(times.los_ms), trace_ms=[float(times.trace_ms[i]) for i in range(n)],
                   shade_ms=[float(times.shade_ms[i]) for i in range(n)],
                   records_ms=[float(times.records_ms[i]) for i in range(n)],
                   scan_ms=[0.0] * n)
        if destroy:
            self.L.hrt_timer_destroy(timer)
        return out

    # ------------------------------------------------------------------ views
    def _view(self, off, n, dtype):
        t = self.ws[off:off + n * 4]
        return t.view(dtype)

    def error_word(self):
        """counts[nb + 1] of the last step: 0, or HRT_ERR_* bits (0x100 / 0x200: the step is void, hrt_kparams.h)"""
        return int(self._view(int(self.layout.off_counts), self.nb + 2, self.torch.int32)[self.nb + 1].item()) & 0xFFFFFFFF

    def counts(self, retrace=True):
        """counts[0 .. nb + 1] of the last step (counts[b + 1] = hits of bounce b).  A VOID step -- a fused launch or
        the chain kernel gave up waiting because the GPU is shared (HRT_ERR_FUSE_TIMEOUT / HRT_ERR_CHAIN_TIMEOUT in
        the error word; the library has switched that kernel off for the process by the time this is read) -- is
        traced again here, like the drop-in calls do (retrace=False: raise instead)."""
        torch = self.torch
        for attempt in range(3):
            c = self._view(int(self.layout.off_counts), self.nb + 2, torch.int32).cpu().numpy()
            c = c.astype(np.int64) & 0xFFFFFFFF
            c[0] = self.ntx * self.num_local
            if not (int(c[self.nb + 1]) & 0x300) or not retrace or attempt == 2:   # (hrt_kparams.h: HRT_ERR_VOID)
                break
            self.trace()
        if int(c[self.nb + 1]) & 0x300:
            raise _lib.HrtError("a fused launch timed out waiting for the workgroups in front of it (the GPU is shared "
                                "with other such kernels): this step is void -- trace() again; the library has switched "
                                "to smaller kernels for the rest of the process")
        if c[self.nb + 1] != 0:   # set by the shade kernel if a trace result was out of range
            raise _lib.HrtError("device reported an internal error flag %d" % int(c[self.nb + 1]))
        return c

    def los(self):
        t = self._view(int(self.layout.off_los), self.nrx * self.ntx * _lib.LOS_FLOATS,
                       self.torch.float32)
        return t.cpu().numpy().reshape(self.nrx, self.ntx, _lib.LOS_FLOATS)

    # whole-block views (all `cap` elements), used by sharding.pack_export
    def counts_tensor(self):
        return self._view(int(self.layout.off_counts), self.nb + 2, self.torch.int32)

    def hit_block(self, b):
        base = int(self.layout.off_hits) + b * int(self.layout.hit_block_bytes)
        nf = len(_lib.HIT_FIELDS)
        return self._view(base, nf * self.cap, self.torch.int32).view(nf, self.cap)

    def rec_block(self, b):
        base = int(self.layout.off_recs) + b * int(self.layout.rec_block_bytes)
        nf = len(_lib.REC_FIELDS)
        return self._view(base, self.nrx * nf * self.cap, self.torch.int32).view(self.nrx, nf, self.cap)

    def mask_block(self, b):
        words = self.cap // 64
        moff = int(self.layout.off_masks) + b * self.nrx * words * 8
        return self.ws[moff:moff + self.nrx * words * 8].view(self.torch.int32).view(self.nrx, 2 * words)

    def hits(self, b, n=None):
        """Fields of hit block b (rays that hit at bounce b, state after the bounce)."""
        torch = self.torch
        if n is None:
            n = int(self.counts()[b + 1])
        base = int(self.layout.off_hits) + b * int(self.layout.hit_block_bytes)
        out = {}
        for f, name in enumerate(_lib.HIT_FIELDS):
            dt = torch.int32 if name in ("ray", "tri") else torch.float32
            out[name] = self._view(base + f * self.cap * 4, self.cap, dt)[:n]
        return out

    def records(self, b, n=None):
        """Scatter records of bounce b: dict name -> [nrx, H_b]; 'unblocked' bool [nrx, H_b]."""
        torch = self.torch
        if n is None:
            n = int(self.counts()[b + 1])
        base = int(self.layout.off_recs) + b * int(self.layout.rec_block_bytes)
        nf = len(_lib.REC_FIELDS)
        blk = self._view(base, self.nrx * nf * self.cap, torch.float32).view(self.nrx, nf, self.cap)
        out = {name: blk[:, f, :n] for f, name in enumerate(_lib.REC_FIELDS)}
        words = self.cap // 64
        moff = int(self.layout.off_masks) + b * self.nrx * words * 8
        m = self.ws[moff:moff + self.nrx * words * 8].view(torch.int64).view(self.nrx, words)
        nw = (n + 63) // 64
        bits = (m[:, :nw, None] >> torch.arange(64, device=self.device)) & 1
        out["unblocked"] = bits.reshape(self.nrx, nw * 64)[:, :n].bool()
        return out

    def global_path(self, local_ray):
        """tx, global path index of local ray ids (numpy or torch int tensor)."""
        ch = int(self.shard.chunk) or 4096
        tx = local_ray // self.num_local
        i = local_ray - tx * self.num_local
        p = ((i // ch) * int(self.shard.count) + int(self.shard.rank)) * ch + i % ch
        return tx, p

    def work(self, counts=None):
        """Algorithmic work of the last trace (hrt_stats fields as a dict)."""
        if counts is None:
            counts = self.counts()
        c32 = np.ascontiguousarray(counts, dtype=np.uint32)
        st = _lib.Stats()
        self.L.hrt_work_from_counts(self.problem, C.byref(self.shard),
                                    c32.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(st))
        return dict(live=[int(st.live[i]) for i in range(self.nb + 1)], records=int(st.records),
                    tests=int(st.tests))

    # ------------------------------------------------------------------ path list (COO)
    def paths(self, nonzero_only=True, with_geometry=False):
        """The resolved scatter paths of the last trace as ONE list (device tensors), the form a
        channel model consumes -- instead of the reference's dense [rx][tx][bounce][path] arrays
        of which > 95 % are never written:

            rx, tx, bounce, path   int64 [n]   indices of the dense slot the record belongs to
                                               (path = GLOBAL path index, also on a shard)
            a_te, a_tm             complex64   gains (a blocked record has zeros)
            tau                    float32     delay; direction_rx float32 [n, 3]
            freq_shift             float32     launch Doppler term of the ray minus the record's
                                               (= the dense array's value for one TX; the
                                               reference's dense fill is undefined for more, Q9)
            unblocked              bool        False: blocked record (all-zero gains)
            mesh, face             int64       (with_geometry) triangle the ray left for the RX

        nonzero_only drops the blocked records (the reference writes zeros there)."""
        torch = self.torch
        counts = self.counts()
        cols = {k: [] for k in ("rx", "tx", "bounce", "path", "a_te", "a_tm", "tau", "direction_rx",
                                "freq_shift", "unblocked")}
        if with_geometry:
            cols["mesh"], cols["face"] = [], []
            T = self.num_tri
            mesh_h, face_h = np.empty(T, np.uint32), np.empty(T, np.uint32)
            _lib.check(self.L.hrt_problem_tri_ids(self.problem,
                                                  mesh_h.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                  face_h.ctypes.data_as(C.POINTER(C.c_uint32))),
                       "hrt_problem_tri_ids")
            mesh_d = torch.from_numpy(mesh_h.astype(np.int64)).to(self.device)
            face_d = torch.from_numpy(face_h.astype(np.int64)).to(self.device)
        for b in range(self.nb):
            n = int(counts[b + 1])
            if n == 0:
                continue
            h = self.hits(b, n)
            r = self.records(b, n)
            ray = h["ray"].to(torch.int64) & 0xFFFFFFFF
            tx, p = self.global_path(ray)
            for rx in range(self.nrx):
                ub = r["unblocked"][rx]
                sel = ub if nonzero_only else torch.ones_like(ub)
                k = int(sel.sum().item())
                if k == 0:
                    continue
                cols["rx"].append(torch.full((k,), rx, dtype=torch.int64, device=self.device))
                cols["tx"].append(tx[sel])
                cols["bounce"].append(torch.full((k,), b, dtype=torch.int64, device=self.device))
                cols["path"].append(p[sel])
                cols["a_te"].append(torch.complex(r["a_te_re"][rx][sel], r["a_te_im"][rx][sel]))
                cols["a_tm"].append(torch.complex(r["a_tm_re"][rx][sel], r["a_tm_im"][rx][sel]))
                cols["tau"].append(r["tau"][rx][sel])
                cols["direction_rx"].append(torch.stack([r["dirx"][rx][sel], r["diry"][rx][sel],
                                                         r["dirz"][rx][sel]], dim=1))
                cols["freq_shift"].append(h["fs0"][sel] - r["dfs"][rx][sel])
                cols["unblocked"].append(ub[sel])
                if with_geometry:
                    tri = h["tri"].to(torch.int64)[sel] & 0xFFFFFFFF
                    cols["mesh"].append(mesh_d[tri])
                    cols["face"].append(face_d[tri])
        out = {}
        for k, v in cols.items():
            if v:
                out[k] = torch.cat(v)
            else:
                shape = (0, 3) if k == "direction_rx" else (0,)
                dt = {"a_te": torch.complex64, "a_tm": torch.complex64, "tau": torch.float32,
                      "direction_rx": torch.float32, "freq_shift": torch.float32,
                      "unblocked": torch.bool}.get(k, torch.int64)
                out[k] = torch.empty(shape, dtype=dt, device=self.device)
        return out

    # ------------------------------------------------------------------ dense (host) view
    def to_dense(self, sentinel_u32=abi.SENTINEL_U32):
        """Assemble the reference's dense [rx][tx][b][p] scatter arrays on the host from the
        compact device result (global path indices; rays of other shards keep the sentinel).
        Slow, for checking only -- the C drop-in compute_paths() has its own dense writer."""
        torch = self.torch
        nrx, ntx, nb, npth = self.nrx, self.ntx, self.nb, self.num_paths
        shp = (nrx, ntx, nb, npth)

        def sent(shape):
            return np.full(shape, sentinel_u32, np.uint32).view(np.float32)

        out = {k: sent(shp) for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau", "dfs")}
        out["directions_rx"] = sent(shp + (3,))
        hit_tri = np.full((nb, ntx, npth), 0xFFFFFFFF, np.uint32)
        hit_theta = sent((nb, ntx, npth))
        fs0 = sent((nb, ntx, npth))
        state = sent((nb, ntx, npth, 11))
        counts = self.counts()
        for b in range(nb):
            n = int(counts[b + 1])
            if n == 0:
                continue
            h = {k: v.cpu().numpy() for k, v in self.hits(b, n).items()}
            ray = h["ray"].astype(np.int64) & 0xFFFFFFFF
            tx, p = self.global_path(ray)
            hit_tri[b, tx, p] = self.tri_order[h["tri"].view(np.uint32)]   # reference's flat index
            hit_theta[b, tx, p] = h["theta"]
            fs0[b, tx, p] = h["fs0"]
            for k, name in enumerate(("ox", "oy", "oz", "dx", "dy", "dz", "a_te_re", "a_te_im",
                                      "a_tm_re", "a_tm_im", "tau")):
                state[b, tx, p, k] = h[name]
            r = {k: v.cpu().numpy() for k, v in self.records(b, n).items()}
            for rx in range(nrx):
                ub = r["unblocked"][rx]
                for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau"):
                    out[k][rx, tx, b, p] = r[k][rx]
                out["dfs"][rx, tx[ub], b, p[ub]] = r["dfs"][rx][ub]
                for c, name in enumerate(("dirx", "diry", "dirz")):
                    out["directions_rx"][rx, tx[ub], b, p[ub], c] = r[name][rx][ub]
        out.update(hit_tri=hit_tri, hit_theta=hit_theta, fs0=fs0, state=state, counts=counts)
        return out
