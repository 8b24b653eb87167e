"""Python front end of the CPU oracle (oracle/hrt_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of hrt_oracle.c.  Imported by tests/,
bench.py's cpu_baseline leg and __graft_entry__.smoke(); never by hermespy-rt_amd/.

Self-contained on purpose: it has its own .hrt reader (format: src/scene.c:36-83 of the
reference) so the checker shares no code with the product under test.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SENTINEL_U32 = 0x7FC0DEAD
NO_HIT = 0xFFFFFFFF

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


class _Scene(C.Structure):
    _fields_ = [("num_meshes", C.c_uint32), ("num_tri", C.c_uint32), ("tri_vtx", _f32p),
                ("tri_mesh", _u32p), ("mesh_material", _u32p), ("mesh_velocity", _f32p)]


class _Chan(C.Structure):
    _fields_ = [(k, _f32p) for k in ("directions_rx", "directions_tx", "a_te_re", "a_te_im",
                                     "a_tm_re", "a_tm_im", "tau", "freq_shift")]


class _Rays(C.Structure):
    _fields_ = [("rays", _f32p), ("rays_active", _u8p)]


class _Opts(C.Structure):
    _fields_ = [("p_begin", C.c_uint64), ("p_end", C.c_uint64), ("p_stride", C.c_uint64),
                ("num_threads", C.c_int), ("hit_tri", _u32p), ("hit_theta", _f32p),
                ("live", _u64p), ("tests", _u64p), ("eta_table", _f32p), ("normals", _f32p),
                ("launch_dirs", _f32p), ("compact", C.c_int)]


def build(force=False):
    """Compile liboracle.so (gcc, seconds).  Building the checker is not using it."""
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "hrt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        # HRT_ORACLE_LIB: another build of the same source (the sanitizer build, `make -C oracle asan`)
        _lib = C.CDLL(os.environ.get("HRT_ORACLE_LIB") or build())
        _lib.hrt_oracle_compute_paths.restype = C.c_int
        _lib.hrt_oracle_compute_paths.argtypes = [
            C.POINTER(_Scene), _f32p, _f32p, _f32p, _f32p, C.c_float,
            C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
            C.POINTER(_Chan), C.POINTER(_Rays), C.POINTER(_Chan), C.POINTER(_Rays),
            C.POINTER(_Opts)]
        _lib.hrt_oracle_max_threads.restype = C.c_int
        # a GPU box shows all 256 cores of its host to a container that owns 16 of them: 256 OpenMP
        # threads there mean barriers between descheduled threads (HRT_ORACLE_THREADS overrides)
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        _lib.hrt_oracle_set_threads.argtypes = [C.c_int]
        _lib.hrt_oracle_set_threads.restype = None
        _lib.hrt_oracle_set_threads(int(os.environ.get("HRT_ORACLE_THREADS", min(16, cores))))
        _lib.hrt_oracle_libm.argtypes = [C.c_int, _f32p, _f32p, C.c_size_t]
        _lib.hrt_oracle_libm.restype = None
    return _lib


def read_hrt(path):
    """Parse a .hrt scene: 'HRT' | u32 num_meshes | per mesh: u32 nv | nv*3 f32 | u32 nt |
    nt*3 u32 | u32 material_index | 3 f32 velocity   (little-endian, unaligned)."""
    raw = open(path, "rb").read()
    if raw[:3] != b"HRT":
        raise ValueError("not an HRT file: %s" % path)
    pos = 3

    def take(dtype, n):
        nonlocal pos
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=pos)
        pos += a.nbytes
        return a.copy()

    nm = int(take("<u4", 1)[0])
    meshes = []
    for _ in range(nm):
        nv = int(take("<u4", 1)[0])
        vs = take("<f4", 3 * nv).reshape(nv, 3)
        nt = int(take("<u4", 1)[0])
        idx = take("<u4", 3 * nt).reshape(nt, 3)
        mat = int(take("<u4", 1)[0])
        vel = take("<f4", 3)
        meshes.append(dict(vs=vs, idx=idx, material_index=mat, velocity=vel))
    if pos != len(raw):
        raise ValueError("trailing bytes in %s" % path)
    return meshes


def flatten(meshes):
    """(mesh, face)-ordered triangle table with gathered vertices."""
    tri_vtx = np.concatenate([m["vs"][m["idx"]].reshape(-1, 9) for m in meshes]).astype(np.float32)
    tri_mesh = np.concatenate([np.full(len(m["idx"]), i, np.uint32) for i, m in enumerate(meshes)])
    tri_face = np.concatenate([np.arange(len(m["idx"]), dtype=np.uint32) for m in meshes])
    mesh_mat = np.array([m["material_index"] for m in meshes], np.uint32)
    mesh_vel = np.stack([m["velocity"] for m in meshes]).astype(np.float32)
    return dict(tri_vtx=np.ascontiguousarray(tri_vtx), tri_mesh=tri_mesh, tri_face=tri_face,
                mesh_material=mesh_mat, mesh_velocity=np.ascontiguousarray(mesh_vel))


def _sentinel(n, dtype=np.float32):
    if dtype == np.uint8:
        return np.full(n, 0xAD, dtype=np.uint8)
    return np.full(n, SENTINEL_U32, dtype=np.uint32).view(np.float32)


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def compute_paths(scene_path, rx_pos, tx_pos, rx_vel, tx_vel, f_ghz, num_paths, num_bounces,
                  zero_freq_shift=None, subset=None, num_threads=0, extras=True):
    """Run the oracle; returns the same dict layout as hermespy_rt_amd.abi.run_compute_paths
    plus 'extras' (hit_tri [nb, ntx, np], hit_theta, live [nb+1], tests, eta_table, normals,
    launch_dirs).  `subset` = (p_begin, p_end, p_stride) restricts the processed paths."""
    L = lib()
    flat = flatten(read_hrt(scene_path))
    T = flat["tri_vtx"].shape[0]
    rx_pos = np.ascontiguousarray(np.asarray(rx_pos, np.float32).reshape(-1, 3))
    tx_pos = np.ascontiguousarray(np.asarray(tx_pos, np.float32).reshape(-1, 3))
    nrx, ntx = rx_pos.shape[0], tx_pos.shape[0]
    rx_vel = np.ascontiguousarray(np.asarray(rx_vel, np.float32).reshape(nrx, 3))
    tx_vel = np.ascontiguousarray(np.asarray(tx_vel, np.float32).reshape(ntx, 3))
    npth, nb = int(num_paths), int(num_bounces)

    sc = _Scene(len(flat["mesh_material"]), T, _p(flat["tri_vtx"]), _p(flat["tri_mesh"], _u32p),
                _p(flat["mesh_material"], _u32p), _p(flat["mesh_velocity"]))

    def chan(n):
        d = dict(directions_rx=_sentinel(3 * n), directions_tx=_sentinel(3 * n),
                 a_te_re=_sentinel(n), a_te_im=_sentinel(n), a_tm_re=_sentinel(n),
                 a_tm_im=_sentinel(n), tau=_sentinel(n), freq_shift=_sentinel(n))
        c = _Chan(*[_p(d[k]) for k, _ in _Chan._fields_])
        return d, c

    n_los, n_scat = nrx * ntx, nrx * ntx * nb * npth
    los, los_c = chan(n_los)
    scat, scat_c = chan(n_scat)
    if zero_freq_shift is None:
        zero_freq_shift = ntx > 1
    if zero_freq_shift:
        scat["freq_shift"][:] = 0.0
    los_rays = _sentinel(6 * n_los)
    los_active = _sentinel(n_los // 8 + 1, np.uint8)
    n_rays_scat = ntx * (nb + 1) * npth
    scat_rays = _sentinel(6 * n_rays_scat)
    scat_active = _sentinel((ntx * nb + 1) * (npth // 8 + 1), np.uint8)
    lr = _Rays(_p(los_rays), _p(los_active, _u8p))
    sr = _Rays(_p(scat_rays), _p(scat_active, _u8p))

    opts = _Opts()
    if subset is not None:
        opts.p_begin, opts.p_end, opts.p_stride = [int(x) for x in subset]
    opts.num_threads = int(num_threads)
    ex = {}
    if extras:
        ex = dict(hit_tri=np.empty(nb * ntx * npth, np.uint32),
                  hit_theta=_sentinel(nb * ntx * npth),
                  live=np.zeros(nb + 1, np.uint64), tests=np.zeros(1, np.uint64),
                  eta_table=np.zeros(17 * 12, np.float32), normals=np.zeros(3 * T, np.float32),
                  launch_dirs=np.zeros(3 * npth, np.float32))
        opts.hit_tri = _p(ex["hit_tri"], _u32p)
        opts.hit_theta = _p(ex["hit_theta"])
        opts.live = _p(ex["live"], _u64p)
        opts.tests = _p(ex["tests"], _u64p)
        opts.eta_table = _p(ex["eta_table"])
        opts.normals = _p(ex["normals"])
        opts.launch_dirs = _p(ex["launch_dirs"])

    rc = L.hrt_oracle_compute_paths(C.byref(sc), _p(rx_pos), _p(tx_pos), _p(rx_vel), _p(tx_vel),
                                    C.c_float(f_ghz), nrx, ntx, npth, nb, C.byref(los_c),
                                    C.byref(lr), C.byref(scat_c), C.byref(sr), C.byref(opts))
    if rc != 0:
        raise RuntimeError("hrt_oracle_compute_paths failed: %d" % rc)

    shp = (nrx, ntx, nb, npth)
    res = dict(
        los={k: (v.reshape(nrx, ntx, 3) if k.startswith("directions") else v.reshape(nrx, ntx))
             for k, v in los.items()},
        scat={k: (v.reshape(*shp, 3) if k.startswith("directions") else v.reshape(shp))
              for k, v in scat.items()},
        los_rays=los_rays.reshape(n_los, 6), los_active=los_active,
        scat_rays=scat_rays.reshape(n_rays_scat, 6), scat_active=scat_active,
    )
    if extras:
        ex["hit_tri"] = ex["hit_tri"].reshape(nb, ntx, npth)
        ex["hit_theta"] = ex["hit_theta"].reshape(nb, ntx, npth)
        ex["eta_table"] = ex["eta_table"].reshape(17, 12)
        ex["normals"] = ex["normals"].reshape(T, 3)
        ex["launch_dirs"] = ex["launch_dirs"].reshape(npth, 3)
        ex["tests"] = int(ex["tests"][0])
        ex["tri_mesh"] = flat["tri_mesh"]
        ex["tri_face"] = flat["tri_face"]
        res["extras"] = ex
    return res


def compute_paths_subset(scene_path, rx_pos, tx_pos, rx_vel, tx_vel, f_ghz, num_paths, num_bounces,
                         subset, num_threads=0):
    """The oracle on a strided subset (p_begin, p_end, p_stride) of a launch set whose dense arrays
    would not fit the host (C5 at full size): arrays of extent n_sub = number of subset paths
    instead of num_paths, slot k <-> path p_begin + k * p_stride.  Returns dict(paths [n_sub],
    scat {a_*, tau, freq_shift: [nrx, ntx, nb, n_sub]; directions_rx [..., 3]} with sentinels in the
    slots the reference does not write, hit_tri / hit_theta [nb, ntx, n_sub], live [nb+1]).
    freq_shift is the record's own value (launch term minus the record's)."""
    L = lib()
    flat = flatten(read_hrt(scene_path))
    T = flat["tri_vtx"].shape[0]
    rx_pos = np.ascontiguousarray(np.asarray(rx_pos, np.float32).reshape(-1, 3))
    tx_pos = np.ascontiguousarray(np.asarray(tx_pos, np.float32).reshape(-1, 3))
    nrx, ntx = rx_pos.shape[0], tx_pos.shape[0]
    rx_vel = np.ascontiguousarray(np.asarray(rx_vel, np.float32).reshape(nrx, 3))
    tx_vel = np.ascontiguousarray(np.asarray(tx_vel, np.float32).reshape(ntx, 3))
    npth, nb = int(num_paths), int(num_bounces)
    pb, pe, ps = [int(x) for x in subset]
    pe = min(pe, npth)
    paths = np.arange(pb, pe, ps, dtype=np.int64)
    ns = len(paths)
    sc = _Scene(len(flat["mesh_material"]), T, _p(flat["tri_vtx"]), _p(flat["tri_mesh"], _u32p),
                _p(flat["mesh_material"], _u32p), _p(flat["mesh_velocity"]))

    def chan(n):
        d = dict(directions_rx=_sentinel(3 * n), directions_tx=_sentinel(3 * n),
                 a_te_re=_sentinel(n), a_te_im=_sentinel(n), a_tm_re=_sentinel(n),
                 a_tm_im=_sentinel(n), tau=_sentinel(n), freq_shift=_sentinel(n))
        return d, _Chan(*[_p(d[k]) for k, _ in _Chan._fields_])

    los, los_c = chan(nrx * ntx)
    scat, scat_c = chan(nrx * ntx * nb * ns)
    los_rays, los_active = _sentinel(6 * nrx * ntx), _sentinel(nrx * ntx // 8 + 1, np.uint8)
    lr = _Rays(_p(los_rays), _p(los_active, _u8p))
    sr = _Rays(None, None)
    opts = _Opts()
    opts.p_begin, opts.p_end, opts.p_stride = pb, pe, ps
    opts.num_threads = int(num_threads)
    opts.compact = 1
    hit_tri = np.empty(nb * ntx * ns, np.uint32)
    hit_theta = _sentinel(nb * ntx * ns)
    live = np.zeros(nb + 1, np.uint64)
    opts.hit_tri, opts.hit_theta, opts.live = _p(hit_tri, _u32p), _p(hit_theta), _p(live, _u64p)
    rc = L.hrt_oracle_compute_paths(C.byref(sc), _p(rx_pos), _p(tx_pos), _p(rx_vel), _p(tx_vel),
                                    C.c_float(f_ghz), nrx, ntx, npth, nb, C.byref(los_c),
                                    C.byref(lr), C.byref(scat_c), C.byref(sr), C.byref(opts))
    if rc != 0:
        raise RuntimeError("hrt_oracle_compute_paths failed: %d" % rc)
    shp = (nrx, ntx, nb, ns)
    return dict(paths=paths,
                scat={k: (v.reshape(*shp, 3) if k.startswith("directions") else v.reshape(shp))
                      for k, v in scat.items()},
                los={k: (v.reshape(nrx, ntx, 3) if k.startswith("directions") else v.reshape(nrx, ntx))
                     for k, v in los.items()},
                hit_tri=hit_tri.reshape(nb, ntx, ns), hit_theta=hit_theta.reshape(nb, ntx, ns), live=live,
                tri_mesh=flat["tri_mesh"], tri_face=flat["tri_face"])


LIBM_FN = dict(sinf=0, cosf=1, expf=2, acosf=3, incidence_angle=4)


def host_libm(fn, x):
    """Evaluate the HOST libm's float function (what the reference calls) over an array."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().hrt_oracle_libm(LIBM_FN[fn], _p(x), _p(out), x.size)
    return out
