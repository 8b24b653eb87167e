"""ctypes mirror of the reference's C ABI for the compute_paths hot path.

This is the host-side (Python) statement of the drop-in boundary.  Every struct below is
byte-compatible with the reference header it cites, so driver code written against these
declarations works with any library that exports the reference's three entry points -- the
product loads its own (lib.py); tests/ use the same declarations to drive the checker builds.

Reference interface mirrored here:
  Vec3            inc/vec3.h:6-8
  Ray             inc/ray.h:6-9
  Mesh, Scene     inc/scene.h:10-32
  ChannelInfo     inc/compute_paths.h:13-23
  RaysInfo        inc/compute_paths.h:26-30
  compute_paths   inc/compute_paths.h:59-74
  scene_load      inc/scene.h:105   (returns Scene BY VALUE)
  scene_save      inc/scene.h:95
"""
import ctypes as C

import numpy as np

c_float_p = C.POINTER(C.c_float)


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Ray(C.Structure):
    _fields_ = [("o", Vec3), ("d", Vec3)]


class Mesh(C.Structure):
    _fields_ = [
        ("num_vertices", C.c_uint32),
        ("vs", C.POINTER(Vec3)),
        ("num_triangles", C.c_uint32),
        ("is_", C.POINTER(C.c_uint32)),
        ("material_index", C.c_uint32),
        ("velocity", Vec3),
        ("ns", C.POINTER(Vec3)),
    ]


class Scene(C.Structure):
    _fields_ = [("num_meshes", C.c_uint32), ("meshes", C.POINTER(Mesh))]


class ChannelInfo(C.Structure):
    _fields_ = [
        ("num_rays", C.c_uint32),
        ("directions_rx", C.POINTER(Vec3)),
        ("directions_tx", C.POINTER(Vec3)),
        ("a_te_re", c_float_p),
        ("a_te_im", c_float_p),
        ("a_tm_re", c_float_p),
        ("a_tm_im", c_float_p),
        ("tau", c_float_p),
        ("freq_shift", c_float_p),
    ]


class RaysInfo(C.Structure):
    _fields_ = [
        ("num_bounces", C.c_uint32),
        ("num_rays", C.c_uint32),
        ("rays", C.POINTER(Ray)),
        ("rays_active", C.POINTER(C.c_uint8)),
    ]


assert C.sizeof(Vec3) == 12 and C.sizeof(Ray) == 24
assert C.sizeof(Mesh) == 56 and C.sizeof(Scene) == 16
assert C.sizeof(ChannelInfo) == 72 and C.sizeof(RaysInfo) == 24

#: bit pattern used to pre-fill every output buffer so "slot not written" is observable
#: (the reference leaves slots of dead rays untouched).
SENTINEL_U32 = 0x7FC0DEAD


def bind_c_abi(lib):
    """Declare argtypes/restype of compute_paths / scene_load / scene_save (the reference's
    signatures, inc/compute_paths.h:59-74, inc/scene.h:95-105) on the loaded library `lib`."""
    lib.scene_load.argtypes = [C.c_char_p]
    lib.scene_load.restype = Scene
    lib.scene_save.argtypes = [C.POINTER(Scene), C.c_char_p]
    lib.scene_save.restype = None
    lib.compute_paths.argtypes = [
        C.POINTER(Scene),
        C.POINTER(Vec3), C.POINTER(Vec3), C.POINTER(Vec3), C.POINTER(Vec3),
        C.c_float,
        C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
        C.POINTER(ChannelInfo), C.POINTER(RaysInfo),
        C.POINTER(ChannelInfo), C.POINTER(RaysInfo),
    ]
    lib.compute_paths.restype = None
    return lib


def _sentinel(n, dtype=np.float32):
    if dtype == np.uint8:
        return np.full(n, 0xAD, dtype=np.uint8)
    return np.full(n, SENTINEL_U32, dtype=np.uint32).view(np.float32)


def _vec3_arg(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(n, 3))
    return a, a.ctypes.data_as(C.POINTER(Vec3))


def free_scene(scene):
    """inc/scene.h:72-86 (static inline in the reference header; uses free())."""
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for i in range(scene.num_meshes):
        m = scene.meshes[i]
        for p in (m.vs, m.is_, m.ns):
            libc.free(C.cast(p, C.c_void_p))
    libc.free(C.cast(scene.meshes, C.c_void_p))


def scene_to_numpy(scene):
    """Deep-copy a loaded Scene into python lists of numpy arrays (for inspection/tests)."""
    out = []
    for i in range(scene.num_meshes):
        m = scene.meshes[i]
        vs = np.ctypeslib.as_array(C.cast(m.vs, c_float_p), shape=(m.num_vertices, 3)).copy()
        idx = np.ctypeslib.as_array(m.is_, shape=(m.num_triangles, 3)).copy()
        ns = None
        if m.ns:
            ns = np.ctypeslib.as_array(C.cast(m.ns, c_float_p), shape=(m.num_triangles, 3)).copy()
        out.append(dict(vs=vs, idx=idx, material_index=int(m.material_index),
                        velocity=np.array([m.velocity.x, m.velocity.y, m.velocity.z], np.float32),
                        ns=ns))
    return out


def run_compute_paths(lib, scene_path, rx_pos, tx_pos, rx_vel, tx_vel, f_ghz, num_paths,
                      num_bounces, zero_freq_shift=None, with_rays=True, stats=None, interleaved=False):
    """Call `lib.compute_paths` the way the reference's own callers do
    (compute_paths_pybind11.cpp:99-186, test/test.c:10-75) and return every output as numpy.

    All outputs are pre-filled with SENTINEL_U32 so unwritten slots are visible.  The scatter
    freq_shift is zero-filled when num_tx > 1 (the reference memcpy-replicates caller memory
    there) unless `zero_freq_shift` overrides.  rays_active is allocated with
    the size the callee really needs, (ntx*nb+1)*(np/8+1).
    """
    rx_pos = np.asarray(rx_pos, np.float32).reshape(-1, 3)
    tx_pos = np.asarray(tx_pos, np.float32).reshape(-1, 3)
    nrx, ntx = rx_pos.shape[0], tx_pos.shape[0]
    npth, nb = int(num_paths), int(num_bounces)
    rxp, rxp_c = _vec3_arg(rx_pos, nrx)
    txp, txp_c = _vec3_arg(tx_pos, ntx)
    rxv, rxv_c = _vec3_arg(rx_vel, nrx)
    txv, txv_c = _vec3_arg(tx_vel, ntx)

    def chan(n, n_dir_tx):
        d = dict(directions_rx=_sentinel(3 * n), directions_tx=_sentinel(3 * n_dir_tx),
                 a_te_re=_sentinel(n), a_te_im=_sentinel(n), a_tm_re=_sentinel(n),
                 a_tm_im=_sentinel(n), tau=_sentinel(n), freq_shift=_sentinel(n))
        ci = ChannelInfo()
        if interleaved:
            # hrt_compute_paths_interleaved: re and im of one polarisation share an array of 2 n floats
            # (what a complex64 array is); the result dict still shows them as two (strided) planes
            for pol in ("a_te", "a_tm"):
                both = _sentinel(2 * n)
                d[pol + "_both"] = both
                d[pol + "_re"], d[pol + "_im"] = both[0::2], both[1::2]
        for k, v in d.items():
            if k.endswith("_both"):
                continue
            ptr_t = C.POINTER(Vec3) if k.startswith("directions") else c_float_p
            if interleaved and k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im"):
                base = d[k[:4] + "_both"].ctypes.data + (4 if k.endswith("_im") else 0)
                setattr(ci, k, C.cast(base, c_float_p))
            else:
                setattr(ci, k, v.ctypes.data_as(ptr_t))
        return d, ci

    n_los = nrx * ntx
    n_scat = nrx * ntx * nb * npth
    los, los_c = chan(n_los, n_los)
    scat, scat_c = chan(n_scat, n_scat)
    los_c.num_rays = 1
    scat_c.num_rays = (nb * npth) & 0xFFFFFFFF
    if zero_freq_shift is None:
        zero_freq_shift = ntx > 1
    if zero_freq_shift:
        scat["freq_shift"][:] = 0.0

    los_rays = _sentinel(6 * n_los)
    los_active = _sentinel(n_los // 8 + 1, np.uint8)
    n_rays_scat = ntx * (nb + 1) * npth
    n_act_scat = (ntx * nb + 1) * (npth // 8 + 1)
    scat_rays = _sentinel(6 * n_rays_scat)
    scat_active = _sentinel(n_act_scat, np.uint8)
    lr = RaysInfo(1, 1, los_rays.ctypes.data_as(C.POINTER(Ray)),
                  los_active.ctypes.data_as(C.POINTER(C.c_uint8)))
    sr = RaysInfo(nb + 1, npth & 0xFFFFFFFF, scat_rays.ctypes.data_as(C.POINTER(Ray)),
                  scat_active.ctypes.data_as(C.POINTER(C.c_uint8)))

    scene = lib.scene_load(str(scene_path).encode())
    try:
        if stats is not None or not with_rays or interleaved:
            # product-only entry points: status code, optional RaysInfo, counters
            entry = lib.hrt_compute_paths_interleaved if interleaved else lib.hrt_compute_paths_ex
            rc = entry(C.byref(scene), rxp_c, txp_c, rxv_c, txv_c,
                                          C.c_float(f_ghz), nrx, ntx, npth, nb, C.byref(los_c),
                                          C.byref(lr) if with_rays else None, C.byref(scat_c),
                                          C.byref(sr) if with_rays else None,
                                          C.byref(stats) if stats is not None else None)
            if rc != 0:
                raise RuntimeError("hrt_compute_paths_ex failed (%d): %s" % (rc, lib.hrt_last_error().decode()))
        else:
            lib.compute_paths(C.byref(scene), rxp_c, txp_c, rxv_c, txv_c, C.c_float(f_ghz),
                              nrx, ntx, npth, nb, C.byref(los_c), C.byref(lr),
                              C.byref(scat_c), C.byref(sr))
        normals = [m["ns"] for m in scene_to_numpy(scene)]
    finally:
        free_scene(scene)

    shp = (nrx, ntx, nb, npth)
    los = {k: v for k, v in los.items() if not k.endswith("_both")}
    scat = {k: v for k, v in scat.items() if not k.endswith("_both")}
    res = dict(
        los={k: (v.reshape(nrx, ntx, 3) if k.startswith("directions") else v.reshape(nrx, ntx))
             for k, v in los.items()},
        scat={k: (v.reshape(*shp, 3) if k.startswith("directions") else v.reshape(shp))
              for k, v in scat.items()},
        los_rays=los_rays.reshape(n_los, 6), los_active=los_active,
        scat_rays=scat_rays.reshape(n_rays_scat, 6), scat_active=scat_active,
        normals=normals,
    )
    return res


def written(a):
    """Boolean mask of slots whose bit pattern is not the sentinel."""
    return a.view(np.uint32) != SENTINEL_U32


class PathList(C.Structure):
    """include/hermespy_rt.h hrt_path_list"""
    _fields_ = [
        ("num", C.c_uint64), ("num_rx", C.c_uint32), ("num_tx", C.c_uint32),
        ("rx", C.POINTER(C.c_uint32)), ("tx", C.POINTER(C.c_uint32)), ("bounce", C.POINTER(C.c_uint32)),
        ("path", C.POINTER(C.c_uint64)),
        ("a_te_re", c_float_p), ("a_te_im", c_float_p), ("a_tm_re", c_float_p), ("a_tm_im", c_float_p),
        ("tau", c_float_p), ("direction_rx", C.POINTER(Vec3)), ("freq_shift", c_float_p),
        ("unblocked", C.POINTER(C.c_uint8)), ("mesh", C.POINTER(C.c_uint32)), ("face", C.POINTER(C.c_uint32)),
        ("los", c_float_p),
    ]


def run_compute_paths_list(lib, scene_path, rx_pos, tx_pos, rx_vel, tx_vel, f_ghz, num_paths,
                           num_bounces, include_blocked=False, stats=None):
    """hrt_compute_paths_list through ctypes -> dict of numpy arrays (copies; the C list is freed)."""
    rx_pos = np.asarray(rx_pos, np.float32).reshape(-1, 3)
    tx_pos = np.asarray(tx_pos, np.float32).reshape(-1, 3)
    nrx, ntx = rx_pos.shape[0], tx_pos.shape[0]
    _, rxp = _vec3_arg(rx_pos, nrx)
    _, txp = _vec3_arg(tx_pos, ntx)
    rxv_a, rxv = _vec3_arg(rx_vel, nrx)
    txv_a, txv = _vec3_arg(tx_vel, ntx)
    lib.hrt_compute_paths_list.restype = C.c_int
    lib.hrt_path_list_free.restype = None
    pl = PathList()
    scene = lib.scene_load(str(scene_path).encode())
    try:
        rc = lib.hrt_compute_paths_list(C.byref(scene), rxp, txp, rxv, txv, C.c_float(f_ghz),
                                        C.c_size_t(nrx), C.c_size_t(ntx), C.c_size_t(int(num_paths)),
                                        C.c_size_t(int(num_bounces)), C.c_int(1 if include_blocked else 0),
                                        C.byref(pl), C.byref(stats) if stats is not None else None)
        if rc != 0:
            raise RuntimeError("hrt_compute_paths_list failed (%d): %s" % (rc, lib.hrt_last_error().decode()))
        n = int(pl.num)

        def arr(ptr, dtype, shape):
            if n == 0:
                return np.zeros(shape, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)),
                                         shape=(int(np.prod(shape)) * np.dtype(dtype).itemsize,)).view(dtype).reshape(shape).copy()

        out = dict(rx=arr(pl.rx, np.uint32, (n,)), tx=arr(pl.tx, np.uint32, (n,)),
                   bounce=arr(pl.bounce, np.uint32, (n,)), path=arr(pl.path, np.uint64, (n,)),
                   tau=arr(pl.tau, np.float32, (n,)), direction_rx=arr(pl.direction_rx, np.float32, (n, 3)),
                   freq_shift=arr(pl.freq_shift, np.float32, (n,)), unblocked=arr(pl.unblocked, np.uint8, (n,)).astype(bool),
                   mesh=arr(pl.mesh, np.uint32, (n,)), face=arr(pl.face, np.uint32, (n,)))
        for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im"):
            out[k] = arr(getattr(pl, k), np.float32, (n,))
        out["los"] = np.ctypeslib.as_array(pl.los, shape=(nrx * ntx * 8,)).copy().reshape(nrx, ntx, 8)
    finally:
        lib.hrt_path_list_free(C.byref(pl))
        free_scene(scene)
    return out
