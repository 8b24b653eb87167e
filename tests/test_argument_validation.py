"""hrt_compute_paths_ex / hrt_problem_create validate their arguments BEFORE touching the device
(the reference validates nothing: SURVEY Q15 asks for validation in `_ex`, behaviour in range
unchanged): bad counts, NULL pointers, a material index beyond the 17-entry table, a vertex index
beyond the mesh, a non-positive frequency -> HRT_E_INVALID with a message, no crash, no exit."""
import ctypes as C

import numpy as np
import pytest

from hermespy_rt_amd import abi

from . import scenes_gen as G

V3 = C.POINTER(abi.Vec3)


def _call(lib, scene, nrx=1, ntx=1, np_=100, nb=2, f=3.0, null_scene=False, null_los=False):
    rx = np.zeros((max(nrx, 1), 3), np.float32)
    tx = np.ones((max(ntx, 1), 3), np.float32)
    los, scat = abi.ChannelInfo(), abi.ChannelInfo()
    return lib.hrt_compute_paths_ex(
        None if null_scene else C.byref(scene), rx.ctypes.data_as(V3), tx.ctypes.data_as(V3),
        rx.ctypes.data_as(V3), tx.ctypes.data_as(V3), C.c_float(f), nrx, ntx, np_, nb,
        None if null_los else C.byref(los), None, C.byref(scat), None, None)


@pytest.fixture
def scene(product_lib, tmp_path):
    p = str(tmp_path / "s.hrt")
    v, f = G._box([0, 0, 2], [4, 4, 4])
    G.write_hrt(p, [dict(vs=v, idx=f, material_index=1, velocity=[0, 0, 0])])
    s = product_lib.scene_load(p.encode())
    yield s
    abi.free_scene(s)


@pytest.mark.parametrize("kw,needle", [
    (dict(np_=0), b"num_rays"), (dict(nb=0), b"num_bounces"), (dict(nb=70000), b"num_bounces"),
    (dict(null_scene=True), b"NULL"), (dict(null_los=True), b"NULL"),
    (dict(nrx=0), b"num_rx"), (dict(ntx=0), b"num_rx/num_tx"), (dict(f=0.0), b"frequency"),
    (dict(f=float("nan")), b"frequency"),
])
def test_bad_arguments_are_refused(product_lib, scene, kw, needle):
    assert _call(product_lib, scene, **kw) == -1          # HRT_E_INVALID
    assert needle in product_lib.hrt_last_error()


def test_bad_material_and_vertex_indices_are_refused(product_lib, scene):
    m = scene.meshes[0]
    keep = m.material_index
    m.material_index = 17
    assert _call(product_lib, scene) == -1 and b"material_index" in product_lib.hrt_last_error()
    m.material_index = keep
    old = m.is_[5]
    m.is_[5] = m.num_vertices          # one past the last vertex
    assert _call(product_lib, scene) == -1 and b"vertex index" in product_lib.hrt_last_error()
    m.is_[5] = old


def test_path_list_entry_validates_too(product_lib, scene):
    rx = np.zeros((1, 3), np.float32)
    pl = abi.PathList()
    product_lib.hrt_compute_paths_list.restype = C.c_int

    def call(np_=100, nb=2, out=pl, sc=scene):
        return product_lib.hrt_compute_paths_list(
            C.byref(sc) if sc is not None else None, rx.ctypes.data_as(V3), rx.ctypes.data_as(V3),
            rx.ctypes.data_as(V3), rx.ctypes.data_as(V3), C.c_float(3.0), C.c_size_t(1), C.c_size_t(1),
            C.c_size_t(np_), C.c_size_t(nb), C.c_int(0), C.byref(out) if out is not None else None, None)

    assert call(np_=0) == -1 and b"num_rays" in product_lib.hrt_last_error()
    assert call(nb=70000) == -1 and b"num_bounces" in product_lib.hrt_last_error()
    assert call(out=None) == -1 and b"NULL" in product_lib.hrt_last_error()
    assert call(sc=None) == -1 and b"NULL" in product_lib.hrt_last_error()
    product_lib.hrt_path_list_free(C.byref(pl))      # freeing an empty list is fine
    product_lib.hrt_path_list_free(None)


def test_unknown_tune_key_is_refused(product_lib, scene, monkeypatch):
    """HRT_TUNE carries the developer switches; a key the library does not know is an error, not ignored"""
    monkeypatch.setenv("HRT_TUNE", "variant=2,no_such_switch=1")
    assert _call(product_lib, scene) == -1 and b"HRT_TUNE" in product_lib.hrt_last_error()
