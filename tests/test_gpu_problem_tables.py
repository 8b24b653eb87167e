"""The host-side tables of the product against the oracle's restatement of the reference:
  * eta table (precompute_materials, src/compute_paths.c:171-206; csrc/host/materials.c) for all
    17 ITU-R P.2040 materials at 3.0 / 3.5 / 70 GHz, bit for bit;
  * triangle normals (precompute_normals, :208-224): what hrt_problem_normals reports and what
    compute_paths() leaves in the caller's Scene (mesh->ns), against the oracle's and -- where
    oracle/_ref is built -- the real reference's;
  * the table order map (hrt_problem_tri_order) is a permutation consistent with hrt_problem_tri_ids.
"""
import ctypes as C
import os

import numpy as np
import pytest

from hermespy_rt_amd import abi, lib
from oracle import oracle

from . import configs as K
from . import scenes_gen as G
from .parity import assert_bit_equal

pytestmark = pytest.mark.gpu


def _all_materials_scene(tmp):
    p = os.path.join(str(tmp), "mats.hrt")
    G.room_with_clutter(p, 24, seed=11, tilt=True)      # 25 meshes: materials 1, 0..16, ...
    used = {m["material_index"] for m in oracle.read_hrt(p)}
    assert used == set(range(17))
    return p


@pytest.mark.parametrize("f_ghz", [3.0, 3.5, 70.0])
def test_eta_table_all_materials(tmp_path, f_ghz):
    from hermespy_rt_amd.device import Tracer
    p = _all_materials_scene(tmp_path)
    c = G.cfg(p, [[5, 3, 1.5]], [[-10, 5, 6.0]], 512, 1, f=f_ghz)
    ref = oracle.compute_paths(*K.args(c))
    tr = Tracer(p, c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], f_ghz, 512, 1)
    eta = np.zeros((17, 12), np.float32)
    lib.check(tr.L.hrt_problem_eta_table(tr.problem, eta.ctypes.data_as(C.POINTER(C.c_float))))
    assert_bit_equal(eta, ref["extras"]["eta_table"], "eta_table @ %g GHz" % f_ghz)
    assert np.all(np.isfinite(eta)) and np.all(eta[:, 8] > 0)      # |eta| of every material
    if f_ghz == 3.0:
        # anchors of SURVEY.md 8(c): concrete (material 1) at 3 GHz, field order of :125-132
        e = eta[1]
        assert np.float32(e[0]) == np.float32(5.23999977) and np.float32(e[4]) == np.float32(0.653726876)
        assert np.float32(e[1]) == np.float32(2.29353666) and np.float32(e[8]) == np.float32(5.28062105)
        assert np.float32(e[11]) == np.float32(0.5)
    tr.close()


@pytest.mark.parametrize("name", ["C3", "C4", "gen"])
def test_normals_product_vs_oracle(product_lib, tmp_path, name):
    from hermespy_rt_amd.device import Tracer
    if name == "gen":
        p = _all_materials_scene(tmp_path)
        c = G.cfg(p, [[5, 3, 1.5]], [[-10, 5, 6.0]], 256, 1)
    else:
        c = K.small(getattr(K, name), 256)
    ref = oracle.compute_paths(*K.args(c))
    n_ref = ref["extras"]["normals"]
    # (1) what the drop-in leaves in the caller's Scene
    got = abi.run_compute_paths(product_lib, *K.args(c))
    assert_bit_equal(np.concatenate(got["normals"]), n_ref, "mesh->ns left by compute_paths")
    # (2) the device-resident problem's own report, and the order map
    tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"], 256, 1)
    T = tr.num_tri
    n = np.zeros((T, 3), np.float32)
    lib.check(tr.L.hrt_problem_normals(tr.problem, n.ctypes.data_as(C.POINTER(C.c_float))))
    assert_bit_equal(n, n_ref, "hrt_problem_normals")
    assert sorted(tr.tri_order[:T].tolist()) == list(range(T))
    mesh, face = np.zeros(T, np.uint32), np.zeros(T, np.uint32)
    lib.check(tr.L.hrt_problem_tri_ids(tr.problem, mesh.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       face.ctypes.data_as(C.POINTER(C.c_uint32))))
    tm, tf = np.asarray(ref["extras"]["tri_mesh"]), np.asarray(ref["extras"]["tri_face"])
    assert np.array_equal(mesh, tm[tr.tri_order[:T]]) and np.array_equal(face, tf[tr.tri_order[:T]])
    tr.close()


def test_normals_product_vs_live_reference(product_lib, ref_lib):
    c = K.small(K.C3, 128)
    got = abi.run_compute_paths(product_lib, *K.args(c))
    ref = abi.run_compute_paths(ref_lib, *K.args(c))
    assert_bit_equal(np.concatenate(got["normals"]), np.concatenate(ref["normals"]), "mesh->ns vs reference")


def test_two_tx_sign_of_zero_freq_shift(product_lib):
    """2 TX, zero velocities, one bounce -- the reference's `+= 0` (Q10) lands on the
    ray's own slot for every TX and flips -0 to +0; compared with the sign bit."""
    c = dict(K.small(K.C4, 3001), num_bounces=1)
    got = abi.run_compute_paths(product_lib, *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    assert_bit_equal(got["scat"]["freq_shift"], ref["scat"]["freq_shift"], "scat.freq_shift (sign bits)")
    c2 = dict(K.small(K.C5, 1501), num_bounces=3)
    got = abi.run_compute_paths(product_lib, *K.args(c2))
    ref = oracle.compute_paths(*K.args(c2))
    assert_bit_equal(got["scat"]["freq_shift"], ref["scat"]["freq_shift"], "scat.freq_shift 8 TX (sign bits)")
