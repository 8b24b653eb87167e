"""Edge sizes the bundled configurations do not reach: 1 / 7 / 9 / 64 / 65 / 257 rays (less than a
byte of the active mask, less and one more than a wavefront, one more than a workgroup), a scene
of ONE triangle, a mesh with ZERO triangles next to a real one, a scene no ray can hit (every
list empty after launch 0), 32 bounces and 40 bounces (the reference's loop has no cap, src/compute_paths.c:591;
the library's per-launch tables are sized by the bounce count, only hrt_stats.live keeps its 34 slots).
CPU part: oracle against the LIVE reference.  GPU part: product against the oracle."""
import os

import numpy as np
import pytest

from hermespy_rt_amd import abi
from oracle import oracle

from . import configs as K
from . import scenes_gen as G
from .parity import compare_dense


def _scenes(tmp):
    tmp = str(tmp)
    one = os.path.join(tmp, "one_triangle.hrt")
    G.write_hrt(one, [dict(vs=[[-3, -3, 0], [3, -3, 0], [0, 4, 0]], idx=[[0, 1, 2]], material_index=3,
                           velocity=[0, 0, 1])])
    v, f = G._box([0, 0, 2], [6, 5, 4])
    hole = os.path.join(tmp, "empty_mesh.hrt")
    G.write_hrt(hole, [dict(vs=np.zeros((0, 3), np.float32), idx=np.zeros((0, 3), np.uint32),
                            material_index=0, velocity=[0, 0, 0]),
                       dict(vs=v, idx=f, material_index=5, velocity=[0, 0, 0]),
                       dict(vs=[[9, 9, 9]], idx=np.zeros((0, 3), np.uint32), material_index=2,
                            velocity=[1, 2, 3])])
    far = os.path.join(tmp, "unreachable.hrt")   # a tiny triangle 10 km away, edge-on
    G.write_hrt(far, [dict(vs=[[1e4, 0, 0], [1e4, 1e-3, 0], [1e4, 0, 1e-3]], idx=[[0, 1, 2]],
                           material_index=1, velocity=[0, 0, 0])])
    return one, hole, far


def _cases(tmp):
    one, hole, far = _scenes(tmp)
    out = {}
    for n in (1, 7, 9, 64, 65, 257):
        out["box_%d_rays" % n] = K.cfg("box.hrt", [[2, 1, 1.5], [-1, -2, 3]], [[0, 0, 2.5]], 3.0, n, 3)
    out["one_triangle"] = G.cfg(one, [[0.5, 0.2, 2.0], [1, 1, -1.0]], [[0, 0, 3.0]], 3000, 2)
    out["empty_mesh"] = G.cfg(hole, [[1, 1, 1.0]], [[-1, 0.5, 2.0], [2, -1, 3.0]], 2000, 4)
    out["unreachable"] = G.cfg(far, [[3, 0, 0]], [[0, 0, 0]], 1500, 3)
    out["box_32_bounces"] = K.cfg("box.hrt", [[2, 1, 1.5]], [[0, 0, 2.5]], 3.0, 300, 32)
    out["box_40_bounces"] = K.cfg("box.hrt", [[2, 1, 1.5]], [[0, 0, 2.5]], 3.0, 300, 40)
    out["box_70_bounces_2tx"] = K.cfg("box.hrt", [[2, 1, 1.5]], [[0, 0, 2.5], [1, -1, 1.0]], 3.0, 130, 70)
    return out


NAMES = ["box_1_rays", "box_7_rays", "box_9_rays", "box_64_rays", "box_65_rays", "box_257_rays",
         "one_triangle", "empty_mesh", "unreachable", "box_32_bounces", "box_40_bounces", "box_70_bounces_2tx"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_reference_edge_sizes(ref_lib, name, tmp_path):
    c = _cases(tmp_path)[name]
    ref = abi.run_compute_paths(ref_lib, *K.args(c))
    got = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), st


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_product_equals_oracle_edge_sizes(product_lib, name, tmp_path):
    c = _cases(tmp_path)[name]
    got = abi.run_compute_paths(product_lib, *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), st
    if name == "unreachable":
        assert int(np.asarray(ref["extras"]["live"])[1]) == 0
