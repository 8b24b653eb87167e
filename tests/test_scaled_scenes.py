"""The bundled scenes translated far from the origin (UTM-like coordinates, where one float ulp
is centimetres) and rescaled (a millimetre box a kilometre away, kilometre cars): the reference's
arithmetic loses precision there, and parity means losing it the same way; the culling proof's
tolerances scale with the distances involved, not with a scene size it assumes.
CPU part: oracle against the LIVE reference.  GPU part: product against the oracle."""
import os

import numpy as np
import pytest

from hermespy_rt_amd import abi
from oracle import oracle

from . import configs as K
from . import scenes_gen as G
from .parity import compare_dense

CASES = {
    "canyon_utm": ("simple_street_canyon_with_cars.hrt", "C3", [5e5, 4e6, 300.0], 1.0),
    "canyon_2km": ("simple_street_canyon_with_cars.hrt", "C3", [-2000.0, 1500.0, 50.0], 1.0),
    "box_mm_at_1km": ("box.hrt", "C1", [1e3, -1e3, 10.0], 1e-3),
    "cars_km": ("2cars.hrt", "C4", [0.0, 0.0, 0.0], 1e3),
}


def _case(tmp, name):
    scene, base, off, scale = CASES[name]
    off = np.asarray(off, np.float64)
    meshes = [dict(vs=(np.asarray(m["vs"], np.float64) * scale + off).astype(np.float32), idx=m["idx"],
                   material_index=m["material_index"], velocity=m["velocity"])
              for m in oracle.read_hrt(os.path.join(K.SC, scene))]
    p = os.path.join(str(tmp), name + ".hrt")
    G.write_hrt(p, meshes)
    b = K.ALL[base]
    rx = (np.asarray(b["rx_pos"], np.float64) * scale + off).astype(np.float32).tolist()
    tx = (np.asarray(b["tx_pos"], np.float64) * scale + off).astype(np.float32).tolist()
    return G.cfg(p, rx, tx, 6001, 3, f=3.5)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_equals_reference_scaled(ref_lib, name, tmp_path):
    c = _case(tmp_path, name)
    ref = abi.run_compute_paths(ref_lib, *K.args(c))
    got = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), st


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_product_equals_oracle_scaled(product_lib, name, tmp_path):
    c = _case(tmp_path, name)
    got = abi.run_compute_paths(product_lib, *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), st
    assert int(np.asarray(ref["extras"]["live"])[1]) > 0
