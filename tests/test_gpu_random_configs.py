"""Randomised parity (seeded): endpoints inside/outside the geometry, frequencies 0.5-100 GHz,
1-3 TX, 1-5 RX, 1-12 bounces, odd ray counts, random velocities, on all bundled scenes --
the product (drop-in C ABI) against the oracle, every output array."""
import numpy as np
import pytest

from hermespy_rt_amd import abi
from oracle import oracle

from . import configs as K
from .parity import compare_dense

pytestmark = pytest.mark.gpu
SCENES = ["box.hrt", "simple_reflector.hrt", "2cars.hrt", "simple_street_canyon_with_cars.hrt"]
BOUNDS = {"box.hrt": ([-4.5, -4.5, 0.2], [4.5, 4.5, 4.8]), "simple_reflector.hrt": ([-1, -1, 0.05], [1, 1, 2]),
          "2cars.hrt": ([-8, -25, 0.3], [8, 25, 6]), "simple_street_canyon_with_cars.hrt": ([-70, -8, 0.5], [70, 8, 25])}


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    scene = SCENES[seed % 4]
    lo, hi = BOUNDS[scene]
    nrx, ntx = int(rng.integers(1, 6)), int(rng.integers(1, 4))
    c = K.cfg(scene, rng.uniform(lo, hi, (nrx, 3)).tolist(), rng.uniform(lo, hi, (ntx, 3)).tolist(),
              float(rng.choice([0.5, 2.4, 3.5, 28.0, 60.0, 100.0])), int(rng.integers(50, 6000)),
              int(rng.integers(1, 13)))
    if seed % 3:
        c["rx_vel"] = rng.uniform(-30, 30, (nrx, 3)).tolist()
        c["tx_vel"] = rng.uniform(-30, 30, (ntx, 3)).tolist()
    return c


@pytest.mark.parametrize("seed", range(16))
def test_random_config(product_lib, seed):
    c = _case(seed)
    got = abi.run_compute_paths(product_lib, *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), st


@pytest.mark.parametrize("nrx,ntx", [(40, 1), (70, 2)])
def test_many_receivers(product_lib, nrx, ntx):
    """More receivers than lanes in a wave / than any bundled configuration: the shadow-trace
    kinds, the theta carry across the RX loop and the per-RX record blocks scale with num_rx."""
    rng = np.random.default_rng(5 + nrx)
    lo, hi = BOUNDS["simple_street_canyon_with_cars.hrt"]
    c = K.cfg("simple_street_canyon_with_cars.hrt", rng.uniform(lo, hi, (nrx, 3)).tolist(),
              rng.uniform(lo, hi, (ntx, 3)).tolist(), 3.5, 700, 3)
    got = abi.run_compute_paths(product_lib, *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), st
