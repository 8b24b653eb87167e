"""A slice of the parity fuzzers (tests/fuzz_parity.py) as a regular GPU test: 32 random triangle
soups (scales 5 cm .. 300 m, coplanar clusters, slivers, moving meshes, all materials), product
against oracle, every output array bit for bit."""
import pytest

from .fuzz_parity import check, soup_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(4))
def test_soup_slice(product_lib, block, tmp_path):
    for seed in range(1000 + 8 * block, 1000 + 8 * block + 8):
        ok, st = check(product_lib, soup_case(seed, str(tmp_path)))
        assert ok, (seed, st)


@pytest.mark.parametrize("ntx,nrx,nb", [(4, 3, 2), (3, 2, 5), (5, 1, 1), (2, 4, 3)])
def test_many_tx_keep_the_reference_order_on_freq_shift(product_lib, ntx, nrx, nb, tmp_path):
    """Several TXs, moving meshes: the reference's `+= 0` on freq_shift[tx*np + path] (Q10) meets record slots of rx 0
    -- of TX 0, and with fewer bounces than TXs of later TXs too -- and a -0 turns into +0 or not by the ORDER.  The
    host scatters rx 0 run by run in that order and rx >= 1 over the whole hit list at once (compute_paths.c,
    run_batch); every array must stay bit-identical, freq_shift down to the sign of zero."""
    import numpy as np
    for seed in (2000 + 7 * ntx + nb, 2100 + 7 * ntx + nb, 2201 + 7 * ntx + nb):
        c = soup_case(seed, str(tmp_path), max_tri=120)
        rng = np.random.default_rng(seed)
        scale = max(abs(v) for p in c["tx_pos"] + c["rx_pos"] for v in p) or 1.0
        c["tx_pos"] = (rng.uniform(-0.6, 0.6, (ntx, 3)) * scale).tolist()
        c["rx_pos"] = (rng.uniform(-0.6, 0.6, (nrx, 3)) * scale).tolist()
        c["tx_vel"] = rng.uniform(-30, 30, (ntx, 3)).tolist()
        c["rx_vel"] = rng.uniform(-30, 30, (nrx, 3)).tolist()
        c["num_bounces"] = nb
        c["num_paths"] = int(rng.integers(3000, 9000))
        ok, st = check(product_lib, c)
        assert ok, (seed, st)
