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
