"""Model of the ORDER in which the host writes freq_shift (csrc/host/compute_paths.c, run_batch).  The reference
(src/compute_paths.c:596-723) loops bounce -> tx -> path and, for a ray that hit, first adds a zero to
freq_shift[tx*np + path] (Q10: `+=` of dot(d - d, v) -- a +0, which turns a -0 into +0) and then, rx by rx,
subtracts the record's Doppler term from the record's dense slot ((rx*ntx+tx)*nb+b)*np+path.  The add's index is a
dense slot too -- of rx 0 -- so the two can meet, and the sign of a zero then depends on the order.
The product scatters a bounce as: for every TX run in order {the run's adds; the run's records of rx 0}, then for
rx >= 1 the records of ALL runs at once.  Claim: bit-identical to the reference's order, for any hit pattern, also
with fewer bounces than TXs (the adds then reach slots of later TXs).  Checked here on random data rich in
signed zeros -- and, as it turns out, true of ANY order (second test)."""
import numpy as np
import pytest


def slot(rx, tx, b, p, ntx, nb, npth):
    return ((rx * ntx + tx) * nb + b) * npth + p


def make(seed, nrx, ntx, nb, npth):
    rnd = np.random.default_rng(seed)
    vals = np.array([0.0, -0.0, 1.5, -2.25, 0.0, -0.0], np.float32)
    fs0 = rnd.choice(vals, nrx * ntx * nb * npth).astype(np.float32)
    hit = rnd.random((nb, ntx, npth)) < 0.6
    for b in range(1, nb):
        hit[b] &= hit[b - 1]      # a ray that missed is dead
    dfs = rnd.choice(vals, (nb, ntx, npth, nrx)).astype(np.float32)
    unblocked = rnd.random((nb, ntx, npth, nrx)) < 0.8
    return fs0, hit, dfs, unblocked


def reference_order(fs0, hit, dfs, unb, nrx, ntx, nb, npth):
    fs = fs0.copy()
    zero = np.float32(0.0)
    for b in range(nb):
        for tx in range(ntx):
            for p in range(npth):
                if not hit[b, tx, p]:
                    continue
                fs[tx * npth + p] = fs[tx * npth + p] + zero
                for rx in range(nrx):
                    if unb[b, tx, p, rx]:
                        s = slot(rx, tx, b, p, ntx, nb, npth)
                        fs[s] = fs[s] - dfs[b, tx, p, rx]
    return fs


def product_order(fs0, hit, dfs, unb, nrx, ntx, nb, npth):
    fs = fs0.copy()
    zero = np.float32(0.0)
    for b in range(nb):
        for tx in range(ntx):                      # the TX runs of the hit list, in order
            ps = np.nonzero(hit[b, tx])[0]
            for p in ps:                           # the run's adds (distinct slots: any order)
                fs[tx * npth + p] = fs[tx * npth + p] + zero
            for p in ps:                           # the run's records of rx 0
                if unb[b, tx, p, 0]:
                    s = slot(0, tx, b, p, ntx, nb, npth)
                    fs[s] = fs[s] - dfs[b, tx, p, 0]
        for rx in range(1, nrx):                   # rx >= 1: the whole hit list at once
            for tx in range(ntx):
                for p in np.nonzero(hit[b, tx])[0]:
                    if unb[b, tx, p, rx]:
                        s = slot(rx, tx, b, p, ntx, nb, npth)
                        fs[s] = fs[s] - dfs[b, tx, p, rx]
    return fs


@pytest.mark.parametrize("nrx,ntx,nb,npth", [(1, 1, 1, 40), (3, 1, 4, 30), (2, 2, 6, 25), (3, 4, 2, 20), (2, 5, 1, 16),
                                              (4, 3, 3, 12), (1, 6, 2, 10)])
def test_block_order_equals_reference_order(nrx, ntx, nb, npth):
    for seed in range(6):
        fs0, hit, dfs, unb = make(seed * 31 + nrx + 7 * ntx, nrx, ntx, nb, npth)
        a = reference_order(fs0, hit, dfs, unb, nrx, ntx, nb, npth)
        b = product_order(fs0, hit, dfs, unb, nrx, ntx, nb, npth)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_a_zero_add_and_a_subtraction_commute():
    """Why no wrong order can be shown on this array: (x + z) - d and (x - d) + z have the same bits for z = +-0 and
    every x, d (a -0 survives neither order once a +0 was added; adding -0 is the identity), NaNs stay NaNs.  The
    product keeps the reference's order on rx 0 all the same -- it costs ntx blocks per bounce, not ntx * nrx."""
    sp = np.array([0.0, -0.0, 1e-45, -1e-45, 1.5, -1.5, 3.0e38, -3.0e38, np.inf, -np.inf, np.nan], np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        for z in (np.float32(0.0), np.float32(-0.0)):
            x, d = np.meshgrid(sp, sp, indexing="ij")
            a = ((x + z) - d).astype(np.float32)
            b = ((x - d) + z).astype(np.float32)
            same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
            assert same.all(), (z, x[~same], d[~same])
