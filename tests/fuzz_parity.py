"""Parity fuzzers (GPU box; not collected by pytest -- run by hand, or in slices by
tests/test_gpu_fuzz_slice.py):

    python tests/fuzz_parity.py configs LO HI    random endpoints / frequencies / ray and bounce
                                                 counts / velocities on the bundled scenes
    python tests/fuzz_parity.py soups LO HI      random triangle soups: 1-900 triangles at scales
                                                 from 5 cm to 300 m, coplanar clusters on shared
                                                 axis-aligned planes, slivers, several meshes, all
                                                 materials, moving meshes

    python tests/fuzz_parity.py big LO HI        150-400 k rays per TX on the bundled scenes
                                                 (narrow packets, many chunks, stable compaction)

    python tests/fuzz_parity.py bigsoups LO HI   the soups with 120-300 k rays per TX: narrow packets on
                                                 random geometry (direction tables, many chunks)

    python tests/fuzz_parity.py inplane LO HI    endpoints placed exactly in the planes of random
                                                 scene triangles (inside or far outside them), odd
                                                 ray counts: the reference's noise regime

Every case: the product through the drop-in C ABI against the oracle, every output array, bit for
bit.  Round 1: configs 100-12700, soups 0-5650, big 0-660, inplane 0-3000: 0 mismatches.
Round 2: 90 900 more cases over every intersection mode: 0 mismatches.
Round 3: 112 670 more (fused kernels; fine leaves forced onto every table; the wide-packet queue on, overflowing,
absent, every packet through it; sliced LoS; logical devices; the shipped defaults): 0 mismatches.
Round 4: 89 860 more (patch tables, records / image / chain kernels, two streams, deep soups -- `deepsoups`: 4-8 bounces
on <= 64 triangles -- and the host's block order with several TXs; six fuzzers at once on one GPU): 0 mismatches."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HRT_TUNE", "rxt_min_rays=0")   # direction / patch tables at every size (default: big launch sets only)
from hermespy_rt_amd import abi, lib          # noqa: E402
from oracle import oracle                      # noqa: E402
from tests import configs as K                 # noqa: E402
from tests import scenes_gen as G              # noqa: E402
from tests.parity import compare_dense         # noqa: E402


def soup_case(seed, tmp, max_tri=900):
    rng = np.random.default_rng(50000 + seed)
    T = int(rng.integers(1, max_tri))
    scale = [1.0, 30.0, 0.05, 300.0][seed % 4]
    c0 = rng.uniform(-1, 1, (T, 3)) * scale
    ext = scale * rng.choice([0.02, 0.2, 1.0])
    tri = c0[:, None, :] + rng.uniform(-1, 1, (T, 3, 3)) * ext
    if seed % 7 == 0:      # axis-aligned planes with shared coordinates (coplanar clusters)
        tri[:, :, seed % 3] = np.round(tri[:, :1, seed % 3] / (scale * 0.25)) * (scale * 0.25)
    if seed % 11 == 0:     # slivers
        tri[:, 2, :] = tri[:, 0, :] + (tri[:, 1, :] - tri[:, 0, :]) * 0.5 + rng.normal(0, 1e-4 * scale, (T, 3))
    nm = int(rng.integers(1, 5))
    cuts = np.sort(rng.integers(0, T + 1, nm - 1)).tolist()
    meshes = []
    for a, b in zip([0] + cuts, cuts + [T]):
        v = tri[a:b].reshape(-1, 3).astype(np.float32)
        idx = np.arange(3 * (b - a), dtype=np.uint32).reshape(-1, 3)
        meshes.append(dict(vs=v, idx=idx, material_index=int(rng.integers(0, 17)),
                           velocity=rng.uniform(-20, 20, 3) if rng.random() < 0.5 else np.zeros(3)))
    p = os.path.join(tmp, "soup_%d.hrt" % seed)
    G.write_hrt(p, meshes)
    nrx, ntx = int(rng.integers(1, 5)), int(rng.integers(1, 3))
    return G.cfg(p, (rng.uniform(-1.2, 1.2, (nrx, 3)) * scale).tolist(),
                 (rng.uniform(-1.2, 1.2, (ntx, 3)) * scale).tolist(), int(rng.integers(100, 4000)),
                 int(rng.integers(1, 7)), f=float(rng.choice([0.7, 3.5, 28.0, 77.0])))


def big_case(seed):
    from tests.test_gpu_random_configs import BOUNDS, SCENES
    rng = np.random.default_rng(777000 + seed)
    scene = SCENES[seed % 4]
    lo, hi = BOUNDS[scene]
    nrx, ntx = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    c = K.cfg(scene, rng.uniform(lo, hi, (nrx, 3)).tolist(), rng.uniform(lo, hi, (ntx, 3)).tolist(),
              float(rng.choice([2.4, 3.5, 28.0])), int(rng.integers(150000, 400000)), int(rng.integers(2, 5)))
    if seed % 2:
        c["rx_vel"] = rng.uniform(-30, 30, (nrx, 3)).tolist()
        c["tx_vel"] = rng.uniform(-30, 30, (ntx, 3)).tolist()
    return c


def inplane_case(seed):
    from tests.test_gpu_random_configs import SCENES
    rng = np.random.default_rng(31000 + seed)
    scene = SCENES[seed % 4]
    tris = np.asarray(oracle.flatten(oracle.read_hrt(os.path.join(K.SC, scene)))["tri_vtx"], np.float64).reshape(-1, 3, 3)

    def point():
        t = tris[int(rng.integers(0, len(tris)))]
        u, v = rng.uniform(-1.5, 2.5, 2) if rng.random() < 0.5 else rng.uniform(0, 0.5, 2)
        return (t[0] + u * (t[1] - t[0]) + v * (t[2] - t[0])).astype(np.float32).tolist()

    nrx, ntx = int(rng.integers(1, 5)), int(rng.integers(1, 3))
    return K.cfg(scene, [point() for _ in range(nrx)], [point() for _ in range(ntx)],
                 float(rng.choice([2.4, 3.5, 28.0])), 2 * int(rng.integers(200, 3000)) + 1, int(rng.integers(1, 6)))


def bigsoup_case(seed, tmp):
    """a soup traced with enough rays for NARROW packets: the per-RX / per-TX direction tables, the
    half-word shadow results and the stable compaction over many chunks, on random geometry"""
    c = soup_case(seed, tmp)
    rng = np.random.default_rng(91000 + seed)
    c["num_paths"] = int(rng.integers(120000, 300000))
    c["num_bounces"] = int(rng.integers(1, 4))
    if seed % 3 == 0:      # endpoints well inside the soup: most shadow packets qualify for the tables
        scale = [1.0, 30.0, 0.05, 300.0][seed % 4]
        c["rx_pos"] = (rng.uniform(-0.5, 0.5, (len(c["rx_pos"]), 3)) * scale).tolist()
        c["tx_pos"] = (rng.uniform(-0.5, 0.5, (len(c["tx_pos"]), 3)) * scale).tolist()
    return c


def deepsoup_case(seed, tmp):
    """a soup traced deep: 4-8 bounces with tens of thousands of rays -- the tail of the launches as one
    persistent kernel (hrt_chain_kernel) on random geometry, lists of many chunks at its first bounces"""
    c = soup_case(seed, tmp, max_tri=65)   # (whole bounces are fused up to 64 triangles)
    rng = np.random.default_rng(57000 + seed)
    c["num_paths"] = int(rng.integers(40000, 160000))
    c["num_bounces"] = int(rng.integers(4, 9))
    if seed % 2 == 0:      # endpoints inside the soup: rays keep hitting
        scale = [1.0, 30.0, 0.05, 300.0][seed % 4]
        c["rx_pos"] = (rng.uniform(-0.4, 0.4, (len(c["rx_pos"]), 3)) * scale).tolist()
        c["tx_pos"] = (rng.uniform(-0.4, 0.4, (len(c["tx_pos"]), 3)) * scale).tolist()
    return c


def check(L, c):
    got = abi.run_compute_paths(L, *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    return all(v == 0 for v in st.values()), st


def main():
    mode, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    L = lib.load()
    tmp = tempfile.mkdtemp()
    bad, t0 = 0, time.time()
    if mode == "configs":
        from tests.test_gpu_random_configs import _case
    for seed in range(lo, hi):
        if mode == "bigsoups":
            c = bigsoup_case(seed, tmp)
        elif mode == "deepsoups":
            c = deepsoup_case(seed, tmp)
        else:
            c = _case(seed) if mode == "configs" else (big_case(seed) if mode == "big" else
                                                      (inplane_case(seed) if mode == "inplane" else soup_case(seed, tmp)))
        ok, st = check(L, c)
        if not ok:
            bad += 1
            print("MISMATCH", mode, "seed", seed, st, flush=True)
        if (seed - lo) % 50 == 49:
            print(mode, "seed", seed, "mismatches", bad, "%.0f s" % (time.time() - t0), flush=True)
    print("FUZZ", mode, hi - lo, "cases, mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
