#!/usr/bin/env python3
"""Record outputs of the REAL reference as fixtures.

Runs oracle/_ref/libhrt_ref.so -- the reference's own src/compute_paths.c, scene.c,
materials.c compiled in place by `make -C oracle ref` (only possible where /root/reference is
mounted) -- through the harness of hermespy_rt_amd.abi (sentinel-prefilled caller buffers)
and writes

  tests/golden/<case>.npz     every output array of a small case, as uint32/uint8 bit
                              patterns (sentinels included, so "not written" is pinned too)
  tests/golden/full_size.json per-bounce counts, written-slot counts and order-independent
                              64-bit checksums of every output array at BASELINE.json's full
                              sizes (see slot_checksum below)

The reference holds no golden values of its own (test/test.py asserts shapes only), so these
ARE the golden vectors of this path.  Nothing of the reference's code is stored: only inputs
(named configs of tests/configs.py) and output data.

    python tests/golden/make_golden.py [--full [--only NAME]]   # --full also regenerates full_size.json
"""
import ctypes
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from hermespy_rt_amd import abi  # noqa: E402
from tests import configs as K  # noqa: E402

SMALL = {
    "c1_box_10k": K.C1,
    "c2_reflector_4096": K.small(K.C2, 4096),
    "c3_canyon_2048": K.small(K.C3, 2048),
    "c3_doppler_1000": K.small(K.C3_DOPPLER, 1000),
    "c4_2cars_2tx_2048": K.small(K.C4, 2048),
    "c4_doppler_2tx_1001": dict(K.small(K.C4_DOPPLER, 1001), num_bounces=3),
    "c5_8x8_256": dict(K.small(K.C5, 256), num_bounces=3),
    "test_py_10k": K.TEST_PY,
    "coincident_3000": K.small(K.COINCIDENT, 3000),
}

FULL = {"C1": K.C1, "C2": K.C2, "C3": K.C3, "C4_1M": K.small(K.C4, 1000000),
        "C5_1M": K.small(K.C5, 1000000),   # 8 TX x 8 RX x 8 bounces: ~25 GB of dense arrays, ~3 min
        "C4": K.C4}                        # BASELINE configs[3] at full size: 2 TX x 8 M rays x 6 bounces, ~12 GB

M64 = (1 << 64) - 1


def mix64(x):
    """splitmix64 finaliser on uint64 arrays."""
    x = x.astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def slot_checksum(slots, values_u32):
    """Order-independent checksum of a sparse array: sum over written slots of
    mix64(slot * 2^32 | value_bits), mod 2^64.  Computable from the dense array (reference)
    and from compact path records in any order (GPU)."""
    with np.errstate(over="ignore"):
        key = (slots.astype(np.uint64) << np.uint64(32)) | values_u32.astype(np.uint64)
        return int(np.sum(mix64(key), dtype=np.uint64))


def dense_checksums(res):
    """checksums + written counts of every scatter array of a dense result."""
    out = {}
    s = res["scat"]
    for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau", "freq_shift"):
        a = s[k].ravel()
        w = np.flatnonzero(abi.written(a))
        out[k] = dict(written=int(w.size), checksum=slot_checksum(w, a.view(np.uint32)[w]))
    d = s["directions_rx"].reshape(-1, 3)
    w = np.flatnonzero(abi.written(d[:, 0]))
    cs = 0
    for c in range(3):
        cs = (cs + slot_checksum(w * 3 + c, np.ascontiguousarray(d[:, c]).view(np.uint32)[w])) & M64
    out["directions_rx"] = dict(written=int(w.size), checksum=cs)
    out["directions_tx"] = dict(written=int(abi.written(s["directions_tx"]).sum()))
    return out


def load_ref():
    p = os.path.join(REPO, "oracle", "_ref", "libhrt_ref.so")
    if not os.path.exists(p):
        sys.exit("oracle/_ref/libhrt_ref.so missing: run `make -C oracle ref` where /root/reference is mounted")
    from tests import refabi
    return refabi.load()


def flat(res):
    out = {}
    for blk in ("los", "scat"):
        for k, v in res[blk].items():
            if blk == "scat" and k == "directions_tx":
                continue   # never written by the reference (quirk Q1); asserted, not stored
            out["%s.%s" % (blk, k)] = v.view(np.uint32)
    for k in ("los_rays", "scat_rays"):
        out[k] = res[k].view(np.uint32)
    for k in ("los_active", "scat_active"):
        out[k] = res[k]
    return out


def main():
    lib = load_ref()
    for name, c in ({} if "--only" in sys.argv else SMALL).items():
        r = abi.run_compute_paths(lib, *K.args(c))
        assert not abi.written(r["scat"]["directions_tx"]).any()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **flat(r))
        print("%-24s %8d bytes" % (name, os.path.getsize(path)))
    if "--full" in sys.argv:
        # --only NAME regenerates one entry and keeps the others
        only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
        path = os.path.join(HERE, "full_size.json")
        summ = json.load(open(path)) if only and os.path.exists(path) else {}
        for name, c in FULL.items():
            if only and name != only:
                continue
            r = abi.run_compute_paths(lib, *K.args(c), with_rays=True)
            nb, npth = c["num_bounces"], c["num_paths"]
            ntx = len(c["tx_pos"])
            hits = []
            for b in range(nb):
                w = abi.written(r["scat"]["a_te_re"][0, :, b, :])
                hits.append(int(w.sum()))
            nz = int((abi.written(r["scat"]["directions_rx"][..., 0])).sum())
            summ[name] = dict(config={k: (v if not isinstance(v, str) else os.path.basename(v)) for k, v in c.items()},
                              hits_per_bounce=hits, records_unblocked=nz, arrays=dense_checksums(r),
                              los={k: [float(x) for x in v.ravel()] for k, v in r["los"].items() if not k.startswith("dir")})
            print(name, hits, nz)
            del r
        json.dump(summ, open(os.path.join(HERE, "full_size.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
