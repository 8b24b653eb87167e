#!/usr/bin/env python3
"""Per wave-trace clock records of the trace kernel (library built with `make EXTRA=-DHRT_UNIT_CLOCKS`):
how long one trace of each kind takes, usable packets against too-wide ones, and how many traces are in
flight over time (the tail of a launch).   python profiles/unit_clocks.py BOXES [RAYS]"""
import ctypes
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hermespy_rt_amd import lib  # noqa: E402
from hermespy_rt_amd.device import Tracer  # noqa: E402
from tests import scenes_gen as G  # noqa: E402

d = tempfile.mkdtemp()
p = os.path.join(d, "scene.hrt")
if sys.argv[1] in ("c1", "c2", "c3", "c4", "c5"):   # a bench workload instead of a generated scene
    from hermespy_rt_amd.workloads import WORKLOADS
    w = WORKLOADS[sys.argv[1]]
    T = -1
    tr = Tracer(w["scene_path"], w["rx_pos"], w["tx_pos"], w["rx_vel"], w["tx_vel"], w["f_ghz"],
                w["num_paths"], w["num_bounces"])
    nb = 0
elif os.environ.get("HRT_SCALING_SCENE", "room") == "city":
    nb = int(sys.argv[1])
    rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    T, _ = G.city(p, nb)
    tr = Tracer(p, [[60.0, 0.0, 1.5], [0.0, -90.0, 1.5], [-150.0, 30.0, 1.5]], [[0.0, 0.0, 25.0]], [[0, 0, 0]] * 3,
                [[0, 0, 0]], 3.5, rays, 2)
else:
    nb = int(sys.argv[1])
    rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    T = G.room_with_clutter(p, nb, seed=7, tilt=True, scale=max(1.0, (nb / 500.0) ** (1.0 / 3.0)))
    tr = Tracer(p, [[5, 3, 1.5], [-8, -4, 2.0], [12, 9, 8.0]], [[-10, 5, 6.0]], [[0, 0, 0]] * 3, [[0, 0, 0]], 3.5, rays, 2)
L = lib.load()
arr = (ctypes.c_uint64 * 48)()
uf = os.path.join(d, "units.bin")
os.environ["HRT_UNIT_FILE"] = uf
tr.trace()
torch.cuda.synchronize()
L.hrt_debug_kernel_stats(0, arr, 1)
tr.trace()
torch.cuda.synchronize()
L.hrt_debug_kernel_stats(0, arr, 0)
u = np.fromfile(uf, dtype=np.uint64).reshape(-1, 2)
u2 = np.fromfile(uf + ".2", dtype=np.uint64).reshape(-1, 2) if os.path.exists(uf + ".2") else None
if u2 is not None:   # phases of a unit around its trace (10 ns ticks), and the workgroup's start-up
    m2 = u[:, 0] != 0
    pre = (u2[m2, 0] & np.uint64(0xffffffff)).astype(np.int64) / 100.0
    post = (u2[m2, 0] >> np.uint64(32)).astype(np.int64) / 100.0
    st = u2[m2, 1].astype(np.int64) / 100.0
    trc = (u[m2, 1] & np.uint64((1 << 24) - 1)).astype(np.int64) / 100.0
    print(f"unit phases (us, mean / median): unit start -> trace start {pre.mean():.2f} / {np.median(pre):.2f}, trace {trc.mean():.2f} / "
          f"{np.median(trc):.2f}, trace end -> unit end {post.mean():.2f} / {np.median(post):.2f}; kernel entry -> first unit "
          f"{st[st > 0].mean():.2f} / {np.median(st[st > 0]):.2f} (n = {(st > 0).sum()})")
u = u[u[:, 0] != 0]   # (slots are hashed, not counted: unused ones are zero)
t0 = u[:, 0].astype(np.int64)
dt = (u[:, 1] & np.uint64((1 << 24) - 1)).astype(np.int64)
cosa = ((u[:, 1] >> np.uint64(24)) & np.uint64(0xff)).astype(np.int64) / 255.0
near = ((u[:, 1] >> np.uint64(32)) & np.uint64(0xfff)).astype(np.int64)
plv = ((u[:, 1] >> np.uint64(44)) & np.uint64(0xfff)).astype(np.int64)
kind = ((u[:, 1] >> np.uint64(56)) & np.uint64(3)).astype(np.int64)
usable = ((u[:, 1] >> np.uint64(60)) & np.uint64(1)).astype(np.int64)
t0 -= t0.min()
print(f"T={T} records={len(u)}  (wall clock: 100 MHz, 10 ns per tick)")
for k, name in enumerate(("primary0", "primary", "shadow")):
    for us in (1, 0):
        m = (kind == k) & (usable == us)
        if not m.any():
            continue
        x = dt[m] / 100.0   # us
        print(f"{name:9s} usable={us}: n={m.sum():7d} sum={x.sum() / 1e3:9.1f} ms  mean={x.mean():8.1f} us  "
              f"p50={np.percentile(x, 50):8.1f} p90={np.percentile(x, 90):8.1f} p99={np.percentile(x, 99):8.1f} max={x.max():9.1f}")
# share of the trace time by duration class (all traces)
edges_us = [0, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 1e9]
tot = dt.sum()
print("duration classes (us: traces, share of all trace time): " + "  ".join(
    f"{a:g}-{b:g}: {int(((dt >= a * 100) & (dt < b * 100)).sum())} {dt[(dt >= a * 100) & (dt < b * 100)].sum() / tot:.3f}"
    for a, b in zip(edges_us[:-1], edges_us[1:])))
bin_us = float(os.environ.get("HRT_UC_BIN_US", "250"))
# what makes a usable trace slow: cone, near spheres, plane leaves by duration class
m = usable == 1
for lo, hi in ((0, 50), (50, 100), (100, 200), (200, 400), (400, 800), (800, 1600), (1600, 3200), (3200, 1e9)):
    c = m & (dt >= lo * 100) & (dt < hi * 100)
    if c.any():
        print(f"usable {lo:5.0f}-{hi:<6.0f} us: n={c.sum():7d} sum={dt[c].sum() / 1e5:9.1f} ms  cos(alpha) mean={cosa[c].mean():.3f} "
              f"min={cosa[c].min():.3f}  near spheres mean={near[c].mean():7.1f} max={near[c].max():5d}  plane leaves mean={plv[c].mean():7.1f} "
              f"max={plv[c].max():5d}  shadow share={np.mean(kind[c] == 2):.2f}")
# traces in flight over time, 0.25 ms bins
end = t0 + dt
bw = int(bin_us * 100)
edges = np.arange(0, end.max() + bw, bw)
line = []
for a in edges[:-1]:
    b = a + bw
    ov = np.clip(np.minimum(end, b) - np.maximum(t0, a), 0, None).sum() / float(bw)
    line.append(int(round(ov)))
print(f"wave-traces in flight per {bin_us:g} us bin:", line)
