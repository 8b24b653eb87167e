#!/usr/bin/env python3
"""Diagnostic counters of the trace kernel on one generated room (library built with `make STATS=1`):
    python profiles/accel_stats.py BOXES [RAYS]"""
import ctypes
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hermespy_rt_amd import lib  # noqa: E402
from hermespy_rt_amd.device import Tracer  # noqa: E402
from tests import scenes_gen as G  # noqa: E402

p = os.path.join(tempfile.mkdtemp(), "room.hrt")
if sys.argv[1] in ("c1", "c2", "c3", "c4", "c5"):   # a bench workload instead of a generated scene
    from hermespy_rt_amd.workloads import WORKLOADS
    w = WORKLOADS[sys.argv[1]]
    T = -1
    tr = Tracer(w["scene_path"], w["rx_pos"], w["tx_pos"], w["rx_vel"], w["tx_vel"], w["f_ghz"], w["num_paths"], w["num_bounces"])
elif os.environ.get("HRT_SCALING_SCENE", "room") == "city":
    nb = int(sys.argv[1])
    rays = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    T, _ = G.city(p, nb)
    tr = Tracer(p, [[60.0, 0.0, 1.5], [0.0, -90.0, 1.5], [-150.0, 30.0, 1.5]], [[0.0, 0.0, 25.0]], [[0, 0, 0]] * 3,
                [[0, 0, 0]], 3.5, rays, 2)
else:
    nb = int(sys.argv[1])
    rays = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    T = G.room_with_clutter(p, nb, seed=7, tilt=True, scale=max(1.0, (nb / 500.0) ** (1.0 / 3.0)))
    tr = Tracer(p, [[5, 3, 1.5], [-8, -4, 2.0], [12, 9, 8.0]], [[-10, 5, 6.0]], [[0, 0, 0]] * 3, [[0, 0, 0]], 3.5, rays, 2)
arr = (ctypes.c_uint64 * 48)()
lib.load().hrt_debug_kernel_stats(0, arr, 1)
tr.trace()
torch.cuda.synchronize()
lib.load().hrt_debug_kernel_stats(0, arr, 0)
cols = ["wave_traces", "usable_full", "candidates", "stage2", "stage3", "exact", "subpackets", "cull_rounds",
        "sphere_rounds", "plane_rounds", "plane_leaves", "plane_leaves_flagged", "flagged", "max_clk", "sum_clk"]
cols = cols[:13] + ["max_clk", "sum_clk", "clk15"]
for k, name in enumerate(("primary0", "primary", "shadow")):
    row = {c: int(arr[k * 16 + j]) for j, c in enumerate(cols)}
    wt = max(1, row["wave_traces"])
    row["per_trace"] = {c: round(row[c] / wt, 2) for c in cols[1:]}
    print(json.dumps(dict(T=T, kind=name, variant=os.environ.get("HRT_TUNE", "default"), **row)))
