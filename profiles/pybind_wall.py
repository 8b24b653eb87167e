#!/usr/bin/env python3
"""Wall time of the drop-in PYTHON call hermespy_rt.compute_paths() on a named workload (GPU box):
    python profiles/pybind_wall.py [c3] [repeats] [list]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hermespy_rt_amd                         # noqa: E402
import torch                                   # noqa: E402,F401  (HIP runtime first)
from hermespy_rt_amd import workloads as W     # noqa: E402

sys.path.insert(0, hermespy_rt_amd.LIB_DIR)
import hermespy_rt                             # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = W.WORKLOADS[name]
f32 = lambda a: np.array(a, dtype=np.float32)
for k in range(reps):
    t0 = time.perf_counter()
    los, scat = hermespy_rt.compute_paths(c["scene_path"], f32(c["rx_pos"]), f32(c["tx_pos"]), f32(c["rx_vel"]),
                                          f32(c["tx_vel"]), c["f_ghz"], len(c["rx_pos"]), len(c["tx_pos"]),
                                          c["num_paths"], c["num_bounces"])
    dt = time.perf_counter() - t0
    nz = int(np.count_nonzero(scat.a_te))
    print("call %d: %.3f s  (%s; non-zero a_te %d)" % (k, dt, W.describe(c), nz), flush=True)
    del los, scat
for k in range(reps if "list" in sys.argv else 0):
    t0 = time.perf_counter()
    P = hermespy_rt.compute_paths_list(c["scene_path"], f32(c["rx_pos"]), f32(c["tx_pos"]), f32(c["rx_vel"]),
                                       f32(c["tx_vel"]), c["f_ghz"], len(c["rx_pos"]), len(c["tx_pos"]),
                                       c["num_paths"], c["num_bounces"])
    dt = time.perf_counter() - t0
    print("list call %d: %.3f s  (%d records)" % (k, dt, P["rx"].size), flush=True)
    del P
