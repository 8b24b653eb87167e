#!/usr/bin/env python3
"""ms per step of rank 0's shard of a workload at a given world size, one GPU (HRT_TUNE is honoured):
    python profiles/tools/step_ms.py <workload> [world=1] [steps=200] [num_bounces]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import hermespy_rt_amd  # noqa: E402,F401
from hermespy_rt_amd.device import Tracer  # noqa: E402
from hermespy_rt_amd.workloads import WORKLOADS  # noqa: E402

w = sys.argv[1]
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
c = dict(WORKLOADS[w])
if len(sys.argv) > 4:
    c["num_bounces"] = int(sys.argv[4])
tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"], c["num_paths"], c["num_bounces"],
            rank=0, world=world)
best = 1e9
for rep in range(3):
    for _ in range(5):
        tr.trace()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.trace()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps * 1e3)
tr.counts()   # (raises on a void step)
print("%s world %d nb %d [%s]: %.4f ms/step" % (w, world, c["num_bounces"], os.environ.get("HRT_TUNE", ""), best))
tr.close()
