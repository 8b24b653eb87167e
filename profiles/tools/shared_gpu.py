#!/usr/bin/env python3
"""N tracer processes on ONE GPU at the same time (a drop-in library is used like that: several worker
processes, or next to torch): ms per step of each, against the solo time.
    python profiles/tools/shared_gpu.py <workload> <rays> <procs> [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(w, rays, steps, bar, q, k):
    import torch
    import hermespy_rt_amd  # noqa: F401
    from hermespy_rt_amd.device import Tracer
    from hermespy_rt_amd.workloads import WORKLOADS
    c = dict(WORKLOADS[w])
    c["num_paths"] = rays
    tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"], c["num_paths"], c["num_bounces"])
    for _ in range(3):
        tr.trace()
    torch.cuda.synchronize()
    bar.wait()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.trace()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    q.put((k, dt))
    bar.wait()
    tr.close()


def run(w, rays, procs, steps):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    bar = ctx.Barrier(procs)
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(w, rays, steps, bar, q, k)) for k in range(procs)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=600) for _ in range(procs))
    for p in ps:
        p.join(60)
    return [d for _, d in out]


if __name__ == "__main__":
    w, rays, procs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    solo = run(w, rays, 1, steps)[0]
    both = run(w, rays, procs, steps)
    print("%s %d rays: solo %.3f ms/step; %d processes at once: %s ms/step each (x%.1f of solo per process, x%.2f of the work-conserving %d x solo)"
          % (w, rays, solo, procs, ", ".join("%.3f" % d for d in both), max(both) / solo, max(both) / (procs * solo), procs))
