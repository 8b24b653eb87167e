#!/usr/bin/env python3
"""Time per step of the hot path on generated rooms full of tilted boxes (tests/scenes_gen.py) as a
function of the triangle count T, for one intersection variant (HRT_TUNE variant=...):

    HRT_TUNE=variant=2 python profiles/accel_scaling.py 24 91 166 500 1000 4000 8333

1 M rays, 3 RX, 2 bounces (the workload of DESIGN.md section 9).  HRT_SCALING_SCENE=city: the
argument is the number of buildings per side of tests/scenes_gen.city (10 n^2 + 2 triangles).  Prints one JSON line per scene."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hermespy_rt_amd.device import Tracer  # noqa: E402
from tests import scenes_gen as G  # noqa: E402

RX = [[5, 3, 1.5], [-8, -4, 2.0], [12, 9, 8.0]]
TX = [[-10, 5, 6.0]]


def main():
    tmp = tempfile.mkdtemp()
    rays = int(os.environ.get("HRT_SCALING_RAYS", "1000000"))
    kind = os.environ.get("HRT_SCALING_SCENE", "room")
    for nb in [int(x) for x in sys.argv[1:]]:
        p = os.path.join(tmp, "%s_%d.hrt" % (kind, nb))
        rx, tx = RX, TX
        if kind == "city":   # nb = buildings per side; TX over a crossing, RX at street level
            T, half = G.city(p, nb)
            tx = [[0.0, 0.0, 25.0]]
            rx = [[60.0, 0.0, 1.5], [0.0, -90.0, 1.5], [-150.0, 30.0, 1.5]]
        else:
            T = G.room_with_clutter(p, nb, seed=7, tilt=True, scale=max(1.0, (nb / 500.0) ** (1.0 / 3.0)))
        t0 = time.time()
        tr = Tracer(p, rx, tx, [[0, 0, 0]] * 3, [[0, 0, 0]], 3.5, rays, 2)
        t_build = time.time() - t0
        for _ in range(2):
            tr.trace()
        torch.cuda.synchronize()
        n = 5 if T < 20000 else 3
        t0 = time.perf_counter()
        for _ in range(n):
            tr.trace()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        los_ms, trace_ms = tr.trace(timed=True)
        c = tr.counts()
        w = tr.work(c)
        print(json.dumps(dict(variant=os.environ.get("HRT_TUNE", "default"), boxes=nb, T=T,
                              ms_per_step=ms, trace_ms=[round(x, 3) for x in trace_ms],
                              shade_ms=[round(x, 3) for x in tr.last_shade_ms], live=w["live"], tests_per_s=w["tests"] / ms * 1e3,
                              setup_s=t_build)), flush=True)
        tr.close()


if __name__ == "__main__":
    main()
