"""Named workloads: BASELINE.json configs with the synthetic endpoints fixed in SURVEY.md 8(d)
(defined in hermespy_rt_amd.workloads, which bench.py uses too) plus test-only cases.
`small(cfg, n)` keeps everything but the ray count (parity-test sizes)."""
import os

from hermespy_rt_amd.workloads import (C1, C2, C3, C3_DOPPLER, C3_RX, C4, C4_DOPPLER, C5, SC, Z,  # noqa: F401
                                       cfg)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the reference's own smoke test (test/test.py:8-17)
TEST_PY = cfg("simple_reflector.hrt", [[0, 0, .15]], [[0, 0, .151]], 3.0, 10000, 3)
# tx == rx (coincident LoS branch), test/test.c:22-23
COINCIDENT = cfg("simple_street_canyon_with_cars.hrt", [[0, 0, .5]], [[0, 0, .5]], 3.0, 30000, 3)

# Endpoints placed EXACTLY in planes of scene triangles (floor/wall planes, the car roofs at
# z = 1.5 / 0.75, the reflector's own plane): rays then travel inside those planes, where the
# reference's u/v/dist are ratios of rounding noise and it can report hits on coplanar triangles
# far from the ray.  What must be reproduced is that noise -- the regime the packet culling's
# numerator-space proof exists for.  Odd ray counts: the middle ray of the
# Fibonacci sphere is then horizontal (d.z = 6e-17), i.e. launched INSIDE the plane z = tx.z.
IN_PLANE = dict(
    box=cfg("box.hrt", [[2, 1, 0], [5, 0, 2.5], [1, 5, 5]], [[0, 0, 0]], 3.0, 4001, 3),
    canyon=cfg("simple_street_canyon_with_cars.hrt",
               [[-10, 1.5, 0], [10, -1.5, 1.5], [35, 0, .75], [20, 4.7, 1.0]],
               [[-40, 0, 0], [-30, -4, 1.5]], 3.5, 6001, 4),
    cars=cfg("2cars.hrt", [[-3.958, 0, .6142], [4.629, 5.0, 2.6142]], [[0, -20, 0]], 70.0, 5001, 3),
    reflector=cfg("simple_reflector.hrt", [[-.3, .2, 0], [2, 0, 0]], [[.2, .1, 0]], 3.0, 3001, 2),
)

ALL = dict(C1=C1, C2=C2, C3=C3, C3_DOPPLER=C3_DOPPLER, C4=C4, C4_DOPPLER=C4_DOPPLER, C5=C5,
           TEST_PY=TEST_PY, COINCIDENT=COINCIDENT)


def small(c, n):
    d = dict(c)
    d["num_paths"] = n
    return d


def args(c):
    """positional args of abi.run_compute_paths / oracle.compute_paths"""
    return (c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
            c["num_paths"], c["num_bounces"])
