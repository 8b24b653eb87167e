"""Named workloads: the BASELINE.json configurations with the synthetic endpoints fixed in
SURVEY.md 8(d) -- what bench.py times (C3 by default) and the parity tests scale down."""
import os

from . import SCENES_DIR as SC
Z = [0.0, 0.0, 0.0]


def cfg(scene, rx, tx, f, np_, nb, rx_vel=None, tx_vel=None):
    return dict(scene_path=os.path.join(SC, scene), rx_pos=rx, tx_pos=tx,
                rx_vel=rx_vel or [Z] * len(rx), tx_vel=tx_vel or [Z] * len(tx),
                f_ghz=f, num_paths=np_, num_bounces=nb)


C1 = cfg("box.hrt", [[2, 1, 1.5]], [[0, 0, 2.5]], 3.0, 10000, 1)
C2 = cfg("simple_reflector.hrt", [[0, 0, .15]], [[0, 0, .151]], 3.0, 1000000, 2)
C3_RX = [[-10, 1.5, 1.5], [10, -1.5, 1.5], [35, 0, 1.5], [50, 2, 3]]
C3 = cfg("simple_street_canyon_with_cars.hrt", C3_RX, [[-40, 0, 10]], 3.5, 4000000, 4)
C3_DOPPLER = cfg("simple_street_canyon_with_cars.hrt", C3_RX, [[-40, 0, 10]], 3.5, 4000000, 4,
                 rx_vel=[[1, 2, 3]] * 4, tx_vel=[[10, 0, 0]])
C4 = cfg("2cars.hrt", [[-2, 0, 1.5], [2, 0, 1.5]], [[0, -20, 3], [0, 20, 3]], 70.0, 8000000, 6)
C4_DOPPLER = cfg("2cars.hrt", [[-2, 0, 1.5], [2, 0, 1.5]], [[0, -20, 3], [0, 20, 3]], 70.0,
                 8000000, 6, rx_vel=[[1, 0, 0], [0, 1, 0]], tx_vel=[[3, 1, 0], [0, -2, 1]])
C5 = cfg("simple_street_canyon_with_cars.hrt",
         [[x, y, 1.5] for x in (-50, -25, 0, 25) for y in (-1.5, 1.5)],
         [[x, y, 10] for x in (-60, -20, 20, 60) for y in (-2, 2)], 3.5, 8000000, 8)

WORKLOADS = dict(c1=C1, c2=C2, c3=C3, c3_doppler=C3_DOPPLER, c4=C4, c4_doppler=C4_DOPPLER, c5=C5)


def describe(c):
    return "%s, %d TX + %d RX, %d rays/TX, %d bounces, %.1f GHz" % (
        os.path.basename(c["scene_path"]), len(c["tx_pos"]), len(c["rx_pos"]), c["num_paths"],
        c["num_bounces"], c["f_ghz"])


def args(c):
    """positional arguments (scene_path, rx_pos, tx_pos, rx_vel, tx_vel, f_ghz, num_paths, num_bounces)"""
    return (c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
            c["num_paths"], c["num_bounces"])
