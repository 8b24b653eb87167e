#!/usr/bin/env python3
"""Design study input (CPU only, oracle = checker used as a data source, nothing shipped):
the live lists of a workload in the order the device holds them -- coherent launch order,
stable compaction -- so that candidate-set designs can be priced on real waves before any
kernel is written.   python profiles/study/live_lists.py c3 [num_paths] -> /tmp/hrt_study/<w>_<n>.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hermespy_rt_amd  # noqa: E402,F401
from hermespy_rt_amd.workloads import WORKLOADS, args  # noqa: E402
from oracle import oracle  # noqa: E402


def launch_order(dirs):
    """hrt_launch_order_host restated (z-bands, serpentine diamond-angle azimuth, stable)."""
    n = len(dirs)
    nb = int(np.sqrt(n / 128.0))
    nb = max(1, min(nb, 4095))
    z = np.clip(dirs[:, 2].astype(np.float64), -1, 1)
    band = np.minimum((0.5 * (1.0 - z) * nb).astype(np.int64), nb - 1)
    ax, ay = np.abs(dirs[:, 0].astype(np.float64)), np.abs(dirs[:, 1].astype(np.float64))
    t = np.where(ax + ay > 0, ay / np.maximum(ax + ay, 1e-300), 0.0)
    xp, yp = dirs[:, 0] >= 0, dirs[:, 1] >= 0
    az = np.where(xp, np.where(yp, t, 4.0 - t), np.where(yp, 2.0 - t, 2.0 + t)) * 0.25
    az = np.clip(az, 0.0, 0.999999)
    az = np.where(band & 1, 0.999999 - az, az)
    key = (band << 20) | (az * 1048576.0).astype(np.int64)
    return np.argsort(key, kind="stable")


def main():
    w = sys.argv[1] if len(sys.argv) > 1 else "c3"
    c = dict(WORKLOADS[w])
    if len(sys.argv) > 2:
        c["num_paths"] = int(sys.argv[2])
    assert len(c["tx_pos"]) == 1, "study handles one TX"
    npth, nb = c["num_paths"], c["num_bounces"]
    r = oracle.compute_paths(*args(c))
    ex = r["extras"]
    order = launch_order(ex["launch_dirs"])
    rays = r["scat_rays"].reshape(nb + 1, npth, 6)
    out = dict(order=order.astype(np.uint32), rx_pos=np.asarray(c["rx_pos"], np.float32),
               tx_pos=np.asarray(c["tx_pos"], np.float32))
    alive = np.ones(npth, bool)
    cur = order
    for b in range(nb):
        ht = ex["hit_tri"][b, 0]
        hit = ht[cur] != oracle.NO_HIT
        cur = cur[hit]                      # stable compaction in launch order
        out["ray%d" % (b + 1)] = cur.astype(np.uint32)
        out["tri%d" % (b + 1)] = ht[cur]    # reference (mesh, face) order index
        out["o%d" % (b + 1)] = rays[b + 1, cur, :3]
        out["d%d" % (b + 1)] = rays[b + 1, cur, 3:]
        print("launch", b + 1, "live", len(cur), "oracle live", int(ex["live"][b + 1]) if b + 1 <= nb else -1)
    os.makedirs("/tmp/hrt_study", exist_ok=True)
    p = "/tmp/hrt_study/%s_%d.npz" % (w, npth)
    np.savez(p, **out)
    print("wrote", p)


if __name__ == "__main__":
    main()
