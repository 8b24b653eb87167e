#!/usr/bin/env python3
"""Design study driver (CPU): candidate statistics of per-ray table lookups against today's packet
culling, on the real waves of a workload (profiles/study/live_lists.py writes them).
    python profiles/study/study.py /tmp/hrt_study/c3_4000000.npz [ratio] """
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import hermespy_rt_amd  # noqa: E402,F401
from hermespy_rt_amd.workloads import WORKLOADS  # noqa: E402
from oracle import oracle  # noqa: E402

os.makedirs("/tmp/hrt_study", exist_ok=True)
so = "/tmp/hrt_study/libcand.so"
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", os.path.join(HERE, "cand.c"), "-o", so, "-lm"])
L = C.CDLL(so)
f32p, u64p, i64p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_double)


def P(a, t=f32p):
    return a.ctypes.data_as(t)


def rows_of(scene_path):
    """the product's 20-float triangle rows (problem.c), reference order"""
    flat = oracle.flatten(oracle.read_hrt(scene_path))
    v = flat["tri_vtx"].reshape(-1, 3, 3).astype(np.float32)
    T = len(v)
    rows = np.zeros((T, 20), np.float32)
    v1, e1, e2 = v[:, 0], v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    n = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    nl = np.linalg.norm(n, axis=1, keepdims=True)
    rows[:, 0:3], rows[:, 3:6], rows[:, 6:9] = v1, e1, e2
    rows[:, 9:12] = (n / np.maximum(nl, 1e-300)).astype(np.float32)
    l1 = np.linalg.norm(e1, axis=1) * 1.000001
    l2 = np.linalg.norm(e2, axis=1) * 1.000001
    l3 = np.linalg.norm(e2 - e1, axis=1) * 1.000001
    ln = nl[:, 0] * 1.000001
    eps, up = 1.1920928955078125e-07, 1.000001
    Ed = 16.0 * eps * l1 * l2
    c2 = 4.0 * eps * (1.0001 * ln + Ed)
    rows[:, 12] = Ed * up
    rows[:, 13] = c2 * up
    rows[:, 14] = (2 * c2 + 2 * Ed + 4e-6 * ln) * up
    rows[:, 15], rows[:, 16], rows[:, 17], rows[:, 18] = l1, l2, l3, ln
    return np.ascontiguousarray(rows), flat


def main():
    path = sys.argv[1]
    ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
    w = os.path.basename(path).split("_")[0]
    c = WORKLOADS[w]
    D = np.load(path)
    rows, flat = rows_of(c["scene_path"])
    T = len(rows)
    W = (T + 63) // 64
    NC = 6 * 48 * 48
    axis = np.zeros((NC, 3), np.float32)
    cs = np.zeros((NC, 2), np.float32)
    L.cell_cones(P(axis), P(cs), C.c_double(0.0))
    # radial bins: [0, r_min, r_min * ratio, ...] up to beyond the scene
    edges = [0.0, 0.25]
    while edges[-1] < 400.0:
        edges.append(edges[-1] * ratio)
    edges[-1] = 1e30
    redge = np.asarray(edges, np.float32)
    nr = len(redge) - 1
    print("T", T, "cells", NC, "rbins", nr, "table bytes per apex", NC * nr * W * 8)
    rx = D["rx_pos"]
    tables = []
    for k in range(len(rx)):
        m = np.zeros((NC, nr, W), np.uint64)
        L.build_table(P(rows), T, P(rx[k]), C.c_float(1e-5), 1, P(axis), P(cs), P(redge), nr, P(m, u64p))
        pc = np.zeros(m.shape[:2])
        for q in range(W):
            pc += np.array([bin(int(x)).count("1") for x in m[:, :, q].ravel()]).reshape(NC, nr)
        print("rx", k, "table: mean candidates per key %.1f" % pc.mean(), "max", pc.max())
        tables.append(m)
    tot = np.zeros(6)
    for b in (1, 2, 3, 4):
        o = np.ascontiguousarray(D["o%d" % b], np.float32)
        n = len(o)
        hm, hu, hp = np.zeros(257, np.int64), np.zeros(257, np.int64), np.zeros(257, np.int64)
        acc = np.zeros(6)
        for k in range(len(rx)):
            out = np.zeros(8)
            L.shadow_eval(P(rows), T, P(o), n, P(rx[k]), P(tables[k], u64p), P(redge), nr, P(out, f64p), P(hm, i64p),
                          P(hu, i64p), P(hp, i64p))
            acc += out[:6]
        nw = acc[0]
        print("launch %d shadow: wave-traces %d | per-lane mean %.2f | max-over-lanes %.2f | union %.2f | packet(today) %.2f | "
              "unusable %.3f" % (b, nw, acc[1] / (n * len(rx)), acc[2] / nw, acc[3] / nw, acc[4] / nw, acc[5] / nw))
        q = lambda h, p: int(np.searchsorted(np.cumsum(h) / h.sum(), p))
        print("    pct 50/90/99: max-over-lanes %d/%d/%d  union %d/%d/%d  packet %d/%d/%d" % (
            q(hm, .5), q(hm, .9), q(hm, .99), q(hu, .5), q(hu, .9), q(hu, .99), q(hp, .5), q(hp, .9), q(hp, .99)))
        # a wave picks the cheaper walk: union walked wave-uniformly (~51 VALU per candidate) or per lane (~90)
        tot += acc
    nw = tot[0]
    print("ALL shadow: per-lane mean %.2f | max-over-lanes %.2f | union %.2f | packet(today) %.2f" % (
        tot[1] / (nw * 64), tot[2] / nw, tot[3] / nw, tot[4] / nw))
    # ---- patch tables ----
    class PD(C.Structure):
        _fields_ = [("nu", C.c_int), ("nv", C.c_int), ("base", C.c_int)]
    for size in (float(x) for x in os.environ.get("PATCH_SIZES", "2,1,0.5").split(",")):
        slack = 2e-3
        pd = (PD * T)()
        base = 0
        for a in range(T):
            nu = int(min(512, max(1, np.ceil(rows[a, 15] / size))))
            nv = int(min(512, max(1, np.ceil(rows[a, 16] / size))))
            pd[a].nu, pd[a].nv, pd[a].base = nu, nv, base
            base += nu * nv
        npatch = base
        print("patch size %.2f m: %d patches, %.1f MB per apex (masks)" % (size, npatch, npatch * W * 8 / 1e6))
        tot = np.zeros(6)
        ptabs = []
        for k in range(len(rx)):
            m = np.zeros((npatch, W), np.uint64)
            L.build_patch_table(P(rows), T, pd, npatch, P(rx[k]), 0, C.c_float(1e-5), 1, C.c_float(slack), P(m, u64p))
            ptabs.append(m)
        for b in (1, 2, 3, 4):
            o = np.ascontiguousarray(D["o%d" % b], np.float32)
            tri = np.ascontiguousarray(D["tri%d" % b], np.uint32)
            n = len(o)
            hm, hu = np.zeros(257, np.int64), np.zeros(257, np.int64)
            acc = np.zeros(6)
            for k in range(len(rx)):
                out = np.zeros(8)
                L.patch_eval(P(rows), T, pd, P(o), P(tri, C.POINTER(C.c_uint32)), n, P(ptabs[k], u64p), C.c_float(slack), P(out, f64p),
                             P(hm, i64p), P(hu, i64p))
                acc += out[:6]
            nw = acc[0]
            q = lambda h, p: int(np.searchsorted(np.cumsum(h) / h.sum(), p))
            print("  launch %d shadow: per-lane mean %.2f | max-over-lanes %.2f | union %.2f | unserved lanes %.5f | pct 50/90/99 max %d/%d/%d union %d/%d/%d" % (
                b, acc[1] / (n * len(rx)), acc[2] / nw, acc[3] / nw, acc[5] / (n * len(rx)), q(hm, .5), q(hm, .9), q(hm, .99), q(hu, .5), q(hu, .9), q(hu, .99)))
            tot += acc
        nw = tot[0]
        print("  ALL shadow: per-lane mean %.2f | max-over-lanes %.2f | union %.2f" % (tot[1] / (nw * 64), tot[2] / nw, tot[3] / nw))
        # launch-1 bounce rays: they left the TX and reflected off A: apex = image of the TX in A's plane
        tx = D["tx_pos"][0].astype(np.float64)
        img = np.zeros((T, 3), np.float32)
        for a in range(T):
            nrm = rows[a, 9:12].astype(np.float64)
            img[a] = tx - 2.0 * np.dot(tx - rows[a, 0:3].astype(np.float64), nrm) * nrm
        m = np.zeros((npatch, W), np.uint64)
        L.build_patch_table(P(rows), T, pd, npatch, P(img), 1, C.c_float(1e-3), 0, C.c_float(slack), P(m, u64p))
        o = np.ascontiguousarray(D["o1"], np.float32)
        tri = np.ascontiguousarray(D["tri1"], np.uint32)
        out = np.zeros(8)
        hm, hu = np.zeros(257, np.int64), np.zeros(257, np.int64)
        L.patch_eval(P(rows), T, pd, P(o), P(tri, C.POINTER(C.c_uint32)), len(o), P(m, u64p), C.c_float(slack), P(out, f64p), P(hm, i64p), P(hu, i64p))
        print("  launch 1 bounce (image apex): per-lane mean %.2f | max-over-lanes %.2f | union %.2f | pct 50/90/99 max %d/%d/%d union %d/%d/%d" % (
            out[1] / len(o), out[2] / out[0], out[3] / out[0], q(hm, .5), q(hm, .9), q(hm, .99), q(hu, .5), q(hu, .9), q(hu, .99)))
    # bounce traces today
    for b in (1, 2, 3):
        o = np.ascontiguousarray(D["o%d" % b], np.float32)
        d = np.ascontiguousarray(D["d%d" % b], np.float32)
        out = np.zeros(8)
        hp = np.zeros(257, np.int64)
        L.bounce_eval_packet(P(rows), T, P(o), P(d), len(o), P(out, f64p), P(hp, i64p))
        print("launch %d bounce: wave-traces %d packet(today) %.2f unusable %.3f" % (b, out[0], out[4] / out[0], out[5] / out[0]))


if __name__ == "__main__":
    main()
