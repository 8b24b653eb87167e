#!/usr/bin/env python3
"""Design study: splitting a bounce wave by reflection sequence (= virtual source) before packet culling.
    python profiles/study/split_study.py /tmp/hrt_study/c3_4000000.npz"""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import hermespy_rt_amd  # noqa
from hermespy_rt_amd.workloads import WORKLOADS
from oracle import oracle
so = "/tmp/hrt_study/libcand.so"
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", os.path.join(HERE, "cand.c"), "-o", so, "-lm"])
L = C.CDLL(so)
import importlib.util
spec = importlib.util.spec_from_file_location("study", os.path.join(HERE, "study.py"))
D = np.load(sys.argv[1])
c = WORKLOADS["c3"]
# rows
sys.argv = [sys.argv[0]]
flat = oracle.flatten(oracle.read_hrt(c["scene_path"]))
st = {"__file__": os.path.join(HERE, "study.py")}
exec(open(os.path.join(HERE, "study.py")).read().split("def main")[0], st)
rows, _ = st["rows_of"](c["scene_path"])
T = len(rows)
P = st["P"]
f64p = C.POINTER(C.c_double)
np4 = len(D["order"])
seq = np.zeros(np4, np.int64)
for b in (1, 2, 3):
    r = D["ray%d" % b]
    t = np.full(np4, 0, np.int64); t[r] = D["tri%d" % b]
    seq = seq * 256 + t     # (triangle sequence: finer than planes, fine for grouping)
    o = np.ascontiguousarray(D["o%d" % b], np.float32); d = np.ascontiguousarray(D["d%d" % b], np.float32)
    gid = np.ascontiguousarray((seq[r] % (2**31 - 1)).astype(np.int32))
    for mode, name in ((0, "no split"), (1, "split wide (cos<0.99)"), (2, "always split")):
        out = np.zeros(8)
        L.bounce_eval_split(P(rows), T, P(o), P(d), gid.ctypes.data_as(C.POINTER(C.c_int32)), len(o), mode, C.c_float(0.99),
                            C.c_double(700.0), C.c_double(51.0), P(out, f64p))
        print("launch %d bounce %-22s: cost/wave %7.0f  groups/wave %.2f  cands/wave %.1f  unusable/wave %.3f" % (
            b, name, out[1] / out[0], out[2] / out[0], out[3] / out[0], out[4] / out[0]))
