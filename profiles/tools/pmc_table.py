#!/usr/bin/env python3
"""Per-kernel table of one step from profiles/pmc_quick2.sh output:  python profiles/tools/pmc_table.py <tag>"""
import csv, glob, os, sys, collections
tag = sys.argv[1]
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def load(kind):
    f = max(glob.glob(os.path.join(REPO, "gpurun_out", "pmcq_%s_%s" % (tag, kind), "*", "*counter_collection.csv")), key=os.path.getmtime)
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "hrt_" not in r["Kernel_Name"]: continue
        by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return by
V, Wv = load("VALU"), load("WAVE")
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:44]
keys = list(V.keys())
is0 = [i for i, k in enumerate(keys) if "hrt_fused_kernel" in k[1] and "true" in k[1].split(",")[2]]
last0 = max(is0) if is0 else 0
wk = list(Wv.keys())
print("%-46s %12s %10s %8s | %6s %6s %6s %6s" % ("kernel", "VALU insts", "cyc/XCD", "cyc/inst", "wait%", "winst%", "valu%", "ldsbc%"))
tot_i = tot_c = 0
for i, k in enumerate(keys[last0:]):
    v = V[k]
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    ins = max(v["SQ_INSTS_VALU"], 1)
    w = Wv.get(wk[last0 + i], {}) if last0 + i < len(wk) else {}
    wc = max(w.get("SQ_WAVE_CYCLES", 0), 1)
    print("%-46s %12d %10d %8.2f | %6.1f %6.1f %6.1f %6.1f" % (short(k[1]), ins, cyc, 1024 * cyc / ins, 100 * w.get("SQ_WAIT_ANY", 0) / wc,
          100 * w.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * w.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * w.get("SQ_LDS_BANK_CONFLICT", 0) / max(w.get("SQ_ACTIVE_INST_ANY", 1), 1)))
    tot_i += ins; tot_c += cyc
print("total VALU %d, cycles/XCD %d (%.3f ms at 2.4 GHz)" % (tot_i, tot_c, tot_c / 2.4e6))
