#!/usr/bin/env python3
"""Turn the two PMC passes of profiles/collect_pmc.sh into profiles/<tag>_pmc_*.csv (trimmed
to our kernels) and profiles/pmc_traffic.json (corrected HBM bytes per launch of the bounce
kernel), which bench.py reports as roofline.traffic for the same workload."""
import csv
import glob
import json
import os
import hashlib
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
workload = sys.argv[2] if len(sys.argv) > 2 else "c3"
TRUE_KIB = 131072.0   # bench.py --calibrate: 32 Mi floats read, 32 Mi floats written

raw = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    # gpurun merges results into gpurun_out/ without deleting older ones: take the newest
    f = max(glob.glob(os.path.join(REPO, "gpurun_out", "pmc_%s_%s" % (tag, ctr), "*", "*counter_collection.csv")),
            key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "hrt_" in r["Kernel_Name"]]
    keep = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
            "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(os.path.join(HERE, "%s_pmc_%s_%s.csv" % (tag, ctr.lower(), workload)), "w", newline="") as fo:
        w = csv.DictWriter(fo, keep)
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in keep})
    cal = [float(r["Counter_Value"]) for r in rows if "selftest" in r["Kernel_Name"]][0]
    n_steps = len([r for r in rows if "los" in r["Kernel_Name"]])
    # one "launch" of the bounce = trace kernel + shade kernel (+ the scan/move kernels of the
    # stable compaction that follow it): sum their counters per trace-kernel dispatch
    per, kinds = [], {}
    for r in rows:
        n = r["Kernel_Name"]
        if "selftest" in n or "los" in n or not any(t in n for t in ("trace_kernel", "shade_kernel", "scan", "move")):
            continue   # (problem-creation kernels -- hrt_rxt_build_kernel -- are not part of a step)
        if "trace" in n:
            per.append(0.0)
        per[-1] += float(r["Counter_Value"])
        k = "trace" if "trace" in n else "shade" if "shade" in n else "compaction"
        kinds[k] = kinds.get(k, 0.0) + float(r["Counter_Value"]) / n_steps
    per_step = len(per) // n_steps
    raw[ctr] = dict(corr=TRUE_KIB / cal, per_launch=[sum(per[i::per_step]) / n_steps for i in range(per_step)],
                    by_kernel_KiB_per_step=kinds)

fetch = [x * raw["FETCH_SIZE"]["corr"] * 1024 for x in raw["FETCH_SIZE"]["per_launch"]]
write = [x * raw["WRITE_SIZE"]["corr"] * 1024 for x in raw["WRITE_SIZE"]["per_launch"]]
path = os.path.join(HERE, "pmc_traffic.json")
allj = json.load(open(path)) if os.path.exists(path) else {}
allj[workload] = dict(
    kernels_sha16=hashlib.sha256(open(os.path.join(REPO, 'hermespy-rt_amd', 'csrc', 'hrt_kernels.hip'), 'rb').read()).hexdigest()[:16],
    n_gpus=1, kernel="hrt_trace_kernel + hrt_shade_kernel (+ scan/move of the compaction)", round=tag,
    by_kernel_bytes_per_step={k: dict(fetch=raw["FETCH_SIZE"]["by_kernel_KiB_per_step"].get(k, 0) * raw["FETCH_SIZE"]["corr"] * 1024,
                                      write=raw["WRITE_SIZE"]["by_kernel_KiB_per_step"].get(k, 0) * raw["WRITE_SIZE"]["corr"] * 1024)
                              for k in ("trace", "shade", "compaction")},
    source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (profiles/%s_pmc_*_%s.csv); "
           "KiB counters, calibrated on a known-traffic launch of the same access pattern: "
           "FETCH_SIZE x%.3f (gfx950 half-count), WRITE_SIZE x%.3f" % (tag, workload, raw["FETCH_SIZE"]["corr"], raw["WRITE_SIZE"]["corr"]),
    fetch_bytes_per_launch=fetch, write_bytes_per_launch=write,
    hbm_bytes_per_step=sum(fetch) + sum(write), hbm_bytes_per_launch_avg=(sum(fetch) + sum(write)) / len(fetch))
json.dump(allj, open(path, "w"), indent=1)
print(json.dumps(allj[workload], indent=1))
