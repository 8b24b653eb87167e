#!/usr/bin/env python3
"""profiles/collect_all.sh <tag> <workload> -> profiles/<tag>_pmc_{fetch,write}_size_<workload>.csv (trimmed to
our kernels), profiles/<tag>_pmc_wave_<workload>.csv, profiles/<tag>_bench_<workload>_kernel_stats.csv and
profiles/pmc_traffic.json: corrected HBM bytes per launch, which bench.py reports as roofline.traffic
for the same workload.  A launch = one hrt_fused_kernel dispatch, or an hrt_trace_kernel dispatch and
the hrt_shade_kernel (+ re-sort kernels) behind it; with patch tables the launch begins with its hrt_records_kernel."""
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
workload = sys.argv[2] if len(sys.argv) > 2 else "c3"
TRUE_KIB = 131072.0   # bench.py --calibrate: 32 Mi floats read, 32 Mi floats written


def newest(pattern):
    # gpurun merges results into gpurun_out/ without deleting older ones: take the newest
    return max(glob.glob(os.path.join(REPO, "gpurun_out", pattern)), key=os.path.getmtime)


def kind(name):
    if "hrt_fused_kernel" in name:
        return "fused"
    if "hrt_chain_kernel" in name:   # (not in the PMC passes: collect_all.sh runs them with no_chain=1)
        return "chain"
    if "hrt_records_kernel" in name:
        return "records"
    if "hrt_image_kernel" in name:
        return "image"
    if "hrt_trace_kernel" in name:
        return "trace"
    if "hrt_shade_kernel" in name:
        return "shade"
    if "hrt_sort" in name or "radix" in name.lower():
        return "sort"
    return None


def is_launch0(name):
    m = re.search(r"hrt_fused_kernel<(\w+), (\d+), (\w+), (\d+)>", name)
    return bool(m) and m.group(3) == "true"


keep = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
        "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
raw = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(os.path.join("pmc_%s_%s_%s" % (tag, workload, ctr), "*", "*counter_collection.csv"))
    rows = [r for r in csv.DictReader(open(f)) if "hrt_" in r["Kernel_Name"]]
    with open(os.path.join(HERE, "%s_pmc_%s_%s.csv" % (tag, ctr.lower(), workload)), "w", newline="") as fo:
        w = csv.DictWriter(fo, keep)
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in keep})
    cal = [float(r["Counter_Value"]) for r in rows if "selftest" in r["Kernel_Name"]][0]
    rows = [r for r in rows if kind(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    n_steps = sum(1 for r in rows if is_launch0(r["Kernel_Name"])) or sum(1 for r in rows if "los" in r["Kernel_Name"])
    per, kinds, prev = [], {}, None
    for r in rows:
        k = kind(r["Kernel_Name"])
        # a launch begins with its fused kernel, with its records kernel (patch tables: shadow traces + records), or
        # with the kernel of its primary rays when no records kernel went before
        if k in ("fused", "chain", "records") or (k in ("trace", "image") and prev != "records"):
            per.append(0.0)
        prev = k
        per[-1] += float(r["Counter_Value"])
        kinds[k] = kinds.get(k, 0.0) + float(r["Counter_Value"]) / n_steps
    per_step = len(per) // n_steps
    raw[ctr] = dict(corr=TRUE_KIB / cal, per_launch=[sum(per[i::per_step]) / n_steps for i in range(per_step)],
                    by_kernel_KiB_per_step=kinds)

fetch = [x * raw["FETCH_SIZE"]["corr"] * 1024 for x in raw["FETCH_SIZE"]["per_launch"]]
write = [x * raw["WRITE_SIZE"]["corr"] * 1024 for x in raw["WRITE_SIZE"]["per_launch"]]
path = os.path.join(HERE, "pmc_traffic.json")
allj = json.load(open(path)) if os.path.exists(path) else {}
allj[workload] = dict(
    kernels_sha16=hashlib.sha256(b"".join(open(os.path.join(REPO, "hermespy-rt_amd", "csrc", f), "rb").read() for f in ("hrt_kernels.hip", "hrt_fused_body.inc"))).hexdigest()[:16],
    n_gpus=1, kernel="per launch: hrt_fused_kernel, or hrt_records_kernel + hrt_image_kernel / hrt_trace_kernel + hrt_shade_kernel", round=tag,
    by_kernel_bytes_per_step={k: dict(fetch=raw["FETCH_SIZE"]["by_kernel_KiB_per_step"].get(k, 0) * raw["FETCH_SIZE"]["corr"] * 1024,
                                      write=raw["WRITE_SIZE"]["by_kernel_KiB_per_step"].get(k, 0) * raw["WRITE_SIZE"]["corr"] * 1024)
                              for k in ("fused", "records", "image", "trace", "shade", "sort")},
    source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (profiles/%s_pmc_*_%s.csv); "
           "KiB counters, calibrated on a known-traffic launch of the same access pattern: "
           "FETCH_SIZE x%.3f (gfx950 half-count), WRITE_SIZE x%.3f" % (tag, workload, raw["FETCH_SIZE"]["corr"], raw["WRITE_SIZE"]["corr"]),
    fetch_bytes_per_launch=fetch, write_bytes_per_launch=write,
    hbm_bytes_per_step=sum(fetch) + sum(write), hbm_bytes_per_launch_avg=(sum(fetch) + sum(write)) / len(fetch))
json.dump(allj, open(path, "w"), indent=1)
print(json.dumps(allj[workload], indent=1))

# wave-state counters, trimmed; kernel stats of the --stats run
try:
    f = newest(os.path.join("pmc_%s_%s_WAVE" % (tag, workload), "*", "*counter_collection.csv"))
    rows = [r for r in csv.DictReader(open(f)) if kind(r["Kernel_Name"])]
    with open(os.path.join(HERE, "%s_pmc_wave_%s.csv" % (tag, workload)), "w", newline="") as fo:
        w = csv.DictWriter(fo, ["Dispatch_Id", "Grid_Size", "Kernel_Name", "VGPR_Count", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in w.fieldnames})
    f = newest(os.path.join("prof_%s_%s" % (tag, workload), "*", "*kernel_stats.csv"))
    shutil.copy(f, os.path.join(HERE, "%s_bench_%s_kernel_stats.csv" % (tag, workload)))
    try:   # the same with the records kernels on their second stream (durations of concurrent kernels stretch)
        f = newest(os.path.join("prof_%s_%s_overlap" % (tag, workload), "*", "*kernel_stats.csv"))
        shutil.copy(f, os.path.join(HERE, "%s_bench_%s_kernel_stats_overlap.csv" % (tag, workload)))
    except ValueError:
        pass
except ValueError as e:
    print("no wave / stats pass found:", e)
