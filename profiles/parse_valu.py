#!/usr/bin/env python3
"""profiles/collect_all.sh <tag> <workload> -> profiles/<tag>_pmc_valu_<workload>.csv (trimmed to our kernels)
and profiles/pmc_valu.json: VALU instructions per step and SIMD-cycles per VALU instruction of the
fused, records, image, trace and shade kernels, which bench.py reports next to the HBM roofline (roofline.valu)."""
import csv
import glob
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
workload = sys.argv[2] if len(sys.argv) > 2 else "c3"
SIMDS = 256 * 4
XCDS = 8   # GRBM_GUI_ACTIVE comes back summed over the 8 XCD instances (8x kernel time x clock)

f = max(glob.glob(os.path.join(REPO, "gpurun_out", "pmc_%s_%s_VALU" % (tag, workload), "*", "*counter_collection.csv")),
        key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "hrt_" in r["Kernel_Name"]]
keep = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "VGPR_Count", "Counter_Name", "Counter_Value"]
with open(os.path.join(HERE, "%s_pmc_valu_%s.csv" % (tag, workload)), "w", newline="") as fo:
    w = csv.DictWriter(fo, keep)
    w.writeheader()
    for r in rows:
        w.writerow({k: r[k] for k in keep})


def launch0(name):
    m = re.search(r"hrt_fused_kernel<(\w+), (\d+), (\w+), (\d+)>", name)
    return bool(m) and m.group(3) == "true"


n_steps = len({r["Dispatch_Id"] for r in rows if launch0(r["Kernel_Name"])}) or \
    len({r["Dispatch_Id"] for r in rows if "los" in r["Kernel_Name"]})
out = {}
for kern in ("fused", "records", "image", "trace", "shade"):
    acc = {}
    for r in rows:
        if ("hrt_%s_kernel" % kern) in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if not acc:
        continue
    insts, cyc = acc["SQ_INSTS_VALU"] / n_steps, acc["GRBM_GUI_ACTIVE"] / n_steps
    out[kern] = dict(valu_insts_per_step=insts, busy_cycles_per_step_per_xcd=cyc / XCDS, cycles_per_valu_inst=SIMDS * (cyc / XCDS) / insts,
                     issue_frac_vs_simd32_peak=insts * 2.0 / (SIMDS * cyc / XCDS))
path = os.path.join(HERE, "pmc_valu.json")
allj = json.load(open(path)) if os.path.exists(path) else {}
allj[workload] = dict(
    kernels_sha16=hashlib.sha256(b"".join(open(os.path.join(REPO, "hermespy-rt_amd", "csrc", f), "rb").read() for f in ("hrt_kernels.hip", "hrt_fused_body.inc"))).hexdigest()[:16],
    round=tag, kernels=out,
    source="rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES (profiles/%s_pmc_valu_%s.csv); "
           "cycles_per_valu_inst = 1024 SIMDs * busy cycles per XCD / insts; the SIMD-32 peak is "
           "2 cycles per wave64 instruction, a busy chip sustains 2.3-2.9 on fma/mul/add, 4.2 on "
           "compares/min/max/DPP, 8.1 on transcendentals, 5.2 on f64 fma "
           "(profiles/microbench/r01_valu_issue.txt)" % (tag, workload))
json.dump(allj, open(path, "w"), indent=1)
print(json.dumps(allj[workload], indent=1))
