#!/bin/bash
# quick PMC look at one workload: VALU instruction count, busy cycles, wave-state counters per kernel
# (two rocprofv3 passes, --kernel-trace only beside --pmc).  bash profiles/pmc_quick.sh <workload> [env...]
set -e
w=${1:-c4}
export TMPDIR=/tmp
mkdir -p gpurun_out
for pass in A B; do
  if [ $pass = A ]; then ctr="SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS"; else ctr="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; fi
  d=gpurun_out/pq_${w}_$pass
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- \
      python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $d.json 2> $d.err \
      || (tail -20 $d.err; exit 1)
done
python3 - <<P
import csv, glob, collections
for p in "AB":
    f = max(glob.glob("gpurun_out/pq_${w}_%s/*/*counter_collection.csv" % p))
    rows = [r for r in csv.DictReader(open(f)) if "hrt_" in r["Kernel_Name"]]
    disp = collections.OrderedDict()
    for r in rows:
        d = int(r["Dispatch_Id"])
        disp.setdefault(d, {"k": r["Kernel_Name"].split("(")[0][-28:] + ("<" + r["Kernel_Name"].split("<")[1][:16] if "<" in r["Kernel_Name"] else ""), "grid": r["Grid_Size"]})[r["Counter_Name"]] = round(float(r["Counter_Value"]))
    los = [d for d in disp if "los" in disp[d]["k"] or "fused" in disp[d]["k"] and "true, 4" in disp[d]["k"]]
    ids = sorted(disp)
    # the dispatches of the last step: from the last launch-0 (largest grid first seen) on
    last0 = max(d for d in ids if disp[d]["grid"] == disp[ids[-1 if False else 0]]["grid"]) if ids else 0
    for d in ids:
        if d >= last0: print(p, d, disp[d])
P
