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
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen=set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hrt_" not in k: continue
        k = k.split("(")[0][-60:] + " g" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[k]+=1
    for k in acc:
        print(p, k, "n=%d" % n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
P
