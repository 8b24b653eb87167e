#!/bin/bash
# VALU + wave-state counters of one bench configuration (one GPU-box call):  bash profiles/pmc_quick2.sh <tag> [workload]
# -> gpurun_out/pmcq_<tag>_{VALU,WAVE}/ ; read with profiles/tools/pmc_table.py
set -e
tag=${1:-x}
w=${2:-c3}
export TMPDIR=/tmp
B="python3 bench.py --workload $w --no-cpu-baseline --no-end-to-end --event-steps 1 --steps 3 --warmup 1"
d=gpurun_out/pmcq_${tag}_VALU; rm -rf $d
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $d -- $B > $d.json 2> $d.err || (tail -20 $d.err; exit 1)
d=gpurun_out/pmcq_${tag}_WAVE; rm -rf $d
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES \
    --kernel-trace --output-format csv -d $d -- $B > $d.json 2> $d.err || (tail -20 $d.err; exit 1)
echo done $tag
