#!/bin/bash
# One GPU-box call for a round's profile evidence (run from the repo root):  bash profiles/collect_all.sh r02
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command -> gpurun_out/prof_<tag>/
#   2. PMC passes (separate runs, --kernel-trace only beside --pmc): FETCH_SIZE, WRITE_SIZE (collect_pmc.sh),
#      SQ_INSTS_VALU + GRBM_GUI_ACTIVE (collect_valu.sh), wave-state counters of the trace kernel
# then, back home:  python profiles/parse_pmc.py <tag>; python profiles/parse_valu.py <tag>
set -e
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
d=gpurun_out/prof_$tag
rm -rf $d
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- \
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $d.json 2> $d.err || (tail -20 $d.err; exit 1)
bash profiles/collect_pmc.sh $tag
bash profiles/collect_valu.sh $tag
d=gpurun_out/pmc_${tag}_WAVE
rm -rf $d
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES \
    --kernel-trace --output-format csv -d $d -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-gather > $d.json 2> $d.err || (tail -20 $d.err; exit 1)
echo collected $tag
