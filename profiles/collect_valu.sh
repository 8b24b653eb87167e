#!/bin/bash
# VALU issue rate of the two kernels from PMC counters (own rocprofv3 pass, --kernel-trace only
# beside --pmc): SQ_INSTS_VALU = VALU instructions issued (per wave), GRBM_GUI_ACTIVE = busy cycles
# of the dispatch (summed over the 8 XCDs).  cycles per VALU instruction per SIMD =
# 1024 SIMDs * (GRBM_GUI_ACTIVE / 8) / SQ_INSTS_VALU; the SIMD-32's own rate is 2 cycles per wave64
# instruction, what a busy chip sustains per instruction class is in profiles/microbench/.
# Run on the GPU box from the repo root:   bash profiles/collect_valu.sh <tag>
# then, back home:                          python profiles/parse_valu.py <tag>
set -e
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
d=gpurun_out/pmc_${tag}_VALU
rm -rf $d
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $d -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-gather > $d.json 2> $d.err \
    || (tail -20 $d.err; exit 1)
