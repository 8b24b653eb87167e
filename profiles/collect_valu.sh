#!/bin/bash
# VALU issue utilisation of the two kernels from PMC counters (own rocprofv3 pass, --kernel-trace
# only beside --pmc): SQ_INSTS_VALU = VALU instructions issued (per wave), GRBM_GUI_ACTIVE = cycles
# the GPU was busy during the dispatch.  A wave64 VALU instruction occupies its SIMD for 4 cycles,
# an MI355X has 256 CUs x 4 SIMDs, so   issue fraction = SQ_INSTS_VALU * 4 / (1024 * GRBM_GUI_ACTIVE / 8)
# (the counter comes back summed over the 8 XCDs)
# (double-precision instructions take 8 cycles: the shade kernel's figure is a lower bound).
# Run on the GPU box from the repo root:   bash profiles/collect_valu.sh <tag>
# then, back home:                          python profiles/parse_valu.py <tag>
set -e
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
d=gpurun_out/pmc_${tag}_VALU
rm -rf $d
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $d -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gather > $d.json 2> $d.err \
    || (tail -20 $d.err; exit 1)
