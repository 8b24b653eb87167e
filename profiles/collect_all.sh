#!/bin/bash
# One GPU-box call for a round's profile evidence of ONE workload (run from the repo root):
#     bash profiles/collect_all.sh r03 c3        (c3 = the default; c2, c4, ... likewise)
#   1. rocprofv3 --kernel-trace --stats of `bench.py --workload W` -> gpurun_out/prof_<tag>_<W>/
#   2. PMC passes (separate runs, --kernel-trace only beside --pmc): FETCH_SIZE, WRITE_SIZE,
#      SQ_INSTS_VALU + GRBM_GUI_ACTIVE, wave-state counters
# then, back home:  python profiles/parse_pmc.py <tag> <W>; python profiles/parse_valu.py <tag> <W>
set -e
tag=${1:-rXX}
w=${2:-c3}
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python3 bench.py --workload $w --no-cpu-baseline --no-end-to-end --event-steps 1"
d=gpurun_out/prof_${tag}_$w
rm -rf $d
# (one stream for the kernel-stats run, HRT_OVERLAP=0: a kernel's duration is then its own and can be held against the
# per-kernel HIP events of the bench line, which come from a one-stream pass too.  In the timed region the records
# kernels run on a second stream beside the others and every duration stretches: second run, *_overlap)
HRT_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- $B --steps 20 --warmup 5 > $d.json 2> $d.err || (tail -20 $d.err; exit 1)
d=gpurun_out/prof_${tag}_${w}_overlap
rm -rf $d
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- $B --steps 20 --warmup 5 > $d.json 2> $d.err || (tail -20 $d.err; exit 1)
# (PMC passes: a kernel per launch -- HRT_TUNE=no_chain=1 -- so that every step of the pass, timed or with events, has
# the same kernel sequence and the per-launch sums mean what the bench line's per-launch times mean; the kernel-stats
# runs above keep the default, i.e. show hrt_chain_kernel where it is used: tables of <= 64 triangles, >= 4 bounces)
export HRT_TUNE=no_chain=1
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_${tag}_${w}_$c
  rm -rf $d
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- $B --steps 3 --warmup 1 --calibrate > $d.json 2> $d.err || (tail -20 $d.err; exit 1)
done
d=gpurun_out/pmc_${tag}_${w}_VALU
rm -rf $d
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $d -- $B --steps 3 --warmup 1 > $d.json 2> $d.err || (tail -20 $d.err; exit 1)
d=gpurun_out/pmc_${tag}_${w}_WAVE
rm -rf $d
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES \
    --kernel-trace --output-format csv -d $d -- $B --steps 3 --warmup 1 > $d.json 2> $d.err || (tail -20 $d.err; exit 1)
echo collected $tag $w
