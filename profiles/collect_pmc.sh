#!/bin/bash
# HBM traffic of the bounce kernel from PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md
# (HBM / rocprofv3) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slot limit),
# --kernel-trace only beside --pmc, counters in KiB, FETCH_SIZE calibrated on a launch of known
# traffic in the same access pattern (bench.py --calibrate: 128 MiB read + 128 MiB written,
# 4 B/lane coalesced) -- on gfx950 it reads exactly half.
# Run on the GPU box from the repo root:   bash profiles/collect_pmc.sh <tag>   (e.g. r01)
# then, back home:                          python profiles/parse_pmc.py <tag>
set -e
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_${tag}_$c
  rm -rf $d
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- \
      python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --calibrate > $d.json 2> $d.err \
      || (tail -20 $d.err; exit 1)
done
