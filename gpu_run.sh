set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle liboracle.so libm_probe > gpurun_out/build.log 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=5 2>&1 | tee gpurun_out/pytest_gpu.log | grep -vE "^$" | tail -12
timeout -k 10 600 python bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err || (tail -30 gpurun_out/bench_c3.err; exit 1)
rm -rf gpurun_out/prof_r01
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err || (tail -30 gpurun_out/bench_prof.err; exit 1)
bash profiles/collect_pmc.sh r01
python bench.py --dropin > gpurun_out/dropin_c3.json
for w in c1 c2 c4; do python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$w.json 2>/dev/null || echo "bench $w failed"; done
