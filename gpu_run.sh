set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r01
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err || (tail -30 gpurun_out/bench_prof.err; exit 1)
bash profiles/collect_pmc.sh r01
python bench.py --dropin > gpurun_out/dropin_c3.json
cat gpurun_out/dropin_c3.json
