set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -C oracle liboracle.so libm_probe > gpurun_out/build.log 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log | grep -vE "^$" | tail -5
timeout -k 10 600 python bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err || (tail -30 gpurun_out/bench_c3.err; exit 1)
cat gpurun_out/bench_c3.json
