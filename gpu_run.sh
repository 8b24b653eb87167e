set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -C oracle liboracle.so > gpurun_out/build.log 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_dense_parity.py -m gpu -q -x 2>&1 | tee gpurun_out/pytest_gpu.log | grep -vE "^$" | tail -4
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_v2.json 2> gpurun_out/bench_v2.err || (tail -30 gpurun_out/bench_v2.err; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/bench_v2.json')); print(d['ms_per_step'], d['ray_tri_tests_per_sec'], d['roofline']['per_launch_ms'], d['roofline']['compaction_ms_per_step'])"
