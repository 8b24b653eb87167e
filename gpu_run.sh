set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -C oracle liboracle.so > gpurun_out/build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_dense_parity.py tests/test_gpu_full_size.py tests/test_generated_scenes.py tests/test_gpu_device_api.py -m gpu -q -x 2>&1 | grep -vE "^$" | tail -4
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err || (tail -30 gpurun_out/b.err; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/b.json')); r=d['roofline']; print(round(d['ms_per_step'],3), 'trace', [round(x,3) for x in r['trace_kernel_ms']], 'shade', [round(x,3) for x in r['shade_kernel_ms']])"
