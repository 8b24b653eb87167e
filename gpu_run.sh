set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -C oracle liboracle.so libm_probe > gpurun_out/build.log 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/pytest_gpu.log | grep -vE "^$" | tail -5
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err || (tail -30 gpurun_out/b.err; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/b.json')); r=d['roofline']; print(round(d['ms_per_step'],3), 'trace', [round(x,3) for x in r['trace_kernel_ms']], 'shade', [round(x,3) for x in r['shade_kernel_ms']], 'scan', round(r['compaction_ms_per_step'],3), 'los', round(r['los_ms'],3))"
