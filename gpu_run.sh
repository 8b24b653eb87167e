set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err || (tail -30 gpurun_out/bench_c3.err; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/bench_c3.json')); r=d['roofline']; print(round(d['ms_per_step'],3), '%.4g' % d['value'], '%.4g' % d['ray_tri_tests_per_sec'], 'trace', [round(x,3) for x in r['trace_kernel_ms']], 'shade', [round(x,3) for x in r['shade_kernel_ms']], r['frac'])"
export HRT_BENCH_REHEARSE=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --workload c2 2>gpurun_out/reh.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rehearse2', d['n_gpus'], round(d['ms_per_step'],3), d.get('gather',{}).get('ms'), d.get('gather_error'))"
