set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HRT_BENCH_REHEARSE=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 3 --steps 3 --warmup 1 --workload c2 > gpurun_out/rehearse3.json 2> gpurun_out/rehearse3.err || (tail -40 gpurun_out/rehearse3.err; exit 1)
cat gpurun_out/rehearse3.json
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --gather-in-step > gpurun_out/rehearse2.json 2> gpurun_out/rehearse2.err || (tail -40 gpurun_out/rehearse2.err; exit 1)
cat gpurun_out/rehearse2.json
