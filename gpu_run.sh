set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -C oracle liboracle.so > gpurun_out/build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_dense_parity.py tests/test_gpu_golden.py tests/test_gpu_pybind.py -m gpu -q -x 2>&1 | grep -vE "^$" | tail -4
python bench.py --dropin | tee gpurun_out/dropin_c3.json
python bench.py --dropin | python -c "import json,sys; d=json.load(sys.stdin); print({k:round(v,4) for k,v in d.items() if k.startswith('t_')})"
