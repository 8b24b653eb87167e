set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_launch_dirs.py -m gpu -q -x -s 2>&1 | grep -vE "^$" | tail -14
