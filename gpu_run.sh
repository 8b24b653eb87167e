set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -C oracle liboracle.so > gpurun_out/build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_generated_scenes.py -m gpu -q -x 2>&1 | grep -vE "^$" | tail -30
