set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
fail=0
run() { name=$1; shift; ( timeout -k 10 570 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -1 gpurun_out/$name.log)" > gpurun_out/$name.rc ) & }
run fz4_soups python tests/fuzz_parity.py soups 318000 322000
run fz4_configs python tests/fuzz_parity.py configs 60000 63000
run fz4_inplane python tests/fuzz_parity.py inplane 50000 53000
run fz4_bigsoups python tests/fuzz_parity.py bigsoups 7300 7700
run fz4_big python tests/fuzz_parity.py big 4000 4300
run fz4_deepsoups python tests/fuzz_parity.py deepsoups 400 800
wait
cat gpurun_out/fz4_*.rc
grep -l MISMATCH gpurun_out/fz4_*.log && exit 1
exit 0
