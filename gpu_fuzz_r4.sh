set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
fail=0
run() { name=$1; shift; ( timeout -k 10 570 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -1 gpurun_out/$name.log)" > gpurun_out/$name.rc ) & }
run fz4_soups python tests/fuzz_parity.py soups 322000 326000
run fz4_configs python tests/fuzz_parity.py configs 63000 66000
run fz4_inplane python tests/fuzz_parity.py inplane 53000 56000
run fz4_bigsoups python tests/fuzz_parity.py bigsoups 7700 8100
run fz4_big python tests/fuzz_parity.py big 4300 4600
run fz4_deepsoups python tests/fuzz_parity.py deepsoups 800 1200
wait
cat gpurun_out/fz4_*.rc
grep -l MISMATCH gpurun_out/fz4_*.log && exit 1
exit 0
