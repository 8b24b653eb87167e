# Six parity fuzzers at once on one GPU (tests/fuzz_parity.py): bash gpu_fuzz_r4.sh [pass]   (pass k = fresh seed ranges)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
k=${1:-0}
run() { name=$1; shift; ( timeout -k 10 570 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -1 gpurun_out/$name.log)" > gpurun_out/$name.rc ) & }
run fz4_soups python tests/fuzz_parity.py soups $((340000 + 4000 * k)) $((344000 + 4000 * k))
run fz4_configs python tests/fuzz_parity.py configs $((80000 + 3000 * k)) $((83000 + 3000 * k))
run fz4_inplane python tests/fuzz_parity.py inplane $((60000 + 3000 * k)) $((63000 + 3000 * k))
run fz4_bigsoups python tests/fuzz_parity.py bigsoups $((9000 + 400 * k)) $((9400 + 400 * k))
run fz4_big python tests/fuzz_parity.py big $((6000 + 300 * k)) $((6300 + 300 * k))
run fz4_deepsoups python tests/fuzz_parity.py deepsoups $((2000 + 400 * k)) $((2400 + 400 * k))
wait
cat gpurun_out/fz4_*.rc
grep -l MISMATCH gpurun_out/fz4_*.log && exit 1
exit 0
