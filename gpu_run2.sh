set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_sq2
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq.json 2> gpurun_out/pmc_sq.err || (tail -20 gpurun_out/pmc_sq.err; exit 1)
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq2.json 2> gpurun_out/pmc_sq2.err || (tail -20 gpurun_out/pmc_sq2.err; exit 1)
