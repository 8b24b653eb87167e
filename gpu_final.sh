cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -x -m gpu > gpurun_out/pytest_full.log 2>&1 || { tail -40 gpurun_out/pytest_full.log; exit 1; }
tail -1 gpurun_out/pytest_full.log
export HRT_ACCEL_FINE_MIN=0 HRT_LDS_TRI_BYTES_MAX=0
( timeout -k 10 400 python tests/fuzz_parity.py soups 130000 132500 > gpurun_out/fz_w2_soups.log 2>&1; tail -1 gpurun_out/fz_w2_soups.log ) &
( HRT_WIDE_COS=2.0 timeout -k 10 400 python tests/fuzz_parity.py soups 132500 134500 > gpurun_out/fz_w2_soups_all.log 2>&1; tail -1 gpurun_out/fz_w2_soups_all.log ) &
( HRT_SORT_RAYS=1 timeout -k 10 400 python tests/fuzz_parity.py inplane 10000 11500 > gpurun_out/fz_w2_inplane.log 2>&1; tail -1 gpurun_out/fz_w2_inplane.log ) &
( HRT_SORT_RAYS=1 timeout -k 10 400 python tests/fuzz_parity.py bigsoups 3200 3400 > gpurun_out/fz_w2_bigsoups.log 2>&1; tail -1 gpurun_out/fz_w2_bigsoups.log ) &
( HRT_WIDE_CAP=5 HRT_SORT_RAYS=1 timeout -k 10 400 python tests/fuzz_parity.py configs 21000 22200 > gpurun_out/fz_w2_configs.log 2>&1; tail -1 gpurun_out/fz_w2_configs.log ) &
wait
