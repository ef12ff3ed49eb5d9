cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -q -rP 2>&1 | grep -E "^\[parity\]|passed|failed|^E  |^FAILED" | cut -c1-400
for wb in 1; do
  SCLDM_WGRAD_BATCH=$wb timeout 300 python bench.py --workload replogle_train_ditl_b1024 --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WGRAD_BATCH=$wb b1024', round(j['ms_per_step'],2), 'ms', round(j['train_tflops_per_gpu'],1), 'TF')"
  SCLDM_WGRAD_BATCH=$wb timeout 300 python bench.py --workload replogle_train_ditl_b256 --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WGRAD_BATCH=$wb b256', round(j['ms_per_step'],2), 'ms', round(j['train_tflops_per_gpu'],1), 'TF')"
done > gpurun_out/r3g2_ditl_ab.txt 2>&1
cat gpurun_out/r3g2_ditl_ab.txt
ROCPROF_ROWS=22 bash tools/rocprof_stats.sh r3g2_ditl1024 tests/perf/train_ditl_profile.py 1024 > gpurun_out/r3g2_train_ditl_b1024_kernel_stats.txt 2>&1; cut -c1-140 gpurun_out/r3g2_train_ditl_b1024_kernel_stats.txt
