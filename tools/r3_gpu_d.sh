cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vae_train.py -q -rP 2>&1 | grep -E "^\[parity\]|passed|failed|^E  |^FAILED|^ERROR" > gpurun_out/r3d_vae_train_tests.txt; tail -14 gpurun_out/r3d_vae_train_tests.txt | cut -c1-300
timeout 300 python tests/perf/vae_train_bench.py 32 128 > gpurun_out/r3d_vae_train_bench.txt 2>&1; grep -v amdgpu.ids gpurun_out/r3d_vae_train_bench.txt
ROCPROF_ROWS=24 bash tools/rocprof_stats.sh r3d_vae_train tests/perf/vae_train_bench.py 32 > gpurun_out/r3d_vae_train_kernel_stats.txt 2>&1; cat gpurun_out/r3d_vae_train_kernel_stats.txt | cut -c1-200
