cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vae_train.py -q 2>&1 | tail -3
timeout 300 python tests/perf/vae_train_bench.py 32 128 512 2>&1 | grep -v -E "amdgpu.ids|Warning|detach|print" > gpurun_out/r3e_vae_train_bench.txt; cat gpurun_out/r3e_vae_train_bench.txt
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dit.py -q -x 2>&1 | tail -3
ROCPROF_ROWS=30 bash tools/rocprof_stats.sh r3e_train tests/perf/train_cliff.py 1024 > gpurun_out/r3e_train_b1024_kernel_stats.txt 2>&1; cut -c1-150 gpurun_out/r3e_train_b1024_kernel_stats.txt
