cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[parity\]|^\[dopri5|passed|failed|^E  |^FAILED|^ERROR" > gpurun_out/r3b_gpu_tests.txt; tail -25 gpurun_out/r3b_gpu_tests.txt | cut -c1-300
timeout 300 python bench.py --precision fp16 --steps 2 --no-extra --no-cpu-baseline > gpurun_out/r3b_bench_fp16.json 2>/dev/null; cut -c1-1500 gpurun_out/r3b_bench_fp16.json
timeout 300 python bench.py --precision bf16 --steps 2 --no-extra --no-cpu-baseline > gpurun_out/r3b_bench_bf16.json 2>/dev/null; cut -c1-1500 gpurun_out/r3b_bench_bf16.json
timeout 300 python tests/perf/train_scale.py > gpurun_out/r3b_train_scale.txt 2>&1; grep -v amdgpu.ids gpurun_out/r3b_train_scale.txt
