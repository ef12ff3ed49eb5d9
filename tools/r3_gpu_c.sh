cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_vae_train.py -q -x -rP 2>&1 | grep -E "^\[parity\]|passed|failed|^E  |^FAILED|^ERROR|Error" > gpurun_out/r3c_vae_train_tests.txt; tail -30 gpurun_out/r3c_vae_train_tests.txt | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_train.py -q -k "full_depth or bf16x3" -rP 2>&1 | grep -E "^\[parity\]|passed|failed|^E  |^FAILED" | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_vae.py -q 2>&1 | tail -3
timeout 300 python tests/perf/train_scale.py > gpurun_out/r3c_train_scale.txt 2>&1; grep -v amdgpu.ids gpurun_out/r3c_train_scale.txt
timeout 300 python tests/perf/train_scale.py nogc > gpurun_out/r3c_train_scale_nogc.txt 2>&1; grep -v amdgpu.ids gpurun_out/r3c_train_scale_nogc.txt
