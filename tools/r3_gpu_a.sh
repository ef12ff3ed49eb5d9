cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x -rP 2>&1 | grep -E "^\[parity\]|^\[dopri5|passed|failed|^E |Error|^FAILED" > gpurun_out/r3a_gpu_tests.txt; tail -15 gpurun_out/r3a_gpu_tests.txt
timeout 300 python tests/perf/train_cliff.py 256 512 1024 > gpurun_out/r3a_cliff_seq.txt 2>&1
timeout 300 python tests/perf/train_cliff.py keep 256 512 1024 > gpurun_out/r3a_cliff_seq_keep.txt 2>&1
timeout 300 python tests/perf/train_cliff.py 512 256 512 > gpurun_out/r3a_cliff_seq2.txt 2>&1
cat gpurun_out/r3a_cliff_seq*.txt | grep -v amdgpu.ids
