#!/bin/bash
# round 3, run L: LDS-DMA GEMM with the vector epilogue, transposed weight copies for the data gradients, bf16 dhid
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3l_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3l_tests.txt
tail -5 gpurun_out/r3l_tests.txt
{
for B in 1024 256 512; do
  for v in 1 0 1 0; do
    SCLDM_BGEMM8=$v SCLDM_DGRAD_WT=$v SCLDM_DHID16=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/new=$v /"
  done
done
} > gpurun_out/r3l_ditl_ab.txt 2>&1
cat gpurun_out/r3l_ditl_ab.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3l_ditl1024" -o r3l_ditl1024 --output-format csv -- python3 "$GRAFT_REPO_ROOT/tests/perf/bgemm_check.py" run 1024 > "$GRAFT_REPO_ROOT/gpurun_out/r3l_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/r3l_ditl1024 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200 > gpurun_out/r3l_train_ditl_b1024_kernel_stats.txt
cat gpurun_out/r3l_train_ditl_b1024_kernel_stats.txt | cut -c1-150
