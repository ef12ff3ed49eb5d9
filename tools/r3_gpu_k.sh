#!/bin/bash
# round 3, run I: LDS-DMA GEMM (bgemm8_kernel) - parity, race screen, A/B on the DiT-L step, vendor yardstick
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -s -k "lds_dma_gemm or wider_shapes or batched_weight" > gpurun_out/r3k_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3k_tests.txt
tail -5 gpurun_out/r3k_tests.txt
{
for B in 1024 256; do
  for v in 1 0 1 0; do
    SCLDM_BGEMM8=$v SCLDM_DGRAD_WT=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/BGEMM8=$v /"
  done
done
} > gpurun_out/r3k_ditl_ab.txt 2>&1
cat gpurun_out/r3k_ditl_ab.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3k_ditl1024" -o r3k_ditl1024 --output-format csv -- python3 "$GRAFT_REPO_ROOT/tests/perf/bgemm_check.py" run 1024 > "$GRAFT_REPO_ROOT/gpurun_out/r3k_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/r3k_ditl1024 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200 > gpurun_out/r3k_train_ditl_b1024_kernel_stats.txt
cat gpurun_out/r3k_train_ditl_b1024_kernel_stats.txt | cut -c1-160
