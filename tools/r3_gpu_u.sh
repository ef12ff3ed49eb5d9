#!/bin/bash
# round 3, run U: bf16 results through LDS (full-line stores)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tests/perf/gemm_probe 16384 2732 1024 4000 1004 520 > gpurun_out/r3u_probe2.txt 2>&1
grep -E "^M=|bf16 out" gpurun_out/r3u_probe2.txt
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3u_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3u_tests.txt
tail -4 gpurun_out/r3u_tests.txt
{
for B in 1024 256; do
  for v in 1 0 1 0; do
    SCLDM_EPI_LDS=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/epi_lds=$v /"
  done
done
env timeout 300 python tests/perf/bgemm_check.py run 512 2>&1 | tail -1
} > gpurun_out/r3u_ditl_ab.txt 2>&1
cat gpurun_out/r3u_ditl_ab.txt
