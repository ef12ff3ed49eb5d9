#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3ac_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3ac_tests.txt
tail -3 gpurun_out/r3ac_tests.txt
{
for B in 1024 256; do
  for v in 1 0 1 0; do
    SCLDM_CAST_SIDE=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/cast_side=$v /"
  done
done
} > gpurun_out/r3ac_cast_side.txt 2>&1
cat gpurun_out/r3ac_cast_side.txt
