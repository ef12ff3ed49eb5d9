#!/bin/bash
# round 3, run S: the batched weight gradient on a side stream beside the data-gradient chain
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3s_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3s_tests.txt
tail -4 gpurun_out/r3s_tests.txt
{
for B in 1024 512; do
  for v in 1 0 1 0; do
    SCLDM_BATCH_SIDE=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/batch_side=$v /"
  done
done
env timeout 300 python tests/perf/bgemm_check.py run 256 2>&1 | tail -1
} > gpurun_out/r3s_ditl_ab.txt 2>&1
cat gpurun_out/r3s_ditl_ab.txt
