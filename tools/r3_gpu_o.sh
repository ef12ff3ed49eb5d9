#!/bin/bash
# round 3, run O: matrix-core attention
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q -s -k "matrix_core or wider_shapes or 24_layer or deep" > gpurun_out/r3o_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3o_tests.txt
grep -E "parity|passed|failed|Error" gpurun_out/r3o_tests.txt | tail -12
{
for B in 1024 256; do
  for v in 1 0 1 0; do
    SCLDM_ATTN_MFMA=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/attn_mfma=$v /"
  done
done
} > gpurun_out/r3o_ditl_ab.txt 2>&1
cat gpurun_out/r3o_ditl_ab.txt
