#!/bin/bash
# round 3, run Q: residual step fused into the next LayerNorm (forward), gate backward fused into the LayerNorm backward
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3q_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3q_tests.txt
tail -4 gpurun_out/r3q_tests.txt
{
for B in 1024; do
  for v in "" "SCLDM_FUSE_RES=0" "SCLDM_FUSE_GATE=0" "" "SCLDM_FUSE_RES=0" "SCLDM_FUSE_GATE=0"; do
    env $v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/[$v] /"
  done
done
for B in 512 256; do env timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1; done
} > gpurun_out/r3q_ditl_ab.txt 2>&1
cat gpurun_out/r3q_ditl_ab.txt
