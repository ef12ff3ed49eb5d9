#!/bin/bash
# round 3, run T: SwiGLU backward in the epilogue of c_proj's data gradient
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3t_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3t_tests.txt
tail -4 gpurun_out/r3t_tests.txt
{
for B in 1024; do
  for v in 1 0 1 0; do
    SCLDM_FUSE_SWIGLU_BWD=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/fuse_swiglu_bwd=$v /"
  done
done
} > gpurun_out/r3t_ditl_ab.txt 2>&1
cat gpurun_out/r3t_ditl_ab.txt
