#!/bin/bash
# round 3, run N: bf16 branch outputs, merged MLP data gradient
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3n_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3n_tests.txt
tail -5 gpurun_out/r3n_tests.txt
{
for B in 1024; do
  for v in "" "SCLDM_Y16=0" "SCLDM_MLP_MERGE=0" "" "SCLDM_Y16=0" "SCLDM_MLP_MERGE=0"; do
    env $v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/[$v] /"
  done
done
for B in 512 256; do env timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1; done
} > gpurun_out/r3n_ditl_ab.txt 2>&1
cat gpurun_out/r3n_ditl_ab.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3n_ditl1024" -o r3n_ditl1024 --output-format csv -- python3 "$GRAFT_REPO_ROOT/tests/perf/bgemm_check.py" run 1024 > "$GRAFT_REPO_ROOT/gpurun_out/r3n_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/r3n_ditl1024 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200 > gpurun_out/r3n_train_ditl_b1024_kernel_stats.txt
cat gpurun_out/r3n_train_ditl_b1024_kernel_stats.txt | cut -c1-150
