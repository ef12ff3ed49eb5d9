#!/bin/bash
# round 3, run P: bf16 dh / dao; full training tests; profile of the DiT-L step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3p_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3p_tests.txt
tail -4 gpurun_out/r3p_tests.txt
{
for B in 1024 256; do
  for v in 1 0 1 0; do
    SCLDM_GRAD16=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/grad16=$v /"
  done
done
env timeout 300 python tests/perf/bgemm_check.py run 512 2>&1 | tail -1
} > gpurun_out/r3p_ditl_ab.txt 2>&1
cat gpurun_out/r3p_ditl_ab.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3p_ditl1024" -o r3p_ditl1024 --output-format csv -- python3 "$GRAFT_REPO_ROOT/tests/perf/bgemm_check.py" run 1024 > "$GRAFT_REPO_ROOT/gpurun_out/r3p_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/r3p_ditl1024 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" | cut -c1-200 > gpurun_out/r3p_train_ditl_b1024_kernel_stats.txt
cat gpurun_out/r3p_train_ditl_b1024_kernel_stats.txt | cut -c1-150
