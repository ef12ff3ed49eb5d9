#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for B in 256 128 384; do
  for v in 96 0 96 0; do
    SCLDM_WT_MIN_TILES=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/wt_min=$v /"
  done
done
} > gpurun_out/r3aa_wt_min.txt 2>&1
cat gpurun_out/r3aa_wt_min.txt
