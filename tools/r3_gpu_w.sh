#!/bin/bash
# round 3, run W: from how many 256-tiles on is the LDS-DMA kernel the better choice (smaller batches)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for B in 512 256 384 768; do
  for v in 224 128 96 64 224 128; do
    SCLDM_MIN_TILES256=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/min_tiles=$v /"
  done
done
} > gpurun_out/r3w_min_tiles.txt 2>&1
cat gpurun_out/r3w_min_tiles.txt
