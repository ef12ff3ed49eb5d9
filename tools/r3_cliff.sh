cd $GRAFT_REPO_ROOT
for B in 256 384 512 640 768 1024; do timeout 120 python tests/perf/train_cliff.py $B; done > gpurun_out/r3_cliff.txt 2>&1
for B in 512 1024; do ROCPROF_ROWS=30 bash tools/rocprof_stats.sh r3_cliff_$B tests/perf/train_cliff.py $B 20 > gpurun_out/r3_cliff_stats_$B.txt 2>&1; done
cat gpurun_out/r3_cliff.txt
