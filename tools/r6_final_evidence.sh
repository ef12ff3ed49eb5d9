#!/bin/bash
# Final-tree verification of round 6 on the GPU box: the whole -m gpu suite, smoke(), the default bench line, and the one-call training
# step's kernel statistics / one-step timelines / step times per size (re-taken because the step changed after the first evidence pass).
#   usage (through gpurun): bash tools/r6_final_evidence.sh      -> gpurun_out/r6_final_*  (copy into profiles/)
set -uo pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
o=gpurun_out/r6_final
timeout 2400 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[parity\]|^\[dopri5|passed|failed|^E  |^FAILED" | cut -c1-400 > ${o}_gpu_tests.txt; tail -2 ${o}_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > ${o}_smoke.txt 2>&1; tail -3 ${o}_smoke.txt
python bench.py > ${o}_bench.json 2> ${o}_bench.err; cp profiles/bench_last.json ${o}_bench_details.json 2>/dev/null; tail -c 600 ${o}_bench.json
# kernel statistics of the headline command itself (the bench line's roofline must agree with the dominant kernel's average here)
ROCPROF_ROWS=12 bash tools/rocprof_stats.sh r6_final_stats_bench bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > ${o}_bench_kernel_stats.txt 2>&1
ROCPROF_ROWS=12 bash tools/rocprof_stats.sh r6_final_stats_bench_fp16 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --precision fp16 > ${o}_bench_fp16_kernel_stats.txt 2>&1
for b in 1024 256; do
  ROCPROF_ROWS=30 bash tools/rocprof_stats.sh r6_final_stats_fused_b$b tests/perf/train_fused_profile.py $b 20 bf16 1 > ${o}_train_fused_b${b}_kernel_stats.txt 2>&1
  python3 tools/kernel_timeline.py $(find gpurun_out/r6_final_stats_fused_b$b -name "*kernel_trace.csv" | head -1) 3 > ${o}_train_fused_b${b}_graph_timeline.txt 2>&1
done
{ for a in "1024 40 bf16 1" "1024 40 bf16 0" "1024 40 fp16 0" "512 40 bf16 0" "256 40 bf16 1" "256 40 bf16 0" "128 40 bf16 0" "64 40 bf16 0" "1024 40 bf16 1 1" "1024 40 bf16 0 1"; do python tests/perf/train_fused_profile.py $a 2>&1 | tail -1; done; } > ${o}_train_fused_sizes.txt
cat ${o}_train_fused_sizes.txt
