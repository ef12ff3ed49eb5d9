#!/bin/bash
# Round evidence on the GPU box: bench lines, rocprofv3 kernel statistics, PMC passes, GPU test log -> gpurun_out/<tag>_*
# (copy what is to be judged into profiles/).   usage: tools/collect_evidence.sh <tag>
tag=${1:-r3}
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out
o=gpurun_out/$tag
timeout 1500 python bench.py --steps 20 --warmup 5 > ${o}_bench_default.json 2> ${o}_bench_default.err; tail -c 600 ${o}_bench_default.json; echo; cp profiles/bench_last.json ${o}_bench_default_details.json 2>/dev/null
timeout 300 python bench.py --precision fp16 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > ${o}_bench_fp16.json 2>/dev/null
timeout 300 python bench.py --precision bf16x3 --steps 10 --warmup 2 --no-extra --no-cpu-baseline > ${o}_bench_bf16x3.json 2>/dev/null
timeout 300 python bench.py --precision fp32 --steps 1 --no-extra --no-cpu-baseline > ${o}_bench_fp32.json 2>/dev/null
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 1 --no-extra --no-cpu-baseline > ${o}_bench_rccl_world1.json 2>/dev/null
timeout 300 python bench.py --workload replogle_train_ditl_b1024 --steps 4 --warmup 2 > ${o}_bench_replogle_train_ditl_b1024.json 2>/dev/null
ROCPROF_ROWS=12 bash tools/rocprof_stats.sh ${tag}_stats_bench bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > ${o}_bench_kernel_stats.txt 2>&1
ROCPROF_ROWS=12 bash tools/rocprof_stats.sh ${tag}_stats_bench_fp16 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --precision fp16 > ${o}_bench_fp16_kernel_stats.txt 2>&1
ROCPROF_ROWS=26 bash tools/rocprof_stats.sh ${tag}_stats_vae_train tests/perf/vae_train_bench.py 32 > ${o}_vae_train_b32_kernel_stats.txt 2>&1
ROCPROF_ROWS=20 bash tools/rocprof_stats.sh ${tag}_stats_train tests/perf/train_cliff.py 1024 > ${o}_train_b1024_kernel_stats.txt 2>&1
ROCPROF_ROWS=16 bash tools/rocprof_stats.sh ${tag}_stats_ditl1024 tests/perf/train_ditl_profile.py 1024 > ${o}_train_ditl_b1024_kernel_stats.txt 2>&1
timeout 300 python tests/perf/train_scale.py > ${o}_train_scale.txt 2>&1
timeout 300 python tests/perf/train_scale.py nogc > ${o}_train_scale_nogc.txt 2>&1
timeout 300 python tests/perf/vae_train_bench.py 32 128 512 > ${o}_vae_train_bench.txt 2>&1
timeout 300 python tests/perf/vae_train_host.py 32 >> ${o}_vae_train_bench.txt 2>&1
ROCPROF_ROWS=12 bash tools/rocprof_stats.sh ${tag}_stats_parse1m bench.py --workload parse1m_b1024_euler100 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > ${o}_parse1m_b1024_kernel_stats.txt 2>&1
ROCPROF_ROWS=12 bash tools/rocprof_stats.sh ${tag}_stats_train_fp16 bench.py --workload replogle_train_b1024 --precision fp16 --steps 20 --warmup 5 > ${o}_train_fp16_b1024_kernel_stats.txt 2>&1
timeout 120 python tests/perf/power_probe.py 4096 30 > ${o}_power_probe.txt 2>&1
K=dit_forward
for prec in bf16 fp16; do
bash tools/rocprof_pmc.sh ${tag}_pmc1_$prec "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" $K tests/perf/dit_profile.py $prec 6 > ${o}_pmc_sq_$prec.txt 2>&1
done
cp ${o}_pmc_sq_bf16.txt ${o}_pmc_sq.txt
bash tools/rocprof_pmc.sh ${tag}_pmc2 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_sq2.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmc3 "FETCH_SIZE" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_fetch.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmc4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_write.txt 2>&1
# round 5: the fused training step's kernels (statistics, one-step timeline, four PMC passes over the backward / weight-gradient / recording-forward kernels)
ROCPROF_ROWS=24 bash tools/rocprof_stats.sh ${tag}_stats_train_bench tools/train_bench.py 1024 25 bf16 > ${o}_train_bench_b1024_kernel_stats.txt 2>&1
python3 tools/kernel_timeline.py $(find gpurun_out/${tag}_stats_train_bench -name "*kernel_trace.csv" | head -1) 3 > ${o}_train_b1024_timeline.txt 2>&1
KT="dit_backward|wgrad_bf16|dit_forward"
bash tools/rocprof_pmc.sh ${tag}_pmct1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "$KT" tools/train_bench.py 1024 8 bf16 > ${o}_pmc_train_sq1.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmct2 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" "$KT" tools/train_bench.py 1024 8 bf16 > ${o}_pmc_train_sq2.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmct3 "FETCH_SIZE" "$KT" tools/train_bench.py 1024 8 bf16 > ${o}_pmc_train_fetch.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmct4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "$KT" tools/train_bench.py 1024 8 bf16 > ${o}_pmc_train_write.txt 2>&1
for b in 1024 512 256; do python tools/train_bench.py $b 30 bf16 2>&1 | grep ms/step; done > ${o}_train_bench_sizes.txt
python tools/train_bench.py 1024 30 fp16 2>&1 | grep ms/step >> ${o}_train_bench_sizes.txt
TRAIN_BENCH_TORCH_ADAMW=1 python tools/train_bench.py 1024 30 bf16 2>&1 | grep ms/step | sed "s/^/torch AdamW: /" >> ${o}_train_bench_sizes.txt
python tools/train_graph_bench.py 256 bf16 2>&1 | grep cells >> ${o}_train_bench_sizes.txt
cp profiles/bench_last.json ${o}_bench_last.json 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[parity\]|^\[dopri5|passed|failed|^E  |^FAILED" | cut -c1-400 > ${o}_gpu_tests.txt; tail -2 ${o}_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > ${o}_smoke.txt 2>&1; tail -3 ${o}_smoke.txt
# round 6: MCAB kernels at the bench shapes (kernel statistics + four PMC passes per precision), the one-call training step (timelines at
# 1 024 and 256 cells, step times per mode and size), the hand-written GEMM against the vendor library per DiT-L shape, DiT-L step per optimizer
bash tools/r6_mcab_evidence.sh ${tag} "fp32 fp16 bf16" > ${o}_mcab_evidence.log 2>&1
for b in 1024 256; do
  ROCPROF_ROWS=30 bash tools/rocprof_stats.sh ${tag}_stats_fused_b$b tests/perf/train_fused_profile.py $b 20 bf16 1 > ${o}_train_fused_b${b}_kernel_stats.txt 2>&1
  python3 tools/kernel_timeline.py $(find gpurun_out/${tag}_stats_fused_b$b -name "*kernel_trace.csv" | head -1) 3 > ${o}_train_fused_b${b}_graph_timeline.txt 2>&1
done
{ for a in "1024 40 bf16 1" "1024 40 bf16 0" "1024 40 fp16 0" "512 40 bf16 0" "256 40 bf16 1" "256 40 bf16 0" "128 40 bf16 0" "64 40 bf16 0" "1024 40 bf16 1 1" "1024 40 bf16 0 1"; do python tests/perf/train_fused_profile.py $a 2>&1 | tail -1; done; } > ${o}_train_fused_sizes.txt
bash tools/r6_gemm_vs_vendor.sh ${tag} > /dev/null 2>&1
python tests/perf/ditl_optimizer_ab.py 2>&1 | grep ms/step > ${o}_ditl_optimizer_ab.txt
