#!/bin/bash
# Re-take the headline evidence after a change to the forward kernel's sources: bench lines, kernel statistics, the four PMC passes and
# their summary (profiles/pmc_dit_forward_kernel.json carries the source hash bench.py checks before it reports `roofline.traffic`).
tag=${1:-r6}
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out
o=gpurun_out/$tag
K=dit_forward
for prec in bf16 fp16; do
bash tools/rocprof_pmc.sh ${tag}_pmc1_$prec "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" $K tests/perf/dit_profile.py $prec 6 > ${o}_pmc_sq_$prec.txt 2>&1
done
cp ${o}_pmc_sq_bf16.txt ${o}_pmc_sq.txt
bash tools/rocprof_pmc.sh ${tag}_pmc2 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_sq2.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmc3 "FETCH_SIZE" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_fetch.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmc4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_write.txt 2>&1
python3 tools/pmc_summary.py $tag 8 > /dev/null 2>&1
cp profiles/pmc_dit_forward_kernel.json gpurun_out/${tag}_pmc_dit_forward_kernel.json
ROCPROF_ROWS=12 bash tools/rocprof_stats.sh ${tag}_stats_bench bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > ${o}_bench_kernel_stats.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 5 > ${o}_bench_default.json 2> ${o}_bench_default.err; cp profiles/bench_last.json ${o}_bench_default_details.json
timeout 300 python bench.py --precision fp16 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > ${o}_bench_fp16.json 2>/dev/null
tail -c 400 ${o}_bench_default.json
