#!/bin/bash
# Round evidence on the GPU box: bench lines, rocprofv3 kernel statistics, PMC passes, GPU test log -> gpurun_out/<tag>_*
# (copy what is to be judged into profiles/).   usage: tools/collect_evidence.sh <tag>
tag=${1:-r2}
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out
o=gpurun_out/$tag
timeout 900 python bench.py --steps 3 > ${o}_bench_default.json 2> ${o}_bench_default.err; tail -c 600 ${o}_bench_default.json; echo
timeout 300 python bench.py --precision bf16x3 --steps 2 --no-extra --no-cpu-baseline > ${o}_bench_bf16x3.json 2>/dev/null
timeout 300 python bench.py --precision fp32 --steps 1 --no-extra --no-cpu-baseline > ${o}_bench_fp32.json 2>/dev/null
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 1 --no-extra --no-cpu-baseline > ${o}_bench_rccl_world1.json 2>/dev/null
timeout 60 python bench.py --gpus 2 --steps 1 > ${o}_bench_gpus2_on_1gpu_box.txt 2>&1; echo "gpus2 on a 1-GPU box: exit $?" >> ${o}_bench_gpus2_on_1gpu_box.txt
ROCPROF_ROWS=12 bash tools/rocprof_stats.sh ${tag}_stats_bench bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > ${o}_bench_kernel_stats.txt 2>&1
ROCPROF_ROWS=12 bash tools/rocprof_stats.sh ${tag}_stats_bench_x3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --precision bf16x3 > ${o}_bench_x3_kernel_stats.txt 2>&1
for g in "17002 6147" "27997 10186"; do set -- $g
  ROCPROF_ROWS=9 bash tools/rocprof_stats.sh ${tag}_stats_vae_$1 tests/perf/vae_profile.py $1 $2 1024 fp32 > ${o}_vae_G$1_fp32_kernel_stats.txt 2>&1
  ROCPROF_ROWS=9 bash tools/rocprof_stats.sh ${tag}_stats_vae_$1b tests/perf/vae_profile.py $1 $2 1024 bf16 > ${o}_vae_G$1_bf16_kernel_stats.txt 2>&1
done
ROCPROF_ROWS=14 bash tools/rocprof_stats.sh ${tag}_stats_ditl256 tests/perf/train_ditl_profile.py 256 > ${o}_train_ditl_b256_kernel_stats.txt 2>&1
ROCPROF_ROWS=14 bash tools/rocprof_stats.sh ${tag}_stats_ditl1024 tests/perf/train_ditl_profile.py 1024 > ${o}_train_ditl_b1024_kernel_stats.txt 2>&1
timeout 300 python tests/perf/bgemm_check.py 256 > ${o}_train_ditl_ab.txt 2>&1
timeout 300 python tests/perf/vae_bench.py > ${o}_vae_bench.txt 2>&1
K=dit_forward
bash tools/rocprof_pmc.sh ${tag}_pmc1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_sq.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmc2 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_sq2.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmc3 "FETCH_SIZE" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_fetch.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmc4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $K tests/perf/dit_profile.py bf16 6 > ${o}_pmc_write.txt 2>&1
bash tools/rocprof_pmc.sh ${tag}_pmc5 "FETCH_SIZE" dec_gene tests/perf/vae_profile.py 17002 6147 1024 fp32 > ${o}_pmc_vae_fetch.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[parity\]|passed|failed|^E " > ${o}_gpu_tests.txt; tail -2 ${o}_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > ${o}_smoke.txt 2>&1; tail -3 ${o}_smoke.txt
