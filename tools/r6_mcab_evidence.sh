#!/bin/bash
# Round 6: rocprofv3 evidence for the MCAB kernels at the bench shapes (decode 8 192 rows x 17 002 genes, encode 4 096 cells x 6 147
# tokens): kernel statistics + four PMC passes per precision -> gpurun_out/<tag>_mcab_*.txt (copy into profiles/).
#   usage: tools/r6_mcab_evidence.sh <tag> [precisions="fp32 fp16"]
tag=${1:-r6}
precs=${2:-"fp32 fp16"}
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out
o=gpurun_out/${tag}_mcab
K="enc_pool|enc_cell|dec_cell|dec_gene|dec_finalize"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE"
for prec in $precs; do
  for mode in decode encode; do
    ROCPROF_ROWS=8 bash tools/rocprof_stats.sh ${tag}_mcab_stats_${mode}_$prec tests/perf/mcab_profile.py $mode $prec > ${o}_${mode}_${prec}_kernel_stats.txt 2>&1
    bash tools/rocprof_pmc.sh ${tag}_mcab_p1_${mode}_$prec "$SQ1" "$K" tests/perf/mcab_profile.py $mode $prec > ${o}_${mode}_${prec}_pmc_sq1.txt 2>&1
    bash tools/rocprof_pmc.sh ${tag}_mcab_p2_${mode}_$prec "$SQ2" "$K" tests/perf/mcab_profile.py $mode $prec > ${o}_${mode}_${prec}_pmc_sq2.txt 2>&1
    bash tools/rocprof_pmc.sh ${tag}_mcab_p3_${mode}_$prec "FETCH_SIZE" "$K" tests/perf/mcab_profile.py $mode $prec > ${o}_${mode}_${prec}_pmc_fetch.txt 2>&1
    bash tools/rocprof_pmc.sh ${tag}_mcab_p4_${mode}_$prec "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "$K" tests/perf/mcab_profile.py $mode $prec > ${o}_${mode}_${prec}_pmc_write.txt 2>&1
  done
done
ROCPROF_ROWS=8 bash tools/rocprof_stats.sh ${tag}_mcab_stats_decode_sample_fp16 tests/perf/mcab_profile.py decode_sample fp16 > ${o}_decode_sample_fp16_kernel_stats.txt 2>&1
ls -la ${o}_*
