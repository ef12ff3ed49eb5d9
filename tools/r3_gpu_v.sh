#!/bin/bash
# round 3, run V: PMC of the GEMM kernels of the DiT-L step (HBM fetch, L2 hit rate, MFMA busy)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
K="bgemm8"
bash tools/rocprof_pmc.sh r3v_pmc_fetch "FETCH_SIZE" $K tests/perf/train_ditl_profile.py 1024 > gpurun_out/r3v_pmc_fetch.txt 2>&1
bash tools/rocprof_pmc.sh r3v_pmc_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $K tests/perf/train_ditl_profile.py 1024 > gpurun_out/r3v_pmc_tcc.txt 2>&1
bash tools/rocprof_pmc.sh r3v_pmc_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" $K tests/perf/train_ditl_profile.py 1024 > gpurun_out/r3v_pmc_sq.txt 2>&1
cat gpurun_out/r3v_pmc_fetch.txt gpurun_out/r3v_pmc_tcc.txt gpurun_out/r3v_pmc_sq.txt
