#!/bin/bash
# round 3, run M: LDS-DMA weight-gradient batch kernel (both operands contiguous along m)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tests/perf/gemm_probe mc 3072 1024 16384 1024 2732 16384 1000 520 4000 > gpurun_out/r3m_probe_mc4.txt 2>&1
grep -v "probe:" gpurun_out/r3m_probe_mc4.txt
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3m_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r3m_tests.txt
tail -5 gpurun_out/r3m_tests.txt
{
for B in 1024 512; do
  for v in 1 0 1 0; do
    SCLDM_BGEMM8_WGRAD=$v timeout 300 python tests/perf/bgemm_check.py run $B 2>&1 | tail -1 | sed "s/^/wgrad_dma=$v /"
  done
done
} > gpurun_out/r3m_ditl_ab.txt 2>&1
cat gpurun_out/r3m_ditl_ab.txt
