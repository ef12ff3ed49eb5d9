#!/bin/bash
# Round 6: the hand-written LDS-DMA GEMM against the vendor library on the SAME box, per DiT-L shape at 1 024 cells
# (16 384 tokens): torch.matmul (hipBLASLt) from tests/perf/gemm_yardstick.py beside tests/perf/gemm_probe (k-contiguous products) and
# `gemm_probe mc` (weight gradients: both operands contiguous along m).   -> gpurun_out/<tag>_gemm_vs_vendor.txt
tag=${1:-r6}
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
o=gpurun_out/${tag}_gemm_vs_vendor.txt
{
  echo "== vendor (torch.matmul -> hipBLASLt), bf16 in / bf16 out"
  python tests/perf/gemm_yardstick.py 2>/dev/null
  echo "== ours, k-contiguous operands (forward products and data gradients): bgemm8_kernel variants, bf16 out = 'through LDS' rows"
  tests/perf/gemm_probe 16384 3072 1024 16384 1024 1024 16384 2736 1024 16384 1024 2736 16384 1024 3072 2>/dev/null | grep -E "^M=|bgemm8_kernel<0>|bf16 out: bgemm8, through LDS|bf16 out: bgemm256"
  echo "== ours, m-contiguous operands (weight gradients)"
  tests/perf/gemm_probe mc 3072 1024 16384 2736 1024 16384 2>/dev/null | grep -E "^M=|vector epilogue|no row sums"
} > $o 2>&1
cat $o
