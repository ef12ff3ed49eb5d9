// Backward of the conditioning chain behind d SiLU(c) in TWO kernels (round 6), fused training route (n_embed 256):
//     c = t_emb + sum_classes table[label],   t_emb = SiLU(freq t_w0^T + t_b0) t_w2^T + t_b2        (src/scldm/layers.py:351-364,
//                                                                                                   src/scldm/nnets.py:283-288)
//   cond_bwd_rows_kernel   (one workgroup per 8 batch rows)   d c = d SiLU(c) SiLU'(c);  d sth = d c t_w2;  d th = d sth SiLU'(th)
//   cond_bwd_wgrad_kernel  (one workgroup per 32 x 32 output tile of either matrix)
//                                                              d t_w2 = d c^T sth,  d t_b2 = colsum(d c);  d t_w0 = d th^T freq,  d t_b0 = colsum(d th)
// Round 5 ran this as a chain of ~9 launches (silu_bwd, split-K GEMM + reduce + column sums twice, a data-gradient GEMM, silu_bwd):
// 80-100 us of 5-14 us kernels and launch gaps at the very end of the step's critical path (profiles/r6_train_fused_b*_timeline.txt),
// a quarter of the tail - 5 % of a 256-cell step.  The three products are 134 MFLOP each at 1 024 cells: exact fp32 on the VALU (the
// generic route rounds their operands to 16 bits), every sum in a fixed order.  The class-table gradients (embed_bwd_*) read the d c
// this writes.
// STATUS: measured +-0 to slower inside the step (train_api.hip, SCLDM_TRAIN_COND_BWD) - opt-in; kept as the exact-fp32 form of the chain.
#pragma once
#include "common.hpp"

namespace scldm {
namespace train {

constexpr int kCbRows = 8;      // batch rows per workgroup of cond_bwd_rows_kernel
constexpr int kCbD = 256;       // n_embed = timestep-MLP width = frequency-embedding size of the fused route

__device__ __forceinline__ float cb_silu_grad(float dy, float v) {   // silu_bwd_kernel's arithmetic
  const float s = 1.0f / (1.0f + __expf(-v));
  return dy * s * (1.0f + v * (1.0f - s));
}

__global__ __launch_bounds__(256) void cond_bwd_rows_kernel(const float* __restrict__ dsc, const float* __restrict__ c, const float* __restrict__ th,
                                                            const float* __restrict__ t_w2, int n, float* __restrict__ dc, float* __restrict__ dth) {
  __shared__ __attribute__((aligned(16))) float DC[kCbRows][kCbD];
  const int i = threadIdx.x, r0 = blockIdx.x * kCbRows;
#pragma unroll
  for (int r = 0; r < kCbRows; ++r) {
    const int row = r0 + r;
    float d = 0.f;
    if (row < n) {
      d = cb_silu_grad(dsc[(size_t)row * kCbD + i], c[(size_t)row * kCbD + i]);
      dc[(size_t)row * kCbD + i] = d;
    }
    DC[r][i] = d;
  }
  __syncthreads();
  // d sth[r][i] = sum_o d c[r][o] t_w2[o][i]      (t_w2 is (out, in): row o is contiguous over i - coalesced, L2-resident).
  // 64 weight rows are requested together (the loop is a chain of L2 round trips otherwise: 32 us at 8 loads in flight, r6f)
  float acc[kCbRows];
#pragma unroll
  for (int r = 0; r < kCbRows; ++r) acc[r] = 0.f;
  for (int o0 = 0; o0 < kCbD; o0 += 64) {
    float w[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) w[j] = t_w2[(size_t)(o0 + j) * kCbD + i];
#pragma unroll
    for (int j4 = 0; j4 < 64; j4 += 4)
#pragma unroll
      for (int r = 0; r < kCbRows; ++r) {
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(&DC[r][o0 + j4]);   // (one address per wave: an LDS broadcast)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r] = fmaf(d4[j], w[j4 + j], acc[r]);
      }
  }
#pragma unroll
  for (int r = 0; r < kCbRows; ++r) {
    const int row = r0 + r;
    if (row < n) dth[(size_t)row * kCbD + i] = cb_silu_grad(acc[r], th[(size_t)row * kCbD + i]);
  }
}

// job 0: dW = d t_w2 (out, in) from (dy = d c, x = sth);  job 1: dW = d t_w0 from (dy = d th, x = freq).  Workgroup = output tile
// [32 outputs o] x [32 inputs i] over ALL n rows (no split: no partial buffer, one fixed summation order); thread (tg = tid >> 5, ti =
// tid & 31) owns outputs 4 tg .. 4 tg + 3 at input ti.  The tiles with i-tile 0 also form the bias gradient of their 32 outputs.
struct CondWgradArgs {
  const float* dy[2];
  const float* x[2];
  float* dW[2];
  float* db[2];
  int n;
};
__global__ __launch_bounds__(256) void cond_bwd_wgrad_kernel(const CondWgradArgs a) {
  constexpr int kChunk = 128;   // rows per stage; the NEXT stage's rows are requested before this stage's arithmetic (register prefetch)
  __shared__ __attribute__((aligned(16))) float A[kChunk][32], B[kChunk][33];
  const int job = blockIdx.z, o0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
  const int tid = threadIdx.x, tg = tid >> 5, ti = tid & 31;
  const float* __restrict__ dy = a.dy[job];
  const float* __restrict__ x = a.x[job];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  // staging map: thread -> rows (tid >> 3) + 32 h, h < 4, four columns (tid & 7) * 4 of each operand: 16-byte loads
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  f32x4 pa[4], pb[4];
  auto fetch = [&](int rbase) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int row = rbase + sr + 32 * h;
      pa[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      pb[h] = pa[h];
      if (row < a.n) {
        pa[h] = *reinterpret_cast<const f32x4*>(dy + (size_t)row * kCbD + o0 + sc);
        pb[h] = *reinterpret_cast<const f32x4*>(x + (size_t)row * kCbD + i0 + sc);
      }
    }
  };
  fetch(0);
  for (int rbase = 0; rbase < a.n; rbase += kChunk) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      *reinterpret_cast<f32x4*>(&A[sr + 32 * h][sc]) = pa[h];
#pragma unroll
      for (int j = 0; j < 4; ++j) B[sr + 32 * h][sc + j] = pb[h][j];
    }
    __syncthreads();
    if (rbase + kChunk < a.n) fetch(rbase + kChunk);
#pragma unroll 8
    for (int r = 0; r < kChunk; ++r) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&A[r][4 * tg]);
      const float b = B[r][ti];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(a4[j], b, acc[j]);
    }
    if (blockIdx.x == 0 && tid < 32) {
#pragma unroll 8
      for (int r = 0; r < kChunk; ++r) bsum += A[r][tid];
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) a.dW[job][(size_t)(o0 + 4 * tg + j) * kCbD + i0 + ti] = acc[j];
  if (blockIdx.x == 0 && tid < 32 && a.db[job]) a.db[job][o0 + tid] = bsum;
}

}  // namespace train
}  // namespace scldm
