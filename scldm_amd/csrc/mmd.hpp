// Generation-evaluation MMD: the kernel matrices of src/scldm/evaluations.py:10-69
// (RBF, Bray-Curtis, Tanimoto, Ruzicka) and MMDLoss (:72-82), fused so that the (Bx, By, D) broadcast tensors the
// reference materialises (its memory blow-up on the GPU) never exist: a workgroup owns a 64 x 64 tile of pairs, streams
// D through LDS once, keeps two fp32 accumulators per pair in registers, finishes k(x_i, y_j) in place and reduces the
// tile to one partial sum (deterministic: partials are summed in index order afterwards).  The matrix itself is
// written only on request.
#pragma once
#include <hip/hip_runtime.h>

namespace scldm {

enum MmdKind { kMmdRbf = 0, kMmdBrayCurtis = 1, kMmdTanimoto = 2, kMmdRuzicka = 3,
               kMmdSqDist = 4, kMmdDist = 5 };   // |x - y|^2 and |x - y|: the cost matrices of evaluations.py:103-105 (torch.cdist)

constexpr int kMmdTile = 64;   // pairs per side of a workgroup tile
constexpr int kMmdDC = 32;     // feature chunk staged per trip

template <int KIND>
__global__ __launch_bounds__(256) void mmd_tile_kernel(const float* __restrict__ x, int nx, const float* __restrict__ y, int ny, int D,
                                                       float scale, float* __restrict__ partial, float* __restrict__ kmat) {
  __shared__ __attribute__((aligned(16))) float xs[kMmdDC][kMmdTile + 4];
  __shared__ __attribute__((aligned(16))) float ys[kMmdDC][kMmdTile + 4];
  __shared__ float red[4];
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;   // thread owns pairs (4 ti + a, 4 tj + b)
  const int i0 = blockIdx.y * kMmdTile, j0 = blockIdx.x * kMmdTile;
  float p[4][4], q[4][4];    // per-kind pair accumulators
  float rx[4] = {0, 0, 0, 0}, ry[4] = {0, 0, 0, 0};   // per-row statistics (squared norms or sums)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) p[a][b] = q[a][b] = 0.f;

  for (int d0 = 0; d0 < D; d0 += kMmdDC) {
    // stage 64 rows x 32 features of each operand, transposed to [d][row] (coalesced 128-byte row segments in, conflict-free out)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = (tid >> 5) + 8 * r, d = tid & 31;
      xs[d][row] = (i0 + row < nx && d0 + d < D) ? x[(long)(i0 + row) * D + d0 + d] : 0.f;
      ys[d][row] = (j0 + row < ny && d0 + d < D) ? y[(long)(j0 + row) * D + d0 + d] : 0.f;
    }
    __syncthreads();
    // blocked summation: this chunk's 32 terms are summed on their own and then added to the running totals, which keeps the
    // rounding error of the 17k-term sums near that of torch's pairwise reduction (the RBF exponent is ill-conditioned)
    float pc[4][4], qc[4][4], rxc[4] = {0, 0, 0, 0}, ryc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) pc[a][b] = qc[a][b] = 0.f;
#pragma unroll 8
    for (int d = 0; d < kMmdDC; ++d) {
      const float4 xv = *reinterpret_cast<const float4*>(&xs[d][4 * ti]);
      const float4 yv = *reinterpret_cast<const float4*>(&ys[d][4 * tj]);
      const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, yb[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (KIND == kMmdRbf || KIND == kMmdSqDist || KIND == kMmdDist) rxc[a] += xa[a] * xa[a];
        if (KIND == kMmdTanimoto) rxc[a] += xa[a];
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (KIND == kMmdRbf || KIND == kMmdSqDist || KIND == kMmdDist) ryc[b] += yb[b] * yb[b];
        if (KIND == kMmdTanimoto) ryc[b] += yb[b];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (KIND == kMmdRbf || KIND == kMmdTanimoto || KIND == kMmdSqDist || KIND == kMmdDist) pc[a][b] += xa[a] * yb[b];
          if (KIND == kMmdBrayCurtis) {
            pc[a][b] += fabsf(xa[a] - yb[b]);
            qc[a][b] += fabsf(xa[a] + yb[b]);
          }
          if (KIND == kMmdRuzicka) {
            pc[a][b] += fminf(xa[a], yb[b]);
            qc[a][b] += fmaxf(xa[a], yb[b]);
          }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      rx[a] += rxc[a];
      ry[a] += ryc[a];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        p[a][b] += pc[a][b];
        q[a][b] += qc[a][b];
      }
    }
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + 4 * ti + a, j = j0 + 4 * tj + b;
      float k;
      if (KIND == kMmdRbf) k = expf(-scale * (rx[a] - 2.f * p[a][b] + ry[b]));             // evaluations.py:17-21
      else if (KIND == kMmdSqDist) k = fmaxf(rx[a] - 2.f * p[a][b] + ry[b], 0.f);
      else if (KIND == kMmdDist) k = sqrtf(fmaxf(rx[a] - 2.f * p[a][b] + ry[b], 0.f));
      else if (KIND == kMmdBrayCurtis) k = 1.f - p[a][b] / (q[a][b] + 1e-8f);              // :34-37
      else if (KIND == kMmdTanimoto) k = p[a][b] / ((rx[a] + ry[b] - p[a][b]) + 1e-8f);    // :50-53
      else k = p[a][b] / (q[a][b] + 1e-8f);                                                // :66-69
      if (i < nx && j < ny) {
        s += k;
        if (kmat) kmat[(long)i * ny + j] = k;
      }
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = sum of partial[0..n) in index order, accumulated in double
__global__ void mmd_sum_kernel(const float* __restrict__ partial, int n, double* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

}  // namespace scldm
