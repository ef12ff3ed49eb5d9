// Entropic optimal transport for the generation-evaluation metrics (second half):
//   wasserstein(x0, x1, method="sinkhorn", reg, power)          src/scldm/evaluations.py:85-108
// which the reference hands to third-party POT: a, b uniform, M = cdist(x0, x1) ** power, ot.sinkhorn2(a, b, M, reg,
// numItermax=1e7), sqrt for power 2.  POT is not vendored (pyproject.toml:36 "pot"), so this is a restatement of the
// published algorithm - Sinkhorn-Knopp matrix scaling (Cuturi, NIPS 2013) exactly as POT's `sinkhorn_knopp` iterates it:
//     K = exp(-M / reg);  u = 1/n, v = 1/m
//     repeat:  v = b / (K^T u);  u = a / (K v);  stop on a zero / non-finite scaling (keep the previous u, v)
//              every 10th iteration: err = || v * (K^T u) - b ||_2 ;  stop when err < stopThr (1e-9)
//     cost = sum_ij u_i K_ij v_j M_ij
// PARITY UNPINNED at this boundary (no POT here, no reference test); pinned instead to oracle/evaluations.py's float64
// restatement of the same published iteration and to closed-form cases.
// The cost matrix comes from the MMD tile kernel (squared / plain Euclidean distance kinds: D is streamed through LDS once);
// the iteration is two HBM-bound sweeps over K per step: K^T u as row-split column sums, K v as one wave per row.
#pragma once
#include <hip/hip_runtime.h>

namespace scldm {

constexpr int kSkRowSplit = 16;   // row slices of the K^T u sweep (partials are combined in the v update, in order)

// K = exp(-M / reg) elementwise
__global__ void sk_gibbs_kernel(const float* __restrict__ M, float* __restrict__ K, size_t n, float neg_inv_reg) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) K[i] = expf(M[i] * neg_inv_reg);
}

// part[s][j] = sum_{i in slice s} K[i][j] u[i]        grid (ceil(m / 256), kSkRowSplit), block 256: coalesced along j
__global__ __launch_bounds__(256) void sk_ktu_kernel(const float* __restrict__ K, const float* __restrict__ u, int n, int m,
                                                     float* __restrict__ part) {
  const int j = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  const int i0 = (int)((long long)n * s / kSkRowSplit), i1 = (int)((long long)n * (s + 1) / kSkRowSplit);
  if (j >= m) return;
  float acc = 0.f;
#pragma unroll 4
  for (int i = i0; i < i1; ++i) acc += K[(size_t)i * m + j] * u[i];
  part[(size_t)s * m + j] = acc;
}
// ktu[j] = sum_s part[s][j];  v[j] = b / ktu[j]   (flag[0] |= 1 when ktu is 0 or v is not finite)
__global__ void sk_v_update_kernel(const float* __restrict__ part, int m, float b, float* __restrict__ ktu, float* __restrict__ v,
                                   int* __restrict__ flag) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < kSkRowSplit; ++r) s += part[(size_t)r * m + j];
  ktu[j] = s;
  const float vj = b / s;
  v[j] = vj;
  if (s == 0.f || !isfinite(vj)) atomicOr(flag, 1);
}
// u[i] = a / sum_j K[i][j] v[j]      one wave per row, 4 rows per 256-thread block
__global__ __launch_bounds__(256) void sk_u_update_kernel(const float* __restrict__ K, const float* __restrict__ v, int n, int m, float a,
                                                          float* __restrict__ u, int* __restrict__ flag) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* row = K + (size_t)i * m;
  float acc = 0.f;
  for (int j = lane; j < m; j += 64) acc += row[j] * v[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    const float ui = a / acc;
    u[i] = ui;
    if (!isfinite(ui)) atomicOr(flag, 1);
  }
}
// err^2 = sum_j (v[j] * ktu[j] - b)^2   (single block; ktu = K^T u for the CURRENT u)
__global__ __launch_bounds__(256) void sk_err_kernel(const float* __restrict__ part, const float* __restrict__ v, int m, float b,
                                                     float* __restrict__ err2) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int j = threadIdx.x; j < m; j += 256) {
    float k = 0.f;
#pragma unroll
    for (int r = 0; r < kSkRowSplit; ++r) k += part[(size_t)r * m + j];
    const double d = (double)v[j] * k - b;
    s += d * d;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) err2[0] = (float)sh[0];
}
// rowcost[i] = u[i] * sum_j K[i][j] v[j] M[i][j]  (double accumulation per row; summed in order by mmd_sum-style reduction)
__global__ __launch_bounds__(256) void sk_cost_kernel(const float* __restrict__ K, const float* __restrict__ M, const float* __restrict__ u,
                                                      const float* __restrict__ v, int n, int m, float* __restrict__ rowcost) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  double acc = 0.0;
  for (int j = lane; j < m; j += 64) acc += (double)(K[(size_t)i * m + j] * v[j]) * M[(size_t)i * m + j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) rowcost[i] = (float)(acc * u[i]);
}
// commit the iteration unless one of its scalings was singular (then the previous u, v stay: POT's "numerical errors" exit)
__global__ void sk_commit_kernel(float* __restrict__ u, const float* __restrict__ u_new, int n, float* __restrict__ v,
                                 const float* __restrict__ v_new, int m, const int* __restrict__ flag) {
  if (*flag) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) u[i] = u_new[i];
  if (i < m) v[i] = v_new[i];
}
__global__ void sk_fill_kernel(float* __restrict__ p, int n, float val) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = val;
}

}  // namespace scldm
