// Small kernels around the fused block: weight packing, conditioning (timestep MLP + class embeddings
// + all-layer adaLN projection), input projection, final layer, CFG blend and ODE state update.
#pragma once
#include "common.hpp"
#include "dit_forward.hpp"

namespace scldm {

// ------------------------------------------------------------------------------------------------
// Weight packing: PyTorch (out, in) row-major fp32 -> one contiguous MFMA-fragment stream per wave
// (layout documented at WStream in dit_forward.hpp).  Element (unit gu, tile ft, lane l, j) lives at
// ((gu*FT + ft)*64 + l)*8 + j; its k index inside the unit's k-step is (l>>5)*8 + j; its row is l&31.
//   units 0-47 : c_attn rows p*256 + w*64 + ft*32 + r          (p = q,k,v)       attn.c_attn.weight (768,256)
//   units 48-63: c_proj rows w*64 + ft*32 + r                                      attn.c_proj.weight (256,256)
//   chunk c, units 0-15: tile rows 0-15 = w1[hid], rows 16-31 = w2[hid], hid = c*128 + w*32 + ft*16 + (r&15)
//   chunk c, units 16-23: mlp.c_proj rows w*64 + ft*32 + r, k = hidden index c*128 + ks*16 + ...
// Hidden indices >= H are zero (exact padding of 684 -> 768).
// ------------------------------------------------------------------------------------------------
template <typename E>
__global__ void pack_layer_kernel(const float* __restrict__ Wqkv, const float* __restrict__ Wproj, const float* __restrict__ W1,
                                  const float* __restrict__ W2, const float* __restrict__ Wcp, E* __restrict__ out, int H,
                                  int n_chunks, int half, int layer, int FT) {
  // FT = 32-row tiles per wave (2: four waves, 1: eight waves); a unit holds FT fragments of 512 elements
  const int UL = units_per_layer(n_chunks, half), NW = 8 / FT, unit_elems = 512 * FT;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)NW * UL * unit_elems) return;
  const int j = idx & 7, l = (idx >> 3) & 63, ft = (int)((idx >> 9) % FT);
  const int gu = (int)(idx / unit_elems), w = gu / UL, u = gu % UL;
  const int r = l & 31, k8 = (l >> 5) * 8 + j;
  const int frow = (w * FT + ft) * 32 + r;  // output feature row of this wave's tile
  float val;
  if (u < 32 && FT == 2) {
    // Q and K of ONE head per pass: units 0-15 = head 0, 16-31 = head 1; inside a unit fragment 0 is the head's Q tile and
    // fragment 1 its K tile of the same k-step (an ordinary two-tile gemm_pass whose "feature tiles" are Q_h and K_h)
    const int head = u >> 4, ks = u & 15;
    val = Wqkv[(size_t)(ft * 256 + (w * 2 + head) * 32 + r) * 256 + ks * 16 + k8];
  } else if (u < 48) {
    const int p = u >> 4, ks = u & 15;
    val = Wqkv[(size_t)(p * 256 + frow) * 256 + ks * 16 + k8];
  } else if (u < 64) {
    const int ks = u - 48;
    val = Wproj[(size_t)frow * 256 + ks * 16 + k8];
  } else if (u >= 64 + n_chunks * kUnitsPerChunk) {
    // trailing half chunk (FT=2): 64 hidden units, wave w owns 16 of them in ONE tile; then a K=64 c_proj pass
    const int vh = u - 64 - n_chunks * kUnitsPerChunk, hid0 = n_chunks * kHC;
    if (vh < 8) {
      const int ks = 2 * vh + ft, hid = hid0 + w * 16 + (r & 15);
      const float* src = (r < 16) ? W1 : W2;
      val = (hid < H) ? src[(size_t)hid * 256 + ks * 16 + k8] : 0.f;
    } else {
      const int hid = hid0 + (vh - 8) * 16 + k8;
      val = (hid < H) ? Wcp[(size_t)frow * H + hid] : 0.f;
    }
  } else {
    const int v = u - 64, c = v / kUnitsPerChunk, vv = v % kUnitsPerChunk;
    if (vv < 16) {
      // FT=2: units 0-7 = tile 0, units 8-15 = tile 1, each unit = k-steps (2j, 2j+1) of that tile (gemm_pass_tile)
      const int tile = (FT == 2) ? (vv >> 3) : ft;
      const int ks = (FT == 2) ? (2 * (vv & 7) + ft) : vv;
      const int hid = c * kHC + (w * FT + tile) * 16 + (r & 15);
      const float* src = (r < 16) ? W1 : W2;
      val = (hid < H) ? src[(size_t)hid * 256 + ks * 16 + k8] : 0.f;
    } else {
      const int hid = c * kHC + (vv - 16) * 16 + k8;
      val = (hid < H) ? Wcp[(size_t)frow * H + hid] : 0.f;
    }
  }
  out[((size_t)layer * NW * UL) * unit_elems + idx] = (E)val;  // [layer][wave][unit][ft][lane][8]
}

// final_layer.linear (din,256) -> 16 fragments of a 32-row tile (rows >= din zero): ((ks*64 + l)*8 + j)
template <typename E>
__global__ void pack_final_kernel(const float* __restrict__ W, E* __restrict__ out, int din) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 16 * 512) return;
  const int j = idx & 7, l = (idx >> 3) & 63, ks = idx >> 9;
  const int row = l & 31, k = ks * 16 + (l >> 5) * 8 + j;
  out[idx] = (E)((row < din) ? W[row * 256 + k] : 0.f);
}

// out[k*ldo + col0 + n] = W[n*K + k]  (transpose so that consecutive threads read consecutive floats)
__global__ void transpose_kernel(const float* __restrict__ W, float* __restrict__ out, int N, int K, int ldo, int col0) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * K) return;
  const int n = idx % N, k = idx / N;
  out[(size_t)k * ldo + col0 + n] = W[(size_t)n * K + k];
}

__global__ void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) dst[idx] = src[idx];
}

// ------------------------------------------------------------------------------------------------
// Conditioning.  One workgroup (256 threads = 256 features) per conditioning row.
//   c = t_mlp(sinusoid(t)) + sum_classes emb_c[label or null]    (layers.py:351-364, nnets.py:380-456)
// writes silu(c) (the input of every adaLN projection, layers.py:206,395).
// ------------------------------------------------------------------------------------------------
constexpr int kMaxClasses = 8;
struct CondArgs {
  const float* t;          // (rows) or scalar when t_stride == 0
  int t_stride;
  const float* w0t;        // (256,256) transposed t_embedder.mlp.0.weight  [k][n]
  const float* b0;
  const float* w2t;        // transposed t_embedder.mlp.2.weight
  const float* b2;
  const float* emb;        // concatenated class tables, (sum(vocab+1), 256)
  int n_classes;
  int emb_row0[kMaxClasses];   // first row of class c in `emb`
  int null_tok[kMaxClasses];   // vocab size of class c (= index of its null token)
  const int64_t* labels[kMaxClasses];  // (rows) or nullptr => null token for every row
  float* silu_c;           // (rows,256)
  int rows;
};

__global__ __launch_bounds__(256) void cond_embed_kernel(const CondArgs a) {
  __shared__ float te[256];
  __shared__ float h1[256];
  const int u = blockIdx.x, n = threadIdx.x;
  const float t = a.t[(size_t)u * a.t_stride];
  {
    const int k = n & 127;
    const float freq = expf(-9.210340371976184f * (float)k / 128.0f);  // exp(-ln(10000) k / half)
    const float arg = t * freq;
    te[n] = (n < 128) ? cosf(arg) : sinf(arg);  // [cos | sin], cos first
  }
  __syncthreads();
  float s = a.b0[n];
#pragma unroll 8
  for (int k = 0; k < 256; ++k) s += a.w0t[k * 256 + n] * te[k];
  h1[n] = silu_f(s);
  __syncthreads();
  float c = a.b2[n];
#pragma unroll 8
  for (int k = 0; k < 256; ++k) c += a.w2t[k * 256 + n] * h1[k];
  for (int ci = 0; ci < a.n_classes; ++ci) {
    int tok = a.null_tok[ci];
    if (a.labels[ci] != nullptr) tok = (int)a.labels[ci][u];
    c += a.emb[(size_t)(a.emb_row0[ci] + tok) * 256 + n];
  }
  a.silu_c[(size_t)u * 256 + n] = silu_f(c);
}

// Scalar-t fast path of the sampler (the ODE solver broadcasts ONE t, integrators.py:103-104): the timestep MLP is the
// same for every conditioning row, so it runs once per step in a single workgroup ...
__global__ __launch_bounds__(256) void t_embed_kernel(const float* __restrict__ t, const float* __restrict__ w0t,
                                                      const float* __restrict__ b0, const float* __restrict__ w2t,
                                                      const float* __restrict__ b2, float* __restrict__ temb) {
  __shared__ float te[256];
  __shared__ float h1[256];
  const int n = threadIdx.x;
  {
    const int k = n & 127;
    const float freq = expf(-9.210340371976184f * (float)k / 128.0f);
    const float arg = t[0] * freq;
    te[n] = (n < 128) ? cosf(arg) : sinf(arg);
  }
  __syncthreads();
  float s = b0[n];
#pragma unroll 8
  for (int k = 0; k < 256; ++k) s += w0t[k * 256 + n] * te[k];
  h1[n] = silu_f(s);
  __syncthreads();
  float c = b2[n];
#pragma unroll 8
  for (int k = 0; k < 256; ++k) c += w2t[k * 256 + n] * h1[k];
  temb[n] = c;
}

// Whole-trajectory variant: evaluation e of a fixed-grid solve sees t = linspace(0,1,steps)[e] (Euler) or
// linspace[e/2 + e%2] (Heun); all of them are embedded by one launch before the loop starts.
__device__ __forceinline__ float linspace01_dev(int idx, int steps) {  // torch.linspace(0,1,steps), fp32, symmetric fill
  const float step = 1.0f / (float)(steps - 1);
  return (idx < steps / 2) ? step * (float)idx : 1.0f - step * (float)(steps - idx - 1);
}
__global__ __launch_bounds__(256) void t_embed_all_kernel(int steps, int heun, const float* __restrict__ w0t,
                                                          const float* __restrict__ b0, const float* __restrict__ w2t,
                                                          const float* __restrict__ b2, float* __restrict__ temb_all) {
  __shared__ float te[256];
  __shared__ float h1[256];
  const int n = threadIdx.x, e = blockIdx.x;
  const float t = heun ? linspace01_dev(e / 2 + (e & 1), steps) : linspace01_dev(e, steps);
  {
    const int k = n & 127;
    const float freq = expf(-9.210340371976184f * (float)k / 128.0f);
    const float arg = t * freq;
    te[n] = (n < 128) ? cosf(arg) : sinf(arg);
  }
  __syncthreads();
  float s = b0[n];
#pragma unroll 8
  for (int k = 0; k < 256; ++k) s += w0t[k * 256 + n] * te[k];
  h1[n] = silu_f(s);
  __syncthreads();
  float c = b2[n];
#pragma unroll 8
  for (int k = 0; k < 256; ++k) c += w2t[k * 256 + n] * h1[k];
  temb_all[(size_t)e * 256 + n] = c;
}

// ... and every row (row 0 = unconditional, row 1 + p*U + u = pass p / unique label row u) only adds its class embeddings.
struct CondRowsArgs {
  const float* temb;
  const float* emb;
  int n_classes, U, rows;
  int emb_row0[kMaxClasses];
  int null_tok[kMaxClasses];
  const int64_t* labels[kMaxClasses];
  uint32_t mask[kMaxClasses];   // per pass: which classes keep their labels
  float* silu_c;
};
__global__ __launch_bounds__(256) void cond_rows_kernel(const CondRowsArgs a) {
  const int r = blockIdx.x, n = threadIdx.x;
  float c = a.temb[n];
  const int p = r > 0 ? (r - 1) / a.U : 0, u = r > 0 ? (r - 1) % a.U : 0;
  for (int ci = 0; ci < a.n_classes; ++ci) {
    int tok = a.null_tok[ci];
    if (r > 0 && a.labels[ci] != nullptr && ((a.mask[p] >> ci) & 1u)) tok = (int)a.labels[ci][u];
    c += a.emb[(size_t)(a.emb_row0[ci] + tok) * 256 + n];
  }
  a.silu_c[(size_t)r * 256 + n] = silu_f(c);
}

// mod[u][n] = bias[n] + sum_k wt[k][n] * silu_c[u][k] for ALL layers at once (n < mod_w).
// Workgroup = 64 columns x 4 k-quarters (split-K, combined through LDS), kAdaRU rows share each weight read;
// grid = (mod_w / 64, rows / kAdaRU): 400 workgroups for the sampler's 15 rows instead of 100.
constexpr int kAdaRU = 8;
__global__ __launch_bounds__(256) void adaln_all_kernel(const float* __restrict__ silu_c, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, float* __restrict__ mod,
                                                        int rows, int mod_w) {
  __shared__ float sc[kAdaRU][256];
  __shared__ float part[4][kAdaRU][64];
  const int col = threadIdx.x & 63, kq = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + col;
  const int u0 = blockIdx.y * kAdaRU;
  for (int i = threadIdx.x; i < kAdaRU * 256; i += 256) {
    const int u = u0 + (i >> 8);
    sc[i >> 8][i & 255] = (u < rows) ? silu_c[(size_t)u * 256 + (i & 255)] : 0.f;
  }
  __syncthreads();
  float acc[kAdaRU];
#pragma unroll
  for (int r = 0; r < kAdaRU; ++r) acc[r] = 0.f;
  if (n < mod_w) {
#pragma unroll 4
    for (int kk = 0; kk < 64; ++kk) {
      const int k = kq * 64 + kk;
      const float w = wt[(size_t)k * mod_w + n];
#pragma unroll
      for (int r = 0; r < kAdaRU; ++r) acc[r] += w * sc[r][k];
    }
  }
#pragma unroll
  for (int r = 0; r < kAdaRU; ++r) part[kq][r][col] = acc[r];
  __syncthreads();
  if (n < mod_w) {
    for (int r = kq; r < kAdaRU; r += 4)
      if (u0 + r < rows) mod[(size_t)(u0 + r) * mod_w + n] = bias[n] + part[0][r][col] + part[1][r][col] + part[2][r][col] + part[3][r][col];
  }
}

// ------------------------------------------------------------------------------------------------
// CFG blend (nnets.py:356-378) + explicit ODE state updates.
// v holds n_direct = 2B unconditional outputs followed by P conditional passes of B rows each.
//   dz[i]      = v[i]                                   i <  B
//   dz[B + i]  = v[B+i] + sum_p scale[p] * (v[2B + pB + i] - v[B+i])
// ------------------------------------------------------------------------------------------------
struct CfgArgs {
  const float* v;
  float* dz;       // (2B, e) may alias nothing else
  int B, e, P;     // e = elements per sample (16*din)
  float scale[kMaxClasses];
};
__global__ void cfg_blend_kernel(const CfgArgs a) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t half = (size_t)a.B * a.e;
  if (idx >= 2 * half) return;
  float r = a.v[idx];
  if (idx >= half) {
    const float u = r;
    for (int p = 0; p < a.P; ++p) r += a.scale[p] * (a.v[2 * half + (size_t)p * half + (idx - half)] - u);
  }
  a.dz[idx] = r;
}

// out = z + h * k
__global__ void axpy_kernel(const float* __restrict__ z, const float* __restrict__ k, float* __restrict__ out, float h, size_t n) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) out[idx] = z[idx] + h * k[idx];
}
// z += hh * (k1 + k2)   (Heun corrector, hh = h/2)
__global__ void heun_kernel(float* __restrict__ z, const float* __restrict__ k1, const float* __restrict__ k2, float hh, size_t n) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) z[idx] = z[idx] + hh * (k1[idx] + k2[idx]);
}

__global__ void fill_row_index_kernel(int32_t* __restrict__ ri, const int32_t* __restrict__ cell_row, int n_direct, int B,
                                      int U, int P) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_direct + P * B) return;
  if (s < n_direct) { ri[s] = 0; return; }
  const int p = (s - n_direct) / B, i = (s - n_direct) % B;
  ri[s] = 1 + p * U + (cell_row ? cell_row[i] : i);
}

__global__ void iota_kernel(int32_t* __restrict__ ri, int n) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) ri[s] = s;
}

}  // namespace scldm
