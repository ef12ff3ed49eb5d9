// The 16-token side of a cell in the VAE training backward (row V1), second version: ONE WORKGROUP OF 256 THREADS PER CELL.
//   trunk Blocks                               src/scldm/layers.py:222-226 (x += c_proj(attn(LN_1 x)); x += MLP(LN_2 x); 8 heads x 4)
//   encoder ends (c_proj of the pooling, LN_2, MLP, pos, latent Linear + LN)   layers.py:326-330, nnets.py:139-144
//   decoder ends (LN, latent Linear, K|V = c_attn(LN_1 h))                     nnets.py:196-208, layers.py:312-318
// The first version (vae_train.hpp: one token per lane, four cells per wave, weights as SGPR operands) walks eight layers in one
// wave: at batch 32 that is 8 waves on 1 024 SIMDs, 3.3-3.7 ms per kernel whatever the batch, with 3.4-4.4 k spilled registers.
// Here a token is spread over 16 lanes (a DPP row: LayerNorm statistics are row rotations), a cell's 16 tokens over the four waves
// of a workgroup, and every operand lives in LDS:
//   * a layer's five weight matrices (50 KB, rows padded to 36 floats) are copied into LDS once per layer by all 256 threads,
//     the next layer's copy in flight (registers) while the current layer's weight gradients are contracted;
//   * y = W x: lane (token, j) owns outputs j, j + 16, ... and reads whole weight rows as 16-byte pieces (conflict-free: 36 j mod 64
//     are distinct multiples of 4) against its token's activation row (broadcast);
//   * dx = W^T dy: lane (token, j) owns input features 4 (j & 7) .. + 3 for the output rows whose bit 3 equals j >> 3 (the two
//     halves are 8 rows = 32 banks apart), one row_ror:8 adds the halves;
//   * dW = sum over the cell's 16 tokens of dy (x) x: 8 lanes per output row, 4 input features each; one partial per CELL, added in
//     index order by reduce_jobs_kernel (deterministic);
//   * attention (8 heads x 4, 16 keys): lane = (token, head, half of the keys); softmax / dq / dk / dv partner sums are quad_perm
//     exchanges; the probabilities stay in registers from the recompute to the backward.
// A token's lanes sit in one wave, so most steps need only wave-level ordering; workgroup barriers stand where tokens meet (K / V,
// the key-side backward, the weight gradients, the weight copy): four per layer.
#pragma once
#include "vae_train.hpp"

namespace scldm {
namespace vtrain {
namespace wide {

constexpr int kThreads = 256;
constexpr int kP = 36;     // floats per 32-wide LDS row
constexpr int kQ = 100;    // floats per 96-wide LDS row (100 mod 64 = 36: the same bank walk as kP)
constexpr int kPS = 17;    // pitch of the probability / d-score rows of the key-side exchange

// ---- LDS map (float offsets) ------------------------------------------------------------------------------------------------------
constexpr int W_QKV = 0, W_P = W_QKV + 96 * kP, W_1 = W_P + 32 * kP, W_2 = W_1 + 96 * kP, W_CT = W_2 + 96 * kP;
constexpr int W_E1 = W_CT + 96 * kP;        // 64 rows: an "end" matrix (c_attn K|V of the decoder's cross attention)
constexpr int W_E2 = W_E1 + 64 * kP;        // 32 rows: latent Linear (either side), padded to 32 x 32
constexpr int A_X = W_E2 + 32 * kP;         // layer input
constexpr int A_HN = A_X + 16 * kP;
constexpr int A_AO = A_HN + 16 * kP;
constexpr int A_H2 = A_AO + 16 * kP;
constexpr int A_DM0 = A_H2 + 16 * kP;       // gradient w.r.t. the layer's output (ping)
constexpr int A_DM1 = A_DM0 + 16 * kP;      // ... (pong)
constexpr int A_DX1 = A_DM1 + 16 * kP;      // gradient w.r.t. x1 = x + c_proj(attn)
constexpr int A_DAO = A_DX1 + 16 * kP;
constexpr int A_DH2 = A_DAO + 16 * kP;
constexpr int A_T2 = A_DH2 + 16 * kP;       // dh2 * xhat2 (LN_2 weight gradient terms)
constexpr int A_DHN = A_T2 + 16 * kP;
constexpr int A_T1 = A_DHN + 16 * kP;
constexpr int A_QKV = A_T1 + 16 * kP;
constexpr int A_DA = A_QKV + 16 * kQ;
constexpr int A_DB = A_DA + 16 * kQ;
constexpr int A_HID = A_DB + 16 * kQ;
constexpr int A_DQKV = A_HID + 16 * kQ;
constexpr int A_PP = A_DQKV + 16 * kQ;      // [head][query][key] probabilities
constexpr int A_DS = A_PP + 8 * 16 * kPS;   // ... d scores (scaled)
constexpr int LDS_FLOATS = A_DS + 8 * 16 * kPS;
constexpr int LDS_BYTES = LDS_FLOATS * 4;
static_assert(LDS_BYTES <= 160 * 1024, "cell-side LDS image");

// ---- synchronisation ----------------------------------------------------------------------------------------------------------------
// tsync: orders LDS traffic among the 16 lanes of a token (one wave: LDS instructions of a wave complete in order, the fences stop
// the compiler from moving them).  SCLDM_VAE_TSYNC_WG=1 builds with full barriers instead (A/B and debugging).
#ifndef SCLDM_VAE_TSYNC_WG
#define SCLDM_VAE_TSYNC_WG 0
#endif
__device__ __forceinline__ void tsync() {
#if SCLDM_VAE_TSYNC_WG
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// ---- lane exchanges -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ror8(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)); }
__device__ __forceinline__ float row16_sum(float v) {
  v += ror8(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
__device__ __forceinline__ float pair_get(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)); }   // lane ^ 1
__device__ __forceinline__ float pair_sum(float v) { return v + pair_get(v); }
__device__ __forceinline__ f32x4 pair_sum4(f32x4 v) { return f32x4{pair_sum(v[0]), pair_sum(v[1]), pair_sum(v[2]), pair_sum(v[3])}; }
__device__ __forceinline__ f32x4 half_sum4(f32x4 v) { return f32x4{v[0] + ror8(v[0]), v[1] + ror8(v[1]), v[2] + ror8(v[2]), v[3] + ror8(v[3])}; }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]))); }
__device__ __forceinline__ f32x4 fma4(float s, f32x4 a, f32x4 c) { return f32x4{fmaf(s, a[0], c[0]), fmaf(s, a[1], c[1]), fmaf(s, a[2], c[2]), fmaf(s, a[3], c[3])}; }
__device__ __forceinline__ const f32x4* v4(const float* p) { return reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4* v4(float* p) { return reinterpret_cast<f32x4*>(p); }
__device__ __forceinline__ f32x4 z4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// ---- weight copies into LDS ---------------------------------------------------------------------------------------------------------
// rows of 32 floats (row-major, contiguous) -> [rows_total][kP]; rows >= rows_valid are zero.  Split into a load half (registers) and
// a store half so that the loads of the next layer fly during the weight-gradient contraction.
template <int ROWS>
struct RowCopy {
  static constexpr int N = (ROWS * 8 + kThreads - 1) / kThreads;
  f32x4 r[N];
  __device__ __forceinline__ void load(const float* __restrict__ src, int rows_valid, int tid) {
    const bool al = (reinterpret_cast<size_t>(src) & 15) == 0;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int idx = tid + n * kThreads, row = idx >> 3;
      f32x4 v = z4();
      if (idx < ROWS * 8 && row < rows_valid) {
        if (al) v = v4(src)[idx];
        else { const float* s = src + (size_t)idx * 4; v = f32x4{s[0], s[1], s[2], s[3]}; }
      }
      r[n] = v;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ dst, int tid) const {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int idx = tid + n * kThreads;
      if (idx < ROWS * 8) *v4(dst + (idx >> 3) * kP + 4 * (idx & 7)) = r[n];
    }
  }
};
struct LayerCopy {
  RowCopy<96> qkv, w1, w2, wct;
  RowCopy<32> wp;
  __device__ __forceinline__ void load(const BlockW& w, int tid) {
    qkv.load(w.wqkv, 96, tid); wp.load(w.wp, 32, tid); w1.load(w.mlp.w1, w.mlp.H, tid); w2.load(w.mlp.w2, w.mlp.H, tid); wct.load(w.mlp.wct, w.mlp.H, tid);
  }
  __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
    qkv.store(S + W_QKV, tid); wp.store(S + W_P, tid); w1.store(S + W_1, tid); w2.store(S + W_2, tid); wct.store(S + W_CT, tid);
  }
};
// a small matrix with its own source pitch -> [rows_total][kP], zero padded (the latent Linears: (32, n_lat) / (n_lat, 32))
__device__ __forceinline__ void copy_small(float* __restrict__ dst, const float* __restrict__ src, int rows_valid, int cols_valid, int ld, int rows_total, int tid) {
  for (int idx = tid; idx < rows_total * 32; idx += kThreads) {
    const int r = idx >> 5, c = idx & 31;
    dst[r * kP + c] = (r < rows_valid && c < cols_valid) ? src[(size_t)r * ld + c] : 0.f;
  }
}

// ---- the three contractions -----------------------------------------------------------------------------------------------------------
// y[o] = W[o][:] . x for o = j, j + 16, ...: f(m, o, value).  k-outer with the next k piece of every output row requested before
// this piece's FMAs (one wave per SIMD: nothing else hides the LDS latency; SCLDM_VAE_LIN_PIPE=0 = one output row at a time).
#ifndef SCLDM_VAE_LIN_PIPE
#define SCLDM_VAE_LIN_PIPE 1
#endif
template <int OUT, class F>
__device__ __forceinline__ void lin32(const float* __restrict__ W, const float* __restrict__ xrow, int j, F f) {
  f32x4 x[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) x[q] = v4(xrow)[q];
#pragma unroll
  for (int m = 0; m < OUT / 16; ++m) {
    const int o = j + 16 * m;
    const f32x4* wr = v4(W + o * kP);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      const f32x4 wa = wr[q], wb = wr[q + 1];
      s0 = fmaf(wa[0], x[q][0], s0); s0 = fmaf(wa[1], x[q][1], s0); s0 = fmaf(wa[2], x[q][2], s0); s0 = fmaf(wa[3], x[q][3], s0);
      s1 = fmaf(wb[0], x[q + 1][0], s1); s1 = fmaf(wb[1], x[q + 1][1], s1); s1 = fmaf(wb[2], x[q + 1][2], s1); s1 = fmaf(wb[3], x[q + 1][3], s1);
    }
    f(m, o, s0 + s1);
  }
}
// the cell-side kernels' form (one wave per SIMD, registers to spare): all output rows advance together
template <int OUT, class F>
__device__ __forceinline__ void lin32p(const float* __restrict__ W, const float* __restrict__ xrow, int j, F f) {
#if SCLDM_VAE_LIN_PIPE
  constexpr int NO = OUT / 16;
  f32x4 x[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) x[q] = v4(xrow)[q];
  f32x4 wc[NO], wn[NO];
  float s[NO];
#pragma unroll
  for (int m = 0; m < NO; ++m) { wc[m] = *v4(W + (j + 16 * m) * kP); s[m] = 0.f; }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (q < 7) {
#pragma unroll
      for (int m = 0; m < NO; ++m) wn[m] = *v4(W + (j + 16 * m) * kP + 4 * (q + 1));
    }
#pragma unroll
    for (int m = 0; m < NO; ++m) {
      s[m] = fmaf(wc[m][0], x[q][0], s[m]); s[m] = fmaf(wc[m][1], x[q][1], s[m]);
      s[m] = fmaf(wc[m][2], x[q][2], s[m]); s[m] = fmaf(wc[m][3], x[q][3], s[m]);
    }
    if (q < 7) {
#pragma unroll
      for (int m = 0; m < NO; ++m) wc[m] = wn[m];
    }
  }
#pragma unroll
  for (int m = 0; m < NO; ++m) f(m, j + 16 * m, s[m]);
#else
  lin32<OUT>(W, xrow, j, f);
#endif
}
// acc[c] += sum over the rows o with ((o >> 3) & 1) == (j >> 3) of W[o][4 (j & 7) + c] dy[o]   (finish with half_sum4)
template <int OUT>
__device__ __forceinline__ void lin32_t_acc(const float* __restrict__ W, const float* __restrict__ dyrow, int j, f32x4& acc) {
  const int ib = j & 7, hf = j >> 3;
#pragma unroll
  for (int g = 0; g < OUT / 16; ++g)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int o0 = 16 * g + 8 * hf + 4 * c;
      const f32x4 d = *v4(dyrow + o0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fma4(d[e], *v4(W + (o0 + e) * kP + 4 * ib), acc);
    }
}
// the cell-side kernels' form: the next group's five loads are requested before this group's FMAs, two accumulator chains
template <int OUT>
__device__ __forceinline__ void lin32_t_accp(const float* __restrict__ W, const float* __restrict__ dyrow, int j, f32x4& acc) {
#if SCLDM_VAE_LIN_PIPE
  const int ib = j & 7, hf = j >> 3;
  constexpr int NG = OUT / 8;          // groups of four output rows: o0 = 16 (t >> 1) + 8 hf + 4 (t & 1)
  f32x4 dc = *v4(dyrow + 8 * hf), wc[4], dn = dc, wn[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { wc[e] = *v4(W + (8 * hf + e) * kP + 4 * ib); wn[e] = wc[e]; }
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    if (t + 1 < NG) {
      const int o1 = 16 * ((t + 1) >> 1) + 8 * hf + 4 * ((t + 1) & 1);
      dn = *v4(dyrow + o1);
#pragma unroll
      for (int e = 0; e < 4; ++e) wn[e] = *v4(W + (o1 + e) * kP + 4 * ib);
    }
    acc = fma4(dc[0], wc[0], acc);
    acc2 = fma4(dc[1], wc[1], acc2);
    acc = fma4(dc[2], wc[2], acc);
    acc2 = fma4(dc[3], wc[3], acc2);
    dc = dn;
#pragma unroll
    for (int e = 0; e < 4; ++e) wc[e] = wn[e];
  }
  acc += acc2;
#else
  lin32_t_acc<OUT>(W, dyrow, j, acc);
#endif
}
// dW[o][i] = sum_t DY[t][o] X[t][i] for o = tid >> 3 (+ 32 m), i = 4 (tid & 7) .. + 3; stored at dst[o * ld_o + i * ld_i]
template <int OUT>
__device__ __forceinline__ void wgrad(const float* __restrict__ DY, int py, const float* __restrict__ X, float* __restrict__ dst, int ld_o, int ld_i, int tid) {
  const int i4 = tid & 7, oo = tid >> 3;
  f32x4 acc[OUT / 32];
#pragma unroll
  for (int m = 0; m < OUT / 32; ++m) acc[m] = z4();
#pragma unroll 4
  for (int t = 0; t < kT; ++t) {
    const f32x4 xv = *v4(X + t * kP + 4 * i4);
#pragma unroll
    for (int m = 0; m < OUT / 32; ++m) acc[m] = fma4(DY[t * py + oo + 32 * m], xv, acc[m]);
  }
#pragma unroll
  for (int m = 0; m < OUT / 32; ++m) {
    const int o = oo + 32 * m;
    if (ld_i == 1) *v4(dst + (size_t)o * ld_o + 4 * i4) = acc[m];
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[(size_t)o * ld_o + (size_t)(4 * i4 + e) * ld_i] = acc[m][e];
    }
  }
}
// column sums over the 16 tokens: threads 0..31 of a 64-thread group -> A, 32..63 -> B
__device__ __forceinline__ float colsum16(const float* __restrict__ A, int f) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < kT; ++t) s += A[t * kP + f];
  return s;
}

// ---- LayerNorm on a lane's two features (j, j + 16) ---------------------------------------------------------------------------------
struct Ln { float h0, h1, r; };
__device__ __forceinline__ Ln ln_own(float v0, float v1, float eps) {
  const float mean = row16_sum(v0 + v1) * (1.0f / 32);
  const float d0 = v0 - mean, d1 = v1 - mean;
  const float var = row16_sum(fmaf(d0, d0, d1 * d1)) * (1.0f / 32);
  Ln s;
  s.r = 1.0f / sqrtf(var + eps);
  s.h0 = d0 * s.r;
  s.h1 = d1 * s.r;
  return s;
}
// gradient through xhat = (x - mean) rstd given d xhat of the two features
__device__ __forceinline__ void ln_back(const Ln& s, float g0, float g1, float& o0, float& o1) {
  const float a = row16_sum(g0 + g1) * (1.0f / 32);
  const float b = row16_sum(fmaf(g0, s.h0, g1 * s.h1)) * (1.0f / 32);
  o0 = s.r * (g0 - a - s.h0 * b);
  o1 = s.r * (g1 - a - s.h1 * b);
}

// ---- per-lane state of a Block between its recompute and its backward -----------------------------------------------------------------
struct BlockState {
  float x0, x1;        // layer input (features j, j + 16)
  Ln n1, n2;
  float m0, m1;        // x1 = x + c_proj(attn)
  float p[8];          // probabilities of (token, head j >> 1) over keys 8 (j & 1) ..
};

// LN_1 -> qkv -> attention -> c_proj -> LN_2: leaves HN, QKV, AO, H2 in LDS
__device__ __forceinline__ void block_front(const BlockW& w, float* __restrict__ S, int tok, int j, float eps, BlockState& b) {
  b.n1 = ln_own(b.x0, b.x1, eps);
  S[A_HN + tok * kP + j] = fmaf(b.n1.h0, w.ln1_w[j], w.ln1_b[j]);
  S[A_HN + tok * kP + j + 16] = fmaf(b.n1.h1, w.ln1_w[j + 16], w.ln1_b[j + 16]);
  tsync();
  lin32p<96>(S + W_QKV, S + A_HN + tok * kP, j, [&](int, int o, float v) { S[A_QKV + tok * kQ + o] = v; });
  __syncthreads();
  {
    const int h = j >> 1, kh = j & 1;
    const f32x4 q = *v4(S + A_QKV + tok * kQ + 4 * h);
    float s[8], mx = -3.0e38f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      s[kk] = dot4(q, *v4(S + A_QKV + (8 * kh + kk) * kQ + 32 + 4 * h)) * kTScale;
      mx = fmaxf(mx, s[kk]);
    }
    mx = fmaxf(mx, pair_get(mx));
    float l = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) l += __expf(s[kk] - mx);
    l = pair_sum(l);
    const float lse = mx + __logf(l);
    f32x4 ao = z4();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      b.p[kk] = __expf(s[kk] - lse);
      ao = fma4(b.p[kk], *v4(S + A_QKV + (8 * kh + kk) * kQ + 64 + 4 * h), ao);
    }
    ao = pair_sum4(ao);
    if (kh == 0) *v4(S + A_AO + tok * kP + 4 * h) = ao;
  }
  tsync();
  lin32p<32>(S + W_P, S + A_AO + tok * kP, j, [&](int m, int, float v) { if (m == 0) b.m0 = b.x0 + v; else b.m1 = b.x1 + v; });
  b.n2 = ln_own(b.m0, b.m1, eps);
  S[A_H2 + tok * kP + j] = fmaf(b.n2.h0, w.ln2_w[j], w.ln2_b[j]);
  S[A_H2 + tok * kP + j + 16] = fmaf(b.n2.h1, w.ln2_w[j + 16], w.ln2_b[j + 16]);
  tsync();
}

// SwiGLU forward from H2: returns MLP(h2) for the lane's two features (through A_DH2 as the exchange row)
__device__ __forceinline__ void mlp_forward(float* __restrict__ S, int tok, int j, float& y0, float& y1) {
  float a[6], g[6];
  lin32p<96>(S + W_1, S + A_H2 + tok * kP, j, [&](int m, int, float v) { a[m] = v; });
  lin32p<96>(S + W_2, S + A_H2 + tok * kP, j, [&](int m, int, float v) { g[m] = v; });
#pragma unroll
  for (int m = 0; m < 6; ++m) S[A_HID + tok * kQ + j + 16 * m] = a[m] * sigm(a[m]) * g[m];
  tsync();
  f32x4 acc = z4();
  lin32_t_accp<96>(S + W_CT, S + A_HID + tok * kQ, j, acc);
  acc = half_sum4(acc);
  if (j < 8) *v4(S + A_DH2 + tok * kP + 4 * j) = acc;
  tsync();
  y0 = S[A_DH2 + tok * kP + j];
  y1 = S[A_DH2 + tok * kP + j + 16];
  tsync();
}

// SwiGLU + LN_2 backward.  In: H2, dm rows at DM (LDS), the lane's dm values d0 / d1, LN_2 state.  Out: DA, DB, HID, DH2, T2 (LDS) and
// the gradient w.r.t. the LN_2 input INCLUDING the residual path (d0 / d1 updated in place).
__device__ __forceinline__ void mlp_back(float* __restrict__ S, int DM, const float* __restrict__ ln2_w, const Ln& n2, int tok, int j, float& d0, float& d1) {
  float a[6], g[6];
  lin32p<96>(S + W_1, S + A_H2 + tok * kP, j, [&](int m, int, float v) { a[m] = v; });
  lin32p<96>(S + W_2, S + A_H2 + tok * kP, j, [&](int m, int, float v) { g[m] = v; });
  lin32p<96>(S + W_CT, S + DM + tok * kP, j, [&](int m, int o, float dh) {
    const float s = sigm(a[m]), sa = a[m] * s;
    S[A_HID + tok * kQ + o] = sa * g[m];
    S[A_DA + tok * kQ + o] = dh * g[m] * (s * (1.0f + a[m] * (1.0f - s)));
    S[A_DB + tok * kQ + o] = dh * sa;
  });
  tsync();
  f32x4 acc = z4();
  lin32_t_accp<96>(S + W_1, S + A_DA + tok * kQ, j, acc);
  lin32_t_accp<96>(S + W_2, S + A_DB + tok * kQ, j, acc);
  acc = half_sum4(acc);
  if (j < 8) *v4(S + A_DH2 + tok * kP + 4 * j) = acc;
  tsync();
  const float h0 = S[A_DH2 + tok * kP + j], h1 = S[A_DH2 + tok * kP + j + 16];
  S[A_T2 + tok * kP + j] = h0 * n2.h0;
  S[A_T2 + tok * kP + j + 16] = h1 * n2.h1;
  float o0, o1;
  ln_back(n2, h0 * ln2_w[j], h1 * ln2_w[j + 16], o0, o1);
  d0 += o0;
  d1 += o1;
}

// attention + LN_1 backward.  In: the lane's d x1 (d0 / d1; also written to DX1).  Out: DAO, DQKV, DHN, T1 in LDS; d0 / d1 = d x.
__device__ __forceinline__ void attn_back(const BlockW& w, float* __restrict__ S, int tok, int j, const BlockState& b, float& d0, float& d1) {
  S[A_DX1 + tok * kP + j] = d0;
  S[A_DX1 + tok * kP + j + 16] = d1;
  tsync();
  {
    f32x4 acc = z4();
    lin32_t_accp<32>(S + W_P, S + A_DX1 + tok * kP, j, acc);
    acc = half_sum4(acc);
    if (j < 8) *v4(S + A_DAO + tok * kP + 4 * j) = acc;
  }
  tsync();
  const int h = j >> 1, kh = j & 1;
  {
    const f32x4 dao = *v4(S + A_DAO + tok * kP + 4 * h);
    float dp[8], dg = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      dp[kk] = dot4(dao, *v4(S + A_QKV + (8 * kh + kk) * kQ + 64 + 4 * h));
      dg = fmaf(b.p[kk], dp[kk], dg);
    }
    dg = pair_sum(dg);
    f32x4 dq = z4();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float ds = b.p[kk] * (dp[kk] - dg) * kTScale;
      dq = fma4(ds, *v4(S + A_QKV + (8 * kh + kk) * kQ + 32 + 4 * h), dq);
      S[A_DS + (h * 16 + tok) * kPS + 8 * kh + kk] = ds;
      S[A_PP + (h * 16 + tok) * kPS + 8 * kh + kk] = b.p[kk];
    }
    dq = pair_sum4(dq);
    if (kh == 0) *v4(S + A_DQKV + tok * kQ + 4 * h) = dq;
  }
  __syncthreads();
  {   // key side: this lane's token is the KEY; queries 8 kh .. 8 kh + 7
    f32x4 dk = z4(), dv = z4();
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const int t = 8 * kh + tt;
      dk = fma4(S[A_DS + (h * 16 + t) * kPS + tok], *v4(S + A_QKV + t * kQ + 4 * h), dk);
      dv = fma4(S[A_PP + (h * 16 + t) * kPS + tok], *v4(S + A_DAO + t * kP + 4 * h), dv);
    }
    dk = pair_sum4(dk);
    dv = pair_sum4(dv);
    if (kh == 0) {
      *v4(S + A_DQKV + tok * kQ + 32 + 4 * h) = dk;
      *v4(S + A_DQKV + tok * kQ + 64 + 4 * h) = dv;
    }
  }
  tsync();
  {
    f32x4 acc = z4();
    lin32_t_accp<96>(S + W_QKV, S + A_DQKV + tok * kQ, j, acc);
    acc = half_sum4(acc);
    if (j < 8) *v4(S + A_DHN + tok * kP + 4 * j) = acc;
  }
  tsync();
  const float h0 = S[A_DHN + tok * kP + j], h1 = S[A_DHN + tok * kP + j + 16];
  S[A_T1 + tok * kP + j] = h0 * b.n1.h0;
  S[A_T1 + tok * kP + j + 16] = h1 * b.n1.h1;
  float o0, o1;
  ln_back(b.n1, h0 * w.ln1_w[j], h1 * w.ln1_w[j + 16], o0, o1);
  d0 += o0;
  d1 += o1;
}

// the MLP's three weight gradients + LN_2's vectors of one partial (threads 0..63 write the vectors)
__device__ __forceinline__ void mlp_wgrads(float* __restrict__ S, int DM, float* __restrict__ P1, float* __restrict__ P2, float* __restrict__ PC,
                                           float* __restrict__ PLW, float* __restrict__ PLB, int tid) {
  wgrad<96>(S + A_DA, kQ, S + A_H2, P1, 32, 1, tid);
  wgrad<96>(S + A_DB, kQ, S + A_H2, P2, 32, 1, tid);
  wgrad<96>(S + A_HID, kQ, S + DM, PC, 1, kHP, tid);      // dWc[i][u] = sum_t dm[t][i] hid[t][u]
  if (tid < 32) PLW[tid] = colsum16(S + A_T2, tid);
  else if (tid < 64) PLB[tid - 32] = colsum16(S + A_DH2, tid - 32);
}

// backward of one Block.  In: x of the layer in b.x0 / b.x1, dm rows at DM (LDS) and d0 / d1; the layer's weights in LDS.
// Out: d0 / d1 = gradient w.r.t. the layer's input, also written as rows to DN; the layer's partial to P.  `next`: the weights to put
// into LDS for whatever runs after this layer (loaded during the weight-gradient contraction), or nullptr.
__device__ __forceinline__ void block_back(const BlockW& w, const BlockW* next, float* __restrict__ S, int DM, int DN, float* __restrict__ P,
                                           int tid, float eps, BlockState& b, float& d0, float& d1) {
  const int tok = tid >> 4, j = tid & 15;
  block_front(w, S, tok, j, eps, b);
  mlp_back(S, DM, w.ln2_w, b.n2, tok, j, d0, d1);
  attn_back(w, S, tok, j, b, d0, d1);
  S[DN + tok * kP + j] = d0;
  S[DN + tok * kP + j + 16] = d1;
  __syncthreads();                       // every token's rows are complete; nobody reads the weights any more
  LayerCopy lc;
  if (next) lc.load(*next, tid);
  wgrad<96>(S + A_DQKV, kQ, S + A_HN, P + TP_WQKV, 32, 1, tid);
  wgrad<32>(S + A_DX1, kP, S + A_AO, P + TP_WP, 32, 1, tid);
  mlp_wgrads(S, DM, P + TP_W1, P + TP_W2, P + TP_WC, P + TP_LN2W, P + TP_LN2B, tid);
  if (tid >= 64 && tid < 96) P[TP_LN1W + tid - 64] = colsum16(S + A_T1, tid - 64);
  else if (tid >= 96 && tid < 128) P[TP_LN1B + tid - 96] = colsum16(S + A_DHN, tid - 96);
  if (next) lc.store(S, tid);
  __syncthreads();
}

// forward of one Block (x in b.x0 / b.x1 -> out in the same)
__device__ __forceinline__ void block_forward(const BlockW& w, float* __restrict__ S, int tok, int j, float eps, BlockState& b) {
  block_front(w, S, tok, j, eps, b);
  float y0, y1;
  mlp_forward(S, tok, j, y0, y1);
  b.x0 = b.m0 + y0;
  b.x1 = b.m1 + y1;
}

// =================================================================================================================================
// Decoder cell side
// =================================================================================================================================
// z -> LN (no affine, over n_lat) for the lane's features j, j + 16 (zero beyond n_lat)
__device__ __forceinline__ void latent_ln(float z0, float z1, int n_lat, int j, float eps, float& zn0, float& zn1, float& rz) {
  const bool in0 = j < n_lat, in1 = j + 16 < n_lat;
  const float inv = 1.0f / n_lat;
  const float mean = row16_sum((in0 ? z0 : 0.f) + (in1 ? z1 : 0.f)) * inv;
  const float e0 = in0 ? z0 - mean : 0.f, e1 = in1 ? z1 - mean : 0.f;
  rz = 1.0f / sqrtf(row16_sum(fmaf(e0, e0, e1 * e1)) * inv + eps);
  zn0 = e0 * rz;
  zn1 = e1 * rz;
}

__global__ __launch_bounds__(kThreads) void dec_cell_fwd_kernel(const DecCellTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const int tid = threadIdx.x, tok = tid >> 4, j = tid & 15, cell = blockIdx.x, L = a.n_layer;
  LayerCopy lc;
  if (L > 0) lc.load(a.blocks.b[0], tid);
  copy_small(S + W_E2, a.w_in, 32, a.n_lat, a.n_lat, 32, tid);
  RowCopy<64> ckv;
  ckv.load(a.wkv, 64, tid);
  ckv.store(S + W_E1, tid);
  if (L > 0) lc.store(S, tid);
  const float* zr = a.z + ((size_t)cell * kT + tok) * a.n_lat;
  float zn0, zn1, rz;
  latent_ln(j < a.n_lat ? zr[j] : 0.f, j + 16 < a.n_lat ? zr[j + 16] : 0.f, a.n_lat, j, a.eps, zn0, zn1, rz);
  S[A_X + tok * kP + j] = zn0;
  S[A_X + tok * kP + j + 16] = zn1;
  __syncthreads();
  BlockState b;
  lin32p<32>(S + W_E2, S + A_X + tok * kP, j, [&](int m, int, float v) { if (m == 0) b.x0 = v; else b.x1 = v; });
  float* XS = a.xsave + ((size_t)cell * (L + 1) * kT + tok) * 32;
  for (int l = 0; l < L; ++l) {
    XS[(size_t)l * kT * 32 + j] = b.x0;
    XS[(size_t)l * kT * 32 + j + 16] = b.x1;
    if (l + 1 < L) lc.load(a.blocks.b[l + 1], tid);
    block_forward(a.blocks.b[l], S, tok, j, a.eps, b);
    __syncthreads();
    if (l + 1 < L) { lc.store(S, tid); __syncthreads(); }
  }
  XS[(size_t)L * kT * 32 + j] = b.x0;
  XS[(size_t)L * kT * 32 + j + 16] = b.x1;
  const Ln n = ln_own(b.x0, b.x1, a.eps);
  S[A_HN + tok * kP + j] = fmaf(n.h0, a.cln1_w[j], a.cln1_b[j]);
  S[A_HN + tok * kP + j + 16] = fmaf(n.h1, a.cln1_w[j + 16], a.cln1_b[j + 16]);
  tsync();
  float* kv = a.kv + ((size_t)cell * kT + tok) * 64;
  lin32p<64>(S + W_E1, S + A_HN + tok * kP, j, [&](int, int o, float v) { kv[o] = v; });
}

__global__ __launch_bounds__(kThreads) void dec_cell_bwd_kernel(const DecCellTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const int tid = threadIdx.x, tok = tid >> 4, j = tid & 15, cell = blockIdx.x, L = a.n_layer;
  float* P = a.part + (size_t)cell * dc_size(L);
  const float* XS = a.xsave + ((size_t)cell * (L + 1) * kT + tok) * 32;
  LayerCopy lc;
  if (L > 0) lc.load(a.blocks.b[L - 1], tid);
  copy_small(S + W_E2, a.w_in, 32, a.n_lat, a.n_lat, 32, tid);
  {
    RowCopy<64> ckv;
    ckv.load(a.wkv, 64, tid);
    ckv.store(S + W_E1, tid);
  }
  // d kv of this token (outputs j, j + 16, j + 32, j + 48): the per-chunk partials in index order
  {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < a.chunks; ++c) {
      const float* src = a.dkv_part + ((size_t)(cell * a.chunks + c) * kT + tok) * 64;
#pragma unroll
      for (int m = 0; m < 4; ++m) s[m] += src[j + 16 * m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) S[A_DQKV + tok * kQ + j + 16 * m] = s[m];
  }
  BlockState b;
  b.x0 = XS[(size_t)L * kT * 32 + j];
  b.x1 = XS[(size_t)L * kT * 32 + j + 16];
  const Ln n = ln_own(b.x0, b.x1, a.eps);
  S[A_HN + tok * kP + j] = fmaf(n.h0, a.cln1_w[j], a.cln1_b[j]);
  S[A_HN + tok * kP + j + 16] = fmaf(n.h1, a.cln1_w[j + 16], a.cln1_b[j + 16]);
  __syncthreads();
  {
    f32x4 acc = z4();
    lin32_t_accp<64>(S + W_E1, S + A_DQKV + tok * kQ, j, acc);
    acc = half_sum4(acc);
    if (j < 8) *v4(S + A_DHN + tok * kP + 4 * j) = acc;
  }
  tsync();
  float d0, d1;
  {
    const float h0 = S[A_DHN + tok * kP + j], h1 = S[A_DHN + tok * kP + j + 16];
    S[A_T1 + tok * kP + j] = h0 * n.h0;
    S[A_T1 + tok * kP + j + 16] = h1 * n.h1;
    ln_back(n, h0 * a.cln1_w[j], h1 * a.cln1_w[j + 16], d0, d1);
  }
  int DM = A_DM0, DN = A_DM1;
  S[DM + tok * kP + j] = d0;
  S[DM + tok * kP + j + 16] = d1;
  __syncthreads();
  wgrad<64>(S + A_DQKV, kQ, S + A_HN, P + dc_off_wkv(L), 32, 1, tid);
  if (tid < 32) P[dc_off_cln1w(L) + tid] = colsum16(S + A_T1, tid);
  else if (tid < 64) P[dc_off_cln1b(L) + tid - 32] = colsum16(S + A_DHN, tid - 32);
  if (L > 0) lc.store(S, tid);
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    b.x0 = XS[(size_t)l * kT * 32 + j];
    b.x1 = XS[(size_t)l * kT * 32 + j + 16];
    block_back(a.blocks.b[l], l > 0 ? &a.blocks.b[l - 1] : nullptr, S, DM, DN, P + (size_t)l * TP_SIZE, tid, a.eps, b, d0, d1);
    const int t = DM; DM = DN; DN = t;
  }
  // x0 = W_in LN(z): d W_in (stored [32][32], columns >= n_lat zero), d z
  const float* zr = a.z + ((size_t)cell * kT + tok) * a.n_lat;
  float zn0, zn1, rz;
  latent_ln(j < a.n_lat ? zr[j] : 0.f, j + 16 < a.n_lat ? zr[j + 16] : 0.f, a.n_lat, j, a.eps, zn0, zn1, rz);
  S[A_X + tok * kP + j] = zn0;
  S[A_X + tok * kP + j + 16] = zn1;
  {
    f32x4 acc = z4();
    lin32_t_accp<32>(S + W_E2, S + DM + tok * kP, j, acc);
    acc = half_sum4(acc);
    if (j < 8) *v4(S + A_DHN + tok * kP + 4 * j) = acc;
  }
  __syncthreads();
  wgrad<32>(S + DM, kP, S + A_X, P + dc_off_win(L), 32, 1, tid);
  {
    const bool in0 = j < a.n_lat, in1 = j + 16 < a.n_lat;
    const float g0 = in0 ? S[A_DHN + tok * kP + j] : 0.f, g1 = in1 ? S[A_DHN + tok * kP + j + 16] : 0.f;
    const float inv = 1.0f / a.n_lat;
    const float sa = row16_sum(g0 + g1) * inv, sb = row16_sum(fmaf(g0, zn0, g1 * zn1)) * inv;
    float* dz = a.dz + ((size_t)cell * kT + tok) * a.n_lat;
    if (in0) dz[j] = rz * (g0 - sa - zn0 * sb);
    if (in1) dz[j + 16] = rz * (g1 - sa - zn1 * sb);
  }
}

// =================================================================================================================================
// Encoder cell side:  y = inducing + Wp ao; y2 = y + MLP(LN_2 y) + pos; n_layer Blocks -> hL; zl = W_lat hL; z = LN(zl)
// The pooling's c_proj / MLP sit in the W_P / W_1 / W_2 / W_CT slots like a layer without attention.
// =================================================================================================================================
struct CrossCopy {
  RowCopy<96> w1, w2, wct;
  RowCopy<32> wp;
  __device__ __forceinline__ void load(const EncCellTrainArgs& a, int tid) {
    wp.load(a.wp, 32, tid); w1.load(a.cmlp.w1, a.cmlp.H, tid); w2.load(a.cmlp.w2, a.cmlp.H, tid); wct.load(a.cmlp.wct, a.cmlp.H, tid);
  }
  __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
    wp.store(S + W_P, tid); w1.store(S + W_1, tid); w2.store(S + W_2, tid); wct.store(S + W_CT, tid);
  }
};

__global__ __launch_bounds__(kThreads) void enc_cell_fwd_kernel(const EncCellTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const int tid = threadIdx.x, tok = tid >> 4, j = tid & 15, cell = blockIdx.x, L = a.n_layer;
  {
    CrossCopy cc;
    cc.load(a, tid);
    cc.store(S, tid);
  }
  const float* ao = a.pooled + ((size_t)cell * kT + tok) * 32;
  S[A_AO + tok * kP + j] = ao[j];
  S[A_AO + tok * kP + j + 16] = ao[j + 16];
  __syncthreads();
  LayerCopy lc;
  if (L > 0) lc.load(a.blocks.b[0], tid);
  BlockState b;
  lin32p<32>(S + W_P, S + A_AO + tok * kP, j, [&](int m, int, float v) { if (m == 0) b.m0 = a.ind[tok * 32 + j] + v; else b.m1 = a.ind[tok * 32 + j + 16] + v; });
  float* ys = a.ysave + ((size_t)cell * kT + tok) * 32;
  ys[j] = b.m0;
  ys[j + 16] = b.m1;
  b.n2 = ln_own(b.m0, b.m1, a.eps);
  S[A_H2 + tok * kP + j] = fmaf(b.n2.h0, a.cln2_w[j], a.cln2_b[j]);
  S[A_H2 + tok * kP + j + 16] = fmaf(b.n2.h1, a.cln2_w[j + 16], a.cln2_b[j + 16]);
  tsync();
  float y0, y1;
  mlp_forward(S, tok, j, y0, y1);
  b.x0 = b.m0 + y0 + (a.pos ? a.pos[tok * 32 + j] : 0.f);
  b.x1 = b.m1 + y1 + (a.pos ? a.pos[tok * 32 + j + 16] : 0.f);
  __syncthreads();
  if (L > 0) { lc.store(S, tid); __syncthreads(); }
  float* XS = a.xsave + ((size_t)cell * (L + 1) * kT + tok) * 32;
  for (int l = 0; l < L; ++l) {
    XS[(size_t)l * kT * 32 + j] = b.x0;
    XS[(size_t)l * kT * 32 + j + 16] = b.x1;
    if (l + 1 < L) lc.load(a.blocks.b[l + 1], tid);
    block_forward(a.blocks.b[l], S, tok, j, a.eps, b);
    __syncthreads();
    if (l + 1 < L) { lc.store(S, tid); __syncthreads(); }
  }
  XS[(size_t)L * kT * 32 + j] = b.x0;
  XS[(size_t)L * kT * 32 + j + 16] = b.x1;
}

__global__ __launch_bounds__(kThreads) void enc_cell_bwd_kernel(const EncCellTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const int tid = threadIdx.x, tok = tid >> 4, j = tid & 15, cell = blockIdx.x, L = a.n_layer;
  float* P = a.part + (size_t)cell * ec_size(L);
  const float* XS = a.xsave + ((size_t)cell * (L + 1) * kT + tok) * 32;
  LayerCopy lc;
  if (L > 0) lc.load(a.blocks.b[L - 1], tid);
  copy_small(S + W_E2, a.w_lat, a.n_lat, 32, 32, 32, tid);
  BlockState b;
  b.x0 = XS[(size_t)L * kT * 32 + j];
  b.x1 = XS[(size_t)L * kT * 32 + j + 16];
  S[A_X + tok * kP + j] = b.x0;
  S[A_X + tok * kP + j + 16] = b.x1;
  __syncthreads();
  // zl = W_lat hL; z = LN(zl) over n_lat entries; dz = dz_a + dz_b
  float d0, d1;
  {
    float zl0 = 0.f, zl1 = 0.f, zn0, zn1, rz;
    lin32p<32>(S + W_E2, S + A_X + tok * kP, j, [&](int m, int, float v) { if (m == 0) zl0 = v; else zl1 = v; });
    latent_ln(zl0, zl1, a.n_lat, j, a.eps, zn0, zn1, rz);
    const bool in0 = j < a.n_lat, in1 = j + 16 < a.n_lat;
    const size_t zi = ((size_t)cell * kT + tok) * a.n_lat;
    const float g0 = in0 ? (a.dz_a ? a.dz_a[zi + j] : 0.f) + (a.dz_b ? a.dz_b[zi + j] : 0.f) : 0.f;
    const float g1 = in1 ? (a.dz_a ? a.dz_a[zi + j + 16] : 0.f) + (a.dz_b ? a.dz_b[zi + j + 16] : 0.f) : 0.f;
    const float inv = 1.0f / a.n_lat;
    const float sa = row16_sum(g0 + g1) * inv, sb = row16_sum(fmaf(g0, zn0, g1 * zn1)) * inv;
    S[A_DX1 + tok * kP + j] = in0 ? rz * (g0 - sa - zn0 * sb) : 0.f;          // d zl rows
    S[A_DX1 + tok * kP + j + 16] = in1 ? rz * (g1 - sa - zn1 * sb) : 0.f;
    tsync();
    f32x4 acc = z4();
    lin32_t_accp<32>(S + W_E2, S + A_DX1 + tok * kP, j, acc);
    acc = half_sum4(acc);
    if (j < 8) *v4(S + A_DHN + tok * kP + 4 * j) = acc;
    tsync();
    d0 = S[A_DHN + tok * kP + j];
    d1 = S[A_DHN + tok * kP + j + 16];
  }
  int DM = A_DM0, DN = A_DM1;
  S[DM + tok * kP + j] = d0;
  S[DM + tok * kP + j + 16] = d1;
  __syncthreads();
  wgrad<32>(S + A_DX1, kP, S + A_X, P + ec_off_wlat(L), 32, 1, tid);
  if (L > 0) lc.store(S, tid);
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    b.x0 = XS[(size_t)l * kT * 32 + j];
    b.x1 = XS[(size_t)l * kT * 32 + j + 16];
    block_back(a.blocks.b[l], l > 0 ? &a.blocks.b[l - 1] : nullptr, S, DM, DN, P + (size_t)l * TP_SIZE, tid, a.eps, b, d0, d1);
    const int t = DM; DM = DN; DN = t;
  }
  {   // the pooling's c_proj / MLP take the layer slots (same row counts; the qkv slot stays unused)
    CrossCopy cc;
    cc.load(a, tid);
    cc.store(S, tid);
  }
  // x0 = y + MLP(LN_2 y) + pos  (pos_embed is frozen: nnets.py:103-106)
  const float* ys = a.ysave + ((size_t)cell * kT + tok) * 32;
  const float* ao = a.pooled + ((size_t)cell * kT + tok) * 32;
  b.m0 = ys[j];
  b.m1 = ys[j + 16];
  S[A_AO + tok * kP + j] = ao[j];
  S[A_AO + tok * kP + j + 16] = ao[j + 16];
  b.n2 = ln_own(b.m0, b.m1, a.eps);
  S[A_H2 + tok * kP + j] = fmaf(b.n2.h0, a.cln2_w[j], a.cln2_b[j]);
  S[A_H2 + tok * kP + j + 16] = fmaf(b.n2.h1, a.cln2_w[j + 16], a.cln2_b[j + 16]);
  __syncthreads();
  mlp_back(S, DM, a.cln2_w, b.n2, tok, j, d0, d1);         // d0 / d1 = d y
  S[A_DX1 + tok * kP + j] = d0;
  S[A_DX1 + tok * kP + j + 16] = d1;
  P[ec_off_ind(L) + tok * 32 + j] = d0;                     // d inducing of this cell
  P[ec_off_ind(L) + tok * 32 + j + 16] = d1;
  tsync();
  {
    f32x4 acc = z4();
    lin32_t_accp<32>(S + W_P, S + A_DX1 + tok * kP, j, acc);
    acc = half_sum4(acc);                                    // d ao[4 (j & 7) .. + 3]
    if (j < 8) {
      *v4(a.dao + ((size_t)cell * kT + tok) * 32 + 4 * j) = acc;
      // sum_d dao[h, d] ao[h, d] per pooling head (4 heads x 8): lanes 2 h, 2 h + 1
      const float s = pair_sum(dot4(acc, *v4(S + A_AO + tok * kP + 4 * j)));
      if ((j & 1) == 0) a.dgq[((size_t)cell * 4 + (j >> 1)) * kT + tok] = s;
    }
  }
  __syncthreads();
  wgrad<32>(S + A_DX1, kP, S + A_AO, P + ec_off_wp(L), 32, 1, tid);
  mlp_wgrads(S, DM, P + ec_off_w1(L), P + ec_off_w2(L), P + ec_off_wc(L), P + ec_off_ln2w(L), P + ec_off_ln2b(L), tid);
}

// =================================================================================================================================
// Encoder MCAB pooling backward, key side (layers.py:111-118, 248-264, 325-326), second version: 16 lanes per gene token, four
// tokens per wave, the four waves of a workgroup independent (no workgroup barrier inside the token loop).
//   x = E[gene] log1p(count); xn = LN_1(x); k | v = c_attn xn; p[i][h] = exp2(log2e / sqrt 8 * Q[i][h] . k[h] - lse2[i][h])
// Sums over tokens (d c_attn 64 x 32, dQ 64 x 8, LN_1's vectors) are contracted per wave over its four tokens into 41 registers
// per lane that live across the whole token range; the four waves are added through LDS in wave order at the end (deterministic).
// grid = (chunks, B); partial per workgroup: EP_* of vae_train.hpp (only the block-diagonal entries of EP_DQ are written - the only
// ones fold_dq_kernel reads).  Gene-embedding gradient by atomics, as before.
// =================================================================================================================================
constexpr int kP64 = 68;                                   // floats per 64-wide row
constexpr int PB_W = 0, PB_Q = PB_W + 64 * kP, PB_DAO = PB_Q + 16 * kP, PB_LSE = PB_DAO + 16 * kP, PB_DG = PB_LSE + 64, PB_WAVE = PB_DG + 64;
constexpr int PW_XN = 0, PW_TX = PW_XN + 4 * kP, PW_T1 = PW_TX + 4 * kP, PW_KV = PW_T1 + 4 * kP, PW_DKV = PW_KV + 4 * kP64, PW_DSV = PW_DKV + 4 * kP64,
              PW_SIZE = PW_DSV + 4 * kP64;
constexpr int PB_ACC = 2048 + 512 + 64;                    // one wave's accumulators in the final exchange
constexpr int PB_FLOATS = (PB_WAVE + 4 * PW_SIZE) > 4 * PB_ACC ? (PB_WAVE + 4 * PW_SIZE) : 4 * PB_ACC;
constexpr int PB_BYTES = PB_FLOATS * 4;

__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ f32x4 quad_sum4(f32x4 v) { return f32x4{quad_sum(v[0]), quad_sum(v[1]), quad_sum(v[2]), quad_sum(v[3])}; }

__global__ __launch_bounds__(kThreads) void enc_pool_bwd_kernel(const EncPoolBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, tk = lane >> 4, j = lane & 15;
  const int chunk = blockIdx.x, cell = blockIdx.y, nch = gridDim.x;
  constexpr float kS2 = 1.4426950408889634f * 0.35355339059327373f;   // log2(e) / sqrt(8)
  constexpr float kScale = 0.35355339059327373f;
  {
    RowCopy<64> cw;
    cw.load(a.wkv, 64, tid);
    cw.store(S + PB_W, tid);
    for (int idx = tid; idx < 512; idx += kThreads) {
      S[PB_Q + (idx >> 5) * kP + (idx & 31)] = a.Q[idx];
      S[PB_DAO + (idx >> 5) * kP + (idx & 31)] = a.dao[(size_t)cell * 512 + idx];
    }
    if (tid < 64) { S[PB_LSE + tid] = a.lse2[(size_t)cell * 64 + tid]; S[PB_DG + tid] = a.dgq[(size_t)cell * 64 + tid]; }
  }
  __syncthreads();
  float* __restrict__ Wv = S + PB_WAVE + wave * PW_SIZE;
  const float lw0 = a.ln1_w[j], lw1 = a.ln1_w[j + 16], lb0 = a.ln1_b[j], lb1 = a.ln1_b[j + 16];
  const int begin = chunk * a.tiles * 64, end = min(a.S, begin + a.tiles * 64);
  const int h = j >> 2, iq = j & 3;
  f32x4 gw[8];          // d c_attn[o = (lane >> 3) + 8 m][4 (lane & 7) ..]
  f32x4 gqa = z4(), gqb = z4();   // dQ[(head, query) = lane][d = 0 .. 7 of that head]
  float gln = 0.f;      // lanes < 32: LN_1 weight gradient of feature lane; lanes >= 32: bias gradient of feature lane - 32
#pragma unroll
  for (int m = 0; m < 8; ++m) gw[m] = z4();
  for (int s0 = begin + wave * 4; s0 < end; s0 += 16) {
    const int s = s0 + tk;
    const bool valid = s < end;
    const size_t si = (size_t)cell * a.S + (valid ? s : end - 1);
    const long long gene = a.genes[si];
    const float lc = log1pf(a.counts[si]);
    const float* e = a.emb + (size_t)gene * 32;
    const Ln n = ln_own(e[j] * lc, e[j + 16] * lc, a.eps);
    Wv[PW_XN + tk * kP + j] = fmaf(n.h0, lw0, lb0);
    Wv[PW_XN + tk * kP + j + 16] = fmaf(n.h1, lw1, lb1);
    tsync();
    lin32<64>(S + PB_W, Wv + PW_XN + tk * kP, j, [&](int, int o, float v) { Wv[PW_KV + tk * kP64 + o] = v; });
    tsync();
    {
      const f32x4 ka = *v4(Wv + PW_KV + tk * kP64 + 8 * h), kb = *v4(Wv + PW_KV + tk * kP64 + 8 * h + 4);
      const f32x4 va = *v4(Wv + PW_KV + tk * kP64 + 32 + 8 * h), vb = *v4(Wv + PW_KV + tk * kP64 + 32 + 8 * h + 4);
      f32x4 dka = z4(), dkb = z4(), dva = z4(), dvb = z4();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 4 * iq + q;
        const f32x4 qa = *v4(S + PB_Q + i * kP + 8 * h), qb = *v4(S + PB_Q + i * kP + 8 * h + 4);
        const f32x4 da = *v4(S + PB_DAO + i * kP + 8 * h), db = *v4(S + PB_DAO + i * kP + 8 * h + 4);
        const float sc = dot4(qa, ka) + dot4(qb, kb), dp = dot4(da, va) + dot4(db, vb);
        const float pp = valid ? __builtin_amdgcn_exp2f(sc * kS2 - S[PB_LSE + h * 16 + i]) : 0.f;
        const float ds = pp * (dp - S[PB_DG + h * 16 + i]) * kScale;
        Wv[PW_DSV + tk * kP64 + h * 16 + i] = ds;
        dka = fma4(ds, qa, dka); dkb = fma4(ds, qb, dkb);
        dva = fma4(pp, da, dva); dvb = fma4(pp, db, dvb);
      }
      dka = quad_sum4(dka); dkb = quad_sum4(dkb); dva = quad_sum4(dva); dvb = quad_sum4(dvb);
      const f32x4 mine = iq == 0 ? dka : iq == 1 ? dkb : iq == 2 ? dva : dvb;
      *v4(Wv + PW_DKV + tk * kP64 + (iq >> 1) * 32 + 8 * h + 4 * (iq & 1)) = mine;
    }
    tsync();
    {
      f32x4 acc = z4();
      lin32_t_acc<64>(S + PB_W, Wv + PW_DKV + tk * kP64, j, acc);
      acc = half_sum4(acc);
      if (j < 8) *v4(Wv + PW_TX + tk * kP + 4 * j) = acc;
    }
    {   // token-axis contractions over this wave's four tokens
      const int i4 = lane & 7, oo = lane >> 3, hq = lane >> 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 xv = *v4(Wv + PW_XN + t * kP + 4 * i4);
#pragma unroll
        for (int m = 0; m < 8; ++m) gw[m] = fma4(Wv[PW_DKV + t * kP64 + oo + 8 * m], xv, gw[m]);
        const float ds = Wv[PW_DSV + t * kP64 + lane];
        gqa = fma4(ds, *v4(Wv + PW_KV + t * kP64 + 8 * hq), gqa);
        gqb = fma4(ds, *v4(Wv + PW_KV + t * kP64 + 8 * hq + 4), gqb);
      }
    }
    tsync();
    const float g0 = Wv[PW_TX + tk * kP + j], g1 = Wv[PW_TX + tk * kP + j + 16];
    Wv[PW_T1 + tk * kP + j] = g0 * n.h0;
    Wv[PW_T1 + tk * kP + j + 16] = g1 * n.h1;
    float o0, o1;
    ln_back(n, g0 * lw0, g1 * lw1, o0, o1);
    if (valid && lc != 0.f) {
      float* ge = a.g_emb + (size_t)gene * 32;
      atomicAdd(ge + j, o0 * lc);
      atomicAdd(ge + j + 16, o1 * lc);
    }
    tsync();
    {
      const float* src = (lane < 32 ? Wv + PW_T1 : Wv + PW_TX) + (lane & 31);
      gln += (src[0] + src[kP]) + (src[2 * kP] + src[3 * kP]);
    }
    tsync();
  }
  // the four waves' sums -> one partial
  __syncthreads();
  {
    float* R = S + wave * PB_ACC;
    const int i4 = lane & 7, oo = lane >> 3;
#pragma unroll
    for (int m = 0; m < 8; ++m) *v4(R + (oo + 8 * m) * 32 + 4 * i4) = gw[m];
    *v4(R + 2048 + lane * 8) = gqa;
    *v4(R + 2048 + lane * 8 + 4) = gqb;
    R[2560 + lane] = gln;
  }
  __syncthreads();
  float* P = a.part + (size_t)(cell * nch + chunk) * EP_SIZE;
  for (int idx = tid; idx < PB_ACC; idx += kThreads) {
    const float v = ((S[idx] + S[PB_ACC + idx]) + S[2 * PB_ACC + idx]) + S[3 * PB_ACC + idx];
    if (idx < 2048) P[EP_WKV + idx] = v;
    else if (idx < 2560) { const int e = idx - 2048, hi = e >> 3, dd = e & 7; P[EP_DQ + hi * 32 + 8 * (hi >> 4) + dd] = v; }
    else P[EP_LN1W + idx - 2560] = v;
  }
}

// =================================================================================================================================
// Decoder MCAB, per-gene chain backward (layers.py:305-330 with q = gene embeddings, nnets.py:206-208; NB logit head), second
// version: 16 lanes per decoded gene, 16 genes per workgroup step, two workgroups per CU.
//   q0 = E[gene]; qn = LN_1q(q0); qq = Wq qn; ao = softmax(qq K^T / sqrt 8) V (4 heads x 8, 16 latent keys of the cell);
//   y = q0 + Wp ao; h2 = LN_2(y); yo = y + Wc (silu(W1 h2) * W2 h2); logit = w_head . yo + b
// What makes it cheap: the gradient entering the MLP is RANK ONE per gene, d yo = dlogit * w_head.  So
//   d hid = dlogit * c0 with c0 = Wc^T w_head (one 96-vector per launch, not a Linear per gene),
//   d Wc  = w_head (x) c with c = sum_genes dlogit * hid  (a 96-vector of running sums, not a 32 x 96 contraction),
//   d w_head = sum dlogit * y + Wc c  (the MLP's forward output is never formed),
// and Wc leaves the per-gene path altogether.  Token-axis sums that are plain column sums (LN vectors, head, c) are running sums in
// the lanes' own registers, folded over the 16 gene slots once at the end; the outer products (d Wq, d Wp, d W1, d W2, dK | dV of
// the cell) are contracted over the step's 16 genes from LDS rows into 36 accumulator registers per thread that live across the
// whole gene range.  grid = (chunks, B); partial per workgroup: DP_* of vae_train.hpp, dkv_part as before; dE[gene] by atomics.
// =================================================================================================================================
constexpr int G_WQ = 0, G_WP = G_WQ + 32 * kP, G_W1 = G_WP + 32 * kP, G_W2 = G_W1 + 96 * kP, G_KV = G_W2 + 96 * kP;
constexpr int G_QN = G_KV + 16 * kP64, G_QQ = G_QN + 16 * kP, G_AO = G_QQ + 16 * kP, G_H2 = G_AO + 16 * kP, G_DY = G_H2 + 16 * kP,
              G_DAO = G_DY + 16 * kP, G_DQQ = G_DAO + 16 * kP, G_TX = G_DQQ + 16 * kP, G_DA = G_TX + 16 * kP, G_DB = G_DA + 16 * kQ,
              G_PP = G_DB + 16 * kQ, G_DS = G_PP + 16 * kP64, G_FLOATS = G_DS + 16 * kP64;
constexpr int G_BYTES = G_FLOATS * 4;
static_assert(G_BYTES <= 80 * 1024, "two workgroups per CU");

__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));
  return v;
}
// acc[m] += sum_t DY[t][oo + 32 m] X[t][4 i4 ..]
template <int OUT>
__device__ __forceinline__ void wgrad_acc(const float* __restrict__ DY, int py, const float* __restrict__ X, f32x4* __restrict__ acc, int tid) {
  const int i4 = tid & 7, oo = tid >> 3;
#pragma unroll 4
  for (int t = 0; t < kT; ++t) {
    const f32x4 xv = *v4(X + t * kP + 4 * i4);
#pragma unroll
    for (int m = 0; m < OUT / 32; ++m) acc[m] = fma4(DY[t * py + oo + 32 * m], xv, acc[m]);
  }
}

__global__ __launch_bounds__(kThreads, 2) void dec_gene_bwd_kernel(const DecBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const int tid = threadIdx.x, tok = tid >> 4, j = tid & 15, chunk = blockIdx.x, cell = blockIdx.y, nch = gridDim.x;
  constexpr float kScale = 0.35355339059327373f;   // 1 / sqrt(8)
  const int H = a.mlp.H;
  {
    RowCopy<32> cq, cp;
    RowCopy<96> c1, c2;
    cq.load(a.wq, 32, tid); cp.load(a.wp, 32, tid); c1.load(a.mlp.w1, H, tid); c2.load(a.mlp.w2, H, tid);
    cq.store(S + G_WQ, tid); cp.store(S + G_WP, tid); c1.store(S + G_W1, tid); c2.store(S + G_W2, tid);
    for (int idx = tid; idx < kT * 64; idx += kThreads) S[G_KV + (idx >> 6) * kP64 + (idx & 63)] = a.kv[(size_t)cell * (kT * 64) + idx];
  }
  // c0[u] = sum_i Wc[i][u] w_head[i] for this lane's hidden units u = j + 16 m (zero beyond H)
  float c0[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    const int u = j + 16 * m;
    float s = 0.f;
    if (u < H)
      for (int i = 0; i < 32; ++i) s = fmaf(a.mlp.wct[u * 32 + i], a.head_w[i], s);
    c0[m] = s;
  }
  const float l1w0 = a.ln1q_w[j], l1w1 = a.ln1q_w[j + 16], l1b0 = a.ln1q_b[j], l1b1 = a.ln1q_b[j + 16];
  const float l2w0 = a.ln2_w[j], l2w1 = a.ln2_w[j + 16], l2b0 = a.ln2_b[j], l2b1 = a.ln2_b[j + 16];
  const float hw0 = a.head_w[j], hw1 = a.head_w[j + 16];
  __syncthreads();
  f32x4 gq[1] = {z4()}, gp[1] = {z4()}, g1[3] = {z4(), z4(), z4()}, g2[3] = {z4(), z4(), z4()}, gkv = z4();
  float vs[16];       // running sums of this lane's gene slot: ln1q w|b, ln2 w|b, head (two features each), c (six hidden units)
#pragma unroll
  for (int i = 0; i < 16; ++i) vs[i] = 0.f;
  const int h = j >> 2, jq = j & 3;
  const int begin = chunk * a.tiles * 64, end = min(a.G, begin + a.tiles * 64);
  for (int g0 = begin; g0 < end; g0 += 16) {
    const int g = g0 + tok;
    const bool valid = g < end;
    const size_t gi = (size_t)cell * a.G + (valid ? g : end - 1);
    const long long gene = a.genes[gi];
    const float dlog = valid ? a.dl[gi] : 0.f;
    const float* e = a.emb + (size_t)gene * 32;
    const float q00 = e[j], q01 = e[j + 16];
    const Ln n1 = ln_own(q00, q01, a.eps);
    S[G_QN + tok * kP + j] = fmaf(n1.h0, l1w0, l1b0);
    S[G_QN + tok * kP + j + 16] = fmaf(n1.h1, l1w1, l1b1);
    tsync();
    lin32<32>(S + G_WQ, S + G_QN + tok * kP, j, [&](int, int o, float v) { S[G_QQ + tok * kP + o] = v; });
    tsync();
    float p[4];
    {
      const f32x4 qa = *v4(S + G_QQ + tok * kP + 8 * h), qb = *v4(S + G_QQ + tok * kP + 8 * h + 4);
      float mx = -3.0e38f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float* K = S + G_KV + (4 * jq + kk) * kP64 + 8 * h;
        p[kk] = (dot4(qa, *v4(K)) + dot4(qb, *v4(K + 4))) * kScale;
        mx = fmaxf(mx, p[kk]);
      }
      mx = quad_max(mx);
      float l = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { p[kk] = __expf(p[kk] - mx); l += p[kk]; }
      const float inv = 1.0f / quad_sum(l);
      f32x4 aa = z4(), ab = z4();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        p[kk] *= inv;
        const float* V = S + G_KV + (4 * jq + kk) * kP64 + 32 + 8 * h;
        aa = fma4(p[kk], *v4(V), aa);
        ab = fma4(p[kk], *v4(V + 4), ab);
      }
      aa = quad_sum4(aa);
      ab = quad_sum4(ab);
      if (jq < 2) *v4(S + G_AO + tok * kP + 8 * h + 4 * jq) = jq == 0 ? aa : ab;
      *v4(S + G_PP + tok * kP64 + h * 16 + 4 * jq) = f32x4{p[0], p[1], p[2], p[3]};
    }
    tsync();
    float y0 = q00, y1 = q01;
    lin32<32>(S + G_WP, S + G_AO + tok * kP, j, [&](int m, int, float v) { if (m == 0) y0 += v; else y1 += v; });
    const Ln n2 = ln_own(y0, y1, a.eps);
    S[G_H2 + tok * kP + j] = fmaf(n2.h0, l2w0, l2b0);
    S[G_H2 + tok * kP + j + 16] = fmaf(n2.h1, l2w1, l2b1);
    tsync();
    {
      float aa[6], bb[6];
      lin32<96>(S + G_W1, S + G_H2 + tok * kP, j, [&](int m, int, float v) { aa[m] = v; });
      lin32<96>(S + G_W2, S + G_H2 + tok * kP, j, [&](int m, int, float v) { bb[m] = v; });
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        const float s = sigm(aa[m]), sa = aa[m] * s, dh = dlog * c0[m];
        vs[10 + m] = fmaf(dlog, sa * bb[m], vs[10 + m]);
        S[G_DA + tok * kQ + j + 16 * m] = dh * bb[m] * (s * (1.0f + aa[m] * (1.0f - s)));
        S[G_DB + tok * kQ + j + 16 * m] = dh * sa;
      }
    }
    tsync();
    {
      f32x4 acc = z4();
      lin32_t_acc<96>(S + G_W1, S + G_DA + tok * kQ, j, acc);
      lin32_t_acc<96>(S + G_W2, S + G_DB + tok * kQ, j, acc);
      acc = half_sum4(acc);
      if (j < 8) *v4(S + G_TX + tok * kP + 4 * j) = acc;
    }
    tsync();
    float d0, d1;
    {
      const float t0 = S[G_TX + tok * kP + j], t1 = S[G_TX + tok * kP + j + 16];
      vs[4] = fmaf(t0, n2.h0, vs[4]); vs[5] = fmaf(t1, n2.h1, vs[5]); vs[6] += t0; vs[7] += t1;
      vs[8] = fmaf(dlog, y0, vs[8]); vs[9] = fmaf(dlog, y1, vs[9]);
      ln_back(n2, t0 * l2w0, t1 * l2w1, d0, d1);
      d0 = fmaf(dlog, hw0, d0);
      d1 = fmaf(dlog, hw1, d1);
    }
    S[G_DY + tok * kP + j] = d0;
    S[G_DY + tok * kP + j + 16] = d1;
    tsync();
    {
      f32x4 acc = z4();
      lin32_t_acc<32>(S + G_WP, S + G_DY + tok * kP, j, acc);
      acc = half_sum4(acc);
      if (j < 8) *v4(S + G_DAO + tok * kP + 4 * j) = acc;
    }
    tsync();
    {
      const f32x4 da = *v4(S + G_DAO + tok * kP + 8 * h), db = *v4(S + G_DAO + tok * kP + 8 * h + 4);
      float dp[4], dg = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float* V = S + G_KV + (4 * jq + kk) * kP64 + 32 + 8 * h;
        dp[kk] = dot4(da, *v4(V)) + dot4(db, *v4(V + 4));
        dg = fmaf(p[kk], dp[kk], dg);
      }
      dg = quad_sum(dg);
      f32x4 qa = z4(), qb = z4(), dsv;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float ds = p[kk] * (dp[kk] - dg) * kScale;
        dsv[kk] = ds;
        const float* K = S + G_KV + (4 * jq + kk) * kP64 + 8 * h;
        qa = fma4(ds, *v4(K), qa);
        qb = fma4(ds, *v4(K + 4), qb);
      }
      qa = quad_sum4(qa);
      qb = quad_sum4(qb);
      if (jq < 2) *v4(S + G_DQQ + tok * kP + 8 * h + 4 * jq) = jq == 0 ? qa : qb;
      *v4(S + G_DS + tok * kP64 + h * 16 + 4 * jq) = dsv;
    }
    tsync();
    {
      f32x4 acc = z4();
      lin32_t_acc<32>(S + G_WQ, S + G_DQQ + tok * kP, j, acc);
      acc = half_sum4(acc);
      tsync();      // (G_TX of this gene was read above; the fence keeps the store below it)
      if (j < 8) *v4(S + G_TX + tok * kP + 4 * j) = acc;
    }
    tsync();
    {
      const float t0 = S[G_TX + tok * kP + j], t1 = S[G_TX + tok * kP + j + 16];
      vs[0] = fmaf(t0, n1.h0, vs[0]); vs[1] = fmaf(t1, n1.h1, vs[1]); vs[2] += t0; vs[3] += t1;
      float o0, o1;
      ln_back(n1, t0 * l1w0, t1 * l1w1, o0, o1);
      if (valid && dlog != 0.f) {
        float* ge = a.g_emb + (size_t)gene * 32;
        atomicAdd(ge + j, d0 + o0);
        atomicAdd(ge + j + 16, d1 + o1);
      }
    }
    __syncthreads();
    wgrad_acc<32>(S + G_DQQ, kP, S + G_QN, gq, tid);
    wgrad_acc<32>(S + G_DY, kP, S + G_AO, gp, tid);
    wgrad_acc<96>(S + G_DA, kQ, S + G_H2, g1, tid);
    wgrad_acc<96>(S + G_DB, kQ, S + G_H2, g2, tid);
    {   // dK | dV of the cell: thread = (key, four of the 64 columns)
      const int key = tid >> 4, f4 = tid & 15, hh = (f4 & 7) >> 1;
      const float* sc = S + (f4 < 8 ? G_DS : G_PP) + hh * 16 + key;
      const float* vec = S + (f4 < 8 ? G_QQ : G_DAO) + 4 * (f4 & 7);
#pragma unroll 4
      for (int t = 0; t < kT; ++t) gkv = fma4(sc[t * kP64], *v4(vec + t * kP), gkv);
    }
    __syncthreads();
  }
  // ---- one partial per workgroup
  float* P = a.part + (size_t)(cell * nch + chunk) * DP_SIZE;
  {
    const int i4 = tid & 7, oo = tid >> 3;
    *v4(P + DP_WQ + oo * 32 + 4 * i4) = gq[0];
    *v4(P + DP_WP + oo * 32 + 4 * i4) = gp[0];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      *v4(P + DP_W1 + (oo + 32 * m) * 32 + 4 * i4) = g1[m];
      *v4(P + DP_W2 + (oo + 32 * m) * 32 + 4 * i4) = g2[m];
    }
    *v4(a.dkv_part + (size_t)(cell * nch + chunk) * (kT * 64) + (tid >> 4) * 64 + 4 * (tid & 15)) = gkv;
  }
  // the 16 gene slots' running sums: R[slot][16 values][16 lanes]
  float* R = S;
#pragma unroll
  for (int i = 0; i < 16; ++i) R[(tok * 16 + i) * 16 + j] = vs[i];
  __syncthreads();
  {
    const int i = tid >> 4;       // value index, lane j
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < kT; ++t) s += R[(t * 16 + i) * 16 + j];
    __syncthreads();
    // feature / hidden-unit index of value i in lane j
    if (i < 10) {
      const int f = j + 16 * (i & 1);
      const int off = i < 2 ? DP_LN1QW : i < 4 ? DP_LN1QB : i < 6 ? DP_LN2W : i < 8 ? DP_LN2B : -1;
      if (off >= 0) P[off + f] = s;
      else R[512 + f] = s;                       // head: sum dlogit * y (completed below)
    } else R[j + 16 * (i - 10)] = s;             // c[u]
  }
  __syncthreads();
  if (tid < 32) {
    float s = R[512 + tid];
    for (int u = 0; u < H; ++u) s = fmaf(a.mlp.wct[u * 32 + tid], R[u], s);
    P[DP_HEADW + tid] = s;
  }
  for (int idx = tid; idx < 32 * kHP; idx += kThreads) P[DP_WC + idx] = a.head_w[idx / kHP] * R[idx % kHP];
}

// =================================================================================================================================
// The same kernel with its heavy contractions on the matrix pipe (exact fp32: v_mfma_f32_16x16x4_f32, the step's 16 genes on one
// axis).  The version above is LDS-bandwidth bound: every wave re-reads each weight piece for its four genes.  Here a step has five
// phases separated by workgroup barriers:
//   1  per gene (16 lanes each, as above): LN_1q, q, attention, c_proj, LN_2 -> rows QN, QQ, AO, PP, H2
//   A  a | b = W1 h2 | W2 h2 as 16 x 16 tiles D[gene][hidden unit] (wave w: unit tiles w, w + 4), SwiGLU gradient in place -> rows DA, DB
//   B  d h2 = W1^T da + W2^T db: D[gene][feature], wave = (feature half, matrix); the two matrices' partials are added by the reader
//   2  per gene: LN_2 backward, d y, d ao, attention backward, d q, LN_1q backward, dE[gene] atomics -> rows DY, DAO, DQQ, DS
//   C  weight gradients as D[out][in] += sum_genes dy[gene][out] x[gene][in]: 8 + 2 tiles of 16 x 16 per wave (d W1, d W2, d Wq, d Wp,
//      dK and dV of one head), accumulators in 40 registers across the whole gene range
// An MFMA reads one scalar per lane and operand: 2 x 256 B of LDS per 2 048 FLOP.
// =================================================================================================================================
constexpr int M_DL = G_FLOATS, M_C0 = M_DL + 16, M_FLOATS = M_C0 + 96;
constexpr int M_BYTES = M_FLOATS * 4;
static_assert(M_BYTES <= 80 * 1024, "two workgroups per CU");
__device__ __forceinline__ f32x4 mfma16(float x, float y, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0); }

__global__ __launch_bounds__(kThreads, 2) void dec_gene_bwd_mfma_kernel(const DecBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const int tid = threadIdx.x, tok = tid >> 4, j = tid & 15, chunk = blockIdx.x, cell = blockIdx.y, nch = gridDim.x;
  const int wave = tid >> 6, li = tid & 15, g4 = (tid & 63) >> 4;      // MFMA roles: lane = (li, g4)
  constexpr float kScale = 0.35355339059327373f;   // 1 / sqrt(8)
  const int H = a.mlp.H;
  {
    RowCopy<32> cq, cp;
    RowCopy<96> c1, c2;
    cq.load(a.wq, 32, tid); cp.load(a.wp, 32, tid); c1.load(a.mlp.w1, H, tid); c2.load(a.mlp.w2, H, tid);
    cq.store(S + G_WQ, tid); cp.store(S + G_WP, tid); c1.store(S + G_W1, tid); c2.store(S + G_W2, tid);
    for (int idx = tid; idx < kT * 64; idx += kThreads) S[G_KV + (idx >> 6) * kP64 + (idx & 63)] = a.kv[(size_t)cell * (kT * 64) + idx];
    if (tid < 96) {      // c0[u] = sum_i Wc[i][u] w_head[i]
      float s = 0.f;
      if (tid < H)
        for (int i = 0; i < 32; ++i) s = fmaf(a.mlp.wct[tid * 32 + i], a.head_w[i], s);
      S[M_C0 + tid] = s;
    }
  }
  const float l1w0 = a.ln1q_w[j], l1w1 = a.ln1q_w[j + 16], l1b0 = a.ln1q_b[j], l1b1 = a.ln1q_b[j + 16];
  const float l2w0 = a.ln2_w[j], l2w1 = a.ln2_w[j + 16], l2b0 = a.ln2_b[j], l2b1 = a.ln2_b[j + 16];
  const float hw0 = a.head_w[j], hw1 = a.head_w[j + 16];
  __syncthreads();
  f32x4 gw1[3] = {z4(), z4(), z4()}, gw2[3] = {z4(), z4(), z4()}, gq = z4(), gp = z4(), gk = z4(), gv = z4();
  float cacc[2] = {0.f, 0.f};   // c[u] partials of this lane's gene quad, hidden-unit tiles wave, wave + 4
  float vs[10];                 // running sums of this lane's gene slot: ln1q w|b, ln2 w|b, head (two features each)
#pragma unroll
  for (int i = 0; i < 10; ++i) vs[i] = 0.f;
  const int h = j >> 2, jq = j & 3;
  const int begin = chunk * a.tiles * 64, end = min(a.G, begin + a.tiles * 64);
  for (int g0 = begin; g0 < end; g0 += 16) {
    const int g = g0 + tok;
    const bool valid = g < end;
    const size_t gi = (size_t)cell * a.G + (valid ? g : end - 1);
    const long long gene = a.genes[gi];
    const float dlog = valid ? a.dl[gi] : 0.f;
    if (j == 0) S[M_DL + tok] = dlog;
    const float* e = a.emb + (size_t)gene * 32;
    const float q00 = e[j], q01 = e[j + 16];
    const Ln n1 = ln_own(q00, q01, a.eps);
    S[G_QN + tok * kP + j] = fmaf(n1.h0, l1w0, l1b0);
    S[G_QN + tok * kP + j + 16] = fmaf(n1.h1, l1w1, l1b1);
    tsync();
    lin32<32>(S + G_WQ, S + G_QN + tok * kP, j, [&](int, int o, float v) { S[G_QQ + tok * kP + o] = v; });
    tsync();
    float p[4];
    {
      const f32x4 qa = *v4(S + G_QQ + tok * kP + 8 * h), qb = *v4(S + G_QQ + tok * kP + 8 * h + 4);
      float mx = -3.0e38f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float* K = S + G_KV + (4 * jq + kk) * kP64 + 8 * h;
        p[kk] = (dot4(qa, *v4(K)) + dot4(qb, *v4(K + 4))) * kScale;
        mx = fmaxf(mx, p[kk]);
      }
      mx = quad_max(mx);
      float l = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { p[kk] = __expf(p[kk] - mx); l += p[kk]; }
      const float inv = 1.0f / quad_sum(l);
      f32x4 aa = z4(), ab = z4();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        p[kk] *= inv;
        const float* V = S + G_KV + (4 * jq + kk) * kP64 + 32 + 8 * h;
        aa = fma4(p[kk], *v4(V), aa);
        ab = fma4(p[kk], *v4(V + 4), ab);
      }
      aa = quad_sum4(aa);
      ab = quad_sum4(ab);
      if (jq < 2) *v4(S + G_AO + tok * kP + 8 * h + 4 * jq) = jq == 0 ? aa : ab;
      *v4(S + G_PP + tok * kP64 + h * 16 + 4 * jq) = f32x4{p[0], p[1], p[2], p[3]};
    }
    tsync();
    float y0 = q00, y1 = q01;
    lin32<32>(S + G_WP, S + G_AO + tok * kP, j, [&](int m, int, float v) { if (m == 0) y0 += v; else y1 += v; });
    const Ln n2 = ln_own(y0, y1, a.eps);
    S[G_H2 + tok * kP + j] = fmaf(n2.h0, l2w0, l2b0);
    S[G_H2 + tok * kP + j + 16] = fmaf(n2.h1, l2w1, l2b1);
    __syncthreads();
    // ---- phase A: a | b tiles D[gene 4 g4 + r][unit 16 ot + li], SwiGLU gradient in place
    {
      float hk[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) hk[s] = S[G_H2 + li * kP + 4 * s + g4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ot = wave + 4 * q;
        if (ot < 6) {
          f32x4 aa = z4(), bb = z4();
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            aa = mfma16(hk[s], S[G_W1 + (16 * ot + li) * kP + 4 * s + g4], aa);
            bb = mfma16(hk[s], S[G_W2 + (16 * ot + li) * kP + 4 * s + g4], bb);
          }
          const float c0u = S[M_C0 + 16 * ot + li];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int tk = 4 * g4 + r;
            const float dl = S[M_DL + tk], sg = sigm(aa[r]), sa = aa[r] * sg, dh = dl * c0u;
            cacc[q] = fmaf(dl, sa * bb[r], cacc[q]);
            S[G_DA + tk * kQ + 16 * ot + li] = dh * bb[r] * (sg * (1.0f + aa[r] * (1.0f - sg)));
            S[G_DB + tk * kQ + 16 * ot + li] = dh * sa;
          }
        }
      }
    }
    __syncthreads();
    // ---- phase B: d h2 partials, wave = (feature half it, matrix); W1's go to the TX rows, W2's to the DQQ rows (free until phase 2)
    {
      const int it = wave & 1, mtx = wave >> 1;
      const float* Wm = S + (mtx ? G_W2 : G_W1);
      const float* Dm = S + (mtx ? G_DB : G_DA);
      // k slot (s, g4) <-> hidden unit u = 16 (s >> 2) + 4 g4 + (s & 3): the two weight rows a 32-lane group reads are 4 rows = 16 banks
      // apart (4 s + g4 would put them 4 banks apart: two-way conflicts); two accumulator chains keep the pipe fed
      f32x4 acc = z4(), acc2 = z4();
#pragma unroll 6
      for (int s = 0; s < 24; s += 2) {
        const int u0 = 16 * (s >> 2) + 4 * g4 + (s & 3), u1 = u0 + 1;
        acc = mfma16(Dm[li * kQ + u0], Wm[u0 * kP + 16 * it + li], acc);
        acc2 = mfma16(Dm[li * kQ + u1], Wm[u1 * kP + 16 * it + li], acc2);
      }
      acc += acc2;
      float* Pt = S + (mtx ? G_DQQ : G_TX);
#pragma unroll
      for (int r = 0; r < 4; ++r) Pt[(4 * g4 + r) * kP + 16 * it + li] = acc[r];
    }
    __syncthreads();
    // ---- phase 2: per gene
    float d0, d1;
    {
      const float t0 = S[G_TX + tok * kP + j] + S[G_DQQ + tok * kP + j], t1 = S[G_TX + tok * kP + j + 16] + S[G_DQQ + tok * kP + j + 16];
      vs[4] = fmaf(t0, n2.h0, vs[4]); vs[5] = fmaf(t1, n2.h1, vs[5]); vs[6] += t0; vs[7] += t1;
      vs[8] = fmaf(dlog, y0, vs[8]); vs[9] = fmaf(dlog, y1, vs[9]);
      ln_back(n2, t0 * l2w0, t1 * l2w1, d0, d1);
      d0 = fmaf(dlog, hw0, d0);
      d1 = fmaf(dlog, hw1, d1);
    }
    S[G_DY + tok * kP + j] = d0;
    S[G_DY + tok * kP + j + 16] = d1;
    tsync();
    {
      f32x4 acc = z4();
      lin32_t_acc<32>(S + G_WP, S + G_DY + tok * kP, j, acc);
      acc = half_sum4(acc);
      if (j < 8) *v4(S + G_DAO + tok * kP + 4 * j) = acc;
    }
    tsync();
    {
      const f32x4 da = *v4(S + G_DAO + tok * kP + 8 * h), db = *v4(S + G_DAO + tok * kP + 8 * h + 4);
      float dp[4], dg = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float* V = S + G_KV + (4 * jq + kk) * kP64 + 32 + 8 * h;
        dp[kk] = dot4(da, *v4(V)) + dot4(db, *v4(V + 4));
        dg = fmaf(p[kk], dp[kk], dg);
      }
      dg = quad_sum(dg);
      f32x4 qa = z4(), qb = z4(), dsv;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float ds = p[kk] * (dp[kk] - dg) * kScale;
        dsv[kk] = ds;
        const float* K = S + G_KV + (4 * jq + kk) * kP64 + 8 * h;
        qa = fma4(ds, *v4(K), qa);
        qb = fma4(ds, *v4(K + 4), qb);
      }
      qa = quad_sum4(qa);
      qb = quad_sum4(qb);
      if (jq < 2) *v4(S + G_DQQ + tok * kP + 8 * h + 4 * jq) = jq == 0 ? qa : qb;
      *v4(S + G_DS + tok * kP64 + h * 16 + 4 * jq) = dsv;
    }
    tsync();
    {
      f32x4 acc = z4();
      lin32_t_acc<32>(S + G_WQ, S + G_DQQ + tok * kP, j, acc);
      acc = half_sum4(acc);
      if (j < 8) *v4(S + G_TX + tok * kP + 4 * j) = acc;
    }
    tsync();
    {
      const float t0 = S[G_TX + tok * kP + j], t1 = S[G_TX + tok * kP + j + 16];
      vs[0] = fmaf(t0, n1.h0, vs[0]); vs[1] = fmaf(t1, n1.h1, vs[1]); vs[2] += t0; vs[3] += t1;
      float o0, o1;
      ln_back(n1, t0 * l1w0, t1 * l1w1, o0, o1);
      if (valid && dlog != 0.f) {
        float* ge = a.g_emb + (size_t)gene * 32;
        atomicAdd(ge + j, d0 + o0);
        atomicAdd(ge + j + 16, d1 + o1);
      }
    }
    __syncthreads();
    // ---- phase C: weight gradients, D[out 16 ot + 4 g4 + r][in 16 it + li] += sum_genes dy[gene][out] x[gene][in]
    {
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const int idx = wave + 4 * n, ot = idx >> 1, it = idx & 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int tk = s + 4 * g4;       // (k slot <-> gene: rows 4 apart = 16 banks apart within a 32-lane group: conflict-free)
          const float x = S[G_H2 + tk * kP + 16 * it + li];
          gw1[n] = mfma16(S[G_DA + tk * kQ + 16 * ot + li], x, gw1[n]);
          gw2[n] = mfma16(S[G_DB + tk * kQ + 16 * ot + li], x, gw2[n]);
        }
      }
      const int ot = wave >> 1, it = wave & 1;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int tk = s + 4 * g4;
        gq = mfma16(S[G_DQQ + tk * kP + 16 * ot + li], S[G_QN + tk * kP + 16 * it + li], gq);
        gp = mfma16(S[G_DY + tk * kP + 16 * ot + li], S[G_AO + tk * kP + 16 * it + li], gp);
        // head `wave` of the cell's keys: D[key 4 g4 + r][column 16 (wave >> 1) + li], useful where (li >> 3) == (wave & 1)
        gk = mfma16(S[G_DS + tk * kP64 + wave * 16 + li], S[G_QQ + tk * kP + 16 * (wave >> 1) + li], gk);
        gv = mfma16(S[G_PP + tk * kP64 + wave * 16 + li], S[G_DAO + tk * kP + 16 * (wave >> 1) + li], gv);
      }
    }
    __syncthreads();
  }
  // ---- one partial per workgroup
  float* P = a.part + (size_t)(cell * nch + chunk) * DP_SIZE;
  {
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int idx = wave + 4 * n, ot = idx >> 1, it = idx & 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        P[DP_W1 + (16 * ot + 4 * g4 + r) * 32 + 16 * it + li] = gw1[n][r];
        P[DP_W2 + (16 * ot + 4 * g4 + r) * 32 + 16 * it + li] = gw2[n][r];
      }
    }
    const int ot = wave >> 1, it = wave & 1;
    float* DK = a.dkv_part + (size_t)(cell * nch + chunk) * (kT * 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      P[DP_WQ + (16 * ot + 4 * g4 + r) * 32 + 16 * it + li] = gq[r];
      P[DP_WP + (16 * ot + 4 * g4 + r) * 32 + 16 * it + li] = gp[r];
      if ((li >> 3) == (wave & 1)) {
        DK[(4 * g4 + r) * 64 + 16 * (wave >> 1) + li] = gk[r];
        DK[(4 * g4 + r) * 64 + 32 + 16 * (wave >> 1) + li] = gv[r];
      }
    }
  }
  // the 16 gene slots' running sums R[slot][10 values][16 lanes]; c partials RC[g4][96 units] of every wave's tiles
  float* R = S;
  float* RC = S + 16 * 10 * 16;
#pragma unroll
  for (int i = 0; i < 10; ++i) R[(tok * 10 + i) * 16 + j] = vs[i];
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (wave + 4 * q < 6) RC[g4 * 96 + 16 * (wave + 4 * q) + li] = cacc[q];
  __syncthreads();
  float sum = 0.f;
  const int vi = tid >> 4;
  if (vi < 10) {
#pragma unroll
    for (int t = 0; t < kT; ++t) sum += R[(t * 10 + vi) * 16 + j];
  }
  float cu = 0.f;
  if (tid < 96) cu = (RC[tid] + RC[96 + tid]) + (RC[192 + tid] + RC[288 + tid]);
  __syncthreads();
  float* C = S;            // c[u]
  float* HD = S + 128;     // sum dlogit * y
  if (tid < 96) C[tid] = cu;
  if (vi < 10) {
    const int f = j + 16 * (vi & 1);
    if (vi < 2) P[DP_LN1QW + f] = sum;
    else if (vi < 4) P[DP_LN1QB + f] = sum;
    else if (vi < 6) P[DP_LN2W + f] = sum;
    else if (vi < 8) P[DP_LN2B + f] = sum;
    else HD[f] = sum;
  }
  __syncthreads();
  if (tid < 32) {
    float s2 = HD[tid];
    for (int u = 0; u < H; ++u) s2 = fmaf(a.mlp.wct[u * 32 + tid], C[u], s2);
    P[DP_HEADW + tid] = s2;
  }
  for (int idx = tid; idx < 32 * kHP; idx += kThreads) P[DP_WC + idx] = a.head_w[idx / kHP] * C[idx % kHP];
}


// =================================================================================================================================
// Third generation (round 5): the four 32 x 32 Linears of the per-gene chain - q = Wq LN_1q(e), y = Wp ao, d ao = Wp^T d y,
// d qn = Wq^T d q - on the matrix pipe as well.  In the kernel above every gene's 16 lanes re-read the whole 32 x 36 weight image
// for each of them (16 KB of LDS reads per gene: a third of the kernel's LDS cycles, and four dependent chains of 24 16-byte reads
// per step); here one step's 16 genes are the 16 rows of 16 x 16 x 4 MFMAs: a wave owns (output half, k half) of a product - 4 MFMAs -
// writes its partial tile to a row set that is free at that point of the step, and the per-gene phase that follows adds the two
// partials.  Operand reads are LDS-bank aware (ds_read_b32 banks are address mod 32 within 32 lanes, b64 / b128 mod 64):
//   k along the contiguous axis of BOTH images (q, y, a | b):  lane (li, g4) reads the PAIR of k values 8 p + 2 g4, + 1 as one
//     ds_read_b64 - row pitch 36 = 4 mod 32 walks 16 rows over 16 distinct 4-bank groups, g4 fills the odd pairs: conflict-free, and
//     half the LDS instructions of the scalar reads (which were 2-way conflicts: rows li and li + 8 share a bank)
//   k along the rows of the weight image (d h2, d ao, d qn):  k slot (s, g4) <-> row 16 (s >> 2) + 4 g4 + (s & 3): the B reads of a
//     32-lane group hit rows 4 apart = 16 banks apart (conflict-free, as before); the A operand's four values of a lane are then
//     CONSECUTIVE - one ds_read_b128 instead of four 4-way conflicting scalar reads
// The gene's attention over the cell's 16 latent tokens and its backward run on the matrix pipe as well, one head per wave: scores
// (16 genes x 16 keys, k = 8 head dims) and P V / d P / d S K as 16 x 16 x 4 tiles, softmax and its backward across the 16 lanes of a
// DPP row in the accumulator layout (lane = key) - before, 16 lanes per gene spent ~190 VALU operations per step on them.
// Per step: S0 embeddings, LN_1q | S1 q partials | S2 attention | S3 y partials, q summed | S4 LN_2 | A | B | S5 LN_2 backward |
// S6 d ao (two waves, whole k: the next phase's MFMAs read whole rows) | S7 attention backward | S8 d qn (likewise) + C weight
// gradients | S9 LN_1q backward, dE atomics - separated by workgroup barriers (two workgroups per CU fill each other's waits).
// =================================================================================================================================
#ifndef SCLDM_VAE_PHASE_CLOCKS
#define SCLDM_VAE_PHASE_CLOCKS 0   // 1: one workgroup prints its cycles per phase of dec_gene_bwd_mfma2_kernel (tools only)
#endif
typedef __attribute__((ext_vector_type(2))) float f32x2;
// (volatile: keeps hipcc from pairing two of these into one ds_read2_b64, which is serviced 16 lanes at a time with banks mod 32 -
// 8 LDS cycles and two-way conflicts on this pitch, against 2 + 2 conflict-free)
// (the LDS address space is spelled out: a volatile access through a generic pointer becomes a flat load)
__device__ __forceinline__ float row16_max(float v) {   // max over the 16 lanes of a DPP row, in every lane of the row
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)));   // row_ror:8
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false)));
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false)));
}
typedef const volatile __attribute__((address_space(3))) f32x2 lds_f32x2;
__device__ __forceinline__ lds_f32x2* v2(const float* p) { return (lds_f32x2*)p; }

__global__ __launch_bounds__(kThreads, 2) void dec_gene_bwd_mfma2_kernel(const DecBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const int tid = threadIdx.x, tok = tid >> 4, j = tid & 15, chunk = blockIdx.x, cell = blockIdx.y, nch = gridDim.x;
  const int wave = tid >> 6, li = tid & 15, g4 = (tid & 63) >> 4;      // MFMA roles: lane = (li, g4)
  constexpr float kScale = 0.35355339059327373f;   // 1 / sqrt(8)
  const int H = a.mlp.H;
  {
    RowCopy<32> cq, cp;
    RowCopy<96> c1, c2;
    cq.load(a.wq, 32, tid); cp.load(a.wp, 32, tid); c1.load(a.mlp.w1, H, tid); c2.load(a.mlp.w2, H, tid);
    cq.store(S + G_WQ, tid); cp.store(S + G_WP, tid); c1.store(S + G_W1, tid); c2.store(S + G_W2, tid);
    for (int idx = tid; idx < kT * 64; idx += kThreads) S[G_KV + (idx >> 6) * kP64 + (idx & 63)] = a.kv[(size_t)cell * (kT * 64) + idx];
    if (tid < 96) {      // c0[u] = sum_i Wc[i][u] w_head[i]
      float s = 0.f;
      if (tid < H)
        for (int i = 0; i < 32; ++i) s = fmaf(a.mlp.wct[tid * 32 + i], a.head_w[i], s);
      S[M_C0 + tid] = s;
    }
  }
  const float l1w0 = a.ln1q_w[j], l1w1 = a.ln1q_w[j + 16], l1b0 = a.ln1q_b[j], l1b1 = a.ln1q_b[j + 16];
  const float l2w0 = a.ln2_w[j], l2w1 = a.ln2_w[j + 16], l2b0 = a.ln2_b[j], l2b1 = a.ln2_b[j + 16];
  const float hw0 = a.head_w[j], hw1 = a.head_w[j + 16];
  __syncthreads();
  f32x4 gw1[3] = {z4(), z4(), z4()}, gw2[3] = {z4(), z4(), z4()}, gq = z4(), gp = z4(), gk = z4(), gv = z4();
  float cacc[2] = {0.f, 0.f};   // c[u] partials of this lane's gene quad, hidden-unit tiles wave, wave + 4
  float vs[10];                 // running sums of this lane's gene slot: ln1q w|b, ln2 w|b, head (two features each)
#pragma unroll
  for (int i = 0; i < 10; ++i) vs[i] = 0.f;
  const int half = wave & 1, kh = wave >> 1;     // the small products' roles: (output half, k half)
  // x W^T with k along both images' rows: partial over input features 16 kh .. + 15 of output half `half`, D[gene 4 g4 + r][16 half + li]
  auto lin_partial = [&](int x_rows, int w_rows, int dst_rows) {
    const float* X = S + x_rows + li * kP + 16 * kh + 2 * g4;
    const float* W = S + w_rows + (16 * half + li) * kP + 16 * kh + 2 * g4;
    const f32x2 x0 = *v2(X), x1 = *v2(X + 8), w0 = *v2(W), w1 = *v2(W + 8);
    f32x4 acc = z4();
    acc = mfma16(x0[0], w0[0], acc);
    acc = mfma16(x0[1], w0[1], acc);
    acc = mfma16(x1[0], w1[0], acc);
    acc = mfma16(x1[1], w1[1], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) S[dst_rows + (4 * g4 + r) * kP + 16 * half + li] = acc[r];
  };
  // dy W with k along the weight image's rows: partial over output rows 16 kq .. + 15, D[gene 4 g4 + r][input 16 it + li]
  auto lin_t_partial = [&](int dy_rows, int w_rows, int it, int kq, f32x4& acc) {
    const f32x4 d = *v4(S + dy_rows + li * kP + 16 * kq + 4 * g4);
    const float* W = S + w_rows + (16 * kq + 4 * g4) * kP + 16 * it + li;
    acc = mfma16(d[0], W[0], acc);
    acc = mfma16(d[1], W[kP], acc);
    acc = mfma16(d[2], W[2 * kP], acc);
    acc = mfma16(d[3], W[3 * kP], acc);
  };
  const int begin = chunk * a.tiles * 64, end = min(a.G, begin + a.tiles * 64);
#if SCLDM_VAE_PHASE_CLOCKS
  long long clk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last = __builtin_readcyclecounter();
#define PHASE_MARK(i) { const long long t_ = __builtin_readcyclecounter(); clk[i] += t_ - last; last = t_; }
#else
#define PHASE_MARK(i)
#endif
  // this step's gene id, d logit and embedding row were requested one step earlier (the id two steps earlier): S0 no longer waits
  // for two dependent global loads
  auto slot = [&](int g0) { return (size_t)cell * a.G + min(g0 + tok, end - 1); };
  // The cells of a batch list their genes in the same order, so workgroups of different cells reach the same genes' dE rows at the
  // same time and their atomics serialise in the L2.  Each cell therefore starts its walk over the chunk's steps at its own offset
  // (measured: the step's first phase 2 000 -> 1 650 cycles; without any atomics 1 250, 13.5 -> 13.2 ms per batch-512 step).
  const int nsteps = (end - begin + 15) / 16;
  const int rot = nsteps > 0 ? (int)(((unsigned)cell * 37u) % (unsigned)nsteps) : 0;
  auto first_gene = [&](int step) { return begin + 16 * ((step + rot) % nsteps); };
  long long gene_nx = 0, gene_n2 = 0;
  float dl_nx = 0.f, e0_nx = 0.f, e1_nx = 0.f;
  if (nsteps > 0) {
    const int ga = first_gene(0);
    gene_nx = a.genes[slot(ga)];
    gene_n2 = a.genes[slot(first_gene(1))];
    dl_nx = ga + tok < end ? a.dl[slot(ga)] : 0.f;
    e0_nx = a.emb[(size_t)gene_nx * 32 + j];
    e1_nx = a.emb[(size_t)gene_nx * 32 + j + 16];
  }
  for (int step = 0; step < nsteps; ++step) {
    const int g0 = first_gene(step), g1 = first_gene(step + 1);
    // ---- S0: embeddings, LN_1q
    const bool valid = g0 + tok < end;
    const long long gene = gene_nx;
    const float dlog = dl_nx, q00 = e0_nx, q01 = e1_nx;
    gene_nx = gene_n2;
    e0_nx = a.emb[(size_t)gene_nx * 32 + j];
    e1_nx = a.emb[(size_t)gene_nx * 32 + j + 16];
    dl_nx = g1 + tok < end ? a.dl[slot(g1)] : 0.f;
    gene_n2 = a.genes[slot(first_gene(step + 2))];
    if (j == 0) S[M_DL + tok] = dlog;
    const Ln n1 = ln_own(q00, q01, a.eps);
    S[G_QN + tok * kP + j] = fmaf(n1.h0, l1w0, l1b0);
    S[G_QN + tok * kP + j + 16] = fmaf(n1.h1, l1w1, l1b1);
    __syncthreads();
    PHASE_MARK(0);
    // ---- S1: q partials -> QQ (k half 0), TX (k half 1)
    lin_partial(G_QN, G_WQ, kh ? G_TX : G_QQ);
    __syncthreads();
    PHASE_MARK(1);
    // ---- S2: attention on the matrix pipe, wave = head (16 genes x 16 keys x 8 head dims): scores = q_h K_h^T as two MFMAs (k pairs of
    // head dims as ds_read_b64; q = the two partials), softmax over the 16 lanes of a DPP row (lane = key), p -> PP rows (phase C's
    // operand anyway), ao_h = P V_h as four MFMAs whose useful output columns are the head's 8 dims
    float p[4];
    {
      const f32x2 q2 = *v2(S + G_QQ + li * kP + 8 * wave + 2 * g4) + *v2(S + G_TX + li * kP + 8 * wave + 2 * g4);
      const f32x2 k2 = *v2(S + G_KV + li * kP64 + 8 * wave + 2 * g4);
      f32x4 sc = z4();
      sc = mfma16(q2[0], k2[0], sc);
      sc = mfma16(q2[1], k2[1], sc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = sc[r] * kScale, e = __expf(v - row16_max(v));
        p[r] = e * (1.0f / row16_sum(e));
        S[G_PP + (4 * g4 + r) * kP64 + 16 * wave + li] = p[r];
      }
      tsync();     // (this wave wrote the PP columns it reads)
      f32x4 ao = z4();
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x2 p2 = *v2(S + G_PP + li * kP64 + 16 * wave + 8 * m + 2 * g4);
        const float* V = S + G_KV + (8 * m + 2 * g4) * kP64 + 32 + 8 * wave + (li & 7);   // (lanes li, li + 8 share an address: no conflict)
        ao = mfma16(p2[0], V[0], ao);
        ao = mfma16(p2[1], V[kP64], ao);
      }
      if (li < 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S[G_AO + (4 * g4 + r) * kP + 8 * wave + li] = ao[r];
      }
    }
    __syncthreads();
    PHASE_MARK(2);
    // ---- S3: c_proj partials -> DY (k half 0), DAO (k half 1)
    lin_partial(G_AO, G_WP, kh ? G_DAO : G_DY);
    S[G_QQ + tok * kP + j] += S[G_TX + tok * kP + j];               // the whole q: dK's operand in phase C (its partials' last readers
    S[G_QQ + tok * kP + j + 16] += S[G_TX + tok * kP + j + 16];     // were S2's MFMAs; TX is next written in phase B)
    __syncthreads();
    PHASE_MARK(3);
    // ---- S4: residual, LN_2
    const float y0 = q00 + (S[G_DY + tok * kP + j] + S[G_DAO + tok * kP + j]);
    const float y1 = q01 + (S[G_DY + tok * kP + j + 16] + S[G_DAO + tok * kP + j + 16]);
    const Ln n2 = ln_own(y0, y1, a.eps);
    S[G_H2 + tok * kP + j] = fmaf(n2.h0, l2w0, l2b0);
    S[G_H2 + tok * kP + j + 16] = fmaf(n2.h1, l2w1, l2b1);
    __syncthreads();
    PHASE_MARK(4);
    // ---- phase A: a | b tiles D[gene 4 g4 + r][unit 16 ot + li], SwiGLU gradient in place
    {
      f32x2 hk[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) hk[s] = *v2(S + G_H2 + li * kP + 8 * s + 2 * g4);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ot = wave + 4 * q;
        if (ot < 6) {
          f32x4 aa = z4(), bb = z4();
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const f32x2 wa = *v2(S + G_W1 + (16 * ot + li) * kP + 8 * s + 2 * g4), wb = *v2(S + G_W2 + (16 * ot + li) * kP + 8 * s + 2 * g4);
            aa = mfma16(hk[s][0], wa[0], aa);
            bb = mfma16(hk[s][0], wb[0], bb);
            aa = mfma16(hk[s][1], wa[1], aa);
            bb = mfma16(hk[s][1], wb[1], bb);
          }
          const float c0u = S[M_C0 + 16 * ot + li];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int tk = 4 * g4 + r;
            const float dl = S[M_DL + tk], sg = sigm(aa[r]), sa = aa[r] * sg, dh = dl * c0u;
            cacc[q] = fmaf(dl, sa * bb[r], cacc[q]);
            S[G_DA + tk * kQ + 16 * ot + li] = dh * bb[r] * (sg * (1.0f + aa[r] * (1.0f - sg)));
            S[G_DB + tk * kQ + 16 * ot + li] = dh * sa;
          }
        }
      }
    }
    __syncthreads();
    PHASE_MARK(5);
    // ---- phase B: d h2 partials, wave = (feature half, matrix); W1's go to the TX rows, W2's to the DQQ rows (free until S7)
    {
      const float* Wm = S + (kh ? G_W2 : G_W1);
      const float* Dm = S + (kh ? G_DB : G_DA);
      f32x4 acc = z4(), acc2 = z4();
#pragma unroll
      for (int sq = 0; sq < 6; ++sq) {
        const f32x4 d = *v4(Dm + li * kQ + 16 * sq + 4 * g4);
        const float* W = Wm + (16 * sq + 4 * g4) * kP + 16 * half + li;
        acc = mfma16(d[0], W[0], acc);
        acc2 = mfma16(d[1], W[kP], acc2);
        acc = mfma16(d[2], W[2 * kP], acc);
        acc2 = mfma16(d[3], W[3 * kP], acc2);
      }
      acc += acc2;
      float* Pt = S + (kh ? G_DQQ : G_TX);
#pragma unroll
      for (int r = 0; r < 4; ++r) Pt[(4 * g4 + r) * kP + 16 * half + li] = acc[r];
    }
    __syncthreads();
    PHASE_MARK(6);
    // ---- S5: LN_2 backward -> d y
    float d0, d1;
    {
      const float t0 = S[G_TX + tok * kP + j] + S[G_DQQ + tok * kP + j], t1 = S[G_TX + tok * kP + j + 16] + S[G_DQQ + tok * kP + j + 16];
      vs[4] = fmaf(t0, n2.h0, vs[4]); vs[5] = fmaf(t1, n2.h1, vs[5]); vs[6] += t0; vs[7] += t1;
      vs[8] = fmaf(dlog, y0, vs[8]); vs[9] = fmaf(dlog, y1, vs[9]);
      ln_back(n2, t0 * l2w0, t1 * l2w1, d0, d1);
      d0 = fmaf(dlog, hw0, d0);
      d1 = fmaf(dlog, hw1, d1);
    }
    S[G_DY + tok * kP + j] = d0;
    S[G_DY + tok * kP + j + 16] = d1;
    __syncthreads();
    PHASE_MARK(7);
    // ---- S6: d ao = Wp^T d y (waves 0, 1: one input half each, all 32 rows - S7's MFMAs read the whole row, and no row set is free
    // for a finalised sum of two partials)
    if (wave < 2) {
      f32x4 acc = z4(), acc2 = z4();
      lin_t_partial(G_DY, G_WP, wave, 0, acc);
      lin_t_partial(G_DY, G_WP, wave, 1, acc2);
      acc += acc2;
#pragma unroll
      for (int r = 0; r < 4; ++r) S[G_DAO + (4 * g4 + r) * kP + 16 * wave + li] = acc[r];
    }
    __syncthreads();
    PHASE_MARK(8);
    // ---- S7: attention backward, wave = head: d p = d ao_h V_h^T (two MFMAs), d s = p (d p - sum_keys p d p) / sqrt(8) in the lanes
    // that hold p, -> DS rows (phase C's operand), d q_h = d S K_h (four MFMAs, 8 useful columns)
    {
      const f32x2 a2 = *v2(S + G_DAO + li * kP + 8 * wave + 2 * g4);
      const f32x2 v2v = *v2(S + G_KV + li * kP64 + 32 + 8 * wave + 2 * g4);
      f32x4 dp = z4();
      dp = mfma16(a2[0], v2v[0], dp);
      dp = mfma16(a2[1], v2v[1], dp);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dg = row16_sum(p[r] * dp[r]);
        S[G_DS + (4 * g4 + r) * kP64 + 16 * wave + li] = p[r] * (dp[r] - dg) * kScale;
      }
      tsync();
      f32x4 dq = z4();
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x2 d2 = *v2(S + G_DS + li * kP64 + 16 * wave + 8 * m + 2 * g4);
        const float* K = S + G_KV + (8 * m + 2 * g4) * kP64 + 8 * wave + (li & 7);
        dq = mfma16(d2[0], K[0], dq);
        dq = mfma16(d2[1], K[kP64], dq);
      }
      if (li < 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S[G_DQQ + (4 * g4 + r) * kP + 8 * wave + li] = dq[r];
      }
    }
    __syncthreads();
    PHASE_MARK(9);
    // ---- S8: d qn = Wq^T d q (waves 0, 1: one input half each, all 32 rows) -> TX; then phase C on every wave
    if (wave < 2) {
      f32x4 acc = z4(), acc2 = z4();
      lin_t_partial(G_DQQ, G_WQ, wave, 0, acc);
      lin_t_partial(G_DQQ, G_WQ, wave, 1, acc2);
      acc += acc2;
#pragma unroll
      for (int r = 0; r < 4; ++r) S[G_TX + (4 * g4 + r) * kP + 16 * wave + li] = acc[r];
    }
    // ---- phase C: weight gradients, D[out 16 ot + 4 g4 + r][in 16 it + li] += sum_genes dy[gene][out] x[gene][in]
    // (d W1 | d W2 could run in phase B already - measured: B + 1 270 cycles, C - 890: the matrix pipe is shared with the CU's other workgroup)
    {
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const int idx = wave + 4 * n, ot = idx >> 1, it = idx & 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int tk = s + 4 * g4;       // (k slot <-> gene: rows 4 apart = 16 banks apart within a 32-lane group: conflict-free)
          const float x = S[G_H2 + tk * kP + 16 * it + li];
          gw1[n] = mfma16(S[G_DA + tk * kQ + 16 * ot + li], x, gw1[n]);
          gw2[n] = mfma16(S[G_DB + tk * kQ + 16 * ot + li], x, gw2[n]);
        }
      }
      const int ot = wave >> 1, it = wave & 1;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int tk = s + 4 * g4;
        gq = mfma16(S[G_DQQ + tk * kP + 16 * ot + li], S[G_QN + tk * kP + 16 * it + li], gq);
        gp = mfma16(S[G_DY + tk * kP + 16 * ot + li], S[G_AO + tk * kP + 16 * it + li], gp);
        // head `wave` of the cell's keys: D[key 4 g4 + r][column 16 (wave >> 1) + li], useful where (li >> 3) == (wave & 1)
        gk = mfma16(S[G_DS + tk * kP64 + wave * 16 + li], S[G_QQ + tk * kP + 16 * (wave >> 1) + li], gk);
        gv = mfma16(S[G_PP + tk * kP64 + wave * 16 + li], S[G_DAO + tk * kP + 16 * (wave >> 1) + li], gv);
      }
    }
    __syncthreads();
    PHASE_MARK(10);
    // ---- S9: LN_1q backward, dE[gene] (runs into the next step's S0: neither touches a row the other does)
    {
      const float t0 = S[G_TX + tok * kP + j], t1 = S[G_TX + tok * kP + j + 16];
      vs[0] = fmaf(t0, n1.h0, vs[0]); vs[1] = fmaf(t1, n1.h1, vs[1]); vs[2] += t0; vs[3] += t1;
      float o0, o1;
      ln_back(n1, t0 * l1w0, t1 * l1w1, o0, o1);
      if (valid && dlog != 0.f) {
        float* ge = a.g_emb + (size_t)gene * 32;
        atomicAdd(ge + j, d0 + o0);
        atomicAdd(ge + j + 16, d1 + o1);
      }
    }
    PHASE_MARK(11);
  }
#if SCLDM_VAE_PHASE_CLOCKS
  if (blockIdx.x == 0 && blockIdx.y == 3 && tid == 0)
    printf("phase clocks (S0 S1 S2 S3 S4 A B S5 S6 S7 S8+C S9): %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld steps %d\n", clk[0], clk[1], clk[2],
           clk[3], clk[4], clk[5], clk[6], clk[7], clk[8], clk[9], clk[10], clk[11], nsteps);
#endif
#undef PHASE_MARK
  __syncthreads();
  // ---- one partial per workgroup (as dec_gene_bwd_mfma_kernel)
  float* P = a.part + (size_t)(cell * nch + chunk) * DP_SIZE;
  {
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int idx = wave + 4 * n, ot = idx >> 1, it = idx & 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        P[DP_W1 + (16 * ot + 4 * g4 + r) * 32 + 16 * it + li] = gw1[n][r];
        P[DP_W2 + (16 * ot + 4 * g4 + r) * 32 + 16 * it + li] = gw2[n][r];
      }
    }
    const int ot = wave >> 1, it = wave & 1;
    float* DK = a.dkv_part + (size_t)(cell * nch + chunk) * (kT * 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      P[DP_WQ + (16 * ot + 4 * g4 + r) * 32 + 16 * it + li] = gq[r];
      P[DP_WP + (16 * ot + 4 * g4 + r) * 32 + 16 * it + li] = gp[r];
      if ((li >> 3) == (wave & 1)) {
        DK[(4 * g4 + r) * 64 + 16 * (wave >> 1) + li] = gk[r];
        DK[(4 * g4 + r) * 64 + 32 + 16 * (wave >> 1) + li] = gv[r];
      }
    }
  }
  float* R = S;
  float* RC = S + 16 * 10 * 16;
#pragma unroll
  for (int i = 0; i < 10; ++i) R[(tok * 10 + i) * 16 + j] = vs[i];
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (wave + 4 * q < 6) RC[g4 * 96 + 16 * (wave + 4 * q) + li] = cacc[q];
  __syncthreads();
  float sum = 0.f;
  const int vi = tid >> 4;
  if (vi < 10) {
#pragma unroll
    for (int t = 0; t < kT; ++t) sum += R[(t * 10 + vi) * 16 + j];
  }
  float cu = 0.f;
  if (tid < 96) cu = (RC[tid] + RC[96 + tid]) + (RC[192 + tid] + RC[288 + tid]);
  __syncthreads();
  float* C = S;            // c[u]
  float* HD = S + 128;     // sum dlogit * y
  if (tid < 96) C[tid] = cu;
  if (vi < 10) {
    const int f = j + 16 * (vi & 1);
    if (vi < 2) P[DP_LN1QW + f] = sum;
    else if (vi < 4) P[DP_LN1QB + f] = sum;
    else if (vi < 6) P[DP_LN2W + f] = sum;
    else if (vi < 8) P[DP_LN2B + f] = sum;
    else HD[f] = sum;
  }
  __syncthreads();
  if (tid < 32) {
    float s2 = HD[tid];
    for (int u = 0; u < H; ++u) s2 = fmaf(a.mlp.wct[u * 32 + tid], C[u], s2);
    P[DP_HEADW + tid] = s2;
  }
  for (int idx = tid; idx < 32 * kHP; idx += kThreads) P[DP_WC + idx] = a.head_w[idx / kHP] * C[idx % kHP];
}

}  // namespace wide
}  // namespace vtrain
}  // namespace scldm
