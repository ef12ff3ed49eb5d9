// Backward of the TransformerVAE training step on gfx950 (judge-added row V1 / BASELINE configs[0]):
//   TransformerVAE.forward                     src/scldm/vae.py:29-56
//   VAE.loss -> -log_nb_positive(counts, mu, theta)   src/scldm/models.py:243-249, src/scldm/distributions.py:6-42
// The reference differentiates that chain with torch autograd; here every gradient is hand-derived and fp32.
//
// Execution model of every kernel in this file: ONE wave (64 threads) per workgroup, ONE token per lane (a gene of the
// decoder's / encoder's gene axis, or one of the 16 latent tokens of a cell - four cells per wave).  A token's 32-wide
// activation vectors live in registers and every Linear is a chain of scalar-operand FMAs: the weights are wave-uniform, so
// hipcc fetches their rows with s_load_dwordx16 through the scalar cache and feeds them to v_fmac as SGPR operands - no LDS
// or VGPR traffic for weights at all.  Sums over TOKENS (every weight gradient, the per-cell dK / dV of the decoder's cross
// attention, the encoder's dQ) are contractions over the lane axis: the two operand vectors of a gradient are staged as
// [token][feature] rows in LDS and contracted by exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32, 32 steps per 64 tokens) into
// accumulator tiles that stay in registers across all the token tiles a workgroup walks; each workgroup then writes ONE
// partial, and a final pass adds the partials in index order (deterministic).  Only the two embedding-table gradients
// (gene_embedding, theta: scatter by gene id) use float atomics, like torch's own embedding backward.
// Forward activations are recomputed from the saved inputs (per-gene chains) or from one saved (16, 32) state per trunk layer.
#pragma once
#include "common.hpp"

namespace scldm {
namespace vtrain {

constexpr int kD = 32;      // n_embed
constexpr int kT = 16;      // latent tokens per cell
constexpr int kSL = 36;     // floats per row of a 32-wide staging tile (16-byte aligned rows, conflict-free 16-byte stores)
constexpr int kHP = 96;     // SwiGLU hidden padded to three 32-wide chunks (88 in the reference)
constexpr int kXL = 84;     // floats per lane row of the attention exchange area of the cell kernels

__device__ __forceinline__ f32x16 z16() {
  f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return z;
}
// single-wave workgroups: the barrier is free, the fences order this wave's LDS traffic for the compiler
__device__ __forceinline__ void wsync() { __syncthreads(); }

// ---- thread-local vector helpers (weights: wave-uniform pointers -> scalar loads) ---------------------------------------------
template <int N>
__device__ __forceinline__ float dotw(const float* __restrict__ w, const float (&x)[N]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) s = fmaf(w[i], x[i], s);
  return s;
}
// y[o] = W[o][:] . x   (W row-major [OUT][IN])
template <int OUT, int IN>
__device__ __forceinline__ void matvec(const float* __restrict__ W, const float (&x)[IN], float (&y)[OUT]) {
#pragma unroll
  for (int o = 0; o < OUT; ++o) y[o] = dotw<IN>(W + o * IN, x);
}
// dx[i] += sum_o W[o][i] dy[o]
template <int OUT, int IN>
__device__ __forceinline__ void matvec_t_acc(const float* __restrict__ W, const float (&dy)[OUT], float (&dx)[IN]) {
#pragma unroll
  for (int o = 0; o < OUT; ++o)
#pragma unroll
    for (int i = 0; i < IN; ++i) dx[i] = fmaf(W[o * IN + i], dy[o], dx[i]);
}
template <int N>
__device__ __forceinline__ void ln_fwd(const float (&x)[N], float (&xhat)[N], float& rstd, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) s += x[i];
  const float mean = s * (1.0f / N);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) { const float d = x[i] - mean; xhat[i] = d; v = fmaf(d, d, v); }
  rstd = 1.0f / sqrtf(v * (1.0f / N) + eps);
#pragma unroll
  for (int i = 0; i < N; ++i) xhat[i] *= rstd;
}
// dx += rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))
template <int N>
__device__ __forceinline__ void ln_bwd_acc(const float (&dxhat)[N], const float (&xhat)[N], float rstd, float (&dx)[N]) {
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) { a += dxhat[i]; b = fmaf(dxhat[i], xhat[i], b); }
  a *= (1.0f / N);
  b *= (1.0f / N);
#pragma unroll
  for (int i = 0; i < N; ++i) dx[i] = fmaf(rstd, dxhat[i] - a - xhat[i] * b, dx[i]);
}
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---- staging + lane-axis contractions ------------------------------------------------------------------------------------------
// row `lane` of a [64][kSL] tile <- v (zeros for an invalid token)
__device__ __forceinline__ void stage32(float* __restrict__ buf, int lane, const float (&v)[32], bool valid) {
  f32x4* row = reinterpret_cast<f32x4*>(buf + lane * kSL);
#pragma unroll
  for (int q = 0; q < 8; ++q) row[q] = valid ? f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]} : f32x4{0.f, 0.f, 0.f, 0.f};
}
// acc[r] (lane l) += sum_tokens A[token][acc_row(r, l >> 5)] * B[token][l & 31]
__device__ __forceinline__ void wgrad32(f32x16& acc, const float* __restrict__ A, const float* __restrict__ B, int lane) {
  const int c = lane & 31, kh = lane >> 5;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(2 * k + kh) * kSL + c], B[(2 * k + kh) * kSL + c], acc, 0, 0, 0);
}
// lanes < 32: column sums of A; lanes >= 32: column sums of B  (feature = lane & 31)
__device__ __forceinline__ float colsum2(const float* __restrict__ A, const float* __restrict__ B, int lane) {
  const float* src = (lane < 32 ? A : B) + (lane & 31);
  float s = 0.f;
#pragma unroll 8
  for (int t = 0; t < 64; ++t) s += src[t * kSL];
  return s;
}
// partial tile store: dst[(row0 + acc_row(r, hh)) * ld + col0 + c32]
__device__ __forceinline__ void flush_tile(float* __restrict__ dst, const f32x16& acc, int lane, int row0, int col0, int ld) {
  const int c = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[(size_t)(row0 + acc_row(r, hh)) * ld + col0 + c] = acc[r];
}

// ---- SwiGLU MLP, forward and backward in ONE streaming pass over the hidden units ------------------------------------------
// m = Wc (silu(W1 h2) * (W2 h2)); given dm: dh2 = W1^T da + W2^T db.  Weight-gradient tiles (3 chunks of 32 hidden units):
// g1[c] += da_c (x) h2, g2[c] += db_c (x) h2, gc[c] += dm (x) hid_c.  WcT = Wc transposed ([H][32]).  bufX / bufD must hold the
// staged h2 / dm tiles; bufS is scratch of THREE tiles (da | db | hid of a chunk).  WANT_M: also return m (the forward value).
struct MlpW { const float* w1; const float* w2; const float* wct; int H; };
// FLUSH (the cell kernels: one token tile per workgroup, nothing to accumulate across tiles): every chunk's three tiles are written
// to the partial at once (P1 / P2: [96][32] row blocks, PC: [32][96] column blocks) instead of living in 144 accumulator registers.
template <bool WANT_M, bool FLUSH = false>
__device__ __forceinline__ void mlp_fwd_bwd(const MlpW w, const float (&h2)[32], const float (&dm)[32], float (&m)[32], float (&dh2)[32],
                                            f32x16 (&g1)[3], f32x16 (&g2)[3], f32x16 (&gc)[3], float* __restrict__ bufX,
                                            float* __restrict__ bufD, float* __restrict__ bufS, int lane, bool valid,
                                            float* __restrict__ P1 = nullptr, float* __restrict__ P2 = nullptr, float* __restrict__ PC = nullptr) {
#pragma unroll
  for (int i = 0; i < 32; ++i) { dh2[i] = 0.f; if (WANT_M) m[i] = 0.f; }
  stage32(bufX, lane, h2, valid);
  stage32(bufD, lane, dm, valid);
  float* __restrict__ S1 = bufS;                 // bufS holds THREE [64][kSL] tiles: da | db | hid of the current chunk
  float* __restrict__ S2 = bufS + 64 * kSL;
  float* __restrict__ S3 = bufS + 2 * 64 * kSL;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    wsync();   // (the previous chunk's MFMAs have read the three tiles)
#pragma unroll 2
    for (int jj = 0; jj < 32; ++jj) {
      const int j = c * 32 + jj;
      float da = 0.f, db = 0.f, hd = 0.f;
      if (j < w.H) {   // wave-uniform
        const float a = dotw<32>(w.w1 + j * 32, h2), b = dotw<32>(w.w2 + j * 32, h2);
        const float s = sigm(a), sa = a * s;
        hd = sa * b;
        const float dh = dotw<32>(w.wct + j * 32, dm);
        da = dh * b * (s * (1.0f + a * (1.0f - s)));
        db = dh * sa;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          dh2[i] = fmaf(w.w1[j * 32 + i], da, fmaf(w.w2[j * 32 + i], db, dh2[i]));
          if (WANT_M) m[i] = fmaf(w.wct[j * 32 + i], hd, m[i]);
        }
      }
      S1[lane * kSL + jj] = valid ? da : 0.f;
      S2[lane * kSL + jj] = valid ? db : 0.f;
      S3[lane * kSL + jj] = valid ? hd : 0.f;
    }
    wsync();
    if constexpr (FLUSH) {
      f32x16 t = z16(); wgrad32(t, S1, bufX, lane); flush_tile(P1, t, lane, 32 * c, 0, 32);
      t = z16(); wgrad32(t, S2, bufX, lane); flush_tile(P2, t, lane, 32 * c, 0, 32);
      t = z16(); wgrad32(t, bufD, S3, lane); flush_tile(PC, t, lane, 0, 32 * c, kHP);
    } else {
      wgrad32(g1[c], S1, bufX, lane);
      wgrad32(g2[c], S2, bufX, lane);
      wgrad32(gc[c], bufD, S3, lane);
    }
  }
  wsync();
}
// forward only
__device__ __forceinline__ void mlp_fwd(const MlpW w, const float (&h2)[32], float (&m)[32]) {
#pragma unroll
  for (int i = 0; i < 32; ++i) m[i] = 0.f;
  for (int j = 0; j < w.H; ++j) {
    const float a = dotw<32>(w.w1 + j * 32, h2), b = dotw<32>(w.w2 + j * 32, h2);
    const float hd = a * sigm(a) * b;
#pragma unroll
    for (int i = 0; i < 32; ++i) m[i] = fmaf(w.wct[j * 32 + i], hd, m[i]);
  }
}

// =================================================================================================================================
// Small preparation kernels
// =================================================================================================================================
// WT[c][r] = W[r][c]
__global__ void transpose_kernel(const float* __restrict__ W, int R, int C, float* __restrict__ WT) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R * C) WT[(i % C) * R + (i / C)] = W[i];
}
// the same for up to 34 matrices of one shape in ONE launch (blockIdx.y = matrix; WT of matrix m at dst + m * R * C... stride given)
struct TransposeJobs { const float* src[34]; };
__global__ void transpose_many_kernel(const TransposeJobs j, int R, int C, float* __restrict__ dst, long stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R * C) dst[(size_t)blockIdx.y * stride + (i % C) * R + (i / C)] = j.src[blockIdx.y][i];
}
// Encoder queries (cell-independent): qn = LN_1q(inducing points), Q = c_attn_q qn  (layers.py:312-313,326; 248-253)
__global__ __launch_bounds__(64) void enc_q_fwd_kernel(const float* __restrict__ ind, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                       const float* __restrict__ wq, float eps, float* __restrict__ Q) {
  const int t = threadIdx.x;
  if (t >= kT) return;
  float x[32], xh[32], q[32], rstd;
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = ind[t * 32 + i];
  ln_fwd<32>(x, xh, rstd, eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) xh[i] = fmaf(xh[i], lnw[i], lnb[i]);
  matvec<32, 32>(wq, xh, q);
#pragma unroll
  for (int i = 0; i < 32; ++i) Q[t * 32 + i] = q[i];
}
// ... and its backward once dQ (sum over all cells and tokens) is known: d c_attn_q, d ln_1q, d inducing points (+= into gind)
__global__ __launch_bounds__(64) void enc_q_bwd_kernel(const float* __restrict__ ind, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                       const float* __restrict__ wq, float eps, const float* __restrict__ dQ,
                                                       float* __restrict__ g_wq, float* __restrict__ g_lnw, float* __restrict__ g_lnb,
                                                       float* __restrict__ g_ind) {
  __shared__ float A[64 * kSL], Bq[64 * kSL], Cn[64 * kSL];
  const int t = threadIdx.x;
  const bool valid = t < kT;
  const int tt = valid ? t : 0;
  float x[32], xh[32], qn[32], dq[32], dqn[32], rstd;
#pragma unroll
  for (int i = 0; i < 32; ++i) { x[i] = ind[tt * 32 + i]; dq[i] = dQ[tt * 32 + i]; dqn[i] = 0.f; }
  ln_fwd<32>(x, xh, rstd, eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) qn[i] = fmaf(xh[i], lnw[i], lnb[i]);
  matvec_t_acc<32, 32>(wq, dq, dqn);
  float dxh[32], prod[32], dx[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { dxh[i] = dqn[i] * lnw[i]; prod[i] = dqn[i] * xh[i]; dx[i] = 0.f; }
  ln_bwd_acc<32>(dxh, xh, rstd, dx);
  if (valid)
#pragma unroll
    for (int i = 0; i < 32; ++i) g_ind[t * 32 + i] += dx[i];
  stage32(A, t, dq, valid);
  stage32(Bq, t, qn, valid);
  wsync();
  f32x16 acc = z16();
  wgrad32(acc, A, Bq, t);
  flush_tile(g_wq, acc, t, 0, 0, 32);
  wsync();
  stage32(A, t, prod, valid);
  stage32(Cn, t, dqn, valid);
  wsync();
  const float s = colsum2(A, Cn, t);
  if (t < 32) g_lnw[t] = s; else g_lnb[t - 32] = s;
}

// =================================================================================================================================
// NB head backward (stochastic_layers.py:102-116): mu = softmax_G(logit / T) lib, theta = exp(Theta[gene])
//   dlogit_g = mu_g (dmu_g - sum_j dmu_j mu_j / lib) / T ;  dTheta[gene] += dtheta theta (atomic scatter)
// One workgroup (256 threads) per cell.  dlogit overwrites `dl`; bsum[cell] = sum_g dlogit_g (the head bias gradient's partial).
// =================================================================================================================================
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ theta, const float* __restrict__ dmu,
                                                       const float* __restrict__ dtheta, const float* __restrict__ lib,
                                                       const int64_t* __restrict__ genes, int G, float inv_temp, float* __restrict__ dl,
                                                       float* __restrict__ g_theta, float* __restrict__ bsum) {
  __shared__ float red[4];
  const int cell = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)cell * G;
  float s = 0.f;
  for (int g = tid; g < G; g += 256) s = fmaf(dmu ? dmu[base + g] : 0.f, mu[base + g], s);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float sc = (red[0] + red[1] + red[2] + red[3]) / lib[cell];
  __syncthreads();
  float b = 0.f;
  for (int g = tid; g < G; g += 256) {
    const float d = dmu ? mu[base + g] * (dmu[base + g] - sc) * inv_temp : 0.f;
    dl[base + g] = d;
    b += d;
    if (dtheta) {
      const float dt = dtheta[base + g] * theta[base + g];
      if (dt != 0.f) atomicAdd(g_theta + genes[base + g], dt);
    }
  }
  b = wave_sum(b);
  if ((tid & 63) == 0) red[tid >> 6] = b;
  __syncthreads();
  if (tid == 0) bsum[cell] = red[0] + red[1] + red[2] + red[3];
}

// =================================================================================================================================
// Decoder MCAB, per-gene chain backward (layers.py:305-330 with q = gene embeddings, nnets.py:206-208; NB logit head)
//   q0 = E[gene]; qn = LN_1q(q0); qq = Wq qn; ao = softmax(qq K^T / sqrt 8) V (4 heads x 8, 16 latent keys of the cell);
//   y = q0 + Wp ao; h2 = LN_2(y); yo = y + MLP(h2); logit = w_head . yo + b
// grid = (chunks, B), one wave per workgroup, `tiles` 64-gene tiles per workgroup.
// Partial per workgroup (floats): see DP_* below; dK|dV of the cell: dkv_part[(cell * chunks + chunk)][16][64].
// =================================================================================================================================
enum : int { DP_WQ = 0, DP_WP = 1024, DP_W1 = 2048, DP_W2 = DP_W1 + kHP * 32, DP_WC = DP_W2 + kHP * 32, DP_LN1QW = DP_WC + 32 * kHP,
             DP_LN1QB = DP_LN1QW + 32, DP_LN2W = DP_LN1QB + 32, DP_LN2B = DP_LN2W + 32, DP_HEADW = DP_LN2B + 32, DP_SIZE = DP_HEADW + 32 };
struct DecBwdArgs {
  const int64_t* genes;    // (B, G)
  const float* emb;        // (n_genes + 1, 32)
  const float* dl;         // (B, G) dlogit
  const float* kv;         // (B, 16, 64): K | V of the cell's latent tokens
  const float *ln1q_w, *ln1q_b, *wq, *wp, *ln2_w, *ln2_b, *head_w;
  MlpW mlp;
  float* g_emb;            // (n_genes + 1, 32), atomically accumulated
  float* part;             // (B * chunks, DP_SIZE)
  float* dkv_part;         // (B * chunks, 16, 64)
  int G, tiles;
  float eps;
};
__global__ __launch_bounds__(64) void dec_gene_bwd_kernel(const DecBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float SA[64 * kSL], SB[64 * kSL], SC[3 * 64 * kSL];
  const int lane = threadIdx.x, chunk = blockIdx.x, cell = blockIdx.y, nch = gridDim.x;
  const float* __restrict__ KV = a.kv + (size_t)cell * (kT * 64);
  f32x16 gq = z16(), gp = z16(), g1[3] = {z16(), z16(), z16()}, g2[3] = {z16(), z16(), z16()}, gc[3] = {z16(), z16(), z16()};
  f32x16 gK[2] = {z16(), z16()}, gV[2] = {z16(), z16()};
  float v_ln1q = 0.f, v_ln2 = 0.f, v_head = 0.f;   // lanes < 32: weight-type sums, lanes >= 32: bias-type sums (v_head: lanes < 32 only)
  constexpr float kScale = 0.35355339059327373f;   // 1 / sqrt(8)
  for (int t = 0; t < a.tiles; ++t) {
    const int g = (chunk * a.tiles + t) * 64 + lane;
    if ((chunk * a.tiles + t) * 64 >= a.G) break;     // wave-uniform
    const bool valid = g < a.G;
    const size_t gi = (size_t)cell * a.G + (valid ? g : a.G - 1);
    const long long gene = a.genes[gi];
    const float dlog = valid ? a.dl[gi] : 0.f;
    float q0[32], xq[32], qn[32], rq;
    {
      const f32x4* e4 = reinterpret_cast<const f32x4*>(a.emb + (size_t)gene * 32);
#pragma unroll
      for (int q = 0; q < 8; ++q) { const f32x4 v = e4[q]; q0[4 * q] = v[0]; q0[4 * q + 1] = v[1]; q0[4 * q + 2] = v[2]; q0[4 * q + 3] = v[3]; }
    }
    ln_fwd<32>(q0, xq, rq, a.eps);
#pragma unroll
    for (int i = 0; i < 32; ++i) qn[i] = fmaf(xq[i], a.ln1q_w[i], a.ln1q_b[i]);
    float qq[32];
    matvec<32, 32>(a.wq, qn, qq);
    // attention forward: p[h][j], ao
    float p[4][16], ao[32];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float mx = -3.0e38f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) s = fmaf(qq[h * 8 + d], KV[j * 64 + h * 8 + d], s);
        p[h][j] = s * kScale;
        mx = fmaxf(mx, p[h][j]);
      }
      float l = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) { p[h][j] = __expf(p[h][j] - mx); l += p[h][j]; }
      const float inv = 1.0f / l;
#pragma unroll
      for (int d = 0; d < 8; ++d) ao[h * 8 + d] = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        p[h][j] *= inv;
#pragma unroll
        for (int d = 0; d < 8; ++d) ao[h * 8 + d] = fmaf(p[h][j], KV[j * 64 + 32 + h * 8 + d], ao[h * 8 + d]);
      }
    }
    float y[32], xh2[32], h2[32], r2;
    matvec<32, 32>(a.wp, ao, y);
#pragma unroll
    for (int i = 0; i < 32; ++i) y[i] += q0[i];
    ln_fwd<32>(y, xh2, r2, a.eps);
#pragma unroll
    for (int i = 0; i < 32; ++i) h2[i] = fmaf(xh2[i], a.ln2_w[i], a.ln2_b[i]);
    // backward from the logit
    float dyo[32], m[32], dh2[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dyo[i] = dlog * a.head_w[i];
    mlp_fwd_bwd<true>(a.mlp, h2, dyo, m, dh2, g1, g2, gc, SA, SB, SC, lane, valid);
    // head weight: sum dlog * yo ; LN_2 affine gradients
    {
      float t1[32], t2[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) { t1[i] = dlog * (y[i] + m[i]); t2[i] = dh2[i] * xh2[i]; }
      stage32(SA, lane, t1, valid);
      stage32(SB, lane, t2, valid);
      stage32(SC, lane, dh2, valid);
      wsync();
      const float s1 = colsum2(SB, SC, lane);   // LN2 weight | bias
      const float s2 = colsum2(SA, SA, lane);
      v_ln2 += s1;
      v_head += s2;
      wsync();
    }
    float dy[32];
    {
      float dxh[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) { dxh[i] = dh2[i] * a.ln2_w[i]; dy[i] = dyo[i]; }
      ln_bwd_acc<32>(dxh, xh2, r2, dy);
    }
    // y = q0 + Wp ao
    float dao[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dao[i] = 0.f;
    matvec_t_acc<32, 32>(a.wp, dy, dao);
    stage32(SA, lane, dy, valid);
    stage32(SB, lane, ao, valid);
    wsync();
    wgrad32(gp, SA, SB, lane);
    wsync();
    // attention backward (query side); dK / dV of the cell through the lane-axis contraction
    float dqq[32], ds[4][16];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float dg = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float dp = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) dp = fmaf(dao[h * 8 + d], KV[j * 64 + 32 + h * 8 + d], dp);
        ds[h][j] = dp;
        dg = fmaf(p[h][j], dp, dg);
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) dqq[h * 8 + d] = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        ds[h][j] = p[h][j] * (ds[h][j] - dg) * kScale;      // d score * scale
#pragma unroll
        for (int d = 0; d < 8; ++d) dqq[h * 8 + d] = fmaf(ds[h][j], KV[j * 64 + h * 8 + d], dqq[h * 8 + d]);
      }
    }
    // dV[(hl, j)][d] += p[2t + hl][j] * dao[d]  (useful block: head(d) == 2t + hl);  dK likewise with ds and qq
    stage32(SB, lane, dao, valid);
    stage32(SC, lane, qq, valid);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      float pr[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) pr[i] = p[2 * tt + (i >> 4)][i & 15];
      stage32(SA, lane, pr, valid);
      wsync();
      wgrad32(gV[tt], SA, SB, lane);
      wsync();
#pragma unroll
      for (int i = 0; i < 32; ++i) pr[i] = ds[2 * tt + (i >> 4)][i & 15];
      stage32(SA, lane, pr, valid);
      wsync();
      wgrad32(gK[tt], SA, SC, lane);
      wsync();
    }
    // qq = Wq qn
    float dqn[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dqn[i] = 0.f;
    matvec_t_acc<32, 32>(a.wq, dqq, dqn);
    stage32(SA, lane, dqq, valid);
    stage32(SB, lane, qn, valid);
    wsync();
    wgrad32(gq, SA, SB, lane);
    wsync();
    {
      float t1[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) t1[i] = dqn[i] * xq[i];
      stage32(SA, lane, t1, valid);
      stage32(SB, lane, dqn, valid);
      wsync();
      v_ln1q += colsum2(SA, SB, lane);
      wsync();
    }
    {
      float dxh[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) dxh[i] = dqn[i] * a.ln1q_w[i];
      ln_bwd_acc<32>(dxh, xq, rq, dy);      // dq0 = dy (residual) + LN_1q backward
    }
    if (valid && dlog != 0.f) {
      float* ge = a.g_emb + (size_t)gene * 32;
#pragma unroll
      for (int i = 0; i < 32; ++i) atomicAdd(ge + i, dy[i]);
    }
  }
  float* P = a.part + (size_t)(cell * nch + chunk) * DP_SIZE;
  flush_tile(P + DP_WQ, gq, lane, 0, 0, 32);
  flush_tile(P + DP_WP, gp, lane, 0, 0, 32);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    flush_tile(P + DP_W1, g1[c], lane, 32 * c, 0, 32);
    flush_tile(P + DP_W2, g2[c], lane, 32 * c, 0, 32);
    flush_tile(P + DP_WC, gc[c], lane, 0, 32 * c, kHP);
  }
  if (lane < 32) { P[DP_LN1QW + lane] = v_ln1q; P[DP_LN2W + lane] = v_ln2; P[DP_HEADW + lane] = v_head; }
  else { P[DP_LN1QB + lane - 32] = v_ln1q; P[DP_LN2B + lane - 32] = v_ln2; }
  // dK | dV: tile tt holds rows (hl, j), cols d; keep the block with head(d) == 2 tt + hl
  float* DK = a.dkv_part + (size_t)(cell * nch + chunk) * (kT * 64);
  {
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hh), hl = row >> 4, j = row & 15;
        if ((c >> 3) == 2 * tt + hl) {
          DK[j * 64 + c] = gK[tt][r];
          DK[j * 64 + 32 + c] = gV[tt][r];
        }
      }
  }
}

// =================================================================================================================================
// The 16-token side of a cell: trunk Blocks (layers.py:222-226: x += c_proj(attn(LN_1 x)); x += MLP(LN_2 x); 8 heads x 4) and the
// per-cell ends of the two MCABs.  One wave = four cells x 16 tokens; K / V and the key-side backward operands are exchanged
// through LDS rows of kXL floats per lane.
// =================================================================================================================================
struct BlockW { const float *ln1_w, *ln1_b, *wqkv, *wp, *ln2_w, *ln2_b; MlpW mlp; };
struct BlockWArr { BlockW b[16]; };   // up to 16 layers per side
constexpr float kTScale = 0.5f;   // 1 / sqrt(4)

// forward of one Block for this lane's token; EX = exchange area [64][kXL]
__device__ __forceinline__ void block_fwd(const BlockW& w, float (&x)[32], float eps, float* __restrict__ EX, int lane) {
  float xh[32], hn[32], r1;
  ln_fwd<32>(x, xh, r1, eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) hn[i] = fmaf(xh[i], w.ln1_w[i], w.ln1_b[i]);
  float qkv[96];
  matvec<96, 32>(w.wqkv, hn, qkv);
  wsync();
#pragma unroll
  for (int i = 0; i < 64; ++i) EX[lane * kXL + i] = qkv[32 + i];
  wsync();
  const float* KVc = EX + (lane & ~15) * kXL;
  float ao[32];
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    __builtin_amdgcn_sched_barrier(0);   /* keep one head's LDS reads from being hoisted over the previous heads (register pressure) */

    float s[16], mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float t = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) t = fmaf(qkv[h * 4 + d], KVc[j * kXL + h * 4 + d], t);
      s[j] = t * kTScale;
      mx = fmaxf(mx, s[j]);
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { s[j] = __expf(s[j] - mx); l += s[j]; }
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < 4; ++d) ao[h * 4 + d] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int d = 0; d < 4; ++d) ao[h * 4 + d] = fmaf(s[j] * inv, KVc[j * kXL + 32 + h * 4 + d], ao[h * 4 + d]);
  }
  float t[32];
  matvec<32, 32>(w.wp, ao, t);
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] += t[i];
  ln_fwd<32>(x, xh, r1, eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) hn[i] = fmaf(xh[i], w.ln2_w[i], w.ln2_b[i]);
  mlp_fwd(w.mlp, hn, t);
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] += t[i];
}

// Partial layout of one trunk layer (floats)
enum : int { TP_WQKV = 0, TP_WP = 3072, TP_W1 = 4096, TP_W2 = TP_W1 + kHP * 32, TP_WC = TP_W2 + kHP * 32, TP_LN1W = TP_WC + 32 * kHP,
             TP_LN1B = TP_LN1W + 32, TP_LN2W = TP_LN1B + 32, TP_LN2B = TP_LN2W + 32, TP_SIZE = TP_LN2B + 32 };

// backward of one Block: x = the layer's saved input, dx = gradient w.r.t. its output on entry, w.r.t. its input on return;
// the layer's weight-gradient partial goes to P (this workgroup's slot)
__device__ __forceinline__ void block_bwd(const BlockW& w, const float (&x)[32], float (&dx)[32], float eps, float* __restrict__ EX,
                                          float* __restrict__ SA, float* __restrict__ SB, float* __restrict__ SC, float* __restrict__ P,
                                          int lane, bool valid) {
  // ---- recompute the forward
  float xh1[32], hn[32], r1;
  ln_fwd<32>(x, xh1, r1, eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) hn[i] = fmaf(xh1[i], w.ln1_w[i], w.ln1_b[i]);
  float qkv[96];
  matvec<96, 32>(w.wqkv, hn, qkv);
  wsync();
#pragma unroll
  for (int i = 0; i < 64; ++i) EX[lane * kXL + i] = qkv[32 + i];
  wsync();
  const float* KVc = EX + (lane & ~15) * kXL;
  float lse[8], ao[32];      // (the probabilities are recomputed from lse in the backward: 128 registers less across the MLP)
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    __builtin_amdgcn_sched_barrier(0);   /* keep one head's LDS reads from being hoisted over the previous heads (register pressure) */

    float pj[16], mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float t = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) t = fmaf(qkv[h * 4 + d], KVc[j * kXL + h * 4 + d], t);
      pj[j] = t * kTScale;
      mx = fmaxf(mx, pj[j]);
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) l += __expf(pj[j] - mx);
    lse[h] = mx + __logf(l);
#pragma unroll
    for (int d = 0; d < 4; ++d) ao[h * 4 + d] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float pp = __expf(pj[j] - lse[h]);
#pragma unroll
      for (int d = 0; d < 4; ++d) ao[h * 4 + d] = fmaf(pp, KVc[j * kXL + 32 + h * 4 + d], ao[h * 4 + d]);
    }
  }
  float x1[32], xh2[32], h2[32], r2;
  matvec<32, 32>(w.wp, ao, x1);
#pragma unroll
  for (int i = 0; i < 32; ++i) x1[i] += x[i];
  ln_fwd<32>(x1, xh2, r2, eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) h2[i] = fmaf(xh2[i], w.ln2_w[i], w.ln2_b[i]);
  // ---- MLP + LN_2 backward
  float dh2[32], dummy[32];
  {
    f32x16 gu[3];   // (unused in FLUSH mode)
    mlp_fwd_bwd<false, true>(w.mlp, h2, dx, dummy, dh2, gu, gu, gu, SA, SB, SC, lane, valid, P + TP_W1, P + TP_W2, P + TP_WC);
  }
  {
    float t2[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t2[i] = dh2[i] * xh2[i];
    stage32(SA, lane, t2, valid);
    stage32(SB, lane, dh2, valid);
    wsync();
    const float s = colsum2(SA, SB, lane);
    if (lane < 32) P[TP_LN2W + lane] = s; else P[TP_LN2B + lane - 32] = s;
    wsync();
    float dxh[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dxh[i] = dh2[i] * w.ln2_w[i];
    ln_bwd_acc<32>(dxh, xh2, r2, dx);     // dx is now d x1
  }
  // ---- x1 = x + Wp ao
  float dao[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) dao[i] = 0.f;
  matvec_t_acc<32, 32>(w.wp, dx, dao);
  {
    stage32(SA, lane, dx, valid);
    stage32(SB, lane, ao, valid);
    wsync();
    f32x16 gp = z16();
    wgrad32(gp, SA, SB, lane);
    flush_tile(P + TP_WP, gp, lane, 0, 0, 32);
    wsync();
  }
  // ---- attention backward, query side
  float dqkv[96], dgq[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    __builtin_amdgcn_sched_barrier(0);   /* keep one head's LDS reads from being hoisted over the previous heads (register pressure) */

    float dp[16], pj[16], dg = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float t = 0.f, sc = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) { t = fmaf(dao[h * 4 + d], KVc[j * kXL + 32 + h * 4 + d], t); sc = fmaf(qkv[h * 4 + d], KVc[j * kXL + h * 4 + d], sc); }
      dp[j] = t;
      pj[j] = __expf(sc * kTScale - lse[h]);
      dg = fmaf(pj[j], t, dg);
    }
    dgq[h] = dg;
#pragma unroll
    for (int d = 0; d < 4; ++d) dqkv[h * 4 + d] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float dsj = pj[j] * (dp[j] - dg) * kTScale;
#pragma unroll
      for (int d = 0; d < 4; ++d) dqkv[h * 4 + d] = fmaf(dsj, KVc[j * kXL + h * 4 + d], dqkv[h * 4 + d]);
    }
  }
  // ---- key side: every token of the cell needs (q, dao, lse, Dg) of the cell's 16 queries
  float kme[32], vme[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { kme[i] = qkv[32 + i]; vme[i] = qkv[64 + i]; }
  wsync();
#pragma unroll
  for (int i = 0; i < 32; ++i) { EX[lane * kXL + i] = qkv[i]; EX[lane * kXL + 32 + i] = valid ? dao[i] : 0.f; }
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    __builtin_amdgcn_sched_barrier(0);   /* keep one head's LDS reads from being hoisted over the previous heads (register pressure) */
 EX[lane * kXL + 64 + h] = lse[h]; EX[lane * kXL + 72 + h] = dgq[h]; }
  wsync();
#pragma unroll
  for (int i = 0; i < 64; ++i) dqkv[32 + i] = 0.f;
#pragma unroll 1
  for (int qi = 0; qi < 16; ++qi) {
    const float* R = KVc + qi * kXL;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
    __builtin_amdgcn_sched_barrier(0);   /* keep one head's LDS reads from being hoisted over the previous heads (register pressure) */

      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) { s = fmaf(R[h * 4 + d], kme[h * 4 + d], s); dp = fmaf(R[32 + h * 4 + d], vme[h * 4 + d], dp); }
      const float pp = __expf(s * kTScale - R[64 + h]);
      const float dsj = pp * (dp - R[72 + h]) * kTScale;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        dqkv[32 + h * 4 + d] = fmaf(dsj, R[h * 4 + d], dqkv[32 + h * 4 + d]);
        dqkv[64 + h * 4 + d] = fmaf(pp, R[32 + h * 4 + d], dqkv[64 + h * 4 + d]);
      }
    }
  }
  wsync();
  // ---- qkv = Wqkv hn ; LN_1 backward
  float dhn[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) dhn[i] = 0.f;
  matvec_t_acc<96, 32>(w.wqkv, dqkv, dhn);
  stage32(SB, lane, hn, valid);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float t[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i] = dqkv[c * 32 + i];
    stage32(SA, lane, t, valid);
    wsync();
    f32x16 g = z16();
    wgrad32(g, SA, SB, lane);
    flush_tile(P + TP_WQKV, g, lane, 32 * c, 0, 32);
    wsync();
  }
  {
    float t2[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t2[i] = dhn[i] * xh1[i];
    stage32(SA, lane, t2, valid);
    stage32(SC, lane, dhn, valid);
    wsync();
    const float s = colsum2(SA, SC, lane);
    if (lane < 32) P[TP_LN1W + lane] = s; else P[TP_LN1B + lane - 32] = s;
    wsync();
    float dxh[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dxh[i] = dhn[i] * w.ln1_w[i];
    ln_bwd_acc<32>(dxh, xh1, r1, dx);     // dx is now d x (input of the layer)
  }
}

// ---- decoder cell side ----------------------------------------------------------------------------------------------------------
// forward with saved layer inputs: z (16 x n_lat) -> LN (no affine) -> Linear -> n_layer Blocks -> h_lat; kv = c_attn(LN_1 h_lat)
// xsave: (B, n_layer + 1, 16, 32)
struct DecCellTrainArgs {
  const float* z;          // (B, 16, n_lat)
  const float* w_in;       // decoder_latent_input.1.weight (32, n_lat)
  BlockWArr blocks;
  const float *cln1_w, *cln1_b, *wkv;   // decoder_cross_attention.ln_1, attn.c_attn (64, 32)
  float* xsave;
  float* kv;               // (B, 16, 64)
  // backward only
  const float* dkv_part;   // (B * chunks, 16, 64)
  int chunks;
  float* dz;               // (B, 16, n_lat): gradient w.r.t. z from the decoder
  float* part;             // (workgroups, DC_SIZE)
  int B, n_lat, n_layer;
  float eps;
};
// partial of the decoder cell kernel: n_layer trunk layers, then the cross-attention's K/V side and the latent input Linear
__host__ __device__ constexpr int dc_off_wkv(int n_layer) { return n_layer * TP_SIZE; }
__host__ __device__ constexpr int dc_off_cln1w(int n_layer) { return dc_off_wkv(n_layer) + 64 * 32; }
__host__ __device__ constexpr int dc_off_cln1b(int n_layer) { return dc_off_cln1w(n_layer) + 32; }
__host__ __device__ constexpr int dc_off_win(int n_layer) { return dc_off_cln1b(n_layer) + 32; }      // stored as [32][32] (cols >= n_lat zero)
__host__ __device__ constexpr int dc_size(int n_layer) { return dc_off_win(n_layer) + 1024; }

template <int NL>   // NL = n_lat rounded up to 16 / 32 (register array size)
__global__ __launch_bounds__(64) void dec_cell_fwd_kernel(const DecCellTrainArgs a) {
  __shared__ __attribute__((aligned(16))) float EX[64 * kXL];
  const int lane = threadIdx.x, cell_raw = blockIdx.x * 4 + (lane >> 4), tok = lane & 15;
  const bool valid = cell_raw < a.B;
  const int cell = valid ? cell_raw : a.B - 1;
  float zv[NL], zn[NL], rz;
#pragma unroll
  for (int i = 0; i < NL; ++i) zv[i] = i < a.n_lat ? a.z[((size_t)cell * kT + tok) * a.n_lat + i] : 0.f;
  if (a.n_lat == NL) ln_fwd<NL>(zv, zn, rz, a.eps);
  else {   // generic width: statistics over the first n_lat entries
    float s = 0.f;
    for (int i = 0; i < a.n_lat; ++i) s += zv[i];
    const float mean = s / a.n_lat;
    float v = 0.f;
    for (int i = 0; i < a.n_lat; ++i) v += (zv[i] - mean) * (zv[i] - mean);
    rz = 1.0f / sqrtf(v / a.n_lat + a.eps);
#pragma unroll
    for (int i = 0; i < NL; ++i) zn[i] = i < a.n_lat ? (zv[i] - mean) * rz : 0.f;
  }
  float x[32];
#pragma unroll
  for (int o = 0; o < 32; ++o) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) if (i < a.n_lat) s = fmaf(a.w_in[o * a.n_lat + i], zn[i], s);
    x[o] = s;
  }
  float* XS = a.xsave + ((size_t)cell * (a.n_layer + 1) * kT + tok) * 32;
  for (int l = 0; l < a.n_layer; ++l) {
    if (valid)
#pragma unroll
      for (int i = 0; i < 32; ++i) XS[(size_t)l * kT * 32 + i] = x[i];
    block_fwd(a.blocks.b[l], x, a.eps, EX, lane);
  }
  if (valid)
#pragma unroll
    for (int i = 0; i < 32; ++i) XS[(size_t)a.n_layer * kT * 32 + i] = x[i];
  float xh[32], hn[32], r;
  ln_fwd<32>(x, xh, r, a.eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) hn[i] = fmaf(xh[i], a.cln1_w[i], a.cln1_b[i]);
  float kv[64];
  matvec<64, 32>(a.wkv, hn, kv);
  if (valid)
#pragma unroll
    for (int i = 0; i < 64; ++i) a.kv[((size_t)cell * kT + tok) * 64 + i] = kv[i];
}

template <int NL>
__global__ __launch_bounds__(64) void dec_cell_bwd_kernel(const DecCellTrainArgs a) {
  __shared__ __attribute__((aligned(16))) float EX[64 * kXL], SA[64 * kSL], SB[64 * kSL], SC[3 * 64 * kSL];
  const int lane = threadIdx.x, cell_raw = blockIdx.x * 4 + (lane >> 4), tok = lane & 15;
  const bool valid = cell_raw < a.B;
  const int cell = valid ? cell_raw : a.B - 1;
  float* P = a.part + (size_t)blockIdx.x * dc_size(a.n_layer);
  const float* XS = a.xsave + ((size_t)cell * (a.n_layer + 1) * kT + tok) * 32;
  // d kv of this token: sum of the per-chunk partials
  float dkv[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) dkv[i] = 0.f;
  if (valid)
    for (int c = 0; c < a.chunks; ++c) {
      const float* src = a.dkv_part + ((size_t)(cell * a.chunks + c) * kT + tok) * 64;
#pragma unroll
      for (int i = 0; i < 64; ++i) dkv[i] += src[i];
    }
  float x[32], xh[32], hn[32], r, dx[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = XS[(size_t)a.n_layer * kT * 32 + i];
  ln_fwd<32>(x, xh, r, a.eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) hn[i] = fmaf(xh[i], a.cln1_w[i], a.cln1_b[i]);
  {
    float dhn[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dhn[i] = 0.f;
    matvec_t_acc<64, 32>(a.wkv, dkv, dhn);
    stage32(SB, lane, hn, valid);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float t[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) t[i] = dkv[c * 32 + i];
      stage32(SA, lane, t, valid);
      wsync();
      f32x16 g = z16();
      wgrad32(g, SA, SB, lane);
      flush_tile(P + dc_off_wkv(a.n_layer), g, lane, 32 * c, 0, 32);
      wsync();
    }
    float t2[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t2[i] = dhn[i] * xh[i];
    stage32(SA, lane, t2, valid);
    stage32(SC, lane, dhn, valid);
    wsync();
    const float s = colsum2(SA, SC, lane);
    if (lane < 32) P[dc_off_cln1w(a.n_layer) + lane] = s; else P[dc_off_cln1b(a.n_layer) + lane - 32] = s;
    wsync();
    float dxh[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { dxh[i] = dhn[i] * a.cln1_w[i]; dx[i] = 0.f; }
    ln_bwd_acc<32>(dxh, xh, r, dx);
  }
  for (int l = a.n_layer - 1; l >= 0; --l) {
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = XS[(size_t)l * kT * 32 + i];
    block_bwd(a.blocks.b[l], x, dx, a.eps, EX, SA, SB, SC, P + (size_t)l * TP_SIZE, lane, valid);
  }
  // x0 = W_in LN(z)
  float zv[NL], zn[32], rz, mean = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) zv[i] = i < a.n_lat ? a.z[((size_t)cell * kT + tok) * a.n_lat + i] : 0.f;
  {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += zv[i];
    mean = s / a.n_lat;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) if (i < a.n_lat) v += (zv[i] - mean) * (zv[i] - mean);
    rz = 1.0f / sqrtf(v / a.n_lat + a.eps);
#pragma unroll
    for (int i = 0; i < 32; ++i) zn[i] = (i < NL && i < a.n_lat) ? (zv[i < NL ? i : 0] - mean) * rz : 0.f;
  }
  stage32(SA, lane, dx, valid);
  stage32(SB, lane, zn, valid);
  wsync();
  {
    f32x16 g = z16();
    wgrad32(g, SA, SB, lane);
    flush_tile(P + dc_off_win(a.n_layer), g, lane, 0, 0, 32);
  }
  float dzn[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) dzn[i] = 0.f;
#pragma unroll
  for (int o = 0; o < 32; ++o)
#pragma unroll
    for (int i = 0; i < NL; ++i) if (i < a.n_lat) dzn[i] = fmaf(a.w_in[o * a.n_lat + i], dx[o], dzn[i]);
  // LN backward over n_lat entries (no affine)
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) if (i < a.n_lat) { sa += dzn[i]; sb = fmaf(dzn[i], zn[i], sb); }
  sa /= a.n_lat;
  sb /= a.n_lat;
  if (valid)
#pragma unroll
    for (int i = 0; i < NL; ++i) if (i < a.n_lat) a.dz[((size_t)cell * kT + tok) * a.n_lat + i] = rz * (dzn[i] - sa - zn[i] * sb);
}

// ---- encoder cell side ----------------------------------------------------------------------------------------------------------
// y = P + Wp ao; y2 = y + MLP(LN_2 y); x0 = y2 + pos; n_layer Blocks -> hL; zl = W_lat hL; z = LN(zl)   (layers.py:326-330, nnets.py:139-144)
struct EncCellTrainArgs {
  const float* pooled;     // (B, 16, 32): attention output of the pooling (heads concatenated)
  const float* ind;        // inducing points (16, 32)
  const float *wp, *cln2_w, *cln2_b;
  MlpW cmlp;
  const float* pos;        // (16, 32) or nullptr
  BlockWArr blocks;
  const float* w_lat;      // encoder_latent_input.0.weight (n_lat, 32)
  float* xsave;            // (B, n_layer + 1, 16, 32)
  float* ysave;            // (B, 16, 32): y (input of LN_2)
  // backward
  const float* dz_a;       // (B, 16, n_lat) or nullptr: gradient w.r.t. z from the decoder
  const float* dz_b;       // (B, 16, n_lat) or nullptr: gradient w.r.t. the returned z
  float* dao;              // (B, 16, 32): gradient w.r.t. the pooled attention output
  float* dgq;              // (B, 4, 16): sum_d dao[i, h, d] * ao[i, h, d]
  float* part;
  int B, n_lat, n_layer;
  float eps;
};
__host__ __device__ constexpr int ec_off_wp(int n_layer) { return n_layer * TP_SIZE; }
__host__ __device__ constexpr int ec_off_w1(int n_layer) { return ec_off_wp(n_layer) + 1024; }
__host__ __device__ constexpr int ec_off_w2(int n_layer) { return ec_off_w1(n_layer) + kHP * 32; }
__host__ __device__ constexpr int ec_off_wc(int n_layer) { return ec_off_w2(n_layer) + kHP * 32; }
__host__ __device__ constexpr int ec_off_ln2w(int n_layer) { return ec_off_wc(n_layer) + 32 * kHP; }
__host__ __device__ constexpr int ec_off_ln2b(int n_layer) { return ec_off_ln2w(n_layer) + 32; }
__host__ __device__ constexpr int ec_off_wlat(int n_layer) { return ec_off_ln2b(n_layer) + 32; }     // [32][32], rows >= n_lat zero
__host__ __device__ constexpr int ec_off_ind(int n_layer) { return ec_off_wlat(n_layer) + 1024; }    // [16][32]
__host__ __device__ constexpr int ec_size(int n_layer) { return ec_off_ind(n_layer) + 512; }

__global__ __launch_bounds__(64) void enc_cell_fwd_kernel(const EncCellTrainArgs a) {
  __shared__ __attribute__((aligned(16))) float EX[64 * kXL];
  const int lane = threadIdx.x, cell_raw = blockIdx.x * 4 + (lane >> 4), tok = lane & 15;
  const bool valid = cell_raw < a.B;
  const int cell = valid ? cell_raw : a.B - 1;
  float ao[32], y[32], xh[32], h2[32], m[32], r;
#pragma unroll
  for (int i = 0; i < 32; ++i) ao[i] = a.pooled[((size_t)cell * kT + tok) * 32 + i];
  matvec<32, 32>(a.wp, ao, y);
#pragma unroll
  for (int i = 0; i < 32; ++i) y[i] += a.ind[tok * 32 + i];
  if (valid)
#pragma unroll
    for (int i = 0; i < 32; ++i) a.ysave[((size_t)cell * kT + tok) * 32 + i] = y[i];
  ln_fwd<32>(y, xh, r, a.eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) h2[i] = fmaf(xh[i], a.cln2_w[i], a.cln2_b[i]);
  mlp_fwd(a.cmlp, h2, m);
#pragma unroll
  for (int i = 0; i < 32; ++i) y[i] += m[i] + (a.pos ? a.pos[tok * 32 + i] : 0.f);
  float* XS = a.xsave + ((size_t)cell * (a.n_layer + 1) * kT + tok) * 32;
  for (int l = 0; l < a.n_layer; ++l) {
    if (valid)
#pragma unroll
      for (int i = 0; i < 32; ++i) XS[(size_t)l * kT * 32 + i] = y[i];
    block_fwd(a.blocks.b[l], y, a.eps, EX, lane);
  }
  if (valid)
#pragma unroll
    for (int i = 0; i < 32; ++i) XS[(size_t)a.n_layer * kT * 32 + i] = y[i];
}

template <int NL>
__global__ __launch_bounds__(64) void enc_cell_bwd_kernel(const EncCellTrainArgs a) {
  __shared__ __attribute__((aligned(16))) float EX[64 * kXL], SA[64 * kSL], SB[64 * kSL], SC[3 * 64 * kSL];
  const int lane = threadIdx.x, cell_raw = blockIdx.x * 4 + (lane >> 4), tok = lane & 15;
  const bool valid = cell_raw < a.B;
  const int cell = valid ? cell_raw : a.B - 1;
  float* P = a.part + (size_t)blockIdx.x * ec_size(a.n_layer);
  const float* XS = a.xsave + ((size_t)cell * (a.n_layer + 1) * kT + tok) * 32;
  float x[32], dx[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = XS[(size_t)a.n_layer * kT * 32 + i];
  // zl = W_lat hL (n_lat x 32); z = LN(zl) without affine
  {
    float zl[32], zn[32], dzl[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) zl[o] = (o < NL && o < a.n_lat) ? dotw<32>(a.w_lat + (o < a.n_lat ? o : 0) * 32, x) : 0.f;
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < NL; ++o) s += zl[o];
    const float mean = s / a.n_lat;
    float v = 0.f;
#pragma unroll
    for (int o = 0; o < NL; ++o) if (o < a.n_lat) v += (zl[o] - mean) * (zl[o] - mean);
    const float rz = 1.0f / sqrtf(v / a.n_lat + a.eps);
    float sa = 0.f, sb = 0.f, dzv[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) {
      const bool in = o < NL && o < a.n_lat;
      zn[o] = in ? (zl[o] - mean) * rz : 0.f;
      const size_t zi = ((size_t)cell * kT + tok) * a.n_lat + (in ? o : 0);
      dzv[o] = in ? ((a.dz_a ? a.dz_a[zi] : 0.f) + (a.dz_b ? a.dz_b[zi] : 0.f)) : 0.f;
      sa += dzv[o];
      sb = fmaf(dzv[o], zn[o], sb);
    }
    sa /= a.n_lat;
    sb /= a.n_lat;
#pragma unroll
    for (int o = 0; o < 32; ++o) dzl[o] = (o < NL && o < a.n_lat) ? rz * (dzv[o] - sa - zn[o] * sb) : 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) dx[i] = 0.f;
#pragma unroll
    for (int o = 0; o < NL; ++o)
      if (o < a.n_lat)
#pragma unroll
        for (int i = 0; i < 32; ++i) dx[i] = fmaf(a.w_lat[o * 32 + i], dzl[o], dx[i]);
    stage32(SA, lane, dzl, valid);
    stage32(SB, lane, x, valid);
    wsync();
    f32x16 g = z16();
    wgrad32(g, SA, SB, lane);
    flush_tile(P + ec_off_wlat(a.n_layer), g, lane, 0, 0, 32);
    wsync();
  }
  for (int l = a.n_layer - 1; l >= 0; --l) {
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = XS[(size_t)l * kT * 32 + i];
    block_bwd(a.blocks.b[l], x, dx, a.eps, EX, SA, SB, SC, P + (size_t)l * TP_SIZE, lane, valid);
  }
  // x0 = y + MLP(LN_2 y) + pos  (pos_embed is frozen: nnets.py:103-106)
  float y[32], xh2[32], h2[32], r2, ao[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { y[i] = a.ysave[((size_t)cell * kT + tok) * 32 + i]; ao[i] = a.pooled[((size_t)cell * kT + tok) * 32 + i]; }
  ln_fwd<32>(y, xh2, r2, a.eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) h2[i] = fmaf(xh2[i], a.cln2_w[i], a.cln2_b[i]);
  float dh2[32], dummy[32];
  {
    f32x16 gu[3];
    mlp_fwd_bwd<false, true>(a.cmlp, h2, dx, dummy, dh2, gu, gu, gu, SA, SB, SC, lane, valid, P + ec_off_w1(a.n_layer), P + ec_off_w2(a.n_layer),
                             P + ec_off_wc(a.n_layer));
  }
  {
    float t2[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t2[i] = dh2[i] * xh2[i];
    stage32(SA, lane, t2, valid);
    stage32(SB, lane, dh2, valid);
    wsync();
    const float s = colsum2(SA, SB, lane);
    if (lane < 32) P[ec_off_ln2w(a.n_layer) + lane] = s; else P[ec_off_ln2b(a.n_layer) + lane - 32] = s;
    wsync();
    float dxh[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dxh[i] = dh2[i] * a.cln2_w[i];
    ln_bwd_acc<32>(dxh, xh2, r2, dx);    // dx = dy
  }
  // y = inducing + Wp ao: d inducing (sum over the four cells of this wave), d Wp, d ao
  stage32(SA, lane, dx, valid);
  stage32(SB, lane, ao, valid);
  wsync();
  {
    f32x16 g = z16();
    wgrad32(g, SA, SB, lane);
    flush_tile(P + ec_off_wp(a.n_layer), g, lane, 0, 0, 32);
    // d inducing[tok][f]: lanes 0..15 (tok) x 32 features, summed over the 4 cells (rows tok, tok + 16, tok + 32, tok + 48 of SA)
    if (lane < 16)
#pragma unroll 4
      for (int f = 0; f < 32; ++f)
        P[ec_off_ind(a.n_layer) + lane * 32 + f] = SA[lane * kSL + f] + SA[(lane + 16) * kSL + f] + SA[(lane + 32) * kSL + f] + SA[(lane + 48) * kSL + f];
  }
  float dao[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) dao[i] = 0.f;
  matvec_t_acc<32, 32>(a.wp, dx, dao);
  if (valid) {
#pragma unroll
    for (int i = 0; i < 32; ++i) a.dao[((size_t)cell * kT + tok) * 32 + i] = dao[i];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) s = fmaf(dao[h * 8 + d], ao[h * 8 + d], s);
      a.dgq[((size_t)cell * 4 + h) * kT + tok] = s;
    }
  }
}

// =================================================================================================================================
// Encoder MCAB pooling backward, key side (layers.py:111-118, 248-264, 325-326): one lane per input gene token
//   x = E[gene] log1p(count); xn = LN_1(x); k | v = c_attn xn; p[i][h] = exp2(log2e / sqrt 8 * Q[i][h] . k[h] - lse2[i][h])
// grid = (chunks, B).  Partial per workgroup: EP_* ; gene-embedding gradient by atomics.
// =================================================================================================================================
enum : int { EP_WKV = 0, EP_LN1W = 2048, EP_LN1B = 2080, EP_DQ = 2112, EP_SIZE = EP_DQ + 64 * 32 };
struct EncPoolBwdArgs {
  const float* counts;     // (B, S)
  const int64_t* genes;    // (B, S)
  const float* emb;
  const float *ln1_w, *ln1_b, *wkv;
  const float* Q;          // (16, 32) = c_attn_q(LN_1q(inducing))
  const float* lse2;       // (B, 4, 16): log2-domain log-sum-exp of the scaled scores
  const float* dao;        // (B, 16, 32)
  const float* dgq;        // (B, 4, 16)
  float* g_emb;
  float* part;             // (B * chunks, EP_SIZE)
  int S, tiles;
  float eps;
};
__global__ __launch_bounds__(64) void enc_pool_bwd_kernel(const EncPoolBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float SA[64 * kSL], SB[64 * kSL], SC[64 * kSL];
  const int lane = threadIdx.x, chunk = blockIdx.x, cell = blockIdx.y, nch = gridDim.x;
  const float* __restrict__ DAO = a.dao + (size_t)cell * kT * 32;
  const float* __restrict__ DG = a.dgq + (size_t)cell * 64;
  const float* __restrict__ LSE = a.lse2 + (size_t)cell * 64;
  constexpr float kS2 = 1.4426950408889634f * 0.35355339059327373f;   // log2(e) / sqrt(8)
  constexpr float kScale = 0.35355339059327373f;
  f32x16 gkv[2] = {z16(), z16()}, gq[2] = {z16(), z16()};
  float v_ln1 = 0.f;
  for (int t = 0; t < a.tiles; ++t) {
    const int s0 = (chunk * a.tiles + t) * 64;
    if (s0 >= a.S) break;
    const int s = s0 + lane;
    const bool valid = s < a.S;
    const size_t si = (size_t)cell * a.S + (valid ? s : a.S - 1);
    const long long gene = a.genes[si];
    const float lc = log1pf(a.counts[si]);
    float x[32], xh[32], xn[32], r;
    {
      const f32x4* e4 = reinterpret_cast<const f32x4*>(a.emb + (size_t)gene * 32);
#pragma unroll
      for (int q = 0; q < 8; ++q) { const f32x4 v = e4[q]; x[4 * q] = v[0] * lc; x[4 * q + 1] = v[1] * lc; x[4 * q + 2] = v[2] * lc; x[4 * q + 3] = v[3] * lc; }
    }
    ln_fwd<32>(x, xh, r, a.eps);
#pragma unroll
    for (int i = 0; i < 32; ++i) xn[i] = fmaf(xh[i], a.ln1_w[i], a.ln1_b[i]);
    float kv[64], dkv[64], dsv[64];
    matvec<64, 32>(a.wkv, xn, kv);
#pragma unroll
    for (int i = 0; i < 64; ++i) dkv[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) { sc = fmaf(a.Q[i * 32 + h * 8 + d], kv[h * 8 + d], sc); dp = fmaf(DAO[i * 32 + h * 8 + d], kv[32 + h * 8 + d], dp); }
        const float pp = valid ? __builtin_amdgcn_exp2f(sc * kS2 - LSE[h * 16 + i]) : 0.f;
        const float ds = pp * (dp - DG[h * 16 + i]) * kScale;
        dsv[h * 16 + i] = ds;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          dkv[h * 8 + d] = fmaf(ds, a.Q[i * 32 + h * 8 + d], dkv[h * 8 + d]);
          dkv[32 + h * 8 + d] = fmaf(pp, DAO[i * 32 + h * 8 + d], dkv[32 + h * 8 + d]);
        }
      }
    // dQ[(h, i)][d] += ds[h][i] * k[d]  (useful block: head(d) == h): tiles of 32 rows = heads (2 tt, 2 tt + 1)
    {
      float kk[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) kk[i] = kv[i];
      stage32(SB, lane, kk, valid);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        float pr[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) pr[i] = dsv[tt * 32 + i];
        stage32(SA, lane, pr, valid);
        wsync();
        wgrad32(gq[tt], SA, SB, lane);
        wsync();
      }
    }
    // k | v = Wkv xn
    float dxn[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dxn[i] = 0.f;
    matvec_t_acc<64, 32>(a.wkv, dkv, dxn);
    stage32(SB, lane, xn, valid);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float tv[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) tv[i] = dkv[c * 32 + i];
      stage32(SA, lane, tv, valid);
      wsync();
      wgrad32(gkv[c], SA, SB, lane);
      wsync();
    }
    {
      float t2[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) t2[i] = dxn[i] * xh[i];
      stage32(SA, lane, t2, valid);
      stage32(SC, lane, dxn, valid);
      wsync();
      v_ln1 += colsum2(SA, SC, lane);
      wsync();
    }
    if (valid && lc != 0.f) {
      float dxh[32], dx[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) { dxh[i] = dxn[i] * a.ln1_w[i]; dx[i] = 0.f; }
      ln_bwd_acc<32>(dxh, xh, r, dx);
      float* ge = a.g_emb + (size_t)gene * 32;
#pragma unroll
      for (int i = 0; i < 32; ++i) atomicAdd(ge + i, dx[i] * lc);
    }
  }
  float* P = a.part + (size_t)(cell * nch + chunk) * EP_SIZE;
  flush_tile(P + EP_WKV, gkv[0], lane, 0, 0, 32);
  flush_tile(P + EP_WKV, gkv[1], lane, 32, 0, 32);
  if (lane < 32) P[EP_LN1W + lane] = v_ln1; else P[EP_LN1B + lane - 32] = v_ln1;
  // dQ: tile tt rows (hl, i), cols d; keep head(d) == 2 tt + hl; stored as [h*16 + i][32] with the other entries zero
  {
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hh), hl = row >> 4;
        P[EP_DQ + (tt * 32 + row) * 32 + c] = ((c >> 3) == 2 * tt + hl) ? gq[tt][r] : 0.f;
      }
  }
}

// =================================================================================================================================
// Partial reduction: dst[i] (+)= sum_p part[p * stride + off + i]   (index order: deterministic)
// =================================================================================================================================
struct RedJob { float* dst; int off, n, accumulate; int rows, ld_src, ld_dst; };   // rows > 1: a [rows][ld_src] block copied to [rows][ld_dst] (n = cols)
constexpr int kMaxRedJobs = 96;   // (3 KB of kernel arguments; a cell side of 8 layers is 76-84 jobs: one launch)
struct RedArgs { const float* part; int n_part; long stride; int n_jobs; RedJob job[kMaxRedJobs]; };
// sum over the partials of one element, eight independent running sums (memory-level parallelism; fixed order: deterministic)
__device__ __forceinline__ float sum_partials(const float* __restrict__ src, int n_part, long stride) {
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int p = 0;
  for (; p + 8 <= n_part; p += 8)
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += src[(size_t)(p + u) * stride];
  for (; p < n_part; ++p) s[p & 7] += src[(size_t)p * stride];
  return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
__global__ __launch_bounds__(256) void reduce_jobs_kernel(const RedArgs a) {
  const RedJob& j = a.job[blockIdx.y];
  const int total = j.rows * j.n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int r = i / j.n, c = i % j.n;
    const float s = sum_partials(a.part + j.off + (size_t)r * j.ld_src + c, a.n_part, a.stride);
    float* d = j.dst + (size_t)r * j.ld_dst + c;
    *d = j.accumulate ? *d + s : s;
  }
}
// dQ (16, 32) from the pooling partial's [h*16 + i][32] block-diagonal rows: dQ[i][d] = row(head(d), i)[d]
__global__ __launch_bounds__(64) void fold_dq_kernel(const float* __restrict__ part, int n_part, long stride, float* __restrict__ dQ) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= 16 * 32) return;
  const int i = idx >> 5, d = idx & 31, h = d >> 3;
  dQ[idx] = sum_partials(part + EP_DQ + (h * 16 + i) * 32 + d, n_part, stride);
}

// =================================================================================================================================
// log_nb_positive with its gradient (src/scldm/distributions.py:6-42), elementwise
// =================================================================================================================================
__device__ __forceinline__ float digammaf_dev(float x) {
  // psi(x) for x > 0: recurrence up to x >= 6, then the asymptotic series
  float r = 0.f;
  while (x < 6.0f) { r -= 1.0f / x; x += 1.0f; }
  const float f = 1.0f / (x * x);
  return r + logf(x) - 0.5f / x - f * (1.0f / 12.0f - f * (1.0f / 120.0f - f * (1.0f / 252.0f - f * (1.0f / 240.0f - f * (1.0f / 132.0f)))));
}
__global__ __launch_bounds__(256) void nb_loglik_kernel(const float* __restrict__ x, const float* __restrict__ mu, const float* __restrict__ theta,
                                                        float eps, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float xv = x[i], m = mu[i], th = theta[i];
    const float ltm = logf(th + m + eps);
    out[i] = th * (logf(th + eps) - ltm) + xv * (logf(m + eps) - ltm) + lgammaf(xv + th) - lgammaf(th) - lgammaf(xv + 1.0f);
  }
}
__global__ __launch_bounds__(256) void nb_loglik_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mu, const float* __restrict__ theta,
                                                            const float* __restrict__ gout, float eps, float* __restrict__ dmu,
                                                            float* __restrict__ dtheta, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float xv = x[i], m = mu[i], th = theta[i], g = gout[i];
    const float inv = 1.0f / (th + m + eps);
    if (dmu) dmu[i] = g * (xv / (m + eps) - (th + xv) * inv);
    if (dtheta)
      dtheta[i] = g * (logf(th + eps) - logf(th + m + eps) + th / (th + eps) - (th + xv) * inv + digammaf_dev(xv + th) - digammaf_dev(th));
  }
}

}  // namespace vtrain
}  // namespace scldm
