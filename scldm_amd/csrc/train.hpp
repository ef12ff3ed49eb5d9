// Training path of the DiT: forward with saved activations and the hand-derived
// backward, as plain (unfused) gfx950 kernels around one exact-fp32 MFMA GEMM.  First correct version: every
// intermediate makes an HBM round trip; the fused inference kernel (dit_forward.hpp) is not used here.
//
// Reference arithmetic being differentiated: Block.forward adaLN branch (src/scldm/layers.py:208-221), modulate
// (:91-94), SelfAttention (:143-158), MLP (:161-174), FinalLayerDit (:397-401), TimestepEmbedder (:351-364),
// DiT.forward (src/scldm/nnets.py:273-297).  The reference gets its gradients from torch autograd.
#pragma once
#include <type_traits>
#include "../../include/scldm_hip.h"
#include "common.hpp"

namespace scldm {
namespace train {

constexpr int kS = 16;     // tokens per sample (seq_len of every reference config)

// ---------------------------------------------------------------------------------------------------------------
// C[m][n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n]),  A(m,k) = A[m*sam + k*sak], B(n,k) = B[n*sbn + k*sbk].
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).  Tiles are staged through LDS k-major
// ([k][m], m contiguous) so the MFMA operand reads (lane = row, two k per instruction) are bank-conflict free
// whatever the source orientation; *_KC says the source is contiguous along k (sak/sbk == 1), otherwise it is
// contiguous along m/n (sam/sbn == 1).  Split-K: blockIdx.z covers k range [z*kchunk, (z+1)*kchunk) and writes its
// partial tile to C + z*M*ldc (the caller reduces deterministically with reduce_partials_kernel).
// ---------------------------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* A; long sam, sak;
  const float* B; long sbn, sbk;
  float* C; long ldc;
  const float* bias;
  int M, N, K;
  int kchunk;      // multiple of kBK; == K rounded up when there is no split
  int accumulate;  // C += (only without split-K)
  float* rowsum;   // optional: rowsum[z*M + m] = sum_k A(m,k) over this split (bias gradient of a weight-gradient GEMM;
                   // needs A contiguous along m); written by the workgroups of the first column of tiles
};

constexpr int kBK = 32;
template <int WTM, int WTN>
constexpr int gemm_smem_bytes() { return 2 * kBK * (64 * WTM + 4 + 64 * WTN + 4) * 4; }

template <int BM, bool KC>
struct TileLoader {
  static constexpr int kVec = BM / 32;  // float4 loads per thread per tile
  f32x4 v[kVec];
  // Loads tile rows [m0, m0+BM) x k [k0, k0+kBK) (zero outside [0,M) x [0,kend)).
  __device__ __forceinline__ void load(const float* __restrict__ P, long sm, long sk, int m0, int M, int k0, int kend) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
      if constexpr (KC) {
        const int m = m0 + (tid >> 3) + 32 * j, k = k0 + (tid & 7) * 4;
        if (m < M) {
          const float* p = P + (long)m * sm + k;
          if (k + 3 < kend) r = *reinterpret_cast<const f32x4*>(p);
          else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (k + i < kend) r[i] = p[i];
          }
        }
      } else {
        constexpr int kPerRow = BM / 4;            // float4 per k row
        constexpr int kRowsPerPass = 256 / kPerRow;
        const int m = m0 + (tid % kPerRow) * 4, k = k0 + tid / kPerRow + kRowsPerPass * j;
        if (k < kend) {
          const float* p = P + (long)k * sk + m;
          if (m + 3 < M) r = *reinterpret_cast<const f32x4*>(p);
          else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (m + i < M) r[i] = p[i];
          }
        }
      }
      v[j] = r;
    }
  }
  // MC only: rs[i] += this tile's values of row (tid % (BM/4))*4 + i
  __device__ __forceinline__ void add_rowsum(float (&rs)[4]) const {
#pragma unroll
    for (int j = 0; j < kVec; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) rs[i] += v[j][i];
  }
  __device__ __forceinline__ void store(float* __restrict__ S /* [kBK][BM+4] */) const {
    constexpr int LD = BM + 4;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      if constexpr (KC) {
        const int m = (tid >> 3) + 32 * j, k = (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) S[(k + i) * LD + m] = v[j][i];
      } else {
        constexpr int kPerRow = BM / 4;
        constexpr int kRowsPerPass = 256 / kPerRow;
        const int m = (tid % kPerRow) * 4, k = tid / kPerRow + kRowsPerPass * j;
        *reinterpret_cast<f32x4*>(S + k * LD + m) = v[j];
      }
    }
  }
};

template <int WTM, int WTN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void sgemm_kernel(GemmArgs g) {
  constexpr int BM = 64 * WTM, BN = 64 * WTN, LDA = BM + 4, LDB = BN + 4;
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];   // As[2][kBK*LDA] | Bs[2][kBK*LDB]
  auto As = [&](int b) { return gemm_smem + b * (kBK * LDA); };
  auto Bs = [&](int b) { return gemm_smem + 2 * kBK * LDA + b * (kBK * LDB); };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  float* __restrict__ C = g.C + (long)blockIdx.z * g.M * g.ldc;

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileLoader<BM, A_KC> la;
  TileLoader<BN, B_KC> lb;
  const bool want_rs = !A_KC && g.rowsum != nullptr && blockIdx.x == 0;   // workgroup-uniform
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  la.load(g.A, g.sam, g.sak, m0, g.M, kbeg, kend);
  lb.load(g.B, g.sbn, g.sbk, n0, g.N, kbeg, kend);
  if constexpr (!A_KC) if (want_rs) la.add_rowsum(rs);
  la.store(As(0));
  lb.store(Bs(0));
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kBK) {
    const bool more = k0 + kBK < kend;
    if (more) {
      la.load(g.A, g.sam, g.sak, m0, g.M, k0 + kBK, kend);
      lb.load(g.B, g.sbn, g.sbk, n0, g.N, k0 + kBK, kend);
    }
    const float* __restrict__ as = As(buf) + (lane >> 5) * LDA + wm * 32 * WTM + (lane & 31);
    const float* __restrict__ bs = Bs(buf) + (lane >> 5) * LDB + wn * 32 * WTN + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < kBK; kk += 2) {
      float a[WTM], b[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) a[i] = as[kk * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < WTN; ++j) b[j] = bs[kk * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      if constexpr (!A_KC) if (want_rs) la.add_rowsum(rs);
      la.store(As(buf ^ 1));
      lb.store(Bs(buf ^ 1));
    }
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (!A_KC) {
    if (want_rs) {   // fold the per-thread row sums: threads sharing tid % (BM/4) hold the same four rows
      constexpr int kPerRow = BM / 4, kGroups = 256 / kPerRow;
      float* red = gemm_smem;   // the tiles are dead (the loop ended on a barrier)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[(threadIdx.x / kPerRow) * BM + (threadIdx.x % kPerRow) * 4 + i] = rs[i];
      __syncthreads();
      if (threadIdx.x < BM && m0 + threadIdx.x < g.M) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < kGroups; ++q) t += red[q * BM + threadIdx.x];
        g.rowsum[(long)blockIdx.z * g.M + m0 + threadIdx.x] = t;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int n = n0 + wn * 32 * WTN + j * 32 + (lane & 31);
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * WTM + i * 32 + acc_row(r, lane >> 5);
        if (m < g.M) {
          float* p = C + (long)m * g.ldc + n;
          float v = acc[i][j][r] + bv;
          if (g.accumulate) v += *p;
          *p = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16-operand variant (v_mfma_f32_32x32x16_bf16, fp32 accumulate): same contract and fp32 sources / outputs as
// sgemm_kernel; operands are rounded to bf16 while they are staged into LDS.  LDS image is [m][k] with k contiguous
// (row = kBKH bf16 + 16 B pad: the 32 rows of a ds_read_b128 fragment read hit 16 distinct 16-byte slots).
// A source that is contiguous along m (the token-major activations of a weight-gradient GEMM) is transposed in
// registers: each thread gathers 8 consecutive k for its m and writes one 16-byte row segment.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kBKH = 32;          // k per staged tile of the bf16 kernel (64 measured slower: 8.7 vs 8.4 ms per base-DiT step)
constexpr int kKG = kBKH / 8;      // 8-element k groups per row
constexpr int kLDH = kBKH + 8;     // bf16 elements per LDS row (144 B: fragment rows fall on distinct 16-byte slots)
template <int WTM, int WTN>
constexpr int hgemm_smem_bytes() { return 2 * (64 * WTM + 64 * WTN) * kLDH * 2; }

template <int BM, bool KC>
struct TileLoaderBF {
  static constexpr int kFPT = BM / 64;            // MC: rows (features) per thread
  static constexpr int kSlots = BM * kKG / 256;   // 8-element row segments per thread (KC: items; MC: kFPT rows x kKG/4 k groups)
  float f[kSlots][8];                             // raw fp32 (converted in store(), so the loads stay in flight during the MFMAs)
  __device__ __forceinline__ void load(const float* __restrict__ P, long sm, long sk, int m0, int M, int k0, int kend) {
    const int tid = threadIdx.x;
    const bool inside = m0 + BM <= M && k0 + kBKH <= kend;   // workgroup-uniform: interior tiles skip every bounds test
    if constexpr (KC) {
#pragma unroll
      for (int j = 0; j < kSlots; ++j) {
        const int item = tid + 256 * j, m = m0 + item / kKG, k = k0 + (item % kKG) * 8;
        const float* p = P + (long)m * sm + k;
        if (inside || (m < M && k + 7 < kend)) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { f[j][i] = lo[i]; f[j][4 + i] = hi[i]; }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[j][i] = (m < M && k + i < kend) ? p[i] : 0.f;
        }
      }
    } else {
      const int m = m0 + (tid & 63) * kFPT;
#pragma unroll
      for (int g = 0; g < kKG / 4; ++g) {
        const int k = k0 + ((tid >> 6) + 4 * g) * 8;
        const float* p = P + (long)k * sk + m;
        if (inside) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if constexpr (kFPT == 2) {
              const auto v2 = *reinterpret_cast<const __attribute__((ext_vector_type(2))) float*>(p + (long)i * sk);
              f[g * kFPT][i] = v2[0];
              f[g * kFPT + 1][i] = v2[1];
            } else {
              f[g][i] = p[(long)i * sk];
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < kFPT; ++e) f[g * kFPT + e][i] = (k + i < kend && m + e < M) ? p[(long)i * sk + e] : 0.f;
        }
      }
    }
  }
  // MC only: rs[e] += this tile's values of row (tid & 63)*kFPT + e (fp32, before the bf16 rounding)
  __device__ __forceinline__ void add_rowsum(float (&rs)[4]) const {
#pragma unroll
    for (int g = 0; g < kKG / 4; ++g)
#pragma unroll
      for (int e = 0; e < kFPT; ++e)
#pragma unroll
        for (int i = 0; i < 8; ++i) rs[e] += f[g * kFPT + e][i];
  }
  // E = __bf16 or _Float16: the 16-bit type the operands are rounded to while staged
  template <typename E>
  __device__ __forceinline__ void store(E* __restrict__ S /* [BM][kLDH] */) const {
    typedef __attribute__((ext_vector_type(8))) E Ex8;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
      Ex8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (E)f[j][i];
      if constexpr (KC) {
        const int item = tid + 256 * j;
        *reinterpret_cast<Ex8*>(S + (item / kKG) * kLDH + (item % kKG) * 8) = v;
      } else {
        const int g = j / kFPT, e = j % kFPT;
        *reinterpret_cast<Ex8*>(S + ((tid & 63) * kFPT + e) * kLDH + ((tid >> 6) + 4 * g) * 8) = v;
      }
    }
  }
};

// F16: fp16 operands (10 mantissa bits = the reference's TF32 class; v_mfma_f32_32x32x16_f16) instead of bf16 - same rate, same staging
template <int WTM, int WTN, bool A_KC, bool B_KC, bool F16 = false>
__global__ __launch_bounds__(256) void hgemm_kernel(GemmArgs g) {
  constexpr int BM = 64 * WTM, BN = 64 * WTN;
  using E = typename std::conditional<F16, _Float16, __bf16>::type;
  typedef __attribute__((ext_vector_type(8))) E Ex8;
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
  E* const smem = reinterpret_cast<E*>(gemm_smem);
  auto As = [&](int b) { return smem + b * (BM * kLDH); };
  auto Bs = [&](int b) { return smem + 2 * BM * kLDH + b * (BN * kLDH); };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  float* __restrict__ C = g.C + (long)blockIdx.z * g.M * g.ldc;

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileLoaderBF<BM, A_KC> la;
  TileLoaderBF<BN, B_KC> lb;
  const bool want_rs = !A_KC && g.rowsum != nullptr && blockIdx.x == 0;   // workgroup-uniform
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  la.load(g.A, g.sam, g.sak, m0, g.M, kbeg, kend);
  lb.load(g.B, g.sbn, g.sbk, n0, g.N, kbeg, kend);
  if constexpr (!A_KC) if (want_rs) la.add_rowsum(rs);
  la.store(As(0));
  lb.store(Bs(0));
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kBKH) {
    const bool more = k0 + kBKH < kend;
    if (more) {
      la.load(g.A, g.sam, g.sak, m0, g.M, k0 + kBKH, kend);
      lb.load(g.B, g.sbn, g.sbk, n0, g.N, k0 + kBKH, kend);
    }
    const E* __restrict__ as = As(buf) + (wm * 32 * WTM + (lane & 31)) * kLDH + (lane >> 5) * 8;
    const E* __restrict__ bs = Bs(buf) + (wn * 32 * WTN + (lane & 31)) * kLDH + (lane >> 5) * 8;
#pragma unroll
    for (int kk = 0; kk < kBKH; kk += 16) {
      Ex8 a[WTM], b[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) a[i] = *reinterpret_cast<const Ex8*>(as + i * 32 * kLDH + kk);
#pragma unroll
      for (int j = 0; j < WTN; ++j) b[j] = *reinterpret_cast<const Ex8*>(bs + j * 32 * kLDH + kk);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
          if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    if (more) {
      if constexpr (!A_KC) if (want_rs) la.add_rowsum(rs);
      la.store(As(buf ^ 1));
      lb.store(Bs(buf ^ 1));
    }
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (!A_KC) {
    if (want_rs) {   // threads (tid & 63) of the four k-groups hold the same kFPT rows
      constexpr int kFPT = BM / 64;
      float* red = gemm_smem;
#pragma unroll
      for (int e = 0; e < kFPT; ++e) red[(threadIdx.x >> 6) * BM + (threadIdx.x & 63) * kFPT + e] = rs[e];
      __syncthreads();
      if (threadIdx.x < BM && m0 + threadIdx.x < g.M)
        g.rowsum[(long)blockIdx.z * g.M + m0 + threadIdx.x] =
            (red[threadIdx.x] + red[BM + threadIdx.x]) + (red[2 * BM + threadIdx.x] + red[3 * BM + threadIdx.x]);
    }
  }
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int n = n0 + wn * 32 * WTN + j * 32 + (lane & 31);
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * WTM + i * 32 + acc_row(r, lane >> 5);
        if (m < g.M) {
          float* p = C + (long)m * g.ldc + n;
          float v = acc[i][j][r] + bv;
          if (g.accumulate) v += *p;
          *p = v;
        }
      }
    }
}

// C[m][n] (+)= bias[n] + sum_z P[z][m][n]   (P has leading dimension N, C has ldc)
__global__ void reduce_partials_kernel(const float* __restrict__ P, int splits, int M, int N, float* __restrict__ C, long ldc,
                                       const float* __restrict__ bias, int accumulate) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i % N);
    float s = bias ? bias[n] : 0.f;
    for (int z = 0; z < splits; ++z) s += P[(long)z * total + i];
    float* p = C + (long)m * ldc + n;
    *p = accumulate ? *p + s : s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Elementwise / per-token kernels.  Width D = n_embed is a runtime multiple of 256 (one float4 per lane per 256
// features); 16 tokens per sample; heads of 32 or 64 features.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMaxNQ = 8;   // D <= 2048: float4 slots per lane of a token row (kernels are instantiated per D / 256)

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// TimestepEmbedder.timestep_embedding (layers.py:351-361): [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(1e4) k / 128)
__global__ void t_freq_kernel(const float* __restrict__ t, int n, float* __restrict__ out) {
  const int i = blockIdx.x, k = threadIdx.x & 127;
  if (i >= n) return;
  const float f = expf(-9.210340371976184f * (float)k / 128.0f);
  const float a = t[i] * f;
  out[(long)i * 256 + threadIdx.x] = threadIdx.x < 128 ? cosf(a) : sinf(a);
}

template <typename TO = float>
__global__ void silu_kernel(const float* __restrict__ x, TO* __restrict__ y, long count) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = (TO)(v * sigmoid_f(v));
  }
}
// dst = bf16(src), count % 4 == 0 (the adaLN vectors' gradient before the stacked bf16-source GEMMs)
__global__ void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long count) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4; i < count; i += (long)gridDim.x * blockDim.x * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
    bf16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x4_t*>(dst + i) = o;
  }
}
// dx = dy * silu'(x)
__global__ void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long count) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i], s = sigmoid_f(v);
    dx[i] = dy[i] * s * (1.0f + v * (1.0f - s));
  }
}

// c[i] = temb[i] + sum_classes table_c[label_c[i] or null_c]     (nnets.py:283-288,380-456; labels already carry
// the dropout decisions; labels[c] == NULL selects the null token for every row).  grid (n, D/256).
struct EmbedArgs {
  const float* table[SCLDM_MAX_CLASSES];
  const int64_t* labels[SCLDM_MAX_CLASSES];
  int vocab[SCLDM_MAX_CLASSES];
  int n_classes;
};
__global__ void cond_sum_kernel(const float* __restrict__ temb, EmbedArgs e, int n, int D, float* __restrict__ c) {
  const int i = blockIdx.x, f = blockIdx.y * 256 + threadIdx.x;
  if (i >= n) return;
  float v = temb[(long)i * D + f];
  for (int k = 0; k < e.n_classes; ++k) {
    long row = e.labels[k] ? (long)e.labels[k][i] : (long)e.vocab[k];
    row = row < 0 ? 0 : (row > e.vocab[k] ? e.vocab[k] : row);
    v += e.table[k][row * D + f];
  }
  c[(long)i * D + f] = v;
}
// d table_c[row] = sum over samples whose label is `row` of dc  (deterministic: one workgroup per (table row, 256-feature
// slice); the 256 threads test 256 labels at a time and exchange wave ballots through LDS, then walk the matches in order)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dc, const int64_t* __restrict__ labels, int vocab,
                                                        int n, int D, float* __restrict__ dtable) {
  __shared__ unsigned long long hit[4];
  const int row = blockIdx.x, tid = threadIdx.x, f = blockIdx.y * 256 + tid, wave = tid >> 6;
  float s = 0.f;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    long l = vocab;
    if (i < n && labels) l = (long)labels[i];
    l = l < 0 ? 0 : (l > vocab ? vocab : l);
    const unsigned long long m = __ballot(i < n && l == row);
    if ((tid & 63) == 0) hit[wave] = m;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned long long mm = hit[w];
      while (mm) {   // up to 32 matches per trip: the loads are independent (a label shared by most of the batch - the null token
        float v[32];  // under label dropout - is one long latency-bound chain otherwise), the adds keep sample order
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int bit = mm ? __ffsll((long long)mm) - 1 : -1;
          mm &= mm - 1;
          v[j] = bit >= 0 ? dc[(long)(i0 + w * 64 + bit) * D + f] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) s += v[j];
      }
    }
    __syncthreads();
  }
  dtable[(long)row * D + f] = s;
}

// Same result (same summation order: ascending sample index), organised by SAMPLE instead of by table row: the table is zeroed
// first (hipMemsetAsync), workgroup i exits unless sample i is the first one carrying its label, and then sums that label's
// samples.  With n samples << table rows (replogle: 1 024 cells, 2 025 gene rows) the row-wise kernel above spends its time
// on 2 025 workgroups that each scan every label to find nothing.  grid (n, D/256).
__global__ __launch_bounds__(256) void embed_bwd_by_sample_kernel(const float* __restrict__ dc, const int64_t* __restrict__ labels,
                                                                  int vocab, int n, int D, float* __restrict__ dtable) {
  __shared__ unsigned long long hit[4];
  __shared__ int earlier;
  const int i = blockIdx.x, tid = threadIdx.x, f = blockIdx.y * 256 + tid, wave = tid >> 6;
  auto label_of = [&](int s) {
    long l = labels ? (long)labels[s] : (long)vocab;
    return l < 0 ? 0L : (l > vocab ? (long)vocab : l);
  };
  const long mine = label_of(i);
  if (tid == 0) earlier = 0;
  __syncthreads();
  bool found = false;
  for (int s = tid; s < i; s += 256) found |= label_of(s) == mine;
  if (found) earlier = 1;   // benign race: every writer stores 1
  __syncthreads();
  if (earlier) return;
  float sum = 0.f;
  for (int i0 = (i / 256) * 256; i0 < n; i0 += 256) {
    const int s = i0 + tid;
    const unsigned long long m = __ballot(s >= i && s < n && label_of(s) == mine);
    if ((tid & 63) == 0) hit[wave] = m;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned long long mm = hit[w];
      while (mm) {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int bit = mm ? __ffsll((long long)mm) - 1 : -1;
          mm &= mm - 1;
          v[j] = bit >= 0 ? dc[(long)(i0 + w * 64 + bit) * D + f] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) sum += v[j];
      }
    }
    __syncthreads();
  }
  dtable[mine * D + f] = sum;
}

// Two-level variant for batches where ONE label can cover most of the samples (the null token under label dropout: 80 % of
// 1 024 cells): the sums above are one latency-bound chain per label.  Level 1: workgroup i, if sample i is the first of its
// label inside its 64-sample segment, sums the segment's samples of that label into part[i].  Level 2: workgroup i, if sample i
// is the first of its label in the whole batch, adds the partials of the segments' first occurrences in segment order.
// Deterministic (fixed order), but the order is segment-wise: the bits differ from the single-level kernels'.
constexpr int kEmbSeg = 64;
constexpr int kEmbMaxSegs = 1024;   // n <= 65 536
__device__ __forceinline__ long emb_label(const int64_t* __restrict__ labels, int s, int vocab) {
  const long l = labels ? (long)labels[s] : (long)vocab;
  return l < 0 ? 0L : (l > vocab ? (long)vocab : l);
}
__global__ __launch_bounds__(256) void embed_bwd_seg_partial_kernel(const float* __restrict__ dc, const int64_t* __restrict__ labels,
                                                                    int vocab, int n, int D, float* __restrict__ part) {
  __shared__ unsigned long long seg_mask;
  const int i = blockIdx.x, tid = threadIdx.x, f = blockIdx.y * 256 + tid, seg0 = (i / kEmbSeg) * kEmbSeg;
  const long mine = emb_label(labels, i, vocab);
  if (tid < 64) {
    const int s = seg0 + tid;
    const unsigned long long m = __ballot(s < n && emb_label(labels, s, vocab) == mine);
    if (tid == 0) seg_mask = m;
  }
  __syncthreads();
  unsigned long long mm = seg_mask;
  if (mm & ((1ull << (i - seg0)) - 1ull)) return;   // an earlier sample of this segment owns the label
  float sum = 0.f;
  while (mm) {
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int bit = mm ? __ffsll((long long)mm) - 1 : -1;
      mm &= mm - 1;
      v[j] = bit >= 0 ? dc[(long)(seg0 + bit) * D + f] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) sum += v[j];
  }
  part[(long)i * D + f] = sum;
}
__global__ __launch_bounds__(256) void embed_bwd_seg_final_kernel(const float* __restrict__ part, const int64_t* __restrict__ labels,
                                                                  int vocab, int n, int D, float* __restrict__ dtable) {
  __shared__ int first_in_seg[kEmbMaxSegs];
  __shared__ int earlier;
  const int i = blockIdx.x, tid = threadIdx.x, f = blockIdx.y * 256 + tid, wave = tid >> 6, lane = tid & 63;
  const long mine = emb_label(labels, i, vocab);
  if (tid == 0) earlier = 0;
  __syncthreads();
  bool found = false;
  for (int s = tid; s < i; s += 256) found |= emb_label(labels, s, vocab) == mine;
  if (found) earlier = 1;
  __syncthreads();
  if (earlier) return;
  const int nseg = (n + kEmbSeg - 1) / kEmbSeg, seg_i = i / kEmbSeg;
  for (int g0 = seg_i; g0 < nseg; g0 += 4) {   // four segments per trip, one per wave
    const int g = g0 + wave, s = g * kEmbSeg + lane;
    const unsigned long long m = __ballot(g < nseg && s < n && emb_label(labels, s, vocab) == mine);
    if (lane == 0 && g < nseg) first_in_seg[g] = m ? g * kEmbSeg + (__ffsll((long long)m) - 1) : -1;
  }
  __syncthreads();
  float sum = 0.f;
  for (int g0 = seg_i; g0 < nseg; g0 += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int g = g0 + j, src = g < nseg ? first_in_seg[g] : -1;
      v[j] = src >= 0 ? part[(long)src * D + f] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += v[j];
  }
  dtable[mine * D + f] = sum;
}

// x[t][f] += pos[t % 16][f]
__global__ void add_pos_kernel(float* __restrict__ x, const float* __restrict__ pos, long tokens, int D) {
  const long total = tokens * D, period = (long)kS * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) x[i] += pos[i % period];
}

// Arrays that only ever feed GEMMs (h1, ao, h2, hid and dy, dqkv, da, db) are written as bf16 when the step runs the
// bf16-source GEMM (bgemm.hpp): TO = float or __bf16 on their producers.
template <typename TO>
__device__ __forceinline__ void put4(TO* __restrict__ p, const f32x4& v) {
  if constexpr (sizeof(TO) == 4) {
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
    bf16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x4_t*>(p) = o;
  }
}

// h = LN(x) * (1 + scale[b]) + shift[b], LayerNorm without affine (nnets.py:257-258), one wave per token.
// stats[t] = (mean, rstd).  mod row of sample b at mod + b*mod_stride; scale/shift at column offsets.
// Optional fused residual (y != nullptr): x_new = x + gate[b] * y is formed first (the gate_res step of the previous branch:
// Block.forward, layers.py:219-222), written to x_out, and normalised from registers - the row is read once instead of twice.
template <int NQ, typename TO = float, typename TY = float>
__global__ __launch_bounds__(256) void ln_mod_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mod, long mod_stride,
                                                         int sc_off, int sh_off, float eps, long tokens, TO* __restrict__ h,
                                                         float* __restrict__ stats, const TY* __restrict__ y = nullptr, int g_off = 0,
                                                         float* __restrict__ x_out = nullptr) {
  constexpr int D = NQ * 256, nq = NQ, kMaxDQ = NQ;
  const int lane = threadIdx.x & 63;
  const long t = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (t >= tokens) return;
  f32x4 v[kMaxDQ];
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < kMaxDQ; ++q)
    if (q < nq) {
      v[q] = *reinterpret_cast<const f32x4*>(x + t * D + q * 256 + lane * 4);
      if (y) {
        f32x4 yv;
        if constexpr (sizeof(TY) == 4) {
          yv = *reinterpret_cast<const f32x4*>(y + t * D + q * 256 + lane * 4);
        } else {
          const bf16x4 yb = *reinterpret_cast<const bf16x4*>(y + t * D + q * 256 + lane * 4);
          yv = f32x4{(float)yb[0], (float)yb[1], (float)yb[2], (float)yb[3]};
        }
        const f32x4 gv = *reinterpret_cast<const f32x4*>(mod + (t / kS) * mod_stride + g_off + q * 256 + lane * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[q][i] += gv[i] * yv[i];
        *reinterpret_cast<f32x4*>(x_out + t * D + q * 256 + lane * 4) = v[q];
      }
      sum += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int q = 0; q < kMaxDQ; ++q)
    if (q < nq) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[q][i] -= mean;
        sq += v[q][i] * v[q][i];
      }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
  const float* m = mod + (t / kS) * mod_stride;
#pragma unroll
  for (int q = 0; q < kMaxDQ; ++q)
    if (q < nq) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(m + sc_off + q * 256 + lane * 4);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(m + sh_off + q * 256 + lane * 4);
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = v[q][i] * rstd * (1.0f + sc[i]) + sh[i];
      put4(h + t * D + q * 256 + lane * 4, o);
    }
  if (lane == 0) {
    stats[t * 2] = mean;
    stats[t * 2 + 1] = rstd;
  }
}

// Backward of ln_mod_fwd for one sample per workgroup (8 waves x 2 tokens):
//   g = dh * (1 + scale);  dx (+)= rstd * (g - mean_f(g) - xhat * mean_f(g * xhat))
//   dscale[b] = sum_t dh * xhat;  dshift[b] = sum_t dh      (written to dmod row b at the same column offsets)
// Round 3: HBM-bound by design (reads dh, x, dx; writes dx: 16 B per element) but it ran at 1.9 TB/s - four waves walking four
// tokens each, every token's loads issued only after the previous token's wave reductions.  Now eight waves, and a wave requests
// BOTH its tokens' rows (and the old dx when accumulating) before any arithmetic.
template <typename T>
__device__ __forceinline__ f32x4 load4f(const T* p) {
  if constexpr (sizeof(T) == 4) {
    return *reinterpret_cast<const f32x4*>(p);
  } else {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
}
template <typename T>
__device__ __forceinline__ void store4f(T* p, const f32x4 v) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
    *reinterpret_cast<bf16x4*>(p) = o;
  }
}
constexpr int kLnBwdWaves = 8, kLnBwdTok = kS / kLnBwdWaves;
template <int NQ, typename TD = float>
__global__ __launch_bounds__(64 * kLnBwdWaves) void ln_mod_bwd_kernel(const TD* __restrict__ dh, const float* __restrict__ x,
                                                         const float* __restrict__ stats, const float* __restrict__ mod,
                                                         long mod_stride, int sc_off, int sh_off, float* __restrict__ dx,
                                                         int accumulate_dx, float* __restrict__ dmod,
                                                         const __bf16* __restrict__ gy = nullptr, int g_off = 0, __bf16* __restrict__ gdy = nullptr) {
  // gy != nullptr: the gate backward of the branch that FOLLOWS in the backward order rides along (gate_bwd_kernel's work on the dx
  // rows this kernel has just produced: gdy = gate[b] * dx as bf16, dmod[g_off ..] = sum_t dx * gy) - dx is not read again
  __shared__ f32x4 red[3][kLnBwdWaves][64];
  constexpr int D = NQ * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long b = blockIdx.x;
  f32x4 sc[NQ], dsc[NQ], dsh[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    sc[q] = *reinterpret_cast<const f32x4*>(mod + b * mod_stride + sc_off + q * 256 + lane * 4);
    dsc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    dsh[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 xv[kLnBwdTok][NQ], dv[kLnBwdTok][NQ], ov[kLnBwdTok][NQ];
  f32x4 gate[NQ], dgate[NQ];
  bf16x4 yv[kLnBwdTok][NQ];
  if (gy) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      gate[q] = *reinterpret_cast<const f32x4*>(mod + b * mod_stride + g_off + q * 256 + lane * 4);
      dgate[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  float mean[kLnBwdTok], rstd[kLnBwdTok];
#pragma unroll
  for (int tt = 0; tt < kLnBwdTok; ++tt) {
    const long t = b * kS + wave * kLnBwdTok + tt;
    mean[tt] = stats[t * 2];
    rstd[tt] = stats[t * 2 + 1];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      xv[tt][q] = *reinterpret_cast<const f32x4*>(x + t * D + q * 256 + lane * 4);
      dv[tt][q] = load4f(dh + t * D + q * 256 + lane * 4);
      if (accumulate_dx) ov[tt][q] = *reinterpret_cast<const f32x4*>(dx + t * D + q * 256 + lane * 4);
      if (gy) yv[tt][q] = *reinterpret_cast<const bf16x4*>(gy + t * D + q * 256 + lane * 4);
    }
  }
#pragma unroll
  for (int tt = 0; tt < kLnBwdTok; ++tt) {
    const long t = b * kS + wave * kLnBwdTok + tt;
    f32x4 xh[NQ], g[NQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xh[q][i] = (xv[tt][q][i] - mean[tt]) * rstd[tt];
        g[q][i] = dv[tt][q][i] * (1.0f + sc[q][i]);
        s1 += g[q][i];
        s2 += g[q][i] * xh[q][i];
        dsc[q][i] += dv[tt][q][i] * xh[q][i];
        dsh[q][i] += dv[tt][q][i];
      }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = rstd[tt] * (g[q][i] - s1 - xh[q][i] * s2) + (accumulate_dx ? ov[tt][q][i] : 0.f);
      *reinterpret_cast<f32x4*>(dx + t * D + q * 256 + lane * 4) = o;
      if (gy) {
        bf16x4 dyv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dyv[i] = (__bf16)(gate[q][i] * o[i]);
          dgate[q][i] += o[i] * (float)yv[tt][q][i];
        }
        *reinterpret_cast<bf16x4*>(gdy + t * D + q * 256 + lane * 4) = dyv;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    red[0][wave][lane] = dsc[q];
    red[1][wave][lane] = dsh[q];
    if (gy) red[2][wave][lane] = dgate[q];
    __syncthreads();
    if (wave == 2 && gy) {
      f32x4 s = red[2][0][lane];
#pragma unroll
      for (int w = 1; w < kLnBwdWaves; ++w) {
        const f32x4 o = red[2][w][lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] += o[i];
      }
      *reinterpret_cast<f32x4*>(dmod + b * mod_stride + g_off + q * 256 + lane * 4) = s;
    }
    if (wave < 2) {
      f32x4 s = red[wave][0][lane];
#pragma unroll
      for (int w = 1; w < kLnBwdWaves; ++w) {
        const f32x4 o = red[wave][w][lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] += o[i];
      }
      *reinterpret_cast<f32x4*>(dmod + b * mod_stride + (wave == 0 ? sc_off : sh_off) + q * 256 + lane * 4) = s;
    }
    __syncthreads();
  }
}

// x_out = x + gate[b] * y
template <typename TY = float>
__global__ void gate_res_kernel(const float* __restrict__ x, const TY* __restrict__ y, const float* __restrict__ mod,
                                long mod_stride, int g_off, long tokens, int D, float* __restrict__ out) {
  const int dq = D / 4;
  const long total = tokens * dq;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i / dq;
    const int f = (int)(i % dq) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(mod + (t / kS) * mod_stride + g_off + f);
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + t * D + f), b = load4f(y + t * D + f);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = a[k] + g[k] * b[k];
    *reinterpret_cast<f32x4*>(out + t * D + f) = o;
  }
}
// dy = gate[b] * dx;  dgate[b] = sum_t dx * y        (grid (samples, D/256), thread = feature)
template <typename TO = float, typename TY = float>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dx, const TY* __restrict__ y,
                                                       const float* __restrict__ mod, long mod_stride, int g_off, int D,
                                                       TO* __restrict__ dy, float* __restrict__ dmod) {
  const long b = blockIdx.x;
  const int f = blockIdx.y * 256 + threadIdx.x;
  const float g = mod[b * mod_stride + g_off + f];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < kS; ++t) {
    const long i = (b * kS + t) * D + f;
    const float d = dx[i];
    dy[i] = (TO)(g * d);
    s += d * (float)y[i];
  }
  dmod[b * mod_stride + g_off + f] = s;
}

// hid = silu(a) * b   (MLP.forward, layers.py:172-174).  a, b: [tokens][H]; the output rows have ldo >= H elements, and the
// thread of a row's last element also zeroes the row's padding (bf16 rows are padded to 16-byte multiples for bgemm_kernel,
// whose last k chunk reads the padding of both operands)
// (round 3: FOUR consecutive hidden units per thread - H % 4 == 0 is a requirement of the training path - so one 8- / 16-byte load
// and store per array and one 32-bit index division per four elements; the scalar form spent its time in a 64-bit division per element)
template <typename TO = float, typename TI = float>
__global__ void swiglu_fwd_kernel(const TI* __restrict__ a, const TI* __restrict__ b, TO* __restrict__ hid, long count, int H, int ldo) {
  const unsigned H4 = (unsigned)H / 4u, n4 = (unsigned)(count / 4);
  for (unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += gridDim.x * blockDim.x) {
    const unsigned t = i4 / H4, c = (i4 - t * H4) * 4u;
    const f32x4 av = load4f(a + (size_t)i4 * 4), bv = load4f(b + (size_t)i4 * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = av[e] * sigmoid_f(av[e]) * bv[e];
    store4f(hid + (size_t)t * ldo + c, o);
    if (c + 4 == (unsigned)H)
      for (int p = H; p < ldo; ++p) hid[(size_t)t * ldo + p] = (TO)0.f;
  }
}
template <typename TO = float, typename TI = float, typename TD = float>
__global__ void swiglu_bwd_kernel(const TD* __restrict__ dhid, const TI* __restrict__ a, const TI* __restrict__ b,
                                  TO* __restrict__ da, TO* __restrict__ db, long count, int H, int ldo, int lpad) {
  // (rows of da / db are ldo apart and zero padded up to lpad elements: lpad == ldo for separate arrays, ldo == 2 lpad when the two
  // live side by side in one row - the k-concatenated operand of the merged MLP data gradient)
  const unsigned H4 = (unsigned)H / 4u, n4 = (unsigned)(count / 4);
  for (unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += gridDim.x * blockDim.x) {
    const unsigned t = i4 / H4, c = (i4 - t * H4) * 4u;
    const f32x4 av = load4f(a + (size_t)i4 * 4), bv = load4f(b + (size_t)i4 * 4), dv = load4f(dhid + (size_t)i4 * 4);
    f32x4 oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s = sigmoid_f(av[e]);
      oa[e] = dv[e] * bv[e] * s * (1.0f + av[e] * (1.0f - s));
      ob[e] = dv[e] * av[e] * s;
    }
    store4f(da + (size_t)t * ldo + c, oa);
    store4f(db + (size_t)t * ldo + c, ob);
    if (c + 4 == (unsigned)H)
      for (int p = H; p < lpad; ++p) da[(size_t)t * ldo + p] = db[(size_t)t * ldo + p] = (TO)0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Self-attention over the 16 tokens of one sample, one wave per (sample, head); qkv rows are [q | k | v] (each D wide)
// with head h at columns h*HD (layers.py:147-151).  softmax(q k^T / sqrt(HD)) v.  Nothing is saved: the backward
// recomputes P.  HD = 32 or 64; lane -> (token = lane / 4, HD/4 consecutive columns).
// ---------------------------------------------------------------------------------------------------------------
template <int HD>
struct AttnTile {
  float q[kS][HD + 1], k[kS][HD + 1], v[kS][HD + 1], p[kS][kS + 1];
};
template <int HD, typename TI>
__device__ __forceinline__ void attn_load(AttnTile<HD>& s, const TI* __restrict__ qkv, long sample, int head, int D, int lane) {
  constexpr int CW = HD / 4;
  const int tok = lane >> 2, c0 = (lane & 3) * CW;
  const TI* base = qkv + (sample * kS + tok) * (3L * D) + head * HD + c0;
#pragma unroll
  for (int i = 0; i < CW; ++i) {
    s.q[tok][c0 + i] = (float)base[i];
    s.k[tok][c0 + i] = (float)base[D + i];
    s.v[tok][c0 + i] = (float)base[2 * D + i];
  }
}
// lane -> (i = lane / 4, j in {j0..j0+3}, j0 = (lane & 3) * 4): returns this lane's 4 probabilities of row i
template <int HD>
__device__ __forceinline__ void attn_probs(const AttnTile<HD>& s, int lane, float* p4) {
  const int i = lane >> 2, j0 = (lane & 3) * 4;
  const float scale = HD == 32 ? 0.17677669529663687f : 0.125f;  // 1/sqrt(HD)
  float mx = -INFINITY;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc += s.q[i][d] * s.k[j0 + jj][d];
    p4[jj] = acc * scale;
    mx = fmaxf(mx, p4[jj]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  float sum = 0.f;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    p4[jj] = __expf(p4[jj] - mx);
    sum += p4[jj];
  }
  sum += __shfl_xor(sum, 1);
  sum += __shfl_xor(sum, 2);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) p4[jj] *= inv;
}

template <int HD>
constexpr int attn_waves() { return HD == 32 ? 4 : 2; }   // waves (= (sample, head) units) per workgroup: LDS-limited for HD 64

template <int HD, typename TO = float, typename TI = float>
__global__ __launch_bounds__(64 * attn_waves<HD>()) void attn_fwd_kernel(const TI* __restrict__ qkv, long n_samples, int n_head, int D,
                                                                         TO* __restrict__ ao) {
  __shared__ AttnTile<HD> tiles[attn_waves<HD>()];
  constexpr int CW = HD / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long unit = blockIdx.x * (long)attn_waves<HD>() + wave;   // (sample, head)
  if (unit >= n_samples * n_head) return;
  const long sample = unit / n_head;
  const int head = (int)(unit % n_head);
  AttnTile<HD>& s = tiles[wave];
  attn_load<HD, TI>(s, qkv, sample, head, D, lane);
  __builtin_amdgcn_wave_barrier();
  float p4[4];
  attn_probs<HD>(s, lane, p4);
  const int i = lane >> 2, j0 = (lane & 3) * 4;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) s.p[i][j0 + jj] = p4[jj];
  __builtin_amdgcn_wave_barrier();
  const int d0 = (lane & 3) * CW;
  float o[CW];
#pragma unroll
  for (int d = 0; d < CW; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < kS; ++j) {
    const float pj = s.p[i][j];
#pragma unroll
    for (int d = 0; d < CW; ++d) o[d] += pj * s.v[j][d0 + d];
  }
  TO* out = ao + (sample * kS + i) * (long)D + head * HD + d0;
#pragma unroll
  for (int d = 0; d < CW; d += 4) put4(out + d, f32x4{o[d], o[d + 1], o[d + 2], o[d + 3]});
}

// dqkv from (qkv, dao):  dP = dao v^T;  dS = P * (dP - rowsum(P * dP));  dq = scale dS k;  dk = scale dS^T q;  dv = P^T dao
template <int HD, typename TO = float, typename TI = float>
__global__ __launch_bounds__(64 * attn_waves<HD>()) void attn_bwd_kernel(const TI* __restrict__ qkv, const float* __restrict__ dao,
                                                                         long n_samples, int n_head, int D, TO* __restrict__ dqkv) {
  __shared__ AttnTile<HD> tiles[attn_waves<HD>()];
  __shared__ float dos[attn_waves<HD>()][kS][HD + 1];
  __shared__ float dss[attn_waves<HD>()][kS][kS + 1];
  constexpr int CW = HD / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long unit = blockIdx.x * (long)attn_waves<HD>() + wave;
  if (unit >= n_samples * n_head) return;
  const long sample = unit / n_head;
  const int head = (int)(unit % n_head);
  AttnTile<HD>& s = tiles[wave];
  attn_load<HD, TI>(s, qkv, sample, head, D, lane);
  const int i = lane >> 2, c0 = (lane & 3) * CW, j0 = (lane & 3) * 4;
  {
    const float* src = dao + (sample * kS + i) * (long)D + head * HD + c0;
#pragma unroll
    for (int d = 0; d < CW; ++d) dos[wave][i][c0 + d] = src[d];
  }
  __builtin_amdgcn_wave_barrier();
  float p4[4], dp4[4];
  attn_probs<HD>(s, lane, p4);
  float dot = 0.f;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc += dos[wave][i][d] * s.v[j0 + jj][d];
    dp4[jj] = acc;
    dot += p4[jj] * acc;
  }
  dot += __shfl_xor(dot, 1);
  dot += __shfl_xor(dot, 2);
  const float scale = HD == 32 ? 0.17677669529663687f : 0.125f;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    s.p[i][j0 + jj] = p4[jj];
    dss[wave][i][j0 + jj] = p4[jj] * (dp4[jj] - dot) * scale;
  }
  __builtin_amdgcn_wave_barrier();
  // this lane now owns row `i` (as query row for dq, as key row for dk / dv) and CW columns starting at c0
  float dq[CW], dk[CW], dv[CW];
#pragma unroll
  for (int d = 0; d < CW; ++d) dq[d] = dk[d] = dv[d] = 0.f;
#pragma unroll
  for (int j = 0; j < kS; ++j) {
    const float ds_ij = dss[wave][i][j], ds_ji = dss[wave][j][i], p_ji = s.p[j][i];
#pragma unroll
    for (int d = 0; d < CW; ++d) {
      dq[d] += ds_ij * s.k[j][c0 + d];
      dk[d] += ds_ji * s.q[j][c0 + d];
      dv[d] += p_ji * dos[wave][j][c0 + d];
    }
  }
  TO* dst = dqkv + (sample * kS + i) * (3L * D) + head * HD + c0;
#pragma unroll
  for (int d = 0; d < CW; d += 4) {
    put4(dst + d, f32x4{dq[d], dq[d + 1], dq[d + 2], dq[d + 3]});
    put4(dst + D + d, f32x4{dk[d], dk[d + 1], dk[d + 2], dk[d + 3]});
    put4(dst + 2 * D + d, f32x4{dv[d], dv[d + 1], dv[d + 2], dv[d + 3]});
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Column sums (bias gradients, pos_embed-style reductions): out[c] = sum_r X[r*ld + c], two deterministic stages.
// Stage 1: grid (ceil(cols/64), splits), 256 threads = 64 columns x 4 row phases.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, long rows, int cols, long ld,
                                                             float* __restrict__ part /* [splits][cols] */) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const long per = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float s = 0.f;
  if (c < cols)
    for (long r = r0 + ph; r < r1; r += 4) s += X[r * ld + c];
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && c < cols) part[(long)blockIdx.y * cols + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int splits, int cols, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += part[(long)z * cols + c];
  out[c] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// Flow-matching interpolation and loss around the model call (transport.py:110-150, path.py:148-151, utils.py:15-17).
// ---------------------------------------------------------------------------------------------------------------
__global__ void fm_mix_kernel(const float* __restrict__ x1, const float* __restrict__ x0, const float* __restrict__ t,
                              float* __restrict__ xt, float* __restrict__ ut, long total, int e) {
#pragma clang fp contract(off)   // te * x1 + (1 - te) * x0 with every intermediate rounded (no fused multiply-add): the eager reference's bits
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float tv = t[i / e], a = x1[i], b = x0[i];
    const float p1 = tv * a, om = 1.0f - tv;
    const float p0 = om * b;
    xt[i] = p1 + p0;
    ut[i] = a - b;
  }
}
// one workgroup per sample: loss[b] = mean_e (pred - ut)^2 (fixed-order tree: deterministic)
__global__ __launch_bounds__(256) void fm_loss_kernel(const float* __restrict__ pred, const float* __restrict__ ut, float* __restrict__ loss, int e) {
  __shared__ float red[4];
  const long b = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < e; i += 256) {
    const float d = pred[b * e + i] - ut[b * e + i];
    s = fmaf(d, d, s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[b] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)e;
}
__global__ void fm_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ ut, const float* __restrict__ gloss,
                                   float* __restrict__ dpred, long total, int e) {
  const float k = 2.0f / (float)e;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    dpred[i] = gloss[i / e] * k * (pred[i] - ut[i]);
}

}  // namespace train
}  // namespace scldm
