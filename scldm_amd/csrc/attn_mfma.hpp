// Self-attention over the 16 tokens of one sample on the matrix cores, for the bf16-array route of the generic training path
// (qkv, ao, dqkv are bf16 arrays there: train_api.hip).  One wave per (sample, head); head_dim 32 or 64; layers.py:147-151.
//
// attn_fwd_kernel / attn_bwd_kernel (train.hpp) keep q, k, v (and dao) of a unit as fp32 tiles in LDS and contract on the VALU:
// 54 / 125 us per DiT-L layer at 1 024 cells against 27 / 53 us of HBM time.  Here a unit's q, k, v (dao) rows are staged once as
// bf16 [16][head_dim] images and every contraction is an MFMA:
//   * over head_dim (scores, dP): v_mfma_f32_16x16x32_bf16, both operands read along their rows (ds_read_b128);
//   * over the 16 tokens (P V, dS K, dS^T Q, P^T dO): v_mfma_f32_16x16x16_bf16 with the A operand taken straight from the
//     accumulator of the previous product and the B operand (a row-major [token][d] image) through ds_read_b64_tr_b16.
// The accumulator of a 16 x 16 product holds (row 4 g + r, column lane % 16) in register r of lane 16 g + lane % 16 - which is
// the A-operand layout (row lane % 16, k = 4 g + r) of the TRANSPOSED matrix.  So the scores are computed in both orientations
// (S^T = K Q^T gives P as the A operand of P V and dS as that of dS K; S = Q K^T gives P^T and dS^T as the A operands of P^T dO
// and dS^T Q): two tiny MFMAs more instead of a transposition through LDS.  Softmax statistics are taken in the S^T orientation
// (a query's 16 scores sit in 4 registers x the 4 lanes 16 apart) and fetched by lane for the other one.
// Results go back through the unit's LDS images so that a lane stores 16 contiguous bytes.
#pragma once
#include "common.hpp"

namespace scldm {
namespace train {

constexpr int kAttnMfmaWaves = 4;

template <int HD>
struct AttnMfmaTile {
  static constexpr int LD = HD + 8;   // bf16 per row: 16-byte pad (the 16 rows of a b128 read group spread over the banks)
  __bf16 q[16][LD], k[16][LD], v[16][LD], d[16][LD];
};

typedef __attribute__((ext_vector_type(4))) short am_s16x4;

// a [16][HD] block of a bf16 array with row stride ld -> LDS image (lane: row lane / 4, 32-byte quarter lane % 4 of a 64-wide row;
// half rows for head_dim 32)
template <int HD>
__device__ __forceinline__ void am_stage_bf16(__bf16 (*img)[HD + 8], const __bf16* __restrict__ src, long ld, int lane) {
  constexpr int PER = HD / 4;   // elements per lane
  const int row = lane >> 2, c0 = (lane & 3) * PER;
  const __bf16* p = src + row * ld + c0;
#pragma unroll
  for (int e = 0; e < PER; e += 8) *reinterpret_cast<bf16x8*>(&img[row][c0 + e]) = *reinterpret_cast<const bf16x8*>(p + e);
}
template <int HD>
__device__ __forceinline__ void am_stage_f32(__bf16 (*img)[HD + 8], const float* __restrict__ src, long ld, int lane) {
  constexpr int PER = HD / 4;
  const int row = lane >> 2, c0 = (lane & 3) * PER;
  const float* p = src + row * ld + c0;
#pragma unroll
  for (int e = 0; e < PER; e += 8) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + e), b = *reinterpret_cast<const f32x4*>(p + e + 4);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i] = (__bf16)a[i];
      o[4 + i] = (__bf16)b[i];
    }
    *reinterpret_cast<bf16x8*>(&img[row][c0 + e]) = o;
  }
}
// LDS image -> a [16][HD] block of a bf16 array (16 contiguous bytes per lane and piece)
template <int HD>
__device__ __forceinline__ void am_unstage(const __bf16 (*img)[HD + 8], __bf16* __restrict__ dst, long ld, int lane) {
  constexpr int PER = HD / 4;
  const int row = lane >> 2, c0 = (lane & 3) * PER;
#pragma unroll
  for (int e = 0; e < PER; e += 8) *reinterpret_cast<bf16x8*>(dst + row * ld + c0 + e) = *reinterpret_cast<const bf16x8*>(&img[row][c0 + e]);
}

// C (+)= X Y^T over head_dim: X, Y row-major [16][HD] images; result (row of X = 4 g + r, row of Y = lane % 16)
template <int HD>
__device__ __forceinline__ f32x4 am_dot_hd(const __bf16 (*X)[HD + 8], const __bf16 (*Y)[HD + 8], int lane) {
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  const int row = lane & 15, g = lane >> 4;
#pragma unroll
  for (int kc = 0; kc < HD; kc += 32) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&X[row][kc + 8 * g]);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Y[row][kc + 8 * g]);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  return c;
}
// the B operand (k = token 4 g + e, column 16 dt + lane % 16) of a row-major [token][d] image
template <int HD>
__device__ __forceinline__ am_s16x4 am_tr_b(const __bf16 (*img)[HD + 8], int dt, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const __bf16* p = &img[4 * g + (i >> 2)][16 * dt + 4 * (i & 3)];
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_s16x4*)p);
}
__device__ __forceinline__ am_s16x4 am_pack4(const f32x4& v) {
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
  return __builtin_bit_cast(am_s16x4, o);
}
// sum / max over the 16 values of a query that sit in 4 registers x the lanes {i, i + 16, i + 32, i + 48}
__device__ __forceinline__ float am_sum4(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float am_max4(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

template <int HD>
__global__ __launch_bounds__(64 * kAttnMfmaWaves) void attn_fwd_mfma_kernel(const __bf16* __restrict__ qkv, long n_samples, int n_head, int D,
                                                                            __bf16* __restrict__ ao) {
  __shared__ __attribute__((aligned(16))) AttnMfmaTile<HD> tiles[kAttnMfmaWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long unit = blockIdx.x * (long)kAttnMfmaWaves + wave;
  if (unit >= n_samples * n_head) return;
  const long sample = unit / n_head;
  const int head = (int)(unit % n_head);
  AttnMfmaTile<HD>& s = tiles[wave];
  const __bf16* base = qkv + sample * 16 * (3L * D) + head * HD;
  am_stage_bf16<HD>(s.q, base, 3L * D, lane);
  am_stage_bf16<HD>(s.k, base + D, 3L * D, lane);
  am_stage_bf16<HD>(s.v, base + 2 * D, 3L * D, lane);
  __builtin_amdgcn_wave_barrier();
  // S^T = K Q^T: register r of lane (g, i) = score of query i against key 4 g + r
  f32x4 st = am_dot_hd<HD>(s.k, s.q, lane);
  const float sl2 = (HD == 32 ? 0.17677669529663687f : 0.125f) * 1.4426950408889634f;
  const float m = am_max4(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])));
  f32x4 p;
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f((st[r] - m) * sl2);
  const float inv = __builtin_amdgcn_rcpf(am_sum4((p[0] + p[1]) + (p[2] + p[3])));
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r] *= inv;
  const am_s16x4 pa = am_pack4(p);   // A operand: row = query lane % 16, k = key 4 g + r
  // O = P V: (query 4 g + r, column 16 dt + lane % 16) -> the q image (free now), then 16-byte stores
  f32x4 o[HD / 16];
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, am_tr_b<HD>(s.v, dt, lane), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s.q[4 * (lane >> 4) + r][16 * dt + (lane & 15)] = (__bf16)o[dt][r];
  __builtin_amdgcn_wave_barrier();
  am_unstage<HD>(s.q, ao + sample * 16 * (long)D + head * HD, D, lane);
}

// dqkv from (qkv, dao):  dP = dao v^T;  dS = P * (dP - rowsum(P * dP)) / sqrt(hd);  dq = dS k;  dk = dS^T q;  dv = P^T dao
template <int HD, typename TD = float>
__global__ __launch_bounds__(64 * kAttnMfmaWaves) void attn_bwd_mfma_kernel(const __bf16* __restrict__ qkv, const TD* __restrict__ dao,
                                                                            long n_samples, int n_head, int D, __bf16* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) AttnMfmaTile<HD> tiles[kAttnMfmaWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long unit = blockIdx.x * (long)kAttnMfmaWaves + wave;
  if (unit >= n_samples * n_head) return;
  const long sample = unit / n_head;
  const int head = (int)(unit % n_head);
  AttnMfmaTile<HD>& s = tiles[wave];
  const __bf16* base = qkv + sample * 16 * (3L * D) + head * HD;
  am_stage_bf16<HD>(s.q, base, 3L * D, lane);
  am_stage_bf16<HD>(s.k, base + D, 3L * D, lane);
  am_stage_bf16<HD>(s.v, base + 2 * D, 3L * D, lane);
  if constexpr (sizeof(TD) == 2) am_stage_bf16<HD>(s.d, dao + sample * 16 * (long)D + head * HD, D, lane);   // (dao written as bf16 by its data gradient)
  else am_stage_f32<HD>(s.d, dao + sample * 16 * (long)D + head * HD, D, lane);
  __builtin_amdgcn_wave_barrier();
  const int g = lane >> 4, i = lane & 15;
  const float scale = HD == 32 ? 0.17677669529663687f : 0.125f, sl2 = scale * 1.4426950408889634f;
  // orientation T: lane = query i, registers = keys 4 g + r.  orientation N: lane = key j, registers = queries 4 g + r.
  const f32x4 st = am_dot_hd<HD>(s.k, s.q, lane), sn = am_dot_hd<HD>(s.q, s.k, lane);
  const f32x4 dpt = am_dot_hd<HD>(s.v, s.d, lane), dpn = am_dot_hd<HD>(s.d, s.v, lane);
  const float m = am_max4(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])));
  f32x4 pt;
#pragma unroll
  for (int r = 0; r < 4; ++r) pt[r] = __builtin_amdgcn_exp2f((st[r] - m) * sl2);
  const float inv = __builtin_amdgcn_rcpf(am_sum4((pt[0] + pt[1]) + (pt[2] + pt[3])));
  float dot = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pt[r] *= inv;
    dot += pt[r] * dpt[r];
  }
  dot = am_sum4(dot);
  f32x4 dst_, pn, dsn;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dst_[r] = pt[r] * (dpt[r] - dot) * scale;
    // the statistics of query 4 g + r live in the lanes whose lane % 16 is that query
    const float mq = __shfl(m, 4 * g + r), iq = __shfl(inv, 4 * g + r), dq_ = __shfl(dot, 4 * g + r);
    pn[r] = __builtin_amdgcn_exp2f((sn[r] - mq) * sl2) * iq;
    dsn[r] = pn[r] * (dpn[r] - dq_) * scale;
  }
  const am_s16x4 a_dst = am_pack4(dst_), a_pn = am_pack4(pn), a_dsn = am_pack4(dsn);
  f32x4 dq[HD / 16], dk[HD / 16], dv[HD / 16];
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    dq[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a_dst, am_tr_b<HD>(s.k, dt, lane), z, 0, 0, 0);   // (query 4 g + r, d)
    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a_dsn, am_tr_b<HD>(s.q, dt, lane), z, 0, 0, 0);   // (key 4 g + r, d)
    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a_pn, am_tr_b<HD>(s.d, dt, lane), z, 0, 0, 0);    // (key 4 g + r, d)
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s.q[4 * g + r][16 * dt + i] = (__bf16)dq[dt][r];
      s.k[4 * g + r][16 * dt + i] = (__bf16)dk[dt][r];
      s.v[4 * g + r][16 * dt + i] = (__bf16)dv[dt][r];
    }
  __builtin_amdgcn_wave_barrier();
  __bf16* out = dqkv + sample * 16 * (3L * D) + head * HD;
  am_unstage<HD>(s.q, out, 3L * D, lane);
  am_unstage<HD>(s.k, out + D, 3L * D, lane);
  am_unstage<HD>(s.v, out + 2 * D, 3L * D, lane);
}

}  // namespace train
}  // namespace scldm
