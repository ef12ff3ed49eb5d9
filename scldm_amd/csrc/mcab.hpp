// MCAB (multi-head cross-attention block) pooling / unpooling of the scLDM TransformerVAE on gfx950, fp32-exact.
//
// Replaces the reference's eager chains (all Python, flex_attention materialising (B,4,16,S) / (B,4,G,16) scores):
//   InputTransformerVAE.forward (log1p)                 src/scldm/layers.py:28-31,111-118
//   CrossAttention.forward                              src/scldm/layers.py:248-264
//   CrossAttentionBlock.forward, non-adaLN branch       src/scldm/layers.py:325-330  (residual from q, not x)
//   Encoder.forward / Decoder.forward                   src/scldm/nnets.py:137-144, 200-208
//   Block.forward, non-adaLN branch (the 16-token trunks) src/scldm/layers.py:222-226
//   NegativeBinomialTransformerLayer.forward (shared theta) src/scldm/stochastic_layers.py:102-116
// Shapes are the reference's only VAE family (vae_base.yaml:8-19): n_embed 32, 16 inducing points, 16 latent
// channels, trunk heads 8x4, cross heads 4x8, bias=False, affine LayerNorms, SwiGLU hidden 88 (padded to 96).
//
// Per-gene work (23 kFLOP per decoded gene, 6 kFLOP per encoded gene) runs as chains of fp32 MFMAs
// (v_mfma_f32_32x32x2_f32, bit-exact fp32 FMA chains) on tiles of 32 genes that never leave the registers:
// a 32x32 accumulator tile T[R][C] (lane = C, register = R) is directly a legal B operand (contracting R) or,
// read as T^T, a legal A operand of the next MFMA, because the k order of an MFMA is free as long as both operands
// use the same one.  Weights are pre-packed into matching A/B fragment order ("k = acc_row(step, lane>>5)").
#pragma once
#include "common.hpp"
#include "nb_sample.hpp"
#include <type_traits>

namespace scldm {

constexpr int kE = 32;         // VAE n_embed
constexpr int kNI = 16;        // inducing points / latent tokens
constexpr int kHPad = 96;      // SwiGLU hidden 88 padded to 6 tiles of 16
constexpr int kHTiles = kHPad / 16;

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return z;
}
__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// Operand policy of the per-gene contractions and of the trunks' Linears (common.hpp): OpF32 = the exact chain; OpBF16 / OpFP16 = 16-bit
// operands, fp32 accumulate.  OpFP16 carries TF32's 10 mantissa bits - the arithmetic class the reference runs MCAB in
// (torch.set_float32_matmul_precision("high"): experiments/scripts/inference.py:26, train.py:18) - at the bf16 MFMA rate.
// What fp16 gives up against TF32 is exponent range.  Weights and the per-cell K / V fragments are converted once per workgroup
// and SATURATE at +-65 504 (mcab_pack8_sat); the per-tile activation operands (LayerNorm outputs, softmax probabilities, attention
// outputs, SwiGLU products of O(1) pre-activations) are converted with one v_cvt_pk_f16_f32 per pair and no clamp - both 16-bit
// kernels are VALU-bound (profiles/r6_mcab_*: VALU ~96 % busy, matrix pipe 14-18 %), and a v_med3 per element was 13 % of the
// instruction stream; the DiT's fp16 policy makes the same choice.
// log1p for the 16-bit operand policies: log(u) c / (u - 1) with u = fl(1 + c) compensates the rounding of 1 + c (the classic
// HP-15C form), so it is accurate to a few ulp for every c >= 0 - v_log_f32 + v_rcp_f32 + 6 VALU against the ~120 instructions of
// libm's double-float log1pf, which was a quarter of the VALU-bound pooling kernel's instruction stream (profiles/r6_mcab_*).
// The exact-fp32 policy keeps log1pf (parity path; that kernel is matrix-pipe-bound).
__device__ __forceinline__ float log1p_fast(float c) {
  const float u = 1.0f + c, d = u - 1.0f;
  const float l = __builtin_amdgcn_logf(u) * 0.6931471805599453f;     // v_log_f32 is log2
  return d == 0.f ? c : l * (c * __builtin_amdgcn_rcpf(d));
}

template <class OP> struct McabOp { static constexpr bool k16 = true; using Frag = typename OP::Frag; };
template <> struct McabOp<OpF32> { static constexpr bool k16 = false; using Frag = bf16x8; };
template <class OP> __device__ __forceinline__ typename McabOp<OP>::Frag mcab_pack8(const float* v) {
  if constexpr (std::is_same<OP, OpFP16>::value) return OpFP16::pack8(v);
  else return OpBF16::pack8(v);
}
template <class OP> __device__ __forceinline__ typename McabOp<OP>::Frag mcab_pack8_sat(const float* v) {
  if constexpr (std::is_same<OP, OpFP16>::value) {
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_fmed3f(v[i], -65504.f, 65504.f);
    return OpFP16::pack8(t);
  } else {
    return OpBF16::pack8(v);
  }
}

// ------------------------------------------------------------------------------------------------
// Fragment packing (run once per weight load).  A "weight fragment" of step j holds, for lane l,
// W[row = l & 31][k = kidx(j, l >> 5)]; fragments are stored 4 steps per lane as float4: ((j/4)*64 + l)*4 + (j&3).
// ------------------------------------------------------------------------------------------------
// plain matrix W (32 rows, ld) with k = acc_row(j, hh), j < 16  (K = 32)
__global__ void pack_frag32_kernel(const float* __restrict__ W, int ld, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 16 * 64) return;
  const int l = idx & 63, j = idx >> 6;
  out[((j >> 2) * 64 + l) * 4 + (j & 3)] = W[(l & 31) * ld + acc_row(j, l >> 5)];
}
// the same with k = 16 (l >> 5) + j: the B operand then is "lane half hh holds features 16 hh .. 16 hh + 15" (the fp32 per-gene
// decoder's attention output leaves the VALU in that order: heads 2 hh and 2 hh + 1)
__global__ void pack_frag32_halves_kernel(const float* __restrict__ W, int ld, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 16 * 64) return;
  const int l = idx & 63, j = idx >> 6;
  out[((j >> 2) * 64 + l) * 4 + (j & 3)] = W[(l & 31) * ld + 16 * (l >> 5) + j];
}
// SwiGLU up-projection, tile u (16 hidden): rows 0-15 = w1[16u + r], rows 16-31 = w2[16u + r]; hidden >= H -> 0
__global__ void pack_frag_w12_kernel(const float* __restrict__ W1, const float* __restrict__ W2, int H, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kHTiles * 16 * 64) return;
  const int l = idx & 63, j = (idx >> 6) & 15, u = idx >> 10;
  const int r = l & 31, hid = 16 * u + (r & 15);
  const float* src = (r < 16) ? W1 : W2;
  out[(size_t)u * 1024 + ((j >> 2) * 64 + l) * 4 + (j & 3)] = (hid < H) ? src[hid * kE + acc_row(j, l >> 5)] : 0.f;
}
// SwiGLU down-projection Wc (32, H): tile u, step r < 8: k = hidden 16u + acc_row(r, hh) (< 16 for r < 8)
__global__ void pack_frag_wc_kernel(const float* __restrict__ Wc, int H, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kHTiles * 8 * 64) return;
  const int l = idx & 63, r = (idx >> 6) & 7, u = idx >> 9;
  const int hid = 16 * u + acc_row(r, l >> 5);
  out[(size_t)u * 512 + ((r >> 2) * 64 + l) * 4 + (r & 3)] = (hid < H) ? Wc[(l & 31) * H + hid] : 0.f;
}

// plain matrix W (rows x K, ld) zero-padded to a 32 x 32 tile: rows >= `rows` and k >= K are zero
__global__ void pack_frag32_pad_kernel(const float* __restrict__ W, int ld, int rows, int K, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 16 * 64) return;
  const int l = idx & 63, j = idx >> 6, row = l & 31, k = acc_row(j, l >> 5);
  out[((j >> 2) * 64 + l) * 4 + (j & 3)] = (row < rows && k < K) ? W[row * ld + k] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// The whole re-pack as ONE launch over a job table (round 4; scldm_vae_load_weights used to issue ~100 small launches and ~50
// device-to-device copies, and the Python face synchronised the host at every encode / decode to compare weight norms).  A job is
// one of the index maps above or a plain copy; every job owns kVaeJobBlocks workgroups.  `dirty` (device) gates the launch: the
// fingerprint of the source tensors is compared on device (vae_fingerprint_kernel / vae_fp_compare_kernel, the DiT's scheme), so a
// refresh costs five small launches and NO host round trip, and `.data` updates of the parameters are still picked up.
// ------------------------------------------------------------------------------------------------
enum : int { VJ_COPY = 0, VJ_FRAG32 = 1, VJ_FRAG32_HALVES = 2, VJ_W12 = 3, VJ_WC = 4, VJ_PAD = 5 };
struct VaePackJob {
  int kind, p0, p1, p2;       // COPY: n; FRAG32 / HALVES: ld; W12 / WC: H; PAD: ld, rows, K
  const float* src0;
  const float* src1;
  float* dst;
};
constexpr int kVaeJobBlocks = 24;   // x 256 threads >= the largest job (w12: kHTiles * 1024 elements; copies up to 6 144 floats)
__global__ __launch_bounds__(256) void vae_pack_jobs_kernel(const VaePackJob* __restrict__ jobs, int n_jobs, const int* __restrict__ dirty) {
  if (dirty && dirty[0] == 0) return;
  const VaePackJob j = jobs[blockIdx.x / kVaeJobBlocks];
  const int idx = (blockIdx.x % kVaeJobBlocks) * 256 + threadIdx.x;
  const int l = idx & 63;
  switch (j.kind) {
    case VJ_COPY:
      if (idx < j.p0) j.dst[idx] = j.src0[idx];
      break;
    case VJ_FRAG32: {
      if (idx >= 16 * 64) return;
      const int s = idx >> 6;
      j.dst[((s >> 2) * 64 + l) * 4 + (s & 3)] = j.src0[(l & 31) * j.p0 + acc_row(s, l >> 5)];
    } break;
    case VJ_FRAG32_HALVES: {
      if (idx >= 16 * 64) return;
      const int s = idx >> 6;
      j.dst[((s >> 2) * 64 + l) * 4 + (s & 3)] = j.src0[(l & 31) * j.p0 + 16 * (l >> 5) + s];
    } break;
    case VJ_W12: {
      if (idx >= kHTiles * 16 * 64) return;
      const int s = (idx >> 6) & 15, u = idx >> 10;
      const int r = l & 31, hid = 16 * u + (r & 15);
      const float* src = (r < 16) ? j.src0 : j.src1;
      j.dst[(size_t)u * 1024 + ((s >> 2) * 64 + l) * 4 + (s & 3)] = (hid < j.p0) ? src[hid * kE + acc_row(s, l >> 5)] : 0.f;
    } break;
    case VJ_WC: {
      if (idx >= kHTiles * 8 * 64) return;
      const int r = (idx >> 6) & 7, u = idx >> 9;
      const int hid = 16 * u + acc_row(r, l >> 5);
      j.dst[(size_t)u * 512 + ((r >> 2) * 64 + l) * 4 + (r & 3)] = (hid < j.p0) ? j.src0[(l & 31) * j.p0 + hid] : 0.f;
    } break;
    case VJ_PAD: {
      if (idx >= 16 * 64) return;
      const int s = idx >> 6, row = l & 31, k = acc_row(s, l >> 5);
      j.dst[((s >> 2) * 64 + l) * 4 + (s & 3)] = (row < j.p1 && k < j.p2) ? j.src0[row * j.p0 + k] : 0.f;
    } break;
  }
}
// 64-bit position-dependent wrapping sum of every source tensor (tensor blockIdx.x split over gridDim.y workgroups - the sum does
// not depend on the split; 64 since round 4: with 8 the 544 k words of the gene table were 266 dependent iterations per thread,
// 98 us per call - a sixth of a bf16 encode at 1 024 cells); see dit_aux.hpp
struct VaeFpSrc { const uint32_t* p; long long n; };
__global__ __launch_bounds__(256) void vae_fingerprint_kernel(const VaeFpSrc* __restrict__ src, unsigned long long* __restrict__ acc) {
  const VaeFpSrc s = src[blockIdx.x];
  if (s.n <= 0) return;
  // (at least 2 048 words per workgroup: the ~170 small tensors are one workgroup = ONE atomic each - same-address 64-bit atomics
  // serialise at ~20 ns, and 2 400 of them were most of this kernel's 53 us)
  long long chunk = ((s.n + gridDim.y - 1) / gridDim.y + 255) / 256 * 256;
  if (chunk < 2048) chunk = 2048;
  const long long lo = (long long)blockIdx.y * chunk, hi = lo + chunk < s.n ? lo + chunk : s.n;
  if (lo >= s.n) return;
  unsigned long long hsum = 0;
  auto mix = [&](unsigned long long v, long long pos) {
    hsum += (v + 0x9E3779B97F4A7C15ull * (unsigned long long)(pos + 1 + blockIdx.x * 7919ll)) * 0xBF58476D1CE4E5B9ull ^ (v << 29);
  };
  // (a thread's loop is a chain of dependent HBM round trips unless several loads are in flight: 16-byte loads, two per iteration;
  // chunk boundaries are multiples of 256 words, so only the tensor's base address decides the alignment)
  if ((reinterpret_cast<size_t>(s.p) & 15) == 0) {
    const uint4* p4 = reinterpret_cast<const uint4*>(s.p);
    const long long q_lo = lo / 4, q_hi = hi / 4;      // whole 16-byte groups of [lo, hi)
    long long q = q_lo + threadIdx.x;
    for (; q + 256 < q_hi; q += 512) {
      const uint4 u = p4[q], w = p4[q + 256];
      mix(u.x, 4 * q); mix(u.y, 4 * q + 1); mix(u.z, 4 * q + 2); mix(u.w, 4 * q + 3);
      mix(w.x, 4 * (q + 256)); mix(w.y, 4 * (q + 256) + 1); mix(w.z, 4 * (q + 256) + 2); mix(w.w, 4 * (q + 256) + 3);
    }
    if (q < q_hi) {
      const uint4 u = p4[q];
      mix(u.x, 4 * q); mix(u.y, 4 * q + 1); mix(u.z, 4 * q + 2); mix(u.w, 4 * q + 3);
    }
    for (long long pos = 4 * q_hi + threadIdx.x; pos < hi; pos += 256) mix(s.p[pos], pos);   // the last 0-3 words
  } else {
    for (long long pos = lo + threadIdx.x; pos < hi; pos += 256) mix(s.p[pos], pos);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hsum += __shfl_xor(hsum, o);
  __shared__ unsigned long long wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = hsum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}
// state[0] = accumulator of this pass, state[1] = fingerprint of the packed copies; dirty[0] = re-pack?, dirty[1] = force
__global__ void vae_fp_compare_kernel(unsigned long long* __restrict__ state, int* __restrict__ dirty) {
  dirty[0] = (state[0] != state[1]) || dirty[1];
  dirty[1] = 0;
  state[1] = state[0];
  state[0] = 0;
}
__global__ void vae_set_word_kernel(int* p, int v) { *p = v; }

// ------------------------------------------------------------------------------------------------
// The 16-token trunk: n_layer non-adaLN Blocks (affine LN, 8 heads x 4, SwiGLU, no biases; layers.py:222-226) on the
// (16, 32) state of a cell.  TWO cells travel as one 32-token MFMA tile through the whole network, carried by the two waves
// of a workgroup (trunk_blocks below says how they share a layer; round 1 ran one cell per 64-thread workgroup on the VALU with
// an LDS read per FMA: a flat 0.5 ms per call, 10-25 % of decode).  The state lives in registers in accumulator layout - lane = token (cell = token >> 4),
// register r <-> feature acc_row(r, lane >> 5) - which is directly the B operand of the next Linear, so every Linear is a
// chain of exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32) whose A operands are pre-packed weight fragments read straight from
// L2 (53 KB per layer, shared by every wave).  Only the 16 x 16 attention of the 8 four-dimensional heads goes through LDS:
// on the matrix pipe its block-sparse P V would cost 8x the useful work.
// Packed per-layer weights (floats): ln1_w 32 | ln1_b 32 | ln2_w 32 | ln2_b 32 | qkv 3 tiles x 1024 | proj 1024 |
//                                    w12 6 tiles x 1024 | c_proj 6 tiles x 512
// ------------------------------------------------------------------------------------------------
enum : int { T_LN1W = 0, T_LN1B = 32, T_LN2W = 64, T_LN2B = 96, T_QKV = 128, T_PROJ = T_QKV + 3 * 1024, T_W12 = T_PROJ + 1024,
             T_WC = T_W12 + kHTiles * 1024, kTrunkLayerFloats = T_WC + kHTiles * 512 };
constexpr int kTrunkLd = 100;                    // floats per token row of the per-wave attention scratch (q | k | v, 16-byte aligned rows)
constexpr int kTrunkWaves = 2;                   // the two waves that share ONE cell pair (see trunk_blocks)
constexpr int kTrunkXchg = 16 * 64;              // floats of one wave's partial MLP output in the exchange area
constexpr int kTrunkSmemFloats = 32 * kTrunkLd + 2 * kTrunkXchg;

// LDS traffic of ONE wave needs no workgroup barrier; the fence keeps the compiler from moving the reads above the writes
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc += W-tile (16 steps = K 32, fragments from global memory) x[16] (accumulator-order registers)
// OP = OpF32: sixteen (eight) exact-fp32 MFMA steps.  OP = OpFP16 / OpBF16 (round 6: the 16-token trunks of the 16-bit policies - the
// reference runs these Linears in TF32 too, and a trunk is a chain of DEPENDENT products: 13 of them per layer at 16 x 64 cycles each in
// fp32): eight consecutive fp32 steps are the 16 k-values of one 16-bit MFMA in the same (step, half-wave) order, so the fp32 fragments
// are converted four at a time (weights saturating) and the chain is two (one) MFMAs.
template <class OP>
__device__ __forceinline__ f32x16 chain_w8(const f32x4& wa, const f32x4& wb, const float* x, f32x16 acc) {   // steps [s0, s0 + 8)
  if constexpr (McabOp<OP>::k16) {
    const float w8[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
    return OP::mma(mcab_pack8_sat<OP>(w8), mcab_pack8<OP>(x), acc);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = mfma2(wa[i], x[i], acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = mfma2(wb[i], x[4 + i], acc);
    return acc;
  }
}
template <class OP = OpF32>
__device__ __forceinline__ f32x16 chain16(const float* __restrict__ frag, const float (&x)[16], f32x16 acc, int lane) {
  const f32x4* F = reinterpret_cast<const f32x4*>(frag);
  f32x4 w[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) w[g4] = F[g4 * 64 + lane];
  acc = chain_w8<OP>(w[0], w[1], x, acc);
  return chain_w8<OP>(w[2], w[3], x + 8, acc);
}
template <class OP = OpF32>
__device__ __forceinline__ f32x16 chain8(const float* __restrict__ frag, const float (&x)[8], f32x16 acc, int lane) {
  const f32x4* F = reinterpret_cast<const f32x4*>(frag);
  const f32x4 w0 = F[lane], w1 = F[64 + lane];
  return chain_w8<OP>(w0, w1, x, acc);
}

// The same with the fragments already in registers: inside the trunk the NEXT tile's fragments are requested before the
// current chain runs (an L2 round trip per 16-step chain would otherwise sit in front of every one of its 26 chains per layer)
struct Frag16 { f32x4 w[4]; };
struct Frag8 { f32x4 w[2]; };
__device__ __forceinline__ Frag16 load16(const float* __restrict__ frag, int lane) {
  const f32x4* F = reinterpret_cast<const f32x4*>(frag);
  Frag16 f;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) f.w[g4] = F[g4 * 64 + lane];
  return f;
}
__device__ __forceinline__ Frag8 load8(const float* __restrict__ frag, int lane) {
  const f32x4* F = reinterpret_cast<const f32x4*>(frag);
  Frag8 f;
  f.w[0] = F[lane];
  f.w[1] = F[64 + lane];
  return f;
}
template <class OP = OpF32>
__device__ __forceinline__ f32x16 chain16(const Frag16& f, const float (&x)[16], f32x16 acc) {
  acc = chain_w8<OP>(f.w[0], f.w[1], x, acc);
  return chain_w8<OP>(f.w[2], f.w[3], x + 8, acc);
}
template <class OP = OpF32>
__device__ __forceinline__ f32x16 chain8(const Frag8& f, const float (&x)[8], f32x16 acc) {
  return chain_w8<OP>(f.w[0], f.w[1], x, acc);
}

// LayerNorm over the first `width` features of every token (registers beyond `width` must be zero); affine if w != nullptr
__device__ __forceinline__ void tile_ln(const float (&x)[16], float (&y)[16], const float* __restrict__ w, const float* __restrict__ b,
                                        int width, float eps, int hh) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += x[r];
  const float mean = xor32_sum(s) / (float)width;
  float ss = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float d = (acc_row(r, hh) < width) ? x[r] - mean : 0.f;
    y[r] = d;
    ss += d * d;
  }
  const float rstd = __builtin_amdgcn_rsqf(xor32_sum(ss) / (float)width + eps);   // v_rsq_f32, 1 ulp
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 w4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (w) {
      w4 = *reinterpret_cast<const f32x4*>(w + q * 8 + hh * 4);
      b4 = *reinterpret_cast<const f32x4*>(b + q * 8 + hh * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) y[q * 4 + i] = y[q * 4 + i] * rstd * w4[i] + b4[i];
  }
}

// The same with the affine vectors already in registers (LnVec: this lane's 16 weights and 16 biases): inside the trunk they are
// requested one LayerNorm ahead - read at the point of use, each LayerNorm started with an exposed L2 round trip
struct LnVec { f32x4 w[4], b[4]; };
__device__ __forceinline__ LnVec load_ln(const float* __restrict__ w, const float* __restrict__ b, int hh) {
  LnVec v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v.w[q] = *reinterpret_cast<const f32x4*>(w + q * 8 + hh * 4);
    v.b[q] = *reinterpret_cast<const f32x4*>(b + q * 8 + hh * 4);
  }
  return v;
}
__device__ __forceinline__ void tile_ln(const float (&x)[16], float (&y)[16], const LnVec& v, float eps) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += x[r];
  const float mean = xor32_sum(s) / (float)kE;
  float ss = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float d = x[r] - mean;
    y[r] = d;
    ss += d * d;
  }
  const float rstd = __builtin_amdgcn_rsqf(xor32_sum(ss) / (float)kE + eps);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) y[q * 4 + i] = y[q * 4 + i] * rstd * v.w[q][i] + v.b[q][i];
}

// SwiGLU MLP on a normalised tile: x += c_proj( silu(w1 yn) * (w2 yn) ), six 16-unit hidden tiles (88 padded to 96 with zeros)
template <class OP = OpF32>
__device__ __forceinline__ void tile_swiglu(const float* __restrict__ w12, const float* __restrict__ wc, const float (&yn)[16],
                                            float (&x)[16], int lane) {
  f32x16 mo = zero16();
#pragma unroll 1
  for (int u = 0; u < kHTiles; ++u) {
    const f32x16 ht = chain16<OP>(w12 + u * 1024, yn, zero16(), lane);   // rows 0-15 = w1 units, 16-31 = the matching w2 units
    float hv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) hv[r] = silu_f(ht[r]) * ht[r + 8];
    mo = chain8<OP>(wc + u * 512, hv, mo, lane);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] += mo[r];
}

// n_layer plain Blocks on a 32-token tile x (in place).  The tile (two cells) is carried by the TWO waves of the workgroup: a
// single wave walks 13 dependent MFMA chains per layer (3 q|k|v + 1 proj + 6 x 1.5 SwiGLU) at one wave per SIMD - 0.155 ms per
// call whatever the batch, with half the SIMDs idle at 1 024 cells.  Both waves hold the state x and run the LayerNorms and
// the output projection redundantly (bit-identical); wave 0 computes the q and k tiles, wave 1 the v tile; each wave runs the
// attention of ONE of the two cells and three of the six SwiGLU hidden tiles, and the two partial down-projections are
// exchanged through LDS and added in a fixed order (partial of wave 0 + partial of wave 1 in both waves): 7.5 chains per layer.
// S = the pair's scratch [32][kTrunkLd], X = the exchange area [2][16][64]; four workgroup barriers per layer.
// The LayerNorm vectors are requested one LayerNorm ahead (LnVec); the kernels are bounded to two waves per SIMD (256 registers):
// same-box A/B at 1 024 / 2 048 / 4 096 cells: decoder trunk 110 / 139 / 259 -> 108 / 126 / 225 us, encoder tail 114 / 143 / 281
// -> 113 / 136 / 260 us (three waves per SIMD spill and lose 10 %; unbounded - 305 registers, one wave - is 22 % faster at
// 1 024 cells and 10 % slower at 4 096).
template <class OP = OpF32>
__device__ __forceinline__ void trunk_blocks(float (&x)[16], float* __restrict__ S, float* __restrict__ X, const float* __restrict__ wts,
                                             int n_layer, float eps, int lane, int hw) {
  const int c32 = lane & 31, hh = lane >> 5;
  if (n_layer <= 0) return;
  Frag16 nxt = load16(wts + T_QKV + (hw ? 2 * 1024 : 0), lane);   // fragments of the wave's next chain, always one chain ahead
  LnVec ln1 = load_ln(wts + T_LN1W, wts + T_LN1B, hh);
  for (int layer = 0; layer < n_layer; ++layer) {
    const float* w = wts + (size_t)layer * kTrunkLayerFloats;
    const LnVec ln2 = load_ln(w + T_LN2W, w + T_LN2B, hh);   // used after the attention
    float yn[16];
    tile_ln(x, yn, ln1, eps);
    if (layer + 1 < n_layer) ln1 = load_ln(w + kTrunkLayerFloats + T_LN1W, w + kTrunkLayerFloats + T_LN1B, hh);   // the next layer's
    // q | k | v (split order of layers.py:147) -> scratch rows [token][q 32 | k 32 | v 32]
    auto put_tile = [&](const f32x16& o, int t) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(S + c32 * kTrunkLd + t * 32 + q * 8 + hh * 4) = f32x4{o[q * 4], o[q * 4 + 1], o[q * 4 + 2], o[q * 4 + 3]};
    };
    if (hw == 0) {
      Frag16 cur = nxt;
      nxt = load16(w + T_QKV + 1024, lane);
      put_tile(chain16<OP>(cur, yn, zero16()), 0);
      cur = nxt;
      nxt = load16(w + T_PROJ, lane);
      put_tile(chain16<OP>(cur, yn, zero16()), 1);
    } else {
      const Frag16 cur = nxt;
      nxt = load16(w + T_PROJ, lane);
      put_tile(chain16<OP>(cur, yn, zero16()), 2);
    }
    __syncthreads();
    // attention: 8 heads x 16 queries of cell `hw` = 128 (head, query) pairs, two per lane; the output replaces q in place
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {
      const int p = lane + 64 * rep, hd = (p >> 4) & 7, qi = p & 15;
      const float* base = S + (hw * 16) * kTrunkLd + hd * 4;
      const f32x4 qv = *reinterpret_cast<const f32x4*>(base + qi * kTrunkLd);
      float sc[16], m = -3.0e38f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(base + j * kTrunkLd + 32);
        sc[j] = (qv[0] * kv[0] + qv[1] * kv[1] + qv[2] * kv[2] + qv[3] * kv[3]) * (0.5f * 1.4426950408889634f);   // log2(e) / sqrt(head_dim 4)
        m = fmaxf(m, sc[j]);
      }
      float sum = 0.f;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float pj = __builtin_amdgcn_exp2f(sc[j] - m);
        sum += pj;
        const f32x4 vv = *reinterpret_cast<const f32x4*>(base + j * kTrunkLd + 64);
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] += pj * vv[d];
      }
      const float inv = __builtin_amdgcn_rcpf(sum);
      *reinterpret_cast<f32x4*>(S + (hw * 16 + qi) * kTrunkLd + hd * 4) = f32x4{o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
    }
    __syncthreads();
    float ao[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(S + c32 * kTrunkLd + q * 8 + hh * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) ao[q * 4 + i] = t4[i];
    }
    __syncthreads();   // the scratch is rewritten by the next layer's q | k | v
    {
      const Frag16 cur = nxt;                     // proj (both waves: each needs the new state)
      nxt = load16(w + T_W12 + hw * 1024, lane);
      const f32x16 po = chain16<OP>(cur, ao, zero16());
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] += po[r];
    }
    tile_ln(x, yn, ln2, eps);
    // SwiGLU: hidden tiles hw, hw + 2, hw + 4, each consumed by the down-projection as soon as it exists
    f32x16 mo = zero16();
    const bool last = layer + 1 == n_layer;
#pragma unroll
    for (int uu = 0; uu < kHTiles / 2; ++uu) {
      const int u = 2 * uu + hw;
      const Frag16 cur = nxt;
      const Frag8 wcf = load8(w + T_WC + u * 512, lane);
      if (uu + 1 < kHTiles / 2) nxt = load16(w + T_W12 + (u + 2) * 1024, lane);
      else if (!last) nxt = load16(w + kTrunkLayerFloats + T_QKV + (hw ? 2 * 1024 : 0), lane);
      const f32x16 ht = chain16<OP>(cur, yn, zero16());   // rows 0-15 = w1 units, 16-31 = the matching w2 units
      float hv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) hv[r] = silu_f(ht[r]) * ht[r + 8];
      mo = chain8<OP>(wcf, hv, mo);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) X[hw * kTrunkXchg + r * 64 + lane] = mo[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float other = X[(hw ^ 1) * kTrunkXchg + r * 64 + lane];
      x[r] += hw ? other + mo[r] : mo[r] + other;   // partial of wave 0 + partial of wave 1, in both waves
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Decoder, per cell pair (one workgroup of two waves, trunk_blocks): LN(z) -> Linear 16->32 -> trunk -> LN1 (cross block) -> c_attn -> K | V,
// PLAIN (fp32 per-gene kernel, whose attention runs on the VALU): written per cell as [key 16][K 32 | V 32] fp32 (4 KiB);
// otherwise packed as MFMA A-operand fragments for the bf16-operand per-gene kernel (48 fragments = 12 KiB per cell):
//   K tile t (heads 2t, 2t+1), step jj < 8:  lane(row = hl*16 + key, hh): K[key][k] if head(k) == 2t + hl else 0,
//                                             k = acc_row(8t + jj, hh)
//   V tile t, step r < 16: lane(row = f, hh): V[key][f] if head(f) == 2t + hl else 0, (hl, key) = acc_row(r, hh) >> 4, & 15
// ------------------------------------------------------------------------------------------------
struct DecCellArgs {
  const float* z;        // (B, 16, n_lat)
  const float* lat_frag; // decoder_latent_input.1.weight (32, n_lat) as a K-padded fragment tile
  const float* trunk;    // packed trunk weights
  const float* ca_ln1_w; const float* ca_ln1_b;  // decoder_cross_attention.ln_1
  const float* kv_frag;  // decoder_cross_attention.attn.c_attn.weight (64, 32): K tile | V tile fragments
  float* kvfrag;         // (B, 48*64) floats (fragments) or (B, 16, 64) (PLAIN)
  int B, n_lat, n_layer;
  float eps;
};
template <class OP>
__global__ __launch_bounds__(64 * kTrunkWaves, 2) void dec_cell_kernel(const DecCellArgs a) {
  constexpr bool PLAIN = !McabOp<OP>::k16;   // fp32 policy: K | V rows for the VALU attention of dec_gene_kernel<OpF32>; 16-bit policies: MFMA fragments
  __shared__ __attribute__((aligned(16))) float SM[kTrunkSmemFloats];
  const int lane = threadIdx.x & 63, hw = threadIdx.x >> 6;   // both waves carry the workgroup's cell pair (trunk_blocks)
  const int c32 = lane & 31, hh = lane >> 5;
  const int pair = blockIdx.x;
  float* S = SM;
  float* X = SM + 32 * kTrunkLd;
  const int cell = min(pair * 2 + (c32 >> 4), a.B - 1);   // an odd batch pads its last tile with a copy of the last cell
  // LN (no affine) over the n_lat latent channels of every token, then Linear n_lat -> 32 (no bias)
  float zr[16], zn[16], x[16];
  const float* zrow = a.z + ((size_t)cell * kNI + (c32 & 15)) * a.n_lat;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = acc_row(r, hh);
    zr[r] = f < a.n_lat ? zrow[f] : 0.f;
  }
  tile_ln(zr, zn, nullptr, nullptr, a.n_lat, a.eps, hh);
  {
    const f32x16 h0 = chain16<OP>(a.lat_frag, zn, zero16(), lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = h0[r];
  }
  trunk_blocks<OP>(x, S, X, a.trunk, a.n_layer, a.eps, lane, hw);
  if (hw) return;   // the tail (no workgroup barrier below) is wave 0's
  float yn[16];
  tile_ln(x, yn, a.ca_ln1_w, a.ca_ln1_b, kE, a.eps, hh);
#pragma unroll
  for (int t = 0; t < 2; ++t) {   // scratch rows [token][K 32 | V 32]
    const f32x16 o = chain16<OP>(a.kv_frag + t * 1024, yn, zero16(), lane);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f32x4*>(S + c32 * kTrunkLd + t * 32 + q * 8 + hh * 4) = f32x4{o[q * 4], o[q * 4 + 1], o[q * 4 + 2], o[q * 4 + 3]};
  }
  wave_lds_sync();
  if constexpr (PLAIN) {
    for (int c = 0; c < 2; ++c) {
      if (pair * 2 + c >= a.B) break;
      float* out = a.kvfrag + (size_t)(pair * 2 + c) * (kNI * 64);
      const int key = lane >> 2, col = (lane & 3) * 16;
      const float* src = S + (c * 16 + key) * kTrunkLd + col;
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(out + key * 64 + col + q * 4) = *reinterpret_cast<const f32x4*>(src + q * 4);
    }
    return;
  }
  const int row = lane & 31;
  for (int c = 0; c < 2; ++c) {
    if (pair * 2 + c >= a.B) break;
    float* out = a.kvfrag + (size_t)(pair * 2 + c) * 48 * 64;
    const float* Sc = S + c * 16 * kTrunkLd;
    for (int t = 0; t < 2; ++t)
      for (int jj = 0; jj < 8; ++jj) {
        const int hl = row >> 4, key = row & 15, k = acc_row(8 * t + jj, hh);
        const int j = t * 8 + jj;
        out[((j >> 2) * 64 + lane) * 4 + (j & 3)] = ((k >> 3) == 2 * t + hl) ? Sc[key * kTrunkLd + k] : 0.f;
      }
    for (int t = 0; t < 2; ++t)
      for (int r = 0; r < 16; ++r) {
        const int rr = acc_row(r, hh), hl = rr >> 4, key = rr & 15;
        const int j = 16 + t * 16 + r;
        out[((j >> 2) * 64 + lane) * 4 + (j & 3)] = ((row >> 3) == 2 * t + hl) ? Sc[key * kTrunkLd + 32 + row] : 0.f;
      }
  }
}

// Per-gene query table (run once per weight load): Qtab[g] = c_attn_q(LN_1q(emb[g])) / sqrt(8)
__global__ void dec_qtab_kernel(const float* __restrict__ emb, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                const float* __restrict__ wq, float* __restrict__ qtab, int rows, float eps,
                                const int* __restrict__ gate = nullptr) {
  if (gate && *gate == 0) return;   // scldm_vae_refresh_weights: the packed copies are up to date
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows) return;
  float v[kE], s = 0.f;
#pragma unroll
  for (int k = 0; k < kE; ++k) { v[k] = emb[(size_t)g * kE + k]; s += v[k]; }
  const float mean = s * (1.0f / kE);
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < kE; ++k) { v[k] -= mean; ss += v[k] * v[k]; }
  const float rstd = 1.0f / sqrtf(ss * (1.0f / kE) + eps);
#pragma unroll
  for (int k = 0; k < kE; ++k) v[k] = v[k] * rstd * lnw[k] + lnb[k];
  for (int n = 0; n < kE; ++n) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < kE; ++k) acc += v[k] * wq[n * kE + k];
    qtab[(size_t)g * kE + n] = acc * 0.35355339059327373f;
  }
}

// ------------------------------------------------------------------------------------------------
// Decoder, per gene: MCAB unpooling + NB head logits.  Workgroup = 4 waves, each wave walks tiles of 32 genes of
// ONE cell; everything between the gene-embedding gather and the logit is a register-resident MFMA chain.
// ------------------------------------------------------------------------------------------------
struct DecGeneArgs {
  const int64_t* genes;   // (B, G)
  const float* emb;       // input_layer.gene_embedding.weight (n_genes+1, 32)
  const float* qtab;      // (n_genes+1, 32) pre-projected, pre-scaled queries
  const float* theta_emb; // decoder_head.theta.weight (n_genes+1, 1)
  const float* kvfrag;    // bf16 path: (B, 48*64) K/V fragments; fp32 path: (B, 16, 64) plain K | V per inducing point
  const float* wfrag;     // packed c_proj (16 steps) | w12 (6*16) | wc (6*8) fragments = 160*64 floats
  const float* wfrag_cproj_halves;   // fp32 path: the 16 c_proj fragments with k in lane-half order (pack_frag32_halves_kernel)
  const float* ln2_w; const float* ln2_b;  // decoder_cross_attention.ln_2
  const float* head_w; const float* head_b;  // decoder_head.params (1,32), (1): the mu logit
  const float* head_w2;   // unshared theta (stochastic_layers.py:94-96,109-111: params is Linear(32, 2), theta = exp of its second
                          // output): row 1 of the weight, its bias is head_b[1]; nullptr = shared theta (the gene-indexed table)
  float* logits;          // (B, G)  (aliases mu)
  float* theta;           // (B, G), or nullptr when the caller only wants drawn counts (scldm_vae_decode_sample)
  float* part;            // (B, n_chunks, 2): running (max, sum exp) of logit / temperature per chunk
  int G, n_chunks, tiles_per_wave;
  float eps, inv_temp;
};

// BF = false: exact-fp32 chain (v_mfma_f32_32x32x2_f32, the parity path).  BF = true: the same chain with bf16 operands
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate; softmax / LayerNorm / SiLU / logits stay fp32): eight consecutive fp32 steps
// contract exactly the 16 k-values of one bf16 MFMA in the same (step, half-wave) order, so the packed fragments are the
// fp32 ones converted eight steps at a time and every activation operand is its accumulator registers 0-7 / 8-15.
// Workgroup = kDecWaves waves sharing one LDS copy of the weight fragments.  Eight waves with the register budget of four
// waves per SIMD (two workgroups per CU; 52 KB of LDS each): the chain of one tile is strictly dependent (MFMA -> softmax
// / LayerNorm / SiLU -> MFMA), so the matrix pipe is kept busy by OTHER waves - three per SIMD left it idle a third of the
// time (r2 PMC: SQ_VALU_MFMA_BUSY_CYCLES 67 % of the kernel, 68 % of wave cycles waiting on the pipe).
#ifndef SCLDM_DEC_WAVES
#define SCLDM_DEC_WAVES 8
#endif
constexpr int kDecWaves = SCLDM_DEC_WAVES;
constexpr int kDecThreads = 64 * kDecWaves;
template <class OP>
#ifndef SCLDM_DEC_UNROLL
#define SCLDM_DEC_UNROLL 1   // the six SwiGLU tiles of a gene tile unrolled: the next tile's up-projection MFMAs issue under this tile's SiLU (+2 % fp32 decode)
#endif
#ifndef SCLDM_DEC_MINW
#define SCLDM_DEC_MINW 4
#endif
__global__ __launch_bounds__(kDecThreads, McabOp<OP>::k16 ? (kDecWaves >= 8 ? kDecWaves / 2 : 1) : SCLDM_DEC_MINW) void dec_gene_kernel(const DecGeneArgs a) {   // two workgroups per CU
  constexpr bool BF = McabOp<OP>::k16;
  using H8 = typename McabOp<OP>::Frag;
  constexpr int kWF4 = BF ? 1 : 40 * 64, kKV4 = BF ? 1 : kNI * 16, kWF8 = BF ? 20 * 64 : 1, kKV8 = BF ? 6 * 64 : 1;
  __shared__ f32x4 WF[kWF4];    // fp32: 160 weight fragments, 4 steps per float4 (the 16 c_proj ones in lane-half k order)
  __shared__ f32x4 KVP[kKV4];   // fp32: this cell's K | V, plain [key][64 floats]
  __shared__ H8 WFh[kWF8];  // 16-bit operands: the 160 fragments, 8 steps per 16-byte fragment
  __shared__ H8 KVh[kKV8];  // 16-bit operands: this cell's 48 K/V fragments
  __shared__ float VEC[4 * kE];   // ln2_w | ln2_b | head_w | head_w row 1 (unshared theta)
  __shared__ float RED[kDecWaves][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c32 = lane & 31, hh = lane >> 5;
  const int cell = blockIdx.y, chunk = blockIdx.x;
  if constexpr (BF) {
    // (the conversion also folds constants the VALU-bound chain would otherwise multiply by per element: the K fragments carry
    // log2(e), so the softmax is exp2(s - m); the SwiGLU up-projection rows carry -log2(e) (w1 units, tile rows 0-15) and
    // -1 / log2(e) (w2 units, rows 16-31), so silu(a) b = a' b' / (1 + 2^a') - the DiT kernel's form, common.hpp kW1Scale)
    auto cvt = [&](const float* src, H8* dst, int nfrag8, auto scale_of) {
      for (int i = tid; i < nfrag8 * 64; i += kDecThreads) {
        const int f = i >> 6, l = i & 63;
        const f32x4 lo = reinterpret_cast<const f32x4*>(src)[(2 * f) * 64 + l], hi = reinterpret_cast<const f32x4*>(src)[(2 * f + 1) * 64 + l];
        const float sc = scale_of(f, l);
        float t[8] = {lo[0] * sc, lo[1] * sc, lo[2] * sc, lo[3] * sc, hi[0] * sc, hi[1] * sc, hi[2] * sc, hi[3] * sc};
        dst[i] = mcab_pack8_sat<OP>(t);
      }
    };
    // weight fragments of 8 steps: 0-1 c_proj | 2-13 w12 (six tiles x two) | 14-19 wc
    cvt(a.wfrag, WFh, 20, [](int f, int l) { return (f >= 2 && f < 14) ? ((l & 31) < 16 ? -1.4426950408889634f : -0.6931471805599453f) : 1.0f; });
    // K / V fragments of 8 steps: 0-1 K | 2-5 V
    cvt(a.kvfrag + (size_t)cell * 48 * 64, KVh, 6, [](int f, int) { return f < 2 ? 1.4426950408889634f : 1.0f; });
  } else {
    for (int i = tid; i < 40 * 64; i += kDecThreads)
      WF[i] = i < 4 * 64 ? reinterpret_cast<const f32x4*>(a.wfrag_cproj_halves)[i] : reinterpret_cast<const f32x4*>(a.wfrag)[i];
    for (int i = tid; i < kNI * 16; i += kDecThreads) KVP[i] = reinterpret_cast<const f32x4*>(a.kvfrag + (size_t)cell * (kNI * 64))[i];
  }
  if (tid < kE) { VEC[tid] = a.ln2_w[tid]; VEC[kE + tid] = a.ln2_b[tid]; VEC[2 * kE + tid] = a.head_w[tid]; VEC[3 * kE + tid] = a.head_w2 ? a.head_w2[tid] : 0.f; }
  __syncthreads();
  // acc += F[steps step0 .. step0+7] (A operand, from LDS) x x[0..7] (B operand, accumulator-order registers)
  auto mm8 = [&](const f32x4* F4, const H8* F8, int step0, const float* x, f32x16 acc) {
    if constexpr (BF) {
      return OP::mma(F8[(step0 >> 3) * 64 + lane], mcab_pack8<OP>(x), acc);
    } else {
#pragma unroll
      for (int g4 = 0; g4 < 2; ++g4) {
        const f32x4 wf = F4[((step0 >> 2) + g4) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = mfma2(wf[i], x[g4 * 4 + i], acc);
      }
      return acc;
    }
  };
  const float hb = a.head_b[0];
  float run_m = -3.0e38f, run_s = 0.f;
  const int tile0 = (chunk * kDecWaves + wave) * a.tiles_per_wave;
  for (int ti = 0; ti < a.tiles_per_wave; ++ti) {
    const int gi = (tile0 + ti) * 32 + c32;
    if ((tile0 + ti) * 32 >= a.G) break;  // wave-uniform
    const bool valid = gi < a.G;
    const long long g = valid ? a.genes[(size_t)cell * a.G + gi] : 0;
    float y[16];
    f32x16 yt = zero16();
    if constexpr (BF) {
      // q^T and the raw embedding in accumulator order: register r <-> feature acc_row(r, hh)
      float q[16];
  #pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.qtab + (size_t)g * kE + qd * 8 + hh * 4);
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.emb + (size_t)g * kE + qd * 8 + hh * 4);
  #pragma unroll
        for (int i = 0; i < 4; ++i) { q[qd * 4 + i] = t4[i]; y[qd * 4 + i] = e4[i]; }
      }
      // S^T[(head, key)][gene] = Kblk q^T  (block sparse: tile t only sees features 16t..16t+15)
      f32x16 st[2];
  #pragma unroll
      for (int t = 0; t < 2; ++t) {
        st[t] = mm8(nullptr, KVh, 8 * t, q + 8 * t, zero16());
      }
      // softmax over the 16 keys of each head: 8 keys in-lane (registers 8hl..8hl+7) + 8 in the other half-wave
      float inv_h[4];
  #pragma unroll
      for (int t = 0; t < 2; ++t)
  #pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
          float m = st[t][8 * hl];
  #pragma unroll
          for (int i = 1; i < 8; ++i) m = fmaxf(m, st[t][8 * hl + i]);
          m = xor32_max(m);
          float sum = 0.f;
  #pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float p = __builtin_amdgcn_exp2f(st[t][8 * hl + i] - m);   // scores are in log2 units (K carries log2 e)
            st[t][8 * hl + i] = p;
            sum += p;
          }
          // P stays un-normalised (in (0, 1]: the same relative operand precision); 1 / sum scales this head's 8 output features
          // after P V - 16 multiplies per lane instead of 32
          inv_h[2 * t + hl] = __builtin_amdgcn_rcpf(xor32_sum(sum));   // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division
        }
      // O^T[f][gene] = Vblk^T P^T
      f32x16 ot = zero16();
  #pragma unroll
      for (int t = 0; t < 2; ++t) {
        float pr[16];
  #pragma unroll
        for (int r = 0; r < 16; ++r) pr[r] = st[t][r];
        ot = mm8(nullptr, KVh, 16 + 16 * t, pr, ot);
        ot = mm8(nullptr, KVh, 16 + 16 * t + 8, pr + 8, ot);
      }
      // y = q_raw + c_proj(O)   (residual from the query, layers.py:327)
      {
        float ov[16];
  #pragma unroll
        for (int r = 0; r < 16; ++r) ov[r] = ot[r] * inv_h[r >> 2];   // register r <-> feature acc_row(r, hh) = 8 (r >> 2) + ...: head r >> 2
        yt = mm8(WF, WFh, 0, ov, yt);
        yt = mm8(WF, WFh, 8, ov + 8, yt);
      }
    } else {
      // the raw embedding in accumulator order (register r <-> feature acc_row(r, hh): it meets the c_proj output there); the
      // pre-projected query as this lane half's two heads (features 16 hh .. 16 hh + 15)
      float q[16];
  #pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.qtab + (size_t)g * kE + hh * 16 + qd * 4);
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.emb + (size_t)g * kE + qd * 8 + hh * 4);
  #pragma unroll
        for (int i = 0; i < 4; ++i) { q[qd * 4 + i] = t4[i]; y[qd * 4 + i] = e4[i]; }
      }
      // Cross attention of one gene over the cell's 16 inducing points, heads 2 hh and 2 hh + 1 in this lane (the other lane of
      // the gene has the other two): 2 x 16 dot products of 8, an in-lane softmax, 2 x 8 outputs - 512 FMAs whose K / V operands
      // are two-address LDS broadcasts.  On 32x32 MFMA tiles the same work cost 48 fp32 MFMAs (3 072 cycles of the matrix pipe,
      // two thirds of them on structural zeros) plus four half-wave exchanges.
      float o[16];
      {
        const f32x4* KP = KVP + hh * 4;
        float sc[2][kNI];
  #pragma unroll
        for (int key = 0; key < kNI; ++key) {
          const f32x4 k0 = KP[key * 16], k1 = KP[key * 16 + 1], k2 = KP[key * 16 + 2], k3 = KP[key * 16 + 3];
          float s0 = q[0] * k0[0], s1 = q[8] * k2[0];
  #pragma unroll
          for (int i = 1; i < 4; ++i) { s0 = fmaf(q[i], k0[i], s0); s1 = fmaf(q[8 + i], k2[i], s1); }
  #pragma unroll
          for (int i = 0; i < 4; ++i) { s0 = fmaf(q[4 + i], k1[i], s0); s1 = fmaf(q[12 + i], k3[i], s1); }
          sc[0][key] = s0;
          sc[1][key] = s1;
        }
  #pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
          float m = sc[hl][0];
  #pragma unroll
          for (int key = 1; key < kNI; ++key) m = fmaxf(m, sc[hl][key]);
          float sum = 0.f;
          const float nm = -m * 1.4426950408889634f;
#pragma unroll
          for (int key = 0; key < kNI; ++key) {
            sc[hl][key] = __builtin_amdgcn_exp2f(fmaf(sc[hl][key], 1.4426950408889634f, nm));   // exp(s - m): one fma + v_exp_f32
            sum += sc[hl][key];
          }
          const float inv = __builtin_amdgcn_rcpf(sum);   // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division
  #pragma unroll
          for (int key = 0; key < kNI; ++key) sc[hl][key] *= inv;
        }
  #pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
        const f32x4* VP = KVP + 8 + hh * 4;
  #pragma unroll
        for (int key = 0; key < kNI; ++key) {
          const f32x4 v0 = VP[key * 16], v1 = VP[key * 16 + 1], v2 = VP[key * 16 + 2], v3 = VP[key * 16 + 3];
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            o[i] = fmaf(sc[0][key], v0[i], o[i]);
            o[4 + i] = fmaf(sc[0][key], v1[i], o[4 + i]);
            o[8 + i] = fmaf(sc[1][key], v2[i], o[8 + i]);
            o[12 + i] = fmaf(sc[1][key], v3[i], o[12 + i]);
          }
        }
      }
      // y = q_raw + c_proj(O)   (residual from the query, layers.py:327); the c_proj fragments are packed for this k order
      yt = mm8(WF, WFh, 0, o, yt);
      yt = mm8(WF, WFh, 8, o + 8, yt);
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[r] += yt[r]; s += y[r]; }
    const float mean = xor32_sum(s) * (1.0f / kE);
    float ss = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float d = y[r] - mean; ss += d * d; }
    const float rstd = __builtin_amdgcn_rsqf(xor32_sum(ss) * (1.0f / kE) + a.eps);
    float yn[16];
    if constexpr (BF) {
      const float nmr = -mean * rstd;     // (y - mean) rstd as one fma, then the affine fma
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = acc_row(r, hh);
        yn[r] = fmaf(fmaf(y[r], rstd, nmr), VEC[f], VEC[kE + f]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = acc_row(r, hh);
        yn[r] = (y[r] - mean) * rstd * VEC[f] + VEC[kE + f];
      }
    }
    // SwiGLU: six tiles of 16 hidden units, each consumed by the down-projection as soon as it exists
    f32x16 mo = zero16();
#if SCLDM_DEC_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int u = 0; u < kHTiles; ++u) {
      f32x16 ht = mm8(WF, WFh, 16 + 16 * u, yn, zero16());
      ht = mm8(WF, WFh, 16 + 16 * u + 8, yn + 8, ht);
      float hv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if constexpr (BF) hv[r] = OP::swiglu(ht[r], ht[r + 8]);   // pre-scaled rows: a' b' / (1 + 2^a')
        else hv[r] = silu_f(ht[r]) * ht[r + 8];                   // x * v_rcp(1 + v_exp(-x)): 4 VALU ops, not ~15
      }
      mo = mm8(WF, WFh, 16 + 16 * kHTiles + 8 * u, hv, mo);
    }
    // NB head: logit = w . (y + mlp) + b
    float lg = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) lg += (y[r] + mo[r]) * VEC[2 * kE + acc_row(r, hh)];
    lg = (xor32_sum(lg) + hb) * a.inv_temp;
    float th = 0.f;
    if (a.head_w2) {   // (wave-uniform) unshared theta: the head's second output
#pragma unroll
      for (int r = 0; r < 16; ++r) th += (y[r] + mo[r]) * VEC[3 * kE + acc_row(r, hh)];
      th = expf(xor32_sum(th) + a.head_b[1]);
    }
    if (valid && hh == 0) {
      a.logits[(size_t)cell * a.G + gi] = lg;
      if (a.theta) a.theta[(size_t)cell * a.G + gi] = a.head_w2 ? th : expf(a.theta_emb[g]);
    }
    if (valid) {  // online (max, sum exp) per lane; both half-waves carry the same value, count it once at the end
      const float nm = fmaxf(run_m, lg);
      run_s = run_s * __expf(run_m - nm) + __expf(lg - nm);
      run_m = nm;
    }
  }
  // reduce (max, sum) over the 32 gene lanes of the wave, then over the 4 waves
  float m = run_m, sacc = run_s;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o), os = __shfl_xor(sacc, o);
    const float nm = fmaxf(m, om);
    sacc = sacc * __expf(m - nm) + os * __expf(om - nm);
    m = nm;
  }
  if (lane == 0) { RED[wave][0] = m; RED[wave][1] = sacc; }
  __syncthreads();
  if (tid == 0) {
    float M = RED[0][0], S = RED[0][1];
    for (int w = 1; w < kDecWaves; ++w) {
      const float nm = fmaxf(M, RED[w][0]);
      S = S * __expf(M - nm) + RED[w][1] * __expf(RED[w][0] - nm);
      M = nm;
    }
    a.part[((size_t)cell * a.n_chunks + chunk) * 2 + 0] = M;
    a.part[((size_t)cell * a.n_chunks + chunk) * 2 + 1] = S;
  }
}

// Merge of a cell's per-chunk (max, sum exp) partials, by every wave on its own (lane = chunk, then a butterfly): no LDS, no barrier.
// Round 5 had thread 0 of every 1 024-element workgroup walk the chunks serially (17-28 dependent expf) in front of a barrier:
// the HBM-bound normalisation pass ran at 2.6 TB/s (profiles/r6a_mcab_decode_*: 0.42 ms for 8 192 x 17 002).
__device__ __forceinline__ void dec_merge_partials(const float* __restrict__ part, int cell, int n_chunks, float& M, float& S) {
  const int lane = threadIdx.x & 63;
  float m = -3.0e38f, s = 0.f;
  for (int c = lane; c < n_chunks; c += 64) {
    const float pm = part[((size_t)cell * n_chunks + c) * 2], ps = part[((size_t)cell * n_chunks + c) * 2 + 1];
    const float nm = fmaxf(m, pm);
    s = s * expf(m - nm) + ps * expf(pm - nm);
    m = nm;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o), os = __shfl_xor(s, o);
    const float nm = fmaxf(m, om);
    s = s * expf(m - nm) + os * expf(om - nm);      // (commutative up to the order of one add: both partners form the same two products)
    m = nm;
  }
  M = m;
  S = s;
}

// mu = softmax_G(logit) * library_size  (stochastic_layers.py:115), in place.  16-byte accesses on the 16-byte-aligned body of the
// row (rows start at cell * G floats: any alignment), the 0-3 leading and trailing elements by workgroup 0.
__global__ __launch_bounds__(256) void dec_finalize_kernel(float* __restrict__ mu, const float* __restrict__ part,
                                                           const float* __restrict__ library, int G, int n_chunks) {
  const int cell = blockIdx.y;
  float M, S;
  dec_merge_partials(part, cell, n_chunks, M, S);
  const float scale = library[cell] / S;
  float* row = mu + (size_t)cell * G;
  const int head = min(G, (int)((4 - ((reinterpret_cast<size_t>(row) >> 2) & 3)) & 3));
  const int nvec = (G - head) >> 2;
  f32x4* v = reinterpret_cast<f32x4*>(row + head);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
    f32x4 x = v[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = expf(x[e] - M) * scale;
    v[i] = x;
  }
  if (blockIdx.x == 0) {
    const int t = threadIdx.x, tail0 = head + 4 * nvec;
    if (t < head) row[t] = expf(row[t] - M) * scale;
    if (tail0 + t < G) row[tail0 + t] = expf(row[tail0 + t] - M) * scale;
  }
}

// The same pass with the negative-binomial draw fused in (nb_sample.hpp): out = count ~ NB(mu, theta = exp(theta_emb[gene])),
// mu and theta stay in registers.  Element index for the RNG = cell * G + gene position.
__global__ __launch_bounds__(256) void dec_finalize_sample_kernel(float* __restrict__ out, const float* __restrict__ part,
                                                                  const float* __restrict__ library, const int64_t* __restrict__ genes,
                                                                  const float* __restrict__ theta_emb, const float* __restrict__ theta_rows,
                                                                  int G, int n_chunks, unsigned long long seed) {
  // theta_rows (B, G): per-element dispersions written by dec_gene_kernel (unshared theta); nullptr = exp(theta_emb[gene])
  const int cell = blockIdx.y;
  float M, S;
  dec_merge_partials(part, cell, n_chunks, M, S);
  const float scale = library[cell] / S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < G; i += gridDim.x * 256) {
    const size_t e = (size_t)cell * G + i;
    const float mu = expf(out[e] - M) * scale;
    const float theta = theta_rows ? theta_rows[e] : expf(theta_emb[genes[e]]);
    out[e] = nb_draw(seed, e, mu, theta);
  }
}

// Stand-alone draw from given (mu, theta) tensors (NegativeBinomial.sample() on the decode outputs; also the test hook)
__global__ __launch_bounds__(256) void nb_sample_kernel(const float* __restrict__ mu, const float* __restrict__ theta, float* __restrict__ out,
                                                        size_t n, unsigned long long seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = nb_draw(seed, i, mu[i], theta[i]);
}

// ------------------------------------------------------------------------------------------------
// Encoder pooling, one workgroup (4 waves) per cell: 16 inducing-point queries x 4 heads attend over the S input
// genes with an online softmax.  Per 32-gene tile:
//   x = LN_1(emb[gene] * log1p(count))            (lane = gene; 16 features per half-wave)
//   K^T = Wk x^T   (A = Wk fragments, B = x)      -> tile (lane = gene, reg = feature) = A operand of the scores
//   V   = x Wv^T   (A = x, B = Wv fragments)      -> tile (lane = d, reg = gene)       = A operand of P V
//   S_t[gene][(hl, q)] = K Qblk^T_t               -> lane = (head, query): running max / sum are per-lane scalars
//   O^T_t[d][(hl, q)] += V^T P_t
// ------------------------------------------------------------------------------------------------
struct EncPoolArgs {
  const float* counts;    // (B, S)
  const int64_t* genes;   // (B, S)
  const float* emb;       // (n_genes+1, 32)
  const float* ln1_w; const float* ln1_b;   // encoder.ca_layer.ln_1
  const float* kfrag;     // Wk (rows 0-31 of c_attn) fragments, 16 steps
  const float* vfrag;     // Wv (rows 32-63 of c_attn) fragments, 16 steps
  const float* qfrag;     // Qblk^T fragments: 2 tiles x 8 steps (built from the inducing points at weight load)
  float* pooled;          // (B, 16, 32): softmax(QK^T/sqrt(8)) V, heads concatenated
  float* lse2;            // optional (B, 4, 16): log2-domain log-sum-exp of the scaled scores (m + log2 l), for the training backward
  int S;
  float eps;
};

// BF: bf16 operands for the four contractions of a gene tile (see dec_gene_kernel); LayerNorm, the online softmax and the
// output accumulators stay fp32.
// Waves per SIMD (same-box A/B, 1 024 cells x 6 147 genes): the bf16-operand kernel is VALU / latency bound and gains 11 % from
// three waves (168 VGPRs, 24 B of scratch) over two; the fp32 kernel is half matrix-pipe bound and is 7 % faster at two waves
// without spills (218 VGPRs) once the gather is software-pipelined.
#ifndef SCLDM_ENC_MINW
#define SCLDM_ENC_MINW 3
#endif
// NW = waves per cell (round 4).  A cell is one workgroup; with four waves of 168 registers a CU holds three cells, so 1 024 cells are
// 1.33 rounds of the chip's 768 slots - the second round a third full.  Six waves per cell (two cells per CU, 512 slots) make it two
// full rounds of cells that each finish in 4/6 of the time; the gene tiles are dealt round-robin over the waves and merged in wave
// order at the end, so NW changes the summation order of the online softmax merge (not its value beyond fp32 rounding).
template <class OP, int NW = 4>
__global__ __launch_bounds__(64 * NW, McabOp<OP>::k16 ? SCLDM_ENC_MINW : 2) void enc_pool_kernel(const EncPoolArgs a) {
  constexpr bool BF = McabOp<OP>::k16;
  using H8 = typename McabOp<OP>::Frag;
  __shared__ f32x4 KF[4 * 64], VF[4 * 64], QF[4 * 64];
  __shared__ float VEC[2 * kE];
  __shared__ float MRG[NW][2][64][18];  // per wave, per column tile, per lane: m, l, O[16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c32 = lane & 31, hh = lane >> 5;
  const int cell = blockIdx.x;
  for (int i = tid; i < 4 * 64; i += 64 * NW) {
    KF[i] = reinterpret_cast<const f32x4*>(a.kfrag)[i];
    VF[i] = reinterpret_cast<const f32x4*>(a.vfrag)[i];
    QF[i] = reinterpret_cast<const f32x4*>(a.qfrag)[i];
  }
  if (tid < kE) { VEC[tid] = a.ln1_w[tid]; VEC[kE + tid] = a.ln1_b[tid]; }
  __syncthreads();
  float m[2] = {-3.0e38f, -3.0e38f}, l[2] = {0.f, 0.f};
  f32x16 O[2] = {zero16(), zero16()};
  const int n_tiles = (a.S + 31) / 32;
  // bf16 operands: the weight / query fragments are the same for every tile - packed once (eight consecutive fp32 steps = the
  // 16 k-values of one bf16 MFMA in the same order: fragment of steps [s0, s0+8))
  H8 kf16[2], vf16[2], qf16[2];
  if constexpr (BF) {
    auto frag8 = [&](const f32x4* F, int s0) {
      const f32x4 lo = F[(s0 >> 2) * 64 + lane], hi = F[((s0 >> 2) + 1) * 64 + lane];
      float t8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return mcab_pack8_sat<OP>(t8);
    };
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) { kf16[h8] = frag8(KF, 8 * h8); vf16[h8] = frag8(VF, 8 * h8); qf16[h8] = frag8(QF, 8 * h8); }
  }
  // Software pipeline of the gather (fp32 kernel, two waves per SIMD; the bf16 kernel's three waves hide it and need the
  // registers): a tile's inputs are two dependent round trips (gene id, then its table row).  The table rows of the NEXT tile
  // and the gene ids / counts of the tile after it are requested before this tile's arithmetic; indices are clamped to the
  // cell's last gene so that every request is unconditional (padding lanes are neutralised by lc = 0 below).
  constexpr bool PF = !BF;
  auto idx_of = [&](int tile) { return (size_t)cell * a.S + max(min(min(tile, n_tiles - 1) * 32 + c32, a.S - 1), 0); };
  long long g_nxt = 0;
  float c_cur = 0.f, c_nxt = 0.f;
  f32x4 e_cur[4];
  if constexpr (PF) {
    g_nxt = a.genes[idx_of(wave + NW)];
    c_cur = a.counts[idx_of(wave)];
    c_nxt = a.counts[idx_of(wave + NW)];
    const long long g0 = a.genes[idx_of(wave)];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) e_cur[qd] = *reinterpret_cast<const f32x4*>(a.emb + (size_t)g0 * kE + qd * 8 + hh * 4);
  }
  for (int tile = wave; tile < n_tiles; tile += NW) {
    const bool valid = tile * 32 + c32 < a.S;
    float x[16], s = 0.f;
    if constexpr (PF) {
      f32x4 e_nxt[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) e_nxt[qd] = *reinterpret_cast<const f32x4*>(a.emb + (size_t)g_nxt * kE + qd * 8 + hh * 4);
      const long long g_n2 = a.genes[idx_of(tile + 2 * NW)];
      const float c_n2 = a.counts[idx_of(tile + 2 * NW)];
      const float lc = valid ? log1pf(c_cur) : 0.f;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[qd * 4 + i] = e_cur[qd][i] * lc; s += x[qd * 4 + i]; }
        e_cur[qd] = e_nxt[qd];
      }
      g_nxt = g_n2;
      c_cur = c_nxt;
      c_nxt = c_n2;
    } else {
      const long long g = a.genes[idx_of(tile)];
      const float lc = valid ? log1p_fast(a.counts[idx_of(tile)]) : 0.f;    // (this branch is the 16-bit policies')
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.emb + (size_t)g * kE + qd * 8 + hh * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[qd * 4 + i] = e4[i] * lc; s += x[qd * 4 + i]; }
      }
    }
    const float mean = xor32_sum(s) * (1.0f / kE);
    float ss = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] -= mean; ss += x[r] * x[r]; }
    const float rstd = __builtin_amdgcn_rsqf(xor32_sum(ss) * (1.0f / kE) + a.eps);
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int f = acc_row(r, hh); x[r] = x[r] * rstd * VEC[f] + VEC[kE + f]; }
    f32x16 kt = zero16(), vt = zero16();
    if constexpr (BF) {
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        const H8 xf = mcab_pack8<OP>(x + 8 * h8);
        kt = OP::mma(kf16[h8], xf, kt);   // K^T[feature][gene]
        vt = OP::mma(xf, vf16[h8], vt);   // V[gene][d]
      }
    } else {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 kf = KF[g4 * 64 + lane], vf = VF[g4 * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          kt = mfma2(kf[i], x[g4 * 4 + i], kt);   // K^T[feature][gene]
          vt = mfma2(x[g4 * 4 + i], vf[i], vt);   // V[gene][d]
        }
      }
    }
    float ktv[16], vtv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ktv[r] = kt[r]; vtv[r] = vt[r]; }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 sc = zero16();
      if constexpr (BF) {
        sc = OP::mma(mcab_pack8<OP>(ktv + 8 * t), qf16[t], sc);  // S[gene][(hl, q)]
      } else {
#pragma unroll
        for (int g4 = 0; g4 < 2; ++g4) {
          const f32x4 qf = QF[(t * 2 + g4) * 64 + lane];
#pragma unroll
          for (int i = 0; i < 4; ++i) sc = mfma2(kt[8 * t + g4 * 4 + i], qf[i], sc);  // S[gene][(hl, q)]
        }
      }
      float tm = -3.0e38f;
      if (tile == n_tiles - 1) {   // (wave-uniform) tile padding
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (tile * 32 + acc_row(r, hh) >= a.S) sc[r] = -3.0e38f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) tm = fmaxf(tm, sc[r]);
      tm = xor32_max(tm);
      // scores are in log2 units (Qblk carries log2(e) / sqrt(8), enc_qfrag_kernel): softmax through v_exp_f32 without a multiply
      const float nm = fmaxf(m[t], tm), alpha = __builtin_amdgcn_exp2f(m[t] - nm);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] - nm); ps += sc[r]; }
      l[t] = l[t] * alpha + xor32_sum(ps);
      m[t] = nm;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
      if constexpr (BF) {
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[r] = sc[r];
        O[t] = OP::mma(mcab_pack8<OP>(vtv), mcab_pack8<OP>(pv), O[t]);
        O[t] = OP::mma(mcab_pack8<OP>(vtv + 8), mcab_pack8<OP>(pv + 8), O[t]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t] = mfma2(vt[r], sc[r], O[t]);  // O^T[d][(hl, q)] += V^T P
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    MRG[wave][t][lane][0] = m[t];
    MRG[wave][t][lane][1] = l[t];
#pragma unroll
    for (int r = 0; r < 16; ++r) MRG[wave][t][lane][2 + r] = O[t][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float M = MRG[0][t][lane][0];
      for (int w = 1; w < NW; ++w) M = fmaxf(M, MRG[w][t][lane][0]);
      float L = 0.f, o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      for (int w = 0; w < NW; ++w) {
        const float sc = __builtin_amdgcn_exp2f(MRG[w][t][lane][0] - M);
        L += MRG[w][t][lane][1] * sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] += MRG[w][t][lane][2 + r] * sc;
      }
      // lane = column (hl, q) of tile t; rows d = acc_row(r, hh); only d in head 2t + hl carry this head's output
      const int hl = c32 >> 4, qi = c32 & 15, head = 2 * t + hl;
      const float inv = 1.0f / L;
      if (a.lse2 && hh == 0) a.lse2[((size_t)cell * 4 + head) * kNI + qi] = M + __log2f(L);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = acc_row(r, hh);
        if ((d >> 3) == head) a.pooled[((size_t)cell * kNI + qi) * kE + d] = o[r] * inv;
      }
    }
  }
}

// Encoder tail, per cell pair (one workgroup of two waves, trunk_blocks): c_proj + inducing-point residual -> LN2 -> SwiGLU -> (+pos_embed) -> trunk ->
// Linear 32 -> n_lat -> LN (no affine)  (layers.py:326-330, nnets.py:139-144); same register-resident tile as the trunk
struct EncCellArgs {
  const float* pooled;     // (B, 16, 32)
  const float* ind;        // encoder.ca_layer.inducing_points (16, 32)
  const float* proj_frag;  // encoder.ca_layer.attn.c_proj.weight (32, 32) fragments
  const float* ca_ln2_w; const float* ca_ln2_b;
  const float* w12_frag; const float* wc_frag;  // encoder.ca_layer.mlp fragments
  const float* pos;        // encoder.pos_embed (16, 32) or nullptr
  const float* trunk;
  const float* lat_frag;   // encoder_latent_input.0.weight (n_lat, 32) as a row-padded fragment tile
  float* z;                // (B, 16, n_lat)
  int B, n_lat, n_layer;
  float eps;
};
template <class OP>
__global__ __launch_bounds__(64 * kTrunkWaves, 2) void enc_cell_kernel(const EncCellArgs a) {
  __shared__ __attribute__((aligned(16))) float SM[kTrunkSmemFloats];
  const int lane = threadIdx.x & 63, hw = threadIdx.x >> 6;   // both waves carry the workgroup's cell pair (trunk_blocks)
  const int c32 = lane & 31, hh = lane >> 5;
  const int pair = blockIdx.x;
  float* S = SM;
  float* X = SM + 32 * kTrunkLd;
  const int cell_raw = pair * 2 + (c32 >> 4), cell = min(cell_raw, a.B - 1), tok = c32 & 15;
  float att[16], x[16], yn[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 p4 = *reinterpret_cast<const f32x4*>(a.pooled + ((size_t)cell * kNI + tok) * kE + q * 8 + hh * 4);
    const f32x4 i4 = *reinterpret_cast<const f32x4*>(a.ind + tok * kE + q * 8 + hh * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { att[q * 4 + i] = p4[i]; x[q * 4 + i] = i4[i]; }
  }
  {
    const f32x16 po = chain16<OP>(a.proj_frag, att, zero16(), lane);   // h = inducing + c_proj(att): the residual is the QUERY (layers.py:327)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] += po[r];
  }
  tile_ln(x, yn, a.ca_ln2_w, a.ca_ln2_b, kE, a.eps, hh);
  tile_swiglu<OP>(a.w12_frag, a.wc_frag, yn, x, lane);
  if (a.pos) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 p4 = *reinterpret_cast<const f32x4*>(a.pos + tok * kE + q * 8 + hh * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) x[q * 4 + i] += p4[i];
    }
  }
  trunk_blocks<OP>(x, S, X, a.trunk, a.n_layer, a.eps, lane, hw);
  if (hw) return;
  // latent head: Linear 32 -> n_lat (no bias; fragment rows >= n_lat are zero), LN without affine over the n_lat channels
  const f32x16 lt = chain16<OP>(a.lat_frag, x, zero16(), lane);
  float lv[16], ln[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) lv[r] = lt[r];
  tile_ln(lv, ln, nullptr, nullptr, a.n_lat, a.eps, hh);
  if (cell_raw < a.B) {
    float* zrow = a.z + ((size_t)cell * kNI + tok) * a.n_lat;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = acc_row(r, hh);
      if (f < a.n_lat) zrow[f] = ln[r];
    }
  }
}

// Qblk^T fragments for the encoder scores (run once per weight load): tile t, step jj < 8:
//   lane(col = hl*16 + q, hh): c_attn_q(LN_1q(inducing[q]))[k] log2(e) / sqrt(8) if head(k) == 2t + hl else 0, k = acc_row(8t + jj, hh)
__global__ __launch_bounds__(64) void enc_qfrag_kernel(const float* __restrict__ ind, const float* __restrict__ lnw,
                                                       const float* __restrict__ lnb, const float* __restrict__ wq,
                                                       float* __restrict__ out, float eps, const int* __restrict__ gate = nullptr) {
  __shared__ float qp[kNI][kE];
  if (gate && *gate == 0) return;   // (uniform: before any barrier)
  const int lane = threadIdx.x;
  if (lane < kNI) {
    float v[kE], s = 0.f;
    for (int k = 0; k < kE; ++k) { v[k] = ind[lane * kE + k]; s += v[k]; }
    const float mean = s * (1.0f / kE);
    float ss = 0.f;
    for (int k = 0; k < kE; ++k) { v[k] -= mean; ss += v[k] * v[k]; }
    const float rstd = 1.0f / sqrtf(ss * (1.0f / kE) + eps);
    for (int k = 0; k < kE; ++k) v[k] = v[k] * rstd * lnw[k] + lnb[k];
    for (int n = 0; n < kE; ++n) {
      float acc = 0.f;
      for (int k = 0; k < kE; ++k) acc += v[k] * wq[n * kE + k];
      qp[lane][n] = acc * (0.35355339059327373f * 1.4426950408889634f);   // log2(e) / sqrt(8): enc_pool_kernel's softmax runs on exp2
    }
  }
  __syncthreads();
  const int col = lane & 31, hh = lane >> 5, hl = col >> 4, qi = col & 15;
  for (int t = 0; t < 2; ++t)
    for (int jj = 0; jj < 8; ++jj) {
      const int k = acc_row(8 * t + jj, hh), j = t * 8 + jj;
      out[((j >> 2) * 64 + lane) * 4 + (j & 3)] = ((k >> 3) == 2 * t + hl) ? qp[qi][k] : 0.f;
    }
}

}  // namespace scldm
