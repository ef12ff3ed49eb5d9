// Shared device helpers for the scLDM gfx950 kernels (MI355X / CDNA4 only).
//
// MFMA conventions used everywhere in this directory (cdna_hip_programming.md section 3):
//   v_mfma_f32_32x32x16_bf16 / 8 x v_mfma_f32_32x32x2_f32 compute D[32x32] += A[32xK16] * B[K16x32]
//   A operand: lane l holds 8 k-values of row (l & 31), k-group (l >> 5)
//   B operand: lane l holds 8 k-values of col (l & 31), k-group (l >> 5)
//   C/D      : lane l holds col (l & 31); register r holds row (r&3) + 8*(r>>2) + 4*(l>>5)
// A and B layouts are symmetric, so swapping the operands yields the transposed product,
// and ANY k permutation is legal as long as A and B use the same one.  We exploit both.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scldm {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int kWave = 64;

__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// x * sigmoid(x); v_exp_f32 + v_rcp_f32 (1 ulp each) - well inside the 1e-4 parity budget
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// Reductions across the two 32-lane halves of a wave with v_permlane32_swap (VALU, no LDS round trip):
// swap(v, v) yields {v[l & 31]} and {v[(l & 31) + 32]} in every lane; sum / max are symmetric in the two.
__device__ __forceinline__ float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Lanes l and l + 32 of an accumulator tile hold adjacent 4-feature groups of the SAME token (features 8q + 4 (l >> 5) ..+3).
// Given one packed dword of quad q (a) and of quad q + 1 (b), the half-wave exchange below leaves lane l < 32 with both halves
// of quad q's 8 features and lane l >= 32 with both halves of quad q + 1's: one 16-byte LDS store per lane instead of two
// 8-byte ones (whose row stride, a multiple of 16 bytes, makes them 2-way bank conflicted: r2 PMC, 47 % of LDS-active cycles).
__device__ __forceinline__ void halfwave_pair(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];   // l < 32: own a (features 8q..8q+3)          l >= 32: lane l-32's b (features 8(q+1)..+3)
  b = r[1];   // l < 32: lane l+32's a (features 8q+4..+7)   l >= 32: own b (features 8(q+1)+4..+7)
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries an all-address-space release fence: with
// global stores (or LDS-bound DMA) in flight it drains vmcnt(0) - and with it the weight-stream prefetch ring.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Operand-precision policies.  E = element type stored in LDS / packed weights,
// Frag = one lane's 8 k-values.  `mma` accumulates a 32x32 tile over 16 k-values.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

struct OpBF16 {
  using E = __bf16;
  using ModE = _Float16;   // LDS copy of the adaLN vectors: fp16 (11-bit mantissa, far below the bf16 GEMM noise) halves its 24 KB
  static __device__ __forceinline__ f32x4 load_mod4(const ModE* p) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  static __device__ __forceinline__ void store_mod4(ModE* p, const f32x4 v) {
    *reinterpret_cast<f16x4*>(p) = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  }
  using Frag = bf16x8;
  using Quad = bf16x4;  // 4 consecutive elements (one accumulator register group)
  static constexpr bool kIsBF16 = true;
  static constexpr bool kTwoPassLN = false;  // sum / sum-of-squares in one sweep (fp32 accumulation)
  static constexpr bool kPin = true;         // pin the per-k-step issue order of a GEMM pass (sched_group_barrier)
  static constexpr bool kMfmaIn = true;      // input projection on the matrix pipe
  static constexpr bool kTwoWG = true;       // two workgroups per CU (LDS and registers allow it for 64-token tiles)
  static constexpr int kRing = 4;            // k-steps of weight-ring run-ahead
  static constexpr int kFragLoads = 1;       // 16-byte loads per fragment
  static constexpr int kMmaOps = 1;          // MFMA instructions per mma()
  static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // SwiGLU with the w1 rows packed times -log2(e) and the w2 rows times -1/log2(e) (kW1Scale / kW2Scale, applied by the weight
  // packer): the up-projection delivers a' = -log2(e) a and b' = -b / log2(e), so silu(a) b = a' b' / (1 + 2^a') - one multiply less
  static constexpr float kW1Scale = -1.4426950408889634f, kW2Scale = -0.6931471805599453f;
  static __device__ __forceinline__ float swiglu(float a, float b) { return (a * b) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a)); }
  // 4 consecutive features starting at feature f (multiple of 4) of an activation row
  static __device__ __forceinline__ void store_quad(E* row, int f, const Quad& q) { *reinterpret_cast<Quad*>(row + f) = q; }
  // quads q (q0) and q + 1 (q1) of a feature tile, f8 = first feature of quad q's 8-feature group: one 16-byte store per lane
  static __device__ __forceinline__ void store_quad_pair(E* row, int f8, int hh, const Quad& q0, const Quad& q1) {
    union { Quad q; unsigned u[2]; } a, b;
    a.q = q0; b.q = q1;
    halfwave_pair(a.u[0], b.u[0]);
    halfwave_pair(a.u[1], b.u[1]);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    *reinterpret_cast<u32x4*>(row + f8 + 8 * hh) = u32x4{a.u[0], a.u[1], b.u[0], b.u[1]};
  }
  static __device__ __forceinline__ Frag pack8(const float* v) {
    Frag f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (__bf16)v[i];
    return f;
  }
  static __device__ __forceinline__ Quad pack4(float a, float b, float c, float d) {
    Quad q;
    q[0] = (__bf16)a; q[1] = (__bf16)b; q[2] = (__bf16)c; q[3] = (__bf16)d;
    return q;
  }
};

// fp16 operands (v_mfma_f32_32x32x16_f16, fp32 accumulate): 10 explicit mantissa bits - exactly TF32's, the arithmetic the
// reference computes in under torch.set_float32_matmul_precision("high") (experiments/scripts/inference.py:26, train_ldm.py:18) -
// at the bf16 MFMA rate.  Same fragment layout, LDS image, weight stream and schedule as OpBF16; what it gives up against TF32
// is exponent range (|v| <= 65 504, gradual underflow below 6.1e-5): weights are range-checked when packed (scldm_dit_fp16_stats),
// activations of a LayerNorm-ed network are O(1-100).
struct OpFP16 {
  using E = _Float16;
  using ModE = _Float16;   // LDS copy of the adaLN vectors in the operand precision (they multiply values that are rounded to it next)
  static __device__ __forceinline__ f32x4 load_mod4(const ModE* p) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  static __device__ __forceinline__ void store_mod4(ModE* p, const f32x4 v) {
    *reinterpret_cast<f16x4*>(p) = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  }
  using Frag = f16x8;
  using Quad = f16x4;  // 4 consecutive elements (one accumulator register group)
  static constexpr bool kIsBF16 = true;   // (= "16-bit throughput policy": selects the weight-ring depth knob)
  static constexpr bool kTwoPassLN = false;  // sum / sum-of-squares in one sweep (fp32 accumulation)
  static constexpr bool kPin = true;         // pin the per-k-step issue order of a GEMM pass (sched_group_barrier)
  static constexpr bool kMfmaIn = true;      // input projection on the matrix pipe
  static constexpr bool kTwoWG = true;       // two workgroups per CU (LDS and registers allow it for 64-token tiles)
  static constexpr int kRing = 4;            // k-steps of weight-ring run-ahead
  static constexpr int kFragLoads = 1;       // 16-byte loads per fragment
  static constexpr int kMmaOps = 1;          // MFMA instructions per mma()
  static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  // SwiGLU with the w1 rows packed times -log2(e) and the w2 rows times -1/log2(e) (kW1Scale / kW2Scale, applied by the weight
  // packer): the up-projection delivers a' = -log2(e) a and b' = -b / log2(e), so silu(a) b = a' b' / (1 + 2^a') - one multiply less
  static constexpr float kW1Scale = -1.4426950408889634f, kW2Scale = -0.6931471805599453f;
  static __device__ __forceinline__ float swiglu(float a, float b) { return (a * b) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a)); }
  // 4 consecutive features starting at feature f (multiple of 4) of an activation row
  static __device__ __forceinline__ void store_quad(E* row, int f, const Quad& q) { *reinterpret_cast<Quad*>(row + f) = q; }
  // quads q (q0) and q + 1 (q1) of a feature tile, f8 = first feature of quad q's 8-feature group: one 16-byte store per lane
  static __device__ __forceinline__ void store_quad_pair(E* row, int f8, int hh, const Quad& q0, const Quad& q1) {
    union { Quad q; unsigned u[2]; } a, b;
    a.q = q0; b.q = q1;
    halfwave_pair(a.u[0], b.u[0]);
    halfwave_pair(a.u[1], b.u[1]);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    *reinterpret_cast<u32x4*>(row + f8 + 8 * hh) = u32x4{a.u[0], a.u[1], b.u[0], b.u[1]};
  }
  static __device__ __forceinline__ Frag pack8(const float* v) {
    Frag f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (_Float16)v[i];
    return f;
  }
  static __device__ __forceinline__ Quad pack4(float a, float b, float c, float d) {
    Quad q;
    q[0] = (_Float16)a; q[1] = (_Float16)b; q[2] = (_Float16)c; q[3] = (_Float16)d;
    return q;
  }
};

struct OpF32 {
  using E = float;
  using ModE = float;
  static __device__ __forceinline__ f32x4 load_mod4(const ModE* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store_mod4(ModE* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
  using Frag = f32x8;
  using Quad = f32x4;
  static constexpr bool kIsBF16 = false;
  static constexpr bool kTwoPassLN = true;   // parity path: mean first, then centred second moment
  static constexpr bool kPin = false;
  static constexpr bool kMfmaIn = false;     // exact fp32 on the VALU (K = din is tiny)
  static constexpr bool kTwoWG = false;
  static constexpr int kRing = 2;
  static constexpr int kFragLoads = 2;
  static constexpr int kMmaOps = 8;
  static constexpr float kW1Scale = 1.0f, kW2Scale = 1.0f;
  static __device__ __forceinline__ float swiglu(float a, float b) { return silu_f(a) * b; }
  static __device__ __forceinline__ void store_quad(E* row, int f, const Quad& q) { *reinterpret_cast<Quad*>(row + f) = q; }
  static __device__ __forceinline__ void store_quad_pair(E* row, int f8, int hh, const Quad& q0, const Quad& q1) {
    store_quad(row, f8 + hh * 4, q0);        // 16-byte stores already
    store_quad(row, f8 + 8 + hh * 4, q1);
  }
  // exact fp32: 8 x v_mfma_f32_32x32x2_f32 (each contracts k-groups 0 and 1 of one element slot)
  static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ Frag pack8(const float* v) {
    Frag f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = v[i];
    return f;
  }
  static __device__ __forceinline__ Quad pack4(float a, float b, float c, float d) {
    Quad q = {a, b, c, d};
    return q;
  }
};

// Split-bf16 ("bf16x3"): every GEMM operand value v is carried as hi = bf16(v), lo = bf16(v - hi) and a product sum is
// three bf16 MFMAs (hi*hi + hi*lo + lo*hi, the lo*lo term ~2^-18 relative is dropped), fp32 accumulation.  This is the
// arithmetic class of the reference's torch.set_float32_matmul_precision("high") (experiments/scripts/inference.py:26;
// TF32 / bf16x3 on its hardware) - gfx950 has no TF32, and exact fp32 MFMA runs at 1/16 of the bf16 rate.
// LDS / weight-stream layout: a k-group of 8 consecutive features is 32 bytes = 16 B of hi followed by 16 B of lo, so a
// lane's fragment is two 16-byte reads (same bytes and addresses as the fp32 policy's f32x8 fragment), and a register
// quad of 4 features is stored as two 8-byte pieces.  E is the 4-byte addressing unit of that layout.
struct bf16x8x2 { bf16x8 hi, lo; };
struct bf16x4x2 { bf16x4 hi, lo; };
struct OpBF16x3 {
  struct E { uint32_t u; };
  using ModE = float;
  static __device__ __forceinline__ f32x4 load_mod4(const ModE* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store_mod4(ModE* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
  using Frag = bf16x8x2;
  using Quad = bf16x4x2;
  static constexpr bool kIsBF16 = false;
  static constexpr bool kTwoPassLN = true;
  static constexpr bool kPin = true;
  static constexpr bool kMfmaIn = true;
  static constexpr bool kTwoWG = false;      // 155 KB of LDS per 64-token tile: one workgroup per CU
  static constexpr int kRing = 2;
  static constexpr int kFragLoads = 2;
  static constexpr int kMmaOps = 3;
  static constexpr float kW1Scale = 1.0f, kW2Scale = 1.0f;
  static __device__ __forceinline__ float swiglu(float a, float b) { return silu_f(a) * b; }
  static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);   // small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void split(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
  }
  static __device__ __forceinline__ Frag pack8(const float* v) {
    Frag f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __bf16 h, l;
      split(v[i], h, l);
      f.hi[i] = h;
      f.lo[i] = l;
    }
    return f;
  }
  static __device__ __forceinline__ Quad pack4(float a, float b, float c, float d) {
    const float v[4] = {a, b, c, d};
    Quad q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __bf16 h, l;
      split(v[i], h, l);
      q.hi[i] = h;
      q.lo[i] = l;
    }
    return q;
  }
  static __device__ __forceinline__ void store_quad(E* row, int f, const Quad& q) {
    char* b = reinterpret_cast<char*>(row) + (f >> 3) * 32 + (f & 7) * 2;
    *reinterpret_cast<bf16x4*>(b) = q.hi;
    *reinterpret_cast<bf16x4*>(b + 16) = q.lo;
  }
  // the k-group of 8 features is [hi x 8][lo x 8]: after the half-wave exchange a lane holds a whole group -> two 16-byte stores
  static __device__ __forceinline__ void store_quad_pair(E* row, int f8, int hh, const Quad& q0, const Quad& q1) {
    union { bf16x4 q; unsigned u[2]; } ah, bh, al, bl;
    ah.q = q0.hi; bh.q = q1.hi; al.q = q0.lo; bl.q = q1.lo;
    halfwave_pair(ah.u[0], bh.u[0]);
    halfwave_pair(ah.u[1], bh.u[1]);
    halfwave_pair(al.u[0], bl.u[0]);
    halfwave_pair(al.u[1], bl.u[1]);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    char* b = reinterpret_cast<char*>(row) + ((f8 >> 3) + hh) * 32;
    *reinterpret_cast<u32x4*>(b) = u32x4{ah.u[0], ah.u[1], bh.u[0], bh.u[1]};
    *reinterpret_cast<u32x4*>(b + 16) = u32x4{al.u[0], al.u[1], bl.u[0], bl.u[1]};
  }
};

}  // namespace scldm
