// GEMM with bf16 SOURCES for the generic training path (row T1 on shapes outside the fused family: the DiT-L of configs[4]).
// C[m][n] (+)= sum_k A(m,k) B(n,k) (+ bias[n]), fp32 accumulate / output, v_mfma_f32_32x32x16_bf16.
//
// hgemm_kernel (train.hpp) reads fp32 operands and rounds them while staging: 2x the bytes, 8 conversions per 16 bytes of LDS
// and, for operands contiguous along m, a register transpose - 140-290 TFLOP/s on the DiT-L products.  Here the producers
// (LayerNorm / attention / SwiGLU / gate kernels, the per-step weight cast) leave bf16 arrays behind, tiles travel
// global -> registers -> LDS as 16-byte pieces exactly as they lie in memory, and the MFMA fragments are taken from LDS by
//   * ds_read_b128 when the operand is contiguous along k ("KC": x[token][k] of a forward product, dy[token][k] of a dgrad),
//   * ds_read_b64_tr_b16 when it is contiguous along m ("MC": both operands of a weight gradient, W[k][n] of a dgrad),
// the same scheme as the fused path's wgrad_bf16_kernel (train_fused.hip), generalised to the three operand orientations.
// 128 x 128 output tile per workgroup of four waves (64 x 64 each), 64 k per stage, two LDS buffers (<= 80 KB: two workgroups
// per CU), two register stages per operand so that the loads of stage i+2 fly over the MFMAs of stage i.
// Requirements (checked by the host): 16-byte aligned bases and leading dimensions that are multiples of 8 elements.  An
// extent that is not a multiple of 8 (DiT-L: hidden 2 732) lives in rows padded to the next multiple WITH ZEROS: the last
// 16-byte piece of a row is loaded whole, so along k it multiplies padding by padding (KC x KC) or padding by the zeros an
// out-of-range k row of an MC operand returns, and along m / n it lands in output rows / columns that are not stored.
#pragma once
#include "common.hpp"

namespace scldm {
namespace train {

struct BGemmArgs {
  const __bf16* A; int lda;   // KC: A(m,k) = A[m*lda + k];  MC: A(m,k) = A[k*lda + m]
  const __bf16* B; int ldb;
  float* C; long ldc;
  __bf16* C16;       // optional: the result is stored as bf16 here (row stride N) instead of fp32 in C (no split-K, no accumulate)
  const float* bias;
  int M, N, K;
  int kchunk;        // multiple of 64; == K rounded up without split-K
  int splits;        // k ranges; partial z goes to C + z*M*ldc (ldc == N then)
  int accumulate;    // C += (only without split-K)
  float* rowsum;     // optional, MC A only: rowsum[z*M + m] = sum_k A(m,k) of this split (first column of tiles)
  int tiles_m, tiles_n, per_xcd;
  // bgemm8 only - fused SwiGLU epilogues (EPI bits 2048 / 4096 of bgemm8_kernel): the pre-activations a, b as dense bf16 [M][ep_n]
  // arrays and two / three bf16 outputs with row stride ep_ldo, zero padded up to ep_pad columns
  const __bf16 *ep_a, *ep_b;
  __bf16 *ep_o1, *ep_o2, *ep_o3;
  int ep_n, ep_ldo, ep_pad;
  int rowsum_split;  // bgemm8 only: tile column tn sums the stages t % tiles_n == tn and writes rowsum[tn * M + m] (the host adds the tiles_n partials)
  int n_blocks;      // logical workgroups of this product (a persistent launch walks them with a grid stride; 0: one per launched workgroup)
};

constexpr int kGK = 64;              // k per stage
constexpr int kGLdMC = 128 + 32;     // bf16 per LDS row of a [k][m] image (see kWLD in train_fused.hip)
constexpr int kGLdKC = kGK + 8;      // bf16 per LDS row of a [m][k] image (144 B: the 16 rows of a quarter-wave b128 read fall on distinct 16-byte slots)
constexpr int kGOperandBytes = (kGK * kGLdMC > 128 * kGLdKC ? kGK * kGLdMC : 128 * kGLdKC) * 2;
constexpr int kBGemmLds = 4 * kGOperandBytes;   // two buffers x two operands

typedef __attribute__((ext_vector_type(4))) unsigned bg_u32x4;
constexpr unsigned kOob = 0x80000000u;   // a vector offset beyond every descriptor's range: the load returns zeros

// 128 (m) x 64 (k) of an operand contiguous along k.  pass p: row 32 p + 8 wave + lane / 8, k chunk lane % 8 (8 elements).
struct LoaderKC {
  bg_u32x4 d[4];
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int ld, int m0, int rows, int k0, int kend) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool k_ok = k0 + (lane & 7) * 8 < kend;
    const unsigned voff = (unsigned)(lane >> 3) * (unsigned)ld * 2u + (unsigned)(lane & 7) * 16u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = m0 + 32 * p + 8 * wave;
      const bool ok = k_ok && r + (lane >> 3) < rows;
      d[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? voff : kOob, ((unsigned)r * (unsigned)ld + (unsigned)k0) * 2u, 0);
    }
  }
  __device__ __forceinline__ void store(__bf16* __restrict__ S) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<bg_u32x4*>(S + (32 * p + 8 * wave + (lane >> 3)) * kGLdKC + (lane & 7) * 8) = d[p];
  }
  __device__ __forceinline__ void add_rowsum(float (&)[8]) const {}
  // rows [f0, f0+32) x k [kk, kk+16): lane l holds row l % 32, k 8 (l / 32) .. +7
  static __device__ __forceinline__ bf16x8 frag(const __bf16* __restrict__ S, int f0, int kk, int lane) {
    return *reinterpret_cast<const bf16x8*>(S + (f0 + (lane & 31)) * kGLdKC + kk + 8 * (lane >> 5));
  }
};

// 64 (k) x 128 (m) of an operand contiguous along m.  pass p: k row 16 p + 4 wave + lane / 16, m chunk lane % 16.
struct LoaderMC {
  bg_u32x4 d[4];
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int ld, int m0, int rows /* m extent */, int k0, int kend) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool m_ok = m0 + (lane & 15) * 8 < rows;
    const unsigned voff = (unsigned)(lane >> 4) * (unsigned)ld * 2u + (unsigned)(lane & 15) * 16u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = k0 + 16 * p + 4 * wave;
      const bool ok = m_ok && k + (lane >> 4) < kend;
      d[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? voff : kOob, ((unsigned)k * (unsigned)ld + (unsigned)m0) * 2u, 0);
    }
  }
  __device__ __forceinline__ void store(__bf16* __restrict__ S) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<bg_u32x4*>(S + (16 * p + 4 * wave + (lane >> 4)) * kGLdMC + (lane & 15) * 8) = d[p];
  }
  // rs[e] += the stage's values of m = 8 (lane % 16) + e (this thread's four k rows)
  __device__ __forceinline__ void add_rowsum(float (&rs)[8]) const {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rs[2 * i] += __uint_as_float(d[p][i] << 16);
        rs[2 * i + 1] += __uint_as_float(d[p][i] & 0xffff0000u);
      }
  }
  // ds_read_b64_tr_b16: lane i of a 16-lane group receives column i of the 4 x 16 block whose 8-byte pieces the group addresses
  static __device__ __forceinline__ bf16x8 frag(const __bf16* __restrict__ S, int f0, int kk, int lane) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int g = lane >> 4, i = lane & 15;
    const __bf16* p = S + (kk + 8 * (g >> 1) + (i >> 2)) * kGLdMC + f0 + 16 * (g & 1) + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * kGLdMC));
    union { s16x4 s[2]; bf16x8 f; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.f;
  }
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void bgemm_kernel(const BGemmArgs g) {
  using LA = typename std::conditional<A_KC, LoaderKC, LoaderMC>::type;
  using LB = typename std::conditional<B_KC, LoaderKC, LoaderMC>::type;
  extern __shared__ __attribute__((aligned(16))) char bgemm_smem[];
  auto As = [&](int b) { return reinterpret_cast<__bf16*>(bgemm_smem + b * kGOperandBytes); };
  auto Bs = [&](int b) { return reinterpret_cast<__bf16*>(bgemm_smem + (2 + b) * kGOperandBytes); };
  // Workgroup -> (tile, k range).  Workgroups are dealt to the 8 XCDs round-robin by linear id.  With split-K the k range is
  // the fast index (8 splits: one range per XCD, every tile's operand rows of that range stay in that XCD's L2); without, XCD x
  // takes the contiguous run of tiles [x * per_xcd, (x+1) * per_xcd): a band of tile rows, i.e. few A panels against all of B.
  int z = 0, tile_id;
  if (g.splits > 1) {
    z = blockIdx.x % g.splits;
    tile_id = blockIdx.x / g.splits;
  } else {
    tile_id = (blockIdx.x & 7) * g.per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= g.per_xcd) return;
  }
  if (tile_id >= g.tiles_m * g.tiles_n) return;
  const int tm = tile_id / g.tiles_n, tn = tile_id % g.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int k_beg = z * g.kchunk, k_end = min(g.K, k_beg + g.kchunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

  auto make_rsrc = [](const __bf16* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A), rb = make_rsrc(g.B);
  const bool want_rs = !A_KC && g.rowsum != nullptr && tn == 0;
  float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int n_it = (k_end - k_beg + kGK - 1) / kGK;
  if (n_it <= 0) return;   // (workgroup-uniform)
  LA la[2];
  LB lb[2];
  la[0].load(ra, g.lda, m0, g.M, k_beg, k_end);
  lb[0].load(rb, g.ldb, n0, g.N, k_beg, k_end);
  la[1].load(ra, g.lda, m0, g.M, k_beg + min(1, n_it - 1) * kGK, k_end);
  lb[1].load(rb, g.ldb, n0, g.N, k_beg + min(1, n_it - 1) * kGK, k_end);
  if (want_rs) la[0].add_rowsum(rs);
  la[0].store(As(0));
  lb[0].store(Bs(0));
  lds_barrier();
  auto iteration = [&](int it, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;   // register slot of stage `it` (already in LDS buffer it & 1): free again
    const int buf = it & 1;
    // unconditional (the last two iterations re-request the last stage): loads under a branch make the waitcnt pass assume the
    // path that issued none, and it then drains the NEW stage too when the old one is needed
    const int ahead = k_beg + min(it + 2, n_it - 1) * kGK;
    la[SLOT].load(ra, g.lda, m0, g.M, ahead, k_end);
    lb[SLOT].load(rb, g.ldb, n0, g.N, ahead, k_end);
#pragma unroll
    for (int kk = 0; kk < kGK; kk += 16) {
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = LA::frag(As(buf), wm * 64 + i * 32, kk, lane);
#pragma unroll
      for (int k = 0; k < 2; ++k) bf[k] = LB::frag(Bs(buf), wn * 64 + k * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[k], acc[i][k], 0, 0, 0);
    }
    if (it + 1 < n_it) {
      if (want_rs) la[SLOT ^ 1].add_rowsum(rs);
      la[SLOT ^ 1].store(As(buf ^ 1));
      lb[SLOT ^ 1].store(Bs(buf ^ 1));
    }
    lds_barrier();
  };
  for (int it = 0; it < n_it; it += 2) {
    iteration(it, std::integral_constant<int, 0>{});
    if (it + 1 < n_it) iteration(it + 1, std::integral_constant<int, 1>{});
  }
  if constexpr (!A_KC) {
    if (want_rs) {   // the 16 threads (tid / 16) that share an m chunk hold partial sums of the same eight rows
      float* red = reinterpret_cast<float*>(bgemm_smem);   // [16][128]
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(threadIdx.x >> 4) * 128 + (threadIdx.x & 15) * 8 + e] = rs[e];
      lds_barrier();
      if (threadIdx.x < 128 && m0 + (int)threadIdx.x < g.M) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q * 128 + threadIdx.x];
        g.rowsum[(long)z * g.M + m0 + threadIdx.x] = t;
      }
    }
  }
  float* __restrict__ C = g.C + (long)z * g.M * g.ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int n = n0 + wn * 64 + k * 32 + (lane & 31);
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + acc_row(r, lane >> 5);
        if (m < g.M) {
          float v = acc[i][k][r] + bv;
          if (g.C16) {
            g.C16[(long)m * g.N + n] = (__bf16)v;
          } else {
            float* p = C + (long)m * g.ldc + n;
            if (g.accumulate) v += *p;
            *p = v;
          }
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 output tile, eight waves (2 along m x 4 along n, 128 x 64 each: 8 MFMAs per 6 fragment reads instead of 4 per 4),
// one workgroup per CU.  Timing proxies of the 128 x 128 kernel on the DiT-L products (45 us per call): without MFMAs -8 %,
// without fragment reads 0...-14 %, without global loads -15 %, without the LDS write pass -31 % - no pipe is the limit, the
// 16-MFMA-per-barrier lockstep of two workgroups is.  Here a stage is 32 MFMAs per wave, and the staging is the one-register-set
// scheme: stage t+1 sits in registers while stage t is multiplied, is written to the other LDS buffer behind the MFMAs, and
// the loads of stage t+2 are issued right after - one barrier per stage.  Used when the 256-tiles (x k splits) fill the chip.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kG2LdMC = 256 + 32;                      // [k][m] image: 576 B rows (64 mod 256, like 320)
constexpr int kG2OperandBytes = 256 * kGLdKC * 2;      // = 64 * kG2LdMC * 2 = 36 864
static_assert(64 * kG2LdMC * 2 == kG2OperandBytes, "both images of a 256-wide operand stage have the same size");
constexpr int kBGemm2Lds = 4 * kG2OperandBytes;        // 147 456 B

struct LoaderKC2 {   // 256 (m) x 64 (k): pass p: row 64 p + 8 wave + lane / 8, k chunk lane % 8
  bg_u32x4 d[4];
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int ld, int m0, int rows, int k0, int kend) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool k_ok = k0 + (lane & 7) * 8 < kend;
    const unsigned voff = (unsigned)(lane >> 3) * (unsigned)ld * 2u + (unsigned)(lane & 7) * 16u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = m0 + 64 * p + 8 * wave;
      const bool ok = k_ok && r + (lane >> 3) < rows;
      d[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? voff : kOob, ((unsigned)r * (unsigned)ld + (unsigned)k0) * 2u, 0);
    }
  }
  __device__ __forceinline__ void store(__bf16* __restrict__ S) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<bg_u32x4*>(S + (64 * p + 8 * wave + (lane >> 3)) * kGLdKC + (lane & 7) * 8) = d[p];
  }
  __device__ __forceinline__ void add_rowsum(float (&)[8]) const {}
  static __device__ __forceinline__ bf16x8 frag(const __bf16* __restrict__ S, int f0, int kk, int lane) { return LoaderKC::frag(S, f0, kk, lane); }
};
struct LoaderMC2 {   // 64 (k) x 256 (m): pass p: k row 16 p + 2 wave + lane / 32, m chunk lane % 32
  bg_u32x4 d[4];
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int ld, int m0, int rows, int k0, int kend) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool m_ok = m0 + (lane & 31) * 8 < rows;
    const unsigned voff = (unsigned)(lane >> 5) * (unsigned)ld * 2u + (unsigned)(lane & 31) * 16u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = k0 + 16 * p + 2 * wave;
      const bool ok = m_ok && k + (lane >> 5) < kend;
      d[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? voff : kOob, ((unsigned)k * (unsigned)ld + (unsigned)m0) * 2u, 0);
    }
  }
  __device__ __forceinline__ void store(__bf16* __restrict__ S) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<bg_u32x4*>(S + (16 * p + 2 * wave + (lane >> 5)) * kG2LdMC + (lane & 31) * 8) = d[p];
  }
  __device__ __forceinline__ void add_rowsum(float (&rs)[8]) const {   // m = 8 (lane % 32) + e
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rs[2 * i] += __uint_as_float(d[p][i] << 16);
        rs[2 * i + 1] += __uint_as_float(d[p][i] & 0xffff0000u);
      }
  }
  static __device__ __forceinline__ bf16x8 frag(const __bf16* __restrict__ S, int f0, int kk, int lane) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int g = lane >> 4, i = lane & 15;
    const __bf16* p = S + (kk + 8 * (g >> 1) + (i >> 2)) * kG2LdMC + f0 + 16 * (g & 1) + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * kG2LdMC));
    union { s16x4 s[2]; bf16x8 f; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.f;
  }
};

// `bid`: the workgroup's index inside this product (blockIdx.x for a single launch; blockIdx.x minus the product's first block in
// a batched launch)
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void bgemm256_body(const BGemmArgs& g, const int bid) {
  using LA = typename std::conditional<A_KC, LoaderKC2, LoaderMC2>::type;
  using LB = typename std::conditional<B_KC, LoaderKC2, LoaderMC2>::type;
  extern __shared__ __attribute__((aligned(16))) char bgemm_smem[];
  auto As = [&](int b) { return reinterpret_cast<__bf16*>(bgemm_smem + b * kG2OperandBytes); };
  auto Bs = [&](int b) { return reinterpret_cast<__bf16*>(bgemm_smem + (2 + b) * kG2OperandBytes); };
  int z = 0, tile_id;   // (same workgroup numbering as bgemm_kernel; tiles_m / tiles_n count 256-tiles)
  if (g.splits > 1) {
    z = bid % g.splits;
    tile_id = bid / g.splits;
  } else {
    tile_id = (bid & 7) * g.per_xcd + (bid >> 3);
    if ((bid >> 3) >= g.per_xcd) return;
  }
  if (tile_id >= g.tiles_m * g.tiles_n) return;
  const int tm = tile_id / g.tiles_n, tn = tile_id % g.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int k_beg = z * g.kchunk, k_end = min(g.K, k_beg + g.kchunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

  auto make_rsrc = [](const __bf16* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A), rb = make_rsrc(g.B);
  const bool want_rs = !A_KC && g.rowsum != nullptr && tn == 0;
  float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int n_it = (k_end - k_beg + kGK - 1) / kGK;
  if (n_it <= 0) return;
  LA la;
  LB lb;
  la.load(ra, g.lda, m0, g.M, k_beg, k_end);
  lb.load(rb, g.ldb, n0, g.N, k_beg, k_end);
  if (want_rs) la.add_rowsum(rs);
  la.store(As(0));
  lb.store(Bs(0));
  la.load(ra, g.lda, m0, g.M, k_beg + min(1, n_it - 1) * kGK, k_end);
  lb.load(rb, g.ldb, n0, g.N, k_beg + min(1, n_it - 1) * kGK, k_end);
  lds_barrier();
  for (int it = 0; it < n_it; ++it) {
    const int buf = it & 1;
#pragma unroll
    for (int kk = 0; kk < kGK; kk += 16) {
      bf16x8 af[4], bf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = LA::frag(As(buf), wm * 128 + i * 32, kk, lane);
#pragma unroll
      for (int k = 0; k < 2; ++k) bf[k] = LB::frag(Bs(buf), wn * 64 + k * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[k], acc[i][k], 0, 0, 0);
    }
    // stage it+1 (in registers since the previous iteration) -> the other buffer, whose readers passed the last barrier; then
    // re-request: stage it+2 (unconditional, see bgemm_kernel)
    if (it + 1 < n_it) {
      if (want_rs) la.add_rowsum(rs);
      la.store(As(buf ^ 1));
      lb.store(Bs(buf ^ 1));
    }
    const int ahead = k_beg + min(it + 2, n_it - 1) * kGK;
    la.load(ra, g.lda, m0, g.M, ahead, k_end);
    lb.load(rb, g.ldb, n0, g.N, ahead, k_end);
    lds_barrier();
  }
  if constexpr (!A_KC) {
    if (want_rs) {   // the 16 threads (tid / 32) that share an m chunk hold partial sums of the same eight rows
      float* red = reinterpret_cast<float*>(bgemm_smem);   // [16][256]
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(threadIdx.x >> 5) * 256 + (threadIdx.x & 31) * 8 + e] = rs[e];
      lds_barrier();
      if (threadIdx.x < 256 && m0 + (int)threadIdx.x < g.M) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q * 256 + threadIdx.x];
        g.rowsum[(long)z * g.M + m0 + threadIdx.x] = t;
      }
    }
  }
  float* __restrict__ C = g.C + (long)z * g.M * g.ldc;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int n = n0 + wn * 64 + k * 32 + (lane & 31);
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 128 + i * 32 + acc_row(r, lane >> 5);
        if (m < g.M) {
          float v = acc[i][k][r] + bv;
          if (g.C16) {
            g.C16[(long)m * g.N + n] = (__bf16)v;
          } else {
            float* p = C + (long)m * g.ldc + n;
            if (g.accumulate) v += *p;
            *p = v;
          }
        }
      }
    }
}

// Persistent form (n_blocks > 0): the launch has one workgroup per CU (a multiple of 8, so a workgroup keeps its XCD) and each walks
// the logical workgroups bid = blockIdx.x, blockIdx.x + gridDim.x, ... - a 256 x 256 tile of a K = 1 024 product is only 16 stages, and
// a fresh workgroup per tile pays launch + first-load latency + store drain each time with nothing else resident on the CU to hide it.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512) void bgemm256_kernel(const BGemmArgs g) {
  if (g.n_blocks <= 0) {
    bgemm256_body<A_KC, B_KC>(g, (int)blockIdx.x);
    return;
  }
  for (int bid = blockIdx.x; bid < g.n_blocks; bid += gridDim.x) {
    bgemm256_body<A_KC, B_KC>(g, bid);
    lds_barrier();   // the next tile's first stage overwrites the LDS buffers (and the row-sum scratch) this one read last
  }
}

// Several independent products in ONE launch (the five weight gradients of a DiT-L layer: 48 + 16 + 3 x 44 = 196 tiles of 256 x 256
// fill the chip WITHOUT split-K - no partial tiles, no reduction kernels, one launch instead of twelve).  Workgroups
// [first[j], first[j+1]) belong to product j; inside a product the numbering (and the XCD-aware tile order) is bgemm256_body's.
constexpr int kBGemmBatchMax = 6;
struct BGemmBatch {
  BGemmArgs job[kBGemmBatchMax];
  int first[kBGemmBatchMax + 1];
  int n;
};
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512) void bgemm256_batch_kernel(const BGemmBatch b) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < kBGemmBatchMax; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first[i]) j = i;
  bgemm256_body<A_KC, B_KC>(b.job[j], (int)blockIdx.x - b.first[j]);
}

// dst[r][c] = bf16(src[r][c]), rows of the copy padded with zeros to ldd elements, one job per blockIdx.y (the per-step bf16
// copies of the layers' weight matrices).  cols % 4 == 0.
// raw != 0: plain fp32 copy of rows * cols floats to (float*)dst (the stacked adaLN bias).
// dst_t (optional): ALSO the transposed copy dst_t[c][r] = bf16(src[r][c]), rows padded with zeros to ldt elements - the operand of
// a data gradient dY W as a k-contiguous B (W^T[in][out]), so that it runs on the same (KC, KC) kernels as the forward.  The source
// is read once: 64 x 64 tiles through LDS, both copies written in 8-byte pieces along their own contiguous dimension.
struct CastJob { const float* src; __bf16* dst; int rows, cols, ldd, raw; __bf16* dst_t; int ldt, tcols; };   // tcols: padded length of a transposed row (<= its stride ldt)
__global__ __launch_bounds__(256) void cast_jobs_kernel(const CastJob* __restrict__ jobs, int n_jobs) {
  const CastJob j = jobs[blockIdx.y];
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  if (j.dst_t) {
    __shared__ __bf16 tile[64][64 + 4];   // 136-byte rows: the 8-byte pieces of a transposed read walk 4 rows
    const int tr = (j.rows + 63) / 64, tc = (j.cols + 63) / 64;
    const int t = threadIdx.x;
    for (int id = blockIdx.x; id < tr * tc; id += gridDim.x) {
      const int r0 = (id / tc) * 64, c0 = (id % tc) * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + (t >> 4) + 16 * i, c = c0 + (t & 15) * 4;
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)0.f;
        if (r < j.rows && c < j.cols) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(j.src + (long)r * j.cols + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
          *reinterpret_cast<bf16x4_t*>(j.dst + (long)r * j.ldd + c) = o;
          if (c + 4 == j.cols)
            for (int p = j.cols; p < j.ldd; ++p) j.dst[(long)r * j.ldd + p] = (__bf16)0.f;
        }
        *reinterpret_cast<bf16x4_t*>(&tile[(t >> 4) + 16 * i][(t & 15) * 4]) = o;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = (t >> 4) + 16 * i, r = (t & 15) * 4;   // dst_t row c0 + c, elements r0 + r .. + 3
        if (c0 + c < j.cols) {
          __bf16* d = j.dst_t + (long)(c0 + c) * j.ldt + r0 + r;
          if (r0 + r + 3 < j.tcols) {   // (tcols % 4 == 0: a piece is either inside the padded row or outside it)
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = tile[r + e][c];   // (rows beyond j.rows hold zeros: the padding of the transposed rows)
            *reinterpret_cast<bf16x4_t*>(d) = o;
          }
        }
      }
      __syncthreads();
    }
    return;
  }
  const int cq = j.cols / 4;
  const long total = (long)j.rows * cq;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cq;
    const int c = (int)(i - r * cq) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(j.src + r * j.cols + c);
    if (j.raw) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(j.dst) + r * j.cols + c) = v;
      continue;
    }
    bf16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x4_t*>(j.dst + r * j.ldd + c) = o;
    if (c + 4 == j.cols)
      for (int p = j.cols; p < j.ldd; ++p) j.dst[r * j.ldd + p] = (__bf16)0.f;
  }
}

}  // namespace train
}  // namespace scldm
