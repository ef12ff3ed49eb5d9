// 128 x 128 bf16-source GEMM on LDS-DMA for two k-contiguous operands - the small-tile companion of bgemm8_kernel, used where
// 256 x 256 tiles would leave most of the chip idle (activation-sized products at small batches, narrow models).
//
// bgemm_kernel (bgemm.hpp) stages through registers (global -> VGPR -> ds_write_b128) and stores its results element by element:
// 470-560 TFLOP/s.  Here, as in bgemm8.hpp: `buffer_load_dwordx4 ... lds` into unpadded 128-byte rows whose 16-byte chunk c of row
// r sits in slot c ^ ((r >> 1) & 7) (same swizzle, same conflict-free ds_read_b128 fragments), transposed accumulator blocks, and
// the result tile goes back through LDS so that a store instruction writes whole rows.
// Four waves of 64 x 64 (16 MFMAs per 64-k stage), two stages of 32 KB = 64 KB of LDS: TWO workgroups per CU.  That is what
// replaces bgemm8's wave-group stagger - the other workgroup multiplies while this one waits for its DMA or stores its tile - so
// the schedule inside a workgroup is the plain one: {fragments of stage t, MFMAs, vmcnt(0): stage t+1 has landed, barrier,
// request stage t+2 into the buffer just read}.  Hazards: RAW - a stage is read behind the barrier that follows every wave's
// vmcnt(0); WAR - a buffer is restaged behind the barrier that follows the MFMAs which consumed its fragments.
// Split-K (partial tiles into g.C + z M ldc), bias, accumulate and bf16 results as in bgemm_kernel; results are bit-identical to it
// (same MFMA, same k order, a b = b a).
#pragma once
#include "bgemm8.hpp"

namespace scldm {
namespace train {

constexpr int kG4Stage = 2 * 128 * 128;    // A rows then B rows: 32 KB
constexpr int kBGemm4Lds = 2 * kG4Stage;   // 65 536

// EPI 0: results through LDS (fp32: [128][128] floats = the whole 64 KB; bf16: 32 KB), row-wise 16-byte stores.
// EPI 1: element-wise stores straight from the (untransposed) accumulators - outputs whose rows are not 16-byte (bf16: 8-byte) aligned.
template <int EPI>
__global__ __launch_bounds__(256, 2) void bgemm4_kernel(const BGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char bgemm_smem[];
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
  const int n_it = (k_end - k_beg + kGK - 1) / kGK;
  if (n_it <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  constexpr bool TRANS = EPI == 0;

  auto make_rsrc = [](const __bf16* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    bg_u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    r[2] = 0x7fffffffu;
    r[3] = 0x00020000u;
    return r;
  };
  const bg_u32x4 ra = make_rsrc(g.A), rb = make_rsrc(g.B);
  const unsigned smem0 = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(bgemm_smem));

  // staging: instruction q of an operand covers rows 32 q + tid / 8 (256 threads x 16 bytes = 32 rows of 128 bytes)
  const int c_src = (tid & 7) ^ ((tid >> 4) & 7);
  const unsigned voff_a = (unsigned)(tid >> 3) * (unsigned)g.lda * 2u + (unsigned)c_src * 16u;
  const unsigned voff_b = (unsigned)(tid >> 3) * (unsigned)g.ldb * 2u + (unsigned)c_src * 16u;
  unsigned ok_a = 0, ok_b = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ok_a |= (unsigned)(m0 + q * 32 + (tid >> 3) < g.M) << q;
    ok_b |= (unsigned)(n0 + q * 32 + (tid >> 3) < g.N) << q;
  }
  auto issue = [&](int tt) {
    const int k0 = k_beg + tt * kGK;
    const bool k_ok = k0 + c_src * 8 < k_end;
    const unsigned base = smem0 + (unsigned)(tt & 1) * kG4Stage + (unsigned)wave * 1024u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned va = (k_ok && ((ok_a >> q) & 1u)) ? voff_a : kOob, vb = (k_ok && ((ok_b >> q) & 1u)) ? voff_b : kOob;
      const unsigned sa = ((unsigned)(m0 + q * 32) * (unsigned)g.lda + (unsigned)k0) * 2u;
      const unsigned sb = ((unsigned)(n0 + q * 32) * (unsigned)g.ldb + (unsigned)k0) * 2u;
      g8_lds_dma16(ra, va, __builtin_amdgcn_readfirstlane(sa), __builtin_amdgcn_readfirstlane(base + q * 4096u));
      g8_lds_dma16(rb, vb, __builtin_amdgcn_readfirstlane(sb), __builtin_amdgcn_readfirstlane(base + 16384u + q * 4096u));
    }
  };
  const char* rd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    rd[ks] = bgemm_smem + (lane & 31) * 128 + (((2 * ks + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
  const int a_off = wm * 64 * 128, b_off = 16384 + wn * 64 * 128;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

  issue(0);
  if (n_it > 1) issue(1);
  if (n_it > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 0 has landed (8 requests of stage 1 may fly)
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  for (int t = 0; t < n_it; ++t) {
    const int st = (t & 1) * kG4Stage;
    bf16x8 fa[2][4], fb[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i][ks] = *reinterpret_cast<const bf16x8*>(rd[ks] + st + a_off + i * 4096);
        fb[i][ks] = *reinterpret_cast<const bf16x8*>(rd[ks] + st + b_off + i * 4096);
      }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if constexpr (TRANS) acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[k][ks], fa[i][ks], acc[i][k], 0, 0, 0);
          else acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb[k][ks], acc[i][k], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage t + 1 has landed (this wave's requests; the barrier covers the others')
    __builtin_amdgcn_s_barrier();                       // ... and every wave's fragments of stage t are in registers
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < n_it) issue(t + 2);
  }

  float* __restrict__ C = g.C + (long)z * g.M * g.ldc;
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  if constexpr (TRANS) {
    // acc[i][k][4 q + e] = C(m0 + 64 wm + 32 i + lane % 32, n0 + 64 wn + 32 k + 8 q + 4 (lane / 32) + e) -> XOR-swizzled row image
    // -> rows.  (The loop's last barrier is behind every fragment read and every DMA: the staging space is free.)
    if (g.C16) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = wn * 64 + k * 32 + 8 * q + 4 * (lane >> 5), n = n0 + nl;
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(acc[i][k][4 * q + e] + ((g.bias && n + e < g.N) ? g.bias[n + e] : 0.f));
            *reinterpret_cast<bf16x4_t*>(bgemm_smem + r * 256 + (((nl >> 3) ^ (r & 15)) << 4) + (nl & 4) * 2) = o;
          }
      }
      lds_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = it * 16 + wave * 4 + (lane >> 4), j = lane & 15;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(bgemm_smem + r * 256 + ((j ^ (r & 15)) << 4));
        if (m0 + r < g.M && n0 + j * 8 < g.N) {
          __bf16* dst = g.C16 + (long)(m0 + r) * g.N + n0 + j * 8;
          if (n0 + j * 8 + 8 <= g.N) {
            *reinterpret_cast<bf16x8*>(dst) = v;
          } else {
            bf16x4_t lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) lo[e] = v[e];
            *reinterpret_cast<bf16x4_t*>(dst) = lo;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = wn * 64 + k * 32 + 8 * q + 4 * (lane >> 5), n = n0 + nl;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[i][k][4 * q + e] + ((g.bias && n + e < g.N) ? g.bias[n + e] : 0.f);
            *reinterpret_cast<f32x4*>(bgemm_smem + r * 512 + (((nl >> 2) ^ (r & 31)) << 4)) = o;
          }
      }
      lds_barrier();
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int r = it * 8 + wave * 2 + (lane >> 5), j = lane & 31;   // 4 columns n0 + 4 j .. + 3 of row r
        f32x4 v = *reinterpret_cast<const f32x4*>(bgemm_smem + r * 512 + ((j ^ (r & 31)) << 4));
        if (m0 + r < g.M && n0 + j * 4 < g.N) {   // (ldc % 4 == 0 and N % 4 == 0 on this form: a piece is inside or outside)
          f32x4* p = reinterpret_cast<f32x4*>(C + (long)(m0 + r) * g.ldc + n0 + j * 4);
          if (g.accumulate) v += *p;
          *p = v;
        }
      }
    }
    return;
  }
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

}  // namespace train
}  // namespace scldm
