// 256 x 256 bf16-source GEMM built around LDS-DMA and a phase-split schedule instead of bgemm256_kernel's register staging:
// both operands contiguous along k (forward products X W^T, data gradients against the transposed bf16 weight copies) or both
// contiguous along m (weight gradients: m-contiguous unit images, transposing fragment reads - see the staging / read notes in the
// body).  HISTORY.md 4.4c; measurements and ablations: tests/perf/gemm_probe.hip, profiles/r3_gemm_probe_*.txt.
//
// What bgemm256_kernel pays per 64-k stage (8 waves, 32 MFMAs each = 2 048 MFMA cycles per SIMD): 192 KB of fragment reads
// (768 LDS cycles) AND 64 ds_write_b128 wave-instructions whose VGPR -> LDS transfer costs 13 cycles each (832 cycles, not
// hidden by interleaved loads: MI355X_MICROARCH.md, LDS), issued by all eight waves in the same part of the stage - every wave
// reads, multiplies, writes and waits at the same time, so the matrix pipe idles while the LDS pipe works and vice versa
// (26-32 % MFMA busy).  Here
//   * tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPRs, no ds_write pass).  The LDS-DMA destination is
//     lane-linear (wave base + 16 lane), so the image cannot be padded: rows are 128 B and the 16-byte chunk c of row r sits in
//     slot c ^ ((r >> 1) & 7) - applied to the per-lane SOURCE address when staging and to the address of the ds_read_b128;
//     the 16 rows of each quarter-wave read group then fall on 16 distinct 16-byte slots of the 256-byte bank row;
//   * a stage is four phases of 8 MFMAs per wave (one 64 x 32 quadrant of the wave's 128 x 64 over the 64 k), each
//     {issue one staging unit, read the fragments the quadrant needs, counted vmcnt, barrier, MFMAs, barrier};
//   * waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave multiplies while the other reads and stages
//     (s_setprio 1 around the MFMAs);
//   * four staging units (16 KB each: 128 operand rows x 64 k) stay in flight across the barriers: the unit needed in phase P+1
//     was issued in phase P-4 or earlier, `s_waitcnt vmcnt(8)` (2 instructions per unit and thread) retires exactly it.
// LDS: 2 stages x 4 units x 16 KB = 128 KB.  Unit order: B rows of the waves' first 32 columns, A rows of their first 64 rows,
// B second 32, A second 64 - the order the phases need them.
// Hazards (the LDS-DMA writes are invisible to the compiler: all ordering here is by construction):
//   RAW  a unit is read one phase after the phase whose vmcnt retired it, i.e. behind a barrier that every wave passed after its
//        own wait (both wave groups: the later group's wait of phase P precedes barrier 2P+1, the earlier group's reads of phase
//        P+1 follow it);
//   WAR  unit u of stage s is restaged two phases (four barriers) after the phase that read it last; the reads are ordinary
//        ds_reads whose results the same phase's MFMAs consume, so they are complete when the wave reaches that phase's second
//        barrier;
//   EXIT every wave drains vmcnt(0) before the epilogue: an LDS-DMA write must not land after the workgroup released its LDS.
// Edges: rows beyond M / N and k chunks beyond K are requested with a vector offset outside the descriptor: the DMA writes zeros.
// Epilogues (EPI = the template's first argument): 0 - accumulators hold C^T blocks (MFMA operands swapped, exact), a lane stores 16
// (fp32) / 8 (bf16) contiguous bytes; 8192 - bf16 results through an XOR-swizzled image in the freed staging space, 2 rows x 512
// contiguous bytes per store instruction; 1024 - untransposed blocks, element-wise (fp32 rows a multiple of 4 KB apart);
// 2048 - opt-in SwiGLU backward on top of 0 (measured slower).  Results are bit-identical to bgemm256_kernel in every form.
#pragma once
#include <type_traits>
#include "bgemm.hpp"

namespace scldm {
namespace train {

constexpr int kG8Unit = 128 * 128;       // bytes: 128 operand rows x 64 k of bf16
constexpr int kBGemm8Lds = 8 * kG8Unit;  // 131 072

// byte offset of a unit: B units first (sub 0 / 1), then A; the two stages of a unit are adjacent (all read offsets fit the
// 16-bit immediate of ds_read_b128 next to one address register per k step)
__device__ __forceinline__ constexpr int g8_base(int is_a, int sub, int stage) { return ((is_a * 2 + sub) * 2 + stage) * kG8Unit; }

__device__ __forceinline__ void g8_lds_dma16(bg_u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff)
               : "memory");
}

// PROBE (tests/perf/gemm_probe.hip only; the product instantiates 0): timing ablations that break the result - 1: no staging
// inside the loop, 2: no fragment reads, 4: no MFMAs, 8: no stagger between the wave groups, 16: no s_setprio, 32: no barriers.
// 64 (result stays exact): the staging unit of a phase is issued between its MFMAs instead of ahead of its fragment reads;
// 128 (exact): two phases of 16 MFMAs per stage; 256 (exact): bf16 results stored element by element; 512: no stores;
// 1024 / 2048 / 8192 (exact): the epilogue forms listed above.
template <int PROBE = 0, bool A_KC = true, bool B_KC = true>
__device__ __forceinline__ void bgemm8_body(const BGemmArgs& g, const int bid) {
  extern __shared__ __attribute__((aligned(16))) char bgemm_smem[];
  int tile_id = (bid & 7) * g.per_xcd + (bid >> 3);   // (no split-K on this kernel; same XCD-aware numbering as bgemm256_body)
  if ((bid >> 3) >= g.per_xcd || tile_id >= g.tiles_m * g.tiles_n) return;
  const int tm = tile_id / g.tiles_n, tn = tile_id % g.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
  const int n_tiles = (g.K + kGK - 1) / kGK;
  constexpr bool LATE = (PROBE & 64) != 0;
  constexpr bool TRANS = (PROBE & 1024) == 0;   // accumulators hold C^T blocks (MFMA operands swapped): a lane owns 4 consecutive columns

  constexpr bool MERGED = (PROBE & 128) != 0;   // two phases of 16 MFMAs per stage (four barriers instead of eight)

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

  // ---- staging ------------------------------------------------------------------------------------------------------------------
  // KC operand: unit image [128 operand rows][64 k] (128-byte rows).  thread -> (unit row j = 64 q + tid / 8, slot tid % 8);
  //   source chunk c = slot ^ ((j >> 1) & 7).
  // MC operand (contiguous along m: both operands of a weight gradient): unit image [64 k][128 operand columns] (256-byte rows).
  //   thread -> (k row 32 q + tid / 16, slot tid % 16); the 64-byte segments of a row are XORed with (k row & 3) - the four k rows
  //   of a ds_read_b64_tr_b16 lane group then cover all 64 banks (what the 576-byte rows of bgemm256_kernel do by padding):
  //   source chunk c = slot ^ ((k row & 3) << 2).  Chunk c holds 8 consecutive m: A unit: wave row wm' = c / 8, m = 128 wm' + 64 sub
  //   + 8 (c % 8); B unit: wn' = c / 4, n = 64 wn' + 32 sub + 8 (c % 4).
  const int c_src = (tid & 7) ^ ((tid >> 4) & 7);
  const int c_mc = (tid & 15) ^ (((tid >> 4) & 3) << 2);
  const int brow = (tid >> 8) * 64 + ((tid >> 3) & 31);
  const int a_mc_m = (c_mc >> 3) * 128 + (c_mc & 7) * 8, b_mc_n = (c_mc >> 2) * 64 + (c_mc & 3) * 8;
  const unsigned voff_a = A_KC ? (unsigned)(tid >> 3) * (unsigned)g.lda * 2u + (unsigned)c_src * 16u
                               : (unsigned)(tid >> 4) * (unsigned)g.lda * 2u + (unsigned)a_mc_m * 2u;
  const unsigned voff_b = B_KC ? (unsigned)brow * (unsigned)g.ldb * 2u + (unsigned)c_src * 16u
                               : (unsigned)(tid >> 4) * (unsigned)g.ldb * 2u + (unsigned)b_mc_n * 2u;
  unsigned ok_a = 0, ok_b = 0;   // bit 2 sub + q: the row (KC) / 8-column chunk (MC: same for both q) this thread stages for (sub, q) exists
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ok_a |= (unsigned)(A_KC ? m0 + q * 128 + sub * 64 + (tid >> 3) < g.M : m0 + sub * 64 + a_mc_m < g.M) << (2 * sub + q);
      ok_b |= (unsigned)(B_KC ? n0 + q * 128 + sub * 32 + brow < g.N : n0 + sub * 32 + b_mc_n < g.N) << (2 * sub + q);
    }
  // unit u: 0 = B sub 0, 1 = A sub 0, 2 = B sub 1, 3 = A sub 1 (the order the phases consume them)
  auto issue = [&](auto u_tag, int tt) {
    constexpr int U = decltype(u_tag)::value;
    constexpr int IS_A = U & 1, SUB = U >> 1;
    constexpr bool KC = IS_A ? A_KC : B_KC;
    const int stage = tt & 1;
    const int k0 = tt * kGK;
    if ((PROBE & 1) && tt >= 2) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const bool k_ok = tt < n_tiles && (KC ? k0 + c_src * 8 < g.K : k0 + q * 32 + (tid >> 4) < g.K);
      const bool ok = k_ok && (((IS_A ? ok_a : ok_b) >> (2 * SUB + q)) & 1u);
      const unsigned voff = ok ? (IS_A ? voff_a : voff_b) : kOob;
      const unsigned ld = (unsigned)(IS_A ? g.lda : g.ldb);
      const unsigned row0 = IS_A ? (unsigned)(m0 + SUB * 64) : (unsigned)(n0 + SUB * 32);
      const unsigned soff = KC ? ((row0 + q * 128u) * ld + (unsigned)k0) * 2u : ((unsigned)(k0 + q * 32) * ld + row0) * 2u;
      const unsigned dst = smem0 + (unsigned)(((IS_A * 2 + SUB) * 2) * kG8Unit) + (unsigned)stage * kG8Unit + q * 8192u + (unsigned)wave * 1024u;
      g8_lds_dma16(IS_A ? ra : rb, voff, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane(dst));
    }
  };
  using U0 = std::integral_constant<int, 0>;
  using U1 = std::integral_constant<int, 1>;
  using U2 = std::integral_constant<int, 2>;
  using U3 = std::integral_constant<int, 3>;

  // ---- fragment reads ---------------------------------------------------------------------------------------------------------
  // KC: lane -> row lane % 32 of a 32-row block, chunk 2 ks + lane / 32, swizzled (ds_read_b128).
  // MC: two ds_read_b64_tr_b16 per fragment (k rows 16 ks + 8 (g / 2) + i / 4 and + 4, g = lane / 16, i = lane % 16; 4 consecutive
  //     columns at 16 (g % 2) + 4 (i % 4) of the block's 32): segment (block's 64-byte segment) ^ (i / 4), see the staging note.
  const char* rd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    rd[ks] = bgemm_smem + (lane & 31) * 128 + (((2 * ks + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
  const int a_wave = wm * 8192, b_wave = wn * 4096;   // 64 / 32 unit rows per wave
  const int tg = lane >> 4, ti = lane & 15;
  const int tr_row = (8 * (tg >> 1) + (ti >> 2)) * 256 + 32 * (tg & 1) + 8 * (ti & 3);
  const char* rd_a_mc[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) rd_a_mc[blk] = bgemm_smem + tr_row + (((wm * 2 + blk) ^ (ti >> 2)) << 6);
  const char* rd_b_mc = bgemm_smem + tr_row + ((wn ^ (ti >> 2)) << 6);
  auto tr_frag = [&](const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * 256));
    union { s16x4 s[2]; bf16x8 f; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.f;
  };
  auto frag_a = [&](int ks, int sub, int blk, int stage) {
    if constexpr ((PROBE & 2) != 0) {
      bf16x8 v;
      asm volatile("; keep" : "=v"(v));
      return v;
    } else if constexpr (A_KC) {
      return *reinterpret_cast<const bf16x8*>(rd[ks] + g8_base(1, sub, stage) + a_wave + blk * 4096);
    } else {
      return tr_frag(rd_a_mc[blk] + g8_base(1, sub, stage) + ks * 4096);
    }
  };
  auto frag_b = [&](int ks, int sub, int stage) {
    if constexpr ((PROBE & 2) != 0) {
      bf16x8 v;
      asm volatile("; keep" : "=v"(v));
      return v;
    } else if constexpr (B_KC) {
      return *reinterpret_cast<const bf16x8*>(rd[ks] + g8_base(0, sub, stage) + b_wave);
    } else {
      return tr_frag(rd_b_mc + g8_base(0, sub, stage) + ks * 4096);
    }
  };
  // bias gradient of a weight-gradient product (A = dy, contiguous along m): rowsum[m] = sum_k A(m, k), taken from the A fragments
  // in the tiles of the first tile column - or, with rowsum_split, by every tile column for its share of the stages (all tiles of the
  // launch then take equally long; the host adds the partial vectors).  The four waves that share a row block (wn = 0..3) each take the k step ks == wn of
  // every stage (one v_dot2c_f32_bf16 per pair: 16 per stage and wave), in the read sections of phases 1 and 3 - beside the other wave
  // group's MFMAs; between the wave's own MFMAs the same 16 instructions cost the tile 40 % - and the partial sums meet in LDS after
  // the loop.
  const bool want_rs = !A_KC && g.rowsum != nullptr && (g.rowsum_split || tn == 0);
  int rs_next = g.rowsum_split ? tn : 0;   // the next stage whose row sums this tile takes (rowsum_split: every tiles_n-th, else all)
  const int rs_stride = g.rowsum_split ? g.tiles_n : 1;
  float rs[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  auto add_rs = [&](const bf16x8& f, float& r) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t one;
    one[0] = (__bf16)1.0f;
    one[1] = (__bf16)1.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bf16x2_t a;
      a[0] = f[2 * e];
      a[1] = f[2 * e + 1];
      r = __builtin_amdgcn_fdot2_f32_bf16(a, one, r, false);
    }
  };
  auto mma = [&](const bf16x8& a, const bf16x8& b, f32x16& c) {
    if constexpr ((PROBE & 4) != 0) asm volatile("; use" ::"v"(a), "v"(b));
    else if constexpr (TRANS) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0);   // (a b = b a exactly: same sums, transposed block)
    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
  bf16x8 fa[2][4], fb0[4], fb1[4];

  // ---- prologue -----------------------------------------------------------------------------------------------------------
  issue(U0{}, 0);
  issue(U1{}, 0);
  issue(U2{}, 0);
  issue(U3{}, 0);
  issue(U0{}, 1);
  issue(U1{}, 1);
  if constexpr (LATE || MERGED) issue(U2{}, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1 && !(PROBE & 8)) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind
  __builtin_amdgcn_sched_barrier(0);

#define G8_PHASE_SYNC()                                    \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         \
  if (!(PROBE & 32)) __builtin_amdgcn_s_barrier();         \
  __builtin_amdgcn_sched_barrier(0);                       \
  if (!(PROBE & 16)) __builtin_amdgcn_s_setprio(1)
#define G8_PHASE_END()                                     \
  if (!(PROBE & 16)) __builtin_amdgcn_s_setprio(0);        \
  __builtin_amdgcn_sched_barrier(0);                       \
  if (!(PROBE & 32)) __builtin_amdgcn_s_barrier();         \
  __builtin_amdgcn_sched_barrier(0)

  auto rowsum_step = [&](int sub, int t) {
    if constexpr (!A_KC) {
      if (want_rs && t == rs_next) {
        if (sub == 1) rs_next += rs_stride;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks == wn) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) add_rs(fa[blk][ks], rs[sub][blk]);
          }
      }
    }
  };
  static_assert(A_KC || !((PROBE & 128) != 0), "the merged-phase schedule carries no row sums");
  auto tile = [&](auto st_tag, int t) {
    constexpr int ST = decltype(st_tag)::value;
    // phase 0: A sub 0 x B sub 0
    if constexpr (!LATE) issue(U2{}, t + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb0[ks] = frag_b(ks, 0, ST);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[blk][ks] = frag_a(ks, 0, blk, ST);
    G8_PHASE_SYNC();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) mma(fa[blk][ks], fb0[ks], acc[blk][0]);
      if constexpr (LATE) {
        if (ks == 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue(U3{}, t + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    G8_PHASE_END();
    // phase 1: A sub 0 x B sub 1
    if constexpr (!LATE) issue(U3{}, t + 1);
    rowsum_step(0, t);   // (A sub 0 is still in fa: VALU work belongs here, beside the other wave group's MFMAs, not between this wave's own)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb1[ks] = frag_b(ks, 1, ST);
    G8_PHASE_SYNC();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) mma(fa[blk][ks], fb1[ks], acc[blk][1]);
      if constexpr (LATE) {
        if (ks == 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue(U0{}, t + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    G8_PHASE_END();
    // phase 2: A sub 1 x B sub 1
    if constexpr (!LATE) issue(U0{}, t + 2);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[blk][ks] = frag_a(ks, 1, blk, ST);
    G8_PHASE_SYNC();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) mma(fa[blk][ks], fb1[ks], acc[2 + blk][1]);
      if constexpr (LATE) {
        if (ks == 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue(U1{}, t + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    G8_PHASE_END();
    // phase 3: A sub 1 x B sub 0 (both in registers)
    if constexpr (!LATE) issue(U1{}, t + 2);
    rowsum_step(1, t);
    G8_PHASE_SYNC();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) mma(fa[blk][ks], fb0[ks], acc[2 + blk][0]);
      if constexpr (LATE) {
        if (ks == 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue(U2{}, t + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    G8_PHASE_END();
  };
  // MERGED: phase A = A sub 0 x (B sub 0, B sub 1), phase B = A sub 1 x (B sub 1, B sub 0).  Units are issued between the MFMAs:
  // in A(t) U3(t+1) [its slot was read last in B(t-1)], in B(t) U0, U1, U2 of t+2 [read last in A(t)]; a wave's issue lies
  // behind the phase's first barrier, which the other group passed after completing the previous phase's reads.
  // Waits: ahead of A's first barrier vmcnt(6) (U0-2 of the next stage may fly, U3 of this stage has landed for B), ahead of B's
  // vmcnt(2) (only U3 of the next stage may fly: U0-2 of the next stage have landed for its A).
  auto tile2 = [&](auto st_tag, int t) {
    constexpr int ST = decltype(st_tag)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb0[ks] = frag_b(ks, 0, ST);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[blk][ks] = frag_a(ks, 0, blk, ST);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb1[ks] = frag_b(ks, 1, ST);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (!(PROBE & 32)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!(PROBE & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) mma(fa[blk][ks], fb0[ks], acc[blk][0]);
      if (ks == 0) {
        __builtin_amdgcn_sched_barrier(0);
        issue(U3{}, t + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) mma(fa[blk][ks], fb1[ks], acc[blk][1]);
    G8_PHASE_END();
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[blk][ks] = frag_a(ks, 1, blk, ST);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (!(PROBE & 32)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!(PROBE & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) mma(fa[blk][ks], fb1[ks], acc[2 + blk][1]);
      if (ks < 3) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) issue(U0{}, t + 2);
        if (ks == 1) issue(U1{}, t + 2);
        if (ks == 2) issue(U2{}, t + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) mma(fa[blk][ks], fb0[ks], acc[2 + blk][0]);
    G8_PHASE_END();
  };
  int t = 0;
  if constexpr (MERGED) {
    for (; t + 1 < n_tiles; t += 2) {
      tile2(std::integral_constant<int, 0>{}, t);
      tile2(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < n_tiles) tile2(std::integral_constant<int, 0>{}, t);
  } else {
    for (; t + 1 < n_tiles; t += 2) {
      tile(std::integral_constant<int, 0>{}, t);
      tile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < n_tiles) tile(std::integral_constant<int, 0>{}, t);
  }
#undef G8_PHASE_SYNC
#undef G8_PHASE_END
  if (wm == 0 && !(PROBE & 8)) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last phases' requests lie beyond K: zeros into stages nobody reads)
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!A_KC) {
    if (want_rs) {   // (uniform over the workgroup)
      __builtin_amdgcn_s_barrier();   // every wave's LDS-DMA has landed (vmcnt(0) above): the staging area is free
      float* red = reinterpret_cast<float*>(bgemm_smem);   // [4 k steps][256 rows]
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const float tot = xor32_sum(rs[sub][blk]);   // lanes l and l + 32 hold the two k halves of row l % 32
          if (lane < 32) red[wn * 256 + wm * 128 + (sub * 2 + blk) * 32 + lane] = tot;
        }
      lds_barrier();
      if (tid < 256 && m0 + tid < g.M)
        g.rowsum[(g.rowsum_split ? (long)tn * g.M : 0L) + m0 + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
    }
  }
  if constexpr ((PROBE & 512) != 0) {   // timing probe: no stores at all
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) asm volatile("; keep" ::"v"(acc[i][k]));
    return;
  }
  if constexpr (TRANS && (PROBE & 8192) != 0) {
    // bf16 result through LDS (the 256 x 256 bf16 tile is exactly the 128 KB of staging space, free now): every lane writes its
    // 8-byte pieces into a [256][256] image whose 16-byte chunks are XORed with the row (32 rows of one chunk column -> 32 different
    // chunks), then the workgroup reads the image row by row - a store instruction covers 2 rows x 512 contiguous bytes instead of
    // 32 rows x 16 bytes.  N % 4 == 0 (rows of N % 8 == 4 elements: 8-byte aligned rows, the host sends those here only when C16 is).
    if (g.C16) {
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
      __builtin_amdgcn_s_barrier();   // every wave is past its last fragment read and its last LDS-DMA has landed (vmcnt(0) above)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 128 + i * 32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = wn * 64 + k * 32 + 8 * q + 4 * (lane >> 5);   // column inside the tile
            const int n = n0 + nl;
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(acc[i][k][4 * q + e] + ((g.bias && n + e < g.N) ? g.bias[n + e] : 0.f));
            const int chunk = (nl >> 3) ^ (r & 31);
            *reinterpret_cast<bf16x4_t*>(bgemm_smem + r * 512 + chunk * 16 + (nl & 4) * 2) = o;
          }
      }
      lds_barrier();
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int r = it * 16 + wave * 2 + (lane >> 5), j = lane & 31;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(bgemm_smem + r * 512 + ((j ^ (r & 31)) << 4));
        if (m0 + r < g.M && n0 + j * 8 < g.N) {
          __bf16* dst = g.C16 + (long)(m0 + r) * g.N + n0 + j * 8;
          if (n0 + j * 8 + 8 <= g.N) {
            *reinterpret_cast<bf16x8*>(dst) = v;
          } else {   // N % 8 == 4: the row's last piece is half a chunk
            bf16x4_t lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) lo[e] = v[e];
            *reinterpret_cast<bf16x4_t*>(dst) = lo;
          }
        }
      }
      return;
    }
  }
  if constexpr (TRANS) {
    // acc[i][k][r] = C(m0 + 128 wm + 32 i + lane % 32, n0 + 64 wn + 32 k + acc_row(r, lane / 32)): registers 4 q .. 4 q + 3 are the
    // four consecutive columns 8 q + 4 (lane / 32) .. + 3 of the lane's row - one 16-byte (fp32) or 8-byte (bf16) store per lane and
    // q instead of four 4- / 2-byte ones (the epilogue of the element-wise form costs as much as the 16 stages of a K = 1 024 tile:
    // tests/perf/gemm_probe.hip).  The host sends only 16-byte aligned C / ldc % 4 == 0 (bf16: N % 4 == 0) products here.
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 128 + i * 32 + (lane & 31);
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + k * 32 + 8 * q + 4 * (lane >> 5);
          if (m >= g.M || n >= g.N) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][k][4 * q + e] + ((g.bias && n + e < g.N) ? g.bias[n + e] : 0.f);
          if constexpr ((PROBE & 2048) != 0) {
            // c_proj's data gradient with the SwiGLU backward in its epilogue: v = d hid[m][n..n+3] never leaves the registers;
            // da = d hid * b * s (1 + a (1 - s)), db = d hid * a * s, s = sigmoid(a)   (MLP.forward, layers.py:172-174)
            const bf16x4_t a4 = *reinterpret_cast<const bf16x4_t*>(g.ep_a + (long)m * g.ep_n + n);
            const bf16x4_t b4 = *reinterpret_cast<const bf16x4_t*>(g.ep_b + (long)m * g.ep_n + n);
            const f32x4 av = {(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]}, bv = {(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
            bf16x4_t oa, ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float sg = 1.0f / (1.0f + __expf(-av[e]));   // (the arithmetic of swiglu_bwd_kernel)
              oa[e] = (__bf16)(v[e] * bv[e] * sg * (1.0f + av[e] * (1.0f - sg)));
              ob[e] = (__bf16)(v[e] * av[e] * sg);
            }
            *reinterpret_cast<bf16x4_t*>(g.ep_o1 + (long)m * g.ep_ldo + n) = oa;
            *reinterpret_cast<bf16x4_t*>(g.ep_o2 + (long)m * g.ep_ldo + n) = ob;
            if (n + 4 >= g.N)
              for (int p = g.N; p < g.ep_pad; ++p) g.ep_o1[(long)m * g.ep_ldo + p] = g.ep_o2[(long)m * g.ep_ldo + p] = (__bf16)0.f;
            continue;
          }
          if (g.C16) {
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4_t*>(g.C16 + (long)m * g.N + n) = o;
          } else if (n + 3 < g.N) {
            f32x4* p = reinterpret_cast<f32x4*>(g.C + (long)m * g.ldc + n);
            if (g.accumulate) v += *p;
            *p = v;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < g.N) {
                float* p = g.C + (long)m * g.ldc + n + e;
                *p = g.accumulate ? *p + v[e] : v[e];
              }
          }
        }
    }
    return;
  }
  if (g.C16 && !(PROBE & 256)) {
    // bf16 result: neighbouring lanes hold neighbouring columns of the same row.  Lanes 2j / 2j+1 trade one register of each
    // (r, r+1) pair (DPP quad_perm 1,0,3,2), so that the even lane owns columns (c, c+1) of row(r) and the odd lane the same two
    // columns of row(r+1): one 4-byte store per lane and register pair instead of two 2-byte ones.
    const int odd = lane & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int n = n0 + wn * 64 + k * 32 + (lane & 30);   // even column of the pair
        const float b0 = (g.bias && n < g.N) ? g.bias[n] : 0.f, b1 = (g.bias && n + 1 < g.N) ? g.bias[n + 1] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float give = odd ? acc[i][k][r] : acc[i][k][r + 1];
          const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
          const float lo = (odd ? got : acc[i][k][r]) + b0, hi = (odd ? acc[i][k][r + 1] : got) + b1;
          const int m = m0 + wm * 128 + i * 32 + acc_row(r + odd, lane >> 5);
          if (m < g.M && n < g.N) {   // (N is a multiple of 8 on this path: the pair is inside or outside together)
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            bf16x2_t o;
            o[0] = (__bf16)lo;
            o[1] = (__bf16)hi;
            *reinterpret_cast<bf16x2_t*>(g.C16 + (long)m * g.N + n) = o;
          }
        }
      }
    return;
  }
  float* __restrict__ C = g.C;
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

// PROBE 0: transposed accumulator blocks + vector epilogue; PROBE 1024: element-wise epilogue (fp32 outputs whose row stride is a
// multiple of 4 KB - the 1 024-wide activations: the 32 rows of a vector store all fall on one memory channel, 52.6 against 43.6 us
// on [16 384 x 1 024 x 1 024]; everything else is 15 % faster with the vector epilogue).  Other values: tests/perf/gemm_probe.hip.
template <int PROBE, bool A_KC = true, bool B_KC = true>
__global__ __launch_bounds__(512) void bgemm8_kernel(const BGemmArgs g) {
  bgemm8_body<PROBE, A_KC, B_KC>(g, (int)blockIdx.x);
}
// Several independent products in one launch (the five weight gradients of a layer, see bgemm256_batch_kernel): both operands
// contiguous along m, element-wise epilogue (four of the five outputs have 4 KB rows).
__global__ __launch_bounds__(512) void bgemm8_batch_kernel(const BGemmBatch b) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < kBGemmBatchMax; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first[i]) j = i;
  bgemm8_body<1024, false, false>(b.job[j], (int)blockIdx.x - b.first[j]);
}

}  // namespace train
}  // namespace scldm
