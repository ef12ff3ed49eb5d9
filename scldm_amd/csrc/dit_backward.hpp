// Fused BACKWARD of one adaLN-Zero DiT layer for gfx950 (training path, bf16 operands / fp32 accumulation): one launch =
// one transformer block over all samples, one workgroup per 64-token tile (4 samples).  The only saved state it reads is the
// record the REC forward kernel wrote (dit_forward.hpp): the layer's input residual (fp32) and the two gated branch outputs
// y1, y2 (bf16); everything else (LayerNorms, q/k/v, softmax, SwiGLU pre-activations) is recomputed in registers / LDS.
//
// Differentiates (reference arithmetic): Block.forward adaLN branch src/scldm/layers.py:208-221, modulate :91-94,
// SelfAttention.forward :143-158, MLP.forward :161-174.  The reference gets these gradients from torch autograd.
//
// Outputs: the gradient w.r.t. the layer input (in place, same private tile layout as the residual hand-off), the gradient
// w.r.t. the six adaLN vectors of every sample (dmod), and - for the weight gradients, which are sums over ALL tokens and so
// cannot be owned by a tile - the bf16 operand pairs (dY, X) of the layer's five weight-gradient GEMMs in plain [token][feature]
// layout; wgrad_bf16_kernel (train_fused.hip) contracts them over the token axis afterwards.
//
// Shape: 8 waves, wave w owns the 32 features [32w, 32w+32) = attention head w; GEMMs are computed transposed exactly as in
// the forward kernel (weights streamed L2 -> VGPR ring as pre-packed fragments, activations as [token][feature] bf16 LDS
// images), so every matrix product below is a gemm_pass over a different packed stream (dit_aux.hpp: pack_bwd_val).
//
// Included once per 16-bit operand policy (round 4): the including translation unit defines SCLDM_BWD_NS (namespace of this
// instantiation) and SCLDM_BWD_OP (OpBF16, or OpFP16 = the reference's TF32 mantissa; its gradients are loss-scaled by the caller,
// train_api.hip) before each inclusion.  Nothing below depends on the element type beyond OP.
#include "bwd_layout.hpp"
#include "dit_forward.hpp"
#if !defined(SCLDM_BWD_NS) || !defined(SCLDM_BWD_OP)
#error "define SCLDM_BWD_NS and SCLDM_BWD_OP before including dit_backward.hpp"
#endif

namespace scldm {
namespace SCLDM_BWD_NS {

using OP = SCLDM_BWD_OP;
using E = OP::E;
using Frag = OP::Frag;
using Quad = OP::Quad;
// SCLDM_BWD_NTT: 32-token row tiles per workgroup.  2 = the 64-token tile of the forward kernel's record; 1 = half of one (round 5:
// while 32-token tiles still get a CU each - at most 512 cells - the layer's walk is shorter with half the MFMAs and LDS fragment
// reads per k-step on the same weight stream; the record and the gradient keep the 64-token geometry, a workgroup reads its half).
#ifndef SCLDM_BWD_NTT
#define SCLDM_BWD_NTT 2
#endif
constexpr int NTT = SCLDM_BWD_NTT, NW = 8, NT = 64 * NW, TM = 32 * NTT, NS = 2 * NTT;
static_assert(NTT == 1 || NTT == 2, "the record is laid out in 64-token tiles");
#ifndef SCLDM_BWD_PF
#define SCLDM_BWD_PF 8
#endif
constexpr int PF = SCLDM_BWD_PF;
// Cache policy of the streaming traffic (A/B switches, round 5): the operand-pair stores (written once here, read once by the
// weight-gradient GEMM) and the record loads (read once per phase) pass through the XCD's 4 MB L2 next to the 2.8 MB backward weight
// stream that every workgroup of the XCD re-reads.  0 = default policy, 1 = nt, 2 = sc1 (stores: write through, drop the line),
// 3 = timing proxy without the stores (98 -> 76 us per launch at 1 024 cells: what the twelve store bursts of a layer cost - vmcnt is
// in-order, so the weight ring's next wait after a burst is also a wait for the burst's write acknowledgements).
#ifndef SCLDM_BWD_PAIR_ST
#define SCLDM_BWD_PAIR_ST 0
#endif
#ifndef SCLDM_BWD_REC_NT
#define SCLDM_BWD_REC_NT 0
#endif
// debugging aid: 1 = wait for every vector-memory operation in flight after each group of operand-pair stores
// which phases request their global reads ahead of use (bit 0: kernel start, bit 1: LayerNorm-2 backward, bit 2: before the K = 768 pass)
#ifndef SCLDM_BWD_HOIST
#define SCLDM_BWD_HOIST 10
#endif
// 1: the LayerNorm reductions alternate between two LDS buffers and need ONE barrier each; 0: one buffer, two barriers
#ifndef SCLDM_BWD_RED2
#define SCLDM_BWD_RED2 1
#endif
// 1: the weight ring is parked (not refilled) across the attention core, whose operand fragments are the kernel's register peak
#ifndef SCLDM_BWD_PARK
#define SCLDM_BWD_PARK 1
#endif
// 1: d x_mid stays in 32 registers from the LayerNorm-2 backward to the LayerNorm-1 backward (one 64 KB store and one load per tile less)
#ifndef SCLDM_BWD_KEEP_DX
#define SCLDM_BWD_KEEP_DX 1
#endif
// 1: x_in stays in registers from the LayerNorm-1 forward recompute to the LayerNorm-1 backward as well
#ifndef SCLDM_BWD_KEEP_XIN
#define SCLDM_BWD_KEEP_XIN 1
#endif
// 1: d x_out stays in registers across the SwiGLU chunks too (needs KEEP_DX: the gradient is then read once and written once per layer)
#ifndef SCLDM_BWD_KEEP_DXOUT
#define SCLDM_BWD_KEEP_DXOUT 1
#endif
// 1: x_in stays in registers across the SwiGLU chunks as well (x_mid is a temporary of phases 1-2 and of the LayerNorm-2 backward):
// the record's residual is then read once per layer
#ifndef SCLDM_BWD_KEEP_XIN_MLP
#define SCLDM_BWD_KEEP_XIN_MLP 1
#endif
#ifndef SCLDM_BWD_ST_WAIT
#define SCLDM_BWD_ST_WAIT 0
#endif   // k-steps of weight-ring run-ahead: one wave gets two MFMAs (64 cycles) out of a fragment, an L2 round trip is ~10 of those
constexpr int XA_LD = kD + 8, DADB_LD = 2 * kBwdChunk + 8, DQKV_LD = 3 * kD + 8;   // bf16 elements per image row (+16 B pad)
constexpr int R0_OFF = 0;                                   // h2 image, later h1 image
constexpr int R1_OFF = R0_OFF + TM * XA_LD * 2;             // dy2 image, later dy1 image
constexpr int R2_OFF = R1_OFF + TM * XA_LD * 2;             // [da | db] image of one chunk
constexpr int DQKV_BYTES = TM * DQKV_LD * 2;                // aliases R0..R2 once the q/k/v/dao passes are done
constexpr int TR_LD = 36;                                   // elements per row of a wave's 32x32 transpose scratch (72 B: conflict-free ds_read_b64)
constexpr int TR_OFF = DQKV_BYTES;                          // inside R2's tail, behind the dqkv image
constexpr int TR_BYTES = 32 * TR_LD * 2;
constexpr int R_END_IMG = R2_OFF + TM * DADB_LD * 2, R_END_TR = TR_OFF + NW * TR_BYTES;
constexpr int R_END = R_END_IMG > R_END_TR ? R_END_IMG : R_END_TR;   // (64-token tiles: the scratch fits in R2's tail; 32-token tiles: 1 KB past it)
static_assert(TR_OFF % 8 == 0 && R_END % 16 == 0, "transpose scratch alignment");
constexpr int MOD_OFF = R_END;
constexpr int MOD_BYTES = NS * kModBlock * 2;               // fp16 copies of the six adaLN vectors of the tile's samples
constexpr int RED_OFF = MOD_OFF + MOD_BYTES;
constexpr int RED_BYTES = 2 * NW * TM * 4;                  // one reduction (two values per token and wave); TWO buffers, used alternately:
constexpr int BIAS_OFF = RED_OFF + 2 * RED_BYTES;           // a reduction's readers are past the NEXT reduction's barrier before its buffer is written again
constexpr int LDS_BYTES = BIAS_OFF + 3 * kD * 4;

struct BwdArgs {
  const float* x_in;      // record: residual entering this layer (tile layout of the 4-wave forward kernel)
  const E* y1;            // record: c_proj(attention) + bias
  const E* y2;            // record: MLP output
  float* dx;              // in: gradient w.r.t. the layer output; out: w.r.t. the layer input (tile layout)
  const float* mod;       // (n, mod_stride) adaLN vectors; this layer's six at mod_off
  float* dmod;            // same indexing: gradient w.r.t. the adaLN vectors
  int mod_stride, mod_off;
  const E* w_stream;      // this layer's backward stream (pack_bwd_val)
  const float* b_qkv;     // (768) c_attn bias of this layer
  // operand pairs of the weight-gradient GEMMs, plain [token][ld] bf16
  E *e_h1, *e_dqkv, *e_ao, *e_dy1, *e_h2, *e_da, *e_db, *e_hid, *e_dy2;
  int n;                  // real samples; the last tile may be padded (its padding samples repeat the last real one, gradient zero)
  float eps, attn_scale, attn_scale_log2e;
  unsigned long long* dbg;   // optional phase stamps [block][wave][16] (SCLDM_BWD_DBG=1; nullptr otherwise)
};

__device__ __forceinline__ void wave_sync() {   // orders this wave's LDS writes before its following LDS reads (compiler + hardware)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// sum over the 16 lanes of a DPP row (= the 16 tokens of one sample), result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}

// Global memory goes through buffer descriptors (round 5): wave-uniform base in SGPRs + ONE per-lane byte offset + immediates.
// With flat 64-bit addresses hipcc kept one VGPR pair per access alive, spilled them, and reloaded each from scratch in front of
// its load behind an s_waitcnt vmcnt(0) - the eight loads of a record tile ran one round trip after the other (and every such
// wait drained the weight ring and the operand-pair stores in flight).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
// destination of an operand-pair tile: rows = tokens (ld elements apart); wave-uniform base pointer (SGPR pair: the token tile's
// first row and the wave's first column folded in) + this lane's row / half offset in `voff` (bytes) -> global_store ... saddr.
// (Round 5: MUBUF stores with an SGPR soffset gave run-to-run different operand arrays on MI355X - the store data registers were
// re-used by VALU writes two issue slots later, a hazard hipcc does not pad for - so stores take the global saddr form.)
typedef __attribute__((address_space(1))) char gchar;       // global address space: a pointer rebuilt from integers is generic (flat_store) otherwise
typedef __attribute__((address_space(1))) u32x4 g_u32x4;
typedef __attribute__((address_space(1))) f32x4 g_f32x4;
struct PairDst {
  gchar* base;
  unsigned voff;   // (c32 * ld + 8 * hh) * 2
  bool on;
};
__device__ __forceinline__ gchar* uniform_ptr(const void* p) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return (gchar*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void pair_store(const PairDst& d, unsigned imm, const u32x4 v) {
  g_u32x4* gp = (g_u32x4*)(d.base + (d.voff + imm));
#if SCLDM_BWD_PAIR_ST == 3   // timing proxy: no operand-pair stores at all (weight gradients are garbage)
  asm volatile("" :: "v"(v), "v"(gp));
#elif SCLDM_BWD_PAIR_ST == 1
  __builtin_nontemporal_store(v, gp);
#else
  *gp = v;
#endif
}
// quads q (q0) and q+1 (q1) of a 32-feature tile -> 16 bytes per lane (features f8 + 8*hh .. +7 of the lane's token), stored
// to the LDS image row and, when g.on, to the same position of the plain global operand array
__device__ __forceinline__ void put_pair(E* lrow, const PairDst& g, int f8, int hh, const Quad& q0, const Quad& q1) {
  union { Quad q; unsigned u[2]; } a, b;
  a.q = q0; b.q = q1;
  halfwave_pair(a.u[0], b.u[0]);
  halfwave_pair(a.u[1], b.u[1]);
  const u32x4 v = u32x4{a.u[0], a.u[1], b.u[0], b.u[1]};
  if (lrow) *reinterpret_cast<u32x4*>(lrow + f8 + 8 * hh) = v;
  if (g.on) pair_store(g, (unsigned)f8 * 2u, v);
}
// a whole accumulator tile (32 features x 32 tokens of token tile tt) -> image columns [col0, col0 + 32) / operand array columns
// [g's first column, + 32)
__device__ __forceinline__ void put_tile(const float (&t)[16], E* lrow, const PairDst& g, int col0, int hh) {
#pragma unroll
  for (int q = 0; q < 4; q += 2) {
    union { Quad q; unsigned u[2]; } a, b;
    a.q = OP::pack4(t[q * 4 + 0], t[q * 4 + 1], t[q * 4 + 2], t[q * 4 + 3]);
    b.q = OP::pack4(t[q * 4 + 4], t[q * 4 + 5], t[q * 4 + 6], t[q * 4 + 7]);
    halfwave_pair(a.u[0], b.u[0]);
    halfwave_pair(a.u[1], b.u[1]);
    const u32x4 v = u32x4{a.u[0], a.u[1], b.u[0], b.u[1]};
    if (lrow) *reinterpret_cast<u32x4*>(lrow + col0 + q * 8 + 8 * hh) = v;
    if (g.on) pair_store(g, (unsigned)(q * 8) * 2u, v);
  }
#if SCLDM_BWD_ST_WAIT
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#endif
}

// LDS byte offsets above 64 KB do not fit a ds instruction's 16-bit immediate: hipcc then builds ONE address register per
// access (base + constant), keeps all of them alive for the next use of the same address and spills them (round 5: 40 of the
// kernel's 50 spilled registers were such addresses).  An offset that went through opaque() is a plain register to the compiler:
// region base + lane part are added once, every access is that register + a small immediate.
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// accumulator tile X^T[row = acc_row(r, hh)][col = c32] -> the wave's transpose scratch T[row][col]; T = scratch + (4 hh * TR_LD + c32)
__device__ __forceinline__ void tr_write(E* T, const f32x16& x) {
#pragma unroll
  for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2)) * TR_LD] = (E)x[r];
}
// the same from the two packed operand fragments of a tile (F[h][j] = accumulator register 8 h + j, already rounded to E)
__device__ __forceinline__ void tr_write_frags(E* T, const Frag (&F)[2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) T[((j & 3) + 8 * (2 * h + (j >> 2))) * TR_LD] = F[h][j];
}
// the query's eight own-sample values (keys 16 sp + acc_row(i, hh)) -> T[key][query]; the other sample's keys are zero
__device__ __forceinline__ void tr_write_own(E* T, const float (&v)[8], int sp) {
  E* own = T + 16 * sp * TR_LD;
  E* other = T + 16 * (1 - sp) * TR_LD;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    own[((i & 3) + 8 * (i >> 2)) * TR_LD] = (E)v[i];
    other[((i & 3) + 8 * (i >> 2)) * TR_LD] = (E)0.f;
  }
}
// fragment h (columns 16h .. 16h+15 as the k axis) of row c32, k slots in the accumulator's row order (slot (hh, j) <-> column
// 16h + (j&3) + 8 (j>>2) + 4 hh): the order pack8() gives the register-built partner operand
__device__ __forceinline__ Frag tr_read(const E* T, int h) {   // T = scratch + (c32 * TR_LD + 4 hh)
  const E* p = T + 16 * h;
  const Quad a = *reinterpret_cast<const Quad*>(p), b = *reinterpret_cast<const Quad*>(p + 8);
  Frag f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
  return f;
}

__global__ __launch_bounds__(NT, 1) void dit_backward_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  E* R0 = reinterpret_cast<E*>(smem + R0_OFF);
  E* R1 = reinterpret_cast<E*>(smem + R1_OFF);
  E* R2 = reinterpret_cast<E*>(smem + R2_OFF);
  E* DQKV = reinterpret_cast<E*>(smem);
  using ModE = OP::ModE;
  ModE* MOD = reinterpret_cast<ModE*>(smem + MOD_OFF);
  float* BIAS = reinterpret_cast<float*>(smem + BIAS_OFF);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c32 = lane & 31, hh = lane >> 5, sp = c32 >> 4;
  const int tile = blockIdx.x;
  const int tok0 = tile * TM, smp0 = tile * NS;
  const int fb = wave * 32;   // first feature / head dimension owned by this wave
  // Every lane-dependent address below is rebuilt AT ITS USE SITE from a laundered lane id (lane_now()): hipcc otherwise hoists
  // the address arithmetic of the whole kernel above the SwiGLU loop, spills the results there and reloads each behind an
  // s_waitcnt vmcnt(0) that drains the record loads in flight (round 5).  A handful of VALU operations per phase instead.
  auto lane_now = [&]() { return opaque(lane); };
  // reduction scratch [buffer][2][wave][token]: this lane's write slot (own wave) / first read slot (wave 0)
  auto red_w = [&]() { return reinterpret_cast<float*>(smem + opaque(RED_OFF + (wave * TM + (lane_now() & 31)) * 4)); };
  auto red_r = [&]() { return reinterpret_cast<const float*>(smem + opaque(RED_OFF + (lane_now() & 31) * 4)); };
  constexpr int RED0 = 0, RED1 = SCLDM_BWD_RED2 ? RED_BYTES / 4 : 0;   // float offset of the two buffers
  auto mod_base = [&]() {
    const int l = lane_now();
    return reinterpret_cast<const ModE*>(smem + opaque(MOD_OFF + (((l & 31) >> 4) * kModBlock + fb + (l >> 5) * 4) * (int)sizeof(ModE)));
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#define BWD_STAMP(i)                                                                                                        \
  do {                                                                                                                      \
    if (a.dbg && lane == 0) a.dbg[((size_t)blockIdx.x * NW + wave) * 16 + (i)] = __builtin_readcyclecounter();             \
  } while (0)
  BWD_STAMP(0);
  WStream<OP, PF, 1> ws;   // the weight ring is requested at the end of phase 2 and parked across every phase without a GEMM pass (SCLDM_BWD_PARK)
  constexpr bool kPark = SCLDM_BWD_PARK && PF <= 8;
  if (!kPark) ws.init(reinterpret_cast<const Frag*>(a.w_stream) + (size_t)wave * kBwdUnitsLayer * 64, lane);

  // record / gradient tiles in the 4-wave forward kernel's layout [tile][fwd wave = wave >> 1][quad (tt*2 + ft)*4 + q][lane][4],
  // ft = wave & 1: a quad is 1 KB (fp32) / 512 B (16-bit) contiguous per wave; the descriptors start at this wave's first quad
  // (a 32-token tile is half tt = tile & 1 of 64-token tile tile >> 1)
  const int rec_tile = NTT == 2 ? tile : tile >> 1, rec_tt0 = NTT == 2 ? 0 : tile & 1;
  const size_t quad0 = (size_t)((rec_tile * 4 + (wave >> 1)) * 16 + (rec_tt0 * 2 + (wave & 1)) * 4) * 64;   // in lane-quads of 4 elements
  const __amdgpu_buffer_rsrc_t r_xin = uniform_rsrc(a.x_in + quad0 * 4), r_dx = uniform_rsrc(a.dx + quad0 * 4);
  const __amdgpu_buffer_rsrc_t r_y1 = uniform_rsrc(a.y1 + quad0 * 4), r_y2 = uniform_rsrc(a.y2 + quad0 * 4);
  // (opaque: with provably disjoint bits hipcc turns `lane16 + q * 1024` into an OR, which it does not fold into the instruction's
  // immediate offset - one address register per access again)
  auto load_f32 = [&](const __amdgpu_buffer_rsrc_t r, float (&dst)[NTT][16]) {
    const unsigned lane16 = (unsigned)opaque(lane_now() * 16);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        union { u32x4 u; f32x4 f; } t4;
        t4.u = __builtin_amdgcn_raw_buffer_load_b128(r, lane16 + q * 1024, tt * 8192, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[tt][q * 4 + i] = t4.f[i];
      }
  };
  gchar* const p_dx = uniform_ptr(a.dx + quad0 * 4);
  auto store_dx = [&](const float (&src)[NTT][16]) {
    const unsigned lane16 = (unsigned)opaque(lane_now() * 16);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(g_f32x4*)(p_dx + (lane16 + (unsigned)(tt * 8192 + q * 1024))) = f32x4{src[tt][q * 4], src[tt][q * 4 + 1], src[tt][q * 4 + 2], src[tt][q * 4 + 3]};
  };
  // a 16-bit record array of the operand type: requested as packed pairs (16 registers in flight), widened where it is used
  auto load_16 = [&](const __amdgpu_buffer_rsrc_t r, u32x2 (&raw)[NTT][4]) {
    const unsigned lane8 = (unsigned)opaque(lane_now() * 8);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q) raw[tt][q] = __builtin_amdgcn_raw_buffer_load_b64(r, lane8 + q * 512, tt * 4096, 0);
  };
  auto widen = [&](const u32x2 (&raw)[NTT][4], int tt, int q, int i) {
    union { u32x2 u; Quad h; } t4;
    t4.u = raw[tt][q];
    return (float)t4.h[i];
  };
  // operand-pair destinations: [token][ld] 16-bit arrays; this wave writes columns first + [0, 32) of token tile tt
  static_assert(kBwdChunks * kBwdChunk == 3 * kD, "the hidden operand arrays share the dqkv row pitch");
  auto pair_dst = [&](const E* base, int ld, int tt, int first) {
    const int l = lane_now();
    return PairDst{uniform_ptr(base + (size_t)(tok0 + tt * 32) * ld + first), (unsigned)opaque(((l & 31) * ld + 8 * (l >> 5)) * 2), true};
  };
  // adaLN vector `vec` of the lane's sample, the four features of register quad q
  auto mod4 = [&](const ModE* mb, int tt, int vec, int q) { return OP::load_mod4(mb + tt * 2 * kModBlock + vec * kD + q * 8); };
  // sum over the 16 tokens of each sample of a per-(feature, token) quantity -> dmod[sample][vec][feature]
  gchar* const p_dmod = uniform_ptr(a.dmod + (size_t)smp0 * a.mod_stride + a.mod_off + fb);
  auto dmod_store = [&](const float (&v)[NTT][16], int vec) {
    const int ld = lane_now();
    const unsigned dmod_voff = (unsigned)opaque(((((ld & 31) >> 4)) * a.mod_stride + (ld >> 5) * 4) * 4);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float tot[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[r] = row16_sum(v[tt][r]);
      if ((c32 & 15) == 0 && smp0 + tt * 2 + sp < a.n) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(g_f32x4*)(p_dmod + (dmod_voff + (unsigned)(((tt * 2) * a.mod_stride + vec * kD + q * 8) * 4))) =
              f32x4{tot[q * 4], tot[q * 4 + 1], tot[q * 4 + 2], tot[q * 4 + 3]};
      }
    }
  };
  // LayerNorm statistics of every token of the tile (one sweep; fp32): in-lane sums -> half-wave exchange -> 8-way LDS combine
  auto ln_stats = [&](const float (&v)[NTT][16], float (&mean)[NTT], float (&rstd)[NTT], int RED) {
    float* const RED_W = red_w();
    const float* const RED_R = red_r();
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s += v[tt][r];
        ss = fmaf(v[tt][r], v[tt][r], ss);
      }
      s = xor32_sum(s);
      ss = xor32_sum(ss);
      if (hh == 0) {
        RED_W[RED + tt * 32] = s;
        RED_W[RED + NW * TM + tt * 32] = ss;
      }
    }
    lds_barrier();
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float m = 0.f, e2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        m += RED_R[RED + w * TM + tt * 32];
        e2 += RED_R[RED + NW * TM + w * TM + tt * 32];
      }
      mean[tt] = m * (1.0f / kD);
      rstd[tt] = __builtin_amdgcn_rsqf(fmaxf(fmaf(-mean[tt], mean[tt], e2 * (1.0f / kD)), 0.f) + a.eps);
    }
#if !SCLDM_BWD_RED2
    lds_barrier();   // RED may be rewritten by the next reduction
#endif
  };
  // y = LN(v) * S + shift (S = 1 + scale, staged) -> LDS image + operand array
  auto ln_modulate = [&](const float (&v)[NTT][16], const float (&mean)[NTT], const float (&rstd)[NTT], int sc_v, int sh_v, E* img, const E* gout) {
    const ModE* const mb = mod_base();
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const float nmr = -mean[tt] * rstd[tt];
      float y[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 sc = mod4(mb, tt, sc_v, q), sh = mod4(mb, tt, sh_v, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[q * 4 + i] = fmaf(fmaf(v[tt][q * 4 + i], rstd[tt], nmr), sc[i], sh[i]);
      }
      put_tile(y, img + (tt * 32 + c32) * XA_LD, pair_dst(gout, kD, tt, fb), fb, hh);
    }
  };
  // backward of y = LN(x) * S + shift given dh = d y (accumulator tiles): dres += d x; dmod[sc_v] = sum_t dh xhat, dmod[sh_v] = sum_t dh
  auto ln_backward = [&](const f32x16 (&dh)[NTT], const float (&x)[NTT][16], const float (&mean)[NTT], const float (&rstd)[NTT], int sc_v,
                         int sh_v, float (&dres)[NTT][16], int RED) {
    float* const RED_W = red_w();
    const float* const RED_R = red_r();
    const ModE* const mb = mod_base();
    float t1[NTT][16];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const float nmr = -mean[tt] * rstd[tt];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 sc = mod4(mb, tt, sc_v, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = q * 4 + i;
          const float xh = fmaf(x[tt][r], rstd[tt], nmr), g = dh[tt][r] * sc[i];
          s1 += g;
          s2 = fmaf(g, xh, s2);
          t1[tt][r] = dh[tt][r] * xh;
        }
      }
      s1 = xor32_sum(s1);
      s2 = xor32_sum(s2);
      if (hh == 0) {
        RED_W[RED + tt * 32] = s1;
        RED_W[RED + NW * TM + tt * 32] = s2;
      }
    }
    dmod_store(t1, sc_v);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) t1[tt][r] = dh[tt][r];
    dmod_store(t1, sh_v);
    lds_barrier();
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        s1 += RED_R[RED + w * TM + tt * 32];
        s2 += RED_R[RED + NW * TM + w * TM + tt * 32];
      }
      s1 *= (1.0f / kD);
      s2 *= (1.0f / kD);
      const float nmr = -mean[tt] * rstd[tt];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 sc = mod4(mb, tt, sc_v, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = q * 4 + i;
          const float xh = fmaf(x[tt][r], rstd[tt], nmr), g = dh[tt][r] * sc[i];
          dres[tt][r] = fmaf(rstd[tt], g - s1 - xh * s2, dres[tt][r]);
        }
      }
    }
#if !SCLDM_BWD_RED2
    lds_barrier();
#endif
  };
  auto bias_tile = [&](int p) {   // bias of c_attn rows p * 256 + fb .. + 31 as an initial accumulator tile
    const float* const BIAS_R = reinterpret_cast<const float*>(smem + opaque(BIAS_OFF + (fb + (lane_now() >> 5) * 4) * 4));
    f32x16 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(BIAS_R + p * kD + q * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) t[q * 4 + i] = b4[i];
    }
    return t;
  };
  auto to_frags = [&](const f32x16& t_acc, Frag (&F)[2]) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = t_acc[r];
    F[0] = OP::pack8(t);
    F[1] = OP::pack8(t + 8);
  };
  // eight own-sample values of a query lane -> the two k-halves (keys of sample 0 | sample 1) of an MFMA operand, cross-sample zeros
  auto masked_frags = [&](const float (&v)[8], Frag (&F)[2]) {
    union FragBits { Frag f; unsigned u[4]; } own, lo_half, hi_half;
    own.f = OP::pack8(v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lo_half.u[i] = sp ? 0u : own.u[i];
      hi_half.u[i] = sp ? own.u[i] : 0u;
    }
    F[0] = lo_half.f;
    F[1] = hi_half.f;
  };

  // ---- phase 1: every global read of the phase is requested up front (one round trip, not one per use): the adaLN vectors and
  // the c_attn bias (staged to LDS), d x_out, y2, x_in, y1 ----
  constexpr int kModQuads = NS * kModBlock / 4;               // float4s of the tile's adaLN vectors
  constexpr int kModLd = (kModQuads + NT - 1) / NT;           // 3 per thread (64-token tile), 1.5 (32-token tile: the second one on half the threads)
  f32x4 mstage[kModLd], bstage = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < kModLd; ++j) {
    const int idx = min(tid + NT * j, kModQuads - 1), sl = idx / (kModBlock / 4), w4 = idx % (kModBlock / 4);
    mstage[j] = *reinterpret_cast<const f32x4*>(a.mod + (size_t)min(smp0 + sl, a.n - 1) * a.mod_stride + a.mod_off + w4 * 4);
  }
  if (tid < 3 * kD / 4) bstage = *reinterpret_cast<const f32x4*>(a.b_qkv + tid * 4);
  // The residual x and its gradient are NOT kept in registers across the GEMM phases (the up-projection tiles, the operand
  // fragments of the attention core and the weight ring need them): each phase re-reads what it needs from the record / from
  // a.dx (L2-resident, 64 KB per tile).
  float mean2[NTT], rstd2[NTT];
  float xr[NTT][16];    // x_mid in phases 1-2, x_in from the LayerNorm-2 backward on
#if SCLDM_BWD_KEEP_DXOUT
  static_assert(SCLDM_BWD_KEEP_DX, "KEEP_DXOUT extends KEEP_DX");
  float dxr[NTT][16];
#endif
  {
  auto stage_mod = [&]() {
#pragma unroll
    for (int j = 0; j < kModLd; ++j) {
      const int idx = tid + NT * j, w4 = idx % (kModBlock / 4);
      const int vec = w4 / (kD / 4);
      f32x4 m = mstage[j];
      if (vec == 0 || vec == 3) m += 1.0f;
      if (kModQuads % NT == 0 || idx < kModQuads) OP::store_mod4(MOD + (size_t)idx * 4, m);
    }
    if (tid < 3 * kD / 4) *reinterpret_cast<f32x4*>(BIAS + tid * 4) = bstage;
  };
#if !SCLDM_BWD_KEEP_DXOUT
  float dxr[NTT][16];
#endif
  u32x2 y2raw[NTT][4], y1raw[NTT][4];
#if !(SCLDM_BWD_HOIST & 8)
  stage_mod();
#endif
  load_f32(r_dx, dxr);
  load_16(r_y2, y2raw);
#if SCLDM_BWD_HOIST & 1
  load_f32(r_xin, xr);
  load_16(r_y1, y1raw);
#endif
#if SCLDM_BWD_HOIST & 8
  stage_mod();
#endif
  lds_barrier();

  // ================= MLP branch: x_out = x_mid + a5 * c_proj(silu(w1 h2) * (w2 h2)),  h2 = LN(x_mid) (1 + a3) + a4 =================
  {
    const ModE* const mb = mod_base();
    float t[NTT][16];
    // d a5 = sum_t d x_out * y2;  d y2 = a5 * d x_out -> image R1 + operand array
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float dy[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g = mod4(mb, tt, 5, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dy[q * 4 + i] = g[i] * dxr[tt][q * 4 + i];
          t[tt][q * 4 + i] = widen(y2raw, tt, q, i) * dxr[tt][q * 4 + i];
        }
      }
      put_tile(dy, R1 + (tt * 32 + c32) * XA_LD, pair_dst(a.e_dy2, kD, tt, fb), fb, hh);
    }
    dmod_store(t, 5);
    // x_mid = x_in + a2 * y1
#if !(SCLDM_BWD_HOIST & 1)
    load_f32(r_xin, xr);
    load_16(r_y1, y1raw);
#endif
#if SCLDM_BWD_KEEP_XIN_MLP
    float xm[NTT][16];
#else
    float (&xm)[NTT][16] = xr;
#endif
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g = mod4(mb, tt, 2, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) xm[tt][q * 4 + i] = fmaf(g[i], widen(y1raw, tt, q, i), xr[tt][q * 4 + i]);
      }
  BWD_STAMP(1);
  ln_stats(xm, mean2, rstd2, RED0);
  ln_modulate(xm, mean2, rstd2, 3, 4, R0, a.e_h2);
  }
  }
  if (kPark) ws.init(reinterpret_cast<const Frag*>(a.w_stream) + (size_t)wave * kBwdUnitsLayer * 64, lane);
  lds_barrier();   // h2 and dy2 images complete
  BWD_STAMP(2);

  f32x16 dh[1][NTT];   // d h2 (this phase), later d h1
  for (int c = 0; c < kBwdChunks; ++c) {
    f32x16 ad[1][NTT], aa[1][NTT], ab[1][NTT];
    gemm_pass<OP, NTT, 1, 16, false, true, PF>(ad, ws, R1, XA_LD, lane_now());   // d hid^T = c_proj^T d y2 (this wave's 32 hidden units)
    gemm_pass<OP, NTT, 1, 16, false, true, PF>(aa, ws, R0, XA_LD, lane_now());   // a^T = w1 h2
    gemm_pass<OP, NTT, 1, 16, false, true, PF>(ab, ws, R0, XA_LD, lane_now());   // b^T = w2 h2
    // (round 5: the order a, b, [SwiGLU factors in place, hid out], d hid, [da, db out] holds one accumulator tile less but measured
    //  5 k cycles per chunk SLOWER - two store bursts and two VALU stretches between passes instead of one)
    if (c == 0) BWD_STAMP(3);
    if (c > 0) lds_barrier();   // every wave has finished the previous chunk's d h2 pass over R2
    const int first = c * kBwdChunk + fb;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float da[16], db[16], hid[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float av = aa[0][tt][r], bv = ab[0][tt][r], d = ad[0][tt][r];
        const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * av));
        const float sl = av * s;
        hid[r] = sl * bv;
        da[r] = d * bv * (s * fmaf(av, 1.0f - s, 1.0f));
        db[r] = d * sl;
      }
      put_tile(da, R2 + (tt * 32 + c32) * DADB_LD, pair_dst(a.e_da, 3 * kD, tt, first), fb, hh);
      put_tile(db, R2 + (tt * 32 + c32) * DADB_LD + kBwdChunk, pair_dst(a.e_db, 3 * kD, tt, first), fb, hh);
      put_tile(hid, nullptr, pair_dst(a.e_hid, 3 * kD, tt, first), fb, hh);
    }
    if (c == 0) BWD_STAMP(4);
    lds_barrier();
    if (c == 0) BWD_STAMP(5);
    if (c == 0) gemm_pass<OP, NTT, 1, 32, false, true, PF>(dh, ws, R2, DADB_LD, lane_now());    // d h2^T (+)= [w1^T | w2^T] [da | db]
    else if (kPark && c == kBwdChunks - 1) gemm_pass<OP, NTT, 1, 32, false, false, PF, kPark>(dh, ws, R2, DADB_LD, lane_now());
    else gemm_pass<OP, NTT, 1, 32, false, false, PF>(dh, ws, R2, DADB_LD, lane_now());
    if (c == 0) BWD_STAMP(6);
  }
  BWD_STAMP(7);
  // d x_mid = d x_out + LN2-backward(d h2);  d a3, d a4
  float mean1[NTT], rstd1[NTT];
#if SCLDM_BWD_KEEP_DX && !SCLDM_BWD_KEEP_DXOUT
  float dxr[NTT][16];
#endif
  {
  // every global read of the two phases is requested here: d x_out (needed at the end of the LayerNorm-2 backward), y1 and x_in.
  // x_mid = x_in + a2 y1 is REBUILT here instead of living in 32 registers across the SwiGLU chunks (round 5: that loop is the
  // kernel's register peak, and x_in / y1 are read again for this phase anyway)
#if !SCLDM_BWD_KEEP_DX
  float dxr[NTT][16];
#endif
  u32x2 y1raw[NTT][4];
#if !SCLDM_BWD_KEEP_DXOUT
  load_f32(r_dx, dxr);
#endif
  load_16(r_y1, y1raw);
#if !SCLDM_BWD_KEEP_XIN_MLP
  load_f32(r_xin, xr);
#endif
  {
    const ModE* const mb = mod_base();
    float xm[NTT][16];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g = mod4(mb, tt, 2, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) xm[tt][q * 4 + i] = fmaf(g[i], widen(y1raw, tt, q, i), xr[tt][q * 4 + i]);
      }
    ln_backward(dh[0], xm, mean2, rstd2, 3, 4, dxr, RED1);
  }

  BWD_STAMP(8);
  // ================= attention branch: x_mid = x_in + a2 * (c_proj(attention(c_attn(h1))) + b),  h1 = LN(x_in) (1 + a0) + a1 =================
  {
    const ModE* const mb = mod_base();
    float t[NTT][16];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float dy[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g = mod4(mb, tt, 2, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dy[q * 4 + i] = g[i] * dxr[tt][q * 4 + i];
          t[tt][q * 4 + i] = widen(y1raw, tt, q, i) * dxr[tt][q * 4 + i];
        }
      }
      put_tile(dy, R1 + (tt * 32 + c32) * XA_LD, pair_dst(a.e_dy1, kD, tt, fb), fb, hh);   // d y1 (R1's last reader was chunk 2's d hid pass)
    }
    dmod_store(t, 2);
  }
#if !SCLDM_BWD_KEEP_DX
  // d x_mid waits in a.dx for the LayerNorm-1 backward at the end
  store_dx(dxr);
#endif
  ln_stats(xr, mean1, rstd1, RED0);
  ln_modulate(xr, mean1, rstd1, 0, 1, R0, a.e_h1);   // h1 (R0's last readers were chunk 2's a / b passes)
  }
  if (kPark) ws.unpark();
  lds_barrier();

  BWD_STAMP(9);
  // q, k, v, d ao of head `wave`: as (token, k = head dim) operands straight from the accumulator tiles, and transposed
  // ((head dim, k = token), through the wave's LDS scratch - R2's tail is free since the last d h2 pass)
  // (round 5: the transposed copies are made in the core, one tile at a time, from the packed fragments - holding all eight next
  //  to the four passes' accumulators and the ring was the kernel's register peak)
  Frag QF[NTT][2], KF[NTT][2], VF[NTT][2], GF[NTT][2];
  {
    f32x16 acc[1][NTT];
    auto keep = [&](Frag (&F)[NTT][2]) {
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) to_frags(acc[0][tt], F[tt]);
    };
    gemm_pass<OP, NTT, 1, 16, false, true, PF>(acc, ws, R1, XA_LD, lane_now());   // d ao^T = c_proj^T d y1
    keep(GF);
    f32x16 b = bias_tile(0);
    gemm_pass<OP, NTT, 1, 16, false, true, PF>(acc, ws, R0, XA_LD, lane_now(), &b);
    keep(QF);
    b = bias_tile(1);
    gemm_pass<OP, NTT, 1, 16, false, true, PF>(acc, ws, R0, XA_LD, lane_now(), &b);
    keep(KF);
    b = bias_tile(2);
    gemm_pass<OP, NTT, 1, 16, false, true, PF, kPark>(acc, ws, R0, XA_LD, lane_now(), &b);
    keep(VF);
  }
  const int l_tr = lane_now();
  E* const TRW = reinterpret_cast<E*>(smem + opaque(TR_OFF + wave * TR_BYTES + (4 * (l_tr >> 5) * TR_LD + (l_tr & 31)) * 2));   // transpose scratch, write side
  E* const TRR = reinterpret_cast<E*>(smem + opaque(TR_OFF + wave * TR_BYTES + ((l_tr & 31) * TR_LD + 4 * (l_tr >> 5)) * 2));   // read side
  auto transposed = [&](const Frag (&F)[2], Frag (&T)[2]) {   // (head dim, k = token) copy of a (token, k = head dim) tile
    tr_write_frags(TRW, F);
    wave_sync();
    T[0] = tr_read(TRR, 0);
    T[1] = tr_read(TRR, 1);
    wave_sync();
  };
  BWD_STAMP(10);
  lds_barrier();   // every wave is done with the h1 / dy1 images: the dqkv image and the transpose scratch may overwrite them

#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    // S^T[key][query] = K Q^T, softmax over the query's own sample
    f32x16 st = OP::mma(KF[tt][0], QF[tt][0], zero16);
    st = OP::mma(KF[tt][1], QF[tt][1], st);
    float p[8];
    {
      float sv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float lo = st[i], hi = st[8 + i];
        asm volatile("" : "+v"(lo), "+v"(hi));
        sv[i] = sp ? hi : lo;
      }
      float m = sv[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) m = fmaxf(m, sv[i]);
      m = xor32_max(m);
      float sum = 0.f;
      const float nms = -m * a.attn_scale_log2e;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], a.attn_scale_log2e, nms));
        sum += sv[i];
      }
      sum = xor32_sum(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = sv[i] * inv;
    }
    Frag Ph[2];
    masked_frags(p, Ph);
    // O^T[d][query] = V^T P^T (operand for the c_proj weight gradient)
    {
      Frag VT[2];
      transposed(VF[tt], VT);
      f32x16 ot = OP::mma(VT[0], Ph[0], zero16);
      ot = OP::mma(VT[1], Ph[1], ot);
      float o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = ot[r];
      put_tile(o, nullptr, pair_dst(a.e_ao, kD, tt, fb), fb, hh);
    }
    // dP^T[key][query] = V dAO^T;  dS = P (dP - sum_key P dP) / sqrt(d)
    f32x16 dpt = OP::mma(VF[tt][0], GF[tt][0], zero16);
    dpt = OP::mma(VF[tt][1], GF[tt][1], dpt);
    float ds[8];
    {
      float dp[8], dot = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float lo = dpt[i], hi = dpt[8 + i];
        asm volatile("" : "+v"(lo), "+v"(hi));
        dp[i] = sp ? hi : lo;
        dot = fmaf(p[i], dp[i], dot);
      }
      dot = xor32_sum(dot);
#pragma unroll
      for (int i = 0; i < 8; ++i) ds[i] = p[i] * (dp[i] - dot) * a.attn_scale;
    }
    Frag dSh[2];
    masked_frags(ds, dSh);
    float o[16];
    E* lrow = DQKV + (tt * 32 + c32) * DQKV_LD;
    // dQ^T[d][query] = K^T dS^T
    {
      Frag KT[2];
      transposed(KF[tt], KT);
      f32x16 dq = OP::mma(KT[0], dSh[0], zero16);
      dq = OP::mma(KT[1], dSh[1], dq);
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = dq[r];
      put_tile(o, lrow, pair_dst(a.e_dqkv, 3 * kD, tt, fb), fb, hh);
    }
    // dK^T[d][key] = Q^T dS
    Frag T1[2];
    tr_write_own(TRW, ds, sp);
    wave_sync();
    T1[0] = tr_read(TRR, 0);
    T1[1] = tr_read(TRR, 1);
    wave_sync();
    {
      Frag QT[2];
      transposed(QF[tt], QT);
      f32x16 dk = OP::mma(QT[0], T1[0], zero16);
      dk = OP::mma(QT[1], T1[1], dk);
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = dk[r];
      put_tile(o, lrow, pair_dst(a.e_dqkv, 3 * kD, tt, kD + fb), kD + fb, hh);
    }
    // dV^T[d][key] = dAO^T P
    tr_write_own(TRW, p, sp);
    wave_sync();
    T1[0] = tr_read(TRR, 0);
    T1[1] = tr_read(TRR, 1);
    wave_sync();
    {
      Frag GT[2];
      transposed(GF[tt], GT);
      f32x16 dv = OP::mma(GT[0], T1[0], zero16);
      dv = OP::mma(GT[1], T1[1], dv);
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = dv[r];
      put_tile(o, lrow, pair_dst(a.e_dqkv, 3 * kD, tt, 2 * kD + fb), 2 * kD + fb, hh);
    }
  }
  BWD_STAMP(11);
  if (kPark) ws.unpark();
  lds_barrier();   // dqkv image complete
  // x_in (and d x_mid) are requested before the K = 768 pass and arrive under it
#if !SCLDM_BWD_KEEP_DX
  float dxr[NTT][16];
#endif
#if (SCLDM_BWD_HOIST & 4) && !SCLDM_BWD_KEEP_XIN
  load_f32(r_xin, xr);
#endif
#if SCLDM_BWD_HOIST & 4
#if !SCLDM_BWD_KEEP_DX
  load_f32(r_dx, dxr);
#endif
#endif
  gemm_pass<OP, NTT, 1, 48, false, true, PF, kPark>(dh, ws, DQKV, DQKV_LD, lane_now());   // d h1^T = c_attn^T d qkv (last pass: the ring ends here)
  BWD_STAMP(12);
#if !(SCLDM_BWD_HOIST & 4) && !SCLDM_BWD_KEEP_XIN
  load_f32(r_xin, xr);
#endif
#if !(SCLDM_BWD_HOIST & 4)
#if !SCLDM_BWD_KEEP_DX
  load_f32(r_dx, dxr);
#endif
#endif
  // d x_in = d x_mid + LN1-backward(d h1);  d a0, d a1
  ln_backward(dh[0], xr, mean1, rstd1, 0, 1, dxr, RED1);

  store_dx(dxr);
  BWD_STAMP(13);
#undef BWD_STAMP
}

}  // namespace bwd
}  // namespace scldm
