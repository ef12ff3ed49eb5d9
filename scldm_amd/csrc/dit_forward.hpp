// Fused adaLN-Zero DiT layer for gfx950: one launch = one transformer block over all sample-forwards
// (the first launch also does the input projection, the last one the final layer).  Inside a layer the
// fp32 residual lives in registers: it is read once and written once per layer.
//
// Why one launch per layer and not one persistent launch for the whole network: every workgroup streams
// the layer's full 1.7 MB of weights; with all CUs of an XCD on the SAME layer that stream is served by the
// 4 MB L2.  A whole-network persistent variant was measured (round 1): workgroups drift across layers, the
// 13.6 MB weight set falls out of L2 and the GEMM phases slow down 1.8x.  Kernel boundaries are the cheapest
// lockstep on this chip.
//
// Replaces the reference's (eager, ~230 launches per forward)
//   DiT.forward trunk                     src/scldm/nnets.py:290-296
//   Block.forward adaLN branch            src/scldm/layers.py:213-221  (+ modulate :91-94, F7 order)
//   SelfAttention.forward                 src/scldm/layers.py:143-158
//   MLP.forward (SwiGLU)                  src/scldm/layers.py:173-174
//   FinalLayerDit.forward                 src/scldm/layers.py:397-401
// Specialised to the reference's only DiT shape family: n_embed 256, 8 heads x 32, seq_len 16
// (experiments/configs/model/ldm_base.yaml:16-25); hidden (684) is zero-padded to a multiple of 128.
//
// Work decomposition (MI355X-first, not a GEMM-library composition):
//   * one workgroup = NW = 8/FT waves owns a tile of TM = 32*NTT tokens (= 2*NTT samples); wave w owns the FT
//     32-row feature tiles [32*FT*w, 32*FT*(w+1)) (FT=2: 4 waves x 2 heads; FT=1: 8 waves x 1 head); the
//     residual x of the tile lives in VGPRs in MFMA accumulator layout for the whole layer (the first version
//     re-read/re-wrote it five times per layer and spent 37 % of its time on that, in phase-locked bursts);
//   * GEMMs are computed TRANSPOSED, Y^T[feature][token] = W[feature][k] * X^T[k][token]:
//       A operand = weights, streamed L2 -> VGPR in a pre-packed per-wave fragment stream that runs
//                   continuously through the layer (each weight byte is read by exactly one wave of the
//                   workgroup -> no LDS staging; a register ring prefetches across phase boundaries),
//       B operand = activations, shared by the four waves through LDS ([token][feature], +16 B row pad
//                   => conflict-free ds_read_b128);
//     every wave sees ALL tokens of the tile for its features, so LayerNorm statistics are an in-lane sum +
//     one half-wave exchange + an NW-way LDS combine;
//   * attention never leaves registers: Q^T and K^T tiles come out of the MFMA in a layout that is
//     directly a valid A/B operand pair for S^T = K Q^T (any k permutation is legal if both sides
//     share it); V is produced with swapped operands (V[token][d]) so that O^T = V^T P^T likewise
//     needs no transpose; softmax over the 16 keys is 8 in-lane values + one xor-32 exchange.
//     Two samples share each 32x32 MFMA tile; cross-sample score blocks are masked to exact zeros;
//   * the SwiGLU hidden dimension is processed in chunks of 128 through a double-buffered LDS
//     stage (w1/w2 rows are interleaved inside each 32-row weight tile so silu(a)*b is in-lane).
#pragma once
#include <type_traits>

#include "common.hpp"

// Issue priority of a wave while it runs a weight-streaming GEMM pass (s_setprio; 0 elsewhere).  Two workgroups share
// each SIMD: letting the wave that is feeding the matrix pipe win instruction issue over its neighbour's LayerNorm /
// softmax / SiLU VALU work measured -2 % kernel time (389 -> 381 us, same session); priority held across the whole layer
// except LayerNorm measured +1 % instead.  -DSCLDM_SETPRIO=0 disables it.
#ifndef SCLDM_SETPRIO
#define SCLDM_SETPRIO 1
#endif
// k-steps of run-ahead of the activation-fragment LDS reads in the one-tile up-projection pass (1 or 2)
#ifndef SCLDM_BPF
#define SCLDM_BPF 1
#endif
// 1: Linear biases enter as the INITIAL accumulator of their GEMM pass (the MFMA's C operand) instead of a VALU add per output
// element afterwards; 2 fewer VALU ops per LayerNorm element (scale and offset folded per token).  A/B switch.
#ifndef SCLDM_LEAN_VALU
#define SCLDM_LEAN_VALU 1
#endif
// 1: activation tiles are stored to LDS as 16-byte pieces after a half-wave exchange (common.hpp: halfwave_pair) instead of
// 8-byte pieces (2-way bank conflicted on 16-byte-aligned rows).  A/B switch.
#ifndef SCLDM_PAIR_STORE
#define SCLDM_PAIR_STORE 1
#endif

// Floating-point contraction is OFF in this file and every fused multiply-add is written out (fmaf): the layer body is
// instantiated once per layer slot of a launch, and with the default contract(fast) hipcc is free to fuse a * b + c in one
// slot's schedule and not in another's - results then depend on which slot (i.e. on SCLDM_LPL) a layer happens to run in.
#pragma clang fp contract(off)

namespace scldm {

constexpr int kD = 256;        // n_embed
constexpr int kHC = 128;       // hidden chunk per workgroup (4 waves x 32)
constexpr int kModBlock = 6 * kD;
constexpr int kDbgStamps = 32;
// layer slots instantiated in the fused kernel (code size grows with it: ~30 KB of ISA per slot).  Round 4: EIGHT for the inference
// kernels - the whole reference network in one launch, no residual hand-off through HBM at all (same-box interleaved A/B,
// profiles/r4a_ab_lpl8_l2warm.txt: +0.8 % at 4 096 cells, +2.0 % at 1 024, +1.9 % at 512 over four); the recording (training)
// instantiation keeps four (-DSCLDM_MAX_LPL=4 restores four everywhere).
#ifndef SCLDM_MAX_LPL
#define SCLDM_MAX_LPL 8
#endif
constexpr int kMaxLayersPerLaunch = SCLDM_MAX_LPL;
#ifndef SCLDM_REC_LPL
#define SCLDM_REC_LPL 4
#endif
constexpr int kMaxLayersPerLaunchRec = SCLDM_REC_LPL;   // layer slots of the RECORDING (training) instantiation: 4, or 8 with -DSCLDM_REC_LPL=8
// 1: every workgroup touches its share of the NEXT layer's weight stream (one 4-byte load per 128-byte line) at the start of a
// layer, so that the stream's first-touch misses (each XCD's 4 MB L2 holds ~2 layers) are taken a layer ahead of the ring.
#ifndef SCLDM_L2WARM
#define SCLDM_L2WARM 0
#endif
// Timing proxies (deliberately WRONG results; libx_* experiment builds only): bit 0 no weight-ring refills, bit 1 no activation-fragment
// LDS reads, bit 2 no SwiGLU transcendental work, bit 3 (round 5) the cost model of "two wave groups share one weight ring through LDS":
// only every SECOND k-step's weight fragments are fetched from L2 (the other group's half) and every k-step reads FT extra 16-byte
// fragments per lane from LDS (where the shared ring would live), bit 4 the LDS half of that alone (extra reads, full L2 stream).  What each costs in time AND clock under the power budget.
#ifndef SCLDM_PROXY
#define SCLDM_PROXY 0
#endif
// 1: the SwiGLU up-projection of a full chunk is ONE two-tile pass (each activation fragment read from LDS feeds two MFMAs instead of
// one: -31 % LDS fragment reads per layer; the packer then orders a chunk's W12 units k-step by k-step).  Round 4 proxy: the fragment
// reads cost 2 % in cycles but 7 % under the power budget (profiles/r4j_timing_proxies_random_vs_zero.txt).  With round 4's kernel the
// two-tile pass spills 178 VGPRs (64 accumulators more are live next to the residual and the x-row registers), so it stays off.
#ifndef SCLDM_W12_PAIR
#define SCLDM_W12_PAIR 0
#endif

// Phase stamps (s_memtime) for the debug build (first layer only); compiles to nothing otherwise.
#ifdef SCLDM_PHASE_TIMING
#define SCLDM_STAMP(i)                                                                                 \
  do {                                                                                                 \
    if (a.dbg && lane == 0)                                                                            \
      a.dbg[((size_t)blockIdx.x * 8 + wave) * kDbgStamps + (i)] = __builtin_readcyclecounter();       \
  } while (0)
#define SCLDM_STAMP_END(i)                                                                             \
  do {                                                                                                 \
    if (a.dbg && lane == 0) a.dbg[((size_t)blockIdx.x * 8 + wave) * kDbgStamps + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define SCLDM_STAMP(i) do {} while (0)
#define SCLDM_STAMP_END(i) do {} while (0)
#endif

struct FwdArgs {
  const float* z;           // latents (n_src, 16, din) fp32 (read by the first layer)
  float* out;               // (n_fwd, 16, din) fp32 (written by the last layer)
  float* x;                 // fp32 residual between layers, private layout [tile][wave][reg-quad 8*NTT][lane][4]
                            // (what each lane holds in registers, so every access is one coalesced 1 KiB)
  const float* mod;         // (rows, mod_stride) adaLN vectors: [layer][6][256] ... [final: shift, scale]
  const int32_t* row_index; // (n_fwd) conditioning row of each sample-forward
  const void* w_stream;     // packed weights of THIS layer, [wave][unit] (pack_layer_kernel)
  const void* w_final;      // final_layer.linear packed as 16 fragments (rows >= din are zero)
  const float* b_qkv;       // (768) of this layer
  const float* b_proj;      // (256) of this layer
  const float* in_wt;       // input_proj weight transposed (din, 256) (fp32 path)
  const float* in_w;        // input_proj weight as stored (256, din) (bf16 path: MFMA operand rows)
  const float* in_b;        // (256)
  const float* pos;         // (16, 256)
  const float* fin_b;       // (din)
  int n_fwd;                // number of sample-forwards (16 tokens each)
  int n_direct, rep;        // sample-forward s reads latent s (s < n_direct) else n_direct - rep + (s - n_direct) % rep
  int din;                  // latent channels (<= 32 handled by one output tile)
  int layer, n_layer;       // first layer of this launch, layers of the network
  int n_here;               // consecutive layers this launch runs (1..kMaxLayersPerLaunch; the residual stays in registers between them)
  long w_layer_elems;       // elements of the packed stream per layer (w_stream, b_qkv, b_proj point at `layer`)
  int tile0;                // first token tile of this launch (a layer may be launched as several tile groups)
  int grid_tiles;           // tiles of this launch (0: all)
  int n_chunks;             // full SwiGLU chunks of 128 hidden units
  int half_chunk;           // 1: a trailing chunk of 64 hidden units follows (FT=2 kernels; 684 -> 5*128 + 64 = 704)
  int mod_stride;
  float eps;
  float attn_scale_log2e;   // log2(e) / sqrt(head_dim)
  // Training record (REC kernels only; dit_backward.hpp reads it): per layer the residual entering it (rec_x[layer], fp32;
  // rec_x[n_layer] = the final layer's input) and the two gated branch outputs y1 = c_proj(attention) + b, y2 = MLP (bf16),
  // all in the hand-off buffer's private tile layout.  rec_stride = elements per layer (= padded tokens * 256).
  float* rec_x;
  void* rec_y1;             // (16-bit elements of the kernel's operand type: bf16, or fp16 for the OpFP16 instantiation)
  void* rec_y2;
  long rec_stride;
  unsigned long long* dbg;  // phase stamps [block][wave][kDbgStamps]; only -DSCLDM_PHASE_TIMING builds write
};

template <typename OP, int NTT, int FT>
struct FwdLayout {
  using E = typename OP::E;
  static constexpr int NW = 8 / FT;             // waves per workgroup (256 features / (32*FT) per wave)
  static constexpr int NT = 64 * NW;            // threads
  static constexpr int TM = 32 * NTT;
  static constexpr int NS = 2 * NTT;            // samples per tile
  static constexpr int PADE = 16 / sizeof(E);
  static constexpr int XA_LD = kD + PADE;       // elements per activation row
  static constexpr int HB_LD = kHC + PADE;      // elements per hidden-chunk row
  static constexpr int XA_BYTES = TM * XA_LD * sizeof(E);   // LN output; later the attention output (AO aliases it)
  static constexpr int HB_BYTES = TM * HB_LD * sizeof(E);   // one SwiGLU hidden chunk
  // Two hidden-chunk buffers (ping-pong): chunk c writes buffer c & 1, which every wave finished reading before it
  // arrived at chunk c-1's "ready" barrier - ONE barrier per chunk instead of two (-3.7 % kernel time).  Before the
  // SwiGLU phase buffer 0 holds the staged c_attn / c_proj biases and buffer 1 the LayerNorm reduction scratch.
  static constexpr int HB_BUFS = NTT <= 2 ? 2 : 1;
  static constexpr int RED_BYTES = 2 * NW * TM * sizeof(float);
  static constexpr int RED_OFF = XA_BYTES + (HB_BUFS == 2 ? HB_BYTES : HB_BUFS * HB_BYTES);   // aliases buffer 1, or its own region
  static constexpr int MOD_OFF = XA_BYTES + HB_BUFS * HB_BYTES + (HB_BUFS == 2 ? 0 : RED_BYTES);
  static constexpr int MOD_BYTES = NS * kModBlock * sizeof(typename OP::ModE);  // the tile's six adaLN vectors per sample
  static constexpr int LDS_BYTES = MOD_OFF + MOD_BYTES;
  static_assert(RED_BYTES <= HB_BYTES && 4 * kD * (int)sizeof(float) <= HB_BYTES, "aliases must fit a hidden-chunk buffer");
};

// ---------------------------------------------------------------------------------------------
// Weight stream.  Every wave consumes ONE contiguous sequence of "units" for the whole network
// (unit = one k-step of 16 for the wave's FT 32-row weight tiles = FT fragments of 1 KiB bf16) per layer:
//     Q (16 units) | K (16) | V (16) | proj (16) | for each hidden chunk: W12 (16) | c_proj (8)
// With FT=2 the W12 units of a chunk are ordered tile 0 (8 units of two consecutive k-steps) then tile 1, so that the
// up-projection can be computed one 32-row tile at a time (half the accumulator registers).
// A PF-deep register ring runs ahead of the MFMAs and persists across passes, so L2 latency is
// hidden across phase boundaries too and nothing is fetched twice.  The ring over-reads PF units
// past the wave's last unit (next wave's stream / allocation slack), which is never consumed.
// ---------------------------------------------------------------------------------------------
constexpr int kUnitsFixed = 64;      // Q,K,V,proj
constexpr int kUnitsPerChunk = 24;   // W12 (16) + c_proj (8)
constexpr int kMaxPF = 8;
constexpr int kUnitsHalfChunk = 12;  // FT=2 only: one 32-row W12 tile (8 units) + a K=64 c_proj pass (4 units)
__host__ __device__ constexpr int units_per_layer(int n_chunks, int half = 0) {
  return kUnitsFixed + n_chunks * kUnitsPerChunk + half * kUnitsHalfChunk;
}

template <typename OP, int NTT, int FT>
struct Prefetch {  // k-steps of run-ahead of the weight ring
#ifdef SCLDM_PF
  static constexpr int PF = OP::kIsBF16 ? SCLDM_PF : OP::kRing;
#else
  // 32-token tiles (launches below one 64-token tile per CU: a single workgroup's walk through the layers IS the launch time) run the
  // 16-bit policies' ring 8 k-steps ahead instead of 4: with one workgroup per CU nothing else hides the L2 latency of the weight
  // stream, and the instantiation has the registers (170 -> 216 VGPRs, no spills).  Same box, interleaved, 128 cells x 50 evaluations:
  // 165.0 -> 150.6 us per launch, 9.24 -> 8.51 ms per trajectory (round 6).  64-token tiles keep 4 (247 VGPRs: 8 would spill; +-0 at
  // full occupancy, HISTORY).  Results do not depend on the depth.
  static constexpr int PF = (NTT == 1 && OP::kIsBF16) ? 8 : OP::kRing;
#endif
};

template <typename OP, int PF, int FT>
struct WStream {
  using Frag = typename OP::Frag;
  // The stream is read through a buffer descriptor (SRSRC in SGPRs): address = wave-uniform base + scalar running offset
  // (advanced on the scalar unit) + this lane's constant byte offset + an immediate - no per-k-step 64-bit VALU pointer
  // arithmetic (128 v_lshl_add_u64 per layer with a per-lane pointer), and no address VGPR pair.
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned soff;       // running byte offset of the next unit (wave-uniform)
  unsigned lane_off;   // lane * sizeof(Frag)
  Frag ring[PF][FT];
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  __device__ __forceinline__ Frag fetch(int ft) const {
    static_assert(sizeof(Frag) == 16 || sizeof(Frag) == 32, "fragment = one or two 16-byte loads per lane");
    union { Frag f; u32x4 q[sizeof(Frag) / 16]; } u;
#pragma unroll
    for (int h = 0; h < (int)(sizeof(Frag) / 16); ++h)
      u.q[h] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + ft * 64 * (int)sizeof(Frag) + h * 16, soff, 0);
    return u.f;
  }
  __device__ __forceinline__ void advance(int frags) { soff += frags * 64 * (unsigned)sizeof(Frag); }
  // after a PARK pass: request the PF units the skipped refills would have fetched
  __device__ __forceinline__ void unpark() {
    soff -= PF * FT * 64 * (unsigned)sizeof(Frag);
#pragma unroll
    for (int s = 0; s < PF; ++s) {
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) ring[s][ft] = fetch(ft);
      advance(FT);
    }
  }
  __device__ __forceinline__ void init(const Frag* unit0, int lane) {
    // the descriptor inputs are made provably wave-uniform (a lane-tainted pointer makes hipcc wrap every load in a waterfall loop)
    const unsigned long long b = reinterpret_cast<unsigned long long>(unit0);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
    soff = 0;
    lane_off = (unsigned)lane * (unsigned)sizeof(Frag);
#pragma unroll
    for (int s = 0; s < PF; ++s) {
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) ring[s][ft] = fetch(ft);
      advance(FT);
    }
  }
};

// One GEMM pass: acc[ft][tt] (+)= W-tile(ft) * B-tile(tt) over KSTEPS*16 k-values; activations from LDS.
// ZERO=true starts from zero accumulators (first k-step uses an inline-zero C operand: no register clearing).
// SWAP=true computes the transposed tile (token rows, feature cols) - used for V.
//
// Schedule per k-step, pinned with sched_group_barrier (hipcc otherwise sinks every refill load to the end of
// the unrolled body and waits for it two MFMAs later, and issues each ds_read right in front of its consumer):
//     NTT ds_read_b128 (B fragments of the NEXT k-step)  |  FT*NTT MFMAs (this k-step)  |  FT global loads
// (refill of the ring slot just consumed = PF k-steps ahead).
// PARK=true: the pass's last ring revolution is not refilled (the offset still advances): the ring registers are free until
// WStream::unpark() re-requests those PF units (dit_backward.hpp parks the ring across the register-hungry attention core).
template <typename OP, int NTT, int FT, int KSTEPS, bool SWAP, bool ZERO, int PF, bool PARK = false>
__device__ __forceinline__ void gemm_pass(f32x16 (&acc)[FT][NTT], WStream<OP, PF, FT>& ws,
                                          const typename OP::E* __restrict__ bsm, int ldb, int lane, const f32x16* init = nullptr) {
  // ZERO with init != nullptr: tile (ft, *) starts from init[ft] (a per-row bias tile) instead of zero
  using Frag = typename OP::Frag;
  // a pass shorter than the ring (the trailing half chunk's down-projection) is legal only as the LAST pass of the stream
  static_assert(KSTEPS % PF == 0 || KSTEPS < PF, "KSTEPS must be a multiple of the prefetch depth (or a final short pass)");
  const int c32 = lane & 31, hh = lane >> 5;
  const typename OP::E* bbase = bsm + c32 * ldb + hh * 8;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  Frag bcur[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) bcur[tt] = *reinterpret_cast<const Frag*>(bbase + tt * 32 * ldb);
  auto step = [&](int ks, int s, bool first, bool refill = true) {
    Frag bnext[NTT];
    // B fragments of k-step ks+1 (after the last k-step this reads the row pad / next row: valid LDS, never used)
#if SCLDM_PROXY & 2    // timing proxy 2 (WRONG RESULTS): the activation fragments are read once per pass - what the LDS fragment reads cost
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) bnext[tt] = bcur[tt];
#else
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) bnext[tt] = *reinterpret_cast<const Frag*>(bbase + tt * 32 * ldb + (ks + 1) * 16);
#endif
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        if (SWAP) acc[ft][tt] = OP::mma(bcur[tt], ws.ring[s][ft], first ? (init ? init[ft] : zero) : acc[ft][tt]);
        else acc[ft][tt] = OP::mma(ws.ring[s][ft], bcur[tt], first ? (init ? init[ft] : zero) : acc[ft][tt]);
      }
    }
#if SCLDM_PROXY & 24   // timing proxies 8 / 16 (WRONG RESULTS): FT more fragment reads from LDS per k-step (a weight ring shared through LDS)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      typedef __attribute__((ext_vector_type(4))) unsigned px_u32x4;
      const px_u32x4 extra = *reinterpret_cast<const px_u32x4*>(bbase + ((ft + 1) & (NTT - 1)) * 32 * ldb + ((ks + 2 + ft) & 7) * 16);
      asm volatile("" ::"v"(extra));
    }
#endif
#if !(SCLDM_PROXY & 1)   // timing proxy 1 (WRONG RESULTS): the weight ring is never refilled - what the L2 -> VGPR weight stream costs
    if (refill && (!(SCLDM_PROXY & 8) || (ks & 1))) {
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) ws.ring[s][ft] = ws.fetch(ft);  // refill the slot just consumed: PF k-steps ahead
    }
#endif
    ws.advance(FT);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) bcur[tt] = bnext[tt];
    if (OP::kPin) {
      __builtin_amdgcn_sched_group_barrier(0x100, NTT * OP::kFragLoads, 0);       // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, FT * NTT * OP::kMmaOps, 0);     // MFMA
      if (refill) __builtin_amdgcn_sched_group_barrier(0x020, FT * OP::kFragLoads, 0);        // VMEM read
    }
  };
  // peeled first ring revolution (so that ZERO needs no accumulator clearing), then the rolled loop
#if SCLDM_SETPRIO
  __builtin_amdgcn_s_setprio(SCLDM_SETPRIO);
#endif
  static_assert(!PARK || (KSTEPS % PF == 0 && KSTEPS >= 2 * PF), "a parking pass has a whole last revolution of its own");
#pragma unroll
  for (int s = 0; s < (PF < KSTEPS ? PF : KSTEPS); ++s) step(s, s, ZERO && s == 0);
#pragma unroll 1
  for (int ks0 = PF; ks0 < KSTEPS - (PARK ? PF : 0); ks0 += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) step(ks0 + s, s, false);
  }
  if (PARK) {
#pragma unroll
    for (int s = 0; s < PF; ++s) step(KSTEPS - PF + s, s, false, false);
  }
#if SCLDM_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// Up-projection pass for ONE 32-row weight tile over K = 256 (FT=2 only): the ring still moves units of two
// fragments, here the two consecutive k-steps (2j, 2j+1) of the same tile.  Halves the accumulator footprint of the
// SwiGLU phase (the residual + down-projection accumulators are live there); B fragments are read once per tile.
template <typename OP, int NTT, int PF>
__device__ __forceinline__ void gemm_pass_tile(f32x16 (&acc)[NTT], WStream<OP, PF, 2>& ws,
                                               const typename OP::E* __restrict__ bsm, int ldb, int lane) {
  using Frag = typename OP::Frag;
  constexpr int UNITS = 8;
  static_assert(UNITS % PF == 0, "8 units must be a multiple of the prefetch depth");
  const int c32 = lane & 31, hh = lane >> 5;
  const typename OP::E* bbase = bsm + c32 * ldb + hh * 8;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  Frag bcur[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) bcur[tt] = *reinterpret_cast<const Frag*>(bbase + tt * 32 * ldb);
#if SCLDM_BPF >= 2
  // activation fragments two k-steps ahead: one k-step of this pass is only NTT MFMAs (64 cycles at NTT = 2), less than an
  // LDS round trip with eight waves reading
  Frag bmid[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) bmid[tt] = *reinterpret_cast<const Frag*>(bbase + tt * 32 * ldb + 16);
#endif
  auto unit = [&](int u, int s, bool first) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int ks = 2 * u + half;
      Frag bnext[NTT];
#if SCLDM_PROXY & 2
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) bnext[tt] = bcur[tt];
#elif SCLDM_BPF >= 2
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) bnext[tt] = *reinterpret_cast<const Frag*>(bbase + tt * 32 * ldb + (ks + 2) * 16);
#else
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) bnext[tt] = *reinterpret_cast<const Frag*>(bbase + tt * 32 * ldb + (ks + 1) * 16);
#endif
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) acc[tt] = OP::mma(ws.ring[s][half], bcur[tt], (first && half == 0) ? zero : acc[tt]);
#if SCLDM_BPF >= 2
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) { bcur[tt] = bmid[tt]; bmid[tt] = bnext[tt]; }
#else
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) bcur[tt] = bnext[tt];
#endif
      if (half == 1) {
#if !(SCLDM_PROXY & 1)
        ws.ring[s][0] = ws.fetch(0);
        ws.ring[s][1] = ws.fetch(1);
#endif
        ws.advance(2);
      }
      if (OP::kPin) {
        __builtin_amdgcn_sched_group_barrier(0x100, NTT * OP::kFragLoads, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NTT * OP::kMmaOps, 0);
        if (half == 1) __builtin_amdgcn_sched_group_barrier(0x020, 2 * OP::kFragLoads, 0);
      }
    }
  };
#if SCLDM_SETPRIO
  __builtin_amdgcn_s_setprio(SCLDM_SETPRIO);
#endif
#pragma unroll
  for (int s = 0; s < PF; ++s) unit(s, s, s == 0);
#pragma unroll 1
  for (int u0 = PF; u0 < UNITS; u0 += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) unit(u0 + s, s, false);
  }
#if SCLDM_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

#ifdef SCLDM_PHASE_TIMING
#define SCLDM_LN_STAMP(i)                                                                              \
  do {                                                                                                 \
    if (SB >= 0 && dbg && lane == 0) dbg[((size_t)blockIdx.x * 8 + wave) * kDbgStamps + SB + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define SCLDM_LN_STAMP(i) do {} while (0)
#endif

// LayerNorm (no affine, biased variance) over the 256 features of every token of the tile, followed by
// y*(1+scale)+shift, written as OP::E into dst[token][feature].  v holds this wave's 32*FT features x TM tokens in
// accumulator layout; scale/shift come from the LDS copy of the tile's adaLN vectors (msm[sample][6*256], vector
// indices sc_v / sh_v).  Statistics: in-lane sums -> one permlane32 exchange -> NW-way combine through LDS.
// `between` runs before the last statistics barrier (used to publish freshly staged LDS data).
template <typename OP, int NTT, int FT, int SB = -1, typename Between>
__device__ __forceinline__ void ln_modulate_store(const float (&v)[FT][NTT][16], const typename OP::ModE* msm, int sc_v, int sh_v,
                                                  float eps, float* red, typename OP::E* dst, int ldd, int wave,
                                                  int lane, unsigned long long* dbg, Between between) {
  constexpr int TM = 32 * NTT;
  constexpr int NW = 8 / FT;
  const int c32 = lane & 31, hh = lane >> 5;
  float* red_a = red;
  float* red_b = red + NW * TM;
  float mean[NTT], rstd[NTT];
  if (OP::kTwoPassLN) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float s = 0.f;
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[ft][tt][r];
      s = xor32_sum(s);
      if (hh == 0) red_a[wave * TM + tt * 32 + c32] = s;
    }
    SCLDM_LN_STAMP(0);
    lds_barrier();
    SCLDM_LN_STAMP(1);
    between();
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const int t = tt * 32 + c32;
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) m += red_a[w * TM + t];
      mean[tt] = m * (1.0f / kD);
      float s = 0.f;
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = v[ft][tt][r] - mean[tt];
          s = fmaf(d, d, s);
        }
      s = xor32_sum(s);
      if (hh == 0) red_b[wave * TM + t] = s;
    }
    SCLDM_LN_STAMP(2);
    lds_barrier();
    SCLDM_LN_STAMP(3);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const int t = tt * 32 + c32;
      float var = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) var += red_b[w * TM + t];
      rstd[tt] = __builtin_amdgcn_rsqf(var * (1.0f / kD) + eps);
    }
  } else {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s += v[ft][tt][r];
          ss = fmaf(v[ft][tt][r], v[ft][tt][r], ss);
        }
      s = xor32_sum(s);
      ss = xor32_sum(ss);
      if (hh == 0) {
        red_a[wave * TM + tt * 32 + c32] = s;
        red_b[wave * TM + tt * 32 + c32] = ss;
      }
    }
    SCLDM_LN_STAMP(0);
    between();  // after the statistics: whatever it publishes has had the whole sweep to arrive
    lds_barrier();
    SCLDM_LN_STAMP(3);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const int t = tt * 32 + c32;
      float m = 0.f, e2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        m += red_a[w * TM + t];
        e2 += red_b[w * TM + t];
      }
      mean[tt] = m * (1.0f / kD);
      rstd[tt] = __builtin_amdgcn_rsqf(fmaxf(fmaf(-mean[tt], mean[tt], e2 * (1.0f / kD)), 0.f) + eps);
    }
  }
  const int sp = c32 >> 4;
  float nmr[NTT];   // -mean * rstd
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) nmr[tt] = -mean[tt] * rstd[tt];
  // 16 (scale, shift) quads per lane: the LDS reads of quad b+1 are issued before quad b is computed (hipcc otherwise emits
  // read -> wait -> compute -> write per quad, ~130 stalled cycles each)
  constexpr int NB = NTT * FT * 4;
  auto quad_ptr = [&](int b, int vec) {
    const int tt = b / (FT * 4), ft = (b / 4) % FT, q = b % 4;
    return msm + (tt * 2 + sp) * kModBlock + vec * kD + (wave * FT + ft) * 32 + q * 8 + hh * 4;
  };
  f32x4 sc_n = OP::load_mod4(quad_ptr(0, sc_v)), sh_n = OP::load_mod4(quad_ptr(0, sh_v));
  typename OP::Quad held;   // the even quad of a pair, kept until its odd neighbour is ready (SCLDM_PAIR_STORE)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int tt = b / (FT * 4), ft = (b / 4) % FT, q = b % 4;
    const f32x4 sc = sc_n, sh = sh_n;
    if (b + 1 < NB) {
      sc_n = OP::load_mod4(quad_ptr(b + 1, sc_v));
      sh_n = OP::load_mod4(quad_ptr(b + 1, sh_v));
    }
    const int f = (wave * FT + ft) * 32 + q * 8 + hh * 4;
    float y[4];
#if SCLDM_LEAN_VALU
    // the LDS copy of a scale vector already holds 1 + scale (added once when the vectors are staged), so an element costs
    //   bf16 path:   n = v * rstd + (-mean * rstd);  y = n * S + sh                      (2 ops)
    //   parity paths: n = (v - mean) * rstd (centred first: no cancellation);  y = n * S + sh   (3 ops)
    if (!OP::kTwoPassLN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = fmaf(fmaf(v[ft][tt][q * 4 + i], rstd[tt], nmr[tt]), sc[i], sh[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = fmaf((v[ft][tt][q * 4 + i] - mean[tt]) * rstd[tt], sc[i], sh[i]);
    }
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = (v[ft][tt][q * 4 + i] - mean[tt]) * rstd[tt] * (1.0f + sc[i]) + sh[i];
#endif
#if SCLDM_PAIR_STORE
    const typename OP::Quad packed = OP::pack4(y[0], y[1], y[2], y[3]);
    if ((q & 1) == 0) held = packed;
    else OP::store_quad_pair(dst + (tt * 32 + c32) * ldd, (wave * FT + ft) * 32 + (q - 1) * 8, hh, held, packed);
#else
    OP::store_quad(dst + (tt * 32 + c32) * ldd, f, OP::pack4(y[0], y[1], y[2], y[3]));
#endif
  }
}

#ifndef SCLDM_FWD_ONE_WG
#define SCLDM_FWD_ONE_WG 0    // experiment builds: 1 = compile every instantiation for ONE wave per SIMD (up to 512 registers: room for -DSCLDM_PF=8 -DSCLDM_W12_PAIR=1)
#endif
template <typename OP, int NTT, int FT, bool REC = false>
__global__ __launch_bounds__(64 * (8 / FT), (!SCLDM_FWD_ONE_WG && ((OP::kTwoWG && NTT <= 2) || NTT == 1)) ? 2 : 1) void dit_forward_kernel(const FwdArgs a) {
  using L = FwdLayout<OP, NTT, FT>;
  using E = typename OP::E;
  using Frag = typename OP::Frag;
  using Quad = typename OP::Quad;
  constexpr int TM = L::TM;
  constexpr int NS = L::NS;
  constexpr int NW = L::NW;
  constexpr int NT = L::NT;
  constexpr int PF = Prefetch<OP, NTT, FT>::PF;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  E* XA = reinterpret_cast<E*>(smem);
  E* HB = reinterpret_cast<E*>(smem + L::XA_BYTES);
  using ModE = typename OP::ModE;
  float* RED = reinterpret_cast<float*>(smem + L::RED_OFF);
  ModE* MOD = reinterpret_cast<ModE*>(smem + L::MOD_OFF);
  E* AO = XA;  // the attention output reuses the LN1 buffer once every wave has finished its Q/K/V passes

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_id = blockIdx.x + a.tile0;
  const int tok0 = tile_id * TM;
  const int smp0 = tile_id * NS;
  const int fbase = wave * FT * 32;  // first feature owned by this wave
  auto nothing = [] {};

  SCLDM_STAMP(0);
  constexpr int kModLd = (NS * kModBlock / 4 + NT - 1) / NT;  // float4 of adaLN vectors per thread
  // the weight stream starts first: its first PF units fly while the prologue runs
  WStream<OP, PF, FT> ws;
  ws.init(reinterpret_cast<const Frag*>(a.w_stream) + (size_t)wave * units_per_layer(a.n_chunks, a.half_chunk) * 64 * FT, lane);

  // samples past n_fwd (tile padding) recompute the last real sample and are never stored to `out`
  // residual hand-off buffer: lane-linear, quad j = (tt*FT + ft)*4 + q  (padded to whole tiles: no predication)
  // (computed where it is used: kept in registers across the layer body it is one more value the allocator has to carry)
  auto xw_ptr = [&]() { return a.x + ((size_t)(tile_id * NW + wave) * (4 * FT * NTT) * 64 + (threadIdx.x & 63)) * 4; };

  // The residual stream (this wave's features x TM tokens, accumulator layout; scalars: it never feeds an MFMA and
  // whole-vector values would be copied around by the compiler).  With FT=2 the allocator parks ~1/3 of it in scratch
  // during attention / MLP.  Parking it explicitly in the hand-off buffer and re-reading it ahead of the preceding GEMM
  // was measured SLOWER (451 vs 396 us): vmcnt retires in order, so every wait on a weight-ring load issued after
  // the slow residual loads/stores also waits for them.
  float xr[FT][NTT][16];
  auto load_x = [&](float (&dst)[FT][NTT][16]) {
    const float* xw = xw_ptr();
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(xw + ((tt * FT + ft) * 4 + q) * 256);
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[ft][tt][q * 4 + i] = t4[i];
        }
  };
  auto store_x = [&](const float (&src)[FT][NTT][16]) {
    float* xw = xw_ptr();
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 t4;
#pragma unroll
          for (int i = 0; i < 4; ++i) t4[i] = src[ft][tt][q * 4 + i];
          *reinterpret_cast<f32x4*>(xw + ((tt * FT + ft) * 4 + q) * 256) = t4;
        }
  };
  // training record (REC): same lane-linear tile layout as the hand-off buffer
  static_assert(!REC || NTT <= 2, "the record is laid out in 64-token tiles");
  const int rec_tile = NTT == 2 ? tile_id : tile_id >> 1, rec_tt0 = NTT == 2 ? 0 : tile_id & 1;
  auto rec_store_x = [&](const float (&src)[FT][NTT][16], int layer_idx) {
    // (the record keeps the 64-token-tile geometry whatever this kernel's tile: a 32-token tile is half tt0 of 64-token tile tile_id / 2)
    float* xw = a.rec_x + (size_t)layer_idx * a.rec_stride + ((size_t)(rec_tile * NW + wave) * (4 * FT * 2) * 64 + (threadIdx.x & 63)) * 4;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 t4;
#pragma unroll
          for (int i = 0; i < 4; ++i) t4[i] = src[ft][tt][q * 4 + i];
          *reinterpret_cast<f32x4*>(xw + (((tt + rec_tt0) * FT + ft) * 4 + q) * 256) = t4;
        }
  };
  auto rec_store_y = [&](const f32x16 (&src)[FT][NTT], void* base, int layer_idx) {
    E* yw = reinterpret_cast<E*>(base) + (size_t)layer_idx * a.rec_stride + ((size_t)(rec_tile * NW + wave) * (4 * FT * 2) * 64 + (threadIdx.x & 63)) * 4;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (sizeof(E) == 2)   // (the recording instantiations are the 16-bit policies)
            *reinterpret_cast<Quad*>(yw + (((tt + rec_tt0) * FT + ft) * 4 + q) * 256) =
                OP::pack4(src[ft][tt][q * 4 + 0], src[ft][tt][q * 4 + 1], src[ft][tt][q * 4 + 2], src[ft][tt][q * 4 + 3]);
        }
  };
  // Up to four consecutive layers per launch: the body below is instantiated once per layer slot (compile-time `li`), the
  // residual `xr` stays in registers between slots and only the first slot reads / the last slot writes the hand-off
  // buffer.  (A run-time loop over layers made the register allocator spill 170 VGPRs.)  Measured, cells/s at the
  // default workload in one session: 1 layer per launch 15.3 k, 2: 16.2 k, 4: 16.5 k, 8: 16.3 k.
  auto layer_body = [&](auto li_tag) {
  constexpr int li = decltype(li_tag)::value;
  const int layer = a.layer + li;
  // The second slot re-derives its per-lane indices from an opaque copy of the lane id: shared with the first slot, the
  // common address arithmetic would stay live across the whole first layer (register pressure -> spills).
  int lane_sh = threadIdx.x & 63;
  if (li > 0) asm volatile("" : "+v"(lane_sh));
  const int lane = lane_sh;
  const int tid = wave * 64 + lane;
  const int c32 = lane & 31, hh = lane >> 5;
  const int sp = c32 >> 4;  // which of a 32-token tile's two samples this lane's token belongs to
  if (li == 0 && layer == 0) {   // (only the first slot of a launch can be layer 0: later slots carry no copy of this code)
    // ---- input projection + positional embedding (nnets.py:290-291) ----
    const int p16 = c32 & 15;  // token position inside its sample
    const float* zrow[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const int s = min((tok0 + tt * 32 + c32) >> 4, a.n_fwd - 1);
      const int src = (s < a.n_direct) ? s : (a.n_direct - a.rep + (s - a.n_direct) % a.rep);
      zrow[tt] = a.z + ((size_t)src * 16 + p16) * a.din;
    }
    if constexpr (OP::kMfmaIn) {
      // bf16 / split-bf16 paths: K = din <= 32 is one or two MFMA k-steps.  Operands straight from global memory (this lane's 8 k-values
      // of its weight row / its token's latent row), all loads issued together.  The former per-k VALU loop with its
      // dependent global loads made the first layer's launch 114 us longer than the others (430 vs 316 us).
      auto frag_of = [&](const float* row, int k0) {
        float t[8];
        if ((a.din & 7) == 0 && k0 + 8 <= a.din) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(row + k0), hi = *reinterpret_cast<const f32x4*>(row + k0 + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { t[i] = lo[i]; t[4 + i] = hi[i]; }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) t[i] = (k0 + i < a.din) ? row[k0 + i] : 0.f;
        }
        return OP::pack8(t);
      };
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 pin[FT][NTT];
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) pin[ft][tt] = zero;
      for (int ks = 0; ks * 16 < a.din; ++ks) {
        Frag wf[FT], zf[NTT];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) wf[ft] = frag_of(a.in_w + (size_t)(fbase + ft * 32 + c32) * a.din, ks * 16 + hh * 8);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) zf[tt] = frag_of(zrow[tt], ks * 16 + hh * 8);
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
          for (int tt = 0; tt < NTT; ++tt) pin[ft][tt] = OP::mma(wf[ft], zf[tt], pin[ft][tt]);
      }
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = fbase + ft * 32 + q * 8 + hh * 4;
          const f32x4 bb = *reinterpret_cast<const f32x4*>(a.in_b + f);
          const f32x4 pp = *reinterpret_cast<const f32x4*>(a.pos + p16 * kD + f);
#pragma unroll
          for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[ft][tt][q * 4 + i] = pin[ft][tt][q * 4 + i] + bb[i] + pp[i];
        }
    } else {
      // fp32 parity path: exact fp32 on the VALU (K = din)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = fbase + ft * 32 + q * 8 + hh * 4;
          const f32x4 bb = *reinterpret_cast<const f32x4*>(a.in_b + f);
          const f32x4 pp = *reinterpret_cast<const f32x4*>(a.pos + p16 * kD + f);
#pragma unroll
          for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[ft][tt][q * 4 + i] = bb[i] + pp[i];
        }
      for (int k = 0; k < a.din; ++k) {
        float zk[NTT];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) zk[tt] = zrow[tt][k];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.in_wt + k * kD + fbase + ft * 32 + q * 8 + hh * 4);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
              for (int i = 0; i < 4; ++i) xr[ft][tt][q * 4 + i] = fmaf(w4[i], zk[tt], xr[ft][tt][q * 4 + i]);
          }
      }
    }
  } else if (li == 0) {
    load_x(xr);
  }
  if constexpr (REC) rec_store_x(xr, layer);

  // the tile's six adaLN vectors per sample: coalesced loads (issued AFTER the residual loads: the row_index -> mod
  // lookup is a dependent chain and would otherwise hold them back), parked in LDS during LN1 (16 lanes share every
  // value, so per-lane global loads would be 16x redundant and - measured - fully latency-exposed)
  f32x4 mstage[kModLd];
#pragma unroll
  for (int j = 0; j < kModLd; ++j) {
    const int idx = min(tid + NT * j, NS * kModBlock / 4 - 1), sl = idx / (kModBlock / 4), w4 = idx % (kModBlock / 4);
    const int s = min(smp0 + sl, a.n_fwd - 1);
    mstage[j] = *reinterpret_cast<const f32x4*>(a.mod + (size_t)a.row_index[s] * a.mod_stride + layer * kModBlock + w4 * 4);
  }


  // c_attn / c_proj biases: staged into LDS too (aliasing HB, which is free until the SwiGLU phase; LN2's barriers
  // separate the last bias read from the first hidden-chunk write).  Read from global memory in the pass epilogues they
  // cost 9 % of the kernel (383 -> 348 us in a no-bias timing proxy): vmcnt retires in order, so waiting for a bias
  // load issued behind the weight ring's run-ahead refills drains the ring at every epilogue.
  constexpr int kBiasLd = (4 * kD / 4 + NT - 1) / NT;  // float4 per thread for 3*kD + kD floats
  f32x4 bstage[kBiasLd];
#pragma unroll
  for (int j = 0; j < kBiasLd; ++j) {
    const int idx = min(tid + NT * j, 4 * kD / 4 - 1);
    bstage[j] = idx < 3 * kD / 4 ? *reinterpret_cast<const f32x4*>(a.b_qkv + li * 3 * kD + idx * 4)
                                 : *reinterpret_cast<const f32x4*>(a.b_proj + li * kD + (idx - 3 * kD / 4) * 4);
  }
  float* BIAS = reinterpret_cast<float*>(HB);
#if SCLDM_L2WARM
  // L2 warm-up of the next layer's stream (the first layer's for the last one: the next evaluation starts there).  The 64
  // workgroups an XCD runs at a time (consecutive blockIdx / 8) split the 1.6 MB into 128-byte lines; one dword per line.
  unsigned warm = 0;
  {
    const long lbytes = a.w_layer_elems * (long)sizeof(E);
    const char* nxt = reinterpret_cast<const char*>(a.w_stream) + (layer + 1 < a.n_layer ? (long)(li + 1) * lbytes : -(long)a.layer * lbytes);
    const int lines = (int)(lbytes >> 7), per = (lines + 63) >> 6;
    const int ln = ((blockIdx.x >> 3) & 63) * per + tid;
    if (tid < per && ln < lines) warm = *reinterpret_cast<const unsigned*>(nxt + ((long)ln << 7));
  }
#endif

  f32x16 acc[FT][NTT];
  const float* bq = BIAS;
  const float* bp = BIAS + 3 * kD;

  // ---- LN1 + modulate(a0 = scale, a1 = shift) -> XA (the staged adaLN vectors are published on the way) ----
  SCLDM_STAMP(21);
  ln_modulate_store<OP, NTT, FT, 22>(xr, MOD, 0, 1, a.eps, RED, XA, L::XA_LD, wave, lane, a.dbg, [&] {
#if SCLDM_L2WARM
    asm volatile("" :: "v"(warm));   // the warm-up load retires with the staged vectors (issued together): no later wait inherits it
#endif
#pragma unroll
    for (int j = 0; j < kModLd; ++j)
      if (tid + NT * j < NS * kModBlock / 4) {
#if SCLDM_LEAN_VALU
        const int vec = ((tid + NT * j) % (kModBlock / 4)) / (kD / 4);   // which of the six vectors this float4 belongs to
        if (vec == 0 || vec == 3) mstage[j] += 1.0f;                      // a0 / a3 act as SCALES (F7): stored as 1 + scale
#endif
        OP::store_mod4(MOD + (size_t)(tid + NT * j) * 4, mstage[j]);
      }
#pragma unroll
    for (int j = 0; j < kBiasLd; ++j)
      if (tid + NT * j < 4 * kD / 4) *reinterpret_cast<f32x4*>(BIAS + (size_t)(tid + NT * j) * 4) = bstage[j];
  });
  SCLDM_STAMP(26);
  lds_barrier();
  SCLDM_STAMP(1);

  // ---- attention for this wave's FT heads, entirely in registers ----
  //   Q pass, K pass -> S^T = K Q^T -> softmax -> P  (Q, K fragments die here: keeps the live MFMA
  //   operand set inside the architectural VGPRs) -> V pass -> O^T = V^T P^T (kept in acc).
  {
    Frag Pf[FT][NTT][2];
    // scores + softmax of one head: S^T = K Q^T (k = head dims), masked to the query's own sample, P packed as the next operand
    auto scores = [&](const Frag (&QFh)[NTT][2], const Frag (&KFh)[NTT][2], Frag (&Ph)[NTT][2]) {
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        f32x16 st = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        st = OP::mma(KFh[tt][0], QFh[tt][0], st);  // S^T[key][query]
        st = OP::mma(KFh[tt][1], QFh[tt][1], st);
        float sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // keys of the query's own sample (opaque copies keep this a v_cndmask,
          float lo = st[i], hi = st[8 + i];  // not a dynamically indexed vector extract)
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
        // P of the query's own sample is packed ONCE; the two k-halves of the operand (keys of sample 0 | keys of sample 1) are
        // that fragment or zero, selected per packed dword: cross-sample blocks of the shared 32x32 tile are exactly zero
        float p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = sv[i] * inv;
        union FragBits { Frag f; unsigned u[sizeof(Frag) / 4]; } own, lo_half, hi_half;
        own.f = OP::pack8(p);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(Frag) / 4); ++i) {
          lo_half.u[i] = sp ? 0u : own.u[i];
          hi_half.u[i] = sp ? own.u[i] : 0u;
        }
        Ph[tt][0] = lo_half.f;
        Ph[tt][1] = hi_half.f;
      }
    };
    // accumulator tile (feature rows x token cols) + per-row bias -> the two k-halves of an MFMA operand
    // the per-row bias of a feature tile as an accumulator tile (register r <-> row acc_row(r, hh)): the pass's initial C operand
    auto bias_tile = [&](const float* brow) {
      f32x16 t;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + q * 8 + hh * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) t[q * 4 + i] = b4[i];
      }
      return t;
    };
    auto to_frags_nobias = [&](const f32x16 (&t_acc)[NTT], Frag (&F)[NTT][2]) {
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = t_acc[tt][r];
        F[tt][0] = OP::pack8(t);
        F[tt][1] = OP::pack8(t + 8);
      }
    };
    auto to_frags = [&](const f32x16 (&t_acc)[NTT], const float* brow, Frag (&F)[NTT][2]) {
      float bias[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + q * 8 + hh * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[q * 4 + i] = b4[i];
      }
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = t_acc[tt][r] + bias[r];
        F[tt][0] = OP::pack8(t);
        F[tt][1] = OP::pack8(t + 8);
      }
    };
    if constexpr (FT == 2) {
      // one head at a time: a two-tile pass whose tiles are the head's Q and K (the activation fragments are read from
      // LDS once for both), then its scores.  Only one head's Q/K fragments are live next to the residual - with both
      // heads' (the natural Q pass, K pass order) the allocator parked the residual in scratch, and scratch traffic shares
      // the in-order vmcnt queue with the weight ring.
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        Frag QFh[NTT][2], KFh[NTT][2];
#if SCLDM_LEAN_VALU
        const f32x16 qk_bias[2] = {bias_tile(bq + 0 * kD + fbase + ft * 32), bias_tile(bq + 1 * kD + fbase + ft * 32)};
        gemm_pass<OP, NTT, FT, 16, false, true, PF>(acc, ws, XA, L::XA_LD, lane, qk_bias);  // acc[0] = Q^T + b_q, acc[1] = K^T + b_k of head ft
        to_frags_nobias(acc[0], QFh);
        to_frags_nobias(acc[1], KFh);
#else
        gemm_pass<OP, NTT, FT, 16, false, true, PF>(acc, ws, XA, L::XA_LD, lane);  // acc[0] = Q^T, acc[1] = K^T of head ft
        to_frags(acc[0], bq + 0 * kD + fbase + ft * 32, QFh);
        to_frags(acc[1], bq + 1 * kD + fbase + ft * 32, KFh);
#endif
        if (ft == 0) SCLDM_STAMP(2);
        if (ft == 0) SCLDM_STAMP(3);
        scores(QFh, KFh, Pf[ft]);
      }
    } else {
      Frag QF[FT][NTT][2], KF[FT][NTT][2];
      gemm_pass<OP, NTT, FT, 16, false, true, PF>(acc, ws, XA, L::XA_LD, lane);  // Q^T (feature x token)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) to_frags(acc[ft], bq + 0 * kD + fbase + ft * 32, QF[ft]);
      SCLDM_STAMP(2);
      gemm_pass<OP, NTT, FT, 16, false, true, PF>(acc, ws, XA, L::XA_LD, lane);  // K^T (feature x token)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) to_frags(acc[ft], bq + 1 * kD + fbase + ft * 32, KF[ft]);
      SCLDM_STAMP(3);
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) scores(QF[ft], KF[ft], Pf[ft]);
    }
    SCLDM_STAMP(4);
    // V (swapped operands: lane = feature, registers = tokens), then O^T = V^T P^T
    gemm_pass<OP, NTT, FT, 16, true, true, PF>(acc, ws, XA, L::XA_LD, lane);
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
#if SCLDM_LEAN_VALU
      // the V bias is added AFTER the P V product, as the initial accumulator: every query's probabilities sum to one over its
      // sample's keys (the cross-sample blocks of P are exact zeros), so P (V + 1 b^T) = P V + b per output row d
      const float* vb_row = bq + 2 * kD + fbase + ft * 32;
#else
      const float bv = bq[2 * kD + fbase + ft * 32 + c32];
      const f32x16 ot0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        float t[16];
#pragma unroll
#if SCLDM_LEAN_VALU
        for (int r = 0; r < 16; ++r) t[r] = acc[ft][tt][r];
#else
        for (int r = 0; r < 16; ++r) t[r] = acc[ft][tt][r] + bv;
#endif
        const Frag v0 = OP::pack8(t), v1 = OP::pack8(t + 8);
#if SCLDM_LEAN_VALU
        const float* vb = vb_row;
        asm volatile("" : "+v"(vb));        // re-read per token tile: holding the bias tile across both costs 16 registers where pressure peaks
        f32x16 ot = bias_tile(vb);
#else
        f32x16 ot = ot0;
#endif
        ot = OP::mma(v0, Pf[ft][tt][0], ot);  // O^T[d][query], k = keys
        ot = OP::mma(v1, Pf[ft][tt][1], ot);
        acc[ft][tt] = ot;
      }
    }
  }
  SCLDM_STAMP(5);
  lds_barrier();  // every wave is done reading XA (Q/K/V passes): it may now be overwritten by the attention output
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
#if SCLDM_PAIR_STORE
      for (int q = 0; q < 4; q += 2)
        OP::store_quad_pair(AO + (tt * 32 + c32) * L::XA_LD, fbase + ft * 32 + q * 8, hh,
                            OP::pack4(acc[ft][tt][q * 4 + 0], acc[ft][tt][q * 4 + 1], acc[ft][tt][q * 4 + 2], acc[ft][tt][q * 4 + 3]),
                            OP::pack4(acc[ft][tt][q * 4 + 4], acc[ft][tt][q * 4 + 5], acc[ft][tt][q * 4 + 6], acc[ft][tt][q * 4 + 7]));
#else
      for (int q = 0; q < 4; ++q)
        OP::store_quad(AO + (tt * 32 + c32) * L::XA_LD, fbase + ft * 32 + q * 8 + hh * 4,
                       OP::pack4(acc[ft][tt][q * 4 + 0], acc[ft][tt][q * 4 + 1], acc[ft][tt][q * 4 + 2], acc[ft][tt][q * 4 + 3]));
#endif
  lds_barrier();  // AO complete
  SCLDM_STAMP(6);

  // ---- attention projection, gated residual (a2), LN2 + modulate(a3 = scale, a4 = shift) -> XA ----
#if SCLDM_LEAN_VALU
  {
    f32x16 pbias[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + fbase + ft * 32 + q * 8 + hh * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) pbias[ft][q * 4 + i] = b4[i];
      }
    gemm_pass<OP, NTT, FT, 16, false, true, PF>(acc, ws, AO, L::XA_LD, lane, pbias);   // c_proj(attention) + bias
  }
  if constexpr (REC) rec_store_y(acc, a.rec_y1, layer);
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = fbase + ft * 32 + q * 8 + hh * 4;
        const f32x4 g = OP::load_mod4(MOD + (tt * 2 + sp) * kModBlock + 2 * kD + f);
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[ft][tt][q * 4 + i] = fmaf(g[i], acc[ft][tt][q * 4 + i], xr[ft][tt][q * 4 + i]);
      }
#else
  gemm_pass<OP, NTT, FT, 16, false, true, PF>(acc, ws, AO, L::XA_LD, lane);
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = fbase + ft * 32 + q * 8 + hh * 4;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + f);
        const f32x4 g = OP::load_mod4(MOD + (tt * 2 + sp) * kModBlock + 2 * kD + f);
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[ft][tt][q * 4 + i] = fmaf(g[i], acc[ft][tt][q * 4 + i] + b4[i], xr[ft][tt][q * 4 + i]);
      }
#endif
  SCLDM_STAMP(7);
  // (the statistics barrier inside also guarantees every wave has finished reading AO before XA is rewritten)
  ln_modulate_store<OP, NTT, FT, 16>(xr, MOD, 3, 4, a.eps, RED, XA, L::XA_LD, wave, lane, a.dbg, nothing);
  SCLDM_STAMP(20);
  lds_barrier();  // XA (MLP input) complete
  SCLDM_STAMP(8);

  // ---- SwiGLU MLP, hidden processed in chunks of 128 (NW waves x FT tiles x 16 hidden) staged through HB ----
  f32x16 accp[FT][NTT];
  // one SwiGLU chunk; HALF (compile time) = the trailing 64-unit chunk: one W12 tile per wave, K = 64 down-projection
  auto do_chunk = [&](auto half_tag, int c) {
    constexpr bool HALF = decltype(half_tag)::value;
    constexpr int TILES = (FT == 2 && HALF) ? 1 : FT;
    // rows 0-15 of each weight tile are w1, rows 16-31 the matching w2 rows => registers r and r+8 pair up
    Quad hq[TILES][NTT][2];
    // one 32-row tile per pass: a two-tile up-projection pass (activation fragments read once) was measured at 30 spilled
    // VGPRs and +1 % kernel time
    constexpr bool kPair = SCLDM_W12_PAIR != 0;
    auto swiglu_pack = [&](const f32x16 (&t_acc)[NTT], Quad (&out)[NTT][2]) {
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float h[4];
#pragma unroll
#if SCLDM_PROXY & 4    // timing proxy 4 (WRONG RESULTS): SwiGLU replaced by a single multiply - what its exposed VALU costs
          for (int i = 0; i < 4; ++i) h[i] = t_acc[tt][q * 4 + i] * t_acc[tt][8 + q * 4 + i];
#else
          for (int i = 0; i < 4; ++i) h[i] = OP::swiglu(t_acc[tt][q * 4 + i], t_acc[tt][8 + q * 4 + i]);
#endif
          out[tt][q] = OP::pack4(h[0], h[1], h[2], h[3]);
        }
    };
    if constexpr (FT == 2) {
      if constexpr (HALF || !kPair) {
#pragma unroll
        for (int ft = 0; ft < TILES; ++ft) {
          f32x16 a1[NTT];
          gemm_pass_tile<OP, NTT, PF>(a1, ws, XA, L::XA_LD, lane);
          swiglu_pack(a1, hq[ft]);
        }
      } else {
        gemm_pass<OP, NTT, FT, 16, false, true, PF>(acc, ws, XA, L::XA_LD, lane);
#pragma unroll
        for (int ft = 0; ft < TILES; ++ft) swiglu_pack(acc[ft], hq[ft]);
      }
    } else {
      gemm_pass<OP, NTT, FT, 16, false, true, PF>(acc, ws, XA, L::XA_LD, lane);
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) swiglu_pack(acc[ft], hq[ft]);
    }
    E* HBc = HB + (L::HB_BUFS == 2 ? (c & 1) * (L::HB_BYTES / (int)sizeof(E)) : 0);
    if (L::HB_BUFS == 1 && c > 0) lds_barrier();  // single buffer: every wave must have finished the previous chunk's c_proj pass
#pragma unroll
    for (int ft = 0; ft < TILES; ++ft) {
      const int col0 = HALF ? wave * 16 : (wave * FT + ft) * 16;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
#if SCLDM_PAIR_STORE
        OP::store_quad_pair(HBc + (tt * 32 + c32) * L::HB_LD, col0, hh, hq[ft][tt][0], hq[ft][tt][1]);
#else
#pragma unroll
        for (int q = 0; q < 2; ++q)
          OP::store_quad(HBc + (tt * 32 + c32) * L::HB_LD, col0 + q * 8 + hh * 4, hq[ft][tt][q]);
#endif
      }
    }
    if (c == 0) SCLDM_STAMP(11);
    lds_barrier();
    if (c == 0) SCLDM_STAMP(12);
    constexpr int KS = HALF ? 4 : 8;
    if (c == 0) gemm_pass<OP, NTT, FT, KS, false, true, PF>(accp, ws, HBc, L::HB_LD, lane);
    else gemm_pass<OP, NTT, FT, KS, false, false, PF>(accp, ws, HBc, L::HB_LD, lane);
    if (c == 0) SCLDM_STAMP(13);
  };
  for (int c = 0; c < a.n_chunks; ++c) do_chunk(std::false_type{}, c);
  if (a.half_chunk) do_chunk(std::true_type{}, a.n_chunks);
  SCLDM_STAMP(9);

  // another layer follows in this launch: its weight ring starts now (the ring registers are free after the last pass),
  // so the first units are in flight during the gated residual, the hand-over barrier and the next LayerNorm
  const bool next_here = layer + 1 < a.n_layer && li + 1 < a.n_here;
  if (next_here)
    ws.init(reinterpret_cast<const Frag*>(a.w_stream) + (size_t)(li + 1) * (a.w_layer_elems / 8) +
            (size_t)wave * units_per_layer(a.n_chunks, a.half_chunk) * 64 * FT, lane);

  if constexpr (REC) rec_store_y(accp, a.rec_y2, layer);
  // ---- gated residual (a5) ----
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = fbase + ft * 32 + q * 8 + hh * 4;
        const f32x4 g = OP::load_mod4(MOD + (tt * 2 + sp) * kModBlock + 5 * kD + f);
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[ft][tt][q * 4 + i] = fmaf(g[i], accp[ft][tt][q * 4 + i], xr[ft][tt][q * 4 + i]);
      }
  SCLDM_STAMP(10);

  if (layer + 1 < a.n_layer) {
    if (li + 1 >= a.n_here) {
      store_x(xr);  // hand the residual to the next launch
    } else {
      // next layer in this launch: every wave is done with this layer's adaLN vectors, biases and hidden buffers before
      // they are overwritten (its weight ring was restarted above)
      lds_barrier();
    }
  } else {
    // ---- final layer (layers.py:397-401): LN -> *(1+scale)+shift with (shift, scale) = chunks (0,1) -> Linear 256->din ----
    if constexpr (REC) rec_store_x(xr, a.n_layer);
    constexpr int kFinLd = (NS * 2 * kD / 4 + NT - 1) / NT;  // float4 per thread for the tile's (shift, scale) vectors
    f32x4 fstage[kFinLd];
#pragma unroll
    for (int j = 0; j < kFinLd; ++j) {
      const int idx = min(tid + NT * j, NS * 2 * kD / 4 - 1), sl = idx / (2 * kD / 4), w4 = idx % (2 * kD / 4);
      const int s = min(smp0 + sl, a.n_fwd - 1);
      fstage[j] = *reinterpret_cast<const f32x4*>(a.mod + (size_t)a.row_index[s] * a.mod_stride + a.n_layer * kModBlock + w4 * 4);
    }
    lds_barrier();  // every wave has consumed its a5 gate: vector slots 0/1 of MOD can be replaced
#pragma unroll
    for (int j = 0; j < kFinLd; ++j) {
      const int idx = tid + NT * j, sl = idx / (2 * kD / 4), w4 = idx % (2 * kD / 4);
#if SCLDM_LEAN_VALU
      if (w4 >= kD / 4) fstage[j] += 1.0f;   // the final layer's second chunk is its scale (layers.py:398-399)
#endif
      if (idx < NS * 2 * kD / 4) OP::store_mod4(MOD + sl * kModBlock + w4 * 4, fstage[j]);
    }
    ln_modulate_store<OP, NTT, FT>(xr, MOD, 1, 0, a.eps, RED, XA, L::XA_LD, wave, lane, a.dbg, nothing);
    lds_barrier();
    // wave w projects token tile tt == w; the 16 weight fragments (rows >= din zero) are tiny and L2-hot
    const Frag* wf = reinterpret_cast<const Frag*>(a.w_final) + lane;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      if (tt != wave) continue;  // wave-uniform (NTT <= NW)
      f32x16 o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {   // fully unrolled: the 16 L2-hot weight fragments are requested together (the residual registers are dead here)
        const Frag wfr = wf[ks * 64];
        const Frag b = *reinterpret_cast<const Frag*>(XA + (tt * 32 + c32) * L::XA_LD + hh * 8 + ks * 16);
        o = OP::mma(wfr, b, o);  // out^T[channel][token]
      }
      if (((tok0 + tt * 32 + c32) >> 4) < a.n_fwd) {   // samples past n_fwd (tile padding) are never stored
        float* op = a.out + ((size_t)(tok0 + tt * 32 + c32)) * a.din;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ch = acc_row(r, hh);
          if (ch < a.din) op[ch] = o[r] + a.fin_b[ch];
        }
      }
    }
  }
  };  // layer_body
  layer_body(std::integral_constant<int, 0>{});
  if (a.n_here > 1) layer_body(std::integral_constant<int, 1>{});
  if (a.n_here > 2) layer_body(std::integral_constant<int, 2>{});
  if (a.n_here > 3) layer_body(std::integral_constant<int, 3>{});
#if SCLDM_MAX_LPL > 4
  if constexpr (!REC || SCLDM_REC_LPL == 8) {
    if (a.n_here > 4) layer_body(std::integral_constant<int, 4>{});
    if (a.n_here > 5) layer_body(std::integral_constant<int, 5>{});
    if (a.n_here > 6) layer_body(std::integral_constant<int, 6>{});
    if (a.n_here > 7) layer_body(std::integral_constant<int, 7>{});
  }
#endif
  static_assert(kMaxLayersPerLaunch == 4 || kMaxLayersPerLaunch == 8, "one layer_body call per slot");
  SCLDM_STAMP_END(14);
}

}  // namespace scldm

#pragma clang fp contract(fast)
