// Fused adaLN-Zero DiT block for gfx950 (one launch = one transformer block over all tokens).
//
// Replaces, per block, the ~25-30 eager kernels of the reference's
//   Block.forward adaLN branch           src/scldm/layers.py:213-221  (+ modulate :91-94, F7 order)
//   SelfAttention.forward                src/scldm/layers.py:143-158
//   MLP.forward (SwiGLU)                 src/scldm/layers.py:173-174
// Specialised to the reference's only DiT shape family: n_embed 256, 8 heads x 32, seq_len 16
// (experiments/configs/model/ldm_base.yaml:16-25); hidden (684) is zero-padded to a multiple of 128.
//
// Work decomposition (MI355X-first, not a GEMM-library composition):
//   * one workgroup = 4 waves (one per SIMD) owns a tile of TM = 32*NTT tokens (= 2*NTT samples);
//   * GEMMs are computed TRANSPOSED, Y^T[feature][token] = W[feature][k] * X^T[k][token]:
//       A operand = weights, streamed straight from L2 into VGPRs in a pre-packed fragment order
//                   (each weight byte is read by exactly one wave of the workgroup -> no LDS staging),
//       B operand = activations, shared by all four waves through LDS ([token][feature], +16 B row pad
//                   => conflict-free ds_read_b128);
//     wave w owns output features [64w, 64w+64) = heads 2w, 2w+1, for ALL tokens of the tile, so
//     LayerNorm statistics are an in-lane sum + one xor-32 exchange + a 4-way LDS combine, and
//   * attention never leaves registers: Q^T and K^T tiles come out of the MFMA in a layout that is
//     directly a valid A/B operand pair for S^T = K Q^T (any k permutation is legal if both sides
//     share it); V is produced with swapped operands (V[token][d]) so that O^T = V^T P^T likewise
//     needs no transpose; softmax over the 16 keys is 8 in-lane values + one xor-32 exchange.
//     Two samples share each 32x32 MFMA tile; cross-sample score blocks are masked to zero.
//   * the SwiGLU hidden dimension is processed in chunks of 128 through a double-buffered LDS
//     stage (w1/w2 rows are interleaved inside each 32-row weight tile so silu(a)*b is in-lane).
#pragma once
#include "common.hpp"

namespace scldm {

constexpr int kD = 256;        // n_embed
constexpr int kHC = 128;       // hidden chunk per workgroup (4 waves x 32)
constexpr int kModBlock = 6 * kD;

struct BlockArgs {
  float* x;                 // (n_fwd*16, 256) fp32 residual stream, updated in place
  const float* mod;         // (rows, mod_stride) adaLN vectors for all layers
  const int32_t* row_index; // (n_fwd) conditioning row of each sample-forward
  const void* w_stream;     // this layer's packed weight stream (pack_layer_kernel in dit_aux.hpp)
  const float* b_qkv;       // (768) fp32
  const float* b_proj;      // (256) fp32
  int n_fwd;                // number of sample-forwards (16 tokens each)
  int mod_stride;           // floats per mod row
  int mod_offset;           // offset of this layer's 6*256 block inside a mod row
  int n_chunks;             // padded hidden / 128
  float eps;
  float attn_scale_log2e;   // log2(e) / sqrt(head_dim)
};

template <typename OP, int NTT>
struct BlockLayout {
  using E = typename OP::E;
  static constexpr int TM = 32 * NTT;
  static constexpr int PADE = 16 / sizeof(E);
  static constexpr int XA_LD = kD + PADE;       // elements per activation row
  static constexpr int HB_LD = kHC + PADE;      // elements per hidden-chunk row
  static constexpr int XA_BYTES = TM * XA_LD * sizeof(E);
  static constexpr int HB_BYTES = 2 * TM * HB_LD * sizeof(E);  // double buffer; also hosts attention output
  static constexpr int RED_BYTES = 2 * 4 * TM * sizeof(float);
  static constexpr int LDS_BYTES = XA_BYTES + HB_BYTES + RED_BYTES;
  static_assert(HB_BYTES >= XA_BYTES, "attention output must fit in the hidden double buffer");
};

// ---------------------------------------------------------------------------------------------
// Weight stream.  Every wave consumes ONE contiguous sequence of "units" for the whole layer
// (unit = one k-step of 16 for the wave's two 32-row weight tiles = 2 fragments = 2 KiB bf16):
//     Q (16 units) | K (16) | V (16) | proj (16) | for each hidden chunk: W12 (16) | c_proj (8)
// A PF-deep register ring runs ahead of the MFMAs and persists across passes, so L2 latency is
// hidden across phase boundaries too and nothing is fetched twice.  The ring over-reads PF units
// past the wave's last unit (next wave's stream / allocation slack), which is never consumed.
// ---------------------------------------------------------------------------------------------
constexpr int kUnitsFixed = 64;      // Q,K,V,proj
constexpr int kUnitsPerChunk = 24;   // W12 (16) + c_proj (8)
constexpr int kMaxPF = 8;
__host__ __device__ constexpr int units_per_wave(int n_chunks) { return kUnitsFixed + n_chunks * kUnitsPerChunk; }

template <typename OP, int NTT>
struct Prefetch {  // k-steps of run-ahead: ~1k cycles of MFMA work per ring depth
  static constexpr int PF = OP::kIsBF16 ? (NTT >= 4 ? 4 : 8) : 2;
};

template <typename OP, int PF>
struct WStream {
  using Frag = typename OP::Frag;
  const Frag* p;  // next unit to fetch (lane offset folded in)
  Frag ring[PF][2];
  __device__ __forceinline__ void init(const Frag* base) {
    p = base;
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      ring[s][0] = p[0];
      ring[s][1] = p[64];
      p += 128;
    }
  }
};

// One GEMM pass: acc[ft][tt] += W-tile(ft) * B-tile(tt) over KSTEPS*16 k-values; activations from LDS.
// SWAP=true computes the transposed tile (token rows, feature cols) - used for V.
template <typename OP, int NTT, int KSTEPS, bool SWAP, int PF>
__device__ __forceinline__ void gemm_pass(f32x16 (&acc)[2][NTT], WStream<OP, PF>& ws,
                                          const typename OP::E* __restrict__ bsm, int ldb, int lane) {
  using Frag = typename OP::Frag;
  static_assert(KSTEPS % PF == 0, "KSTEPS must be a multiple of the prefetch depth");
  const int c32 = lane & 31, hh = lane >> 5;
  const typename OP::E* bbase = bsm + c32 * ldb + hh * 8;
#pragma unroll 1
  for (int ks0 = 0; ks0 < KSTEPS; ks0 += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        const Frag b = *reinterpret_cast<const Frag*>(bbase + tt * 32 * ldb + (ks0 + s) * 16);
        if (SWAP) {
          acc[0][tt] = OP::mma(b, ws.ring[s][0], acc[0][tt]);
          acc[1][tt] = OP::mma(b, ws.ring[s][1], acc[1][tt]);
        } else {
          acc[0][tt] = OP::mma(ws.ring[s][0], b, acc[0][tt]);
          acc[1][tt] = OP::mma(ws.ring[s][1], b, acc[1][tt]);
        }
      }
      ws.ring[s][0] = ws.p[0];  // refill the slot just consumed: PF k-steps ahead
      ws.ring[s][1] = ws.p[64];
      ws.p += 128;
    }
  }
}

template <int NTT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NTT]) {
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ft][tt][r] = 0.f;
}

// LayerNorm (no affine, biased variance, two-pass) over the 256 features of every token of the tile,
// followed by y*(1+scale)+shift, written as OP::E into dst[token][feature].
// v holds this wave's 64 features x TM tokens in accumulator layout.
template <typename OP, int NTT>
__device__ __forceinline__ void ln_modulate_store(const f32x16 (&v)[2][NTT], const float* const (&mrow)[NTT],
                                                  int scale_off, int shift_off, float eps, float* red,
                                                  typename OP::E* dst, int ldd, int wave, int lane) {
  constexpr int TM = 32 * NTT;
  const int c32 = lane & 31, hh = lane >> 5;
  float* red_a = red;
  float* red_b = red + 4 * TM;
  float mean[NTT], rstd[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += v[ft][tt][r];
    s = xor32_sum(s);
    if (hh == 0) red_a[wave * TM + tt * 32 + c32] = s;
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const int t = tt * 32 + c32;
    mean[tt] = (red_a[t] + red_a[TM + t] + red_a[2 * TM + t] + red_a[3 * TM + t]) * (1.0f / kD);
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = v[ft][tt][r] - mean[tt];
        s += d * d;
      }
    s = xor32_sum(s);
    if (hh == 0) red_b[wave * TM + t] = s;
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const int t = tt * 32 + c32;
    const float var = (red_b[t] + red_b[TM + t] + red_b[2 * TM + t] + red_b[3 * TM + t]) * (1.0f / kD);
    rstd[tt] = __builtin_amdgcn_rsqf(var + eps);
  }
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = wave * 64 + ft * 32 + q * 8 + hh * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(mrow[tt] + scale_off + f);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(mrow[tt] + shift_off + f);
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          y[i] = (v[ft][tt][q * 4 + i] - mean[tt]) * rstd[tt] * (1.0f + sc[i]) + sh[i];
        *reinterpret_cast<typename OP::Quad*>(dst + (tt * 32 + c32) * ldd + f) = OP::pack4(y[0], y[1], y[2], y[3]);
      }
    }
  }
}

template <typename OP, int NTT>
__global__ __launch_bounds__(256, (OP::kIsBF16 && NTT <= 2) ? 2 : 1) void dit_block_kernel(const BlockArgs a) {
  using L = BlockLayout<OP, NTT>;
  using E = typename OP::E;
  using Frag = typename OP::Frag;
  using Quad = typename OP::Quad;
  constexpr int TM = L::TM;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  E* XA = reinterpret_cast<E*>(smem);
  E* HB = reinterpret_cast<E*>(smem + L::XA_BYTES);
  float* RED = reinterpret_cast<float*>(smem + L::XA_BYTES + L::HB_BYTES);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c32 = lane & 31, hh = lane >> 5;
  const int tok0 = blockIdx.x * TM;

  // per-token-tile bookkeeping.  The residual buffer is padded (and zero-filled by input_proj) to a whole
  // number of 128-token tiles, so loads/stores need no predication; padded samples reuse the last adaLN row.
  const float* mrow[NTT];
  size_t xoff[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const int tok = tok0 + tt * 32 + c32;
    const int smp = min(tok >> 4, a.n_fwd - 1);
    mrow[tt] = a.mod + (size_t)a.row_index[smp] * a.mod_stride + a.mod_offset;
    xoff[tt] = (size_t)tok * kD + wave * 64 + hh * 4;
  }

  f32x16 acc[2][NTT];

  // start the weight stream now: the first PF units fly while LN1 runs
  constexpr int PF = Prefetch<OP, NTT>::PF;
  WStream<OP, PF> ws;
  ws.init(reinterpret_cast<const Frag*>(a.w_stream) + (size_t)wave * units_per_wave(a.n_chunks) * 128 + lane);

  // ---- phase 0: load x (accumulator layout), LN1 + modulate(a0 = scale, a1 = shift) -> XA ----
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.x + xoff[tt] + ft * 32 + q * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ft][tt][q * 4 + i] = t4[i];
      }
  ln_modulate_store<OP, NTT>(acc, mrow, 0 * kD, 1 * kD, a.eps, RED, XA, L::XA_LD, wave, lane);
  __syncthreads();

  // ---- phases 1+2: attention for heads 2w, 2w+1, entirely in registers ----
  //   Q pass, K pass -> S^T = K Q^T -> softmax -> P  (Q, K fragments die here: keeps the live MFMA
  //   operand set inside the 256 architectural VGPRs) -> V pass -> O^T = V^T P^T -> AO (LDS).
  E* AO = HB;  // attention output aliases the (still unused) hidden double buffer
  {
    const int sp = c32 >> 4;            // which of the tile's two samples this lane's query token belongs to
    Frag Pf[2][NTT][2];
    {
      Frag QF[2][NTT][2], KF[2][NTT][2];
      // Q^T (feature x token)
      zero_acc<NTT>(acc);
      gemm_pass<OP, NTT, 16, false, PF>(acc, ws, XA, L::XA_LD, lane);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        float bq[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b_qkv + 0 * kD + wave * 64 + ft * 32 + q * 8 + hh * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) bq[q * 4 + i] = b4[i];
        }
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          float t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = acc[ft][tt][r] + bq[r];
          QF[ft][tt][0] = OP::pack8(t);
          QF[ft][tt][1] = OP::pack8(t + 8);
        }
      }
      // K^T (feature x token)
      zero_acc<NTT>(acc);
      gemm_pass<OP, NTT, 16, false, PF>(acc, ws, XA, L::XA_LD, lane);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        float bk[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b_qkv + 1 * kD + wave * 64 + ft * 32 + q * 8 + hh * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) bk[q * 4 + i] = b4[i];
        }
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          float t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = acc[ft][tt][r] + bk[r];
          KF[ft][tt][0] = OP::pack8(t);
          KF[ft][tt][1] = OP::pack8(t + 8);
        }
      }
      // scores + softmax
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          f32x16 st;
#pragma unroll
          for (int r = 0; r < 16; ++r) st[r] = 0.f;
          st = OP::mma(KF[ft][tt][0], QF[ft][tt][0], st);  // S^T[key][query], k = head dims
          st = OP::mma(KF[ft][tt][1], QF[ft][tt][1], st);
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
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            sv[i] = __builtin_amdgcn_exp2f((sv[i] - m) * a.attn_scale_log2e);
            sum += sv[i];
          }
          sum = xor32_sum(sum);
          const float inv = __builtin_amdgcn_rcpf(sum);
          float p[16];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float pv = sv[i] * inv;
            p[i] = sp ? 0.f : pv;       // cross-sample blocks of the shared 32x32 tile are exactly zero
            p[8 + i] = sp ? pv : 0.f;
          }
          Pf[ft][tt][0] = OP::pack8(p);
          Pf[ft][tt][1] = OP::pack8(p + 8);
        }
      }
    }
    // V (swapped operands: lane = feature, registers = tokens), then O^T = V^T P^T
    zero_acc<NTT>(acc);
    gemm_pass<OP, NTT, 16, true, PF>(acc, ws, XA, L::XA_LD, lane);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      const float bv = a.b_qkv[2 * kD + wave * 64 + ft * 32 + c32];
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = acc[ft][tt][r] + bv;
        const Frag v0 = OP::pack8(t), v1 = OP::pack8(t + 8);
        f32x16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
        ot = OP::mma(v0, Pf[ft][tt][0], ot);  // O^T[d][query], k = keys
        ot = OP::mma(v1, Pf[ft][tt][1], ot);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<Quad*>(AO + (tt * 32 + c32) * L::XA_LD + wave * 64 + ft * 32 + q * 8 + hh * 4) =
              OP::pack4(ot[q * 4 + 0], ot[q * 4 + 1], ot[q * 4 + 2], ot[q * 4 + 3]);
      }
    }
  }
  __syncthreads();  // AO complete; every wave is also done reading XA

  // ---- phase 3: attention projection, gated residual (a2), LN2 + modulate(a3 = scale, a4 = shift) -> XA ----
  zero_acc<NTT>(acc);
  gemm_pass<OP, NTT, 16, false, PF>(acc, ws, AO, L::XA_LD, lane);
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = wave * 64 + ft * 32 + q * 8 + hh * 4;
        const f32x4 bp = *reinterpret_cast<const f32x4*>(a.b_proj + f);
        const f32x4 g = *reinterpret_cast<const f32x4*>(mrow[tt] + 2 * kD + f);
        f32x4 xr = *reinterpret_cast<const f32x4*>(a.x + xoff[tt] + ft * 32 + q * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xr[i] += g[i] * (acc[ft][tt][q * 4 + i] + bp[i]);
          acc[ft][tt][q * 4 + i] = xr[i];
        }
        *reinterpret_cast<f32x4*>(a.x + xoff[tt] + ft * 32 + q * 8) = xr;
      }
  ln_modulate_store<OP, NTT>(acc, mrow, 3 * kD, 4 * kD, a.eps, RED, XA, L::XA_LD, wave, lane);
  __syncthreads();  // XA (MLP input) complete; AO no longer needed -> HB free

  // ---- phase 4: SwiGLU MLP, hidden processed in chunks of 128 through the HB double buffer ----
  f32x16 accp[2][NTT];
  zero_acc<NTT>(accp);
  for (int c = 0; c < a.n_chunks; ++c) {
    E* hb = HB + (c & 1) * (TM * L::HB_LD);
    zero_acc<NTT>(acc);
    gemm_pass<OP, NTT, 16, false, PF>(acc, ws, XA, L::XA_LD, lane);
    // rows 0-15 of each weight tile are w1, rows 16-31 the matching w2 rows => registers r and r+8 pair up
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float h[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) h[i] = silu_f(acc[ft][tt][q * 4 + i]) * acc[ft][tt][8 + q * 4 + i];
          *reinterpret_cast<Quad*>(hb + (tt * 32 + c32) * L::HB_LD + wave * 32 + ft * 16 + q * 8 + hh * 4) =
              OP::pack4(h[0], h[1], h[2], h[3]);
        }
    __syncthreads();
    gemm_pass<OP, NTT, 8, false, PF>(accp, ws, hb, L::HB_LD, lane);
  }

  // ---- phase 5: gated residual (a5) ----
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = wave * 64 + ft * 32 + q * 8 + hh * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(mrow[tt] + 5 * kD + f);
        f32x4 xr = *reinterpret_cast<const f32x4*>(a.x + xoff[tt] + ft * 32 + q * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[i] += g[i] * accp[ft][tt][q * 4 + i];
        *reinterpret_cast<f32x4*>(a.x + xoff[tt] + ft * 32 + q * 8) = xr;
      }
}

}  // namespace scldm
