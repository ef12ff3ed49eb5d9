// Fused training path of the base DiT shape (see train_fused.hpp): REC forward launches, the fused backward layer
// (dit_backward.hpp) and the batched bf16 weight-gradient GEMMs.  gfx950 only.
#include "train_fused.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "api_common.hpp"
// the fused backward layer, once per 16-bit operand policy (see the header of dit_backward.hpp)
#define SCLDM_BWD_NS bwd
#define SCLDM_BWD_OP OpBF16
#include "dit_backward.hpp"
#undef SCLDM_BWD_NS
#undef SCLDM_BWD_OP
#define SCLDM_BWD_NS bwdh
#define SCLDM_BWD_OP OpFP16
#include "dit_backward.hpp"
#undef SCLDM_BWD_NS
#undef SCLDM_BWD_OP
// ... and on 32-token tiles (batches of at most 512 cells)
#define SCLDM_BWD_NTT 1
#define SCLDM_BWD_NS bwd32
#define SCLDM_BWD_OP OpBF16
#include "dit_backward.hpp"
#undef SCLDM_BWD_NS
#undef SCLDM_BWD_OP
#define SCLDM_BWD_NS bwdh32
#define SCLDM_BWD_OP OpFP16
#include "dit_backward.hpp"
#undef SCLDM_BWD_NS
#undef SCLDM_BWD_OP
#undef SCLDM_BWD_NTT
#include "dit_forward.hpp"

namespace scldm {
namespace fused {

namespace {

constexpr int kSplits = 16;         // most token-axis splits of the weight-gradient GEMMs the partial buffer is sized for
constexpr int kSplitsDefault = 8;   // measured per layer at 1 024 cells: 4 splits 75 us, 8: 57 us, 16: 64 us + a longer reduction (SCLDM_WGRAD_SPLITS)
constexpr int kHP = kBwdChunks * kBwdChunk;   // padded hidden width of the operand arrays (768)
constexpr int kOverlapTiles = 160;            // <= this many 64-token tiles (640 cells): weight gradients of layer l beside the backward of layer l - 1

struct Carver {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += align256(count * sizeof(T));
    return p;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// plain [token][256] fp32 <-> the forward kernel's tile layout [tile][wave 0..3][quad j 0..15][lane][4],
// quad j = (tt*2 + ft)*4 + q holds features wave*64 + ft*32 + q*8 + (lane>>5)*4 .. +3 of token tile*64 + tt*32 + (lane&31)
// ---------------------------------------------------------------------------------------------------------------
template <bool TO_TILE>
__global__ void relayout_kernel(const float* __restrict__ src, float* __restrict__ dst, long quads, long real_tokens) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= quads) return;
  const int lane = (int)(i & 63), j = (int)((i >> 6) & 15), wave = (int)((i >> 10) & 3);
  const long tile = i >> 12;
  const int tt = j >> 3, ft = (j >> 2) & 1, q = j & 3;
  const long tok = tile * 64 + tt * 32 + (lane & 31);
  const int f = wave * 64 + ft * 32 + q * 8 + (lane >> 5) * 4;
  // tokens past the real batch (tile padding): zeros into the tile layout, nothing out of it
  if (TO_TILE) *reinterpret_cast<f32x4*>(dst + i * 4) = tok < real_tokens ? *reinterpret_cast<const f32x4*>(src + tok * kD + f) : f32x4{0.f, 0.f, 0.f, 0.f};
  else if (tok < real_tokens) *reinterpret_cast<f32x4*>(dst + tok * kD + f) = *reinterpret_cast<const f32x4*>(src + i * 4);
}

// ---------------------------------------------------------------------------------------------------------------
// Batched weight-gradient GEMM: C_j[m][n] = sum_t A_j[t][m] * B_j[t][n] for the five products of a layer in one launch.
// Operands are bf16 [token][ld] (what dit_backward_kernel emits); a workgroup owns one 128x128 output tile of one job and one
// of kSplits token ranges, stages 32 tokens at a time ([feature][token] LDS images, transposed in registers on the way:
// each thread gathers 8 tokens of its two features), and writes its partial tile to part[job][split][M][N].
// Optional row sums of A (bias gradients) ride along in the workgroups of the first column of tiles.
// ---------------------------------------------------------------------------------------------------------------
struct WgradJob {
  const __bf16* A; int lda;
  const __bf16* B; int ldb;
  int M, N;
  int tile0, tiles_n;     // first linear tile of this job, tiles along N
  long part_off;          // floats: partials [split][M][N]
  long rs_off;            // floats: row-sum partials [split][M] (-1: none)
};
struct WgradArgs {
  WgradJob job[5];
  int n_jobs, T, kchunk, splits;
  float* part;
};
#ifndef SCLDM_WGRAD_STAGES
#define SCLDM_WGRAD_STAGES 2
#endif
constexpr int kWStages = SCLDM_WGRAD_STAGES;   // register stages in flight per operand (2 or 3)
static_assert(kWStages == 2 || kWStages == 3, "two or three register stages");
constexpr int kWK = 64;            // tokens per staged tile
constexpr int kWLD = 128 + 32;     // bf16 elements per LDS row of a [token][feature] image (320 B: the 8-byte pieces of the transposing
                                   // fragment reads of a 32-lane half - 2 groups x 4 token rows - fall on eight distinct 32-byte bank ranges)

// 64 tokens x 128 features of a [token][ld] bf16 operand: 16-byte buffer loads (row offset on the scalar unit), stored to LDS as
// they are; the k-major MFMA fragments come out of ds_read_b64_tr_b16, which hands lane i of a 16-lane group column i of the
// 4 x 16 block whose 8-byte pieces the group's lanes address (verified on the part: lane i supplies row i/4, columns 4(i%4)..+3).
// The first version gathered 4-byte pieces per token and transposed in registers: 64 VMEM + 32 permute instructions per wave
// and stage against 16 + 0 here (timing proxies: the loads alone were 16 of its 55 us).
struct OperandLoader {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 d[4];   // pass p: token 16 p + 4 wave + lane / 16, features 8 (lane % 16) .. +7
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int ld, int m0, int t0) {
    const int lane = threadIdx.x & 63;
    const unsigned voff = (unsigned)(lane >> 4) * (unsigned)ld * 2u + (unsigned)(lane & 15) * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int p = 0; p < 4; ++p)
      d[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((unsigned)(t0 + 16 * p + 4 * wave) * (unsigned)ld + (unsigned)m0) * 2u, 0);
  }
  // rs[e] += the stage's values of feature 8 (lane % 16) + e (this thread's four tokens); F16: the operands are fp16, else bf16
  template <bool F16>
  __device__ __forceinline__ void add_rowsum(float (&rs)[8]) const {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (F16) {
          typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
          union { unsigned u; f16x2 h; } c;
          c.u = d[p][i];
          rs[2 * i] += (float)c.h[0];
          rs[2 * i + 1] += (float)c.h[1];
        } else {
          rs[2 * i] += __uint_as_float(d[p][i] << 16);
          rs[2 * i + 1] += __uint_as_float(d[p][i] & 0xffff0000u);
        }
      }
  }
  __device__ __forceinline__ void store(__bf16* __restrict__ S) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(S + (16 * p + 4 * wave + (lane >> 4)) * kWLD + (lane & 15) * 8) = d[p];
  }
};
// fragment of the 32 features [f0, f0 + 32) (MFMA rows / columns) over the 16 tokens [kk, kk + 16) of a [token][feature] image
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* __restrict__ S, int f0, int kk, int lane) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  const int g = lane >> 4, i = lane & 15;
  const __bf16* p = S + (kk + 8 * (g >> 1) + (i >> 2)) * kWLD + f0 + 16 * (g & 1) + 4 * (i & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * kWLD));
  union { s16x4 s[2]; bf16x8 f; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.f;
}

constexpr int kWgradLds = 2 * 2 * kWK * kWLD * 2;   // two buffers x two operands = 80 KB: two workgroups per CU
template <bool F16 = false>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) char wgrad_smem[];
  auto As = [&](int b) { return reinterpret_cast<__bf16*>(wgrad_smem) + b * (kWK * kWLD); };
  auto Bs = [&](int b) { return reinterpret_cast<__bf16*>(wgrad_smem) + (2 + b) * (kWK * kWLD); };
  // XCD-aware numbering: workgroups are dealt to the 8 XCDs round-robin by linear id, so with id = tile * splits + split every
  // XCD gets ONE token range (for 8 splits) of all tiles: the tiles that share operand rows re-read them from that XCD's L2
  // instead of 2-6 times from the fabric (436 MB of tile loads per launch against 139 MB of operands).
  const int z = blockIdx.x % g.splits, tile_id = blockIdx.x / g.splits;
  int ji = 0;
#pragma unroll
  for (int k = 1; k < 5; ++k)
    if (k < g.n_jobs && tile_id >= g.job[k].tile0) ji = k;
  const WgradJob& j = g.job[ji];
  const int tl = tile_id - j.tile0, tm = tl / j.tiles_n, tn = tl % j.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int t_beg = z * g.kchunk, t_end = min(g.T, t_beg + g.kchunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

  // kWStages register stages per operand: the loads of stage i + kWStages are issued while stage i is multiplied and stage i + 1
  // (requested kWStages - 1 whole iterations earlier) is written to LDS.  Round 5: 42 % of the kernel's wave cycles are vmcnt waits
  // (profiles/r5_pmc_train_sq1.txt), but a third stage (-DSCLDM_WGRAD_STAGES=3) measured +-0 (45.4 against 45.2 us): the loads wait on
  // bandwidth (3.9 TB/s at the fabric with a 66 % L2 hit rate), not on too short a prefetch distance.  Barriers order LDS only
  // (lds_barrier): __syncthreads() would drain the loads in flight.
  OperandLoader la[kWStages], lb[kWStages];
  auto make_rsrc = [](const __bf16* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(j.A), rb = make_rsrc(j.B);
  const bool want_rs = j.rs_off >= 0 && tn == 0;
  float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int n_it = (t_end - t_beg + kWK - 1) / kWK;
#pragma unroll
  for (int q = 0; q < kWStages; ++q) {
    la[q].load(ra, j.lda, m0, t_beg + min(q, n_it - 1) * kWK);
    lb[q].load(rb, j.ldb, n0, t_beg + min(q, n_it - 1) * kWK);
  }
  if (want_rs) la[0].template add_rowsum<F16>(rs);
  la[0].store(As(0));
  lb[0].store(Bs(0));
  lds_barrier();
  auto iteration = [&](int it, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;     // register slot of stage `it` (already in LDS buffer it & 1): free again
    constexpr int NEXT = (SLOT + 1) % kWStages;         // register slot of stage it + 1
    const int buf = it & 1;
    // unconditional (the last iterations re-request the last stage): with the loads under a branch the compiler's waitcnt
    // pass must assume the path that issued none, and then waits for the NEW stage as well when it needs the old one
    const int ahead = t_beg + min(it + kWStages, n_it - 1) * kWK;
    la[SLOT].load(ra, j.lda, m0, ahead);
    lb[SLOT].load(rb, j.ldb, n0, ahead);
#pragma unroll
    for (int kk = 0; kk < kWK; kk += 16) {
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = tr_frag(As(buf), wm * 64 + i * 32, kk, lane);
#pragma unroll
      for (int k = 0; k < 2; ++k) bf[k] = tr_frag(Bs(buf), wn * 64 + k * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if constexpr (F16) {
            union { bf16x8 b; f16x8 h; } ua, ub;   // (the 16-bit pieces travel untyped through the loads and the transposing reads)
            ua.b = af[i];
            ub.b = bf[k];
            acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.h, ub.h, acc[i][k], 0, 0, 0);
          } else {
            acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[k], acc[i][k], 0, 0, 0);
          }
        }
    }
    if (it + 1 < n_it) {
      if (want_rs) la[NEXT].template add_rowsum<F16>(rs);
      la[NEXT].store(As(buf ^ 1));
      lb[NEXT].store(Bs(buf ^ 1));
    }
    lds_barrier();
  };
  for (int it = 0; it < n_it; it += kWStages) {
    iteration(it, std::integral_constant<int, 0>{});
    if (it + 1 < n_it) iteration(it + 1, std::integral_constant<int, 1>{});
    if constexpr (kWStages > 2)
      if (it + 2 < n_it) iteration(it + 2, std::integral_constant<int, 2 % kWStages>{});
  }
  if (want_rs) {   // the 16 threads (tid / 16) that share a feature chunk hold partial sums of the same eight rows
    float* red = reinterpret_cast<float*>(wgrad_smem);   // [16][128]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(threadIdx.x >> 4) * 128 + (threadIdx.x & 15) * 8 + e] = rs[e];
    lds_barrier();
    if (threadIdx.x < 128 && m0 + (int)threadIdx.x < j.M) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += red[q * 128 + threadIdx.x];
      g.part[j.rs_off + (long)z * j.M + m0 + threadIdx.x] = t;
    }
  }
  float* __restrict__ C = g.part + j.part_off + (long)z * j.M * j.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int n = n0 + wn * 64 + k * 32 + (lane & 31);
      if (n >= j.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + acc_row(r, lane >> 5);
        if (m < j.M) C[(long)m * j.N + n] = acc[i][k][r];
      }
    }
}

// dst_j[m*ldc + n] = sum_z part_j[z][m][n] for up to seven jobs (five weight gradients, two bias gradients) in one launch
struct ReduceJob {
  long part_off, first;   // first linear element of this job
  float* dst;
  int M, N, ldc;
};
struct ReduceArgs {
  ReduceJob job[7];
  int n_jobs, splits;
  long total;
  const float* part;
};
__global__ void wgrad_reduce_kernel(const ReduceArgs g) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < g.total; i += (long)gridDim.x * blockDim.x) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < 7; ++k)
      if (k < g.n_jobs && i >= g.job[k].first) ji = k;
    const ReduceJob& j = g.job[ji];
    const long e = i - j.first, mn = (long)j.M * j.N;
    float s = 0.f;
    for (int z = 0; z < g.splits; ++z) s += g.part[j.part_off + z * mn + e];
    j.dst[(e / j.N) * j.ldc + (e % j.N)] = s;
  }
}

// The same for up to eight layers in ONE launch (blockIdx.y = layer): at batches whose backward kernel fills the chip the eight
// per-layer reductions (10 us each + a launch gap, in series between backward layers) leave the layer loop - every layer keeps its
// own partial block and they are summed behind the last layer (round 6; opt-in SCLDM_TRAIN_WGRAD_DEFER=1: measured +-0 to slower, below).
struct ReduceAll { ReduceArgs layer[8]; };
__global__ void wgrad_reduce_all_kernel(const ReduceAll all) {
  const ReduceArgs& g = all.layer[blockIdx.y];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < g.total; i += (long)gridDim.x * blockDim.x) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < 7; ++k)
      if (k < g.n_jobs && i >= g.job[k].first) ji = k;
    const ReduceJob& j = g.job[ji];
    const long e = i - j.first, mn = (long)j.M * j.N;
    float s = 0.f;
    for (int z = 0; z < g.splits; ++z) s += g.part[j.part_off + z * mn + e];
    j.dst[(e / j.N) * j.ldc + (e % j.N)] = s;
  }
}
static_assert(sizeof(ReduceAll) <= 4000, "kernel arguments must stay under 4 KB");

__global__ void iota32_kernel(int32_t* __restrict__ ri, int n) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) ri[s] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// The two ends of the backward, each as ONE kernel on the tile layout (they were eight and four small launches: relayouts,
// LayerNorm forward / backward, two skinny GEMMs with K = all tokens and their split-K reductions).
// Shape: a workgroup walks samples b, b + grid, ...; wave w takes tokens 4w .. 4w+3 of the sample, lane l features 4l .. 4l+3.
// The skinny weight gradients (din x 256) are accumulated in registers across the workgroup's samples and leave as one
// partial per wave; edge_reduce_kernel sums the partials in a fixed order.  fp32 arithmetic throughout (din <= 32).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kEdgeMaxDin = 32;
__device__ __forceinline__ size_t tile_addr(long tok, int f) {   // float offset of feature f (multiple of 4) of token tok in the tile layout
  const long tile = tok >> 6;
  const int tt = (int)(tok >> 5) & 1, c32 = (int)tok & 31;
  const int wave = f >> 6, ft = (f >> 5) & 1, q = (f >> 3) & 3, hh = (f >> 2) & 1;
  return ((size_t)((tile * 4 + wave) * 16 + (tt * 2 + ft) * 4 + q) * 64 + c32 + 32 * hh) * 4;
}
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Final layer (layers.py:397-401): out = (LN(x) (1 + scale) + shift) fin_w^T + b.  Given dout: d x (tile layout), the final adaLN
// vectors' gradients, and per-wave partials of d fin_w (din x 256) and d fin_b (din).
template <int DIN>
__global__ __launch_bounds__(256) void final_bwd_kernel(const float* __restrict__ x_last, const float* __restrict__ mod, int mod_stride, int of,
                                                        const float* __restrict__ dout, const float* __restrict__ fin_w, float eps, int n,
                                                        float* __restrict__ dx, float* __restrict__ dmod, float* __restrict__ part) {
  __shared__ f32x4 red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, f = lane * 4;
  f32x4 wq[DIN];
#pragma unroll
  for (int c = 0; c < DIN; ++c) wq[c] = *reinterpret_cast<const f32x4*>(fin_w + (size_t)c * kD + f);
  f32x4 accw[DIN];
#pragma unroll
  for (int c = 0; c < DIN; ++c) accw[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float accb = 0.f;   // lane c < DIN: sum_t dout[t][c]
  for (int b = blockIdx.x; b < n; b += gridDim.x) {
    const f32x4 shift = *reinterpret_cast<const f32x4*>(mod + (size_t)b * mod_stride + of + f);
    f32x4 scale = *reinterpret_cast<const f32x4*>(mod + (size_t)b * mod_stride + of + kD + f);
    scale += 1.0f;
    f32x4 dsc = {0.f, 0.f, 0.f, 0.f}, dsh = {0.f, 0.f, 0.f, 0.f};
    for (int tt = 0; tt < 4; ++tt) {
      const long tok = (long)b * 16 + wave * 4 + tt;
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x_last + tile_addr(tok, f));
      const float mean = wave_sum64((xv[0] + xv[1]) + (xv[2] + xv[3])) * (1.0f / kD);
      f32x4 xh = xv - mean;
      const float rstd = 1.0f / sqrtf(wave_sum64((xh[0] * xh[0] + xh[1] * xh[1]) + (xh[2] * xh[2] + xh[3] * xh[3])) * (1.0f / kD) + eps);
      xh *= rstd;
      const f32x4 hv = xh * scale + shift;
      const float dlane = lane < DIN ? dout[tok * DIN + lane] : 0.f;
      accb += dlane;
      f32x4 dh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < DIN; ++c) {
        const float d = __shfl(dlane, c);
        dh += d * wq[c];
        accw[c] += d * hv;
      }
      dsc += dh * xh;
      dsh += dh;
      const f32x4 g = dh * scale;
      const float s1 = wave_sum64((g[0] + g[1]) + (g[2] + g[3])) * (1.0f / kD);
      const float s2 = wave_sum64((g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3])) * (1.0f / kD);
      *reinterpret_cast<f32x4*>(dx + tile_addr(tok, f)) = rstd * (g - s1 - xh * s2);
    }
    red[0][wave][lane] = dsc;
    red[1][wave][lane] = dsh;
    __syncthreads();
    if (wave < 2) {
      const f32x4 t = (red[wave][0][lane] + red[wave][1][lane]) + (red[wave][2][lane] + red[wave][3][lane]);
      // (shift, scale) = chunks (0, 1) of the final adaLN vector
      *reinterpret_cast<f32x4*>(dmod + (size_t)b * mod_stride + of + (wave == 0 ? kD : 0) + f) = t;
    }
    __syncthreads();
  }
  float* p = part + (size_t)(blockIdx.x * 4 + wave) * (DIN * kD + kEdgeMaxDin);
#pragma unroll
  for (int c = 0; c < DIN; ++c) *reinterpret_cast<f32x4*>(p + c * kD + f) = accw[c];
  if (lane < kEdgeMaxDin) p[DIN * kD + lane] = lane < DIN ? accb : 0.f;
}

// Input projection (nnets.py:290): x0 = x in_w^T + in_b + pos.  Given d x0 (tile layout): per-wave partials of d in_w (256 x din,
// stored [c][f]), d in_b (256) and d pos_embed (16 x 256: wave w owns token positions 4w .. 4w+3).
template <int DIN>
__global__ __launch_bounds__(256) void inproj_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ x, int n, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, f = lane * 4;
  f32x4 accw[DIN], accp[4], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < DIN; ++c) accw[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) accp[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int b = blockIdx.x; b < n; b += gridDim.x) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const long tok = (long)b * 16 + wave * 4 + tt;
      const f32x4 d = *reinterpret_cast<const f32x4*>(dx + tile_addr(tok, f));
      const float xl = lane < DIN ? x[tok * DIN + lane] : 0.f;
      accb += d;
      accp[tt] += d;
#pragma unroll
      for (int c = 0; c < DIN; ++c) accw[c] += __shfl(xl, c) * d;
    }
  }
  // partial row: [DIN][256] weight | [256] bias | [16][256] positions (this wave's four rows, zeros elsewhere are implied by the reducer)
  float* p = part + (size_t)(blockIdx.x * 4 + wave) * ((DIN + 1 + 4) * kD);
#pragma unroll
  for (int c = 0; c < DIN; ++c) *reinterpret_cast<f32x4*>(p + c * kD + f) = accw[c];
  *reinterpret_cast<f32x4*>(p + DIN * kD + f) = accb;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(p + (DIN + 1 + tt) * kD + f) = accp[tt];
}

// out[j] = sum_p part[p * ld + j] (p ascending), 16 columns x 16 row phases per workgroup; `period` > 0: only rows p with
// p % period == phase0 contribute (the per-wave position rows of inproj_bwd_kernel)
// (columns >= split go to out2[c - split]: the final layer's weight and bias gradients leave for their two tensors directly - round 5
// reduced into scratch and issued two device-to-device copies behind it, ~10 us of copy engine + gaps in front of the layer loop)
__global__ __launch_bounds__(256) void edge_reduce_kernel(const float* __restrict__ part, int rows, int ld, int cols, float* __restrict__ out,
                                                          float* __restrict__ out2 = nullptr, int split = 0) {
  __shared__ float red[16][17];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), ph = threadIdx.x >> 4;
  float s = 0.f;
  if (c < cols)
    for (int p = ph; p < rows; p += 16) s += part[(size_t)p * ld + c];
  red[ph][threadIdx.x & 15] = s;
  __syncthreads();
  if (ph == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
    if (out2 && c >= split) out2[c - split] = t;
    else out[c] = t;
  }
}
// d in_w arrives as [c][f] (the kernel's accumulation order); the parameter is (256, din): out[f * din + c]
__global__ void transpose_small_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * cols) dst[(i % cols) * rows + i / cols] = src[i];
}
// d pos_embed[p][f] = sum over workgroups of the partial of wave p / 4, row p % 4
__global__ __launch_bounds__(256) void pos_reduce_kernel(const float* __restrict__ part, int groups, int ld, int off, float* __restrict__ out) {
  const int p = blockIdx.x, f = threadIdx.x;   // p = token position 0..15
  float s = 0.f;
  for (int g = 0; g < groups; ++g) s += part[(size_t)(g * 4 + p / 4) * ld + off + (p % 4) * kD + f];
  out[p * kD + f] = s;
}

constexpr int kMaxScatterLayers = 64;
struct ScatterArgs {
  float* w[kMaxScatterLayers + 1];   // [L] = final layer
  float* b[kMaxScatterLayers + 1];
  int n_layer;
};
__global__ void scatter_ada_kernel(const float* __restrict__ dw_all, const float* __restrict__ db_all, const ScatterArgs a) {
  const int mw = a.n_layer * kModBlock + 2 * kD;
  const long total = (long)mw * (kD + 1);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const bool is_b = i >= (long)mw * kD;
    const long e = is_b ? i - (long)mw * kD : i;
    const int row = (int)(is_b ? e : e / kD);
    const int l = min(row / kModBlock, a.n_layer), r = row - l * kModBlock;
    if (is_b) a.b[l][r] = db_all[e];
    else a.w[l][(long)r * kD + (e % kD)] = dw_all[e];
  }
}



// ---- loss scaling of the fp16 backward -------------------------------------------------------------------------------------------
// State (16 x 32 bit on the handle, persists across steps):
//   [0] S (float, a power of two)   [1] 1 / S   [2] non-finite gradient values of the LAST backward (int)   [3] headroom h <= 0 (int)
//   [4] clean backwards since the last overflow (int)   [5] max |dout| bits of the running reduction (uint)   [6] block ticket (uint)
//   [7] backwards that overflowed so far (int)
// S = 2^(floor(log2(8 / max |dout|)) + h): the nominal scale puts max |S dout| in [8, 16); an intermediate of the backward can still
// leave the fp16 range - unscale_kernel counts the non-finite gradient values, the next backward lowers h by one (and the
// caller's found_inf flag lets the optimizer skip the poisoned step), and h recovers one power of two per kHeadroomRecover clean steps.
constexpr int kLsBlocks = 128, kHeadroomMin = -24, kHeadroomRecover = 2000;
__global__ __launch_bounds__(1024) void loss_scale_kernel(const float* __restrict__ dout, long n, float* __restrict__ state, float* __restrict__ found_inf) {
  __shared__ float red[16];
  __shared__ bool last;
  unsigned* st = reinterpret_cast<unsigned*>(state);
  float m = 0.f;
  for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n; i += (long)gridDim.x * 1024) m = fmaxf(m, fabsf(dout[i]));   // (fmaxf drops NaN, inf is caught below)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    atomicMax(st + 5, __float_as_uint(m));          // non-negative floats order like their bit patterns
    __threadfence();
    last = atomicAdd(st + 6, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    const float mx = __uint_as_float(atomicMax(st + 5, 0u));
    int h = (int)st[3], clean = (int)st[4];
    if ((int)st[2] > 0) {                            // the previous backward produced non-finite gradients
      h = max(h - 1, kHeadroomMin);
      clean = 0;
      st[7] = st[7] + 1;
    } else if (h < 0 && ++clean >= kHeadroomRecover) {
      ++h;
      clean = 0;
    }
    float S = 1.0f;
    if (mx > 0.f && mx < 3.0e38f) S = exp2f(fminf(fmaxf(floorf(log2f(8.0f / mx)) + (float)h, -24.f), 60.f));
    state[0] = S;
    state[1] = 1.0f / S;
    st[2] = 0u;
    st[3] = (unsigned)h;
    st[4] = (unsigned)clean;
    st[5] = 0u;
    st[6] = 0u;
    if (found_inf) *found_inf = 0.f;
  }
}
__global__ void scale_copy_kernel(const float* __restrict__ src, const float* __restrict__ scale, float* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * scale[0];
}
// every gradient tensor of the fused fp16 backward: 9 per layer + the class tables + 11 of the two ends + dx
constexpr int kMaxUnscale = kMaxFp16TrainLayers * 9 + SCLDM_MAX_CLASSES + 12;
struct UnscaleArgs {
  float* p[kMaxUnscale];
  int n[kMaxUnscale];
  int count;
  float* state;
  float* found_inf;
};
__global__ __launch_bounds__(256) void unscale_kernel(const UnscaleArgs a) {
  float* __restrict__ p = a.p[blockIdx.x];
  const int n = a.n[blockIdx.x];
  const float inv = a.state[1];
  int bad = 0;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) {
    const float v = p[i] * inv;
    p[i] = v;
    bad += !(fabsf(v) <= 3.0e38f);      // inf or NaN
  }
  const unsigned long long any = __ballot(bad != 0);
  if (any) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(reinterpret_cast<int*>(a.state) + 2, bad);
      if (a.found_inf) *a.found_inf = 1.0f;
    }
  }
}

}  // namespace

static inline int pad4(int n) { return (n + 3) / 4 * 4; }   // whole 64-token tiles; samples past n are tile padding

Record carve_record(const scldm_dit* h, int n, void* base) {
  const size_t TD = (size_t)pad4(n) * 16 * kD, L = h->cfg.n_layer;
  Carver c{reinterpret_cast<char*>(base)};
  Record r;
  r.x = c.take<float>((L + 1) * TD);
  r.y1 = c.take<__bf16>(L * TD);
  r.y2 = c.take<__bf16>(L * TD);
  r.bytes = c.off;
  return r;
}

size_t edge_part_floats(const scldm_dit* h);
static size_t part_floats(const scldm_dit* h) {
  const size_t H = h->cfg.hidden_dim;
  return (size_t)kSplits * (3 * kD * kD + kD * kD + 3 * H * kD + 3 * kD + kD);
}

Scratch carve_scratch(const scldm_dit* h, int n, void* base) {
  const size_t T = (size_t)pad4(n) * 16;
  Carver c{reinterpret_cast<char*>(base)};
  Scratch s;
  s.handoff = c.take<float>(T * kD);
  s.dx = c.take<float>(T * kD);
  s.ridx = c.take<int32_t>(n);
  const size_t set_off0 = c.off;
  s.e_h1 = c.take<__bf16>(T * kD);
  s.e_dqkv = c.take<__bf16>(T * 3 * kD);
  s.e_ao = c.take<__bf16>(T * kD);
  s.e_dy1 = c.take<__bf16>(T * kD);
  s.e_h2 = c.take<__bf16>(T * kD);
  s.e_da = c.take<__bf16>(T * kHP);
  s.e_db = c.take<__bf16>(T * kHP);
  s.e_hid = c.take<__bf16>(T * kHP);
  s.e_dy2 = c.take<__bf16>(T * kD);
  // a second set of the nine arrays for batches whose backward kernel leaves CUs idle: layer l's weight gradients then run on a side
  // stream beside layer l - 1's backward kernel (backward_layers_t)
  s.e_set_elems = (c.off - set_off0) / sizeof(__bf16);
  s.e_set1 = pad4(n) / 4 <= kOverlapTiles ? c.take<__bf16>(s.e_set_elems) : nullptr;
  s.part_floats = part_floats(h);
  // one partial block per layer where the per-layer reductions are deferred to the end of the layer loop (no second operand-pair set:
  // the backward kernel fills the chip), one block otherwise
  static const bool defer_on = [] { const char* e = getenv("SCLDM_TRAIN_WGRAD_DEFER"); return e && e[0] == '1'; }();
  s.part_layers = (s.e_set1 || !defer_on) ? 1 : h->cfg.n_layer;
  // (+ one more block when deferring: the backward's tail borrows a block for the split-K partials of d t_w2 while the deferred
  // reduction is still reading the layers' blocks on its side stream)
  s.part = c.take<float>(s.part_floats * (size_t)(s.part_layers + (s.part_layers > 1 ? 1 : 0)));
  s.ada_dw = c.take<float>((size_t)h->mod_w * (kD + 1));
  s.edge_part = c.take<float>(edge_part_floats(h));
  s.dout_s = c.take<float>(T * 32);
  s.scale = h->d_ls;   // (loss-scale state: on the handle, it persists across steps)
  s.bytes = c.off;
  return s;
}

bool eligible(const scldm_dit* h, int n, int precision) {
  // h->train_fused: SCLDM_TRAIN_FUSED=0 at handle creation keeps the base shape on the generic GEMM-based path (A/B runs, tests)
  // precision: bf16, or fp16 (the reference's TF32 mantissa; its backward is loss-scaled by the caller and un-scaled through a by-value
  // pointer table, hence the layer bound)
  if (precision != SCLDM_PREC_BF16 && precision != SCLDM_PREC_FP16) return false;
  if (precision == SCLDM_PREC_FP16 && (h == nullptr || h->cfg.n_layer > kMaxFp16TrainLayers)) return false;
  return h && h->train_fused && h->bwd_stream && h->fused && n >= 1 && n <= 65536 /* operand arrays stay under the 2 GB a buffer descriptor addresses */ && h->stream[precision][1] != nullptr &&
         h->cfg.hidden_dim <= kHP && h->cfg.hidden_dim % 2 == 0 && h->lpl >= 1 && h->cfg.n_layer <= kMaxScatterLayers;
}

// side streams of the training step (experiment switch SCLDM_TRAIN_SIDE_PRIO=low: lowest HIP priority, so that what is queued on the
// caller's stream - the step's critical path - wins the dispatcher when both have workgroups ready)
static int make_side_stream(hipStream_t* s) {
  static const bool low = [] { const char* e = getenv("SCLDM_TRAIN_SIDE_PRIO"); return e && e[0] == 'l'; }();
  if (low) {
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(s, hipStreamNonBlocking, least));
  } else {
    HIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
  }
  return SCLDM_OK;
}
// everything of a step that is not a kernel launch or an event: the pack-job tables of the live parameters, the side streams
// and their events (scldm_dit_train_prepare calls it ahead of the first step; prepare() re-checks it per step for free)
int prepare_tables(scldm_dit* h, const scldm_dit_weights* w, hipStream_t st) {
  int rc = scldm_build_pack_tables(h, w, st);   // (the backward stream itself is allocated with the handle)
  if (rc) return rc;
  for (int k = 0; k < 3; ++k)
    if (!h->side[k]) {
      { const int rc_s = make_side_stream(&h->side[k]); if (rc_s) return rc_s; }
      HIP_TRY(hipEventCreateWithFlags(&h->join_ev[k], hipEventDisableTiming));
    }
  if (!h->fork_ev) HIP_TRY(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
  if (!h->bwd_pack_ev) HIP_TRY(hipEventCreateWithFlags(&h->bwd_pack_ev, hipEventDisableTiming));
  return SCLDM_OK;
}
int prepare(scldm_dit* h, const scldm_dit_weights* w, hipStream_t st, int precision) {
  int rc = prepare_tables(h, w, st);
  if (rc) return rc;
  // The re-pack (90 us, memory bound) runs on a side stream next to the conditioning MLP, which reads the live parameters:
  // forked here, joined by prepare_join() before the first consumer of a packed copy.
  HIP_TRY(hipEventRecord(h->fork_ev, st));            // everything queued so far (the previous step's optimizer update) comes first
  HIP_TRY(hipStreamWaitEvent(h->side[0], h->fork_ev, 0));
  // Two launches over the same job table: what the FORWARD reads first (its weight stream, biases, stacked adaLN copies - the main
  // stream waits for these in prepare_join()), then the backward weight stream, which runs beside the adaLN GEMM and the recording
  // forward and is awaited by the first backward layer (backward_join).  Measured +-0 (2.46 ms per step either way at 1 024 cells: the
  // conditioning chain on the main stream takes as long as the whole pack), so the default stays ONE launch; SCLDM_TRAIN_PACK_SPLIT=1.
  // bit 9: the backward stream is packed as fp16 (same buffer: it is re-packed every step, in the step's operand type)
  static const bool split = [] { const char* e = getenv("SCLDM_TRAIN_PACK_SPLIT"); return e && e[0] == '1'; }();
  const unsigned bwd_bits = 0x100u | (precision == SCLDM_PREC_FP16 ? 0x200u : 0u);
  rc = scldm_run_pack(h, true, (1u << precision) | (split ? 0u : bwd_bits), h->side[0]);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(h->join_ev[0], h->side[0]));
  if (split && (rc = scldm_run_pack(h, true, bwd_bits | 0x400u, h->side[0]))) return rc;
  HIP_TRY(hipEventRecord(h->bwd_pack_ev, h->side[0]));
  return SCLDM_OK;
}
// side stream k (created on first use) ordered after everything queued on `st` so far / `st` ordered after side stream k
int fork_side(scldm_dit* h, hipStream_t st, int k, hipStream_t* out) {
  if (!h->side[k]) {
    { const int rc_s = make_side_stream(&h->side[k]); if (rc_s) return rc_s; }
    HIP_TRY(hipEventCreateWithFlags(&h->join_ev[k], hipEventDisableTiming));
  }
  if (!h->fork_ev) HIP_TRY(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(h->fork_ev, st));
  HIP_TRY(hipStreamWaitEvent(h->side[k], h->fork_ev, 0));
  *out = h->side[k];
  return SCLDM_OK;
}
int join_side(scldm_dit* h, hipStream_t st, int k) {
  HIP_TRY(hipEventRecord(h->join_ev[k], h->side[k]));
  HIP_TRY(hipStreamWaitEvent(st, h->join_ev[k], 0));
  return SCLDM_OK;
}
int prepare_join(scldm_dit* h, hipStream_t st) {
  HIP_TRY(hipStreamWaitEvent(st, h->join_ev[0], 0));
  return SCLDM_OK;
}
int backward_join(scldm_dit* h, hipStream_t st) {   // the backward weight stream of this step is packed
  if (h->bwd_pack_ev) HIP_TRY(hipStreamWaitEvent(st, h->bwd_pack_ev, 0));
  return SCLDM_OK;
}

int forward(scldm_dit* h, const float* x, const float* mod, int n, float* out, const Record& rec, const Scratch& s, hipStream_t st, int precision) {
  using L = FwdLayout<OpBF16, 2, 2>;   // (the fp16 policy has the same layout)
  using L1 = FwdLayout<OpBF16, 1, 2>;
  static_assert(FwdLayout<OpFP16, 2, 2>::LDS_BYTES == L::LDS_BYTES && FwdLayout<OpFP16, 2, 2>::NT == L::NT, "fp16 = the bf16 kernel's shape");
  static_assert(FwdLayout<OpFP16, 1, 2>::LDS_BYTES == L1::LDS_BYTES && L1::NT == L::NT, "fp16 = the bf16 kernel's shape");
  const bool f16 = precision == SCLDM_PREC_FP16;
  // Small batches: while 32-token tiles still get a CU each, the 32-token-tile instantiation's walk is a quarter shorter (api.hip,
  // trunk()); it writes the same record (64-token-tile geometry) and the same bits.  SCLDM_TRAIN_SMALL_NTT=0: off.
  static const bool small_ok = [] { const char* e = getenv("SCLDM_TRAIN_SMALL_NTT"); return !(e && e[0] == '0'); }();
  const int tiles64 = pad4(n) / 4;
  const bool small = small_ok && tiles64 <= 128;
  void (*kern)(const FwdArgs) = small ? (f16 ? dit_forward_kernel<OpFP16, 1, 2, true> : dit_forward_kernel<OpBF16, 1, 2, true>)
                                      : (f16 ? dit_forward_kernel<OpFP16, 2, 2, true> : dit_forward_kernel<OpBF16, 2, 2, true>);
  const int lds_bytes = small ? L1::LDS_BYTES : L::LDS_BYTES;
  static bool attr_set[2][2][64] = {};
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[small][f16][dev]) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    if (dev >= 0 && dev < 64) attr_set[small][f16][dev] = true;
  }
  const scldm_dit_config& c = h->cfg;
  if (h->iota_n < n) {   // identity row index, kept on the handle (grown on demand)
    HIP_TRY(hipStreamSynchronize(st));
    if (h->iota) (void)hipFree(h->iota);
    h->iota = nullptr;
    h->iota_n = 0;
    const int cap = std::max(n, 4096);
    HIP_TRY(hipMalloc((void**)&h->iota, (size_t)cap * sizeof(int32_t)));
    iota32_kernel<<<cdiv(cap, 256), 256, 0, st>>>(h->iota, cap);
    LAUNCH_CHECK();
    h->iota_n = cap;
  }
  const size_t layer_elems = (size_t)4 * units_per_layer(h->n_chunks[1], h->half[1]) * 1024;
  FwdArgs a{};
  a.z = x;
  a.out = out;
  a.x = s.handoff;
  a.mod = mod;
  a.row_index = h->iota;
  a.w_final = h->wfinal[precision];
  a.in_wt = h->in_wt;
  a.in_w = h->in_w;
  a.in_b = h->in_b;
  a.pos = h->pos;
  a.fin_b = h->fin_b;
  a.n_fwd = n;
  a.n_direct = n;
  a.rep = 1;
  a.din = c.n_embed_input;
  a.n_layer = c.n_layer;
  a.n_chunks = h->n_chunks[1];
  a.half_chunk = h->half[1];
  a.mod_stride = h->mod_w;
  a.eps = c.layernorm_eps;
  a.attn_scale_log2e = 1.4426950408889634f / sqrtf(32.0f);
  a.w_layer_elems = (long)layer_elems;
  a.rec_x = rec.x;
  a.rec_y1 = rec.y1;
  a.rec_y2 = rec.y2;
  a.rec_stride = (long)pad4(n) * 16 * kD;
  // (the kernel clamps the samples past n to the last real one and never stores their output; both halves of the last 64-token tile
  // are launched, so every record row the backward reads is written)
  const int tiles = small ? 2 * tiles64 : tiles64;
  const int lpl_rec = std::min(h->lpl, kMaxLayersPerLaunchRec);   // the recording instantiation has four layer slots
  for (int i = 0; i < c.n_layer; i += lpl_rec) {
    a.layer = i;
    a.n_here = std::min(lpl_rec, c.n_layer - i);
    a.w_stream = (const char*)h->stream[precision][1] + (size_t)i * layer_elems * 2;
    a.b_qkv = h->b_qkv + (size_t)i * 768;
    a.b_proj = h->b_proj + (size_t)i * 256;
    kern<<<tiles, L::NT, lds_bytes, st>>>(a);
    LAUNCH_CHECK();
  }
  return SCLDM_OK;
}

int to_tile(const float* plain, float* tile, int n, hipStream_t st) {
  const long quads = (long)pad4(n) * 16 * kD / 4;
  relayout_kernel<true><<<cdiv(quads, 256), 256, 0, st>>>(plain, tile, quads, (long)n * 16);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
int to_plain(const float* tile, float* plain, int n, hipStream_t st) {
  const long quads = (long)pad4(n) * 16 * kD / 4;
  relayout_kernel<false><<<cdiv(quads, 256), 256, 0, st>>>(tile, plain, quads, (long)n * 16);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

template <int DIN>
static int final_backward_t(scldm_dit* h, const float* x_last, const float* mod, const float* dout, const float* fin_w, int n, float* dx,
                            float* dmod, float* gw, float* gb, float* part, hipStream_t st) {
  const int of = h->cfg.n_layer * kModBlock, groups = std::min(n, 256), rows = groups * 4, ld = DIN * kD + kEdgeMaxDin;
  if (n % 4)   // ragged batch: the padding samples of the last 64-token tile must enter the layers with a zero gradient
    HIP_TRY(hipMemsetAsync(dx + (size_t)(n / 4) * 64 * kD, 0, (size_t)64 * kD * sizeof(float), st));
  final_bwd_kernel<DIN><<<groups, 256, 0, st>>>(x_last, mod, h->mod_w, of, dout, fin_w, h->cfg.layernorm_eps, n, dx, dmod, part);
  LAUNCH_CHECK();
  // The reduction of the per-wave partials (d fin_w, d fin_b) feeds nothing in the layer loop: it runs on side stream 0 beside the
  // first backward layer (round 6; it was 20 us + two copies on the critical path); the caller joins side 0 after the layers.
  hipStream_t ss = st;
  const int rc = fork_side(h, st, 0, &ss);
  if (rc) return rc;
  edge_reduce_kernel<<<cdiv(DIN * kD + DIN, 16), 256, 0, ss>>>(part, rows, ld, DIN * kD + DIN, gw, gb, DIN * kD);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
int final_backward(scldm_dit* h, const float* x_last, const float* mod, const float* dout, const float* fin_w, int n, float* dx,
                   float* dmod, float* gw, float* gb, float* part, hipStream_t st) {
  switch (h->cfg.n_embed_input) {
    case 16: return final_backward_t<16>(h, x_last, mod, dout, fin_w, n, dx, dmod, gw, gb, part, st);
    case 32: return final_backward_t<32>(h, x_last, mod, dout, fin_w, n, dx, dmod, gw, gb, part, st);
    case 8: return final_backward_t<8>(h, x_last, mod, dout, fin_w, n, dx, dmod, gw, gb, part, st);
    default: return fail(SCLDM_ERR_SHAPE, "fused edge kernels are instantiated for n_embed_input 8 / 16 / 32");
  }
}
bool edge_kernels_available(const scldm_dit* h) { const int d = h->cfg.n_embed_input; return d == 8 || d == 16 || d == 32; }
size_t edge_part_floats(const scldm_dit* h) {
  const size_t din = h->cfg.n_embed_input;
  return (size_t)256 * 4 * ((din + 5) * kD + kEdgeMaxDin) + (din + 5) * kD + kEdgeMaxDin;
}

template <int DIN>
static int inproj_backward_t(const float* dx, const float* x, int n, float* gw, float* gb, float* gpos, float* part, hipStream_t st) {
  const int groups = std::min(n, 256), rows = groups * 4, ld = (DIN + 5) * kD;
  inproj_bwd_kernel<DIN><<<groups, 256, 0, st>>>(dx, x, n, part);
  LAUNCH_CHECK();
  float* red = part + (size_t)rows * ld;   // [(DIN + 1) * 256] reduced: d in_w as [c][f], then d in_b
  edge_reduce_kernel<<<cdiv((DIN + 1) * kD, 16), 256, 0, st>>>(part, rows, ld, (DIN + 1) * kD, red);
  LAUNCH_CHECK();
  transpose_small_kernel<<<cdiv(DIN * kD, 256), 256, 0, st>>>(red, DIN, kD, gw);
  LAUNCH_CHECK();
  HIP_TRY(hipMemcpyAsync(gb, red + DIN * kD, (size_t)kD * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (gpos) {
    pos_reduce_kernel<<<16, 256, 0, st>>>(part, groups, ld, (DIN + 1) * kD, gpos);
    LAUNCH_CHECK();
  }
  return SCLDM_OK;
}
int inproj_backward(scldm_dit* h, const float* dx, const float* x, int n, float* gw, float* gb, float* gpos, float* part, hipStream_t st) {
  switch (h->cfg.n_embed_input) {
    case 16: return inproj_backward_t<16>(dx, x, n, gw, gb, gpos, part, st);
    case 32: return inproj_backward_t<32>(dx, x, n, gw, gb, gpos, part, st);
    case 8: return inproj_backward_t<8>(dx, x, n, gw, gb, gpos, part, st);
    default: return fail(SCLDM_ERR_SHAPE, "fused edge kernels are instantiated for n_embed_input 8 / 16 / 32");
  }
}

int scatter_ada_grads(scldm_dit* h, const scldm_dit_grads* g, const float* dw_all, const float* db_all, hipStream_t st) {
  const int L = h->cfg.n_layer;
  if (L > kMaxScatterLayers) return fail(SCLDM_ERR_SHAPE, "fused training path supports up to %d layers", kMaxScatterLayers);
  ScatterArgs a{};
  for (int l = 0; l < L; ++l) {
    a.w[l] = g->ada_w[l];
    a.b[l] = g->ada_b[l];
  }
  a.w[L] = g->fin_ada_w;
  a.b[L] = g->fin_ada_b;
  a.n_layer = L;
  scatter_ada_kernel<<<2048, 256, 0, st>>>(dw_all, db_all, a);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

namespace {
#define SCLDM_BWD_POLICY(NAME, NS_, ETYPE, F16)                                                                                          \
  struct NAME {                                                                                                                         \
    using Args = NS_::BwdArgs;                                                                                                          \
    using E = ETYPE;                                                                                                                    \
    static constexpr bool kF16 = F16;                                                                                                   \
    static constexpr int NW = NS_::NW, NT = NS_::NT, LDS_BYTES = NS_::LDS_BYTES, TILES_PER_64 = 2 / NS_::NTT;                           \
    static void launch(int tiles, hipStream_t st, const Args& a) { NS_::dit_backward_kernel<<<tiles, NT, LDS_BYTES, st>>>(a); }         \
    static const void* kernel() { return (const void*)NS_::dit_backward_kernel; }                                                       \
  }
SCLDM_BWD_POLICY(BwdBF16, bwd, __bf16, false);
SCLDM_BWD_POLICY(BwdFP16, bwdh, _Float16, true);
SCLDM_BWD_POLICY(BwdBF16Small, bwd32, __bf16, false);   // 32-token tiles (<= 512 cells)
SCLDM_BWD_POLICY(BwdFP16Small, bwdh32, _Float16, true);
#undef SCLDM_BWD_POLICY
}  // namespace

template <typename BW>
static int backward_layers_t(scldm_dit* h, const scldm_dit_grads* g, const float* mod, float* dmod, int n, const Record& rec, const Scratch& s,
                             hipStream_t st, const std::function<int(int)>& after_layer) {
  using E16 = typename BW::E;
  static bool attr_set[64] = {};   // (per instantiation)
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HIP_TRY(hipFuncSetAttribute(BW::kernel(), hipFuncAttributeMaxDynamicSharedMemorySize, BW::LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void*)wgrad_bf16_kernel<BW::kF16>, hipFuncAttributeMaxDynamicSharedMemorySize, kWgradLds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const scldm_dit_config& c = h->cfg;
  // (the padding tokens carry zero gradients into the operand pairs; both halves of the last 64-token tile are launched)
  const int tiles = pad4(n) / 4 * BW::TILES_PER_64, T = pad4(n) * 16, H = c.hidden_dim;
  const size_t TD = (size_t)T * kD;
  const size_t bwd_layer_elems = (size_t)BW::NW * kBwdUnitsLayer * 512;
  // Small batches (round 5): the backward kernel runs one workgroup per tile on its own CU, so at <= 640 cells it leaves a third or more
  // of the chip idle and layer l's weight-gradient launches can run there: they go to a side stream, the operand pairs alternate
  // between two sets (a set is rewritten two layers later, behind an event of the launches that read it).  SCLDM_TRAIN_WGRAD_OVERLAP=0
  // keeps everything on one stream.
  static const bool overlap_off = [] { const char* e = getenv("SCLDM_TRAIN_WGRAD_OVERLAP"); return e && e[0] == '0'; }();
  const bool overlap = s.e_set1 != nullptr && !overlap_off && c.n_layer > 1;
  hipStream_t sw = st;
  if (overlap) {
    for (int q = 0; q < 2; ++q)
      if (!h->wg_ev[q]) HIP_TRY(hipEventCreateWithFlags(&h->wg_ev[q], hipEventDisableTiming));
  }
  const ptrdiff_t set_delta = overlap ? s.e_set1 - s.e_h1 : 0;
  // MEASURED (round 6, same box, interleaved, 1 024 cells): one deferred launch on the main stream +-0 (2.019 / 2.000 against 2.013 / 2.015
  // ms per step), on side stream 2 beside the tail 1.985 / 1.995 against 1.966 / 1.968 - SLOWER: the 200 MB pass competes with the tail's
  // bandwidth-bound adaLN products.  The per-layer launches stay the default; SCLDM_TRAIN_WGRAD_DEFER=1 selects the deferred form.
  const bool defer = !overlap && s.part_layers >= c.n_layer && c.n_layer > 1;
  ReduceAll pending{};
  int n_pending = 0;
  for (int l = c.n_layer - 1; l >= 0; --l) {
    const int set = overlap ? ((c.n_layer - 1 - l) & 1) : 0;
    const ptrdiff_t sd = set ? set_delta : 0;
    if (overlap && l < c.n_layer - 2) HIP_TRY(hipStreamWaitEvent(st, h->wg_ev[set], 0));   // this set's previous readers are done
    typename BW::Args a{};
    a.x_in = rec.x + (size_t)l * TD;
    a.y1 = reinterpret_cast<const E16*>(rec.y1) + (size_t)l * TD;
    a.y2 = reinterpret_cast<const E16*>(rec.y2) + (size_t)l * TD;
    a.dx = s.dx;
    a.mod = mod;
    a.dmod = dmod;
    a.mod_stride = h->mod_w;
    a.mod_off = l * kModBlock;
    a.w_stream = reinterpret_cast<const E16*>(h->bwd_stream) + (size_t)l * bwd_layer_elems;
    a.b_qkv = h->b_qkv + (size_t)l * 768;
    auto e16 = [sd](__bf16* p) { return reinterpret_cast<E16*>(p + sd); };   // (16-bit slots; the element type is the step's operand type)
    a.e_h1 = e16(s.e_h1); a.e_dqkv = e16(s.e_dqkv); a.e_ao = e16(s.e_ao); a.e_dy1 = e16(s.e_dy1); a.e_h2 = e16(s.e_h2);
    a.e_da = e16(s.e_da); a.e_db = e16(s.e_db); a.e_hid = e16(s.e_hid); a.e_dy2 = e16(s.e_dy2);
    a.n = n;
    a.eps = c.layernorm_eps;
    a.attn_scale = 1.0f / sqrtf(32.0f);
    a.attn_scale_log2e = 1.4426950408889634f / sqrtf(32.0f);
    static unsigned long long* dbg_buf = nullptr;   // SCLDM_BWD_DBG=1: phase stamps of layer 0's launch, printed per step (debug aid)
    const bool want_dbg = l == 0 && h->bwd_dbg;
    if (want_dbg && !dbg_buf) HIP_TRY(hipMalloc(&dbg_buf, (size_t)16384 * BW::NW * 16 * 8));
    a.dbg = (want_dbg && tiles <= 16384) ? dbg_buf : nullptr;
    BW::launch(tiles, st, a);
    LAUNCH_CHECK();
    if (after_layer) {
      const int rc = after_layer(l);
      if (rc) return rc;
    }
    if (a.dbg) {
      HIP_TRY(hipStreamSynchronize(st));
      std::vector<unsigned long long> hst((size_t)tiles * BW::NW * 16);
      HIP_TRY(hipMemcpy(hst.data(), dbg_buf, hst.size() * 8, hipMemcpyDeviceToHost));
      double acc[16] = {0};
      for (int b = 0; b < tiles; ++b)
        for (int i = 1; i < 14; ++i) acc[i] += (double)(hst[((size_t)b * BW::NW) * 16 + i] - hst[((size_t)b * BW::NW) * 16 + i - 1]);
      fprintf(stderr, "[bwd phases, wave 0, mean cycles]");
      for (int i = 1; i < 14; ++i) fprintf(stderr, " %d:%.0f", i, acc[i] / tiles);
      fprintf(stderr, "\n");
    }

    // the layer's five weight gradients (+ two bias gradients as row sums of dqkv / dy1)
    WgradArgs wa{};
    ReduceArgs ra{};
    struct Spec { const __bf16* A; int lda; const __bf16* B; int ldb; int M, N; float* dst; float* bias; };
    if (overlap) {   // the side stream waits for this layer's backward kernel
      const int rc = fork_side(h, st, 2, &sw);
      if (rc) return rc;
    }
    const Spec specs[5] = {
        {s.e_dqkv + sd, 3 * kD, s.e_h1 + sd, kD, 3 * kD, kD, g->attn_w[l], g->attn_b[l]},
        {s.e_dy1 + sd, kD, s.e_ao + sd, kD, kD, kD, g->proj_w[l], g->proj_b[l]},
        {s.e_da + sd, kHP, s.e_h2 + sd, kD, H, kD, g->w1[l], nullptr},
        {s.e_db + sd, kHP, s.e_h2 + sd, kD, H, kD, g->w2[l], nullptr},
        {s.e_dy2 + sd, kD, s.e_hid + sd, kHP, kD, H, g->cproj[l], nullptr},
    };
    long part_off = 0, first = 0;
    int tile0 = 0, nr = 0;
    for (int k = 0; k < 5; ++k) {
      const Spec& sp = specs[k];
      WgradJob& j = wa.job[k];
      j.A = sp.A; j.lda = sp.lda; j.B = sp.B; j.ldb = sp.ldb; j.M = sp.M; j.N = sp.N;
      j.tile0 = tile0;
      j.tiles_n = cdiv(sp.N, 128);
      tile0 += cdiv(sp.M, 128) * j.tiles_n;
      j.part_off = part_off;
      ReduceJob& r = ra.job[nr++];
      r.part_off = part_off; r.first = first; r.dst = sp.dst; r.M = sp.M; r.N = sp.N; r.ldc = sp.N;
      first += (long)sp.M * sp.N;
      part_off += (long)kSplits * sp.M * sp.N;
      j.rs_off = -1;
      if (sp.bias) {
        j.rs_off = part_off;
        ReduceJob& rb = ra.job[nr++];
        rb.part_off = part_off; rb.first = first; rb.dst = sp.bias; rb.M = sp.M; rb.N = 1; rb.ldc = 1;
        first += sp.M;
        part_off += (long)kSplits * sp.M;
      }
    }
    wa.n_jobs = 5;
    wa.T = T;
    const int splits_req = h->wgrad_splits > 0 ? std::min(kSplits, h->wgrad_splits) : kSplitsDefault;
    wa.kchunk = cdiv(cdiv(T, splits_req), kWK) * kWK;
    float* part_l = s.part + (defer ? (size_t)l * s.part_floats : 0);
    wa.part = part_l;
    const int splits = cdiv(T, wa.kchunk);
    wa.splits = splits;
    wgrad_bf16_kernel<BW::kF16><<<tile0 * splits, 256, kWgradLds, sw>>>(wa);
    LAUNCH_CHECK();
    ra.n_jobs = nr;
    ra.splits = splits;
    ra.total = first;
    ra.part = part_l;
    if (defer) {
      pending.layer[n_pending++] = ra;
      if (n_pending == 8 || l == 0) {
        // on side stream 2, behind this layer's weight-gradient launch: nothing in the backward's tails reads a layer weight gradient, so
        // the 200 MB pass (85 us for eight layers) runs beside them; scldm_dit_train_backward joins side 2 before its gradient events
        hipStream_t sr = st;
        const int rc = fork_side(h, st, 2, &sr);
        if (rc) return rc;
        wgrad_reduce_all_kernel<<<dim3((unsigned)std::min<long>(cdiv(first, 256), 1024), n_pending), 256, 0, sr>>>(pending);
        LAUNCH_CHECK();
        h->wgrad_reduce_on_side = true;
        n_pending = 0;
      }
    } else {
      wgrad_reduce_kernel<<<(unsigned)std::min<long>(cdiv(first, 256), 4096), 256, 0, sw>>>(ra);
      LAUNCH_CHECK();
    }
    if (overlap) HIP_TRY(hipEventRecord(h->wg_ev[set], sw));
  }
  if (overlap) {   // every weight gradient is final before the caller's tails / gradient events
    const int rc = join_side(h, st, 2);
    if (rc) return rc;
  }
  return SCLDM_OK;
}

int backward_layers(scldm_dit* h, const scldm_dit_grads* g, const float* mod, float* dmod, int n, const Record& rec, const Scratch& s,
                    hipStream_t st, int precision, const std::function<int(int)>& after_layer) {
  // <= 320 cells: 32-token tiles (the same switch as the recording forward's, SCLDM_TRAIN_SMALL_NTT=0: off).  Measured per launch:
  // 256 cells 83 -> 71 us; at 512 cells the 256 half tiles take every CU, 93 us either way, and the weight gradients of the next layer
  // no longer find room beside them (graphed step 1.73 -> 1.75 ms) - so the rule stops where the side-stream overlap needs the CUs.
  static const bool small_ok = [] { const char* e = getenv("SCLDM_TRAIN_SMALL_NTT"); return !(e && e[0] == '0'); }();
  if (small_ok && pad4(n) / 4 * 2 <= kOverlapTiles)
    return precision == SCLDM_PREC_FP16 ? backward_layers_t<BwdFP16Small>(h, g, mod, dmod, n, rec, s, st, after_layer)
                                        : backward_layers_t<BwdBF16Small>(h, g, mod, dmod, n, rec, s, st, after_layer);
  return precision == SCLDM_PREC_FP16 ? backward_layers_t<BwdFP16>(h, g, mod, dmod, n, rec, s, st, after_layer)
                                      : backward_layers_t<BwdBF16>(h, g, mod, dmod, n, rec, s, st, after_layer);
}

int scale_dout(scldm_dit* h, const float* dout, long n_elem, const Scratch& s, hipStream_t st) {
  loss_scale_kernel<<<(unsigned)std::min<long>(kLsBlocks, cdiv(n_elem, 1024)), 1024, 0, st>>>(dout, n_elem, s.scale, h->found_inf);
  scale_copy_kernel<<<cdiv(n_elem, 256), 256, 0, st>>>(dout, s.scale, s.dout_s, n_elem);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

int unscale_grads(scldm_dit* h, const scldm_dit_grads* g, float* dx_out, long dx_elems, const Scratch& s, hipStream_t st) {
  const scldm_dit_config& c = h->cfg;
  const int L = c.n_layer, H = c.hidden_dim, din = c.n_embed_input;
  UnscaleArgs a{};
  bool full = false;
  auto add = [&](float* p, long n) {
    if (!p || n <= 0) return;
    if (a.count >= kMaxUnscale) { full = true; return; }   // (a dropped tensor would stay multiplied by S)
    a.p[a.count] = p; a.n[a.count] = (int)n; ++a.count;
  };
  for (int l = 0; l < L; ++l) {
    add(g->attn_w[l], 3 * kD * kD); add(g->attn_b[l], 3 * kD); add(g->proj_w[l], kD * kD); add(g->proj_b[l], kD);
    add(g->w1[l], (long)H * kD); add(g->w2[l], (long)H * kD); add(g->cproj[l], (long)H * kD);
    add(g->ada_w[l], 6 * kD * kD); add(g->ada_b[l], 6 * kD);
  }
  for (int ci = 0; ci < c.n_classes; ++ci) add(g->class_emb[ci], (long)h->tab_rows[ci] * kD);
  add(g->fin_ada_w, 2 * kD * kD); add(g->fin_ada_b, 2 * kD); add(g->t_w0, kD * 256); add(g->t_b0, kD); add(g->t_w2, kD * kD); add(g->t_b2, kD);
  add(g->in_w, (long)kD * din); add(g->in_b, kD); add(g->fin_w, (long)din * kD); add(g->fin_b, din); add(g->pos_embed, 16 * kD);
  add(dx_out, dx_elems);
  if (full) return fail(SCLDM_ERR_SHAPE, "fp16 training: more gradient tensors than the un-scale table holds (%d)", kMaxUnscale);
  a.state = s.scale;
  a.found_inf = h->found_inf;
  unscale_kernel<<<dim3(a.count, 32), 256, 0, st>>>(a);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

}  // namespace fused
}  // namespace scldm
