// C ABI of the training path (scldm_dit_train_*; see include/scldm_hip.h).  Host-side sequencing of the kernels in
// train.hpp: forward with saved activations, then the backward in reverse order.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "api_common.hpp"
#include "dit_handle.hpp"
#include "train.hpp"
#include "bgemm.hpp"
#include "bgemm8.hpp"
#include "bgemm4.hpp"
#include "attn_mfma.hpp"
#include "train_fused.hpp"
#include "cond_bwd.hpp"

using namespace scldm;
using namespace scldm::train;

namespace {

constexpr int kMaxDevices = 64;

// ---- activation record ---------------------------------------------------------------------------------------
struct LayerSaved {
  float *x_in, *st1, *h1, *qkv, *ao, *y1, *x_mid, *st2, *h2, *a, *b, *hid, *y2;
};
struct Saved {
  float *freq, *th, *sth, *c, *sc, *mod;
  std::vector<LayerSaved> layer;
  float *x_last, *st_f, *h_f;
  size_t bytes;
};
struct Scratch {
  float *dx, *dh, *dy, *dy2, *dao, *dqkv, *dhid, *da, *db, *dmod, *dsc, *dc, *dsth, *dth, *part, *temb;
  size_t part_floats;
  size_t bytes;
};

struct Carver {
  char* base;
  size_t off = 0;
  float* take(size_t floats) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align256(floats * sizeof(float));
    return p;
  }
};

// fused_path: the layers' activations are the record of train_fused.hpp (layer inputs + y1, y2 in tile layout) instead of the
// twelve per-layer arrays; the per-layer pointers other than layer[0].x_in (the record's base) stay null
Saved carve_saved(const scldm_dit* h, int n, void* base, bool fused_path) {
  const size_t T = (size_t)n * kS, H = h->cfg.hidden_dim, kD = h->cfg.n_embed;
  Carver c{reinterpret_cast<char*>(base)};
  Saved s;
  s.freq = c.take((size_t)n * 256);
  s.th = c.take((size_t)n * kD);
  s.sth = c.take((size_t)n * kD);
  s.c = c.take((size_t)n * kD);
  s.sc = c.take((size_t)n * kD);
  s.mod = c.take((size_t)n * ((size_t)h->cfg.n_layer * 6 * kD + 2 * kD));
  s.layer.resize(h->cfg.n_layer);
  if (fused_path) {
    for (auto& l : s.layer) l = LayerSaved{};
    s.layer[0].x_in = c.take(fused::carve_record(h, n, nullptr).bytes / sizeof(float) + 64);
  } else
  for (auto& l : s.layer) {
    l.x_in = c.take(T * kD);
    l.st1 = c.take(T * 2);
    l.h1 = c.take(T * kD);
    l.qkv = c.take(T * 3 * kD);
    l.ao = c.take(T * kD);
    l.y1 = c.take(T * kD);
    l.x_mid = c.take(T * kD);
    l.st2 = c.take(T * 2);
    l.h2 = c.take(T * kD);
    l.a = c.take(T * H);
    l.b = c.take(T * H);
    l.hid = c.take(T * H);
    l.y2 = c.take(T * kD);
  }
  if (fused_path && fused::edge_kernels_available(h)) {   // the final layer's backward is one kernel on the record's tile layout
    s.x_last = s.st_f = s.h_f = nullptr;
  } else {
    s.x_last = c.take(T * kD);
    s.st_f = c.take(T * 2);
    s.h_f = c.take(T * kD);
  }
  s.bytes = c.off;
  return s;
}

constexpr int kMaxSplit = 32;
// fused_path: the per-token gradient arrays of the layers (dy, dao, dqkv, dhid, da, db) are not used
Scratch carve_scratch(const scldm_dit* h, int n, void* base, bool fused_path) {
  const size_t T = (size_t)n * kS, H = h->cfg.hidden_dim, kD = h->cfg.n_embed;
  Carver c{reinterpret_cast<char*>(base)};
  Scratch s;
  s.dx = c.take(T * kD);
  s.dh = c.take(T * kD);
  if (fused_path) {
    s.dy = s.dy2 = s.dao = s.dqkv = s.dhid = s.da = s.db = nullptr;
  } else {
    s.dy = c.take(T * kD);
    s.dy2 = c.take(T * kD / 2);     // bf16 (T, D): the attention branch's gated gradient when a layer's weight gradients are batched
    s.dao = c.take(T * kD);
    s.dqkv = c.take(T * 3 * kD);
    s.dhid = c.take(T * H);
    s.da = c.take(T * H);
    s.db = c.take(T * H);
  }
  s.dmod = c.take((size_t)n * ((size_t)h->cfg.n_layer * 6 * kD + 2 * kD));
  s.dsc = c.take((size_t)n * kD);
  s.dc = c.take((size_t)n * kD);
  s.dsth = c.take((size_t)n * kD);
  s.dth = c.take((size_t)n * kD);
  s.temb = c.take((size_t)n * kD);
  // split-K partials of the largest weight gradient (6D x D, or H x D) and column-sum partials
  s.part_floats = (size_t)kMaxSplit * (std::max<size_t>((size_t)6 * kD * kD, std::max<size_t>(H, 3 * kD) * kD) + 6 * kD);  // + row sums
  s.part = c.take(s.part_floats);
  s.bytes = c.off;
  return s;
}

// ---- GEMM dispatch --------------------------------------------------------------------------------------------
template <int BF, int WTM, int WTN, bool A_KC, bool B_KC>
int launch_gemm(const GemmArgs& g, int splits, hipStream_t st) {   // BF: 0 exact fp32, 1 bf16 operands, 2 fp16 operands
  static std::atomic<bool> attr_set[kMaxDevices];   // the attribute is per device (and per function): one flag each
  constexpr int smem = BF ? hgemm_smem_bytes<WTM, WTN>() : gemm_smem_bytes<WTM, WTN>();
  auto kern = BF == 2 ? hgemm_kernel<WTM, WTN, A_KC, B_KC, true> : BF == 1 ? hgemm_kernel<WTM, WTN, A_KC, B_KC, false> : sgemm_kernel<WTM, WTN, A_KC, B_KC>;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev].load(std::memory_order_acquire)) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev].store(true, std::memory_order_release);
  }
  dim3 grid(cdiv(g.N, 64 * WTN), cdiv(g.M, 64 * WTM), splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, g);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

// C[M,N] (ldc) (+)= A(m,k) B(n,k) (+ bias[n]); element strides as in train.hpp.  `part` = split-K scratch.
thread_local int g_bf16 = 0;   // operand precision of the GEMMs of the call in progress: 0 exact fp32, 1 bf16, 2 fp16 (the fused fp16 route)

int gemm(hipStream_t st, const float* A, long sam, long sak, const float* B, long sbn, long sbk, float* C, long ldc, int M,
         int N, int K, const float* bias, bool accumulate, float* part, size_t part_floats, float* rowsum_out = nullptr) {
  if (M <= 0 || N <= 0 || K <= 0) return SCLDM_OK;
  const bool a_kc = sak == 1, b_kc = sbk == 1;
  if ((!a_kc && sam != 1) || (!b_kc && sbn != 1)) return fail(SCLDM_ERR_SHAPE, "gemm: operand needs a unit stride");
  // 128x128 tiles whenever both extents fill them (arithmetic intensity); split-K restores the workgroup count
  // ... unless they would leave most of the chip idle on a short product (the conditioning MLP's 1 024 x 256 x 256 Linears: 16 tiles
  // of 128 = 27 us of latency; 64 tiles of 64 finish in a third of that)
  const bool big = M >= 128 && N >= 128 && !((long)cdiv(M, 128) * cdiv(N, 128) < 48 && K <= 1024);
  // (128 x 256 tiles for d SiLU(c) = dmod W_all - N = 256, K = 13 824: the (M x K) operand read once - measured 75.8 + 13.8 us against
  // 59.4 + 20.5 us: 268 registers, one wave per SIMD)
  const long tiles = big ? (long)cdiv(M, 128) * cdiv(N, 128) : (long)cdiv(M, 64) * cdiv(N, 64);
  int splits = 1;
  // split K only when the output tiles alone cannot fill the GPU: a 200-tile, K = 1 024 product (the stacked adaLN weight
  // gradient) is faster without the partials round trip
  if (tiles < 160 && K >= 512) {
    splits = (int)std::min<long>(std::min<long>(cdiv(768, tiles), K / 256), kMaxSplit);
    while (splits > 1 && (size_t)splits * M * (N + 1) > part_floats) --splits;
  }
  if (rowsum_out && a_kc) return fail(SCLDM_ERR_SHAPE, "gemm: row sums need the A operand contiguous along m");
  const int bk = g_bf16 ? kBKH : kBK;
  int kchunk = cdiv(cdiv(K, splits), bk) * bk;
  splits = cdiv(K, kchunk);
  GemmArgs g{A, sam, sak, B, sbn, sbk, C, ldc, bias, M, N, K, kchunk, accumulate ? 1 : 0, rowsum_out};
  if (splits > 1) {
    g.C = part;
    g.ldc = N;
    g.bias = nullptr;
    g.accumulate = 0;
    if (rowsum_out) g.rowsum = part + (size_t)splits * M * N;   // [splits][M] behind the C partials
  }
  int rc;
#define SCLDM_GEMM_CASE(BF, WT)                                                                                             \
  (a_kc ? (b_kc ? launch_gemm<BF, WT, WT, true, true>(g, splits, st) : launch_gemm<BF, WT, WT, true, false>(g, splits, st)) \
        : (b_kc ? launch_gemm<BF, WT, WT, false, true>(g, splits, st) : launch_gemm<BF, WT, WT, false, false>(g, splits, st)))
  if (g_bf16 == 2) rc = big ? SCLDM_GEMM_CASE(2, 2) : SCLDM_GEMM_CASE(2, 1);
  else if (g_bf16) rc = big ? SCLDM_GEMM_CASE(1, 2) : SCLDM_GEMM_CASE(1, 1);
  else rc = big ? SCLDM_GEMM_CASE(0, 2) : SCLDM_GEMM_CASE(0, 1);
#undef SCLDM_GEMM_CASE
  if (rc != SCLDM_OK) return rc;
  if (splits > 1) {
    const long total = (long)M * N;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)std::min<long>(cdiv(total, 256), 2048)), dim3(256), 0, st, part,
                       splits, M, N, C, ldc, bias, accumulate ? 1 : 0);
    LAUNCH_CHECK();
    if (rowsum_out) {
      hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, part + (size_t)splits * M * N, splits, M, rowsum_out);
      LAUNCH_CHECK();
    }
  }
  return SCLDM_OK;
}

// ---- bf16-source GEMM (bgemm.hpp) -------------------------------------------------------------------------------
// SCLDM_BGEMM256 (read when the library is loaded): 0 = 128-tile kernel only, 1 = pick by fill (default), 2 = the 256-tile kernel
// whenever both extents reach 256 (tests: ragged tiles at small sizes)
const bool g_overlap = [] { const char* e = getenv("SCLDM_TRAIN_OVERLAP"); return !e || atoi(e) != 0; }();   // wgrad side stream (A/B switch)
const int g_bgemm256 = [] { const char* e = getenv("SCLDM_BGEMM256"); return e ? atoi(e) : 1; }();
const bool g_epi_lds = [] { const char* e = getenv("SCLDM_EPI_LDS"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
// 256-tiles from this many tiles on.  224 was the crossover of the register-staged bgemm256_kernel; the LDS-DMA kernel wins from ~96
// tiles (DiT-L step, profiles/r3_train_ditl_min_tiles.txt: 768 cells 43.0 -> 37.6 ms, 512 cells 30.75 -> 29.1, 384 cells 26.4 -> 25.8,
// 256 cells 21.3 -> 20.5; at 64 the 256-cell step is back at 21.3)
const int g_min_tiles256 = [] { const char* e = getenv("SCLDM_MIN_TILES256"); return e ? atoi(e) : 96; }();
const bool g_bgemm8 = [] { const char* e = getenv("SCLDM_BGEMM8"); return !e || atoi(e) != 0; }();   // LDS-DMA phase-split kernel for (KC, KC) (A/B switch)
const bool g_bgemm8_wgrad = [] { const char* e = getenv("SCLDM_BGEMM8_WGRAD"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
const bool g_bgemm_persist = [] { const char* e = getenv("SCLDM_BGEMM_PERSIST"); return e && atoi(e) != 0; }();   // A/B switch (read at load)

template <bool BIG, bool A_KC, bool B_KC>
int launch_bgemm(const BGemmArgs& g, int blocks, hipStream_t st) {
  static std::atomic<bool> attr_set[kMaxDevices];
  constexpr int smem = BIG ? kBGemm2Lds : kBGemmLds;
  auto kern = BIG ? bgemm256_kernel<A_KC, B_KC> : bgemm_kernel<A_KC, B_KC>;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev].load(std::memory_order_acquire)) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(BIG ? 512 : 256), smem, st, g);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

template <int EPI>
int launch_bgemm8_t(const BGemmArgs& g, int blocks, hipStream_t st) {
  static std::atomic<bool> attr_set[kMaxDevices];
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev].load(std::memory_order_acquire)) {
    HIP_TRY(hipFuncSetAttribute((const void*)bgemm8_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, kBGemm8Lds));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(bgemm8_kernel<EPI>, dim3(blocks), dim3(512), kBGemm8Lds, st, g);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
// vector epilogue (transposed accumulator blocks) unless the 32 rows of a store would all fall on one memory channel (fp32 rows
// that are a multiple of 4 KB apart) or the output is not aligned for 16-byte stores; bf16 results: rows of N % 4 == 0 elements (8-byte stores)
int launch_bgemm8(const BGemmArgs& g, int blocks, hipStream_t st) {
  const bool vec_ok = g.C16 ? (g.N % 4 == 0 && (reinterpret_cast<uintptr_t>(g.C16) & 7) == 0)
                            : (g.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (g.ldc * 4) % 4096 != 0);
  if (g.ep_a) return launch_bgemm8_t<2048>(g, blocks, st);   // SwiGLU backward in the epilogue (vector form; the host checked the alignment)
  // bf16 results go through LDS (2 rows x 512 contiguous bytes per store instruction; rows of N % 8 == 4 elements are only 8-byte
  // aligned: global 16-byte stores need dword alignment)
  if (g.C16 && g_epi_lds && g.N % 4 == 0 && (reinterpret_cast<uintptr_t>(g.C16) & 15) == 0) return launch_bgemm8_t<8192>(g, blocks, st);
  return vec_ok ? launch_bgemm8_t<0>(g, blocks, st) : launch_bgemm8_t<1024>(g, blocks, st);
}

int launch_bgemm8_mc(const BGemmArgs& g, int blocks, hipStream_t st) {   // both operands contiguous along m (weight gradients)
  static std::atomic<bool> attr_set[kMaxDevices];
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev].load(std::memory_order_acquire)) {
    HIP_TRY(hipFuncSetAttribute((const void*)bgemm8_kernel<0, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kBGemm8Lds));
    HIP_TRY(hipFuncSetAttribute((const void*)bgemm8_kernel<1024, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kBGemm8Lds));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev].store(true, std::memory_order_release);
  }
  const bool vec_ok = !g.C16 && g.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (g.ldc * 4) % 4096 != 0;
  if (vec_ok) hipLaunchKernelGGL((bgemm8_kernel<0, false, false>), dim3(blocks), dim3(512), kBGemm8Lds, st, g);
  else hipLaunchKernelGGL((bgemm8_kernel<1024, false, false>), dim3(blocks), dim3(512), kBGemm8Lds, st, g);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

// 128 x 128 LDS-DMA kernel (bgemm4.hpp) for (KC, KC) products that do not fill the chip with 256-tiles
// SCLDM_BGEMM4: 0 = never (default), 1 = every (KC, KC) product of the small-tile class, 2 = the forward's only (read at load).
// Opt-in: stand-alone the kernel is bit-identical to bgemm_kernel and 5-35 % faster per product (profiles/r3_gemm_probe_small.txt), inside
// the training step it measured +-0 (forward only) to -3 % (everywhere): DiT-L at 256 cells 19.4 / 19.8 against 19.25 ms, 512 x 12 at
// 512 cells 8.1 / 8.5 against 8.16 (profiles/r3_train_bgemm4_modes.txt) - its one-stage prefetch suffers next to the kernels it shares
// the chip with, where bgemm_kernel keeps two register stages in flight.
const int g_bgemm4 = [] { const char* e = getenv("SCLDM_BGEMM4"); return e ? atoi(e) : 0; }();
thread_local bool t_in_backward = false;   // set by scldm_dit_train_backward around its launches
template <int EPI>
int launch_bgemm4_t(const BGemmArgs& g, int blocks, hipStream_t st) {
  static std::atomic<bool> attr_set[kMaxDevices];
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev].load(std::memory_order_acquire)) {
    HIP_TRY(hipFuncSetAttribute((const void*)bgemm4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, kBGemm4Lds));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(bgemm4_kernel<EPI>, dim3(blocks), dim3(256), kBGemm4Lds, st, g);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
int launch_bgemm4(const BGemmArgs& g, int blocks, hipStream_t st) {
  // results through LDS need whole 16-byte pieces per row: fp32 - N, ldc multiples of 4 and a 16-byte aligned base; bf16 - N % 4 == 0
  // (rows of N % 8 == 4 elements are 8-byte aligned: global 16-byte stores need dword alignment only)
  const bool lds_ok = g.C16 ? (g.N % 4 == 0 && (reinterpret_cast<uintptr_t>(g.C16) & 15) == 0)
                            : (g.N % 4 == 0 && g.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
  return lds_ok ? launch_bgemm4_t<0>(g, blocks, st) : launch_bgemm4_t<1>(g, blocks, st);
}

// C[M,N] (ldc) (+)= A(m,k) B(n,k) (+ bias[n]); *_kc: the operand is contiguous along k (else along m / n).  Operand
// orientations in use: (KC, KC) forward, (KC, MC) data gradient, (MC, MC) weight gradient.
int bgemm(hipStream_t st, const __bf16* A, int lda, bool a_kc, const __bf16* B, int ldb, bool b_kc, float* C, long ldc, int M, int N,
          int K, const float* bias, bool accumulate, float* part, size_t part_floats, float* rowsum_out = nullptr, __bf16* C16 = nullptr,
          const BGemmArgs* ep = nullptr) {   // ep: fused-epilogue fields (ep_*) of a product that must run on bgemm8_kernel
  if (M <= 0 || N <= 0 || K <= 0) return SCLDM_OK;
  if (lda % 8 || ldb % 8 || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15))
    return fail(SCLDM_ERR_SHAPE, "bgemm: operands need 16-byte aligned rows");
  if (rowsum_out && a_kc) return fail(SCLDM_ERR_SHAPE, "bgemm: row sums need the A operand contiguous along m");
  if (!a_kc && b_kc) return fail(SCLDM_ERR_SHAPE, "bgemm: operand orientation (MC, KC) is not instantiated");
  BGemmArgs g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.bias = bias; g.C16 = C16;
  g.M = M; g.N = N; g.K = K;
  if (ep) { g.ep_a = ep->ep_a; g.ep_b = ep->ep_b; g.ep_o1 = ep->ep_o1; g.ep_o2 = ep->ep_o2; g.ep_o3 = ep->ep_o3; g.ep_n = ep->ep_n; g.ep_ldo = ep->ep_ldo; g.ep_pad = ep->ep_pad; }
  // split K when the output tiles leave most of the workgroup slots empty and the partials are small (weight gradients:
  // K = all tokens); an activation-sized output pays more for the partials round trip than it gains
  const bool may_split = (long)M * N <= (4L << 20) && !C16 && !ep;   // (a bf16 result is written by the product's own epilogue: no partials)
  auto pick_splits = [&](long tiles, long slots) {
    int sp = 1;
    if (tiles < slots / 2 && may_split) sp = (int)std::max<long>(1, std::min<long>(std::min<long>(slots / tiles, K / 512), kMaxSplit));
    while (sp > 1 && (size_t)sp * M * (N + 1) > part_floats) --sp;
    return sp;
  };
  // 256 x 256 tiles (one workgroup per CU, 8 MFMAs per 6 fragment reads: 700-820 TFLOP/s on long products against ~470) when
  // they fill the chip, else 128 x 128 (two per CU).  Measured on the DiT-L step (tests/perf/bgemm_check.py): with fewer than
  // ~224 workgroups (192 / 176 at 256 cells) or short k ranges per split (832 at 256 cells) the big tile is 0-4 % slower.
  const long tiles256 = (long)cdiv(M, 256) * cdiv(N, 256);
  const int splits256 = pick_splits(tiles256, 256);
  const bool fills = splits256 == 1 ? tiles256 >= g_min_tiles256 : (tiles256 * splits256 >= 200 && K / splits256 >= 1536);
  const bool big = g_bgemm256 && M >= 256 && N >= 256 && (fills || g_bgemm256 == 2);
  const int tile = big ? 256 : 128;
  g.tiles_m = cdiv(M, tile);
  g.tiles_n = cdiv(N, tile);
  const long tiles = (long)g.tiles_m * g.tiles_n;
  int splits = big ? splits256 : pick_splits(tiles, 512);
  g.kchunk = cdiv(cdiv(K, splits), kGK) * kGK;
  splits = cdiv(K, g.kchunk);
  g.splits = splits;
  g.accumulate = accumulate ? 1 : 0;
  g.rowsum = rowsum_out;
  g.per_xcd = (int)cdiv(tiles, 8);
  int blocks = splits > 1 ? (int)tiles * splits : 8 * g.per_xcd;
  if (splits > 1) {
    g.C = part;
    g.ldc = N;
    g.bias = nullptr;
    g.accumulate = 0;
    if (rowsum_out) g.rowsum = part + (size_t)splits * M * N;
  }
  int rc;
  // (bf16 results of the LDS-DMA kernel are stored in pairs at least: rows of an even number of elements)
  const bool use8 = big && a_kc == b_kc && splits == 1 && g_bgemm8 && (a_kc || g_bgemm8_wgrad) && (!C16 || N % 2 == 0);   // LDS-DMA kernel (bgemm8.hpp): (KC, KC) and (MC, MC)
  if (big && !use8 && g_bgemm_persist && splits == 1 && blocks > 256) {   // persistent: one workgroup per CU walks the tiles (see bgemm256_kernel)
    g.n_blocks = blocks;
    blocks = 256;
  }
  if (ep && !use8) return fail(SCLDM_ERR_SHAPE, "bgemm: a fused epilogue needs the 256-tile LDS-DMA kernel");
  bool rs_split = false;
  if (use8 && !a_kc && rowsum_out && g.tiles_n > 1 && (size_t)g.tiles_n * M <= part_floats) {   // bias gradient split over the tile columns (see wgrad_batch)
    g.rowsum = part;
    g.rowsum_split = 1;
    rs_split = true;
  }
  if (use8) rc = a_kc ? launch_bgemm8(g, blocks, st) : launch_bgemm8_mc(g, blocks, st);
  else if (big) rc = a_kc ? (b_kc ? launch_bgemm<true, true, true>(g, blocks, st) : launch_bgemm<true, true, false>(g, blocks, st))
                     : launch_bgemm<true, false, false>(g, blocks, st);
  else if (a_kc && b_kc && !ep && (g_bgemm4 == 1 || (g_bgemm4 == 2 && !t_in_backward))) rc = launch_bgemm4(g, blocks, st);
  else rc = a_kc ? (b_kc ? launch_bgemm<false, true, true>(g, blocks, st) : launch_bgemm<false, true, false>(g, blocks, st))
                 : launch_bgemm<false, false, false>(g, blocks, st);
  if (rc != SCLDM_OK) return rc;
  if (rs_split) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, part, g.tiles_n, M, rowsum_out);
    LAUNCH_CHECK();
  }
  if (splits > 1) {
    const long total = (long)M * N;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)std::min<long>(cdiv(total, 256), 2048)), dim3(256), 0, st, part,
                       splits, M, N, C, ldc, bias, accumulate ? 1 : 0);
    LAUNCH_CHECK();
    if (rowsum_out) {
      hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, part + (size_t)splits * M * N, splits, M, rowsum_out);
      LAUNCH_CHECK();
    }
  }
  return SCLDM_OK;
}

// One launch for several weight gradients dW_j[out][in] = sum_t dy_j[t][out] x_j[t][in] (both operands m-contiguous, K = tokens),
// 256 x 256 tiles, no split-K: see bgemm256_batch_kernel.  db_j (optional) receives the row sums of dy_j (bias gradient).
struct WgradJobH { const __bf16* dy; int lddy; const __bf16* x; int ldx; int out_f, in_f; float* dW; float* db; };
bool wgrad_batch_eligible(const WgradJobH* j, int n, long T) {
  if (!g_bgemm256 || n < 1 || n > kBGemmBatchMax || T < 2048) return false;
  long tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (j[i].out_f < 256 || j[i].in_f < 256 || j[i].lddy % 8 || j[i].ldx % 8) return false;
    tiles += (long)cdiv(j[i].out_f, 256) * cdiv(j[i].in_f, 256);
  }
  return tiles >= 128;   // (fewer: the per-product split-K launches fill the chip better)
}
int wgrad_batch(hipStream_t st, const WgradJobH* j, int n, long T, float* part, size_t part_floats) {
  static std::atomic<bool> attr_set[kMaxDevices];
  const bool dma = g_bgemm8 && g_bgemm8_wgrad;   // LDS-DMA form (bgemm8.hpp) of the same launch
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev].load(std::memory_order_acquire)) {
    HIP_TRY(hipFuncSetAttribute((const void*)bgemm256_batch_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kBGemm2Lds));
    HIP_TRY(hipFuncSetAttribute((const void*)bgemm8_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kBGemm8Lds));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev].store(true, std::memory_order_release);
  }
  BGemmBatch b{};
  b.n = n;
  int blocks = 0;
  size_t part_used = 0;
  for (int i = 0; i < n; ++i) {
    BGemmArgs& g = b.job[i];
    g.A = j[i].dy; g.lda = j[i].lddy; g.B = j[i].x; g.ldb = j[i].ldx; g.C = j[i].dW; g.ldc = j[i].in_f; g.bias = nullptr;
    g.M = j[i].out_f; g.N = j[i].in_f; g.K = (int)T;
    g.kchunk = cdiv(T, kGK) * kGK;
    g.splits = 1;
    g.accumulate = 0;
    g.rowsum = j[i].db;
    g.tiles_m = cdiv(g.M, 256);
    g.tiles_n = cdiv(g.N, 256);
    if (dma && j[i].db && g.tiles_n > 1 && part_used + (size_t)g.tiles_n * g.M <= part_floats) {
      // bias gradient: every tile column sums its share of the token stages into its own partial vector (all tiles of the
      // launch take equally long); colsum_final_kernel below adds the tiles_n vectors in a fixed order
      g.rowsum = part + part_used;
      g.rowsum_split = 1;
      part_used += (size_t)g.tiles_n * g.M;
    }
    g.per_xcd = cdiv((long)g.tiles_m * g.tiles_n, 8);
    b.first[i] = blocks;
    blocks += 8 * g.per_xcd;
  }
  for (int i = n; i <= kBGemmBatchMax; ++i) b.first[i] = blocks;
  if (dma) hipLaunchKernelGGL(bgemm8_batch_kernel, dim3(blocks), dim3(512), kBGemm8Lds, st, b);
  else hipLaunchKernelGGL((bgemm256_batch_kernel<false, false>), dim3(blocks), dim3(512), kBGemm2Lds, st, b);
  LAUNCH_CHECK();
  for (int i = 0; i < n; ++i)
    if (b.job[i].rowsum_split) {
      hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(b.job[i].M, 256)), dim3(256), 0, st, b.job[i].rowsum, b.job[i].tiles_n, b.job[i].M, j[i].db);
      LAUNCH_CHECK();
    }
  return SCLDM_OK;
}

// The generic path keeps bf16 arrays (and runs bgemm) when the step asks for bf16 operands.  Rows of `hidden` elements
// (hid, da, db, c_proj's weight) are padded with zeros to hidden16 = the next multiple of 8 (DiT-L: 2 732 -> 2 736).
inline int hidden16(const scldm_dit* h) { return (h->cfg.hidden_dim + 7) / 8 * 8; }
bool src16_eligible(const scldm_dit* h, int n, int precision) {
  return h->bf16_sources && precision == SCLDM_PREC_BF16 && !fused::eligible(h, n, precision) && (long)n * kS >= 128 &&
         (long)n * kS * std::max(3 * h->cfg.n_embed, hidden16(h)) < (1L << 30);   // buffer descriptors address 2 GB
}

// the stacked adaLN products also run on bf16 sources when the bf16 copy of dmod (n x mod_w) fits the scratch it borrows
// (dhid | da | db, free once the layers' backward is done): up to ~42 layers
bool ada16_eligible(const scldm_dit* h, int n, int precision) {
  return src16_eligible(h, n, precision) && (size_t)h->mod_w * sizeof(__bf16) <= 3 * (size_t)kS * h->cfg.hidden_dim * sizeof(float);
}

// per-step bf16 copies of the layers' weight matrices: [layer][attn_w | proj_w | w1 | w2 | cproj (D x hidden16)]
struct W16 {
  const __bf16 *attn_w, *proj_w, *w1, *w2, *cproj;
};
W16 w16_layer(const scldm_dit* h, int l) {
  const size_t D = h->cfg.n_embed, H = h->cfg.hidden_dim;
  const __bf16* base = reinterpret_cast<const __bf16*>(h->w16) + (size_t)l * h->w16_layer_elems;
  return W16{base, base + 3 * D * D, base + 4 * D * D, base + 4 * D * D + H * D, base + 4 * D * D + 2 * H * D};
}
// transposed copies W^T[in][out] (row = input feature, `out` rounded up to a multiple of 8 elements per row, zero padded):
// [layer][attn_w^T (D x 3D) | proj_w^T (D x D) | (w1^T | w2^T) (D x 2 hidden16: side by side in one row, the k-concatenated operand
// of the merged MLP data gradient) | cproj^T (hidden x D)]
// the branch outputs y1 = proj(ao), y2 = c_proj(hid) of the bf16 route are bf16 arrays (the fused route records them as bf16 too:
// rec_y1 / rec_y2): half the bytes in the producing epilogue, in gate_res and in the gate backward
const bool g_y16 = [] { const char* e = getenv("SCLDM_Y16"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
// (opt-in: measured SLOWER, 46.9 -> 49.3 ms at 1 024 cells - the epilogue of a 256 x 256 tile is the one part of the kernel that nothing
// overlaps, and the product leaves the group of kernels that run beside the previous layer's batched weight gradient; HISTORY.md 4.4c)
const bool g_fuse_swiglu_bwd = [] { const char* e = getenv("SCLDM_FUSE_SWIGLU_BWD"); return e && atoi(e) != 0; }();
const bool g_batch_side = [] { const char* e = getenv("SCLDM_BATCH_SIDE"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
const bool g_ada_stacked = [] { const char* e = getenv("SCLDM_ADA_STACKED"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
const bool g_fuse_gate = [] { const char* e = getenv("SCLDM_FUSE_GATE"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
const bool g_fuse_res = [] { const char* e = getenv("SCLDM_FUSE_RES"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
const bool g_grad16 = [] { const char* e = getenv("SCLDM_GRAD16"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
const bool g_mlp_merge = [] { const char* e = getenv("SCLDM_MLP_MERGE"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
const bool g_dhid16 = [] { const char* e = getenv("SCLDM_DHID16"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
const bool g_dgrad_wt = [] { const char* e = getenv("SCLDM_DGRAD_WT"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
W16 wt16_layer(const scldm_dit* h, int l) {
  const size_t D = h->cfg.n_embed, H = h->cfg.hidden_dim, Hp = (H + 7) / 8 * 8;
  const __bf16* base = reinterpret_cast<const __bf16*>(h->wt16) + (size_t)l * h->wt16_layer_elems;
  return W16{base, base + 3 * D * D, base + 4 * D * D, base + 4 * D * D + Hp, base + 4 * D * D + 2 * D * Hp};   // (w1^T / w2^T rows are 2 Hp apart)
}
// allocations and the cast-job table of the bf16 weight mirror (synchronises `st` when the table is re-uploaded): everything
// that is not a kernel launch.  scldm_dit_train_prepare runs it ahead of the first step; refresh_w16 falls back to it when
// the parameters' device pointers are not the ones it was prepared for.
// transposed copies of the weights for the data gradients (k-contiguous B operands; the two MLP data gradients as one product)
bool want_wt(const scldm_dit* h, int n) {
  // (from 32 tiles = 128 cells of a 1 024-wide model on: the merged MLP data gradient pays for the 0.2 ms transposing cast even on the
  // 128-tile kernel - DiT-L step at 256 cells 20.3 -> 19.3 ms, at 128 cells 16.23 -> 16.19)
  static const int wt_min = [] { const char* e = getenv("SCLDM_WT_MIN_TILES"); return e ? atoi(e) : 32; }();
  return g_dgrad_wt && (g_bgemm256 == 2 || cdiv((long)n * kS, 256L) * cdiv((long)h->cfg.n_embed, 256L) >= wt_min);   // (2: the tests force 256-tiles)
}
constexpr int kCastAhead = 2;
// (opt-in: measured +-0 - 45.7 / 45.8 against 45.8 / 45.8 ms at 1 024 cells, 19.4 / 19.45 against 19.4 / 19.6 at 256: the cast's 2.7 GB
// slow the kernels it runs beside by what it saves)
const bool g_cast_side = [] { const char* e = getenv("SCLDM_CAST_SIDE"); return e && atoi(e) != 0; }();
int prepare_w16(scldm_dit* h, const scldm_dit_weights* w, int n, hipStream_t st) {
  const size_t D = h->cfg.n_embed, H = h->cfg.hidden_dim, Hp = hidden16(h);
  const int L = h->cfg.n_layer;
  if (L == 0) return SCLDM_OK;
  h->w16_layer_elems = 4 * D * D + 2 * H * D + Hp * D;
  if (!h->w16) HIP_TRY(hipMalloc(&h->w16, (size_t)L * h->w16_layer_elems * sizeof(__bf16)));
  h->wt16_layer_elems = 4 * D * D + 2 * D * Hp + H * D;
  const bool wt = want_wt(h, n);
  if (wt && !h->wt16) HIP_TRY(hipMalloc(&h->wt16, (size_t)L * h->wt16_layer_elems * sizeof(__bf16)));
  h->wt16_live = wt;
  if (!h->ada16) HIP_TRY(hipMalloc(&h->ada16, (size_t)h->mod_w * D * sizeof(__bf16)));
  if (!h->ada_ball) HIP_TRY(hipMalloc(&h->ada_ball, (size_t)h->mod_w * sizeof(float)));
  std::vector<const void*> key;
  for (int l = 0; l < L; ++l)
    for (const float* p : {w->attn_w[l], w->proj_w[l], w->w1[l], w->w2[l], w->cproj[l], w->ada_w[l], w->ada_b[l]}) key.push_back(p);
  key.push_back(w->fin_ada_w);
  key.push_back(w->fin_ada_b);
  key.push_back(wt ? h->wt16 : nullptr);
  if (key != h->w16_key || !h->d_cast_jobs) {
    std::vector<CastJob> jobs;
    // every adaLN Linear stacked in the order of the modulation vector's columns: layer l rows [6 D l, 6 D (l+1)), then the final layer's 2 D
    __bf16* a16 = reinterpret_cast<__bf16*>(h->ada16);
    for (int l = 0; l <= L; ++l) {
      const int rows = (int)(l < L ? 6 * D : 2 * D);
      jobs.push_back(CastJob{l < L ? w->ada_w[l] : w->fin_ada_w, a16 + (size_t)l * 6 * D * D, rows, (int)D, (int)D, 0, nullptr, 0, 0});
      jobs.push_back(CastJob{l < L ? w->ada_b[l] : w->fin_ada_b, reinterpret_cast<__bf16*>(h->ada_ball + (size_t)l * 6 * D), 1, rows, rows, 1, nullptr, 0, 0});
    }
    // ... then the layers in order: the first two are cast ahead of the forward, the rest beside it (refresh_w16)
    int n_first = (int)jobs.size();
    for (int l = 0; l < L; ++l) {
      const W16 d = w16_layer(h, l);
      const W16 dt = wt ? wt16_layer(h, l) : W16{};
      auto tp = [](const __bf16* p) { return const_cast<__bf16*>(p); };
      jobs.push_back(CastJob{w->attn_w[l], tp(d.attn_w), (int)(3 * D), (int)D, (int)D, 0, tp(dt.attn_w), (int)(3 * D), (int)(3 * D)});
      jobs.push_back(CastJob{w->proj_w[l], tp(d.proj_w), (int)D, (int)D, (int)D, 0, tp(dt.proj_w), (int)D, (int)D});
      jobs.push_back(CastJob{w->w1[l], tp(d.w1), (int)H, (int)D, (int)D, 0, tp(dt.w1), (int)(2 * Hp), (int)Hp});
      jobs.push_back(CastJob{w->w2[l], tp(d.w2), (int)H, (int)D, (int)D, 0, tp(dt.w2), (int)(2 * Hp), (int)Hp});
      jobs.push_back(CastJob{w->cproj[l], tp(d.cproj), (int)D, (int)H, (int)Hp, 0, tp(dt.cproj), (int)D, (int)D});
      if (l < kCastAhead) n_first = (int)jobs.size();
    }
    if (!h->d_cast_jobs) HIP_TRY(hipMalloc(&h->d_cast_jobs, jobs.size() * sizeof(CastJob)));
    // (synchronous copy of a pageable vector: only when the parameters' device pointers changed)
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(h->d_cast_jobs, jobs.data(), jobs.size() * sizeof(CastJob), hipMemcpyHostToDevice));
    h->n_cast_jobs = (int)jobs.size();
    h->n_cast_first = n_first;
    h->w16_key = key;
  }
  return SCLDM_OK;
}
int refresh_w16(scldm_dit* h, const scldm_dit_weights* w, int n, hipStream_t st) {
  if (h->cfg.n_layer == 0) return SCLDM_OK;
  int rc = prepare_w16(h, w, n, st);   // no-op (a pointer-list compare) once prepared for these parameters
  if (rc != SCLDM_OK) return rc;
  // the adaLN matrices and the first kCastAhead layers on `st`; the other layers on a side stream beside the forward of those
  // layers (2.7 GB of HBM traffic for DiT-L that nothing waits for until layer kCastAhead: the forward joins there)
  const CastJob* jobs = (const CastJob*)h->d_cast_jobs;
  const int first = g_cast_side ? h->n_cast_first : h->n_cast_jobs, rest = h->n_cast_jobs - first;
  hipLaunchKernelGGL(cast_jobs_kernel, dim3(64, first), dim3(256), 0, st, jobs, first);
  LAUNCH_CHECK();
  if (rest > 0) {
    hipStream_t ss = st;
    rc = fused::fork_side(h, st, 0, &ss);
    if (rc != SCLDM_OK) return rc;
    hipLaunchKernelGGL(cast_jobs_kernel, dim3(64, rest), dim3(256), 0, ss, jobs + first, rest);
    LAUNCH_CHECK();
    h->cast_side_busy = true;
  }
  return SCLDM_OK;
}
int join_cast(scldm_dit* h, hipStream_t st) {   // the forward, before the first layer whose copies the side stream writes
  if (!h->cast_side_busy) return SCLDM_OK;
  h->cast_side_busy = false;
  return fused::join_side(h, st, 0);
}

// (ldw: elements per row of the bf16 weight copy = `in` rounded up to a multiple of 8)
// y16: the result is stored as a dense bf16 array (rows, out) at y instead of fp32 (arrays that only attention / SwiGLU kernels read:
// qkv, a, b - half the bytes written here, half the bytes read by the forward and the backward consumer)
int linear_fwd16(hipStream_t st, const __bf16* x, int ldx, const __bf16* W, int rows, int out, int in, const float* b, float* y, long ldy,
                 Scratch& s, bool y16 = false) {
  return bgemm(st, x, ldx, true, W, (in + 7) / 8 * 8, true, y, ldy, rows, out, in, b, false, s.part, s.part_floats, nullptr,
               y16 ? reinterpret_cast<__bf16*>(y) : nullptr);
}
int linear_dgrad16(hipStream_t st, const __bf16* dy, int lddy, const __bf16* W, int rows, int out, int in, float* dx, long lddx,
                   bool accumulate, Scratch& s, const __bf16* WT = nullptr, __bf16* dx16 = nullptr, int ldwt = 0) {
  // dx16: the gradient is stored as a dense bf16 array (rows, in) instead of fp32 (dhid: only the SwiGLU backward reads it)
  // with the transposed copy W^T[in][out] the product is (KC, KC) like a forward one: dx[t][i] = sum_o dy[t][o] W^T[i][o]
  if (WT) return bgemm(st, dy, lddy, true, WT, ldwt ? ldwt : (out + 7) / 8 * 8, true, dx, lddx, rows, in, out, nullptr, accumulate, s.part, s.part_floats, nullptr, dx16);
  return bgemm(st, dy, lddy, true, W, (in + 7) / 8 * 8, false, dx, lddx, rows, in, out, nullptr, accumulate, s.part, s.part_floats, nullptr, dx16);
}
int linear_wgrad16(hipStream_t st, const __bf16* dy, int lddy, const __bf16* x, int ldx, int rows, int out, int in, float* dW, Scratch& s,
                   float* db = nullptr) {
  return bgemm(st, dy, lddy, false, x, ldx, false, dW, in, out, in, rows, nullptr, false, s.part, s.part_floats, db);
}

// y[rows, out] = x[rows, in] W[out, in]^T + b   (nn.Linear forward)
int linear_fwd(hipStream_t st, const float* x, long ldx, const float* W, int rows, int out, int in, const float* b, float* y,
               long ldy, Scratch& s) {
  return gemm(st, x, ldx, 1, W, in, 1, y, ldy, rows, out, in, b, false, s.part, s.part_floats);
}
// dx[rows, in] (+)= dy[rows, out] W[out, in]
int linear_dgrad(hipStream_t st, const float* dy, long lddy, const float* W, int rows, int out, int in, float* dx, long lddx,
                 bool accumulate, Scratch& s) {
  return gemm(st, dy, lddy, 1, W, 1, in, dx, lddx, rows, in, out, nullptr, accumulate, s.part, s.part_floats);
}
// dW[out, in] = dy[rows, out]^T x[rows, in];  optionally db[out] = sum_rows dy (row sums of the A operand, same pass)
int linear_wgrad(hipStream_t st, const float* dy, long lddy, const float* x, long ldx, int rows, int out, int in, float* dW,
                 Scratch& s, float* db = nullptr) {
  return gemm(st, dy, 1, lddy, x, 1, ldx, dW, in, out, in, rows, nullptr, false, s.part, s.part_floats, db);
}
// out[cols] = sum_r X[r, c]
int colsum(hipStream_t st, const float* X, long rows, int cols, long ld, float* out, Scratch& s) {
  int splits = (int)std::max<long>(1, std::min<long>(64, rows / 64));
  while ((size_t)splits * cols > s.part_floats) --splits;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(cols, 64), splits), dim3(256), 0, st, X, rows, cols, ld, s.part);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, st, s.part, splits, cols, out);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

inline unsigned ew_grid(long count) { return (unsigned)std::max<long>(1, std::min<long>(cdiv(count, 256), 4096)); }

// ---- launches of the width-templated kernels -------------------------------------------------------------------
#define SCLDM_NQ_SWITCH(nq, CALL)  \
  switch (nq) {                    \
    case 1: { CALL(1); break; }    \
    case 2: { CALL(2); break; }    \
    case 3: { CALL(3); break; }    \
    case 4: { CALL(4); break; }    \
    case 5: { CALL(5); break; }    \
    case 6: { CALL(6); break; }    \
    case 7: { CALL(7); break; }    \
    default: { CALL(8); break; }   \
  }

template <typename TO>
int ln_fwd(hipStream_t st, int D, const float* x, const float* mod, long mw, int sc_off, int sh_off, float eps, long T, TO* h,
           float* stats) {
#define CALL(NQ) hipLaunchKernelGGL((ln_mod_fwd_kernel<NQ, TO, float>), dim3(cdiv(T, 4)), dim3(256), 0, st, x, mod, mw, sc_off, sh_off, eps, T, h, stats, (const float*)nullptr, 0, (float*)nullptr)
  SCLDM_NQ_SWITCH(D / 256, CALL)
#undef CALL
  LAUNCH_CHECK();
  return SCLDM_OK;
}
// x_out = x + gate * y, then LayerNorm-modulate of x_out (one pass over the row: ln_mod_fwd_kernel's fused residual)
template <typename TO, typename TY>
int ln_fwd_res(hipStream_t st, int D, const float* x, const TY* y, int g_off, float* x_out, const float* mod, long mw, int sc_off, int sh_off,
               float eps, long T, TO* h, float* stats) {
#define CALL(NQ) hipLaunchKernelGGL((ln_mod_fwd_kernel<NQ, TO, TY>), dim3(cdiv(T, 4)), dim3(256), 0, st, x, mod, mw, sc_off, sh_off, eps, T, h, stats, y, g_off, x_out)
  SCLDM_NQ_SWITCH(D / 256, CALL)
#undef CALL
  LAUNCH_CHECK();
  return SCLDM_OK;
}
int ln_bwd(hipStream_t st, int D, int n, const float* dh, const float* x, const float* stats, const float* mod, long mw, int sc_off,
           int sh_off, float* dx, int accumulate, float* dmod, bool dh16 = false, const __bf16* gy = nullptr, int g_off = 0,
           __bf16* gdy = nullptr) {
  // gy / g_off / gdy: the gate backward of the next branch in backward order, fused (see ln_mod_bwd_kernel)
#define CALL(NQ)                                                                                                                              \
  if (dh16) hipLaunchKernelGGL((ln_mod_bwd_kernel<NQ, __bf16>), dim3(n), dim3(64 * kLnBwdWaves), 0, st, reinterpret_cast<const __bf16*>(dh), x, \
                               stats, mod, mw, sc_off, sh_off, dx, accumulate, dmod, gy, g_off, gdy);                                           \
  else hipLaunchKernelGGL((ln_mod_bwd_kernel<NQ, float>), dim3(n), dim3(64 * kLnBwdWaves), 0, st, dh, x, stats, mod, mw, sc_off, sh_off, dx, accumulate, dmod, gy, g_off, gdy)
  SCLDM_NQ_SWITCH(D / 256, CALL)
#undef CALL
  LAUNCH_CHECK();
  return SCLDM_OK;
}
const bool g_attn_mfma = [] { const char* e = getenv("SCLDM_ATTN_MFMA"); return !e || atoi(e) != 0; }();   // A/B switch (read at load)
template <typename TO, typename TI>
int attn_fwd(hipStream_t st, int D, int n_head, long n, const TI* qkv, TO* ao) {
  if constexpr (sizeof(TO) == 2 && sizeof(TI) == 2) {   // bf16 arrays: the matrix-core kernels (attn_mfma.hpp)
    if (g_attn_mfma) {
      const dim3 grid(cdiv(n * n_head, kAttnMfmaWaves)), block(64 * kAttnMfmaWaves);
      if (D / n_head == 32) hipLaunchKernelGGL(attn_fwd_mfma_kernel<32>, grid, block, 0, st, qkv, n, n_head, D, ao);
      else hipLaunchKernelGGL(attn_fwd_mfma_kernel<64>, grid, block, 0, st, qkv, n, n_head, D, ao);
      LAUNCH_CHECK();
      return SCLDM_OK;
    }
  }
  if (D / n_head == 32) hipLaunchKernelGGL((attn_fwd_kernel<32, TO, TI>), dim3(cdiv(n * n_head, attn_waves<32>())), dim3(64 * attn_waves<32>()), 0, st, qkv, n, n_head, D, ao);
  else hipLaunchKernelGGL((attn_fwd_kernel<64, TO, TI>), dim3(cdiv(n * n_head, attn_waves<64>())), dim3(64 * attn_waves<64>()), 0, st, qkv, n, n_head, D, ao);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
template <typename TO, typename TI>
int attn_bwd(hipStream_t st, int D, int n_head, long n, const TI* qkv, const float* dao, TO* dqkv, bool dao16 = false) {
  if constexpr (sizeof(TO) == 2 && sizeof(TI) == 2) {
    if (g_attn_mfma) {
      const dim3 grid(cdiv(n * n_head, kAttnMfmaWaves)), block(64 * kAttnMfmaWaves);
      const __bf16* d16 = reinterpret_cast<const __bf16*>(dao);
      if (dao16 && D / n_head == 32) hipLaunchKernelGGL((attn_bwd_mfma_kernel<32, __bf16>), grid, block, 0, st, qkv, d16, n, n_head, D, dqkv);
      else if (dao16) hipLaunchKernelGGL((attn_bwd_mfma_kernel<64, __bf16>), grid, block, 0, st, qkv, d16, n, n_head, D, dqkv);
      else if (D / n_head == 32) hipLaunchKernelGGL((attn_bwd_mfma_kernel<32, float>), grid, block, 0, st, qkv, dao, n, n_head, D, dqkv);
      else hipLaunchKernelGGL((attn_bwd_mfma_kernel<64, float>), grid, block, 0, st, qkv, dao, n, n_head, D, dqkv);
      LAUNCH_CHECK();
      return SCLDM_OK;
    }
  }
  if (D / n_head == 32) hipLaunchKernelGGL((attn_bwd_kernel<32, TO, TI>), dim3(cdiv(n * n_head, attn_waves<32>())), dim3(64 * attn_waves<32>()), 0, st, qkv, dao, n, n_head, D, dqkv);
  else hipLaunchKernelGGL((attn_bwd_kernel<64, TO, TI>), dim3(cdiv(n * n_head, attn_waves<64>())), dim3(64 * attn_waves<64>()), 0, st, qkv, dao, n, n_head, D, dqkv);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

// The split-bf16 policy exists in the fused inference kernel only; on the training / generic entry points a request for it is served by
// the exact-fp32 GEMM route (same parity class: fp32 products are a superset of bf16x3's accuracy).  fp16 - the reference's own
// arithmetic class, TF32's mantissa (train_ldm.py:18) - TRAINS on the fused route of the base shape since round 4 (fp16 operands in the
// recording forward, the fused backward layer, the weight-gradient GEMMs and - by default - the small conditioning / adaLN GEMMs around
// them (SCLDM_TRAIN_FP16_EXACT_SMALL=1 keeps those exact fp32, check_common below); the backward loss-scaled on device with an overflow
// guard (train_fused.hip); shapes outside the fused family keep the exact-fp32 route for it.
inline int train_precision(const scldm_dit* h, int n, int precision) {
  if (precision == SCLDM_PREC_FP16 && h && fused::eligible(h, n, SCLDM_PREC_FP16)) return SCLDM_PREC_FP16;
  return (precision == SCLDM_PREC_BF16X3 || precision == SCLDM_PREC_FP16) ? SCLDM_PREC_FP32 : precision;
}

int check_common(const scldm_dit* h, const scldm_dit_weights* w, int n, int precision, const void* saved, const void* ws) {
  if (!h || !w || !saved || !ws) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (precision != SCLDM_PREC_FP32 && precision != SCLDM_PREC_BF16 && precision != SCLDM_PREC_FP16) return fail(SCLDM_ERR_SHAPE, "unknown precision %d", precision);
  // (fp16 reaches here only on the fused route - train_precision maps it to fp32 elsewhere: its conditioning / adaLN GEMMs run on
  // fp16 operands too, the arithmetic class of the whole step; SCLDM_TRAIN_FP16_EXACT_SMALL=1 keeps them exact fp32)
  static const bool f16_exact = [] { const char* e = getenv("SCLDM_TRAIN_FP16_EXACT_SMALL"); return e && e[0] == '1'; }();
  g_bf16 = precision == SCLDM_PREC_BF16 ? 1 : (precision == SCLDM_PREC_FP16 && !f16_exact) ? 2 : 0;
  if (n < 1) return fail(SCLDM_ERR_SHAPE, "n must be >= 1");
  const scldm_dit_config& c = h->cfg;
  const int hd = c.n_head > 0 ? c.n_embed / c.n_head : 0;
  if (c.n_embed % 256 != 0 || c.n_embed > 256 * kMaxNQ || c.seq_len != kS || c.n_embed % c.n_head != 0 || (hd != 32 && hd != 64))
    return fail(SCLDM_ERR_SHAPE, "training path supports n_embed %% 256 == 0 (<= %d), seq_len 16, head_dim 32 or 64 (got n_embed %d, n_head %d, seq_len %d)",
                256 * kMaxNQ, c.n_embed, c.n_head, c.seq_len);
  if (h->cfg.n_embed_input % 4 != 0) return fail(SCLDM_ERR_SHAPE, "training path needs n_embed_input %% 4 == 0 (got %d)", h->cfg.n_embed_input);
  if (h->cfg.hidden_dim % 4 != 0) return fail(SCLDM_ERR_SHAPE, "training path needs hidden_dim %% 4 == 0 (got %d)", h->cfg.hidden_dim);
  return SCLDM_OK;
}

#define TRY(expr)                   \
  do {                              \
    int rc_ = (expr);               \
    if (rc_ != SCLDM_OK) return rc_; \
  } while (0)

}  // namespace

extern "C" size_t scldm_dit_train_saved_bytes(const scldm_dit* h, int n) {
  if (!h || n < 1) return 0;
  return carve_saved(h, n, nullptr, false).bytes;   // (precision unknown here: the larger, generic layout)
}
extern "C" size_t scldm_dit_train_saved_bytes_for(const scldm_dit* h, int n, int precision) {
  if (!h || n < 1) return 0;
  precision = train_precision(h, n, precision);
  return carve_saved(h, n, nullptr, fused::eligible(h, n, precision)).bytes;
}
extern "C" size_t scldm_dit_train_workspace_bytes_for(const scldm_dit* h, int n, int precision) {
  if (!h || n < 1) return 0;
  precision = train_precision(h, n, precision);
  const bool f = fused::eligible(h, n, precision);
  return carve_scratch(h, n, nullptr, f).bytes + (f ? fused::carve_scratch(h, n, nullptr).bytes : 0);
}
extern "C" size_t scldm_dit_train_workspace_bytes(const scldm_dit* h, int n) {
  if (!h || n < 1) return 0;
  // the fused bf16 path (train_fused.hpp) carves its own scratch behind the generic one
  return carve_scratch(h, n, nullptr, false).bytes + (h->fused ? fused::carve_scratch(h, n, nullptr).bytes : 0);
}

extern "C" int scldm_dit_train_prepare(scldm_dit* h, const scldm_dit_weights* w, int n, int precision, void* stream_) {
  if (!h || !w) return fail(SCLDM_ERR_SHAPE, "null argument");
  precision = train_precision(h, n, precision);
  if (precision != SCLDM_PREC_FP32 && precision != SCLDM_PREC_BF16 && precision != SCLDM_PREC_FP16) return fail(SCLDM_ERR_SHAPE, "unknown precision %d", precision);
  hipStream_t st = (hipStream_t)stream_;
  if (fused::eligible(h, n, precision)) return fused::prepare_tables(h, w, st);
  if (src16_eligible(h, n, precision)) return prepare_w16(h, w, n, st);
  return SCLDM_OK;
}

extern "C" int scldm_dit_train_forward(scldm_dit* h, const scldm_dit_weights* w, const float* x, const float* t,
                                       const int64_t* const* labels, int n, float* out, int precision, void* saved_, void* ws,
                                       void* stream_) {
  precision = train_precision(h, n, precision);
  TRY(check_common(h, w, n, precision, saved_, ws));
  if (!x || !t || !out) return fail(SCLDM_ERR_SHAPE, "null argument");
  hipStream_t st = (hipStream_t)stream_;
  const scldm_dit_config& cfg = h->cfg;
  const int L = cfg.n_layer, din = cfg.n_embed_input, H = cfg.hidden_dim, kD = cfg.n_embed, kNH = cfg.n_head;
  const int mw = L * 6 * kD + 2 * kD;
  const long T = (long)n * kS;
  const bool use_fused = fused::eligible(h, n, precision);
  Saved s = carve_saved(h, n, saved_, use_fused);
  Scratch k = carve_scratch(h, n, ws, use_fused);

  if (use_fused) TRY(fused::prepare(h, w, st, precision));   // weight re-pack on a side stream, next to the conditioning below
  // bf16-source route (bgemm.hpp): h1, ao, h2, hid and SiLU(c) live as bf16 arrays in their (fp32-sized) slots of the saved block,
  // the weights as per-step bf16 copies; same sequence of kernels otherwise
  const bool src16 = src16_eligible(h, n, precision);
  const int Hp = hidden16(h);
  const bool ada16 = ada16_eligible(h, n, precision);
  if (src16) TRY(refresh_w16(h, w, n, st));
  // conditioning: c = t_embedder(t) + sum class embeddings; every adaLN vector (layers.py:351-364,206-216,395-398)
  hipLaunchKernelGGL(t_freq_kernel, dim3(n), dim3(256), 0, st, t, n, s.freq);
  LAUNCH_CHECK();
  TRY(linear_fwd(st, s.freq, 256, w->t_w0, n, kD, 256, w->t_b0, s.th, kD, k));
  hipLaunchKernelGGL(silu_kernel<float>, dim3(ew_grid((long)n * kD)), dim3(256), 0, st, s.th, s.sth, (long)n * kD);
  LAUNCH_CHECK();
  TRY(linear_fwd(st, s.sth, kD, w->t_w2, n, kD, kD, w->t_b2, k.temb, kD, k));
  if (!cfg.has_null_row)
    for (int c = 0; c < cfg.n_classes; ++c)
      if (!labels || !labels[c])
        return fail(SCLDM_ERR_SHAPE, "class %d needs its null token, but the class tables have no null row (cfg_dropout_prob == 0)", c);
  EmbedArgs e{};
  e.n_classes = cfg.n_classes;
  for (int c = 0; c < cfg.n_classes; ++c) {
    e.table[c] = w->class_emb[c];
    e.labels[c] = labels ? labels[c] : nullptr;
    e.vocab[c] = h->tab_rows[c] - 1;   // last table row: the null token when the tables have one (nnets.py:241-243)
  }
  hipLaunchKernelGGL(cond_sum_kernel, dim3(n, kD / 256), dim3(256), 0, st, k.temb, e, n, kD, s.c);
  LAUNCH_CHECK();
  if (ada16) hipLaunchKernelGGL(silu_kernel<__bf16>, dim3(ew_grid((long)n * kD)), dim3(256), 0, st, s.c, reinterpret_cast<__bf16*>(s.sc), (long)n * kD);
  else hipLaunchKernelGGL(silu_kernel<float>, dim3(ew_grid((long)n * kD)), dim3(256), 0, st, s.c, s.sc, (long)n * kD);
  LAUNCH_CHECK();
  if (use_fused) {
    // every adaLN Linear of the network in ONE GEMM over the handle's all-layer transposed copy (ada_t is (D, mod_w), refreshed
    // from the live parameters by prepare()): mod = SiLU(c) W_all^T + b_all
    TRY(fused::prepare_join(h, st));
    TRY(gemm(st, s.sc, kD, 1, h->ada_t, 1, mw, s.mod, mw, n, mw, kD, h->ada_b, false, k.part, k.part_floats));
  } else if (ada16) {
    // every adaLN Linear of the network in ONE bf16-source GEMM over the stacked weight copy (refresh_w16): 2 336 tiles instead of
    // 25 launches of 96
    TRY(bgemm(st, reinterpret_cast<const __bf16*>(s.sc), kD, true, reinterpret_cast<const __bf16*>(h->ada16), kD, true, s.mod, mw, n, mw, kD,
              h->ada_ball, false, k.part, k.part_floats));
  } else {
    for (int l = 0; l < L; ++l)
      TRY(linear_fwd(st, s.sc, kD, w->ada_w[l], n, 6 * kD, kD, w->ada_b[l], s.mod + (long)l * 6 * kD, mw, k));
    TRY(linear_fwd(st, s.sc, kD, w->fin_ada_w, n, 2 * kD, kD, w->fin_ada_b, s.mod + (long)L * 6 * kD, mw, k));
  }

  if (use_fused) {
    // base shape, bf16 operands: the whole trunk is the fused inference kernel with a training record (layer inputs + the two
    // gated branch outputs; 32 KB per cell per layer) instead of ~290 KB of saved activations
    const fused::Record rec = fused::carve_record(h, n, s.layer[0].x_in);
    const fused::Scratch fs = fused::carve_scratch(h, n, reinterpret_cast<char*>(ws) + k.bytes);
    return fused::forward(h, x, s.mod, n, out, rec, fs, st, precision);
  }
  // x_0 = input_proj(x) + pos_embed (nnets.py:290)
  float* x0 = L > 0 ? s.layer[0].x_in : s.x_last;
  TRY(linear_fwd(st, x, din, w->in_w, (int)T, kD, din, w->in_b, x0, kD, k));
  hipLaunchKernelGGL(add_pos_kernel, dim3(ew_grid(T * kD)), dim3(256), 0, st, x0, w->pos_embed, T, kD);
  LAUNCH_CHECK();

  const bool overlap = src16 && g_overlap;   // the two up-projections of the MLP (same input, 1.4 rounds of tiles each) side by side
  Scratch k2 = k;                            // the side stream's half of the split-K scratch (small batches split the forward products too)
  if (overlap) {
    const size_t half = (k.part_floats / 2) & ~(size_t)63;
    k2.part = k.part + half;
    k2.part_floats = k.part_floats - half;
    k.part_floats = half;
  }
  const bool y16 = src16 && g_y16;
  const bool fuse_res = y16 && g_fuse_res;   // residual step fused into the following LayerNorm-modulate (bf16 branch outputs)
  auto lin = [&](const float* xin, int ldx, const float* W, const __bf16* Wh, int out_f, int in_f, const float* b, float* y,
                 hipStream_t sx = nullptr, bool y16 = false) {
    return src16 ? linear_fwd16(sx ? sx : st, reinterpret_cast<const __bf16*>(xin), ldx, Wh, (int)T, out_f, in_f, b, y, out_f, sx ? k2 : k, y16)
                 : linear_fwd(st, xin, ldx, W, (int)T, out_f, in_f, b, y, out_f, k);
  };
  for (int l = 0; l < L; ++l) {
    LayerSaved& a = s.layer[l];
    const int o = l * 6 * kD;   // a0..a5 at o + i*D (layers.py:214-216)
    if (src16 && l == kCastAhead) TRY(join_cast(h, st));
    const W16 wh = src16 ? w16_layer(h, l) : W16{};
    float* x_next = l + 1 < L ? s.layer[l + 1].x_in : s.x_last;
    // (fuse_res: the previous layer's second residual step already produced x_in, h1 and st1 in one pass)
    if (src16 && !(fuse_res && l > 0)) TRY(ln_fwd(st, kD, a.x_in, s.mod, (long)mw, o, o + kD, cfg.layernorm_eps, T, reinterpret_cast<__bf16*>(a.h1), a.st1));
    else if (!src16) TRY(ln_fwd(st, kD, a.x_in, s.mod, (long)mw, o, o + kD, cfg.layernorm_eps, T, a.h1, a.st1));
    TRY(lin(a.h1, kD, w->attn_w[l], wh.attn_w, 3 * kD, kD, w->attn_b[l], a.qkv, nullptr, src16));   // (bf16 route: qkv itself is a bf16 array)
    if (src16) TRY(attn_fwd(st, kD, kNH, n, reinterpret_cast<const __bf16*>(a.qkv), reinterpret_cast<__bf16*>(a.ao)));
    else TRY(attn_fwd(st, kD, kNH, n, (const float*)a.qkv, a.ao));
    TRY(lin(a.ao, kD, w->proj_w[l], wh.proj_w, kD, kD, w->proj_b[l], a.y1, nullptr, y16));
    if (fuse_res) {   // x_mid = x_in + a2 * y1 and h2 = LN(x_mid)(1 + a3) + a4 in one pass over the row
      TRY(ln_fwd_res(st, kD, a.x_in, reinterpret_cast<const __bf16*>(a.y1), o + 2 * kD, a.x_mid, s.mod, (long)mw, o + 3 * kD, o + 4 * kD, cfg.layernorm_eps, T,
                     reinterpret_cast<__bf16*>(a.h2), a.st2));
    } else {
      if (y16) hipLaunchKernelGGL(gate_res_kernel<__bf16>, dim3(ew_grid(T * kD / 4)), dim3(256), 0, st, a.x_in, reinterpret_cast<const __bf16*>(a.y1), s.mod, (long)mw, o + 2 * kD, T, kD, a.x_mid);
      else hipLaunchKernelGGL(gate_res_kernel<float>, dim3(ew_grid(T * kD / 4)), dim3(256), 0, st, a.x_in, a.y1, s.mod, (long)mw, o + 2 * kD, T, kD, a.x_mid);
      LAUNCH_CHECK();
      if (src16) TRY(ln_fwd(st, kD, a.x_mid, s.mod, (long)mw, o + 3 * kD, o + 4 * kD, cfg.layernorm_eps, T, reinterpret_cast<__bf16*>(a.h2), a.st2));
      else TRY(ln_fwd(st, kD, a.x_mid, s.mod, (long)mw, o + 3 * kD, o + 4 * kD, cfg.layernorm_eps, T, a.h2, a.st2));
    }
    hipStream_t s2 = nullptr;
    if (overlap) TRY(fused::fork_side(h, st, 2, &s2));
    TRY(lin(a.h2, kD, w->w2[l], wh.w2, H, kD, nullptr, a.b, s2, src16));   // (bf16 route: the pre-activations a, b are bf16 arrays)
    TRY(lin(a.h2, kD, w->w1[l], wh.w1, H, kD, nullptr, a.a, nullptr, src16));
    if (overlap) TRY(fused::join_side(h, st, 2));
    if (src16) hipLaunchKernelGGL((swiglu_fwd_kernel<__bf16, __bf16>), dim3(ew_grid(T * H)), dim3(256), 0, st, reinterpret_cast<const __bf16*>(a.a),
                                  reinterpret_cast<const __bf16*>(a.b), reinterpret_cast<__bf16*>(a.hid), T * H, H, Hp);
    else hipLaunchKernelGGL((swiglu_fwd_kernel<float, float>), dim3(ew_grid(T * H)), dim3(256), 0, st, (const float*)a.a, (const float*)a.b, a.hid, T * H, H, H);
    LAUNCH_CHECK();
    TRY(lin(a.hid, src16 ? Hp : H, w->cproj[l], wh.cproj, kD, H, nullptr, a.y2, nullptr, y16));
    if (fuse_res) {   // x_next = x_mid + a5 * y2 and the NEXT LayerNorm-modulate (the next layer's first, or the final layer's) in one pass
      const int on = (l + 1) * 6 * kD;
      if (l + 1 < L) TRY(ln_fwd_res(st, kD, a.x_mid, reinterpret_cast<const __bf16*>(a.y2), o + 5 * kD, x_next, s.mod, (long)mw, on, on + kD, cfg.layernorm_eps, T,
                                    reinterpret_cast<__bf16*>(s.layer[l + 1].h1), s.layer[l + 1].st1));
      else TRY(ln_fwd_res(st, kD, a.x_mid, reinterpret_cast<const __bf16*>(a.y2), o + 5 * kD, x_next, s.mod, (long)mw, on + kD, on, cfg.layernorm_eps, T, s.h_f, s.st_f));
    } else {
      if (y16) hipLaunchKernelGGL(gate_res_kernel<__bf16>, dim3(ew_grid(T * kD / 4)), dim3(256), 0, st, a.x_mid, reinterpret_cast<const __bf16*>(a.y2), s.mod, (long)mw, o + 5 * kD, T, kD, x_next);
      else hipLaunchKernelGGL(gate_res_kernel<float>, dim3(ew_grid(T * kD / 4)), dim3(256), 0, st, a.x_mid, a.y2, s.mod, (long)mw, o + 5 * kD, T, kD, x_next);
      LAUNCH_CHECK();
    }
  }
  // FinalLayerDit (layers.py:397-401): [shift | scale] = adaLN(c); LN(x) * (1 + scale) + shift; Linear
  const int of = L * 6 * kD;
  if (!(fuse_res && L > 0)) TRY(ln_fwd(st, kD, s.x_last, s.mod, (long)mw, of + kD, of, cfg.layernorm_eps, T, s.h_f, s.st_f));
  TRY(linear_fwd(st, s.h_f, kD, w->fin_w, (int)T, din, kD, w->fin_b, out, din, k));
  return SCLDM_OK;
}

extern "C" int scldm_dit_train_set_grad_events(scldm_dit* h, void* const* events, const int* kinds, const int* layers, int n) {
  if (!h || n < 0 || (n > 0 && (!events || !kinds || !layers))) return fail(SCLDM_ERR_SHAPE, "bad argument");
  h->grad_events.clear();
  for (int i = 0; i < n; ++i) {
    if (kinds[i] < SCLDM_GRAD_LAYER || kinds[i] > SCLDM_GRAD_END || !events[i]) return fail(SCLDM_ERR_SHAPE, "bad gradient event %d", i);
    h->grad_events.push_back(scldm_dit::GradEvent{(hipEvent_t)events[i], kinds[i], layers[i], false});
  }
  return SCLDM_OK;
}

extern "C" int scldm_dit_train_backward(scldm_dit* h, const scldm_dit_weights* w, const scldm_dit_grads* g, const float* x,
                                        const int64_t* const* labels, const float* dout, int n, float* dx_out, int precision,
                                        void* saved_, void* ws, void* stream_) {
  struct InBackward { InBackward() { t_in_backward = true; } ~InBackward() { t_in_backward = false; } } in_backward_scope;
  // the gradient-ready events belong to THIS call only: whatever way it returns, none of the raw hipEvent_t handles survives on the
  // handle (the caller may destroy them right after; a later backward without set_grad_events must not record into them)
  struct DropEvents { scldm_dit* h; ~DropEvents() { if (h) h->grad_events.clear(); } } drop_events_scope{h};
  precision = train_precision(h, n, precision);
  TRY(check_common(h, w, n, precision, saved_, ws));
  if (!g || !x || !dout) return fail(SCLDM_ERR_SHAPE, "null argument");
  hipStream_t st = (hipStream_t)stream_;
  const scldm_dit_config& cfg = h->cfg;
  const int L = cfg.n_layer, din = cfg.n_embed_input, H = cfg.hidden_dim, kD = cfg.n_embed, kNH = cfg.n_head;
  const int mw = L * 6 * kD + 2 * kD;
  const long T = (long)n * kS;
  Saved s = carve_saved(h, n, saved_, fused::eligible(h, n, precision));
  Scratch k = carve_scratch(h, n, ws, fused::eligible(h, n, precision));

  const bool use_fused = fused::eligible(h, n, precision);
  const long T_pad = (long)((n + 3) / 4 * 4) * kS;   // tokens incl. the padding of the last 64-token tile (tile layouts)
  const fused::Record rec = use_fused ? fused::carve_record(h, n, s.layer[0].x_in) : fused::Record{};
  const fused::Scratch fs = use_fused ? fused::carve_scratch(h, n, reinterpret_cast<char*>(ws) + k.bytes) : fused::Scratch{};
  // ---- final layer ----
  const int of = L * 6 * kD;
  const bool edge = use_fused && fused::edge_kernels_available(h);   // both ends of the backward as single kernels on the tile layout
  // fp16 policy: gradients of a mean loss are ~1e-6 - below fp16's normal range.  The whole backward is linear in dout, so it runs on
  // S * dout (S a power of two chosen ON DEVICE so that max |S dout| lands in [8, 16): no host read) and every gradient it produced is
  // multiplied by 1 / S at the end (exact: powers of two) - before any gradient-ready event is recorded.
  const bool f16 = use_fused && precision == SCLDM_PREC_FP16;
  if (f16) {
    TRY(fused::scale_dout(h, dout, (long)T * din, fs, st));
    dout = fs.dout_s;
  }
  if (edge) {
    TRY(fused::final_backward(h, rec.x + (size_t)L * T_pad * kD, s.mod, dout, w->fin_w, n, fs.dx, k.dmod, g->fin_w, g->fin_b, fs.edge_part, st));
  } else {
    if (use_fused) {   // the record holds the final layer's input in tile layout; its LayerNorm output is recomputed
      TRY(fused::to_plain(rec.x + (size_t)L * T_pad * kD, s.x_last, n, st));
      TRY(ln_fwd(st, kD, s.x_last, s.mod, (long)mw, of + kD, of, cfg.layernorm_eps, T, s.h_f, s.st_f));
    }
    TRY(linear_wgrad(st, dout, din, s.h_f, kD, (int)T, din, kD, g->fin_w, k, g->fin_b));
    TRY(linear_dgrad(st, dout, din, w->fin_w, (int)T, din, kD, k.dh, kD, false, k));
    TRY(ln_bwd(st, kD, n, k.dh, s.x_last, s.st_f, s.mod, (long)mw, of + kD, of, k.dx, 0, k.dmod));
    if (use_fused) TRY(fused::to_tile(k.dx, fs.dx, n, st));
  }
  // Fused route, adaLN Linears mod_l = SiLU(c) W_l^T + b_l (round 5): the two products of every layer's slice of dmod - the weight
  // gradient d W_l = dmod_l^T SiLU(c) (+ row sums = bias gradient) and the running sum d SiLU(c) += dmod_l W_l - are queued on a side
  // stream as soon as that layer's backward kernel is (the final layer's slice first), so they run beside the layers still being
  // differentiated instead of as two K = 13 824 / M = 13 824 products in the tail after the last layer (140 us of a 2.2 ms step).
  // MEASURED SLOWER at 1 024 cells (2.77 against 2.15 ms per step: the small products take workgroup slots - and with them whole CUs'
  // worth of LDS - from the backward layer, which needs one CU per tile) and +-0 at 256 cells: opt-in, SCLDM_TRAIN_ADA_STREAM=1.
  static const bool ada_opt = [] { const char* e = getenv("SCLDM_TRAIN_ADA_STREAM"); return e && e[0] == '1'; }();
  const bool ada_stream = use_fused && ada_opt;
  hipStream_t s_ada_l = st;
  auto ada_slice = [&](int l, bool first) -> int {
    const int width = l < L ? 6 * kD : 2 * kD;
    const size_t off = (size_t)l * 6 * kD;
    float* dw_all = fs.ada_dw;
    float* db_all = fs.ada_dw + (size_t)mw * kD;
    TRY(gemm(s_ada_l, k.dmod + off, 1, mw, s.sc, 1, kD, dw_all + off * kD, kD, width, kD, n, nullptr, false, nullptr, 0, db_all + off));
    TRY(gemm(s_ada_l, k.dmod + off, mw, 1, h->ada_t + off, mw, 1, k.dsc, kD, n, kD, width, nullptr, !first, nullptr, 0));
    return SCLDM_OK;
  };
  if (use_fused) {
    TRY(fused::backward_join(h, st));
    if (ada_stream) {
      TRY(fused::fork_side(h, st, 1, &s_ada_l));     // (the final layer's dmod slice is complete here)
      TRY(ada_slice(L, true));
      TRY(fused::backward_layers(h, g, s.mod, k.dmod, n, rec, fs, st, precision, [&](int l) -> int {
        TRY(fused::fork_side(h, st, 1, &s_ada_l));   // the side stream waits for layer l's kernel
        return ada_slice(l, false);
      }));
    } else {
      TRY(fused::backward_layers(h, g, s.mod, k.dmod, n, rec, fs, st, precision));
    }
    if (edge) TRY(fused::join_side(h, st, 0));   // the final layer's weight / bias gradient reduction (side stream 0, fused::final_backward)
    if (!edge || dx_out) TRY(fused::to_plain(fs.dx, k.dx, n, st));
  }
  // bf16-source route: dy, dqkv, da, db (consumed only by GEMMs) are bf16 arrays in their slots of the scratch block
  const bool src16 = src16_eligible(h, n, precision);
  if (src16 && !h->w16) return fail(SCLDM_ERR_SHAPE, "training backward without the forward of the same step");
  const int Hl = src16 ? hidden16(h) : H;   // elements per row of hid / da / db
  // bf16-source route: the weight gradients of a layer run on a side stream next to the data-gradient chain.  Each of the two
  // is a few hundred workgroups (240-workgroup split-K products, 256-tile activation-sized ones at 256 cells) that leave half of
  // the chip's workgroup slots empty on their own.  fork = the side stream waits for what the main stream holds (the operand a
  // weight gradient reads is ready); join = the main stream waits for the side stream before it overwrites such an operand.
  // The two streams split the split-K scratch.
  // Batched weight gradients (bf16-source route, every product of a layer at least 256 x 256: the DiT-L of configs[4]): the five
  // weight gradients of a layer are ONE launch of 256 x 256 tiles at the end of the layer - 196 tiles for the DiT-L shape fill the
  // chip without split-K, so there are no partial tiles and no reduction kernels (7 % of the step at 1 024 cells before).  The
  // attention branch's gated gradient then gets its own array (dy2): both dy arrays must live until the launch.
  WgradJobH wj[kBGemmBatchMax];
  int n_wj = 0;
  bool batched = false;
  if (src16 && L > 0 && h->wgrad_batch) {
    const WgradJobH probe[5] = {{nullptr, kD, nullptr, Hl, kD, H, nullptr, nullptr}, {nullptr, Hl, nullptr, kD, H, kD, nullptr, nullptr},
                                {nullptr, Hl, nullptr, kD, H, kD, nullptr, nullptr}, {nullptr, kD, nullptr, kD, kD, kD, nullptr, nullptr},
                                {nullptr, 3 * kD, nullptr, kD, 3 * kD, kD, nullptr, nullptr}};
    batched = wgrad_batch_eligible(probe, 5, T);
  }
  const bool overlap = src16 && g_overlap && !batched;
  // Batched launch BESIDE the chain (side stream): 196 tiles of a DiT-L layer occupy 196 of the 256 CUs for ~490 us; the next kernels of
  // the data-gradient chain (c_attn's data gradient, the LayerNorm backward, the next layer's gate backward and c_proj data gradient)
  // do not touch the batch's operands - with the ONE exception of dy, which therefore alternates between the two halves of its
  // (fp32-sized) slot - and fill the idle quarter of the chip.  The chain waits for the batch before the first kernel that
  // overwrites an operand (the next layer's SwiGLU backward: da | db).
  const bool batch_side = batched && g_batch_side;
  Scratch kb = k;   // the batch's row-sum partials: the tail of the split-K scratch (the chain's products keep the rest)
  if (batch_side) {
    const size_t tail = std::min<size_t>(k.part_floats / 2, (size_t)64 << 10);
    kb.part = k.part + (k.part_floats - tail);
    kb.part_floats = tail;
    k.part_floats -= tail;
  }
  bool batch_busy = false;
  auto join_batch = [&]() -> int {
    if (!batch_busy) return SCLDM_OK;
    batch_busy = false;
    return fused::join_side(h, st, 2);
  };
  Scratch kw = k;
  if (overlap) {
    const size_t half = (k.part_floats / 2) & ~(size_t)63;
    kw.part = k.part + half;
    kw.part_floats = k.part_floats - half;
    k.part_floats = half;
  }
  hipStream_t sw = st;
  bool side_busy = false;
  auto fork = [&]() -> int {
    if (!overlap) return SCLDM_OK;
    side_busy = true;
    return fused::fork_side(h, st, 2, &sw);
  };
  auto join = [&]() -> int {
    if (!side_busy) return SCLDM_OK;
    side_busy = false;
    return fused::join_side(h, st, 2);
  };
  auto wgrad = [&](const float* dyp, int lddy, const float* xs, int ldx, int out_f, int in_f, float* dW, float* db) {
    if (batched) {   // recorded; launched with the layer's other weight gradients (wgrad_batch below)
      wj[n_wj++] = WgradJobH{reinterpret_cast<const __bf16*>(dyp), lddy, reinterpret_cast<const __bf16*>(xs), ldx, out_f, in_f, dW, db};
      return (int)SCLDM_OK;
    }
    return src16 ? linear_wgrad16(sw, reinterpret_cast<const __bf16*>(dyp), lddy, reinterpret_cast<const __bf16*>(xs), ldx, (int)T, out_f, in_f, dW, kw, db)
                 : linear_wgrad(st, dyp, lddy, xs, ldx, (int)T, out_f, in_f, dW, k, db);
  };
  auto dgrad = [&](const float* dyp, int lddy, const float* W, const __bf16* Wh, const __bf16* WhT, int out_f, int in_f, float* dxp, bool acc,
                   bool out16 = false, int ldwt = 0) {
    return src16 ? linear_dgrad16(st, reinterpret_cast<const __bf16*>(dyp), lddy, Wh, (int)T, out_f, in_f, dxp, in_f, acc, k, WhT,
                                  out16 ? reinterpret_cast<__bf16*>(dxp) : nullptr, ldwt)
                 : linear_dgrad(st, dyp, lddy, W, (int)T, out_f, in_f, dxp, in_f, acc, k);
  };
  auto gate_bwd = [&](const float* yv, int g_off, float* dst) {
    if (src16 && g_y16) hipLaunchKernelGGL((gate_bwd_kernel<__bf16, __bf16>), dim3(n, kD / 256), dim3(256), 0, st, k.dx, reinterpret_cast<const __bf16*>(yv), s.mod, (long)mw, g_off, kD, reinterpret_cast<__bf16*>(dst), k.dmod);
    else if (src16) hipLaunchKernelGGL((gate_bwd_kernel<__bf16, float>), dim3(n, kD / 256), dim3(256), 0, st, k.dx, yv, s.mod, (long)mw, g_off, kD, reinterpret_cast<__bf16*>(dst), k.dmod);
    else hipLaunchKernelGGL((gate_bwd_kernel<float, float>), dim3(n, kD / 256), dim3(256), 0, st, k.dx, yv, s.mod, (long)mw, g_off, kD, dst, k.dmod);
  };
  float* const dy_attn = batched ? k.dy2 : k.dy;
  // gradient-ready events (scldm_dit_train_set_grad_events): `st` is ordered after every kernel that writes the gradients an event
  // stands for when it is recorded.  Events the route cannot time individually fire at the end of the call.
  auto fire = [&](int kind, int layer) -> int {
    if (f16 && kind != SCLDM_GRAD_END) return SCLDM_OK;   // (loss-scaled backward: nothing is final before the un-scaling pass at the end)
    for (auto& e : h->grad_events)
      if (!e.fired && (kind == SCLDM_GRAD_END || (e.kind == kind && (kind == SCLDM_GRAD_LAYER ? e.layer >= layer : e.layer <= layer)))) {
        HIP_TRY(hipEventRecord(e.ev, st));
        e.fired = true;
      }
    return SCLDM_OK;
  };
  bool dy_ready = false;
  for (int l = use_fused ? -1 : L - 1; l >= 0; --l) {
    LayerSaved& a = s.layer[l];
    const int o = l * 6 * kD;
    const W16 wh = src16 ? w16_layer(h, l) : W16{};
    const W16 wt = (src16 && h->wt16_live) ? wt16_layer(h, l) : W16{};
    // the gradients that only an elementwise kernel reads next (dh -> LayerNorm backward, dao -> attention backward) leave their data
    // gradient's epilogue as bf16 arrays, like dhid: half the bytes on both sides
    const bool g16 = src16 && g_grad16;
    bool dh_mlp16 = false;
    // x_out = x_mid + a5 * y2,  y2 = c_proj(hid),  hid = silu(w1 h2) * (w2 h2),  h2 = LN(x_mid)(1 + a3) + a4
    TRY(join());   // (the previous layer's weight gradients read dy / da / db / dqkv)
    if (l + 1 < L && !batch_side) TRY(fire(SCLDM_GRAD_LAYER, l + 1));   // the main weight gradients of layers l + 1 .. L - 1 are queued before this point
    float* const dy_alt = reinterpret_cast<float*>(reinterpret_cast<__bf16*>(k.dy) + (size_t)T * kD);   // (second half of dy's fp32-sized slot)
    float* const dy_cur = (batch_side && ((L - 1 - l) & 1)) ? dy_alt : k.dy;
    float* const dy_nxt = (batch_side && ((L - 1 - l) & 1)) ? k.dy : (batch_side ? dy_alt : k.dy);
    if (!dy_ready) {   // (else: written by the previous iteration's last LayerNorm backward, see fuse_gate below)
      gate_bwd(a.y2, o + 5 * kD, dy_cur);
      LAUNCH_CHECK();
    }
    dy_ready = false;
    TRY(fork());
    TRY(wgrad(dy_cur, kD, a.hid, Hl, kD, H, g->cproj[l], nullptr));
    // bf16 route: da and db live side by side in one row (da16[t][0 .. Hl) | [Hl .. 2 Hl)): with the transposed copies (w1^T | w2^T laid
    // out the same way) the two data gradients of the MLP are ONE product over k = 2 Hl - one epilogue instead of two and no
    // read-modify-write of dh
    float* const da_p = k.da;
    float* const db_p = src16 ? reinterpret_cast<float*>(reinterpret_cast<__bf16*>(k.da) + Hl) : k.db;
    const int ldab = src16 ? 2 * Hl : H;
    // the SwiGLU backward in the epilogue of c_proj's data gradient (LDS-DMA kernel, transposed copies: large batches): d hid never
    // leaves the registers, da | db are written by the product itself
    const bool fuse_swiglu = src16 && wt.cproj && g_fuse_swiglu_bwd && g_bgemm8 && H % 4 == 0;
    if (fuse_swiglu) {
      if (batch_side) {   // (the product's epilogue overwrites da | db, which the previous layer's batch reads)
        TRY(join_batch());
        if (l + 1 < L) TRY(fire(SCLDM_GRAD_LAYER, l + 1));
      }
      BGemmArgs ep{};
      ep.ep_a = reinterpret_cast<const __bf16*>(a.a);
      ep.ep_b = reinterpret_cast<const __bf16*>(a.b);
      ep.ep_o1 = reinterpret_cast<__bf16*>(da_p);
      ep.ep_o2 = reinterpret_cast<__bf16*>(db_p);
      ep.ep_n = H; ep.ep_ldo = ldab; ep.ep_pad = Hl;
      TRY(bgemm(st, reinterpret_cast<const __bf16*>(dy_cur), kD, true, wt.cproj, (kD + 7) / 8 * 8, true, k.dhid, H, (int)T, H, kD, nullptr, false,
                k.part, k.part_floats, nullptr, nullptr, &ep));
    } else {
    TRY(dgrad(dy_cur, kD, w->cproj[l], wh.cproj, wt.cproj, kD, H, k.dhid, false, g_dhid16));   // (bf16 route: dhid is a bf16 array, like a, b, da, db)
    if (batch_side) {   // the previous layer's batch read da | db, dy2, dqkv: from here on they are overwritten
      TRY(join_batch());
      if (l + 1 < L) TRY(fire(SCLDM_GRAD_LAYER, l + 1));
    }
    if (src16 && g_dhid16) hipLaunchKernelGGL((swiglu_bwd_kernel<__bf16, __bf16, __bf16>), dim3(ew_grid(T * H)), dim3(256), 0, st, reinterpret_cast<const __bf16*>(k.dhid), reinterpret_cast<const __bf16*>(a.a),
                                  reinterpret_cast<const __bf16*>(a.b), reinterpret_cast<__bf16*>(da_p), reinterpret_cast<__bf16*>(db_p), T * H, H, ldab, Hl);
    else if (src16) hipLaunchKernelGGL((swiglu_bwd_kernel<__bf16, __bf16>), dim3(ew_grid(T * H)), dim3(256), 0, st, k.dhid, reinterpret_cast<const __bf16*>(a.a),
                                  reinterpret_cast<const __bf16*>(a.b), reinterpret_cast<__bf16*>(da_p), reinterpret_cast<__bf16*>(db_p), T * H, H, ldab, Hl);
    else hipLaunchKernelGGL((swiglu_bwd_kernel<float, float>), dim3(ew_grid(T * H)), dim3(256), 0, st, k.dhid, (const float*)a.a, (const float*)a.b, k.da, k.db, T * H, H, H, H);
    }
    LAUNCH_CHECK();
    TRY(fork());
    TRY(wgrad(da_p, ldab, a.h2, kD, H, kD, g->w1[l], nullptr));
    TRY(wgrad(db_p, ldab, a.h2, kD, H, kD, g->w2[l], nullptr));
    if (src16 && wt.w1 && g_mlp_merge) {
      TRY(linear_dgrad16(st, reinterpret_cast<const __bf16*>(da_p), ldab, nullptr, (int)T, 2 * Hl, kD, k.dh, kD, false, k, wt.w1,
                         g16 ? reinterpret_cast<__bf16*>(k.dh) : nullptr));
      dh_mlp16 = g16;
    } else {
      TRY(dgrad(da_p, ldab, w->w1[l], wh.w1, wt.w1, H, kD, k.dh, false, false, 2 * Hl));
      TRY(dgrad(db_p, ldab, w->w2[l], wh.w2, wt.w2, H, kD, k.dh, true, false, 2 * Hl));
    }
    // x_mid = x_in + a2 * y1,  y1 = c_proj(ao) + b,  ao = attention(qkv),  qkv = c_attn(h1) + b,  h1 = LN(x_in)(1 + a0) + a1
    const bool fuse_gate = batched && src16 && g_y16 && g_fuse_gate;   // (batched: dy_attn is its own array, nobody waits on it)
    if (fuse_gate) {   // the attention branch's gate backward rides on the LayerNorm backward that produces its dx
      TRY(ln_bwd(st, kD, n, k.dh, a.x_mid, a.st2, s.mod, (long)mw, o + 3 * kD, o + 4 * kD, k.dx, 1, k.dmod, dh_mlp16,
                 reinterpret_cast<const __bf16*>(a.y1), o + 2 * kD, reinterpret_cast<__bf16*>(dy_attn)));
    } else {
      TRY(ln_bwd(st, kD, n, k.dh, a.x_mid, a.st2, s.mod, (long)mw, o + 3 * kD, o + 4 * kD, k.dx, 1, k.dmod, dh_mlp16));
      TRY(join());   // (c_proj's weight gradient read dy)
      gate_bwd(a.y1, o + 2 * kD, dy_attn);
      LAUNCH_CHECK();
    }
    TRY(fork());
    TRY(wgrad(dy_attn, kD, a.ao, kD, kD, kD, g->proj_w[l], g->proj_b[l]));
    const bool dao16 = g16 && g_attn_mfma;   // (the matrix-core attention backward rounds dao to bf16 anyway)
    TRY(dgrad(dy_attn, kD, w->proj_w[l], wh.proj_w, wt.proj_w, kD, kD, k.dao, false, dao16));
    if (src16) TRY(attn_bwd(st, kD, kNH, n, reinterpret_cast<const __bf16*>(a.qkv), k.dao, reinterpret_cast<__bf16*>(k.dqkv), dao16));
    else TRY(attn_bwd(st, kD, kNH, n, (const float*)a.qkv, k.dao, k.dqkv));
    TRY(fork());
    TRY(wgrad(k.dqkv, 3 * kD, a.h1, kD, 3 * kD, kD, g->attn_w[l], g->attn_b[l]));
    if (batch_side) {   // the layer's five weight gradients, one launch on the side stream; the chain goes on beside it
      hipStream_t sb = st;
      TRY(fused::fork_side(h, st, 2, &sb));
      TRY(wgrad_batch(sb, wj, n_wj, T, kb.part, kb.part_floats));
      batch_busy = true;
      n_wj = 0;
    }
    TRY(dgrad(k.dqkv, 3 * kD, w->attn_w[l], wh.attn_w, wt.attn_w, 3 * kD, kD, k.dh, false, g16));
    if (batched && !batch_side) {   // ... or on the same stream (the next layer overwrites their operands after it)
      TRY(wgrad_batch(st, wj, n_wj, T, k.part, k.part_floats));
      n_wj = 0;
    }
    if (fuse_gate && l > 0) {   // ... and layer l - 1's MLP gate backward rides on this layer's last LayerNorm backward (dy is free: the batch above read it)
      TRY(ln_bwd(st, kD, n, k.dh, a.x_in, a.st1, s.mod, (long)mw, o, o + kD, k.dx, 1, k.dmod, g16,
                 reinterpret_cast<const __bf16*>(s.layer[l - 1].y2), (l - 1) * 6 * kD + 5 * kD, reinterpret_cast<__bf16*>(dy_nxt)));
      dy_ready = true;
    } else {
      TRY(ln_bwd(st, kD, n, k.dh, a.x_in, a.st1, s.mod, (long)mw, o, o + kD, k.dx, 1, k.dmod, g16));
    }
  }
  TRY(join_batch());
  TRY(join());
  if (!use_fused) TRY(fire(SCLDM_GRAD_LAYER, 0));

  // ---- input projection + pos_embed ----
  // Fused route: three independent tails follow the layers - the input projection (reads d x0), the stacked adaLN weight
  // gradient (reads dmod) and the chain d SiLU(c) -> embeddings / timestep MLP.  They are small, latency-bound kernels with
  // disjoint scratch, so the first two go to side streams and the chain stays on `st`.
  hipStream_t s_in = st, s_ada = st;
  if (edge && !dx_out) TRY(fused::fork_side(h, st, 0, &s_in));
  if (use_fused && !ada_stream) TRY(fused::fork_side(h, st, 1, &s_ada));
  if (ada_stream) s_ada = s_ada_l;
  if (edge) {
    TRY(fused::inproj_backward(h, fs.dx, x, n, g->in_w, g->in_b, g->pos_embed, fs.edge_part, s_in));
  } else {
    TRY(linear_wgrad(st, k.dx, kD, x, din, (int)T, kD, din, g->in_w, k, g->in_b));
    if (g->pos_embed) TRY(colsum(st, k.dx, n, kS * kD, (long)kS * kD, g->pos_embed, k));
  }
  if (dx_out) TRY(linear_dgrad(st, k.dx, kD, w->in_w, (int)T, kD, din, dx_out, din, false, k));

  // ---- adaLN Linears: mod_l = SiLU(c) W_l^T + b_l ----
  if (use_fused) {
    // all layers at once: d SiLU(c) = dmod W_all (K = mod_w), d W_all = dmod^T SiLU(c) (+ row sums = bias gradients) into a
    // contiguous (mod_w, D) scratch, scattered to the per-layer gradient tensors by one kernel
    float* dw_all = fs.ada_dw;
    float* db_all = fs.ada_dw + (size_t)mw * kD;
    if (ada_stream) {   // every slice's products are queued on the side stream already: scatter, and wait for the running sum
      TRY(fused::scatter_ada_grads(h, g, dw_all, db_all, s_ada));
      TRY(fused::join_side(h, st, 1));
    } else {
      // (200 output tiles, K = n: no split-K, so this GEMM needs no partial buffer and can run beside the chain below)
      TRY(gemm(s_ada, k.dmod, 1, mw, s.sc, 1, kD, dw_all, kD, mw, kD, n, nullptr, false, nullptr, 0, db_all));
      TRY(fused::scatter_ada_grads(h, g, dw_all, db_all, s_ada));
      TRY(gemm(st, k.dmod, mw, 1, h->ada_t, mw, 1, k.dsc, kD, n, kD, mw, nullptr, false, k.part, k.part_floats));
    }
  }
  // bf16-source route: one cast of dmod, then d SiLU(c) = dmod W_all as ONE split-K product and the per-layer weight
  // gradients straight into their tensors (column slices of dmod as the m-contiguous A operand, row sums = bias gradients).
  // The bf16 copy of dmod (n x mod_w) borrows the dhid | da | db scratch, which is free once the layers are done.
  const bool ada16 = ada16_eligible(h, n, precision);
  if (ada16) {
    __bf16* dmod16 = reinterpret_cast<__bf16*>(k.dhid);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_grid((long)n * mw / 4)), dim3(256), 0, st, k.dmod, dmod16, (long)n * mw);
    LAUNCH_CHECK();
    TRY(bgemm(st, dmod16, mw, true, reinterpret_cast<const __bf16*>(h->ada16), kD, false, k.dsc, kD, n, kD, mw, nullptr, false, k.part, k.part_floats));
    // one product for every layer's adaLN weight gradient when the caller laid the gradient tensors out as the stacked matrix
    // (DiT.grad_segments does: all adaLN weights in layer order, then all adaLN biases): 2 336 tiles of 256 x 256 over k = cells
    // in one launch instead of 25 launches of 96
    bool stacked = L > 0;
    for (int l = 0; l < L && stacked; ++l) {
      const float* nw = l + 1 < L ? g->ada_w[l + 1] : g->fin_ada_w;
      const float* nb = l + 1 < L ? g->ada_b[l + 1] : g->fin_ada_b;
      stacked = nw == g->ada_w[l] + (size_t)6 * kD * kD && nb == g->ada_b[l] + (size_t)6 * kD;
    }
    if (stacked && g_ada_stacked) {
      TRY(bgemm(st, dmod16, mw, false, reinterpret_cast<const __bf16*>(s.sc), kD, false, g->ada_w[0], kD, mw, kD, n, nullptr, false, k.part, k.part_floats,
                g->ada_b[0]));
      TRY(fire(SCLDM_GRAD_ADA, L));
    } else {
      for (int l = 0; l <= L; ++l) {
        TRY(bgemm(st, dmod16 + (size_t)l * 6 * kD, mw, false, reinterpret_cast<const __bf16*>(s.sc), kD, false, l < L ? g->ada_w[l] : g->fin_ada_w, kD,
                  l < L ? 6 * kD : 2 * kD, kD, n, nullptr, false, k.part, k.part_floats, l < L ? g->ada_b[l] : g->fin_ada_b));
        TRY(fire(SCLDM_GRAD_ADA, l));   // (one stream: issue order is completion order)
      }
    }
  }
  for (int l = (use_fused || ada16) ? L + 1 : 0; l <= L; ++l) {
    const int width = l < L ? 6 * kD : 2 * kD;
    const float* dm = k.dmod + (long)l * 6 * kD;
    float* gw = l < L ? g->ada_w[l] : g->fin_ada_w;
    float* gb = l < L ? g->ada_b[l] : g->fin_ada_b;
    const float* wl = l < L ? w->ada_w[l] : w->fin_ada_w;
    TRY(linear_wgrad(st, dm, mw, s.sc, kD, n, width, kD, gw, k, gb));
    TRY(fire(SCLDM_GRAD_ADA, l));
    TRY(linear_dgrad(st, dm, mw, wl, n, width, kD, k.dsc, kD, l > 0, k));
  }
  // Fused route (round 6): everything behind d SiLU(c) except the class tables as two exact-fp32 kernels (cond_bwd.hpp) instead of a chain
  // of nine launches.  MEASURED +-0 to slower (same box, interleaved: 1 024 cells 1.984-1.998 against 1.972 ms per step, 256 cells 1.246
  // against 1.208): the chain's launches already run on three streams beside the stacked adaLN weight gradient, and the two kernels
  // (27 + 28 us: 128 workgroups each, latency-bound) share the chip with that product.  Opt-in: SCLDM_TRAIN_COND_BWD=1.
  static const bool cond_bwd_on = [] { const char* e = getenv("SCLDM_TRAIN_COND_BWD"); return e && e[0] == '1'; }();
  const bool cond2 = use_fused && cond_bwd_on && kD == kCbD;
  if (cond2) {
    hipLaunchKernelGGL(cond_bwd_rows_kernel, dim3(cdiv(n, kCbRows)), dim3(256), 0, st, k.dsc, s.c, s.th, w->t_w2, n, k.dc, k.dth);
  } else {
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_grid((long)n * kD)), dim3(256), 0, st, k.dsc, s.c, k.dc, (long)n * kD);
  }
  LAUNCH_CHECK();

  // ---- class embeddings and the timestep MLP (c = temb + sum emb) ----
  // Fused route (round 5): three independent consumers of d c run beside each other instead of as one chain of ~17 small launches -
  // the class tables on the input-projection stream (partials in an operand-pair array, free after the last layer), d t_w2 on
  // the adaLN stream (split-K partials in the fused weight-gradient scratch, free after the last layer), the timestep MLP's data
  // gradient chain on `st` (the generic split-K scratch).
  hipStream_t s_emb = st, s_tw2 = st;
  float* emb_part = k.part;
  size_t emb_part_floats = k.part_floats;
  Scratch k_tw2 = k;
  static const bool tails_serial = [] { const char* e = getenv("SCLDM_TRAIN_TAILS_SERIAL"); return e && e[0] == '1'; }();   // A/B switch
  const bool par_tails = use_fused && !tails_serial;
  if (par_tails && edge && !dx_out) {
    // (partials in the h1 operand-pair array: 8 KB per cell, free since the last layer's weight-gradient launch earlier on `st`)
    TRY(fused::fork_side(h, st, 0, &s_emb));
    emb_part = reinterpret_cast<float*>(fs.e_h1);
    emb_part_floats = (size_t)T_pad * kD / 2;
  }
  if (par_tails) {
    TRY(fused::fork_side(h, st, 1, &s_tw2));
    k_tw2.part = fs.part + (fs.part_layers > 1 ? (size_t)fs.part_layers * fs.part_floats : 0);   // (the spare block: the layers' blocks may still be read by the deferred reduction)
    k_tw2.part_floats = fs.part_floats;
  }
  for (int c = 0; c < cfg.n_classes; ++c) {
    const int64_t* lab = labels ? labels[c] : nullptr;
    if (n >= 256 && n <= kEmbSeg * kEmbMaxSegs && (size_t)n * kD <= emb_part_floats) {
      // training-size batch: two-level sums (a label shared by most samples - the null token - is not one serial chain)
      HIP_TRY(hipMemsetAsync(g->class_emb[c], 0, (size_t)h->tab_rows[c] * kD * sizeof(float), s_emb));
      hipLaunchKernelGGL(embed_bwd_seg_partial_kernel, dim3(n, kD / 256), dim3(256), 0, s_emb, k.dc, lab, h->tab_rows[c] - 1, n, kD, emb_part);
      hipLaunchKernelGGL(embed_bwd_seg_final_kernel, dim3(n, kD / 256), dim3(256), 0, s_emb, emb_part, lab, h->tab_rows[c] - 1, n, kD, g->class_emb[c]);
    } else if (h->tab_rows[c] > n) {   // more table rows than samples: organise the sum by sample (same order, same bits)
      HIP_TRY(hipMemsetAsync(g->class_emb[c], 0, (size_t)h->tab_rows[c] * kD * sizeof(float), s_emb));
      hipLaunchKernelGGL(embed_bwd_by_sample_kernel, dim3(n, kD / 256), dim3(256), 0, s_emb, k.dc, lab, h->tab_rows[c] - 1, n, kD, g->class_emb[c]);
    } else {
      hipLaunchKernelGGL(embed_bwd_kernel, dim3(h->tab_rows[c], kD / 256), dim3(256), 0, s_emb, k.dc, lab, h->tab_rows[c] - 1, n, kD, g->class_emb[c]);
    }
    LAUNCH_CHECK();
  }
  if (cond2) {
    CondWgradArgs ca{};
    ca.dy[0] = k.dc; ca.x[0] = s.sth; ca.dW[0] = g->t_w2; ca.db[0] = g->t_b2;
    ca.dy[1] = k.dth; ca.x[1] = s.freq; ca.dW[1] = g->t_w0; ca.db[1] = g->t_b0;
    ca.n = n;
    hipLaunchKernelGGL(cond_bwd_wgrad_kernel, dim3(kCbD / 32, kCbD / 32, 2), dim3(256), 0, st, ca);
    LAUNCH_CHECK();
  } else {
    TRY(linear_wgrad(s_tw2, k.dc, kD, s.sth, kD, n, kD, kD, g->t_w2, k_tw2, g->t_b2));
    TRY(linear_dgrad(st, k.dc, kD, w->t_w2, n, kD, kD, k.dsth, kD, false, k));
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_grid((long)n * kD)), dim3(256), 0, st, k.dsth, s.th, k.dth, (long)n * kD);
    LAUNCH_CHECK();
    TRY(linear_wgrad(st, k.dth, kD, s.freq, 256, n, kD, 256, g->t_w0, k, g->t_b0));
  }
  if (s_emb != st && s_emb != s_in) TRY(fused::join_side(h, st, 0));
  if (s_in != st) TRY(fused::join_side(h, st, 0));
  if (s_ada != st) TRY(fused::join_side(h, st, 1));
  if (h->wgrad_reduce_on_side) {   // the layers' deferred weight-gradient reduction (fused::backward_layers)
    h->wgrad_reduce_on_side = false;
    TRY(fused::join_side(h, st, 2));
  }
  if (f16) TRY(fused::unscale_grads(h, g, dx_out, (long)T * din, fs, st));
  TRY(fire(SCLDM_GRAD_END, 0));
  h->grad_events.clear();
  return SCLDM_OK;
}

extern "C" int scldm_fm_mix(const float* x1, const float* x0, const float* t, float* xt, float* ut, int n, int e, void* stream_) {
  if (!x1 || !x0 || !t || !xt || !ut || n < 1 || e < 1) return fail(SCLDM_ERR_SHAPE, "scldm_fm_mix: bad argument");
  const long total = (long)n * e;
  hipLaunchKernelGGL(fm_mix_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream_, x1, x0, t, xt, ut, total, e);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
extern "C" int scldm_fm_loss(const float* pred, const float* ut, float* loss, int n, int e, void* stream_) {
  if (!pred || !ut || !loss || n < 1 || e < 1) return fail(SCLDM_ERR_SHAPE, "scldm_fm_loss: bad argument");
  hipLaunchKernelGGL(fm_loss_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream_, pred, ut, loss, e);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
extern "C" int scldm_fm_loss_bwd(const float* pred, const float* ut, const float* gloss, float* dpred, int n, int e, void* stream_) {
  if (!pred || !ut || !gloss || !dpred || n < 1 || e < 1) return fail(SCLDM_ERR_SHAPE, "scldm_fm_loss_bwd: bad argument");
  const long total = (long)n * e;
  hipLaunchKernelGGL(fm_loss_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream_, pred, ut, gloss, dpred, total, e);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
