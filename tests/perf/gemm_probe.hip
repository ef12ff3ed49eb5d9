// Stand-alone timing / ablation harness for the bf16-source GEMM kernels of the generic training path (bgemm.hpp, bgemm8.hpp).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tests/perf/gemm_probe.hip -o tests/perf/gemm_probe
//   tests/perf/gemm_probe [M N K]...
// Every (KC, KC) product is run through bgemm256_kernel<true, true> (the reference for the bitwise check), bgemm8_kernel and the
// PROBE ablations of bgemm8 (wrong results by construction: timing only).  Not part of the product or of the tests.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#include <algorithm>
#include <cmath>
#include <array>
#include "../../scldm_amd/csrc/bgemm4.hpp"
using namespace scldm;
using namespace scldm::train;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

template <typename K> static void set_lds(K kern, int bytes) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); }

struct Variant {
  const char* name;
  std::function<void(int, const BGemmArgs&)> launch;
  bool exact;   // the result must equal the register-staged kernel's bit for bit
  bool c16 = false;   // bf16 output (C16) instead of fp32
  bool split = false; // mc: row sums split over the tile columns
  bool t128 = false;  // 128 x 128 tiles, 256 threads (bgemm_kernel / bgemm4_kernel)
  std::vector<float> us;
};
template <typename K>
static Variant variant(const char* name, K kern, int smem, bool exact, int threads = 512) {
  set_lds(kern, smem);
  Variant v{name, [kern, smem, threads](int blocks, const BGemmArgs& g) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, 0, g); }, exact, false, {}};
  v.t128 = threads == 256;
  return v;
}
#define PROBE(name, bits, exact) variant(name, bgemm8_kernel<bits>, kBGemm8Lds, exact)

int main(int argc, char** argv) {
  const bool mc = argc > 1 && !strcmp(argv[1], "mc");   // weight-gradient orientation: both operands contiguous along m (A[k][m], B[k][n])
  const bool small = argc > 1 && !strcmp(argv[1], "small");   // the 128 x 128-tile kernels (bgemm_kernel against bgemm4_kernel)
  if (mc || small) { --argc; ++argv; }
  std::vector<std::array<int, 3>> shapes;
  for (int i = 1; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])});
  const bool explicit_shapes = !shapes.empty();
  if (shapes.empty()) shapes = {{16384, 3072, 1024}, {16384, 1024, 1024}, {16384, 1024, 2736}, {16384, 2736, 1024}, {16384, 1024, 8192}};
  std::vector<Variant> vs;
  if (small) {
    if (!explicit_shapes) shapes = {{4096, 1024, 1024}, {4096, 1024, 5472}, {4096, 1024, 3072}, {2048, 1024, 2736}, {8192, 512, 512}, {1000, 520, 328}};
    vs.push_back(variant("bgemm_kernel<KC,KC> (register staging, 128 tiles)", bgemm_kernel<true, true>, kBGemmLds, true, 256));
    vs.push_back(variant("bgemm4_kernel<0> (LDS-DMA, results through LDS)", bgemm4_kernel<0>, kBGemm4Lds, true, 256));
    vs.push_back(variant("bgemm4_kernel<1> (LDS-DMA, element-wise stores)", bgemm4_kernel<1>, kBGemm4Lds, true, 256));
    vs.push_back(variant("bf16 out: bgemm_kernel", bgemm_kernel<true, true>, kBGemmLds, false, 256)); vs.back().c16 = true;
    vs.push_back(variant("bf16 out: bgemm4_kernel<0>", bgemm4_kernel<0>, kBGemm4Lds, false, 256)); vs.back().c16 = true;
    vs.push_back(variant("bf16 out: bgemm4_kernel<1>", bgemm4_kernel<1>, kBGemm4Lds, false, 256)); vs.back().c16 = true;
  } else if (mc) {
    if (!explicit_shapes) shapes = {{3072, 1024, 16384}, {2732, 1024, 16384}, {1024, 2732, 16384}, {1000, 520, 4000}};
    vs.push_back(variant("bgemm256_kernel<MC,MC> (register staging)", bgemm256_kernel<false, false>, kBGemm2Lds, true));
    vs.push_back(variant("bgemm8_kernel<MC,MC>, element-wise epilogue", bgemm8_kernel<1024, false, false>, kBGemm8Lds, true));
    vs.push_back(variant("bgemm8_kernel<MC,MC>, vector epilogue", bgemm8_kernel<0, false, false>, kBGemm8Lds, true));
    vs.push_back(variant("bgemm8_kernel<MC,MC> ew, row sums split over tn", bgemm8_kernel<1024, false, false>, kBGemm8Lds, false)); vs.back().split = true;
    vs.push_back(variant("bgemm256_kernel<MC,MC>, no row sums", bgemm256_kernel<false, false>, kBGemm2Lds, false)); vs.back().c16 = true;
    vs.push_back(variant("bgemm8_kernel<MC,MC> vector, no row sums", bgemm8_kernel<0, false, false>, kBGemm8Lds, false)); vs.back().c16 = true;
    vs.push_back(variant("  probe: no staging in the loop", bgemm8_kernel<1024 | 1, false, false>, kBGemm8Lds, false));
    vs.push_back(variant("  probe: no fragment reads", bgemm8_kernel<1024 | 2, false, false>, kBGemm8Lds, false));
    vs.push_back(variant("  probe: MFMAs + barriers only", bgemm8_kernel<1024 | 3, false, false>, kBGemm8Lds, false));
    vs.push_back(variant("  probe: staging + barriers only", bgemm8_kernel<1024 | 6, false, false>, kBGemm8Lds, false));
    vs.push_back(variant("  probe: reads + barriers only", bgemm8_kernel<1024 | 5, false, false>, kBGemm8Lds, false));
  } else {
  vs.push_back(variant("bgemm256_kernel<KC,KC> (register staging)", bgemm256_kernel<true, true>, kBGemm2Lds, true));
  vs.push_back(variant("bgemm8_kernel<0> (vector epilogue)", bgemm8_kernel<0>, kBGemm8Lds, true));
  vs.push_back(PROBE("8 phases, staging ahead of the reads", 0, true));
  vs.push_back(variant("bf16 out: bgemm256_kernel", bgemm256_kernel<true, true>, kBGemm2Lds, false)); vs.back().c16 = true;
  vs.push_back(PROBE("8 phases, element-wise fp32 epilogue", 1024, true));
  vs.push_back(PROBE("bf16 out: bgemm8, transposed blocks 8-byte", 0, false)); vs.back().c16 = true;
  vs.push_back(PROBE("bf16 out: bgemm8, through LDS, 16-byte row stores", 8192, false)); vs.back().c16 = true;
  vs.push_back(PROBE("bf16 out: bgemm8, paired 4-byte stores", 1024, false)); vs.back().c16 = true;
  vs.push_back(PROBE("bf16 out: bgemm8, 2-byte stores", 1024 | 256, false)); vs.back().c16 = true;
  vs.push_back(PROBE("bf16 out: barriers + epilogue only", 7, false)); vs.back().c16 = true;
  vs.push_back(PROBE("  probe: no stores", 512, false));
  vs.push_back(PROBE("  probe: barriers only (no stores)", 512 | 7, false));
  vs.push_back(PROBE("  probe: barriers + fp32 epilogue", 7, false));
  vs.push_back(PROBE("  probe: MFMAs + barriers, no stores", 512 | 3, false));
  vs.push_back(PROBE("  8ph probe: MFMAs + barriers only", 3, false));
  vs.push_back(PROBE("  8ph probe: no staging in the loop", 1, false));
  }
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    const int lda = mc ? (M + 7) / 8 * 8 : (K + 7) / 8 * 8, ldb = mc ? (N + 7) / 8 * 8 : lda;
    std::vector<unsigned short> ha((size_t)(mc ? K : M) * lda, 0), hb((size_t)(mc ? K : N) * ldb, 0);
    unsigned x = 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 32768.0f - 1.0f; };
    if (mc) {   // (the padding columns M..lda stay zero, as the producers of the real arrays leave them)
      for (int k = 0; k < K; ++k) for (int m = 0; m < M; ++m) ha[(size_t)k * lda + m] = f2bf(rnd());
      for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) hb[(size_t)k * ldb + n] = f2bf(rnd());
    } else {
      for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) ha[(size_t)m * lda + k] = f2bf(rnd());
      for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) hb[(size_t)n * ldb + k] = f2bf(rnd());
    }
    __bf16 *A, *B; float *C0, *C1;
    CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&B, hb.size() * 2));
    CK(hipMalloc(&C0, (size_t)M * N * 4)); CK(hipMalloc(&C1, (size_t)M * N * 4));
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    float *RS0 = nullptr, *RS1 = nullptr;
    CK(hipMalloc(&RS0, (size_t)M * 4)); CK(hipMalloc(&RS1, (size_t)M * 4));
    BGemmArgs g{};
    if (mc) g.rowsum = RS0;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.kchunk = (K + 63) / 64 * 64; g.splits = 1;
    const int TS = small ? 128 : 256;
    g.tiles_m = (M + TS - 1) / TS; g.tiles_n = (N + TS - 1) / TS;
    const int tiles = g.tiles_m * g.tiles_n;
    g.per_xcd = (tiles + 7) / 8;
    const int blocks = 8 * g.per_xcd;
    const double flop = 2.0 * M * N * K;
    printf("M=%d N=%d K=%d (%d tiles of %d x %d, %.2f rounds of %d workgroup slots)\n", M, N, K, tiles, TS, TS, tiles / (small ? 512.0 : 256.0), small ? 512 : 256);
    // exactness first (one launch each), then interleaved timing rounds (median of 7 rounds x 10 launches per variant)
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    g.C = C0;
    CK(hipMemset(C0, 0xff, (size_t)M * N * 4));
    vs[0].launch(blocks, g);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
    g.C = C1;
    if (mc) g.rowsum = RS1;
    std::vector<float> r0(M), r1(M);
    if (mc) CK(hipMemcpy(r0.data(), RS0, (size_t)M * 4, hipMemcpyDeviceToHost));
    std::vector<long> bad(vs.size(), -1);
    for (size_t v = 1; v < vs.size(); ++v) {
      if (!vs[v].exact) continue;
      CK(hipMemset(C1, 0xff, (size_t)M * N * 4));
      if (mc) CK(hipMemset(RS1, 0xff, (size_t)M * 4));
      for (int rep = 0; rep < 3; ++rep) vs[v].launch(blocks, g);   // (repeated: a race that corrupts one launch in three shows)
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
      long b = 0;
      for (size_t i = 0; i < h0.size(); ++i) b += memcmp(&h0[i], &h1[i], 4) != 0;
      bad[v] = b;
      if (mc) {   // row sums: other summation order than the register-staged kernel's - compared relatively
        CK(hipMemcpy(r1.data(), RS1, (size_t)M * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (int m = 0; m < M; ++m) { worst = std::max(worst, (double)fabsf(r0[m] - r1[m])); scale = std::max(scale, (double)fabsf(r0[m])); }
        printf("  [%s] row sums: max |diff| %.3g against max |value| %.3g\n", vs[v].name, worst, scale);
      }
    }
    if (!mc) {   // bf16 outputs: the three kernels must agree bit for bit
      std::vector<unsigned short> r0((size_t)M * N), r1((size_t)M * N);
      bool first = true;
      for (size_t v = 0; v < vs.size(); ++v) {
        if (!vs[v].c16 || strstr(vs[v].name, "only")) continue;
        BGemmArgs gv = g;
        gv.C16 = reinterpret_cast<__bf16*>(C1);
        CK(hipMemset(C1, 0xff, (size_t)M * N * 4));
        vs[v].launch(blocks, gv);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(first ? r0.data() : r1.data(), C1, r0.size() * 2, hipMemcpyDeviceToHost));
        if (!first) {
          long b = 0;
          for (size_t i = 0; i < r0.size(); ++i) b += r0[i] != r1[i];
          bad[v] = b;
        }
        first = false;
      }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : vs) v.us.clear();
    for (int i = 0; i < 30; ++i) vs[1].launch(blocks, g);   // clocks
    for (int round = 0; round < 7; ++round)
      for (auto& v : vs) {
        BGemmArgs gv = g;
        if (v.c16 && !mc) gv.C16 = reinterpret_cast<__bf16*>(C1);
        if (v.c16 && mc) gv.rowsum = nullptr;
        if (v.split) { gv.rowsum = reinterpret_cast<float*>(C0); gv.rowsum_split = 1; }   // (scratch: C0 is not read any more)
        v.launch(blocks, gv);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) v.launch(blocks, gv);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        v.us.push_back(ms * 100.f);
      }
    CK(hipGetLastError());
    for (size_t v = 0; v < vs.size(); ++v) {
      auto u = vs[v].us;
      std::sort(u.begin(), u.end());
      const float med = u[u.size() / 2];
      printf("  %-46s %8.1f us (%6.1f..%6.1f) %6.0f TFLOP/s", vs[v].name, med, u.front(), u.back(), flop / med / 1e6);
      if (bad[v] >= 0) printf("  %ld elements differ", bad[v]);
      printf("\n");
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C0)); CK(hipFree(C1));
  }
  return 0;
}
