// What does the matrix pipe sustain on THIS box under its power budget?  A register-only v_mfma_f32_32x32x16_bf16 loop (no memory
// traffic at all: 2 waves per SIMD, four independent accumulators per wave, operands rotated among eight fragments held in registers)
// with (a) zero operands, (b) uniform random [-1, 1) bf16 operands, (c) N(0, 1)-like operands.  The instruction stream is identical;
// the difference is clock (DVFS).  Prints TFLOP/s of each fill - the random-data figure is the practical ceiling every MFMA-bound
// kernel on this box is priced against in DESIGN.md (the nominal 2.5 PFLOP/s assumes 2.4 GHz).
//   build: hipcc --offload-arch=gfx950 -O3 tests/perf/mfma_power.hip -o tests/perf/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256, 2) void mfma_loop(const bf16x8* __restrict__ frags, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = frags[(i * 2 + 0) * 64 + lane];
    b[i] = frags[(i * 2 + 1) * 64 + lane];
  }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[(i + 2 * u + 1) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;   // keeps the loop alive, never true in practice
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int blocks = 512;   // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  bf16x8* d_frags;
  float* d_out;
  hipMalloc(&d_frags, 8 * 64 * sizeof(bf16x8));
  hipMalloc(&d_out, blocks * 256 * sizeof(float));
  const char* names[3] = {"zero", "uniform[-1,1)", "normal(0,1)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int fill = 0; fill < 3; ++fill) {
      std::vector<__bf16> h(8 * 64 * 8);
      srand(7);
      for (auto& v : h) {
        float x = 0.f;
        if (fill == 1) x = 2.f * rand() / (float)RAND_MAX - 1.f;
        if (fill == 2) { float u1 = (rand() + 1.f) / ((float)RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX; x = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
        v = (__bf16)x;
      }
      hipMemcpy(d_frags, h.data(), h.size() * sizeof(__bf16), hipMemcpyHostToDevice);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      mfma_loop<<<blocks, 256>>>(d_frags, d_out, 2000);   // warm-up
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int k = 0; k < 5; ++k) mfma_loop<<<blocks, 256>>>(d_frags, d_out, iters);   // ~1 s of continuous MFMA: the sustained regime
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = 5.0 * blocks * 4.0 * iters * 16.0 * 2.0 * 32 * 32 * 16;
      printf("fill %-14s: %8.1f ms  %8.1f TFLOP/s  (%.3f of 2.5 PF)\n", names[fill], ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0);
      fflush(stdout);
    }
  return 0;
}
