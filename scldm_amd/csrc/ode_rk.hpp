// Dormand-Prince 5(4) state arithmetic on the device (round 5): the stage combinations, the error ratio and the dense-output
// polynomial of the adaptive solver behind the reference's default `sample_ode()` (src/scldm/transport/transport.py:324-331,
// integrators.py:100-112 -> torchdiffeq.odeint(method="dopri5")).  The host driver (scldm_amd/transport: Sampler._sample_dopri5)
// keeps the step-size control; these kernels replace ~40 elementwise torch launches per step by four.  Every kernel performs the
// SAME fp32 operations in the SAME order as the host-composed expressions they replace (separately rounded multiplies and adds:
// contraction is off), so trajectories are unchanged.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

#pragma clang fp contract(off)

namespace scldm {
namespace rk {

constexpr int kMaxK = 7;
struct CombArgs {
  const float* k[kMaxK];
  float c[kMaxK];     // h * tableau weight, already rounded to fp32 (structural zeros are not passed)
  int n_k;
};

// out = (((y0 + c0 k0) + c1 k1) + ...)            (y0 == nullptr: the sum alone, first term exact)
__global__ __launch_bounds__(256) void combine_kernel(float* __restrict__ out, const float* __restrict__ y0, const CombArgs a, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 acc = y0 ? reinterpret_cast<const f32x4*>(y0)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < kMaxK; ++j)
    if (j < a.n_k) {
      const f32x4 t = reinterpret_cast<const f32x4*>(a.k[j])[i] * a.c[j];
      acc = (j == 0 && !y0) ? t : acc + t;
    }
  reinterpret_cast<f32x4*>(out)[i] = acc;
}

// err = sum_j c_j k_j;  q = err / (atol + rtol * max(|y0|, |y1|));  partial[block] = sum q^2 (double); error_final_kernel (ONE workgroup,
// the next launch) adds the partials in block order and writes mean(q^2) to *out (deterministic: no floating-point atomics).  Round 6:
// two plain launches instead of a last-block ticket - the ticket cost an agent-scope fence per workgroup and the last workgroup's thread 0
// fetched up to 1 022 partials one after the other: 150 us per step of the default sampler (profiles/r6_dopri5_*).
__global__ __launch_bounds__(256) void error_kernel(const float* __restrict__ y0, const float* __restrict__ y1, const CombArgs a, long n, float atol,
                                                    float rtol, double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float e = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxK; ++j)
      if (j < a.n_k) {
        const float t = a.k[j][i] * a.c[j];
        e = j == 0 ? t : e + t;
      }
    const float den = atol + rtol * fmaxf(fabsf(y0[i]), fabsf(y1[i]));
    const double q = (double)(e / den);
    s += q * q;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// the partials fetched by the whole workgroup at once, then added one after the other in block order (the order of the one-kernel form: same bits)
__global__ __launch_bounds__(256) void error_final_kernel(const double* __restrict__ partial, unsigned n_part, long n, double* __restrict__ out) {
  __shared__ double p[1024];
  for (unsigned b = threadIdx.x; b < n_part; b += 256) p[b] = partial[b];
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (unsigned b = 0; b < n_part; ++b) t += p[b];
    *out = t / (double)n;
  }
}

// Quartic dense output of an accepted step (torchdiffeq's _interp_fit: y0, y1, the mid-point combination, f(t0, y0) = k[0] and
// f(t1, y1) = k[6]):  c1 = h fa;  c2 = h (fb - 4 fa) - 11 y0 - 5 y1 + 16 ymid;  c3 = h (5 fa - 3 fb) + 18 y0 + 14 y1 - 32 ymid;
// c4 = 2 h (fb - fa) - 8 (y1 + y0) + 16 ymid - evaluated left to right as written
__global__ __launch_bounds__(256) void dense_kernel(const float* __restrict__ y0, const float* __restrict__ y1, const CombArgs mid, const float* __restrict__ fa_,
                                                    const float* __restrict__ fb_, float h, float two_h, long n, float* __restrict__ c1,
                                                    float* __restrict__ c2, float* __restrict__ c3, float* __restrict__ c4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a0 = y0[i], a1 = y1[i], fa = fa_[i], fb = fb_[i];
  float ym = a0;
#pragma unroll
  for (int j = 0; j < kMaxK; ++j)
    if (j < mid.n_k) ym = ym + mid.k[j][i] * mid.c[j];
  c1[i] = h * fa;
  c2[i] = ((h * (fb - 4.0f * fa) - 11.0f * a0) - 5.0f * a1) + 16.0f * ym;
  c3[i] = ((h * (5.0f * fa - 3.0f * fb) + 18.0f * a0) + 14.0f * a1) - 32.0f * ym;
  c4[i] = (two_h * (fb - fa) - 8.0f * (a1 + a0)) + 16.0f * ym;
}

// out = (((c0 + s1 c1) + s2 c2) + s3 c3) + s4 c4      (s_j = s^j rounded to fp32 by the caller)
__global__ __launch_bounds__(256) void poly_kernel(float* __restrict__ out, const float* __restrict__ c0, const float* __restrict__ c1,
                                                   const float* __restrict__ c2, const float* __restrict__ c3, const float* __restrict__ c4, float s1,
                                                   float s2, float s3, float s4, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  auto ld = [&](const float* p) { return reinterpret_cast<const f32x4*>(p)[i]; };
  f32x4 t = ld(c0) + ld(c1) * s1;
  t = t + ld(c2) * s2;
  t = t + ld(c3) * s3;
  t = t + ld(c4) * s4;
  reinterpret_cast<f32x4*>(out)[i] = t;
}

}  // namespace rk
}  // namespace scldm
