// Negative-binomial draw fused into the decoder's second softmax pass:
//   counts ~ Poisson( Gamma(concentration = theta, rate = theta / mu) )        clamp(gamma, max = 1e8)
// - the Gamma-Poisson mixture scvi-tools' NegativeBinomial.sample() draws after TransformerVAE.decode in the reference
// (src/scldm/models.py:819 `nb.sample()`; distribution built at src/scldm/vae.py:87).  The reference does it with three
// eager passes over the (2B, G) mean / dispersion tensors (Gamma sample, clamp, Poisson sample) after writing both; here the
// pass that normalises the logits draws the count in registers, so neither mu nor theta ever reaches HBM.
// Transcendentals of the samplers run on the hardware units (v_log / v_exp / v_sqrt / v_cos, ~1 ulp) since round 6.
// RNG: Philox4x32-10 (Salmon et al., SC'11), counter = (element index, draw index), key = (seed) - a result depends only
// on (seed, element), not on the launch geometry.  Samplers: Marsaglia-Tsang (2000) for Gamma (shape < 1 through the
// U^(1/a) boost), inversion by multiplication for Poisson(lambda < 10), Hoermann's PTRS (1993) above.  RNG-dependent by
// nature: outside the bit-parity claim; tested statistically against the moments and against torch's
// Gamma / Poisson samplers.
#pragma once
#include "common.hpp"

namespace scldm {

struct Philox {
  uint32_t ctr[4], key[2], out[4];
  int have;   // unread words of `out`
  __device__ __forceinline__ void init(unsigned long long seed, unsigned long long element, uint32_t stream) {
    ctr[0] = (uint32_t)element; ctr[1] = (uint32_t)(element >> 32); ctr[2] = 0; ctr[3] = stream;
    key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
    have = 0;
  }
  __device__ __forceinline__ void refill() {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
      const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
      c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    ++ctr[2];
    have = 4;
  }
  __device__ __forceinline__ uint32_t next() {
    if (have == 0) refill();
    --have;
    // static indexing (a dynamically indexed register array would go to scratch)
    return have == 3 ? out[0] : have == 2 ? out[1] : have == 1 ? out[2] : out[3];
  }
  __device__ __forceinline__ float uniform() { return ((next() >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1), 24 bits
  __device__ __forceinline__ float normal() {   // Box-Muller, one value per call (the sine branch is not kept: registers)
    const float u1 = uniform(), u2 = uniform();
    // hardware transcendentals (round 6: v_log_f32 / v_sqrt_f32 / v_cos_f32, ~1 ulp; the argument of v_cos_f32 is in revolutions, so
    // u2 needs no 2 pi and no range reduction) instead of libm's ~30-instruction logf / cosf: the draw was a quarter of decode_sample
    return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);   // -2 ln 2 log2(u1)
  }
};
// natural log / exp / power on the hardware units (inputs here are positive and far from the subnormal range)
__device__ __forceinline__ float nb_log(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float nb_exp(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }

// Gamma(shape a, scale 1), Marsaglia & Tsang: d = a - 1/3, c = 1 / sqrt(9 d); accept d v when log u < x^2/2 + d - d v + d log v
__device__ __forceinline__ float gamma_draw(Philox& g, float a) {
  float boost = 1.0f;
  if (a < 1.0f) {
    boost = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(g.uniform()) * __builtin_amdgcn_rcpf(a));   // u^(1/a)
    a += 1.0f;
  }
  const float d = a - (1.0f / 3.0f), c = rsqrtf(9.0f * d);
  for (int it = 0; it < 64; ++it) {
    const float x = g.normal();
    float v = 1.0f + c * x;
    if (v <= 0.f) continue;
    v = v * v * v;
    const float u = g.uniform();
    if (u < 1.0f - 0.0331f * (x * x) * (x * x) || nb_log(u) < 0.5f * x * x + d * (1.0f - v + nb_log(v))) return d * v * boost;
  }
  return d * boost;   // 64 rejections in a row: probability < 1e-80
}

__device__ __forceinline__ float poisson_draw(Philox& g, float lam) {
  if (!(lam > 0.f)) return 0.f;
  if (lam < 10.0f) {   // multiply uniforms until the product drops below exp(-lambda)
    const float L = nb_exp(-lam);
    float p = g.uniform();
    int k = 0;
    while (p > L && k < 200) { p *= g.uniform(); ++k; }
    return (float)k;
  }
  // PTRS: transformed rejection with squeeze (Hoermann 1993); the form numpy / torch use for large lambda
  const float slam = __builtin_amdgcn_sqrtf(lam), loglam = nb_log(lam);
  const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
  const float invalpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.0f);
  for (int it = 0; it < 256; ++it) {
    const float U = g.uniform() - 0.5f, V = g.uniform();
    const float us = 0.5f - fabsf(U);
    const float k = floorf((2.0f * a / us + b) * U + lam + 0.43f);
    if (us >= 0.07f && V <= vr) return k;
    if (k < 0.f || (us < 0.013f && V > us)) continue;
    if (nb_log(V * invalpha / (a / (us * us) + b)) <= -lam + k * loglam - lgammaf(k + 1.0f)) return k;   // (the slow path: ~14 % of the large-lambda draws)
  }
  return floorf(lam);
}

__device__ __forceinline__ float nb_draw(unsigned long long seed, unsigned long long element, float mu, float theta) {
  if (!(mu > 0.f)) return 0.f;
  Philox g;
  g.init(seed, element, 0x5c1dbu);
  const float rate_inv = fmaxf(mu, 1e-8f) / theta;   // Gamma(concentration theta, rate theta / mu) = Gamma(theta, 1) * mu / theta
  const float lam = fminf(gamma_draw(g, theta) * rate_inv, 1e8f);
  return poisson_draw(g, lam);
}

}  // namespace scldm
