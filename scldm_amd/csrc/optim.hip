// AdamW over every parameter tensor of a model in ONE launch (round 5).  The training step of the reference is Lightning +
// torch.optim.AdamW (experiments/configs/training/default.yaml, src/scldm/models.py:configure_optimizers); torch's fused multi-tensor
// AdamW runs the base DiT's 84 tensors as four launches of ~150 workgroups (~200 us per step at 9.75 M parameters: latency-bound on a
// 256-CU part).  This kernel gives every 4 096-element chunk of every tensor its own workgroup (2 400 workgroups for the base DiT):
// one pass at HBM speed.  Arithmetic = torch's fused AdamW (ATen/native/cuda/fused_adam_utils.cuh, ADAM_MODE::ADAMW, fp32 opmath):
//   p -= lr wd p;  m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step count t lives on the device (capturable in a HIP graph); a non-zero *found_inf (GradScaler protocol; the fp16 training
// backward's overflow flag) skips the update AND the step increment.
#include <hip/hip_runtime.h>

#include "api_common.hpp"
#include "common.hpp"

#pragma clang fp contract(off)

namespace scldm {
namespace optim {

constexpr int kMaxTensors = 88;    // per launch (by-value table: 88 x 40 B + scalars < 4 KB of kernel arguments; the base DiT trains 84)
constexpr int kChunk = 4096;       // elements per workgroup
struct AdamArgs {
  float* p[kMaxTensors];
  const float* g[kMaxTensors];
  float* m[kMaxTensors];
  float* v[kMaxTensors];
  int n[kMaxTensors];
  int first[kMaxTensors + 1];      // first workgroup of each tensor
  int count;
  const float* step;               // device: t (already incremented for this step)
  const float* found_inf;          // device, may be null
  double lr, beta1, beta2, eps, weight_decay;   // (double like torch's kernel arguments: the bias corrections are formed in double)
  int maximize;
};

__global__ void step_inc_kernel(float* __restrict__ step, const float* __restrict__ found_inf) {
  if (!found_inf || *found_inf == 0.f) *step += 1.0f;
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamArgs a) {
  if (a.found_inf && *a.found_inf != 0.f) return;
  int lo = 0, hi = a.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int t = lo, base = ((int)blockIdx.x - a.first[t]) * kChunk, n = a.n[t];
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const double step = (double)*a.step;
  const float bc1 = (float)(1.0 - pow(a.beta1, step)), bc2 = (float)(1.0 - pow(a.beta2, step));
  const float step_size = (float)(a.lr / (double)bc1), bc2_sqrt = sqrtf(bc2), lr_wd = (float)(a.lr * a.weight_decay);
  const float omb1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, omb2 = (float)(1.0 - a.beta2), eps = (float)a.eps;
  auto one = [&](float pv, float gv, float& mv, float& vv) {
    if (a.maximize) gv = -gv;
    pv = pv - lr_wd * pv;                              // decoupled weight decay
    mv = mv + omb1 * (gv - mv);                        // lerp(m, g, 1 - b1)
    vv = b2 * vv + omb2 * gv * gv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    return pv - step_size * mv / denom;
  };
  const bool vec = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) | reinterpret_cast<size_t>(v)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < kChunk / (256 * 4); ++it) {
    const int i = base + (it * 256 + (int)threadIdx.x) * 4;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float me = mv[e], ve = vv[e];
        pv[e] = one(pv[e], gv[e], me, ve);
        mv[e] = me;
        vv[e] = ve;
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (int e = i; e < min(i + 4, n); ++e) {
        float me = m[e], ve = v[e];
        p[e] = one(p[e], g[e], me, ve);
        m[e] = me;
        v[e] = ve;
      }
    }
  }
}

static_assert(sizeof(AdamArgs) <= 4000, "kernel arguments must stay under 4 KB");
}  // namespace optim
}  // namespace scldm

extern "C" int scldm_adamw_step(const scldm_adamw_entry* e, int count, float* step, const float* found_inf, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int maximize, void* stream_) {
  using namespace scldm::optim;
  if (!e || count < 1 || !step) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_step: bad argument");
  hipStream_t st = (hipStream_t)stream_;
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, step, found_inf);
  for (int i0 = 0, i = 0; i0 < count; i0 = i) {   // (i0 = the first entry this launch has not consumed: empty tensors are skipped, never re-visited)
    AdamArgs a{};
    int blocks = 0, k = 0;
    for (i = i0; i < count && k < kMaxTensors; ++i) {
      if (e[i].n <= 0) continue;
      if (!e[i].p || !e[i].g || !e[i].m || !e[i].v) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_step: tensor %d has a NULL pointer", i);
      if (e[i].n > 0x7fffffffLL) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_step: tensor %d has more than 2^31 elements", i);
      a.p[k] = e[i].p; a.g[k] = e[i].g; a.m[k] = e[i].m; a.v[k] = e[i].v; a.n[k] = (int)e[i].n;
      a.first[k] = blocks;
      blocks += (int)((e[i].n + kChunk - 1) / kChunk);
      ++k;
    }
    if (!k) continue;
    a.first[k] = blocks;
    a.count = k;
    a.step = step; a.found_inf = found_inf;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.maximize = maximize;
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, st, a);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(SCLDM_ERR_HIP, "scldm_adamw_step: %s", hipGetErrorString(err));
  return SCLDM_OK;
}
