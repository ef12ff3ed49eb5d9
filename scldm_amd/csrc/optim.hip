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

// ---- round 6: the launch table in DEVICE memory, the hyper-parameters that change per step in device memory, the EMA of the reference's
// trainer in the same pass.  (a) Any number of tensors is one launch (the DiT-L shape's ~250 tensors were three); (b) `hyper` is read
// by the kernel, so a captured HIP graph follows a learning-rate schedule (the reference's LambdaLR, src/scldm/models.py:603-605) and
// the EMA schedule (ema_pytorch 0.7.7 as LatentDiffusion uses it, models.py:446-453 + on_train_batch_end, models.py:83-87) without
// re-capture - ADVICE r5: lr was a by-value kernel argument frozen at capture; (c) the EMA's `ema.lerp_(p, 1 - decay)` (or the copy
// of its warm-up phase) reads p from the registers of the AdamW update instead of a second pass over both models.
struct TableTensor { float* p; const float* g; float* m; float* v; float* e; long long n; };
struct TableBlock { int tensor, chunk; };
struct TableArgs {
  const TableTensor* tensors;
  const TableBlock* blocks;
  const float* step;
  const float* found_inf;
  const float* hyper;             // device float[4] or null: lr, weight_decay, EMA mode (0 none | 1 copy | 2 lerp), EMA lerp weight
  double lr, beta1, beta2, eps, weight_decay;
  int maximize;
  const float* clip_coef;         // device, or null: every gradient is read times *clip_coef (clip_grad_norm_'s clip_coef_clamped)
};
// Total L2 norm of every gradient of the table and the clipping coefficient torch.nn.utils.clip_grad_norm_ derives from it
// (torch/nn/utils/clip_grad.py: clip_coef = max_norm / (total_norm + 1e-6), clamped to <= 1; error_if_nonfinite = False), as TWO plain
// launches: one workgroup per 4 096-element chunk (the update's own map) writes its sum of squares, one workgroup sums the partials in a
// fixed order.  ws = [norm, coef, -, - | one partial per chunk].  Measured inside the step at 1 024 cells (2 400 chunks): a single kernel
// with a last-workgroup ticket costs 68 us (one same-address atomic and two agent-scope fences per workgroup), <= 512 striding workgroups
// 27 us (three dependent round trips per chunk with little occupancy to hide them), two-level tickets 95 us; two launches without
// atomics or fences are the cheapest form.
__global__ __launch_bounds__(256) void grad_sq_partials_kernel(const TableArgs a, float* __restrict__ part) {
  __shared__ float red[256];
  const TableBlock blk = a.blocks[blockIdx.x];
  const TableTensor tt = a.tensors[blk.tensor];
  const float* __restrict__ g = tt.g;
  const long long n = tt.n;
  const int base = blk.chunk * kChunk;
  float s = 0.f;
  const bool vec = (reinterpret_cast<size_t>(g) & 15) == 0;
#pragma unroll
  for (int it = 0; it < kChunk / (256 * 4); ++it) {
    const long long i = base + (it * 256 + (int)threadIdx.x) * 4;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
      s += gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2] + gv[3] * gv[3];
    } else {
      for (long long e = i; e < (i + 4 < n ? i + 4 : n); ++e) s += g[e] * g[e];
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// ONE workgroup: the partials summed in double (thread-strided, then a fixed tree) -> ws[0] = norm, ws[1] = clip_coef_clamped
__global__ __launch_bounds__(256) void grad_norm_final_kernel(float* __restrict__ ws, int n_blocks, float max_norm) {
  __shared__ double redd[256];
  const float* part = ws + 4;
  double t = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += 256) t += (double)part[i];
  redd[threadIdx.x] = t;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) redd[threadIdx.x] += redd[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(redd[0]);
    const float coef = max_norm / (norm + 1e-6f);
    ws[0] = norm;
    ws[1] = coef != coef ? coef : (coef < 1.0f ? coef : 1.0f);   // (a NaN norm - non-finite gradients - gives a NaN coefficient, as torch's clamp does)
  }
}
// torch.lerp (ATen/native/Lerp.h): weight < 0.5 ? start + weight (end - start) : end - (end - start) (1 - weight); ATen's device code is
// built with floating-point contraction on, so each form is ONE fma (tests/test_gpu_train.py compares the bits with Tensor.lerp_)
__device__ __forceinline__ float torch_lerp(float start, float end, float w, float omw) {
  const float d = end - start;
  return fabsf(w) < 0.5f ? __builtin_fmaf(w, d, start) : __builtin_fmaf(-d, omw, end);
}
__global__ __launch_bounds__(256) void adamw_table_kernel(const TableArgs a) {
  const bool skip = a.found_inf && *a.found_inf != 0.f;
  const int ema_mode = a.hyper ? (int)a.hyper[2] : 0;
  if (skip && ema_mode == 0) return;
  const TableBlock blk = a.blocks[blockIdx.x];
  const TableTensor tt = a.tensors[blk.tensor];
  const int base = blk.chunk * kChunk;
  const long long n = tt.n;
  float* __restrict__ p = tt.p;
  const float* __restrict__ g = tt.g;
  float* __restrict__ m = tt.m;
  float* __restrict__ v = tt.v;
  float* __restrict__ ema = ema_mode ? tt.e : nullptr;
  if (skip && !ema) return;
  const double lr = a.hyper ? (double)a.hyper[0] : a.lr, wd = a.hyper ? (double)a.hyper[1] : a.weight_decay;
  const float ew = a.hyper ? a.hyper[3] : 0.f, eomw = 1.0f - ew;
  const double step = (double)*a.step;
  const float bc1 = (float)(1.0 - pow(a.beta1, step)), bc2 = (float)(1.0 - pow(a.beta2, step));
  const float step_size = (float)(lr / (double)bc1), bc2_sqrt = sqrtf(bc2), lr_wd = (float)(lr * wd);
  const float omb1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, omb2 = (float)(1.0 - a.beta2), eps = (float)a.eps;
  const bool clip = a.clip_coef != nullptr;
  const float cc = clip ? *a.clip_coef : 1.0f;
  auto one = [&](float pv, float gv, float& mv, float& vv) {     // (the arithmetic of adamw_kernel, bit for bit)
    if (clip) gv = gv * cc;      // clip_grad_norm_ multiplies every gradient by the clamped coefficient, 1.0 included
    if (a.maximize) gv = -gv;
    pv = pv - lr_wd * pv;
    mv = mv + omb1 * (gv - mv);
    vv = b2 * vv + omb2 * gv * gv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    return pv - step_size * mv / denom;
  };
  auto ema_of = [&](float ev, float pv) { return ema_mode == 1 ? pv : torch_lerp(ev, pv, ew, eomw); };
  const bool vec = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) | reinterpret_cast<size_t>(v) |
                     reinterpret_cast<size_t>(ema)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < kChunk / (256 * 4); ++it) {
    const long long i = base + (it * 256 + (int)threadIdx.x) * 4;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
      if (!skip) {
        f32x4 mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
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
      }
      if (ema) {
        f32x4 ev = ema_mode == 1 ? pv : *reinterpret_cast<f32x4*>(ema + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) ev[e] = ema_of(ev[e], pv[e]);
        *reinterpret_cast<f32x4*>(ema + i) = ev;
      }
    } else {
      for (long long e = i; e < (i + 4 < n ? i + 4 : n); ++e) {
        float pe = p[e];
        if (!skip) {
          float me = m[e], ve = v[e];
          pe = one(pe, g[e], me, ve);
          p[e] = pe;
          m[e] = me;
          v[e] = ve;
        }
        if (ema) ema[e] = ema_of(ema_mode == 1 ? pe : ema[e], pe);
      }
    }
  }
}
}  // namespace optim
}  // namespace scldm

extern "C" size_t scldm_adamw_table_bytes(const scldm_adamw_entry* e, int count) {
  using namespace scldm::optim;
  if (!e || count < 1) return 0;
  size_t blocks = 0;
  for (int i = 0; i < count; ++i)
    if (e[i].n > 0) blocks += (size_t)((e[i].n + kChunk - 1) / kChunk);
  return sizeof(TableTensor) * (size_t)count + sizeof(TableBlock) * blocks;
}
extern "C" int scldm_adamw_table_build(const scldm_adamw_entry* e, float* const* ema, int count, void* table_host, size_t bytes, int* n_blocks) {
  using namespace scldm::optim;
  if (!e || count < 1 || !table_host || !n_blocks) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_table_build: bad argument");
  if (bytes < scldm_adamw_table_bytes(e, count)) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_table_build: the table buffer is too small");
  TableTensor* tt = reinterpret_cast<TableTensor*>(table_host);
  TableBlock* tb = reinterpret_cast<TableBlock*>(tt + count);
  long long blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (e[i].n > 0 && (!e[i].p || !e[i].g || !e[i].m || !e[i].v)) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_table_build: tensor %d has a NULL pointer", i);
    if (e[i].n > 0x7fffffffLL) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_table_build: tensor %d has more than 2^31 elements", i);
    tt[i] = TableTensor{e[i].p, e[i].g, e[i].m, e[i].v, ema ? ema[i] : nullptr, e[i].n};
    for (long long c = 0; c * kChunk < e[i].n; ++c) tb[blocks++] = TableBlock{i, (int)c};
  }
  if (blocks > 0x7fffffffLL) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_table_build: too many workgroups");
  *n_blocks = (int)blocks;
  return SCLDM_OK;
}
// Only the per-tensor records (the head of the table: count x 48 bytes) depend on addresses; the workgroup map behind them depends on
// the tensor SIZES alone.  A caller whose gradient buffers move between steps (autograd allocating a fresh flat buffer) re-uploads the
// records - 12 KB for the DiT-L shape's ~250 tensors - instead of the whole table (0.9 MB of workgroup map at 459 M parameters).
extern "C" size_t scldm_adamw_clip_workspace_bytes(int n_blocks) { return n_blocks > 0 ? sizeof(float) * (4 + (size_t)n_blocks) : 0; }
extern "C" size_t scldm_adamw_table_records_bytes(int count) { return count > 0 ? sizeof(scldm::optim::TableTensor) * (size_t)count : 0; }
extern "C" int scldm_adamw_table_update(const scldm_adamw_entry* e, float* const* ema, int count, void* records_host, size_t bytes) {
  using namespace scldm::optim;
  if (!e || count < 1 || !records_host || bytes < sizeof(TableTensor) * (size_t)count) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_table_update: bad argument");
  TableTensor* tt = reinterpret_cast<TableTensor*>(records_host);
  for (int i = 0; i < count; ++i) {
    if (e[i].n > 0 && (!e[i].p || !e[i].g || !e[i].m || !e[i].v)) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_table_update: tensor %d has a NULL pointer", i);
    if (e[i].n > 0x7fffffffLL) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_table_update: tensor %d has more than 2^31 elements", i);
    tt[i] = TableTensor{e[i].p, e[i].g, e[i].m, e[i].v, ema ? ema[i] : nullptr, e[i].n};
  }
  return SCLDM_OK;
}
extern "C" int scldm_adamw_table_step(const scldm_adamw_launch* l, void* stream_) {
  using namespace scldm::optim;
  if (!l || !l->table || !l->step || l->count < 1 || l->n_blocks < 0) return fail(SCLDM_ERR_SHAPE, "scldm_adamw_table_step: bad argument");
  hipStream_t st = (hipStream_t)stream_;
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, l->step, l->found_inf);
  if (l->n_blocks) {
    TableArgs a{};
    a.tensors = reinterpret_cast<const TableTensor*>(l->table);
    a.blocks = reinterpret_cast<const TableBlock*>(a.tensors + l->count);
    a.step = l->step; a.found_inf = l->found_inf; a.hyper = l->hyper;
    a.lr = l->lr; a.beta1 = l->beta1; a.beta2 = l->beta2; a.eps = l->eps; a.weight_decay = l->weight_decay; a.maximize = l->maximize;
    if (l->max_grad_norm > 0.f && l->clip_ws) {
      hipLaunchKernelGGL(grad_sq_partials_kernel, dim3(l->n_blocks), dim3(256), 0, st, a, l->clip_ws + 4);
      hipLaunchKernelGGL(grad_norm_final_kernel, dim3(1), dim3(256), 0, st, l->clip_ws, l->n_blocks, l->max_grad_norm);
      a.clip_coef = l->clip_ws + 1;
    }
    hipLaunchKernelGGL(adamw_table_kernel, dim3(l->n_blocks), dim3(256), 0, st, a);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(SCLDM_ERR_HIP, "scldm_adamw_table_step: %s", hipGetErrorString(err));
  return SCLDM_OK;
}

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
