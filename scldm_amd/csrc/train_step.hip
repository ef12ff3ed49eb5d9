// The flow-matching training step of the reference around the DiT call, as device code with no host decision inside (round 6):
//   Transport.sample          src/scldm/transport/transport.py:97-108   x0 ~ N(0, I), t ~ U[0, 1]
//   ICPlan.plan               src/scldm/transport/path.py:148-151       xt = t x1 + (1 - t) x0, ut = x1 - x0
//   label dropout             src/scldm/nnets.py:395-402 (mutually_exclusive: one class drawn per call, one mask per row),
//                             nnets.py:440-452 (joint: one mask per row over every class)
//   loss                      transport.py:122-150 + models.py:651      mean_b mean_e (pred - ut)^2, and its gradient w.r.t. pred
// The eager mirror spent ~20 launches of ~5 us on these (profiles/r5_train_b1024_graph_timeline.txt: 86 us ahead of the
// conditioning chain, 25 us around the loss) and made three decisions on the host that a captured HIP graph froze at capture time
// (the class of a multi-class mutually_exclusive model, the label mask, t).  Here: ONE kernel prepares the batch from a
// device-resident Philox state (seed, step counter), ONE kernel forms the loss and its gradient and advances the counter.
// RNG: Philox4x32-10 (Salmon et al., SC'11), counter = (index, step counter, stream tag), key = seed - results depend on
// (seed, step, element) only, not on the launch geometry; RNG-dependent by nature and outside the bit-parity claim:
// tested statistically, and everything downstream of the draws is compared bit for bit with the composed route on the same draws.
#include <hip/hip_runtime.h>

#include "api_common.hpp"
#include "common.hpp"

#pragma clang fp contract(off)

namespace scldm {
namespace fmstep {

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t w) { return ((w >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1), 24 bits
__device__ __forceinline__ float u01_closed0(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }   // [0, 1): torch.rand's range

enum : uint32_t { TAG_X0 = 0x78300000u, TAG_ROW = 0x726f7700u, TAG_STEP = 0x73746570u };

struct PrepArgs {
  const float* x1;                  // (n, e)
  const int64_t* labels_in[8];      // per class (sorted-name order): (n) labels, or nullptr = class absent from `condition`
  int null_token[8];                // class vocabulary size = its null token
  int n_classes, strategy;          // strategy 0 mutually_exclusive, 1 joint
  int drop;                         // apply label dropout (training mode / force_drop_ids)
  float p_drop;
  const unsigned long long* rng;    // device: [0] seed, [1] step counter
  float* t;                         // (n)
  float* x0;                        // (n, e) or nullptr
  float* xt; float* ut;             // (n, e)
  int64_t* labels_out;              // (n_classes, n): what the model sees
  int n, e;
};

// thread = 4 consecutive elements of (n, e); the first thread of a row also writes t and the row's labels
__global__ __launch_bounds__(256) void fm_prepare_kernel(const PrepArgs a) {
  const unsigned long long seed = a.rng[0], ctr = a.rng[1];
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), s0 = (uint32_t)ctr, s1 = (uint32_t)(ctr >> 32);
  const long q = blockIdx.x * 256l + threadIdx.x, i = q * 4, total = (long)a.n * a.e;
  if (i >= total) return;
  const int row = (int)(i / a.e);
  // per-row draws: t and the dropout decision (every thread of the row forms the same t: one Philox call, no exchange)
  const U4 r = philox4((uint32_t)row, s0, s1, TAG_ROW, k0, k1);
  const float tv = u01_closed0(r.x);
  const U4 g = philox4((uint32_t)q, s0 ^ (uint32_t)(q >> 32), s1, TAG_X0, k0, k1);
  // Box-Muller, both branches: four normals from four uniforms
  const float r0 = sqrtf(-2.0f * logf(u01(g.x))), r1 = sqrtf(-2.0f * logf(u01(g.z)));
  float s_a, c_a, s_b, c_b;
  sincosf(6.283185307179586f * u01(g.y), &s_a, &c_a);
  sincosf(6.283185307179586f * u01(g.w), &s_b, &c_b);
  const float z[4] = {r0 * c_a, r0 * s_a, r1 * c_b, r1 * s_b};
  const float om = 1.0f - tv;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (i + j >= total) break;
    const float x1v = a.x1[i + j];
    const float p1 = tv * x1v, p0 = om * z[j];     // every intermediate rounded on its own: the eager reference's bits (fm_mix_kernel)
    a.xt[i + j] = p1 + p0;
    a.ut[i + j] = x1v - z[j];
    if (a.x0) a.x0[i + j] = z[j];
  }
  if (i % a.e == 0) {
    a.t[row] = tv;
    const bool dropped = a.drop && u01_closed0(r.y) < a.p_drop;       // torch.rand(n) < cfg_dropout_prob: one mask per batch row
    int chosen = -1;                                                   // joint: every class
    if (a.strategy == 0) {      // one class per CALL among those present (nnets.py:395): the same draw in every row
      int avail = 0;
      for (int c = 0; c < a.n_classes; ++c) avail += a.labels_in[c] != nullptr;
      const U4 s = philox4(0u, s0, s1, TAG_STEP, k0, k1);
      int pick = avail > 1 ? (int)(s.x % (uint32_t)avail) : 0;
      for (int c = 0; c < a.n_classes; ++c)
        if (a.labels_in[c] && pick-- == 0) chosen = c;
    }
    for (int c = 0; c < a.n_classes; ++c) {
      const bool live = a.labels_in[c] && (a.strategy == 1 || c == chosen) && !dropped;
      a.labels_out[(size_t)c * a.n + row] = live ? a.labels_in[c][row] : (int64_t)a.null_token[c];
    }
  }
}

// One WAVE per batch row, eight rows per workgroup: loss_rows[b] = mean_e (pred - ut)^2 in fm_loss_kernel's summation order (its thread
// i takes elements i, i + 256, ...; its four waves are summed ((0 + 1) + (2 + 3)): lane l here carries the four partial sums of threads
// l, 64 + l, 128 + l, 192 + l and reduces each with the same butterfly), dpred = (1/n) (2/e) (pred - ut) (what autograd hands back through
// mean() and fm_loss_bwd_kernel: gloss[b] = 1/n); the LAST workgroup to finish (ticket) sums the rows in a fixed order into loss_mean and
// advances the Philox step counter - deterministic, one launch.  (One workgroup per row was 1 024 same-address ticket atomics at ~20 ns
// each: 26 us for a 1 MB pass, profiles/r6d; eight rows per workgroup: n / 8 of them.)
constexpr int kLossRows = 8;
__global__ __launch_bounds__(256) void fm_loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ ut, float* __restrict__ loss_rows,
                                                           float* __restrict__ loss_mean, float* __restrict__ dpred, int n, int e,
                                                           unsigned int* __restrict__ ticket, unsigned long long* __restrict__ rng) {
  __shared__ float red[4];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float gl = 1.0f / (float)n, k = 2.0f / (float)e;
  for (int r = wave; r < kLossRows; r += 4) {
    const long b = (long)blockIdx.x * kLossRows + r;
    if (b >= n) break;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < e; base += 256) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = base + 64 * q + lane;
        if (i < e) {
          const float d = pred[b * e + i] - ut[b * e + i];
          s[q] = fmaf(d, d, s[q]);
          dpred[b * e + i] = gl * k * d;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o);
    if (lane == 0) loss_rows[b] = ((s[0] + s[1]) + (s[2] + s[3])) / (float)e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += __builtin_nontemporal_load(loss_rows + i);   // (written by other workgroups: bypass this CU's cache)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    *loss_mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
    *ticket = 0;
    if (rng) rng[1] += 1ull;
  }
}

}  // namespace fmstep
}  // namespace scldm

using namespace scldm::fmstep;

extern "C" int scldm_fm_prepare(const float* x1, const int64_t* const* labels_in, const int* null_tokens, int n_classes, int strategy, int drop,
                                float p_drop, const unsigned long long* rng_state, int n, int e, float* t, float* x0, float* xt, float* ut,
                                int64_t* labels_out, void* stream_) {
  if (!x1 || !labels_in || !null_tokens || !rng_state || !t || !xt || !ut || !labels_out || n < 1 || e < 1)
    return fail(SCLDM_ERR_SHAPE, "scldm_fm_prepare: bad argument");
  if (n_classes < 1 || n_classes > 8) return fail(SCLDM_ERR_SHAPE, "scldm_fm_prepare: 1..8 condition classes (got %d)", n_classes);
  if (strategy != 0 && strategy != 1) return fail(SCLDM_ERR_SHAPE, "scldm_fm_prepare: strategy 0 (mutually_exclusive) or 1 (joint)");
  if (e % 4) return fail(SCLDM_ERR_SHAPE, "scldm_fm_prepare: the row length must be a multiple of 4 (got %d)", e);
  PrepArgs a{};
  a.x1 = x1; a.n_classes = n_classes; a.strategy = strategy; a.drop = drop; a.p_drop = p_drop; a.rng = rng_state;
  a.t = t; a.x0 = x0; a.xt = xt; a.ut = ut; a.labels_out = labels_out; a.n = n; a.e = e;
  int present = 0;
  for (int c = 0; c < n_classes; ++c) { a.labels_in[c] = labels_in[c]; a.null_token[c] = null_tokens[c]; present += labels_in[c] != nullptr; }
  if (!present) return fail(SCLDM_ERR_SHAPE, "scldm_fm_prepare: condition holds none of the model's classes");
  if (strategy == 1 && present != n_classes) return fail(SCLDM_ERR_SHAPE, "scldm_fm_prepare: the joint strategy needs every class (nnets.py:449)");
  const long quads = ((long)n * e + 3) / 4;
  hipLaunchKernelGGL(fm_prepare_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

extern "C" int scldm_fm_loss_grad(const float* pred, const float* ut, int n, int e, float* loss_rows, float* loss_mean, float* dpred,
                                  unsigned int* ticket, unsigned long long* rng_state, void* stream_) {
  if (!pred || !ut || !loss_rows || !loss_mean || !dpred || !ticket || n < 1 || e < 1) return fail(SCLDM_ERR_SHAPE, "scldm_fm_loss_grad: bad argument");
  hipLaunchKernelGGL(fm_loss_grad_kernel, dim3((n + kLossRows - 1) / kLossRows), dim3(256), 0, (hipStream_t)stream_, pred, ut, loss_rows, loss_mean, dpred, n,
                     e, ticket, rng_state);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

// The whole optimisation step of LatentDiffusion.training_step (src/scldm/models.py:628-663) + Lightning's backward and
// optimizer step, for a batch of latents: batch preparation -> DiT forward with the training record -> loss and its gradient ->
// DiT backward (every parameter gradient into `grads`) -> [AdamW + EMA when `opt` is given].  One C call, kernel launches and
// event records only; every decision that depends on the step (draws, class choice, learning rate, EMA schedule) is read from
// device memory, so the call can be captured once in a HIP graph and replayed.
extern "C" int scldm_dit_train_step(scldm_dit* h, const scldm_dit_weights* w, const scldm_dit_grads* grads, const float* x1,
                                    const int64_t* const* labels_in, const int* null_tokens, int n_classes, int strategy, float p_drop,
                                    unsigned long long* rng_state, int n, int precision, const scldm_train_step_buffers* b,
                                    const scldm_adamw_launch* opt, void* stream) {
  if (!h || !w || !grads || !x1 || !b || !b->t || !b->xt || !b->ut || !b->pred || !b->dpred || !b->labels || !b->loss_rows || !b->loss_mean ||
      !b->ticket || !b->saved || !b->ws)
    return fail(SCLDM_ERR_SHAPE, "scldm_dit_train_step: bad argument");
  const int e = b->row_elems;
  int rc = scldm_fm_prepare(x1, labels_in, null_tokens, n_classes, strategy, 1, p_drop, rng_state, n, e, b->t, b->x0, b->xt, b->ut, b->labels, stream);
  if (rc) return rc;
  const int64_t* lab[8];
  for (int c = 0; c < n_classes; ++c) lab[c] = b->labels + (size_t)c * n;
  rc = scldm_dit_train_forward(h, w, b->xt, b->t, lab, n, b->pred, precision, b->saved, b->ws, stream);
  if (rc) return rc;
  rc = scldm_fm_loss_grad(b->pred, b->ut, n, e, b->loss_rows, b->loss_mean, b->dpred, b->ticket, rng_state, stream);
  if (rc) return rc;
  rc = scldm_dit_train_backward(h, w, grads, b->xt, lab, b->dpred, n, nullptr, precision, b->saved, b->ws, stream);
  if (rc) return rc;
  if (opt) rc = scldm_adamw_table_step(opt, stream);
  return rc;
}
