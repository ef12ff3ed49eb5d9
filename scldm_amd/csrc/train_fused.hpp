// Fused training path of the base DiT shape (n_embed 256, 8 heads, seq_len 16) with bf16 operands: host interface between
// train_api.hip (conditioning, final layer, input projection: the small GEMM-based pieces) and train_fused.hip (the fused
// forward with a training record, the fused backward layer, the batched weight-gradient GEMMs).
#pragma once
#include <functional>
#include <hip/hip_runtime.h>

#include "dit_handle.hpp"

namespace scldm {
namespace fused {

// Activation record of one training step: (L+1) fp32 residuals + 2L bf16 branch outputs per token row of 256 features,
// in the forward kernel's tile layout.  n must be a multiple of 4 (whole 64-token tiles).
struct Record {
  float* x;        // [L+1][T*256]
  __bf16* y1;      // [L][T*256]
  __bf16* y2;      // [L][T*256]
  size_t bytes;
};
Record carve_record(const scldm_dit* h, int n, void* base);

// Per-step scratch: residual hand-off between forward launches, the gradient of the residual stream (tile layout), the
// operand pairs of one layer's weight-gradient GEMMs and their split-K partial sums.
struct Scratch {
  float* handoff;   // [T*256]
  float* dx;        // [T*256] tile layout
  int32_t* ridx;    // [n]
  __bf16 *e_h1, *e_dqkv, *e_ao, *e_dy1, *e_h2, *e_da, *e_db, *e_hid, *e_dy2;   // operand-pair set 0
  __bf16* e_set1;   // second set of the nine arrays (same order and sizes, contiguous), or nullptr: lets layer l's weight gradients run on a
                    // side stream beside layer l-1's backward kernel when that kernel leaves CUs idle (<= 128 tiles, i.e. <= 512 cells)
  size_t e_set_elems;   // elements from e_h1 to the end of e_dy2 (= offset of an array in set 1 relative to set 0's, from e_set1 - e_h1)
  float* part;      // split-K partials of one layer's five weight gradients + two bias gradients
  float* ada_dw;    // (mod_w, 256) + (mod_w): gradient of the stacked adaLN Linears before it is scattered to the per-layer tensors
  float* edge_part; // per-wave partials of the final-layer / input-projection weight gradients
  float* dout_s;    // fp16 policy: the loss-scaled copy of d loss / d output (n x 16 x n_embed_input)
  size_t part_floats;   // floats of ONE layer's partial block behind `part`
  int part_layers;      // blocks behind `part`: n_layer where the per-layer reductions are deferred to the end of the layer loop, else 1
  float* scale;     // fp16 policy: the handle's loss-scale state ([0] S, a power of two decided on device from max |dout|, [1] 1 / S, ...)
  size_t bytes;
};
Scratch carve_scratch(const scldm_dit* h, int n, void* base);

constexpr int kMaxFp16TrainLayers = 16;   // the fp16 backward un-scales every gradient tensor through a by-value pointer table
bool eligible(const scldm_dit* h, int n, int precision);   // shape, precision (bf16 | fp16) and batch served by the fused path

// refresh the packed forward (bf16) and backward weight streams from the live parameters
int prepare_tables(scldm_dit* h, const scldm_dit_weights* w, hipStream_t st);   // pack-job tables, side streams, events (no kernel work)
int prepare(scldm_dit* h, const scldm_dit_weights* w, hipStream_t st, int precision);   // forks the re-pack (of that precision's streams) onto a side stream
int prepare_join(scldm_dit* h, hipStream_t st);
int backward_join(scldm_dit* h, hipStream_t st);                           // `st` waits for it (before the first packed copy is read)
// trunk forward (input projection .. final layer) with the record; mod = (n, mod_w) adaLN vectors
int forward(scldm_dit* h, const float* x, const float* mod, int n, float* out, const Record& rec, const Scratch& s, hipStream_t st, int precision);
// plain [T][256] fp32 <-> tile layout
int to_tile(const float* plain, float* tile, int n, hipStream_t st);
int to_plain(const float* tile, float* plain, int n, hipStream_t st);
// all L layers, last to first: s.dx (tile layout) holds d loss / d x_L on entry and d loss / d x_0 on return; dmod gets the
// gradients of the layers' adaLN vectors; g receives attn_w/attn_b/proj_w/proj_b/w1/w2/cproj of every layer
// after_layer(l), when given, runs right after layer l's kernel is queued (its dmod slice is final once that kernel ends): the caller
// forks the per-layer adaLN products onto a side stream there (round 5)
int backward_layers(scldm_dit* h, const scldm_dit_grads* g, const float* mod, float* dmod, int n, const Record& rec, const Scratch& s,
                    hipStream_t st, int precision, const std::function<int(int)>& after_layer = {});

// The two ends of the backward as single kernels on the tile layout (n_embed_input 8 / 16 / 32):
//   final_backward: LayerNorm + Linear of the final layer: dx (tile layout), d(shift, scale) into dmod, d fin_w, d fin_b
//   inproj_backward: d in_w, d in_b and (gpos != NULL) d pos_embed from d x0 (tile layout) and the input latents x
bool edge_kernels_available(const scldm_dit* h);
int final_backward(scldm_dit* h, const float* x_last, const float* mod, const float* dout, const float* fin_w, int n, float* dx,
                   float* dmod, float* gw, float* gb, float* part, hipStream_t st);
int inproj_backward(scldm_dit* h, const float* dx, const float* x, int n, float* gw, float* gb, float* gpos, float* part, hipStream_t st);

// independent tails of the backward run next to each other: side stream k starts after what `st` holds / `st` waits for it
int fork_side(scldm_dit* h, hipStream_t st, int k, hipStream_t* out);
int join_side(scldm_dit* h, hipStream_t st, int k);

// fp16 policy: loss scaling of the backward.  scale_dout: S = 2^floor(log2(8 / max |dout|)) (device side, no host read), dout_s = S dout;
// unscale: every tensor of `g` (and dx_out, when given) times 1 / S in one launch over a by-value pointer table; the same launch counts
// the non-finite values it meets (overflow guard: the next scale_dout lowers the scale, h->found_inf lets the optimizer skip the step).
int scale_dout(scldm_dit* h, const float* dout, long n_elem, const Scratch& s, hipStream_t st);
int unscale_grads(scldm_dit* h, const scldm_dit_grads* g, float* dx_out, long dx_elems, const Scratch& s, hipStream_t st);

// (mod_w, 256) stacked weight gradient + (mod_w) stacked bias gradient -> g->ada_w[l] / ada_b[l] / fin_ada_w / fin_ada_b
int scatter_ada_grads(scldm_dit* h, const scldm_dit_grads* g, const float* dw_all, const float* db_all, hipStream_t st);

}  // namespace fused
}  // namespace scldm
