// The opaque handle behind `scldm_dit*` (shared by api.hip and train_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/scldm_hip.h"

struct scldm_dit {
  scldm_dit_config cfg;
  int n_chunks[2], half[2];   // [FT-1]: full 128-unit SwiGLU chunks and the optional trailing 64-unit chunk (FT=2)
  int mod_w;
  bool loaded;
  bool fused;   // shape served by the fused inference kernels (otherwise only the scldm_dit_train_* path)
  void* stream[4][2];  // [precision][FT-1] packed weight streams, [layer][wave][unit][tile] (+ ring over-read slack); NULL = shape unused
  void* wfinal[4];   // [precision] packed final_layer.linear
  float *b_qkv, *b_proj;  // (n_layer,768), (n_layer,256)
  float *w0t, *b0, *w2t, *b2;      // timestep MLP (transposed weights)
  float* emb;                      // concatenated class tables
  int emb_row0[SCLDM_MAX_CLASSES];
  float *ada_t, *ada_b;            // (256, mod_w), (mod_w)
  void* ada_x3;                    // the same matrix as split-bf16 B fragments [mod_w / 32][16][64][hi 8 | lo 8] (adaln_x3_kernel)
  float *in_wt, *in_b, *pos;       // (Din,256), (256), (16,256)
  float* in_w;                     // (256,Din) as stored
  float* fin_b;                    // (Din)
  // timing hook
  bool timing;
  std::vector<hipEvent_t> ev;
  size_t ev_used;
  int lpl;         // DiT layers per fused-kernel launch (1..4)
  bool adaln_rowtile = false;  // SCLDM_ADALN_ROWTILE: adaln_x3_kernel<4> instead of the 128 x 128-block kernel above 128 rows (A/B)
  bool adaln_exact = false;  // SCLDM_ADALN_EXACT: adaln_mfma_kernel (exact fp32) for every precision policy (A/B)
  bool adaln_valu = false;   // SCLDM_ADALN_VALU: adaln_all_kernel instead of adaln_mfma_kernel (A/B)
  int cfg1_direct = 0;  // SCLDM_OPT_CFG1_DIRECT (scldm_dit_set_option)
  int tail_split = 0;   // SCLDM_OPT_TAIL_SPLIT: the partial last round of a trunk launch runs as 32-token tiles (api.hip: trunk; measured slower)
  int n_cu = 256;       // compute units of the device (rounds of 2 x n_cu tiles)
  int groups;      // tile groups per layer launch (SCLDM_GROUPS, read once at create)
  int dbg_layer;   // SCLDM_DBG_LAYER (read once at create; -1: middle layer)
  int tab_rows[SCLDM_MAX_CLASSES];  // rows of each class table (vocab + has_null_row)
  int* label_err;  // device: sticky count of clamped out-of-range labels
  float* d_ls = nullptr;   // device, 16 x 32 bit: loss-scale state of the fp16 backward (train_fused.hip: LossScaleState); persists across steps
  float* found_inf = nullptr;   // caller-owned device float (scldm_dit_train_set_found_inf): 1.0 after a backward that produced non-finite gradients
  int* d_fp16_stats;  // device: [0] packed fp16 weights beyond +-65504, [1] below the smallest normal, [2] non-zero values packed
  int* d_plan;     // device: [0] t is uniform, [1] conditioning rows of the chosen plan (scldm_dit_forward_cfg, t_stride 2)
  // packing job table + fingerprint state (scldm_dit_load_weights / scldm_dit_refresh_weights)
  void* d_jobs;    // device PackJob[n_jobs]
  int n_jobs, job_blocks, jobs_cap;
  void* d_fp_src;  // device FpSrc[n_fp]
  int n_fp, fp_cap;
  unsigned long long* d_fp_state;  // [0] running accumulator, [1] fingerprint of the packed weights
  int* d_dirty;    // [0] re-pack flag written by the compare kernel, [1] force flag
  hipStream_t side[3];     // secondary streams for tile-group launches (created on first use)
  bool wgrad_reduce_on_side = false;   // fused training backward: the deferred weight-gradient reduction is in flight on side[2] (joined by scldm_dit_train_backward)
  hipEvent_t fork_ev, join_ev[3];
  hipEvent_t wg_ev[2] = {nullptr, nullptr};   // fused training, small batches: layer l's weight-gradient launches (side stream) are done with operand-pair set l & 1
  hipEvent_t bwd_pack_ev = nullptr;   // the training step's backward weight stream is packed (second pack launch of fused::prepare)
  // fused sampler: the conditioning of evaluation e + 1 (timestep rows + adaLN projection: independent of the state z) runs on its own
  // stream beside the trunk of evaluation e, into the other of two buffer sets (scldm_sample_ode)
  hipStream_t cond_stream = nullptr;
  hipEvent_t ev_cond[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr}, ev_ready = nullptr;
  // round 6: the adaLN vectors of EVERY evaluation of a fixed-grid solve in one batched pass ahead of the loop (they depend on t and the
  // labels only): SCLDM_COND_ALL=0 switches it off; cond_all = [mod | silu | split rows] for n_evals x n_rows rows, grown on demand
  bool cond_all_on = true;
  void* cond_all = nullptr;
  size_t cond_all_bytes = 0;
  bool cond_ahead = false;  // SCLDM_COND_AHEAD=1: opt-in (measured +0.5 % at 1 024 joint-conditioned cells, -0.8 % at 512, -0.2 % at 4 096: off)
  int force_ft, force_x3_ft, force_x3_ntt;
  bool small_ntt = true;   // 32-token tiles for launches of at most 256 of them (SCLDM_SMALL_NTT=0: off)
  unsigned long long* dbg;  // device buffer for phase stamps (debug builds)
  // fused training path (train_fused.hip)
  void* bwd_stream;         // bf16 backward weight stream [layer][8 waves][kBwdUnitsLayer][512] (+ ring slack); allocated on first use
  std::vector<const void*> table_key;  // every device pointer of the scldm_dit_weights the job / fingerprint tables were built from
  bool tables_built;
  bool train_fused;         // SCLDM_TRAIN_FUSED (read once at create; 0 keeps the base shape on the generic training path)
  int wgrad_splits;         // SCLDM_WGRAD_SPLITS (read once at create; 0: default)
  bool bwd_dbg;             // SCLDM_BWD_DBG (read once at create): print the backward kernel's phase stamps
  int32_t* iota;            // identity row index 0..iota_n-1 (the training forward's conditioning rows are the samples themselves)
  int iota_n;
  void* d_tjobs;            // device PackJob table of the training step's subset
  int n_tjobs, tjob_blocks;
  bool partial_pack;        // the last pack refreshed only some precisions' streams: the next inference refresh is unconditional
  // generic training path with bf16 operands (train_api.hip, bgemm.hpp): per-step bf16 copies of the layers' five weight matrices
  void* w16;                // [layer][attn_w | proj_w | w1 | w2 | cproj] bf16, allocated on first use
  size_t w16_layer_elems;
  void* wt16;               // the same matrices transposed ([in][out], rows padded to a multiple of 8): k-contiguous operands of the data gradients
  size_t wt16_layer_elems;
  int n_cast_first;         // cast jobs that run ahead of the forward (adaLN + the first layers); the rest runs beside it
  bool cast_side_busy;      // the side-stream cast of this step has not been joined yet
  bool wt16_live;           // the current step's forward refreshed the transposed copies (large batches only)
  void* ada16;              // [mod_w][D] bf16: every adaLN Linear's weight stacked (one GEMM for all layers' modulation vectors)
  float* ada_ball;          // [mod_w] fp32: their biases, stacked
  void* d_cast_jobs;        // device CastJob table (rebuilt when the weights' device pointers change)
  int n_cast_jobs;
  std::vector<const void*> w16_key;
  // gradient-ready events of the NEXT scldm_dit_train_backward (scldm_dit_train_set_grad_events): recorded on the call's stream
  struct GradEvent { hipEvent_t ev; int kind, layer; bool fired; };
  std::vector<GradEvent> grad_events;
  bool wgrad_batch;         // SCLDM_WGRAD_BATCH (read once at create; 0: per-product weight-gradient launches with split-K, the round-2 scheme)
  bool bf16_sources;        // SCLDM_TRAIN_BF16_SOURCES (read once at create; 0 keeps fp32 activations + hgemm_kernel)
};

// api.hip internals shared with train_fused.hip
int scldm_build_pack_tables(scldm_dit* h, const scldm_dit_weights* w, hipStream_t st);   // (re)builds the device job tables if `w` changed
int scldm_run_pack(scldm_dit* h, bool force, unsigned prec_mask, hipStream_t st);
