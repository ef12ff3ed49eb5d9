/*
 * scldm_hip.h - C ABI of libscldm_hip.so: the MI355X (gfx950) implementation of the scLDM
 * latent-diffusion hot path.  Plain pointers and sizes only; no torch types.
 *
 * The reference (czi-ai/scldm) has no FFI: its seam is Hydra `_target_` instantiation of
 * nn.Modules.  Each entry point below therefore cites the reference
 * Python method it replaces; the ctypes binding a maintainer would add is shown in INTEGRATION.md
 * and implemented in scldm_amd/_lib.py.
 *
 * Conventions
 *  - every pointer named `d_*` or documented "device" is a device (HBM) pointer; tensors are
 *    contiguous, row-major, fp32 unless stated; labels are int64 (torch.long);
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it;
 *  - no hidden allocation in hot calls: the caller provides `ws` of at least
 *    scldm_dit_workspace_bytes(...) bytes (256-byte aligned);
 *  - return value: 0 on success, SCLDM_ERR_* (<0) otherwise; scldm_last_error() returns a
 *    thread-local message;
 *  - a handle belongs to one device and one host thread at a time.
 */
#ifndef SCLDM_HIP_H
#define SCLDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCLDM_OK 0
#define SCLDM_ERR_SHAPE (-1)   /* unsupported / inconsistent shape or argument */
#define SCLDM_ERR_HIP (-2)     /* HIP runtime error (text in scldm_last_error) */
#define SCLDM_ERR_STATE (-3)   /* e.g. weights not loaded */

#define SCLDM_MAX_CLASSES 8

#define SCLDM_PREC_FP32 0  /* exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): parity path, <=1e-4 vs reference */
#define SCLDM_PREC_BF16 1  /* bf16 operands, fp32 accumulate / LN / softmax / residual: throughput path */
#define SCLDM_PREC_BF16X3 2 /* split-bf16: every GEMM operand as hi + lo bf16, three v_mfma_f32_32x32x16_bf16 per k-step (hi*hi, hi*lo,
                             * lo*hi), fp32 accumulate - the arithmetic class of the reference's
                             * torch.set_float32_matmul_precision("high") (experiments/scripts/inference.py:26); ~1e-5 vs fp32,
                             * inside the 1e-4 parity gate at 5x the exact-fp32 MFMA rate.  Fused DiT inference kernels; the training /
                             * generic entry points serve it with the exact-fp32 GEMM route (same parity class). */

#define SCLDM_PREC_FP16 3  /* fp16 operands (v_mfma_f32_32x32x16_f16), fp32 accumulate / LN / softmax / residual: 10 mantissa bits = the
                            * TF32 arithmetic the reference runs under set_float32_matmul_precision("high") (inference.py:26,
                            * train_ldm.py:18) at the bf16 MFMA rate.  Fused DiT inference kernels; range |w| <= 65 504 checked at
                            * pack time (scldm_dit_fp16_stats).  Training / generic entry points: exact-fp32 route. */

#define SCLDM_METHOD_EULER 0
#define SCLDM_METHOD_HEUN 1

typedef struct scldm_dit scldm_dit;

/* Mirror of scldm.nnets.DiT.__init__ kwargs (reference src/scldm/nnets.py:219-234).
 * Classes are listed in SORTED NAME ORDER (the reference iterates sorted(class_vocab_sizes), :393,:404,:438). */
typedef struct {
  int n_embed;        /* 256 for the fused inference kernels; any multiple of 256 (<= 2048) on the generic scldm_dit_train_* path */
  int n_embed_input;  /* latent channels, <= 64 */
  int n_layer;
  int n_head;         /* 8 (head_dim 32) fused; head_dim 32 or 64 on the generic path */
  int seq_len;        /* 16 */
  int hidden_dim;     /* SwiGLU hidden (684 for n_embed 256, multiple_of 4), layers.py:165-167 */
  float layernorm_eps;
  int n_classes;
  int class_vocab[SCLDM_MAX_CLASSES]; /* vocab size per class; the null token index equals it */
  int has_null_row;   /* 1: class tables have vocab+1 rows (cfg_dropout_prob > 0, nnets.py:241-243); 0: vocab rows - every entry point
                       * that would need a null token (unconditional CFG pass, unselected class) then returns SCLDM_ERR_SHAPE, where
                       * the reference's nn.Embedding raises IndexError */
} scldm_dit_config;

/* Device pointers to the reference's parameters in their PyTorch layouts (Linear.weight is (out,in)).
 * state_dict keys: SURVEY.md section 8b.  Per-layer arrays are HOST arrays of n_layer device pointers. */
typedef struct {
  const float* pos_embed;                 /* pos_embed (1,S,D) */
  const float* t_w0; const float* t_b0;   /* t_embedder.mlp.0 (D,256),(D) */
  const float* t_w2; const float* t_b2;   /* t_embedder.mlp.2 (D,D),(D) */
  const float* in_w; const float* in_b;   /* input_proj (D,Din),(D) */
  const float* fin_w; const float* fin_b; /* final_layer.linear (Din,D),(Din) */
  const float* fin_ada_w; const float* fin_ada_b; /* final_layer.adaln_modulation.1 (2D,D),(2D) */
  const float* const* class_emb;          /* n_classes x class_embeddings.<name>.weight (vocab+1,D), sorted-name order */
  const float* const* attn_w; const float* const* attn_b;   /* blocks.i.attn.c_attn (3D,D),(3D) */
  const float* const* proj_w; const float* const* proj_b;   /* blocks.i.attn.c_proj (D,D),(D) */
  const float* const* w1; const float* const* w2;           /* blocks.i.mlp.w1/w2 (H,D) */
  const float* const* cproj;                                /* blocks.i.mlp.c_proj (D,H) */
  const float* const* ada_w; const float* const* ada_b;     /* blocks.i.adaln_modulation.1 (6D,D),(6D) */
} scldm_dit_weights;

const char* scldm_last_error(void);
int scldm_version(void);

/* A handle for a shape outside the fused family (n_embed 256, n_head 8, seq_len 16, n_embed_input <= 32) is valid for
 * scldm_dit_train_forward / _backward only; the fused entry points return SCLDM_ERR_SHAPE for it. */
int scldm_dit_create(const scldm_dit_config* cfg, scldm_dit** out);
void scldm_dit_destroy(scldm_dit* h);

/* Re-pack weights from the caller's parameter tensors into MFMA fragment streams (fp32, bf16 and split-bf16
 * copies) in one launch.  Sources stay owned by the caller and must stay allocated while the handle is used:
 * their addresses are remembered for scldm_dit_refresh_weights. */
int scldm_dit_load_weights(scldm_dit* h, const scldm_dit_weights* w, void* stream);

/* Cheap staleness guard for callers that cannot know whether the parameter tensors were modified in place
 * (e.g. ema_pytorch updates through `.data`, which does not bump torch's version counter; reference
 * src/scldm/models.py:446,690): fingerprints the tensors given to the last scldm_dit_load_weights ON DEVICE (every
 * element of every tensor: one read of the parameters, ~10 us for the base DiT) and re-packs, in the same stream and without a
 * host synchronisation, only if the fingerprint changed.  Three small launches when nothing changed. */
int scldm_dit_refresh_weights(scldm_dit* h, void* stream);

/* Range report of the fp16 weight stream packed by the last load / refresh (synchronises `stream`): values beyond +-65 504
 * (stored as inf: SCLDM_PREC_FP16 must not be used), non-zero values below the smallest fp16 normal 6.1e-5 (stored with fewer
 * than 10 mantissa bits) and the non-zero values packed. */
int scldm_dit_fp16_stats(scldm_dit* h, long long* overflow, long long* subnormal, long long* nonzero, void* stream);

/* AdamW step over every parameter tensor in ONE launch (csrc/optim.hip) - the optimizer of the reference's trainer is
 * torch.optim.AdamW (src/scldm/models.py configure_optimizers; experiments/configs/training/default.yaml); same arithmetic as torch's
 * fused AdamW.  `entries` is a HOST array (device pointers inside); `step` a device float holding the number of steps taken so far
 * (incremented here unless *found_inf != 0, in which case nothing is updated: torch.cuda.amp.GradScaler's protocol and the fp16
 * backward's overflow flag).  Python binding: scldm_amd.optim.AdamW. */
typedef struct {
  float* p;         /* parameter (updated in place) */
  const float* g;   /* gradient */
  float* m;         /* exp_avg */
  float* v;         /* exp_avg_sq */
  long long n;      /* elements */
} scldm_adamw_entry;
int scldm_adamw_step(const scldm_adamw_entry* entries, int count, float* step, const float* found_inf /* may be NULL */, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int maximize, void* stream);

/* The same step driven by a launch table in DEVICE memory (any number of tensors is one launch), with the per-step hyper-parameters
 * read from device memory and the exponential moving average of the reference's trainer in the same pass:
 *   - LatentDiffusion wraps the diffusion model in ema_pytorch.EMA (src/scldm/models.py:446-453; ema-pytorch==0.7.7, pyproject.toml:30)
 *     and calls ema_model.update() after every optimizer step (models.py:83-87); update() copies the online parameters while
 *     step <= update_after_step and does `ema.lerp_(online, 1 - decay)` every update_every steps after (ldm_base.yaml:51-55);
 *   - configure_optimizers attaches a per-step LambdaLR (models.py:603-605).
 * `hyper` (device float[4], may be NULL): [0] learning rate, [1] weight decay, [2] EMA action of THIS step (0 none, 1 copy, 2 lerp),
 * [3] lerp weight.  The caller refreshes it before the call (scldm_amd.optim.AdamW: one 16-byte asynchronous copy); a captured HIP
 * graph therefore follows both schedules.  *found_inf != 0 skips the AdamW update (and the step count) but not the EMA action, as
 * the reference's hook runs after every batch.  The EMA is torch.lerp's arithmetic bit for bit (Tensor.lerp_ / torch._foreach_lerp_).
 * scldm_adamw_table_build fills a HOST buffer of scldm_adamw_table_bytes(...) bytes from the entries (+ one EMA tensor per entry, or
 * NULL); the caller copies it to the device once and passes that copy as `table`. */
size_t scldm_adamw_table_bytes(const scldm_adamw_entry* entries, int count);
int scldm_adamw_table_build(const scldm_adamw_entry* entries, float* const* ema /* count pointers or NULL */, int count, void* table_host,
                            size_t bytes, int* n_blocks);
/* The table is [count tensor records (addresses, size) | workgroup map (sizes only)]: when only addresses changed (autograd handed out
 * another gradient buffer), scldm_adamw_table_update fills just the records (scldm_adamw_table_records_bytes(count) bytes) for the caller
 * to copy over the head of the device table. */
size_t scldm_adamw_table_records_bytes(int count);
int scldm_adamw_table_update(const scldm_adamw_entry* entries, float* const* ema, int count, void* records_host, size_t bytes);
typedef struct {
  const void* table;       /* device copy of the table */
  int count, n_blocks;
  float* step;             /* device: steps taken so far */
  const float* found_inf;  /* device, may be NULL */
  const float* hyper;      /* device float[4], may be NULL (then lr / weight_decay below, no EMA) */
  float lr, beta1, beta2, eps, weight_decay;
  int maximize;
  /* Gradient-norm clipping of the reference's trainer (experiments/configs/training/default.yaml:15-16: gradient_clip_val 10,
   * gradient_clip_algorithm norm -> Lightning calls torch.nn.utils.clip_grad_norm_(parameters, 10.0) between backward and
   * optimizer.step): max_grad_norm > 0 runs two more launches over the same table ahead of the update - the sum of squares of every
   * 4 096-element chunk, then one workgroup summing those partials in a fixed order - and the update reads
   * g * min(1, max_grad_norm / (norm + 1e-6)), torch's clip_coef_clamped.  clip_ws: device memory of
   * scldm_adamw_clip_workspace_bytes(n_blocks) bytes; afterwards clip_ws[0] = the step's total gradient norm (before clipping),
   * clip_ws[1] = the coefficient applied.
   * max_grad_norm <= 0 or clip_ws == NULL: no clipping (the struct of version 4 zero-extended). */
  float max_grad_norm;
  float* clip_ws;
} scldm_adamw_launch;
size_t scldm_adamw_clip_workspace_bytes(int n_blocks);
int scldm_adamw_table_step(const scldm_adamw_launch* launch, void* stream);

/* State arithmetic of the adaptive Dormand-Prince 5(4) solver - the reference's DEFAULT sampler (src/scldm/models.py:793 ->
 * transport/transport.py:324-331 -> integrators.py:100-112 -> torchdiffeq.odeint(method="dopri5")) - as four kernels instead of ~40
 * elementwise launches per step; the step-size control stays on the host (scldm_amd/transport: Sampler._sample_dopri5).  All tensors
 * fp32, contiguous, `n` elements on the device; `k` / `coef` are HOST arrays of n_k <= 7 device pointers / h-scaled tableau weights.
 *   scldm_rk_combine: out = y0 + sum_j coef_j k_j (left to right; y0 may be NULL)                      - stage points
 *   scldm_rk_error:   ws[1022] (double) = mean((sum_j coef_j k_j / (atol + rtol max(|y0|, |y1|)))^2)    - error ratio^2; ws = 8 KB of
 *                     device memory (block partials, summed in block order by a second one-workgroup launch)
 *   scldm_rk_dense:   coefficients c1..c4 of the quartic interpolant of an accepted step (c0 = y0)
 *   scldm_rk_poly:    out = c0 + s1 c1 + s2 c2 + s3 c3 + s4 c4                                           - a save point */
int scldm_rk_combine(float* out, const float* y0, const float* const* k, const float* coef, int n_k, long long n, void* stream);
int scldm_rk_error(const float* y0, const float* y1, const float* const* k, const float* coef, int n_k, long long n, float atol, float rtol,
                   void* ws, void* stream);
int scldm_rk_dense(const float* y0, const float* y1, const float* const* k, const float* mid_coef, int n_k, const float* fa, const float* fb,
                   float h, float two_h, long long n, float* c1, float* c2, float* c3, float* c4, void* stream);
int scldm_rk_poly(float* out, const float* c0, const float* c1, const float* c2, const float* c3, const float* c4, float s1, float s2, float s3,
                  float s4, long long n, void* stream);

/* Overflow guard of the fp16 (loss-scaled) training backward - the counterpart of torch.cuda.amp.GradScaler's found_inf for the
 * reference's trainer (experiments/scripts/train_ldm.py: the reference trains in TF32 and needs none; fp16 operands have TF32's
 * mantissa but 5 exponent bits).  The last launch of scldm_dit_train_backward counts the non-finite values among ALL gradients it
 * produced; when there are any, the NEXT backward lowers its loss scale by one more power of two (the headroom recovers one step
 * per 2 000 clean backwards), and - if the caller registered one with scldm_dit_train_set_found_inf - a device float is set to 1.0
 * (reset to 0.0 at the start of every fp16 backward) so that an optimizer can skip the step without a host read (torch's fused
 * Adam/AdamW take it as `optimizer.found_inf`).  scldm_dit_train_fp16_state synchronises `stream` and reports the loss scale S of
 * the last backward, its non-finite count, the headroom (<= 0, powers of two below the nominal scale) and the number of
 * backwards that overflowed so far. */
int scldm_dit_train_fp16_state(scldm_dit* h, float* scale, long long* nonfinite_last, int* headroom, long long* overflow_steps, void* stream);
int scldm_dit_train_set_found_inf(scldm_dit* h, float* found_inf /* device, caller-owned, may be NULL */);

/* Labels outside [0, vocab] (or == vocab without a null row) are clamped by the conditioning kernels and counted in a
 * device-side sticky counter instead of reading another class's table (the reference's nn.Embedding raises).  This call
 * synchronises `stream`, returns the count since the last call in *count and resets it. */
int scldm_dit_label_errors(scldm_dit* h, int* count, void* stream);

/* Floats per conditioning row: n_layer*6*D + 2*D (all adaLN vectors of all layers + final layer). */
int scldm_dit_mod_width(const scldm_dit* h);

/* Workspace bytes for calls that process up to n_fwd sample-forwards, n_rows conditioning rows and
 * an ODE state of n_state samples (0 when not sampling). */
size_t scldm_dit_workspace_bytes(const scldm_dit* h, int n_fwd, int n_rows, int n_state);

/* Conditioning rows: c = TimestepEmbedder(t) + sum_classes Embedding(label or null), then
 * SiLU(c) W_adaLN^T + b for every block and the final layer in one pass.
 * Replaces layers.py:351-364, nnets.py:283-288,380-456 and the nine adaln_modulation Linears
 * (layers.py:206,214-216,395,398).  t: device (n_rows) with t_stride 1, or one device float with
 * t_stride 0.  labels[c]: device (n_rows) int64, or NULL => class c uses its null token.
 * mod_out: device (n_rows, mod_width). */
int scldm_dit_cond_rows(scldm_dit* h, const float* t, int t_stride, const int64_t* const* labels, int n_rows,
                        float* mod_out, void* ws, void* stream);

/* DiT trunk on prepared conditioning: input_proj + pos_embed, n_layer fused blocks, final layer
 * (nnets.py:290-296).  Sample-forward s < n_direct reads latent x[s]; s >= n_direct re-reads the last
 * `rep` latents (x[n_direct - rep + (s - n_direct) % rep]) - the conditional CFG passes.
 * row_index[s] selects the conditioning row of `mod`.  out: device (n_fwd, S, Din). */
int scldm_dit_forward_rows(scldm_dit* h, const float* x, int n_direct, int rep, int n_fwd, const float* mod,
                           const int32_t* row_index, float* out, int precision, void* ws, void* stream);

/* DiT.forward in eval mode (nnets.py:273-297): x (n,S,Din), t (n), labels[c] (n) or NULL. */
int scldm_dit_forward(scldm_dit* h, const float* x, const float* t, const int64_t* const* labels, float* out, int n,
                      int precision, void* ws, void* stream);

/* DiT.forward_with_cfg (nnets.py:336-378) in one call.  x, out: (2B,S,Din).  t: (2B) with t_stride 1 or a
 * single device float with t_stride 0 (the ODE solver broadcasts one scalar, integrators.py:103-104).
 * The unconditional pass covers all 2B rows; conditional pass p re-runs the second half with the classes
 * in pass_mask[p] (bit c = class c keeps its labels, others null) and is blended with pass_scale[p]:
 *   joint:              n_pass = 1, mask = all classes, scale = mean(cfg_scale)        (:364-369)
 *   mutually_exclusive: one pass per cfg_scale entry, mask = that class, scale = its value (:372-376)
 * Conditional labels are given for n_urows UNIQUE label rows (ulabels[c]: device (n_urows) int64) and
 * cell_row (device (B) int32, or NULL when n_urows == B) maps each cell of the second half to its row.
 * With t_stride 1 the rows must be per-cell (n_urows == B, cell_row NULL).
 * t_stride 2: t is a dense (2B) vector whose uniformity is decided ON DEVICE (what torchdiffeq's `ones(B) * t` looks like
 * to a callee, integrators.py:103-104): a one-workgroup kernel compares the entries, the conditioning kernels of both plans
 * are enqueued and the ones of the plan that does not apply exit at once - no host synchronisation; the workspace must be
 * sized for the dense plan (n_rows = 2B + n_pass * B); de-duplicated label rows + cell_row are accepted. */
int scldm_dit_forward_cfg(scldm_dit* h, const float* x, const float* t, int t_stride, const int64_t* const* ulabels,
                          int n_urows, const int32_t* cell_row, int B, int n_pass, const uint32_t* pass_mask,
                          const float* pass_scale, float* out, int precision, void* ws, void* stream);

/* Fixed-grid ODE sampling of dz/dt = forward_with_cfg(z, t) from t=0 to 1 (transport.py:324-369,
 * integrators.py:100-112 with sampling_method euler|heun): t_i = i/n_steps, h = 1/n_steps;
 * euler: one evaluation per step; heun: two (k1 at t_i, k2 at t_i+h).  z (2B,S,Din) is updated in place
 * (first B rows unconditional, last B guided - models.py:801-812).  `n_steps` = reference num_steps - 1. */
int scldm_sample_ode(scldm_dit* h, float* z, const int64_t* const* ulabels, int n_urows, const int32_t* cell_row, int B,
                     int n_pass, const uint32_t* pass_mask, const float* pass_scale, int n_steps, int method,
                     int precision, void* ws, void* stream);

/* Options of a handle (read by the calls that follow).
 * SCLDM_OPT_CFG1_DIRECT (default 0): with ONE conditional pass of guidance scale exactly 1.0 and a scalar t (scldm_sample_ode,
 *   scldm_dit_forward_cfg with t_stride 0), DiT.forward_with_cfg's guided half u2 + 1.0 * (c2 - u2) (nnets.py:368,376) is c2 up to
 *   one fp32 rounding of that expression: the unconditional forward over the second half is not run (2B sample-forwards per
 *   evaluation instead of 3B) and the conditional output is returned as the guided rows.  Off: the reference's arithmetic, term by
 *   term.  The reference's own configs use guidance 1.0 for dentate_gyrus / parse1m (datamodule/default.yaml:46-47). */
#define SCLDM_OPT_CFG1_DIRECT 1
/* SCLDM_OPT_TAIL_SPLIT (default 0; bf16 / fp16 policies): the tiles of a trunk launch's partial last round that would run beside an
 * empty workgroup slot are launched as 32-token tiles (two per 64-token tile) next to the 64-token launch.  Results are bit-identical
 * either way (tested); measured 2-12 % SLOWER (a 32-token tile streams twice the weight bytes per token), so it stays an A/B knob. */
#define SCLDM_OPT_TAIL_SPLIT 2
int scldm_dit_set_option(scldm_dit* h, int option, int value);

/* DiT layers one fused-kernel launch runs (8 by default, SCLDM_LPL=1..8: the residual stays in registers between them; 0 for a
 * handle outside the fused shape family).  bench.py uses it to state the algorithmic FLOPs of a launch. */
int scldm_dit_layers_per_launch(const scldm_dit* h);

/* Measurement aid (bench.py): TFLOP/s a register-only v_mfma_f32_32x32x16_bf16 loop sustains on the current device for about
 * `iters` x 5 x 16 MFMAs per wave (fill 0: zero operands, 1: uniform [-1, 1), 2: N(0, 1); fill | 4: the same loop and values on
 * v_mfma_f32_32x32x16_f16 - the ceiling of the fp16 policy); same instruction stream for every fill -
 * the part clocks to its power budget, so the figure for realistic operands is the ceiling an MFMA-bound kernel can reach on this
 * device, below the nominal 2.5 PFLOP/s.  Synchronises the device. */
int scldm_mfma_sustained_tflops(int fill, int iters, double* tflops);

/* Timing hook for bench.py: when enabled, every fused-block launch is bracketed by HIP events on its
 * own stream; scldm_dit_block_timing drains them (synchronises) and returns launches and total ms. */
void scldm_dit_block_timing_enable(scldm_dit* h, int enable);
int scldm_dit_block_timing(scldm_dit* h, int* n_launches, double* total_ms);

/* ------------------------------------------------------------------------------------------------
 * Training path: DiT.forward with saved activations and its backward,
 * GEMMs on MFMA with fp32 (exact, parity) or bf16 operands (SCLDM_PREC_*; fp32 accumulation, fp32 activations,
 * LayerNorm / softmax / SiLU / residual in fp32 either way, as bf16-mixed autocast would keep them; the reference's
 * trainer default is `precision: 32`, experiments/configs/training/default.yaml:6).  The reference differentiates DiT.forward (nnets.py:273-297) with torch autograd
 * inside Transport.training_losses (transport/transport.py:110-150); these two calls are the pair a
 * torch.autograd.Function binds (scldm_amd/nnets.py).  Weights are read LIVE from the caller's parameter
 * tensors (no packing, nothing to refresh after an optimiser step).  Labels are the ones the model sees:
 * training-mode label dropout (nnets.py:300-334) is applied by the caller by substituting null tokens.
 * ------------------------------------------------------------------------------------------------ */
/* Writable mirror of scldm_dit_weights: every pointer receives d loss / d parameter in the parameter's own
 * PyTorch layout (overwritten, not accumulated).  pos_embed may be NULL (frozen in the reference, nnets.py:248). */
typedef struct {
  float* pos_embed;
  float* t_w0; float* t_b0;
  float* t_w2; float* t_b2;
  float* in_w; float* in_b;
  float* fin_w; float* fin_b;
  float* fin_ada_w; float* fin_ada_b;
  float* const* class_emb;
  float* const* attn_w; float* const* attn_b;
  float* const* proj_w; float* const* proj_b;
  float* const* w1; float* const* w2;
  float* const* cproj;
  float* const* ada_w; float* const* ada_b;
} scldm_dit_grads;

/* Bytes of the activation record written by train_forward and read by train_backward for n samples. */
size_t scldm_dit_train_saved_bytes(const scldm_dit* h, int n);
/* Scratch bytes for either call (gradient temporaries, split-K partials). */
size_t scldm_dit_train_workspace_bytes(const scldm_dit* h, int n);
/* The same for a known precision: the fused bf16 route of the base shape keeps 32 KB per cell per layer (layer inputs + the two
 * gated branch outputs) instead of the generic route's 295 KB, and needs none of its per-token gradient temporaries; buffers
 * sized by the two functions above (precision unknown: the larger, generic layout) are accepted by every route. */
size_t scldm_dit_train_saved_bytes_for(const scldm_dit* h, int n, int precision);
size_t scldm_dit_train_workspace_bytes_for(const scldm_dit* h, int n, int precision);

/* One-time set-up for training on the parameter tensors `w` at batch size n and `precision`: everything the two calls below
 * would otherwise create on first use - the bf16 weight mirror of the bf16-source route (DiT-L: 0.9 GB) with its cast-job
 * table, or the fused route's pack-job tables, side streams and events.  May allocate and synchronise `stream`; after it,
 * train_forward / train_backward on the same parameter pointers issue kernel launches and event records only.  Optional: a
 * forward on parameters it was not prepared for prepares itself (scldm_amd.nnets calls this whenever the parameters' storage
 * moved).  SCLDM_PREC_BF16X3 is accepted on the training entry points and served by the exact-fp32 GEMM route. */
int scldm_dit_train_prepare(scldm_dit* h, const scldm_dit_weights* w, int n, int precision, void* stream);

/* Gradient-ready events for the NEXT scldm_dit_train_backward on this handle - what a data-parallel caller needs to start
 * all-reducing a bucket of gradients while the rest of the backward still runs (the reference gets this from DDP's autograd
 * hooks, experiments/scripts/train_ldm.py:101).  events[i] is a hipEvent_t the call records on its stream once the gradients it
 * stands for are complete:
 *   SCLDM_GRAD_LAYER, l : attn / proj / w1 / w2 / cproj gradients (weights and biases) of every layer >= l
 *   SCLDM_GRAD_ADA,   l : adaLN weight / bias gradients of every layer <= l (l == n_layer: the final layer's adaLN as well)
 *   SCLDM_GRAD_END      : every gradient
 * The backward walks the layers from last to first and computes the adaLN gradients afterwards, first layer first.  Routes that
 * produce everything at once (the fused base-shape route) record all events at the end.  The list is consumed by that call. */
#define SCLDM_GRAD_LAYER 0
#define SCLDM_GRAD_ADA 1
#define SCLDM_GRAD_END 2
int scldm_dit_train_set_grad_events(scldm_dit* h, void* const* events, const int* kinds, const int* layers, int n);

/* out (n,S,Din) = DiT.forward(x (n,S,Din), t (n), labels) keeping every intermediate the backward needs in `saved`. */
int scldm_dit_train_forward(scldm_dit* h, const scldm_dit_weights* w, const float* x, const float* t,
                            const int64_t* const* labels, int n, float* out, int precision, void* saved, void* ws,
                            void* stream);

/* Given dout = d loss / d out (n,S,Din): all parameter gradients into `grads`, and d loss / d x into dx (n,S,Din)
 * unless dx is NULL.  x, labels and `saved` must be those of the matching train_forward call. */
int scldm_dit_train_backward(scldm_dit* h, const scldm_dit_weights* w, const scldm_dit_grads* grads, const float* x,
                             const int64_t* const* labels, const float* dout, int n, float* dx, int precision, void* saved,
                             void* ws, void* stream);

/* Flow-matching training step around the model call (Transport.training_losses, src/scldm/transport/transport.py:110-150 with
 * ICPlan.plan, path.py:148-151): the eager reference spends ~20 elementwise launches here per step.
 *   scldm_fm_mix:      xt = t*x1 + (1-t)*x0 (each product and the sum rounded separately, as eager torch does), ut = x1 - x0;
 *                      x1, x0, xt, ut are (n, e) fp32, t is (n).
 *   scldm_fm_loss:     loss[b] = mean_e (pred - ut)^2                              (mean_flat, utils.py:15-17)
 *   scldm_fm_loss_bwd: dpred[b][:] = gloss[b] * 2/e * (pred - ut)                  (its gradient w.r.t. pred) */
int scldm_fm_mix(const float* x1, const float* x0, const float* t, float* xt, float* ut, int n, int e, void* stream);

/* The same pieces with nothing left to the host (csrc/train_step.hip; round 6):
 *   scldm_fm_prepare   - Transport.sample + ICPlan.plan + the label dropout of DiT.forward in ONE kernel (transport.py:97-108,
 *                        path.py:148-151, nnets.py:395-402,440-452): t ~ U[0,1) and a drop decision per row, x0 ~ N(0, I), xt, ut,
 *                        and labels_out (n_classes, n) = what the model sees (null token where dropped; mutually_exclusive: ONE class
 *                        drawn per call among the non-NULL labels_in, every other class all-null - the reference draws it on the
 *                        host, nnets.py:395).  Draws come from Philox4x32-10 keyed by rng_state[0] (seed) at step rng_state[1]
 *                        (device memory, read only): reproducible, independent of the launch geometry.  x0 may be NULL.
 *   scldm_fm_loss_grad - loss_rows[b] = mean_e (pred - ut)^2, loss_mean = mean_b, dpred = d loss_mean / d pred, ONE kernel
 *                        (fixed-order sums); advances rng_state[1] when rng_state is given.  ticket: a zero-initialised device word.
 * Captured in a HIP graph both follow the device-side step counter: every replay draws a fresh batch. */
int scldm_fm_prepare(const float* x1, const int64_t* const* labels_in, const int* null_tokens, int n_classes, int strategy /* 0 mutually_exclusive, 1 joint */,
                     int drop, float p_drop, const unsigned long long* rng_state, int n, int e, float* t, float* x0, float* xt, float* ut,
                     int64_t* labels_out, void* stream);
int scldm_fm_loss_grad(const float* pred, const float* ut, int n, int e, float* loss_rows, float* loss_mean, float* dpred, unsigned int* ticket,
                       unsigned long long* rng_state, void* stream);

/* One optimisation step of LatentDiffusion.training_step on a batch of latents (src/scldm/models.py:628-663 + Lightning's backward
 * and optimizer.step(), + the EMA hook when `opt` carries one): scldm_fm_prepare -> scldm_dit_train_forward -> scldm_fm_loss_grad ->
 * scldm_dit_train_backward (every gradient into `grads`) -> scldm_adamw_table_step (skipped when opt is NULL: a data-parallel caller
 * reduces `grads` first).  Kernel launches and event records only; every per-step decision is read from device memory.  Buffers are
 * the caller's: */
typedef struct {
  int row_elems;            /* seq_len * n_embed_input */
  float* t;                 /* (n) */
  float* x0;                /* (n, row_elems) or NULL */
  float* xt; float* ut; float* pred; float* dpred;   /* (n, row_elems) */
  int64_t* labels;          /* (n_classes, n) */
  float* loss_rows;         /* (n) */
  float* loss_mean;         /* (1): the step's loss */
  unsigned int* ticket;     /* zero-initialised word */
  void* saved; void* ws;    /* scldm_dit_train_saved_bytes_for / _workspace_bytes_for */
} scldm_train_step_buffers;
int scldm_dit_train_step(scldm_dit* h, const scldm_dit_weights* w, const scldm_dit_grads* grads, const float* x1,
                         const int64_t* const* labels_in, const int* null_tokens, int n_classes, int strategy, float p_drop,
                         unsigned long long* rng_state, int n, int precision, const scldm_train_step_buffers* buffers,
                         const scldm_adamw_launch* opt /* may be NULL */, void* stream);
int scldm_fm_loss(const float* pred, const float* ut, float* loss, int n, int e, void* stream);
int scldm_fm_loss_bwd(const float* pred, const float* ut, const float* gloss, float* dpred, int n, int e, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TransformerVAE encode / decode (MCAB pooling / unpooling + negative-binomial head), fp32.
 * Shape family of the reference (experiments/configs/model/vae_base.yaml:8-19,64-73): n_embed 32, 16 inducing
 * points, trunk heads 8x4, cross heads 4x8, bias=False, shared gene embedding, shared theta, agg_func log1p.
 * ------------------------------------------------------------------------------------------------ */
typedef struct scldm_vae scldm_vae;

typedef struct {
  int n_genes;          /* vocabulary size; tables have n_genes+1 rows */
  int n_embed;          /* 32 */
  int n_inducing;       /* 16 */
  int n_embed_latent;   /* <= 32 (16) */
  int n_layer;          /* trunk Blocks per side */
  int n_head;           /* 8 */
  int n_head_cross;     /* 4 */
  int hidden_dim;       /* SwiGLU hidden: 88 */
  float layernorm_eps;
  int positional_encoding; /* Encoder.pos_embed present (nnets.py:103-106) */
  float nb_temperature; /* NegativeBinomialTransformerLayer.t (stochastic_layers.py:85) */
} scldm_vae_config;

/* One packed plain Block (state_dict prefix `encoder.encoder_layers.i.` / `decoder.decoder_layers.i.`). */
typedef struct {
  const float* ln1_w; const float* ln1_b; /* ln_1 (32) */
  const float* attn_w;                    /* attn.c_attn.weight (96,32) */
  const float* proj_w;                    /* attn.c_proj.weight (32,32) */
  const float* ln2_w; const float* ln2_b;
  const float* w1; const float* w2;       /* mlp.w1/w2.weight (H,32) */
  const float* cproj;                     /* mlp.c_proj.weight (32,H) */
} scldm_vae_block;

/* Cross-attention block parameters (`encoder.ca_layer.` / `decoder.decoder_cross_attention.`). */
typedef struct {
  const float* ln1_w; const float* ln1_b;   /* ln_1 (applied to x) */
  const float* ln1q_w; const float* ln1q_b; /* ln_1q (applied to the queries) */
  const float* attn_kv;                     /* attn.c_attn.weight (64,32): k | v */
  const float* attn_q;                      /* attn.c_attn_q.weight (32,32) */
  const float* attn_proj;                   /* attn.c_proj.weight (32,32) */
  const float* ln2_w; const float* ln2_b;
  const float* w1; const float* w2; const float* cproj; /* mlp */
} scldm_vae_cross;

typedef struct {
  const float* gene_embedding;     /* input_layer.gene_embedding.weight (n_genes+1, 32) */
  const float* inducing_points;    /* encoder.ca_layer.inducing_points (16,32) */
  const float* enc_pos_embed;      /* encoder.pos_embed (1,16,32) or NULL */
  const float* enc_latent_w;       /* encoder.encoder_latent_input.0.weight (n_lat,32) */
  const float* dec_latent_w;       /* decoder.decoder_latent_input.1.weight (32,n_lat) */
  const float* theta;              /* decoder_head.theta.weight (n_genes+1,1) */
  const float* head_w; const float* head_b; /* decoder_head.params (1,32),(1) */
  scldm_vae_cross enc_cross, dec_cross;
  const scldm_vae_block* enc_blocks; /* HOST array of n_layer entries */
  const scldm_vae_block* dec_blocks;
} scldm_vae_weights;

int scldm_vae_create(const scldm_vae_config* cfg, scldm_vae** out);
void scldm_vae_destroy(scldm_vae* h);
int scldm_vae_load_weights(scldm_vae* h, const scldm_vae_weights* w, void* stream);
/* Same parameter tensors as the last scldm_vae_load_weights (the pointers are remembered), possibly updated in place: a 64-bit
 * fingerprint of every source tensor is compared ON DEVICE with that of the packed copies and the re-pack (one launch over a job
 * table + the two derived tables) runs only if it moved - five small launches, no host synchronisation.  What the Python face calls
 * before every encode / decode whose parameter storages and version counters are unchanged (`.data` updates are invisible to both). */
int scldm_vae_refresh_weights(scldm_vae* h, void* stream);
size_t scldm_vae_workspace_bytes(const scldm_vae* h, int B, int G);

/* TransformerVAE.encode (vae.py:58-69): counts (B,S) fp32, genes (B,S) int64 -> z (B,16,n_lat). */
int scldm_vae_encode(scldm_vae* h, const float* counts, const int64_t* genes, int B, int S, float* z, int precision, void* ws,
                     void* stream);

/* TransformerVAE.decode (vae.py:71-87) up to the distribution parameters: z (B,16,n_lat), genes (B,G) int64,
 * library_size (B) -> mu (B,G) = softmax_G(logit / t) * library_size, theta (B,G) = exp(theta_emb[genes]). */
/* precision (encode, decode, decode_sample): SCLDM_PREC_FP32 = exact-fp32 MFMA chain (parity path, <= 1e-4);
 * SCLDM_PREC_FP16 = fp16 operands for the per-gene MCAB / SwiGLU contractions (10 mantissa bits = TF32, the arithmetic class the
 * reference runs CrossAttention / MLP in under set_float32_matmul_precision("high"): experiments/scripts/inference.py:26,
 * src/scldm/layers.py:248-264,305-330), fp32 accumulate / softmax / LayerNorm / NB head, weights saturated at +-65 504;
 * SCLDM_PREC_BF16 = the same with bf16 operands (8 bits: narrower than the reference).  Both 16-bit policies run about 3x the fp32
 * decode rate; the Linears of the 16-token cell trunks take the policy's operand type too (the reference runs them in TF32 as well;
 * their 16 x 16 attention stays fp32 on the VALU).  Other values: SCLDM_ERR_SHAPE. */
int scldm_vae_decode(scldm_vae* h, const float* z, const int64_t* genes, const float* library_size, int B, int G, float* mu,
                     float* theta, int precision, void* ws, void* stream);

/* TransformerVAE.decode followed by the negative-binomial draw of LatentDiffusion.sample (src/scldm/models.py:819,
 * `nb.sample()` on the distribution vae.py:87 builds): counts (B,G) fp32 ~ Poisson(Gamma(concentration = theta,
 * rate = theta / mu)), the Gamma-Poisson mixture scvi-tools' NegativeBinomial samples.  The draw happens inside the pass that
 * normalises the logits, so mu / theta are never written to HBM.  Philox4x32-10 keyed by `seed`, one counter per output element:
 * reproducible for a given (seed, B, G), independent of the launch geometry.  RNG-dependent: outside the bit-parity claim. */
int scldm_vae_decode_sample(scldm_vae* h, const float* z, const int64_t* genes, const float* library_size, int B, int G,
                            float* counts, unsigned long long seed, int precision, void* ws, void* stream);

/* Measurement hook for bench.py (the MCAB counterpart of scldm_dit_block_timing): when enabled, every launch of an MCAB kernel by
 * scldm_vae_encode / _decode / _decode_sample is bracketed by HIP events on the stream it is launched on (up to 64 launches per
 * kernel kind, later ones are not recorded); scldm_vae_kernel_timing drains one kind (synchronises on its events) and returns the
 * number of recorded launches and their summed duration. */
enum {
  SCLDM_VAE_K_ENC_POOL = 0,   /* enc_pool_kernel: MCAB pooling over the S input genes of a cell */
  SCLDM_VAE_K_ENC_CELL = 1,   /* enc_cell_kernel: c_proj + MLP + 16-token trunk + latent head */
  SCLDM_VAE_K_DEC_CELL = 2,   /* dec_cell_kernel: latent input + 16-token trunk + K | V of the unpooling */
  SCLDM_VAE_K_DEC_GENE = 3,   /* dec_gene_kernel: MCAB unpooling + SwiGLU + NB-head logit per decoded gene */
  SCLDM_VAE_K_DEC_FINAL = 4   /* dec_finalize(_sample)_kernel: softmax over genes x library size (+ the NB draw) */
};
#define SCLDM_VAE_KERNEL_KINDS 5
void scldm_vae_kernel_timing_enable(scldm_vae* h, int enable);
int scldm_vae_kernel_timing(scldm_vae* h, int kind, int* n_launches, double* total_ms);

/* The same draw from explicit parameter tensors: out[i] ~ NB(mu[i], theta[i]), i < n  (NegativeBinomial(mu, theta).sample()). */
int scldm_nb_sample(const float* mu, const float* theta, float* out, size_t n, unsigned long long seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TransformerVAE TRAINING step (BASELINE configs[0]; the reference trains through torch autograd over
 * TransformerVAE.forward, src/scldm/vae.py:29-56, inside VAE.training_step, src/scldm/models.py:249-290, with the loss
 * -log_nb_positive(counts, mu, theta), models.py:243, src/scldm/distributions.py:6-42).  fp32.
 *   forward : (mu, theta, z) = TransformerVAE.forward(counts, genes, library_size, counts_subset, genes_subset) - the inference
 *             kernels - and leaves in `saved` what the backward cannot recompute cheaply (the pooling's attention output and
 *             log-sum-exp) or should not wait for (the decoder's per-cell K | V); `ws` need not survive the call; uses the packed weights of the last scldm_vae_load_weights (call it after every optimiser step).
 *   backward: given d loss / d mu, d theta (B, G; either may be NULL) and optionally d loss / d z (B, 16, n_lat), writes d loss / d
 *             parameter for EVERY tensor named in `g` (same struct and layouts as scldm_vae_weights, pointers writable; overwritten,
 *             not accumulated; enc_pos_embed is ignored: frozen in the reference, nnets.py:103-106).  Parameters are read LIVE from
 *             `w`.  The two embedding tables' gradients (gene_embedding, theta) are scatter-added with float atomics (like torch's
 *             embedding backward); every other gradient is a deterministic two-stage sum.
 * mu / theta / z passed to the backward are the forward's outputs.  Buffers: scldm_vae_train_saved_bytes / _workspace_bytes. */
size_t scldm_vae_train_saved_bytes(const scldm_vae* h, int B);
size_t scldm_vae_train_workspace_bytes(const scldm_vae* h, int B, int S, int G);
int scldm_vae_train_forward(scldm_vae* h, const float* counts_subset, const int64_t* genes_subset, int B, int S, const int64_t* genes,
                            const float* library_size, int G, float* mu, float* theta, float* z, void* saved, void* ws, void* stream);
int scldm_vae_train_backward(scldm_vae* h, const scldm_vae_weights* w, const scldm_vae_weights* g, const float* counts_subset,
                             const int64_t* genes_subset, int B, int S, const int64_t* genes, const float* library_size, int G,
                             const float* mu, const float* theta, const float* z, const float* dmu, const float* dtheta, const float* dz,
                             void* saved, void* ws, void* stream);

/* log_nb_positive (src/scldm/distributions.py:6-42) elementwise over n values, and its gradient w.r.t. mu and theta given the
 * upstream gradient of the log-likelihood (either output may be NULL):
 *   res = theta (log(theta + eps) - L) + x (log(mu + eps) - L) + lgamma(x + theta) - lgamma(theta) - lgamma(x + 1),  L = log(theta + mu + eps) */
int scldm_nb_loglik(const float* x, const float* mu, const float* theta, float eps, float* out, size_t n, void* stream);
int scldm_nb_loglik_bwd(const float* x, const float* mu, const float* theta, const float* gout, float eps, float* dmu, float* dtheta,
                        size_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder input path: tokenize_cells(sample_genes="expressed"),
 * reference src/scldm/datamodule.py:660-731.  counts: device (N,G) fp32; gene_idx: device int64 token ids,
 * one shared row (gene_row_stride 0) or per cell (gene_row_stride G) - the reference tiles one row N times (:691).
 * Outputs: genes_subset (N,S) int64 and counts_subset (N,S) fp32 - expressed genes (counts > 0) in gene order,
 * padded with mask_idx / 0 (:709-716); num_expressed (N) int32; library_size (N) fp32 = row sums (:692).
 * Rows with num_expressed > S keep their first S expressed genes; the caller raises like the reference (:706-707).
 * ------------------------------------------------------------------------------------------------ */
int scldm_tokenize_expressed(const float* counts, const int64_t* gene_idx, long gene_row_stride, int N, int G, int S,
                             int64_t mask_idx, int64_t* genes_subset, float* counts_subset, int32_t* num_expressed,
                             float* library_size, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Output assembly: dense generated counts (N,G) fp32 -> CSR on device, replacing the
 * per-batch scipy.sparse.csr_matrix(counts.cpu().numpy()) of src/scldm/_utils.py:192-197 (after models.py:742).
 * Entries != 0 are kept in column order.  scldm_csr_count writes nnz per row; the caller forms
 * indptr (N+1) int64 = exclusive scan of it; scldm_csr_fill writes indices (nnz) int32 and data (nnz) fp32.
 * ------------------------------------------------------------------------------------------------ */
int scldm_csr_count(const float* dense, int N, int G, int32_t* row_nnz, void* stream);
int scldm_csr_fill(const float* dense, int N, int G, const int64_t* indptr, int32_t* indices, float* data, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Generation-evaluation MMD: kernel matrices of src/scldm/evaluations.py:10-69 and the
 * sums MMDLoss needs (:72-82), without the (Bx,By,D) broadcast tensors of the reference.
 * x (nx,D), y (ny,D) device fp32.  sum_out: device double = sum_ij k(x_i, y_j).  kmat: device (nx,ny) fp32 or NULL.
 * ------------------------------------------------------------------------------------------------ */
#define SCLDM_MMD_RBF 0         /* exp(-scale * |x - y|^2), evaluations.py:10-21 */
#define SCLDM_MMD_BRAYCURTIS 1  /* 1 - sum|x - y| / (sum|x + y| + 1e-8), :24-37 */
#define SCLDM_MMD_TANIMOTO 2    /* sum xy / (sum(x + y - xy) + 1e-8), :40-53 */
#define SCLDM_MMD_RUZICKA 3     /* sum min / (sum max + 1e-8), :56-69 */
#define SCLDM_MMD_SQDIST 4      /* |x - y|^2: the cost matrix of wasserstein(power=2), :103-105 */
#define SCLDM_MMD_DIST 5        /* |x - y|  (torch.cdist): wasserstein(power=1) */
size_t scldm_mmd_workspace_bytes(int nx, int ny);
int scldm_mmd_kernel_sum(const float* x, int nx, const float* y, int ny, int D, int kind, float scale, double* sum_out,
                         float* kmat, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Entropic optimal transport for the Wasserstein generation metrics: wasserstein(x0, x1, method="sinkhorn", reg, power) of
 * src/scldm/evaluations.py:85-108 (models.py:47-48 binds power 1 and 2) up to the final sqrt: uniform marginals,
 * M = cdist(x0, x1) ** power, Sinkhorn-Knopp scaling as third-party POT's ot.sinkhorn2 iterates it (error checked every
 * 10th iteration against stop_thr; POT default 1e-9; additionally ends when that error, already below 1e-6, has stopped
 * improving for three checks - the fp32 floor, which may sit above 1e-9), returns the HOST double *cost_out = <P, M>.  POT is not vendored:
 * parity unpinned, pinned to oracle/evaluations.py.  x0 (n,D), x1 (m,D) device fp32.  Synchronises the stream (the error
 * check is a host decision).  status: 0 converged, 1 iteration limit reached, 2 a scaling became zero / non-finite (as in
 * POT the previous scalings are kept and the loop ends - what reg = 0.05 on 17k-dimensional inputs usually does).
 * ------------------------------------------------------------------------------------------------ */
size_t scldm_sinkhorn_workspace_bytes(int n, int m);
int scldm_wasserstein_sinkhorn(const float* x0, int n, const float* x1, int m, int D, int power, float reg, long long num_iter_max,
                               float stop_thr, double* cost_out, long long* iters_out, int* status_out, void* ws, void* stream);

/* Debug hook (tools/phase_timing.py): device buffer receiving 16 x u64 s_memtime phase stamps per
 * (workgroup, wave) of each fused-block launch.  Only builds with -DSCLDM_PHASE_TIMING record; the
 * production library returns SCLDM_ERR_STATE. */
int scldm_dit_set_debug_buffer(scldm_dit* h, void* dev_buf);

#ifdef __cplusplus
}
#endif
#endif /* SCLDM_HIP_H */
