// The opaque handle behind `scldm_vae*` and the layout of its packed buffers (shared by vae_api.hip and vae_train_api.hip; the layout constants of the packed
// buffers live in vae_api.hip, the only translation unit that includes mcab.hpp).
#pragma once
#include "api_common.hpp"

struct scldm_vae {
  scldm_vae_config cfg;
  bool loaded;
  // owned device buffers
  float* enc_trunk;   // n_layer * kTrunkLayerFloats (LayerNorm vectors + MFMA fragments of every Linear)
  float* dec_trunk;
  float* frag_cell;   // fragments of the per-cell Linears around the trunks (layout: F_* below)
  float* frag_dec;    // c_proj 16 | w12 96 | wc 48 fragments (160*64 floats)
  float* frag_dec_halves;   // the 16 c_proj fragments with k in lane-half order (fp32 per-gene kernel)
  float* frag_enc_k;  // 16*64
  float* frag_enc_v;  // 16*64
  float* frag_enc_q;  // 16*64
  float* qtab;        // (n_genes+1, 32)
  float* small;       // copies of the small vectors / matrices (layout below)
  // borrowed (caller-owned, must stay alive): the big tables
  const float* emb;
  const float* theta;
  // re-pack job table + fingerprint state (scldm_vae_load_weights builds them, scldm_vae_refresh_weights runs them)
  void* d_jobs;                 // device VaePackJob[n_jobs]
  int n_jobs, jobs_cap;
  void* d_fp_src;               // device VaeFpSrc[n_fp]
  int n_fp, fp_cap;
  unsigned long long* d_fp_state;   // [0] accumulator, [1] fingerprint of the packed copies
  int* d_dirty;                 // [0] re-pack?, [1] force
  // sources of the two derived tables (enc_qfrag_kernel, dec_qtab_kernel)
  const float *q_ind, *q_eln_w, *q_eln_b, *q_ewq, *q_dln_w, *q_dln_b, *q_dwq;
  // training backward: the recompute forwards of the 16-token sides and the per-gene partial reduction run on a second stream
  // beside the per-gene backward (created on first use by scldm_vae_train_backward)
  hipStream_t side;
  hipEvent_t ev_fork, ev_gene, ev_join;
  // measurement hook (scldm_vae_kernel_timing_enable): HIP event pairs around each MCAB kernel launch, per kernel kind
  bool timing;
  hipEvent_t tev[SCLDM_VAE_KERNEL_KINDS][2 * 64];
  int tev_made[SCLDM_VAE_KERNEL_KINDS], tev_used[SCLDM_VAE_KERNEL_KINDS];
};

// vae_api.hip internals used by the training entry points: TransformerVAE.encode that also leaves the pooling's attention output
// (B, 16, 32) and the log2-domain log-sum-exp of its scaled scores (B, 4, 16) in caller-provided buffers (either may be NULL)
int scldm_vae_refresh(scldm_vae* h, bool force, hipStream_t st);
int scldm_vae_encode_ex(scldm_vae* h, const float* counts, const int64_t* genes, int B, int S, float* z, int precision, float* pooled,
                        float* lse2, hipStream_t st);
