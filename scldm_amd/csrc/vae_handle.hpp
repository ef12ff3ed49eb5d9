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
};

// vae_api.hip internals used by the training entry points: TransformerVAE.encode that also leaves the pooling's attention output
// (B, 16, 32) and the log2-domain log-sum-exp of its scaled scores (B, 4, 16) in caller-provided buffers (either may be NULL)
int scldm_vae_encode_ex(scldm_vae* h, const float* counts, const int64_t* genes, int B, int S, float* z, int precision, float* pooled,
                        float* lse2, hipStream_t st);
