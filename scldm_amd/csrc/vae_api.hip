// C ABI of the TransformerVAE encode / decode path (see include/scldm_hip.h).
#include <algorithm>
#include <cstring>
#include <vector>

#include "vae_handle.hpp"
#include "mcab.hpp"

using namespace scldm;

// offsets (floats) into `frag_cell`
enum : int {
  F_DEC_LAT = 0,                        // decoder_latent_input.1.weight (32, n_lat), K padded to 32
  F_DEC_KV = F_DEC_LAT + 1024,          // decoder_cross_attention.attn.c_attn: K tile | V tile
  F_ENC_PROJ = F_DEC_KV + 2048,         // encoder.ca_layer.attn.c_proj
  F_ENC_W12 = F_ENC_PROJ + 1024,        // encoder.ca_layer.mlp w1 | w2, six hidden tiles
  F_ENC_WC = F_ENC_W12 + kHTiles * 1024,
  F_ENC_LAT = F_ENC_WC + kHTiles * 512, // encoder_latent_input.0.weight (n_lat, 32), rows padded to 32
  F_TOTAL = F_ENC_LAT + 1024
};

// offsets (floats) into `small`
enum : int {
  S_ENC_LN1W = 0, S_ENC_LN1B = 32, S_ENC_LN2W = 64, S_ENC_LN2B = 96, S_ENC_PROJ = 128,          // (32,32)
  S_ENC_W1 = S_ENC_PROJ + 1024, S_ENC_W2 = S_ENC_W1 + 128 * 32, S_ENC_CP = S_ENC_W2 + 128 * 32,  // up to H=128
  S_ENC_IND = S_ENC_CP + 32 * 128, S_ENC_POS = S_ENC_IND + 512, S_ENC_LAT = S_ENC_POS + 512,    // (n_lat<=32, 32)
  S_DEC_LAT = S_ENC_LAT + 1024, S_DEC_LN1W = S_DEC_LAT + 1024, S_DEC_LN1B = S_DEC_LN1W + 32,
  S_DEC_KV = S_DEC_LN1B + 32, S_DEC_LN2W = S_DEC_KV + 64 * 32, S_DEC_LN2B = S_DEC_LN2W + 32,
  S_HEAD_W = S_DEC_LN2B + 32, S_HEAD_B = S_HEAD_W + 32, S_HEAD_W2 = S_HEAD_B + 32, S_TOTAL = S_HEAD_W2 + 32
};



extern "C" int scldm_vae_create(const scldm_vae_config* cfg, scldm_vae** out) {
  if (!cfg || !out) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (cfg->n_embed != 32 || cfg->n_inducing != 16 || cfg->n_head != 8 || cfg->n_head_cross != 4)
    return fail(SCLDM_ERR_SHAPE, "MCAB kernels support n_embed=32, n_inducing=16, n_head=8, n_head_cross=4 (got %d,%d,%d,%d)",
                cfg->n_embed, cfg->n_inducing, cfg->n_head, cfg->n_head_cross);
  if (cfg->n_embed_latent < 1 || cfg->n_embed_latent > 32) return fail(SCLDM_ERR_SHAPE, "n_embed_latent must be in [1,32]");
  if (cfg->hidden_dim < 1 || cfg->hidden_dim > kHPad) return fail(SCLDM_ERR_SHAPE, "hidden_dim must be in [1,%d]", kHPad);
  if (cfg->n_layer < 0 || cfg->n_genes < 1) return fail(SCLDM_ERR_SHAPE, "bad n_layer / n_genes");
  scldm_vae* h = new scldm_vae();
  memset(h, 0, sizeof(*h));
  h->cfg = *cfg;
  const size_t tl = (size_t)(cfg->n_layer > 0 ? cfg->n_layer : 1) * kTrunkLayerFloats * 4;
  hipError_t e = hipMalloc((void**)&h->enc_trunk, tl);
  if (e == hipSuccess) e = hipMalloc((void**)&h->dec_trunk, tl);
  if (e == hipSuccess) e = hipMalloc((void**)&h->frag_cell, (size_t)F_TOTAL * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&h->frag_dec, 160 * 64 * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&h->frag_dec_halves, 16 * 64 * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&h->frag_enc_k, 16 * 64 * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&h->frag_enc_v, 16 * 64 * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&h->frag_enc_q, 16 * 64 * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&h->qtab, (size_t)(cfg->n_genes + 1) * 32 * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&h->small, (size_t)S_TOTAL * 4);
  if (e == hipSuccess) e = hipMemset(h->small, 0, (size_t)S_TOTAL * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_fp_state, 16);
  if (e == hipSuccess) e = hipMemset(h->d_fp_state, 0, 16);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_dirty, 8);
  if (e == hipSuccess) e = hipMemset(h->d_dirty, 0, 8);
  if (e != hipSuccess) {
    int rc = fail(SCLDM_ERR_HIP, "hipMalloc failed in scldm_vae_create: %s", hipGetErrorString(e));
    scldm_vae_destroy(h);
    return rc;
  }
  *out = h;
  return SCLDM_OK;
}

extern "C" void scldm_vae_destroy(scldm_vae* h) {
  if (!h) return;
  float* ptrs[] = {h->enc_trunk, h->dec_trunk, h->frag_cell, h->frag_dec, h->frag_dec_halves, h->frag_enc_k, h->frag_enc_v, h->frag_enc_q, h->qtab, h->small};
  for (float* p : ptrs)
    if (p) (void)hipFree(p);
  void* more[] = {h->d_jobs, h->d_fp_src, h->d_fp_state, h->d_dirty};
  for (void* p : more)
    if (p) (void)hipFree(p);
  if (h->side) (void)hipStreamDestroy(h->side);
  for (hipEvent_t e : {h->ev_fork, h->ev_gene, h->ev_join})
    if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < SCLDM_VAE_KERNEL_KINDS; ++k)
    for (int i = 0; i < h->tev_made[k]; ++i) (void)hipEventDestroy(h->tev[k][i]);
  delete h;
}

// ---- measurement hook: event pairs around the MCAB kernel launches (include/scldm_hip.h) -------------------------------------------
namespace {
struct KernelTimer {   // records on construction and destruction when the handle's hook is on and there is room
  scldm_vae* h; int kind; hipStream_t st; bool on;
  KernelTimer(scldm_vae* h_, int kind_, hipStream_t st_) : h(h_), kind(kind_), st(st_), on(false) {
    if (!h->timing || h->tev_used[kind] + 2 > 2 * 64) return;
    while (h->tev_made[kind] < h->tev_used[kind] + 2) {
      if (hipEventCreate(&h->tev[kind][h->tev_made[kind]]) != hipSuccess) return;
      ++h->tev_made[kind];
    }
    on = hipEventRecord(h->tev[kind][h->tev_used[kind]], st) == hipSuccess;
  }
  ~KernelTimer() {
    if (!on) return;
    (void)hipEventRecord(h->tev[kind][h->tev_used[kind] + 1], st);
    h->tev_used[kind] += 2;
  }
};
}  // namespace

extern "C" void scldm_vae_kernel_timing_enable(scldm_vae* h, int enable) {
  if (!h) return;
  h->timing = enable != 0;
  for (int k = 0; k < SCLDM_VAE_KERNEL_KINDS; ++k) h->tev_used[k] = 0;
}
extern "C" int scldm_vae_kernel_timing(scldm_vae* h, int kind, int* n_launches, double* total_ms) {
  if (!h || kind < 0 || kind >= SCLDM_VAE_KERNEL_KINDS) return fail(SCLDM_ERR_SHAPE, "scldm_vae_kernel_timing: bad argument");
  double tot = 0;
  int n = 0;
  for (int i = 0; i + 1 < h->tev_used[kind]; i += 2) {
    HIP_TRY(hipEventSynchronize(h->tev[kind][i + 1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->tev[kind][i], h->tev[kind][i + 1]));
    tot += ms;
    ++n;
  }
  if (n_launches) *n_launches = n;
  if (total_ms) *total_ms = tot;
  h->tev_used[kind] = 0;
  return SCLDM_OK;
}

// ---- weight (re-)packing: one job table, one launch (mcab.hpp: vae_pack_jobs_kernel) -------------------------------------------
namespace {
struct JobList {
  std::vector<VaePackJob> jobs;
  std::vector<VaeFpSrc> fps;
  void src(const float* p, long long n) {
    if (p && n > 0) fps.push_back(VaeFpSrc{reinterpret_cast<const uint32_t*>(p), n});
  }
  void copy(float* dst, const float* s, int n) { jobs.push_back(VaePackJob{VJ_COPY, n, 0, 0, s, nullptr, dst}); }
  void frag32(const float* W, int ld, float* dst) { jobs.push_back(VaePackJob{VJ_FRAG32, ld, 0, 0, W, nullptr, dst}); }
  void halves(const float* W, int ld, float* dst) { jobs.push_back(VaePackJob{VJ_FRAG32_HALVES, ld, 0, 0, W, nullptr, dst}); }
  void w12(const float* w1, const float* w2, int H, float* dst) { jobs.push_back(VaePackJob{VJ_W12, H, 0, 0, w1, w2, dst}); }
  void wc(const float* Wc, int H, float* dst) { jobs.push_back(VaePackJob{VJ_WC, H, 0, 0, Wc, nullptr, dst}); }
  void pad(const float* W, int ld, int rows, int K, float* dst) { jobs.push_back(VaePackJob{VJ_PAD, ld, rows, K, W, nullptr, dst}); }
};
void trunk_jobs(JobList& jl, float* dst, const scldm_vae_block* blocks, int n_layer, int H) {
  for (int i = 0; i < n_layer; ++i) {
    float* w = dst + (size_t)i * kTrunkLayerFloats;
    const scldm_vae_block& b = blocks[i];
    jl.copy(w + T_LN1W, b.ln1_w, 32); jl.copy(w + T_LN1B, b.ln1_b, 32); jl.copy(w + T_LN2W, b.ln2_w, 32); jl.copy(w + T_LN2B, b.ln2_b, 32);
    for (int t = 0; t < 3; ++t) jl.frag32(b.attn_w + t * 32 * 32, 32, w + T_QKV + t * 1024);   // q | k | v rows
    jl.frag32(b.proj_w, 32, w + T_PROJ);
    jl.w12(b.w1, b.w2, H, w + T_W12);
    jl.wc(b.cproj, H, w + T_WC);
    jl.src(b.ln1_w, 32); jl.src(b.ln1_b, 32); jl.src(b.ln2_w, 32); jl.src(b.ln2_b, 32); jl.src(b.attn_w, 96 * 32); jl.src(b.proj_w, 1024);
    jl.src(b.w1, (long long)H * 32); jl.src(b.w2, (long long)H * 32); jl.src(b.cproj, (long long)32 * H);
  }
}
template <typename T>
int upload(void** dev, int* cap, const std::vector<T>& v, hipStream_t st) {
  // a pack / fingerprint launch of an earlier refresh may still be reading the table on the caller's (non-blocking) stream: the
  // null-stream copy below would not wait for it.  Loading weights is a rare path: drain the stream first.
  HIP_TRY(hipStreamSynchronize(st));
  if ((int)v.size() > *cap) {
    if (*dev) (void)hipFree(*dev);
    *dev = nullptr;
    HIP_TRY(hipMalloc(dev, v.size() * sizeof(T)));
    *cap = (int)v.size();
  }
  HIP_TRY(hipMemcpy(*dev, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return SCLDM_OK;
}
}  // namespace

// fingerprint of every source tensor -> device-side compare -> the gated re-pack: five small launches, no host round trip
int scldm_vae_refresh(scldm_vae* h, bool force, hipStream_t st) {
  const scldm_vae_config& c = h->cfg;
  if (force) vae_set_word_kernel<<<1, 1, 0, st>>>(h->d_dirty + 1, 1);
  vae_fingerprint_kernel<<<dim3(h->n_fp, 64), 256, 0, st>>>((const VaeFpSrc*)h->d_fp_src, h->d_fp_state);
  vae_fp_compare_kernel<<<1, 1, 0, st>>>(h->d_fp_state, h->d_dirty);
  vae_pack_jobs_kernel<<<h->n_jobs * kVaeJobBlocks, 256, 0, st>>>((const VaePackJob*)h->d_jobs, h->n_jobs, h->d_dirty);
  enc_qfrag_kernel<<<1, 64, 0, st>>>(h->q_ind, h->q_eln_w, h->q_eln_b, h->q_ewq, h->frag_enc_q, c.layernorm_eps, h->d_dirty);
  dec_qtab_kernel<<<cdiv(c.n_genes + 1, 128), 128, 0, st>>>(h->emb, h->q_dln_w, h->q_dln_b, h->q_dwq, h->qtab, c.n_genes + 1, c.layernorm_eps,
                                                            h->d_dirty);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

extern "C" int scldm_vae_load_weights(scldm_vae* h, const scldm_vae_weights* w, void* stream_) {
  if (!h || !w) return fail(SCLDM_ERR_SHAPE, "null argument");
  hipStream_t st = (hipStream_t)stream_;
  const scldm_vae_config& c = h->cfg;
  const int H = c.hidden_dim, nl = c.n_embed_latent;
  if (c.positional_encoding && !w->enc_pos_embed) return fail(SCLDM_ERR_SHAPE, "positional_encoding set but enc_pos_embed is NULL");
  JobList jl;
  trunk_jobs(jl, h->enc_trunk, w->enc_blocks, c.n_layer, H);
  trunk_jobs(jl, h->dec_trunk, w->dec_blocks, c.n_layer, H);
  float* s = h->small;
  const scldm_vae_cross &ec = w->enc_cross, &dc = w->dec_cross;
  jl.copy(s + S_ENC_LN1W, ec.ln1_w, 32); jl.copy(s + S_ENC_LN1B, ec.ln1_b, 32); jl.copy(s + S_ENC_LN2W, ec.ln2_w, 32); jl.copy(s + S_ENC_LN2B, ec.ln2_b, 32);
  jl.copy(s + S_ENC_PROJ, ec.attn_proj, 1024); jl.copy(s + S_ENC_W1, ec.w1, H * 32); jl.copy(s + S_ENC_W2, ec.w2, H * 32); jl.copy(s + S_ENC_CP, ec.cproj, 32 * H);
  jl.copy(s + S_ENC_IND, w->inducing_points, 512); jl.copy(s + S_ENC_LAT, w->enc_latent_w, nl * 32); jl.copy(s + S_DEC_LAT, w->dec_latent_w, 32 * nl);
  jl.copy(s + S_DEC_LN1W, dc.ln1_w, 32); jl.copy(s + S_DEC_LN1B, dc.ln1_b, 32); jl.copy(s + S_DEC_KV, dc.attn_kv, 64 * 32);
  jl.copy(s + S_DEC_LN2W, dc.ln2_w, 32); jl.copy(s + S_DEC_LN2B, dc.ln2_b, 32); jl.copy(s + S_HEAD_W, w->head_w, 32); jl.copy(s + S_HEAD_B, w->head_b, w->theta ? 1 : 2);
  if (!w->theta) jl.copy(s + S_HEAD_W2, w->head_w + 32, 32);   // unshared theta: decoder_head.params is Linear(32, 2) (stochastic_layers.py:94-96)
  if (c.positional_encoding) jl.copy(s + S_ENC_POS, w->enc_pos_embed, 512);
  // MFMA fragments
  float* fc = h->frag_cell;
  jl.pad(w->dec_latent_w, nl, 32, nl, fc + F_DEC_LAT);
  jl.frag32(dc.attn_kv, 32, fc + F_DEC_KV);
  jl.frag32(dc.attn_kv + 32 * 32, 32, fc + F_DEC_KV + 1024);
  jl.frag32(ec.attn_proj, 32, fc + F_ENC_PROJ);
  jl.w12(ec.w1, ec.w2, H, fc + F_ENC_W12);
  jl.wc(ec.cproj, H, fc + F_ENC_WC);
  jl.pad(w->enc_latent_w, 32, nl, 32, fc + F_ENC_LAT);
  jl.frag32(dc.attn_proj, 32, h->frag_dec);                                  // 16 fragments
  jl.halves(dc.attn_proj, 32, h->frag_dec_halves);                           // the same, k in lane-half order (fp32 path)
  jl.w12(dc.w1, dc.w2, H, h->frag_dec + 16 * 64);                            // 96
  jl.wc(dc.cproj, H, h->frag_dec + (16 + 96) * 64);                          // 48
  jl.frag32(ec.attn_kv, 32, h->frag_enc_k);                                  // K rows 0-31
  jl.frag32(ec.attn_kv + 32 * 32, 32, h->frag_enc_v);                        // V rows 32-63
  // every tensor a packed copy or a derived table depends on (theta and the gene embedding are read live by the kernels; the
  // embedding also feeds the decoder's query table)
  for (const scldm_vae_cross* cr : {&ec, &dc}) {
    jl.src(cr->ln1_w, 32); jl.src(cr->ln1_b, 32); jl.src(cr->ln1q_w, 32); jl.src(cr->ln1q_b, 32); jl.src(cr->attn_kv, 64 * 32); jl.src(cr->attn_q, 1024);
    jl.src(cr->attn_proj, 1024); jl.src(cr->ln2_w, 32); jl.src(cr->ln2_b, 32); jl.src(cr->w1, (long long)H * 32); jl.src(cr->w2, (long long)H * 32);
    jl.src(cr->cproj, (long long)32 * H);
  }
  jl.src(w->inducing_points, 512); jl.src(w->enc_latent_w, (long long)nl * 32); jl.src(w->dec_latent_w, (long long)32 * nl);
  jl.src(w->head_w, w->theta ? 32 : 64); jl.src(w->head_b, w->theta ? 1 : 2); jl.src(w->gene_embedding, (long long)(c.n_genes + 1) * 32);
  if (c.positional_encoding) jl.src(w->enc_pos_embed, 512);
  for (const VaePackJob& j : jl.jobs)
    if (!j.src0 || !j.dst || (j.kind == VJ_W12 && !j.src1)) return fail(SCLDM_ERR_SHAPE, "scldm_vae_load_weights: a weight pointer is NULL");
  int rc;
  if ((rc = upload(&h->d_jobs, &h->jobs_cap, jl.jobs, st)) || (rc = upload(&h->d_fp_src, &h->fp_cap, jl.fps, st))) return rc;
  h->n_jobs = (int)jl.jobs.size();
  h->n_fp = (int)jl.fps.size();
  h->q_ind = w->inducing_points; h->q_eln_w = ec.ln1q_w; h->q_eln_b = ec.ln1q_b; h->q_ewq = ec.attn_q;
  h->q_dln_w = dc.ln1q_w; h->q_dln_b = dc.ln1q_b; h->q_dwq = dc.attn_q;
  h->emb = w->gene_embedding;
  h->theta = w->theta;
  if ((rc = scldm_vae_refresh(h, true, st))) return rc;
  h->loaded = true;
  return SCLDM_OK;
}

extern "C" int scldm_vae_refresh_weights(scldm_vae* h, void* stream_) {
  if (!h) return fail(SCLDM_ERR_SHAPE, "null handle");
  if (!h->loaded) return fail(SCLDM_ERR_STATE, "scldm_vae_load_weights has not been called");
  return scldm_vae_refresh(h, false, (hipStream_t)stream_);
}

static int vae_ready(const scldm_vae* h) {
  if (!h) return fail(SCLDM_ERR_SHAPE, "null handle");
  if (!h->loaded) return fail(SCLDM_ERR_STATE, "scldm_vae_load_weights has not been called");
  return SCLDM_OK;
}

static const int kDecTilesPerWave = getenv("SCLDM_DEC_TPW") ? atoi(getenv("SCLDM_DEC_TPW")) : (32 + kDecWaves - 1) / kDecWaves;  // kDecWaves waves x tiles x 32 genes ~ 1024 genes per workgroup
static inline int dec_chunks(int G) { return cdiv(G, kDecWaves * kDecTilesPerWave * 32); }

extern "C" size_t scldm_vae_workspace_bytes(const scldm_vae* h, int B, int G) {
  if (!h) return 0;
  size_t kv = align256((size_t)B * 48 * 64 * 4);
  size_t part = align256((size_t)B * dec_chunks(G > 0 ? G : 1) * 2 * 4);
  size_t pooled = align256((size_t)B * 16 * 32 * 4);
  // unshared theta: scldm_vae_decode_sample keeps the (B, G) dispersions of dec_gene_kernel here for the fused draw
  size_t theta_rows = (h->loaded && !h->theta) ? align256((size_t)B * (G > 0 ? G : 1) * 4) : 0;
  return kv + part + pooled + theta_rows;
}

int scldm_vae_encode_ex(scldm_vae* h, const float* counts, const int64_t* genes, int B, int S, float* z, int precision, float* pooled,
                        float* lse2, hipStream_t st) {
  int rc = vae_ready(h);
  if (rc) return rc;
  if (precision != SCLDM_PREC_FP32 && precision != SCLDM_PREC_BF16 && precision != SCLDM_PREC_FP16)
    return fail(SCLDM_ERR_SHAPE, "unsupported MCAB precision %d (fp32, bf16, fp16)", precision);
  if (B <= 0 || S <= 0 || !counts || !genes || !z || !pooled) return fail(SCLDM_ERR_SHAPE, "bad argument");
  const scldm_vae_config& c = h->cfg;
  EncPoolArgs p;
  p.counts = counts; p.genes = genes; p.emb = h->emb;
  p.ln1_w = h->small + S_ENC_LN1W; p.ln1_b = h->small + S_ENC_LN1B;
  p.kfrag = h->frag_enc_k; p.vfrag = h->frag_enc_v; p.qfrag = h->frag_enc_q;
  p.pooled = pooled; p.lse2 = lse2; p.S = S; p.eps = c.layernorm_eps;
  static const int enc_nw = getenv("SCLDM_ENC_WAVES") ? atoi(getenv("SCLDM_ENC_WAVES")) : 4;   // waves per cell of the bf16-operand pooling kernel (A/B: 4 | 6 | 8)
  {
  KernelTimer kt(h, SCLDM_VAE_K_ENC_POOL, st);
  if (precision == SCLDM_PREC_BF16) {
    if (enc_nw == 6) enc_pool_kernel<OpBF16, 6><<<B, 384, 0, st>>>(p);
    else if (enc_nw == 8) enc_pool_kernel<OpBF16, 8><<<B, 512, 0, st>>>(p);
    else enc_pool_kernel<OpBF16><<<B, 256, 0, st>>>(p);
  } else if (precision == SCLDM_PREC_FP16) {   // TF32's mantissa: the reference's own arithmetic class for MCAB
    if (enc_nw == 6) enc_pool_kernel<OpFP16, 6><<<B, 384, 0, st>>>(p);
    else if (enc_nw == 8) enc_pool_kernel<OpFP16, 8><<<B, 512, 0, st>>>(p);
    else enc_pool_kernel<OpFP16><<<B, 256, 0, st>>>(p);
  } else enc_pool_kernel<OpF32><<<B, 256, 0, st>>>(p);
  }
  LAUNCH_CHECK();
  EncCellArgs e;
  e.pooled = pooled; e.ind = h->small + S_ENC_IND; e.proj_frag = h->frag_cell + F_ENC_PROJ;
  e.ca_ln2_w = h->small + S_ENC_LN2W; e.ca_ln2_b = h->small + S_ENC_LN2B;
  e.w12_frag = h->frag_cell + F_ENC_W12; e.wc_frag = h->frag_cell + F_ENC_WC;
  e.pos = c.positional_encoding ? h->small + S_ENC_POS : nullptr;
  e.trunk = h->enc_trunk; e.lat_frag = h->frag_cell + F_ENC_LAT; e.z = z;
  e.B = B; e.n_lat = c.n_embed_latent; e.n_layer = c.n_layer; e.eps = c.layernorm_eps;
  {
    KernelTimer kt(h, SCLDM_VAE_K_ENC_CELL, st);
    if (precision == SCLDM_PREC_FP16) enc_cell_kernel<OpFP16><<<cdiv(B, 2), 64 * kTrunkWaves, 0, st>>>(e);
    else if (precision == SCLDM_PREC_BF16) enc_cell_kernel<OpBF16><<<cdiv(B, 2), 64 * kTrunkWaves, 0, st>>>(e);
    else enc_cell_kernel<OpF32><<<cdiv(B, 2), 64 * kTrunkWaves, 0, st>>>(e);
  }
  LAUNCH_CHECK();
  return SCLDM_OK;
}

extern "C" int scldm_vae_encode(scldm_vae* h, const float* counts, const int64_t* genes, int B, int S, float* z, int precision,
                                void* ws_, void* stream_) {
  if (!ws_) return fail(SCLDM_ERR_SHAPE, "bad argument");
  // (B,16,32) pooled attention output; encode and decode calls are stream-ordered, so they may share the workspace
  return scldm_vae_encode_ex(h, counts, genes, B, S, z, precision, (float*)ws_, nullptr, (hipStream_t)stream_);
}

static int vae_decode_impl(scldm_vae* h, const float* z, const int64_t* genes, const float* library_size, int B, int G, float* mu,
                           float* theta, bool draw, unsigned long long seed, int precision, void* ws_, void* stream_) {
  int rc = vae_ready(h);
  if (rc) return rc;
  if (precision != SCLDM_PREC_FP32 && precision != SCLDM_PREC_BF16 && precision != SCLDM_PREC_FP16)
    return fail(SCLDM_ERR_SHAPE, "unsupported MCAB precision %d (fp32, bf16, fp16)", precision);
  if (B <= 0 || G <= 0 || !z || !genes || !library_size || !mu || (!theta && !draw) || !ws_) return fail(SCLDM_ERR_SHAPE, "bad argument");
  hipStream_t st = (hipStream_t)stream_;
  const scldm_vae_config& c = h->cfg;
  float* kv = (float*)ws_;
  float* part = (float*)((char*)ws_ + align256((size_t)B * 48 * 64 * 4));
  DecCellArgs d;
  d.z = z; d.lat_frag = h->frag_cell + F_DEC_LAT; d.trunk = h->dec_trunk;
  d.ca_ln1_w = h->small + S_DEC_LN1W; d.ca_ln1_b = h->small + S_DEC_LN1B; d.kv_frag = h->frag_cell + F_DEC_KV;
  d.kvfrag = kv; d.B = B; d.n_lat = c.n_embed_latent; d.n_layer = c.n_layer; d.eps = c.layernorm_eps;
  {
    KernelTimer kt(h, SCLDM_VAE_K_DEC_CELL, st);
    if (precision == SCLDM_PREC_FP16) dec_cell_kernel<OpFP16><<<cdiv(B, 2), 64 * kTrunkWaves, 0, st>>>(d);
    else if (precision == SCLDM_PREC_BF16) dec_cell_kernel<OpBF16><<<cdiv(B, 2), 64 * kTrunkWaves, 0, st>>>(d);
    else dec_cell_kernel<OpF32><<<cdiv(B, 2), 64 * kTrunkWaves, 0, st>>>(d);
  }
  LAUNCH_CHECK();
  const int nch = dec_chunks(G);
  DecGeneArgs g;
  g.genes = genes; g.emb = h->emb; g.qtab = h->qtab; g.theta_emb = h->theta; g.kvfrag = kv; g.wfrag = h->frag_dec; g.wfrag_cproj_halves = h->frag_dec_halves;
  g.ln2_w = h->small + S_DEC_LN2W; g.ln2_b = h->small + S_DEC_LN2B; g.head_w = h->small + S_HEAD_W; g.head_b = h->small + S_HEAD_B;
  const bool unshared = h->theta == nullptr;
  g.head_w2 = unshared ? h->small + S_HEAD_W2 : nullptr;
  float* theta_rows = (unshared && draw) ? (float*)((char*)ws_ + align256((size_t)B * 48 * 64 * 4) + align256((size_t)B * dec_chunks(G) * 2 * 4) +
                                                    align256((size_t)B * 16 * 32 * 4)) : nullptr;
  g.logits = mu; g.theta = draw ? theta_rows : theta; g.part = part; g.G = G; g.n_chunks = nch; g.tiles_per_wave = kDecTilesPerWave;
  g.eps = c.layernorm_eps; g.inv_temp = 1.0f / c.nb_temperature;
  {
    KernelTimer kt(h, SCLDM_VAE_K_DEC_GENE, st);
    if (precision == SCLDM_PREC_BF16) dec_gene_kernel<OpBF16><<<dim3(nch, B), kDecThreads, 0, st>>>(g);
    else if (precision == SCLDM_PREC_FP16) dec_gene_kernel<OpFP16><<<dim3(nch, B), kDecThreads, 0, st>>>(g);
    else dec_gene_kernel<OpF32><<<dim3(nch, B), kDecThreads, 0, st>>>(g);
  }
  LAUNCH_CHECK();
  {
    KernelTimer kt(h, SCLDM_VAE_K_DEC_FINAL, st);
    if (draw) dec_finalize_sample_kernel<<<dim3(cdiv(G, 256 * 4), B), 256, 0, st>>>(mu, part, library_size, genes, h->theta, theta_rows, G, nch, seed);
    else dec_finalize_kernel<<<dim3(cdiv(G, 256 * 16), B), 256, 0, st>>>(mu, part, library_size, G, nch);   // four 16-byte groups per thread
  }
  LAUNCH_CHECK();
  return SCLDM_OK;
}

extern "C" int scldm_vae_decode(scldm_vae* h, const float* z, const int64_t* genes, const float* library_size, int B, int G,
                                float* mu, float* theta, int precision, void* ws_, void* stream_) {
  return vae_decode_impl(h, z, genes, library_size, B, G, mu, theta, false, 0ull, precision, ws_, stream_);
}

extern "C" int scldm_vae_decode_sample(scldm_vae* h, const float* z, const int64_t* genes, const float* library_size, int B, int G,
                                       float* counts, unsigned long long seed, int precision, void* ws_, void* stream_) {
  return vae_decode_impl(h, z, genes, library_size, B, G, counts, nullptr, true, seed, precision, ws_, stream_);
}

extern "C" int scldm_nb_sample(const float* mu, const float* theta, float* out, size_t n, unsigned long long seed, void* stream_) {
  if (!mu || !theta || !out) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (n == 0) return SCLDM_OK;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 256 * 32);
  nb_sample_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(mu, theta, out, n, seed);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

// ---- encoder input path: tokenize_cells(sample_genes="expressed") --------------------------------------------------
#include "tokenize.hpp"

extern "C" int scldm_tokenize_expressed(const float* counts, const int64_t* gene_idx, long gene_row_stride, int N, int G, int S,
                                        int64_t mask_idx, int64_t* genes_subset, float* counts_subset, int32_t* num_expressed,
                                        float* library_size, void* stream_) {
  if (!counts || !gene_idx || !genes_subset || !counts_subset || !num_expressed || !library_size) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (N < 0 || G < 1 || S < 1) return fail(SCLDM_ERR_SHAPE, "need N >= 0, G >= 1, genes_seq_len >= 1 (got %d, %d, %d)", N, G, S);
  if (gene_row_stride != 0 && gene_row_stride != G) return fail(SCLDM_ERR_SHAPE, "gene_row_stride must be 0 (shared row) or G");
  if (N == 0) return SCLDM_OK;
  hipLaunchKernelGGL(tokenize_expressed_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream_, counts, gene_idx, gene_row_stride, G, S,
                     mask_idx, genes_subset, counts_subset, num_expressed, library_size);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

// ---- output assembly: dense generated counts -> CSR ----------------------------------------------------------------
extern "C" int scldm_csr_count(const float* dense, int N, int G, int32_t* row_nnz, void* stream_) {
  if (!dense || !row_nnz) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (N < 1 || G < 1) return fail(SCLDM_ERR_SHAPE, "need N >= 1 and G >= 1 (got %d, %d)", N, G);
  hipLaunchKernelGGL(csr_count_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream_, dense, G, row_nnz);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
extern "C" int scldm_csr_fill(const float* dense, int N, int G, const int64_t* indptr, int32_t* indices, float* data, void* stream_) {
  if (!dense || !indptr || !indices || !data) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (N < 1 || G < 1) return fail(SCLDM_ERR_SHAPE, "need N >= 1 and G >= 1 (got %d, %d)", N, G);
  hipLaunchKernelGGL(csr_fill_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream_, dense, G, indptr, indices, data);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

// ---- generation-evaluation MMD kernels -------------------------------------------------------------------------------
#include "mmd.hpp"

extern "C" size_t scldm_mmd_workspace_bytes(int nx, int ny) {
  if (nx < 1 || ny < 1) return 0;
  return align256((size_t)cdiv(nx, kMmdTile) * cdiv(ny, kMmdTile) * sizeof(float));
}

extern "C" int scldm_mmd_kernel_sum(const float* x, int nx, const float* y, int ny, int D, int kind, float scale, double* sum_out,
                                    float* kmat, void* ws, void* stream_) {
  if (!x || !y || !sum_out || !ws) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (nx < 1 || ny < 1 || D < 1) return fail(SCLDM_ERR_SHAPE, "need nx, ny, D >= 1 (got %d, %d, %d)", nx, ny, D);
  hipStream_t st = (hipStream_t)stream_;
  dim3 grid(cdiv(ny, kMmdTile), cdiv(nx, kMmdTile));
  float* partial = reinterpret_cast<float*>(ws);
  switch (kind) {
    case kMmdRbf: hipLaunchKernelGGL(mmd_tile_kernel<kMmdRbf>, grid, dim3(256), 0, st, x, nx, y, ny, D, scale, partial, kmat); break;
    case kMmdBrayCurtis: hipLaunchKernelGGL(mmd_tile_kernel<kMmdBrayCurtis>, grid, dim3(256), 0, st, x, nx, y, ny, D, scale, partial, kmat); break;
    case kMmdTanimoto: hipLaunchKernelGGL(mmd_tile_kernel<kMmdTanimoto>, grid, dim3(256), 0, st, x, nx, y, ny, D, scale, partial, kmat); break;
    case kMmdRuzicka: hipLaunchKernelGGL(mmd_tile_kernel<kMmdRuzicka>, grid, dim3(256), 0, st, x, nx, y, ny, D, scale, partial, kmat); break;
    case kMmdSqDist: hipLaunchKernelGGL(mmd_tile_kernel<kMmdSqDist>, grid, dim3(256), 0, st, x, nx, y, ny, D, scale, partial, kmat); break;
    case kMmdDist: hipLaunchKernelGGL(mmd_tile_kernel<kMmdDist>, grid, dim3(256), 0, st, x, nx, y, ny, D, scale, partial, kmat); break;
    default: return fail(SCLDM_ERR_SHAPE, "unknown MMD kernel kind %d", kind);
  }
  LAUNCH_CHECK();
  hipLaunchKernelGGL(mmd_sum_kernel, dim3(1), dim3(256), 0, st, partial, (int)(grid.x * grid.y), sum_out);
  LAUNCH_CHECK();
  return SCLDM_OK;
}

// ---- entropic optimal transport (Sinkhorn) for the Wasserstein generation metrics -----------------------------------
#include "sinkhorn.hpp"

static size_t sk_layout(int n, int m, size_t* off) {   // M | K | u | v | ktu | part | rowcost | scalars | mmd scratch | u_new | v_new
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += align256(bytes); return r; };
  off[0] = take((size_t)n * m * 4);
  off[1] = take((size_t)n * m * 4);
  off[2] = take((size_t)n * 4);
  off[3] = take((size_t)m * 4);
  off[4] = take((size_t)m * 4);
  off[5] = take((size_t)kSkRowSplit * m * 4);
  off[6] = take((size_t)n * 4);
  off[7] = take(256);
  off[8] = take(scldm_mmd_workspace_bytes(n, m));
  off[9] = take((size_t)n * 4);
  off[10] = take((size_t)m * 4);
  return o;
}
extern "C" size_t scldm_sinkhorn_workspace_bytes(int n, int m) {
  if (n < 1 || m < 1) return 0;
  size_t off[11];
  return sk_layout(n, m, off);
}

extern "C" int scldm_wasserstein_sinkhorn(const float* x0, int n, const float* x1, int m, int D, int power, float reg,
                                          long long num_iter_max, float stop_thr, double* cost_out, long long* iters_out,
                                          int* status_out, void* ws, void* stream_) {
  if (!x0 || !x1 || !cost_out || !ws) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (n < 1 || m < 1 || D < 1) return fail(SCLDM_ERR_SHAPE, "need n, m, D >= 1 (got %d, %d, %d)", n, m, D);
  if (power != 1 && power != 2) return fail(SCLDM_ERR_SHAPE, "power must be 1 or 2 (evaluations.py:92)");
  if (!(reg > 0.f)) return fail(SCLDM_ERR_SHAPE, "reg must be positive");
  hipStream_t st = (hipStream_t)stream_;
  size_t off[11];
  sk_layout(n, m, off);
  char* base = (char*)ws;
  float *M = (float*)(base + off[0]), *K = (float*)(base + off[1]), *u = (float*)(base + off[2]), *v = (float*)(base + off[3]);
  float *ktu = (float*)(base + off[4]), *part = (float*)(base + off[5]), *rowcost = (float*)(base + off[6]);
  float* err2 = (float*)(base + off[7]);
  int* flag = (int*)(base + off[7] + 64);
  double* dsum = (double*)(base + off[7] + 128);
  void* mmd_ws = base + off[8];
  // M = cdist(x0, x1) ** power  (evaluations.py:103-105)
  int rc = scldm_mmd_kernel_sum(x0, n, x1, m, D, power == 2 ? kMmdSqDist : kMmdDist, 1.0f, dsum, M, mmd_ws, st);
  if (rc) return rc;
  const size_t nm = (size_t)n * m;
  sk_gibbs_kernel<<<(int)std::min<size_t>((nm + 255) / 256, 256 * 16), 256, 0, st>>>(M, K, nm, -1.0f / reg);
  const float a = 1.0f / n, b = 1.0f / m;
  sk_fill_kernel<<<cdiv(n, 256), 256, 0, st>>>(u, n, a);
  sk_fill_kernel<<<cdiv(m, 256), 256, 0, st>>>(v, m, b);
  HIP_TRY(hipMemsetAsync(flag, 0, 4, st));
  LAUNCH_CHECK();
  // an iteration computes candidate scalings and commits them on device unless one of them was singular: the state then
  // freezes at the last good (u, v), which is what POT's per-iteration "numerical errors" exit returns
  float *u_new = (float*)(base + off[9]), *v_new = (float*)(base + off[10]);   // (in the caller's workspace: no allocation in the call)
  long long it = 0;
  int status = 0;   // 0 converged, 1 iteration limit, 2 numerical breakdown (previous scalings kept)
  bool have_ktu = false;
  float best_err = 3.0e38f;
  int stalled = 0;
  const dim3 gk(cdiv(m, 256), kSkRowSplit);
  for (; it < num_iter_max; ++it) {
    if (!have_ktu) sk_ktu_kernel<<<gk, 256, 0, st>>>(K, u, n, m, part);
    sk_v_update_kernel<<<cdiv(m, 256), 256, 0, st>>>(part, m, b, ktu, v_new, flag);
    sk_u_update_kernel<<<cdiv(n, 4), 256, 0, st>>>(K, v_new, n, m, a, u_new, flag);
    sk_commit_kernel<<<cdiv(std::max(n, m), 256), 256, 0, st>>>(u, u_new, n, v, v_new, m, flag);
    have_ktu = false;
    if (it % 10 == 0) {
      sk_ktu_kernel<<<gk, 256, 0, st>>>(K, u, n, m, part);   // K^T u for the committed u: the error term now, the next iteration's input
      have_ktu = true;
      sk_err_kernel<<<1, 256, 0, st>>>(part, v, m, b, err2);
      LAUNCH_CHECK();
      float h_err2 = 0.f;
      int h_flag = 0;
      HIP_TRY(hipMemcpyAsync(&h_err2, err2, 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (h_flag) { status = 2; break; }
      const float err = sqrtf(h_err2);
      if (!(err >= stop_thr)) { ++it; break; }   // converged
      // fp32 floor: the marginal error of an fp32 iterate stops improving around 1e-8..1e-9 * sqrt(m) and may never cross
      // POT's 1e-9; three checks (30 iterations) without a 1 % improvement end the loop instead of spinning to numItermax
      if (err < 0.99f * best_err) { best_err = err; stalled = 0; }
      else if (++stalled >= 3 && err < 1e-6f) { ++it; break; }
    }
  }
  if (it >= num_iter_max && status == 0) status = 1;
  sk_cost_kernel<<<cdiv(n, 4), 256, 0, st>>>(K, M, u, v, n, m, rowcost);
  mmd_sum_kernel<<<1, 256, 0, st>>>(rowcost, n, dsum);
  LAUNCH_CHECK();
  HIP_TRY(hipMemcpyAsync(cost_out, dsum, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (iters_out) *iters_out = it;
  if (status_out) *status_out = status;
  return SCLDM_OK;
}
