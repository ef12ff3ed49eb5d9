// C ABI of the TransformerVAE training step (scldm_vae_train_*, scldm_nb_loglik*; see include/scldm_hip.h): host-side
// sequencing of the kernels in vae_train.hpp.  The forward is the inference path (mcab.hpp) with two extra outputs; the backward
// reads the parameters LIVE from the caller's tensors (PyTorch layouts) and writes every gradient in the same layout.
#include <algorithm>
#include <atomic>
#include <vector>

#include "vae_handle.hpp"
#include "vae_train_wide.hpp"

using namespace scldm;
using namespace scldm::vtrain;

namespace {

struct Carver {
  char* base;
  size_t off = 0;
  float* take(size_t floats) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align256(floats * sizeof(float));
    return p;
  }
};

// tiles of 64 tokens per workgroup / workgroups per cell of the two gene-axis kernels: about 2048 workgroups in flight, so that
// one partial per workgroup stays small next to the work it summarises
void split_tiles(int n_tok, int B, int* tiles, int* chunks, int wgs = 2048) {
  const int per_cell = cdiv(n_tok, 64);
  const int want = std::max(1, std::min(per_cell, wgs / std::max(B, 1)));
  *tiles = cdiv(per_cell, want);
  *chunks = cdiv(per_cell, *tiles);
}

struct Saved { float *pooled, *lse2, *kv; size_t bytes; };
Saved carve_saved(int B, void* base) {
  Carver c{reinterpret_cast<char*>(base)};
  Saved s;
  s.pooled = c.take((size_t)B * kT * 32);
  s.lse2 = c.take((size_t)B * 64);
  s.kv = c.take((size_t)B * kT * 64);   // K | V of the cells' latent tokens (the decode's plain (B, 16, 64) copy), for the per-gene backward
  s.bytes = c.off;
  return s;
}

// SCLDM_VAE_CELL_WIDE=0: the first version of the cell-side kernels (one token per lane, four cells per wave) for A/B runs
bool cell_wide() {
  static const bool on = [] { const char* e = getenv("SCLDM_VAE_CELL_WIDE"); return !(e && e[0] == '0'); }();
  return on;
}

// SCLDM_VAE_GENE_WIDE=0: the first version of the gene-axis backward kernels (one token per lane)
bool gene_wide() {
  static const bool on = [] { const char* e = getenv("SCLDM_VAE_GENE_WIDE"); return !(e && e[0] == '0'); }();
  return on;
}

struct Ws {
  float *wct, *Q, *dQ, *xs_enc, *xs_dec, *ysave, *kv, *dl, *dz_dec, *dao, *dgq, *bsum, *p_gene, *p_dkv, *p_dcell, *p_ecell, *p_pool;
  int tilesD, chunksD, tilesE, chunksE, quads, cparts;
  size_t bytes;
};
Ws carve_ws(const scldm_vae* h, int B, int S, int G, void* base) {
  const scldm_vae_config& c = h->cfg;
  const int L = c.n_layer;
  Carver k{reinterpret_cast<char*>(base)};
  Ws w;
  // (the second version's workgroups are four waves, two per CU: one full round of 512 measured best at batch 32 - per-gene kernel
  // 606 us against 633-645 us with 1 024-4 096 workgroups, pooling 123 us against 130-138 us - and equal at batch 512;
  // SCLDM_VAE_GENE_WGS / SCLDM_VAE_POOL_WGS override the target counts)
  static const int wgs_d = [] { const char* e = getenv("SCLDM_VAE_GENE_WGS"); return e ? atoi(e) : 512; }();
  static const int wgs_e = [] { const char* e = getenv("SCLDM_VAE_POOL_WGS"); return e ? atoi(e) : 512; }();
  split_tiles(G, B, &w.tilesD, &w.chunksD, gene_wide() ? wgs_d : 2048);
  split_tiles(S, B, &w.tilesE, &w.chunksE, gene_wide() ? wgs_e : 2048);
  w.quads = cdiv(B, 4);
  w.wct = k.take((size_t)(2 + 2 * L) * kHP * 32);
  w.Q = k.take(512);
  w.dQ = k.take(512);
  w.xs_enc = k.take((size_t)B * (L + 1) * kT * 32);
  w.xs_dec = k.take((size_t)B * (L + 1) * kT * 32);
  w.ysave = k.take((size_t)B * kT * 32);
  w.kv = k.take((size_t)B * kT * 64);
  w.dl = k.take((size_t)B * G);
  w.dz_dec = k.take((size_t)B * kT * c.n_embed_latent);
  w.dao = k.take((size_t)B * kT * 32);
  w.dgq = k.take((size_t)B * 64);
  w.bsum = k.take((size_t)B);
  w.p_gene = k.take((size_t)B * w.chunksD * DP_SIZE);
  w.p_dkv = k.take((size_t)B * w.chunksD * kT * 64);
  w.cparts = cell_wide() ? B : w.quads;     // cell-side partials: one per cell (vae_train_wide.hpp) or one per four cells
  w.p_dcell = k.take((size_t)w.cparts * dc_size(L));
  w.p_ecell = k.take((size_t)w.cparts * ec_size(L));
  w.p_pool = k.take((size_t)B * w.chunksE * EP_SIZE);
  w.bytes = k.off;
  return w;
}

int check_train(const scldm_vae* h, const scldm_vae_weights* w, int B, int S, int G) {
  if (!h || !w) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (!h->loaded) return fail(SCLDM_ERR_STATE, "scldm_vae_load_weights has not been called");
  if (B < 1 || S < 1 || G < 1) return fail(SCLDM_ERR_SHAPE, "need B, S, G >= 1 (got %d, %d, %d)", B, S, G);
  if (h->cfg.n_layer > 16) return fail(SCLDM_ERR_SHAPE, "the VAE training kernels take at most 16 trunk layers per side (got %d)", h->cfg.n_layer);
  if (h->cfg.hidden_dim > kHP) return fail(SCLDM_ERR_SHAPE, "SwiGLU hidden width %d exceeds %d", h->cfg.hidden_dim, kHP);
  if (B > 65535) return fail(SCLDM_ERR_SHAPE, "batch too large for one launch (%d)", B);
  return SCLDM_OK;
}

MlpW mlp_of(const float* w1, const float* w2, const float* wct, int H) { return MlpW{w1, w2, wct, H}; }

BlockWArr blocks_of(const scldm_vae_block* b, int L, const float* wct0, int H) {
  BlockWArr a{};
  for (int l = 0; l < L; ++l) {
    a.b[l].ln1_w = b[l].ln1_w; a.b[l].ln1_b = b[l].ln1_b; a.b[l].wqkv = b[l].attn_w; a.b[l].wp = b[l].proj_w;
    a.b[l].ln2_w = b[l].ln2_w; a.b[l].ln2_b = b[l].ln2_b;
    a.b[l].mlp = mlp_of(b[l].w1, b[l].w2, wct0 + (size_t)l * kHP * 32, H);
  }
  return a;
}

// reduction jobs of one partial array (launched in groups of at most kMaxRedJobs)
struct Jobs {
  std::vector<RedJob> v;
  void vec(float* dst, int off, int n, bool acc = false) { if (dst) v.push_back(RedJob{dst, off, n, acc ? 1 : 0, 1, 0, 0}); }
  void mat(float* dst, int off, int rows, int cols, int ld_src, int ld_dst, bool acc = false) {
    if (dst) v.push_back(RedJob{dst, off, cols, acc ? 1 : 0, rows, ld_src, ld_dst});
  }
  int run(const float* part, int n_part, long stride, hipStream_t st) {
    for (size_t i = 0; i < v.size(); i += kMaxRedJobs) {
      RedArgs a{};
      a.part = part; a.n_part = n_part; a.stride = stride;
      a.n_jobs = (int)std::min<size_t>(kMaxRedJobs, v.size() - i);
      for (int j = 0; j < a.n_jobs; ++j) a.job[j] = v[i + j];
      reduce_jobs_kernel<<<dim3(12, a.n_jobs), 256, 0, st>>>(a);   // (3 072 threads per job: one element each for the largest blocks)
      LAUNCH_CHECK();
    }
    return SCLDM_OK;
  }
};

// gradient pointer of a trunk layer's tensors, in the partial layout of block_bwd
void trunk_jobs(Jobs& j, const scldm_vae_block& g, int off0, int H) {
  j.mat(const_cast<float*>(g.attn_w), off0 + TP_WQKV, 96, 32, 32, 32);
  j.mat(const_cast<float*>(g.proj_w), off0 + TP_WP, 32, 32, 32, 32);
  j.mat(const_cast<float*>(g.w1), off0 + TP_W1, H, 32, 32, 32);
  j.mat(const_cast<float*>(g.w2), off0 + TP_W2, H, 32, 32, 32);
  j.mat(const_cast<float*>(g.cproj), off0 + TP_WC, 32, H, kHP, H);
  j.vec(const_cast<float*>(g.ln1_w), off0 + TP_LN1W, 32);
  j.vec(const_cast<float*>(g.ln1_b), off0 + TP_LN1B, 32);
  j.vec(const_cast<float*>(g.ln2_w), off0 + TP_LN2W, 32);
  j.vec(const_cast<float*>(g.ln2_b), off0 + TP_LN2B, 32);
}

}  // namespace

extern "C" size_t scldm_vae_train_saved_bytes(const scldm_vae* h, int B) {
  if (!h || B < 1) return 0;
  return carve_saved(B, nullptr).bytes;
}
extern "C" size_t scldm_vae_train_workspace_bytes(const scldm_vae* h, int B, int S, int G) {
  if (!h || B < 1 || S < 1 || G < 1) return 0;
  // the forward borrows the inference workspace layout; the backward carves its own
  return std::max(carve_ws(h, B, S, G, nullptr).bytes, scldm_vae_workspace_bytes(h, B, G));
}

extern "C" int scldm_vae_train_forward(scldm_vae* h, const float* counts_subset, const int64_t* genes_subset, int B, int S,
                                       const int64_t* genes, const float* library_size, int G, float* mu, float* theta, float* z,
                                       void* saved_, void* ws, void* stream_) {
  if (!h || !counts_subset || !genes_subset || !genes || !library_size || !mu || !theta || !z || !saved_ || !ws)
    return fail(SCLDM_ERR_SHAPE, "null argument");
  if (B < 1 || S < 1 || G < 1) return fail(SCLDM_ERR_SHAPE, "need B, S, G >= 1");
  if (h->loaded && !h->theta)
    return fail(SCLDM_ERR_STATE, "the training backward is built for the shared-theta NB head (vae_base.yaml:62); the unshared head decodes only");
  hipStream_t st = (hipStream_t)stream_;
  Saved s = carve_saved(B, saved_);
  int rc = scldm_vae_encode_ex(h, counts_subset, genes_subset, B, S, z, SCLDM_PREC_FP32, s.pooled, s.lse2, st);
  if (rc) return rc;
  rc = scldm_vae_decode(h, z, genes, library_size, B, G, mu, theta, SCLDM_PREC_FP32, ws, stream_);
  if (rc) return rc;
  // the decode's fp32 route leaves the cells' plain K | V at the start of its workspace (vae_api.hip: vae_decode_impl): keep a copy in
  // `saved` so that the per-gene backward does not wait for the recompute forward (4 KB per cell)
  HIP_TRY(hipMemcpyAsync(s.kv, ws, (size_t)B * kT * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));
  return SCLDM_OK;
}

extern "C" int scldm_vae_train_backward(scldm_vae* h, const scldm_vae_weights* w, const scldm_vae_weights* g, const float* counts_subset,
                                        const int64_t* genes_subset, int B, int S, const int64_t* genes, const float* library_size, int G,
                                        const float* mu, const float* theta, const float* z, const float* dmu, const float* dtheta,
                                        const float* dz, void* saved_, void* ws_, void* stream_) {
  int rc = check_train(h, w, B, S, G);
  if (rc) return rc;
  if (!g || !counts_subset || !genes_subset || !genes || !library_size || !mu || !theta || !z || !saved_ || !ws_)
    return fail(SCLDM_ERR_SHAPE, "null argument");
  if (!g->gene_embedding || !g->theta) return fail(SCLDM_ERR_SHAPE, "the gene_embedding and theta gradient tables are required");
  hipStream_t st = (hipStream_t)stream_;
  const scldm_vae_config& c = h->cfg;
  const int L = c.n_layer, H = c.hidden_dim, nl = c.n_embed_latent;
  const float eps = c.layernorm_eps;
  Saved sv = carve_saved(B, saved_);
  Ws k = carve_ws(h, B, S, G, ws_);
  float* g_emb = const_cast<float*>(g->gene_embedding);
  float* g_theta = const_cast<float*>(g->theta);
  HIP_TRY(hipMemsetAsync(g_emb, 0, (size_t)(c.n_genes + 1) * 32 * 4, st));
  HIP_TRY(hipMemsetAsync(g_theta, 0, (size_t)(c.n_genes + 1) * 4, st));

  // ---- transposed c_proj copies (the streaming SwiGLU reads columns of c_proj as rows): enc cross, dec cross, enc layers, dec layers
  auto wct = [&](int i) { return k.wct + (size_t)i * kHP * 32; };
  {
    TransposeJobs tj{};     // slots: enc cross, dec cross, enc layers, dec layers (one launch)
    tj.src[0] = w->enc_cross.cproj;
    tj.src[1] = w->dec_cross.cproj;
    for (int l = 0; l < L; ++l) { tj.src[2 + l] = w->enc_blocks[l].cproj; tj.src[2 + L + l] = w->dec_blocks[l].cproj; }
    transpose_many_kernel<<<dim3(cdiv(32 * H, 256), 2 + 2 * L), 256, 0, st>>>(tj, 32, H, k.wct, (long)kHP * 32);
    enc_q_fwd_kernel<<<1, 64, 0, st>>>(w->inducing_points, w->enc_cross.ln1q_w, w->enc_cross.ln1q_b, w->enc_cross.attn_q, eps, k.Q);
    LAUNCH_CHECK();
  }
  // ---- recompute the 16-token sides with saved layer inputs
  EncCellTrainArgs ea{};
  ea.pooled = sv.pooled; ea.ind = w->inducing_points; ea.wp = w->enc_cross.attn_proj; ea.cln2_w = w->enc_cross.ln2_w; ea.cln2_b = w->enc_cross.ln2_b;
  ea.cmlp = mlp_of(w->enc_cross.w1, w->enc_cross.w2, wct(0), H);
  ea.pos = c.positional_encoding ? w->enc_pos_embed : nullptr;
  ea.blocks = blocks_of(w->enc_blocks, L, wct(2), H);
  ea.w_lat = w->enc_latent_w; ea.xsave = k.xs_enc; ea.ysave = k.ysave;
  ea.dz_a = k.dz_dec; ea.dz_b = dz; ea.dao = k.dao; ea.dgq = k.dgq; ea.part = k.p_ecell;
  ea.B = B; ea.n_lat = nl; ea.n_layer = L; ea.eps = eps;
  const bool wd = cell_wide();
  {   // (the attribute is per device: a process that drives several GPUs sets it on each)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
      for (const void* f : {(const void*)wide::enc_cell_fwd_kernel, (const void*)wide::enc_cell_bwd_kernel, (const void*)wide::dec_cell_fwd_kernel,
                            (const void*)wide::dec_cell_bwd_kernel})
        HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, wide::LDS_BYTES));
      HIP_TRY(hipFuncSetAttribute((const void*)wide::enc_pool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wide::PB_BYTES));
      HIP_TRY(hipFuncSetAttribute((const void*)wide::dec_gene_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wide::G_BYTES));
      HIP_TRY(hipFuncSetAttribute((const void*)wide::dec_gene_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wide::M_BYTES));
      HIP_TRY(hipFuncSetAttribute((const void*)wide::dec_gene_bwd_mfma2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wide::M_BYTES));
      if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
  }
  // The recompute forwards of the two 16-token sides (B workgroups each) only feed the cell-side backward kernels: they run on a second
  // stream beside the head and per-gene backward, which take the cells' K | V from the copy the forward left in `saved`.
  static const bool overlap = [] { const char* e = getenv("SCLDM_VAE_TRAIN_OVERLAP"); return !(e && e[0] == '0'); }();
  hipStream_t s2 = st;
  if (overlap) {
    if (!h->side) {
      HIP_TRY(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_gene, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    s2 = h->side;
    HIP_TRY(hipEventRecord(h->ev_fork, st));
    HIP_TRY(hipStreamWaitEvent(s2, h->ev_fork, 0));
  }
  // whatever path leaves this function, the caller's stream waits for everything the second stream was given (the caller frees the
  // workspace in stream order of `st` only)
  struct JoinSide {
    scldm_vae* h; hipStream_t st; bool on;
    ~JoinSide() {
      if (!on) return;
      if (hipEventRecord(h->ev_fork, h->side) == hipSuccess) (void)hipStreamWaitEvent(st, h->ev_fork, 0);
    }
  } join_side{h, st, overlap};
  if (wd) {
    wide::enc_cell_fwd_kernel<<<B, wide::kThreads, wide::LDS_BYTES, s2>>>(ea);
  } else enc_cell_fwd_kernel<<<k.quads, 64, 0, s2>>>(ea);
  LAUNCH_CHECK();
  DecCellTrainArgs da{};
  da.z = z; da.w_in = w->dec_latent_w; da.blocks = blocks_of(w->dec_blocks, L, wct(2 + L), H);
  da.cln1_w = w->dec_cross.ln1_w; da.cln1_b = w->dec_cross.ln1_b; da.wkv = w->dec_cross.attn_kv;
  da.xsave = k.xs_dec; da.kv = k.kv; da.dkv_part = k.p_dkv; da.chunks = k.chunksD; da.dz = k.dz_dec; da.part = k.p_dcell;
  da.B = B; da.n_lat = nl; da.n_layer = L; da.eps = eps;
  if (wd) wide::dec_cell_fwd_kernel<<<B, wide::kThreads, wide::LDS_BYTES, s2>>>(da);
  else if (nl <= 16) dec_cell_fwd_kernel<16><<<k.quads, 64, 0, s2>>>(da);
  else dec_cell_fwd_kernel<32><<<k.quads, 64, 0, s2>>>(da);
  LAUNCH_CHECK();
  if (overlap) HIP_TRY(hipEventRecord(h->ev_join, s2));
  // ---- NB head, then the per-gene decoder chain
  head_bwd_kernel<<<B, 256, 0, st>>>(mu, theta, dmu, dtheta, library_size, genes, G, 1.0f / c.nb_temperature, k.dl, g_theta, k.bsum);
  LAUNCH_CHECK();
  DecBwdArgs ga{};
  ga.genes = genes; ga.emb = w->gene_embedding; ga.dl = k.dl; ga.kv = overlap ? sv.kv : k.kv;
  ga.ln1q_w = w->dec_cross.ln1q_w; ga.ln1q_b = w->dec_cross.ln1q_b; ga.wq = w->dec_cross.attn_q; ga.wp = w->dec_cross.attn_proj;
  ga.ln2_w = w->dec_cross.ln2_w; ga.ln2_b = w->dec_cross.ln2_b; ga.head_w = w->head_w;
  ga.mlp = mlp_of(w->dec_cross.w1, w->dec_cross.w2, wct(1), H);
  ga.g_emb = g_emb; ga.part = k.p_gene; ga.dkv_part = k.p_dkv; ga.G = G; ga.tiles = k.tilesD; ga.eps = eps;
  // 2 (default): every contraction of the per-gene chain on the matrix pipe; 1: the MLP and the weight gradients only; 0: the VALU form
  static const int gene_mfma = [] { const char* e = getenv("SCLDM_VAE_GENE_MFMA"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 2; }();
  if (gene_wide() && gene_mfma == 2) wide::dec_gene_bwd_mfma2_kernel<<<dim3(k.chunksD, B), wide::kThreads, wide::M_BYTES, st>>>(ga);
  else if (gene_wide() && gene_mfma) wide::dec_gene_bwd_mfma_kernel<<<dim3(k.chunksD, B), wide::kThreads, wide::M_BYTES, st>>>(ga);
  else if (gene_wide()) wide::dec_gene_bwd_kernel<<<dim3(k.chunksD, B), wide::kThreads, wide::G_BYTES, st>>>(ga);
  else dec_gene_bwd_kernel<<<dim3(k.chunksD, B), 64, 0, st>>>(ga);
  LAUNCH_CHECK();
  if (overlap) {
    HIP_TRY(hipEventRecord(h->ev_gene, st));
    HIP_TRY(hipStreamWaitEvent(st, h->ev_join, 0));
  }
  if (wd) wide::dec_cell_bwd_kernel<<<B, wide::kThreads, wide::LDS_BYTES, st>>>(da);
  else if (nl <= 16) dec_cell_bwd_kernel<16><<<k.quads, 64, 0, st>>>(da);
  else dec_cell_bwd_kernel<32><<<k.quads, 64, 0, st>>>(da);
  LAUNCH_CHECK();
  if (wd) wide::enc_cell_bwd_kernel<<<B, wide::kThreads, wide::LDS_BYTES, st>>>(ea);
  else if (nl <= 16) enc_cell_bwd_kernel<16><<<k.quads, 64, 0, st>>>(ea);
  else enc_cell_bwd_kernel<32><<<k.quads, 64, 0, st>>>(ea);
  LAUNCH_CHECK();
  EncPoolBwdArgs pa{};
  pa.counts = counts_subset; pa.genes = genes_subset; pa.emb = w->gene_embedding;
  pa.ln1_w = w->enc_cross.ln1_w; pa.ln1_b = w->enc_cross.ln1_b; pa.wkv = w->enc_cross.attn_kv; pa.Q = k.Q; pa.lse2 = sv.lse2;
  pa.dao = k.dao; pa.dgq = k.dgq; pa.g_emb = g_emb; pa.part = k.p_pool; pa.S = S; pa.tiles = k.tilesE; pa.eps = eps;
  if (gene_wide()) wide::enc_pool_bwd_kernel<<<dim3(k.chunksE, B), wide::kThreads, wide::PB_BYTES, st>>>(pa);
  else enc_pool_bwd_kernel<<<dim3(k.chunksE, B), 64, 0, st>>>(pa);
  LAUNCH_CHECK();

  // ---- partial sums -> parameter gradients
  auto G_ = [](const float* p) { return const_cast<float*>(p); };
  {
    Jobs j;   // per-gene decoder chain
    const scldm_vae_cross& gc = g->dec_cross;
    j.mat(G_(gc.attn_q), DP_WQ, 32, 32, 32, 32);
    j.mat(G_(gc.attn_proj), DP_WP, 32, 32, 32, 32);
    j.mat(G_(gc.w1), DP_W1, H, 32, 32, 32);
    j.mat(G_(gc.w2), DP_W2, H, 32, 32, 32);
    j.mat(G_(gc.cproj), DP_WC, 32, H, kHP, H);
    j.vec(G_(gc.ln1q_w), DP_LN1QW, 32);
    j.vec(G_(gc.ln1q_b), DP_LN1QB, 32);
    j.vec(G_(gc.ln2_w), DP_LN2W, 32);
    j.vec(G_(gc.ln2_b), DP_LN2B, 32);
    j.vec(G_(g->head_w), DP_HEADW, 32);
    // (on the second stream, beside the cell-side backward kernels: it only needs the per-gene kernel's partials)
    if (overlap) HIP_TRY(hipStreamWaitEvent(s2, h->ev_gene, 0));
    if ((rc = j.run(k.p_gene, B * k.chunksD, DP_SIZE, s2))) return rc;
    Jobs jb;
    jb.vec(G_(g->head_b), 0, 1);
    if ((rc = jb.run(k.bsum, B, 1, s2))) return rc;
  }
  {
    Jobs j;   // decoder cell side
    for (int l = 0; l < L; ++l) trunk_jobs(j, g->dec_blocks[l], l * TP_SIZE, H);
    j.mat(G_(g->dec_cross.attn_kv), dc_off_wkv(L), 64, 32, 32, 32);
    j.vec(G_(g->dec_cross.ln1_w), dc_off_cln1w(L), 32);
    j.vec(G_(g->dec_cross.ln1_b), dc_off_cln1b(L), 32);
    j.mat(G_(g->dec_latent_w), dc_off_win(L), 32, nl, 32, nl);
    if ((rc = j.run(k.p_dcell, k.cparts, dc_size(L), st))) return rc;
  }
  {
    Jobs j;   // encoder cell side
    for (int l = 0; l < L; ++l) trunk_jobs(j, g->enc_blocks[l], l * TP_SIZE, H);
    const scldm_vae_cross& gc = g->enc_cross;
    j.mat(G_(gc.attn_proj), ec_off_wp(L), 32, 32, 32, 32);
    j.mat(G_(gc.w1), ec_off_w1(L), H, 32, 32, 32);
    j.mat(G_(gc.w2), ec_off_w2(L), H, 32, 32, 32);
    j.mat(G_(gc.cproj), ec_off_wc(L), 32, H, kHP, H);
    j.vec(G_(gc.ln2_w), ec_off_ln2w(L), 32);
    j.vec(G_(gc.ln2_b), ec_off_ln2b(L), 32);
    j.mat(G_(g->enc_latent_w), ec_off_wlat(L), nl, 32, 32, 32);
    j.vec(G_(g->inducing_points), ec_off_ind(L), 512);
    if ((rc = j.run(k.p_ecell, k.cparts, ec_size(L), st))) return rc;
  }
  {
    Jobs j;   // encoder pooling
    j.mat(G_(g->enc_cross.attn_kv), EP_WKV, 64, 32, 32, 32);
    j.vec(G_(g->enc_cross.ln1_w), EP_LN1W, 32);
    j.vec(G_(g->enc_cross.ln1_b), EP_LN1B, 32);
    if ((rc = j.run(k.p_pool, B * k.chunksE, EP_SIZE, st))) return rc;
    fold_dq_kernel<<<8, 64, 0, st>>>(k.p_pool, B * k.chunksE, EP_SIZE, k.dQ);
    LAUNCH_CHECK();
    // (the inducing-point gradient was set by the encoder-cell reduction above; this adds the query-projection branch)
    if (!g->inducing_points || !g->enc_cross.attn_q || !g->enc_cross.ln1q_w || !g->enc_cross.ln1q_b)
      return fail(SCLDM_ERR_SHAPE, "the encoder query-branch gradient pointers are required");
    enc_q_bwd_kernel<<<1, 64, 0, st>>>(w->inducing_points, w->enc_cross.ln1q_w, w->enc_cross.ln1q_b, w->enc_cross.attn_q, eps, k.dQ,
                                       G_(g->enc_cross.attn_q), G_(g->enc_cross.ln1q_w), G_(g->enc_cross.ln1q_b), G_(g->inducing_points));
    LAUNCH_CHECK();
  }
  return SCLDM_OK;   // (join_side: `st` waits for the second stream)
}

extern "C" int scldm_nb_loglik(const float* x, const float* mu, const float* theta, float eps, float* out, size_t n, void* stream_) {
  if (!x || !mu || !theta || !out) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (n == 0) return SCLDM_OK;
  nb_loglik_kernel<<<(int)std::min<size_t>((n + 255) / 256, 256 * 32), 256, 0, (hipStream_t)stream_>>>(x, mu, theta, eps, out, n);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
extern "C" int scldm_nb_loglik_bwd(const float* x, const float* mu, const float* theta, const float* gout, float eps, float* dmu,
                                   float* dtheta, size_t n, void* stream_) {
  if (!x || !mu || !theta || !gout) return fail(SCLDM_ERR_SHAPE, "null argument");
  if (n == 0) return SCLDM_OK;
  nb_loglik_bwd_kernel<<<(int)std::min<size_t>((n + 255) / 256, 256 * 32), 256, 0, (hipStream_t)stream_>>>(x, mu, theta, gout, eps, dmu, dtheta, n);
  LAUNCH_CHECK();
  return SCLDM_OK;
}
