// Small kernels around the fused block: weight packing, conditioning (timestep MLP + class embeddings
// + all-layer adaLN projection), input projection, final layer, CFG blend and ODE state update.
#pragma once
#include "common.hpp"
#include "dit_forward.hpp"
#include "bwd_layout.hpp"

namespace scldm {

// ------------------------------------------------------------------------------------------------
// Weight packing: PyTorch (out, in) row-major fp32 -> one contiguous MFMA-fragment stream per wave
// (layout documented at WStream in dit_forward.hpp).  Element (unit gu, tile ft, lane l, j) lives at
// ((gu*FT + ft)*64 + l)*8 + j; its k index inside the unit's k-step is (l>>5)*8 + j; its row is l&31.
//   units 0-47 : c_attn rows p*256 + w*64 + ft*32 + r          (p = q,k,v)       attn.c_attn.weight (768,256)
//   units 48-63: c_proj rows w*64 + ft*32 + r                                      attn.c_proj.weight (256,256)
//   chunk c, units 0-15: tile rows 0-15 = w1[hid], rows 16-31 = w2[hid], hid = c*128 + w*32 + ft*16 + (r&15)
//   chunk c, units 16-23: mlp.c_proj rows w*64 + ft*32 + r, k = hidden index c*128 + ks*16 + ...
// Hidden indices >= H are zero (exact padding of 684 -> 768).
// ------------------------------------------------------------------------------------------------
// One element of a layer's packed stream (index documented above); FT = 32-row tiles per wave.
// s1 / s2: factors applied to the w1 / w2 rows (OP::kW1Scale / kW2Scale of the stream's precision policy)
__device__ __forceinline__ float pack_layer_val(const float* __restrict__ Wqkv, const float* __restrict__ Wproj,
                                                const float* __restrict__ W1, const float* __restrict__ W2,
                                                const float* __restrict__ Wcp, long long idx, int H, int n_chunks, int half, int FT,
                                                float s1, float s2) {
  // FT = 32-row tiles per wave (2: four waves, 1: eight waves); a unit holds FT fragments of 512 elements
  const int UL = units_per_layer(n_chunks, half), unit_elems = 512 * FT;
  const int j = idx & 7, l = (idx >> 3) & 63, ft = (int)((idx >> 9) % FT);
  const int gu = (int)(idx / unit_elems), w = gu / UL, u = gu % UL;
  const int r = l & 31, k8 = (l >> 5) * 8 + j;
  const int frow = (w * FT + ft) * 32 + r;  // output feature row of this wave's tile
  float val;
  if (u < 32 && FT == 2) {
    // Q and K of ONE head per pass: units 0-15 = head 0, 16-31 = head 1; inside a unit fragment 0 is the head's Q tile and
    // fragment 1 its K tile of the same k-step (an ordinary two-tile gemm_pass whose "feature tiles" are Q_h and K_h)
    const int head = u >> 4, ks = u & 15;
    val = Wqkv[(size_t)(ft * 256 + (w * 2 + head) * 32 + r) * 256 + ks * 16 + k8];
  } else if (u < 48) {
    const int p = u >> 4, ks = u & 15;
    val = Wqkv[(size_t)(p * 256 + frow) * 256 + ks * 16 + k8];
  } else if (u < 64) {
    const int ks = u - 48;
    val = Wproj[(size_t)frow * 256 + ks * 16 + k8];
  } else if (u >= 64 + n_chunks * kUnitsPerChunk) {
    // trailing half chunk (FT=2): 64 hidden units, wave w owns 16 of them in ONE tile; then a K=64 c_proj pass
    const int vh = u - 64 - n_chunks * kUnitsPerChunk, hid0 = n_chunks * kHC;
    if (vh < 8) {
      const int ks = 2 * vh + ft, hid = hid0 + w * 16 + (r & 15);
      const float* src = (r < 16) ? W1 : W2;
      val = (hid < H) ? src[(size_t)hid * 256 + ks * 16 + k8] * ((r < 16) ? s1 : s2) : 0.f;
    } else {
      const int hid = hid0 + (vh - 8) * 16 + k8;
      val = (hid < H) ? Wcp[(size_t)frow * H + hid] : 0.f;
    }
  } else {
    const int v = u - 64, c = v / kUnitsPerChunk, vv = v % kUnitsPerChunk;
    if (vv < 16) {
      // FT=2: units 0-7 = tile 0, units 8-15 = tile 1, each unit = k-steps (2j, 2j+1) of that tile (gemm_pass_tile)
      // (SCLDM_W12_PAIR: unit vv = k-step vv of BOTH tiles, an ordinary two-tile gemm_pass)
      const int tile = (FT == 2 && !SCLDM_W12_PAIR) ? (vv >> 3) : ft;
      const int ks = (FT == 2 && !SCLDM_W12_PAIR) ? (2 * (vv & 7) + ft) : vv;
      const int hid = c * kHC + (w * FT + tile) * 16 + (r & 15);
      const float* src = (r < 16) ? W1 : W2;
      val = (hid < H) ? src[(size_t)hid * 256 + ks * 16 + k8] * ((r < 16) ? s1 : s2) : 0.f;
    } else {
      const int hid = c * kHC + (vv - 16) * 16 + k8;
      val = (hid < H) ? Wcp[(size_t)frow * H + hid] : 0.f;
    }
  }
  return val;
}

// ------------------------------------------------------------------------------------------------
// Weight stream of the fused BACKWARD layer (dit_backward.hpp; bf16, eight waves, one 32-row tile per wave: a unit is ONE
// fragment of 512 elements, element (gu, l, j) at (gu*64 + l)*8 + j, row l&31, k = ks*16 + (l>>5)*8 + j).  Wave w, units
//   chunk c (3 chunks of 256 hidden units, hid = c*256 + w*32 + r; hid >= H is exact zero padding):
//     0-15  c_proj^T   row r = hidden unit hid,        k = output feature            (d hid = c_proj^T dy2)
//     16-31 w1         row r = hidden unit hid,        k = input feature             (recompute a)
//     32-47 w2         row r = hidden unit hid,        k = input feature             (recompute b)
//     48-79 [w1^T|w2^T] row r = feature w*32 + r,      k < 256: w1 unit c*256 + k, else w2 unit c*256 + k - 256   (d h2)
//   240-255 c_proj(attn)^T  row = feature w*32 + r (head w's d), k = output feature  (d ao = c_proj^T dy1)
//   256-303 c_attn rows p*256 + w*32 + r, p = q, k, v (16 units each)                (recompute q, k, v of head w)
//   304-351 c_attn^T  row = feature w*32 + r,          k = c_attn output row (768)   (d h1 = c_attn^T d qkv)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pack_bwd_val(const float* __restrict__ Wqkv, const float* __restrict__ Wproj,
                                              const float* __restrict__ W1, const float* __restrict__ W2,
                                              const float* __restrict__ Wcp, long long idx, int H) {
  const int j = idx & 7, l = (idx >> 3) & 63;
  const int gu = (int)(idx >> 9), w = gu / kBwdUnitsLayer, u = gu % kBwdUnitsLayer;
  const int r = l & 31, k8 = (l >> 5) * 8 + j;
  const int frow = w * 32 + r;
  if (u < kBwdChunks * kBwdUnitsChunk) {
    const int c = u / kBwdUnitsChunk, v = u % kBwdUnitsChunk;
    const int hid = c * kBwdChunk + w * 32 + r;
    if (v < 16) return hid < H ? Wcp[(size_t)(v * 16 + k8) * H + hid] : 0.f;
    if (v < 32) return hid < H ? W1[(size_t)hid * 256 + (v - 16) * 16 + k8] : 0.f;
    if (v < 48) return hid < H ? W2[(size_t)hid * 256 + (v - 32) * 16 + k8] : 0.f;
    const int kk = (v - 48) * 16 + k8;                      // 0..511
    const int hk = c * kBwdChunk + (kk & 255);
    return hk < H ? (kk < 256 ? W1 : W2)[(size_t)hk * 256 + frow] : 0.f;
  }
  const int v = u - kBwdChunks * kBwdUnitsChunk;
  if (v < 16) return Wproj[(size_t)(v * 16 + k8) * 256 + frow];
  if (v < 64) {
    const int p = (v - 16) >> 4, ks = (v - 16) & 15;
    return Wqkv[(size_t)(p * 256 + frow) * 256 + ks * 16 + k8];
  }
  return Wqkv[(size_t)((v - 64) * 16 + k8) * 256 + frow];
}

// Store one lane fragment (packed indices idx8 * 8 .. + 7) in the stream's precision: 16-byte stores (round 5: the re-pack runs every
// training step; one thread per ELEMENT meant a 2-byte store and a full index decode per element: 117 us for the base DiT).
__device__ __forceinline__ void pack_store8(void* out, long long idx8, const float (&v)[8], int prec, int* fp16_stats = nullptr) {
  typedef __attribute__((ext_vector_type(4))) unsigned ps_u32x4;
  if (prec == 0) {
    float* o = reinterpret_cast<float*>(out) + idx8 * 8;
    *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else if (prec == 1) {
    union { __bf16 h[8]; ps_u32x4 u; } t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.h[i] = (__bf16)v[i];
    *reinterpret_cast<ps_u32x4*>(reinterpret_cast<__bf16*>(out) + idx8 * 8) = t.u;
  } else if (prec == 3) {
    union { _Float16 h[8]; ps_u32x4 u; } t;
    int over = 0, sub = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      t.h[i] = (_Float16)v[i];
      const float av = fabsf(v[i]);
      over += av > 65504.f;
      sub += (v[i] != 0.f) && av < 6.103515625e-05f;
      nz += v[i] != 0.f;
    }
    *reinterpret_cast<ps_u32x4*>(reinterpret_cast<_Float16*>(out) + idx8 * 8) = t.u;
    if (fp16_stats) {   // wave-aggregated counters: one atomic per wave and non-zero counter
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { over += __shfl_xor(over, o); sub += __shfl_xor(sub, o); nz += __shfl_xor(nz, o); }
      if ((threadIdx.x & 63) == 0) {
        if (over) atomicAdd(fp16_stats + 0, over);
        if (sub) atomicAdd(fp16_stats + 1, sub);
        if (nz) atomicAdd(fp16_stats + 2, nz);
      }
    }
  } else {
    union { __bf16 h[8]; ps_u32x4 u; } hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) OpBF16x3::split(v[i], hi.h[i], lo.h[i]);
    __bf16* o = reinterpret_cast<__bf16*>(out) + idx8 * 16;
    *reinterpret_cast<ps_u32x4*>(o) = hi.u;
    *reinterpret_cast<ps_u32x4*>(o + 8) = lo.u;
  }
}

// Store packed element `idx` (fragment-major: 8 consecutive indices = one lane's 8 k-values) in the stream's precision.
//   prec 0: fp32   1: bf16   2: split-bf16 (the 8 hi values then the 8 lo values of a lane: 32 bytes per lane-fragment)   3: fp16
// fp16_stats (prec 3): [0] += values beyond the fp16 range (stored as +-inf), [1] += non-zero values below the smallest normal
// (6.1e-5: stored with fewer than 10 mantissa bits), [2] += non-zero values packed
__device__ __forceinline__ void pack_store(void* out, long long idx, float val, int prec, int* fp16_stats = nullptr) {
  if (prec == 0) {
    reinterpret_cast<float*>(out)[idx] = val;
  } else if (prec == 1) {
    reinterpret_cast<__bf16*>(out)[idx] = (__bf16)val;
  } else if (prec == 3) {
    reinterpret_cast<_Float16*>(out)[idx] = (_Float16)val;
    if (fp16_stats && val != 0.f) {
      const float av = fabsf(val);
      // wave-aggregated counters: one atomic per wave and counter
      const unsigned long long over = __ballot(av > 65504.f), sub = __ballot(av < 6.103515625e-05f), nz = __ballot(true);
      const int lane = threadIdx.x & 63;
      if (over && lane == __ffsll((long long)over) - 1) atomicAdd(fp16_stats + 0, __popcll(over));
      if (sub && lane == __ffsll((long long)sub) - 1) atomicAdd(fp16_stats + 1, __popcll(sub));
      if (lane == __ffsll((long long)nz) - 1) atomicAdd(fp16_stats + 2, __popcll(nz));
    }
  } else {
    __bf16 hi, lo;
    OpBF16x3::split(val, hi, lo);
    __bf16* o = reinterpret_cast<__bf16*>(out) + (idx >> 3) * 16 + (idx & 7);
    o[0] = hi;
    o[8] = lo;
  }
}

// ------------------------------------------------------------------------------------------------
// All packing of one scldm_dit_load_weights call is ONE launch over a table of jobs (was ~80 launches), so that it can
// also be re-run conditionally, on device, by scldm_dit_refresh_weights: every block first reads the `dirty` word.
// ------------------------------------------------------------------------------------------------
enum PackKind : int { kPackCopy = 0, kPackTranspose = 1, kPackLayer = 2, kPackFinal = 3, kPackLayerBwd = 4, kPackAdaX3 = 5 };
struct PackJob {
  int kind;
  int first_block;      // first 256-thread block of this job
  long long n;          // elements (pack_job_threads(kind, n, p) threads: layer streams take a lane fragment of 8 per thread)
  const float* s[5];    // sources (parameter tensors)
  void* d;              // destination
  int p[6];             // kind-specific: copy -; transpose N,K,ldo,col0; layer H,n_chunks,half,FT,prec; final din,prec
  long long d_off;      // element offset added to the destination index (layer: start of the layer's stream)
};
// threads a job needs: the layer streams (forward and backward) handle one lane fragment = 8 consecutive elements per thread; a
// transpose whose extents are multiples of 32 runs as 32 x 32 tiles through LDS (four elements per thread)
__host__ __device__ inline bool pack_transpose_tiled(const int* p) { return p[0] % 32 == 0 && p[1] % 32 == 0; }
__host__ __device__ inline long long pack_job_threads(int kind, long long n, const int* p) {
  if (kind == kPackLayer || kind == kPackLayerBwd) return n / 8;
  if (kind == kPackTranspose && pack_transpose_tiled(p)) return n / 4;
  return n;
}
// prec_mask: bit p set = pack the streams of precision p (kPackLayer / kPackFinal jobs of other precisions are skipped - the
// training step re-packs only what it reads); kPackLayerBwd jobs run when bit 8 is set (bit 9: as fp16 instead of bf16); bit 10: ONLY
// the kPackLayerBwd jobs (the training step's second launch: the backward stream is packed beside the forward, not ahead of it).
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* __restrict__ jobs, int n_jobs, const int* __restrict__ dirty,
                                                        unsigned prec_mask, int* __restrict__ fp16_stats = nullptr) {
  if (dirty && *dirty == 0) return;
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackJob& j = jobs[lo];
  const long long idx = (long long)((int)blockIdx.x - j.first_block) * 256 + threadIdx.x;
  if ((prec_mask & 0x400u) && j.kind != kPackLayerBwd) return;
  if (j.kind == kPackTranspose && pack_transpose_tiled(j.p)) {   // out[k*ldo + col0 + n] = W[n*K + k], a 32 x 32 tile per block
    __shared__ float tile[32][33];
    const int N = j.p[0], K = j.p[1], ldo = j.p[2], col0 = j.p[3];
    const int b = (int)blockIdx.x - j.first_block, kt = K / 32, n0 = (b / kt) * 32, k0 = (b % kt) * 32;   // (whole block: no early exit before the barrier)
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[r0 + 8 * q][c] = j.s[0][(size_t)(n0 + r0 + 8 * q) * K + k0 + c];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) reinterpret_cast<float*>(j.d)[(size_t)(k0 + r0 + 8 * q) * ldo + col0 + n0 + c] = tile[c][r0 + 8 * q];
    (void)N;
    return;
  }
  if (idx >= pack_job_threads(j.kind, j.n, j.p)) return;
  switch (j.kind) {
    case kPackCopy:
      reinterpret_cast<float*>(j.d)[idx] = j.s[0][idx];
      break;
    case kPackTranspose: {  // out[k*ldo + col0 + n] = W[n*K + k]
      const int N = j.p[0], K = j.p[1], ldo = j.p[2], col0 = j.p[3];
      const int n = (int)(idx % N), k = (int)(idx / N);
      reinterpret_cast<float*>(j.d)[(size_t)k * ldo + col0 + n] = j.s[0][(size_t)n * K + k];
      break;
    }
    case kPackAdaX3: {  // adaLN weight (N, 256) -> split-bf16 B fragments of adaln_x3_kernel: [n tile][k step][lane][hi x 8 | lo x 8]
      const int n = (int)(idx >> 8) + j.p[0], k = (int)(idx & 255);
      const float v = j.s[0][idx];
      __bf16 hi, lo;
      OpBF16x3::split(v, hi, lo);
      const size_t base = ((((size_t)(n >> 5) * 16 + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) << 4) + (k & 7);
      reinterpret_cast<__bf16*>(j.d)[base] = hi;
      reinterpret_cast<__bf16*>(j.d)[base + 8] = lo;
      break;
    }
    case kPackLayerBwd: {   // bit 9: the step's operand type is fp16 (same 16-bit stream, re-packed every training step)
      if (!(prec_mask & 0x100u)) return;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = pack_bwd_val(j.s[0], j.s[1], j.s[2], j.s[3], j.s[4], idx * 8 + e, j.p[0]);
      pack_store8(j.d, j.d_off / 8 + idx, v, (prec_mask & 0x200u) ? 3 : 1);
      break;
    }
    case kPackLayer:
      if (!((prec_mask >> j.p[4]) & 1u)) return;
      {
        const float s1 = (j.p[4] == 1 || j.p[4] == 3) ? OpBF16::kW1Scale : 1.0f, s2 = (j.p[4] == 1 || j.p[4] == 3) ? OpBF16::kW2Scale : 1.0f;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = pack_layer_val(j.s[0], j.s[1], j.s[2], j.s[3], j.s[4], idx * 8 + e, j.p[0], j.p[1], j.p[2], j.p[3], s1, s2);
        pack_store8(j.d, j.d_off / 8 + idx, v, j.p[4], fp16_stats);
      }
      break;
    case kPackFinal: {  // final_layer.linear (din,256) -> 16 fragments of a 32-row tile (rows >= din zero): ((ks*64 + l)*8 + j)
      if (!((prec_mask >> j.p[1]) & 1u)) return;
      const int jj = idx & 7, l = (idx >> 3) & 63, ks = (int)(idx >> 9);
      const int row = l & 31, k = ks * 16 + (l >> 5) * 8 + jj;
      pack_store(j.d, idx, (row < j.p[0]) ? j.s[0][row * 256 + k] : 0.f, j.p[1], fp16_stats);
      break;
    }
  }
}

// Fingerprint of the parameter tensors (scldm_dit_refresh_weights): EVERY element of every tensor enters one 64-bit
// position-dependent, order-independent (wrapping) sum - tensor blockIdx.x is split over kFpSplit workgroups.  A partial
// `.data` write (a few rows of a class table, a masked update) therefore always moves it; the pass reads the parameters once
// (39 MB for the base DiT: ~10 us, against >= 300 us for the smallest forward).
// Round 4: 32 workgroups per tensor (at least 2 048 words each), 16-byte loads with two in flight, one atomic per workgroup - with 8
// workgroups of scalar loads a thread's loop was ~190 dependent HBM round trips on the largest tensors and the pass took 62-68 us,
// which every Python-level forward_with_cfg (the dopri5 sampler: 110 per solve) pays.
constexpr int kFpSplit = 32;
struct FpSrc {
  const uint32_t* p;
  long long n;
};
__global__ __launch_bounds__(256) void fingerprint_kernel(const FpSrc* __restrict__ src, unsigned long long* __restrict__ acc) {
  const FpSrc s = src[blockIdx.x];
  if (s.n <= 0) return;
  long long chunk = ((s.n + kFpSplit - 1) / kFpSplit + 255) / 256 * 256;
  if (chunk < 2048) chunk = 2048;
  const long long lo = (long long)blockIdx.y * chunk, hi = lo + chunk < s.n ? lo + chunk : s.n;
  if (lo >= s.n) return;
  unsigned long long hsum = 0;
  auto mix = [&](unsigned long long v, long long pos) {
    hsum += (v + 0x9E3779B97F4A7C15ull * (unsigned long long)(pos + 1 + blockIdx.x * 7919ll)) * 0xBF58476D1CE4E5B9ull ^ (v << 29);
  };
  if ((reinterpret_cast<size_t>(s.p) & 15) == 0) {       // (chunk boundaries are multiples of 256 words: only the base decides)
    const uint4* p4 = reinterpret_cast<const uint4*>(s.p);
    const long long q_hi = hi / 4;
    long long q = lo / 4 + threadIdx.x;
    for (; q + 256 < q_hi; q += 512) {
      const uint4 u = p4[q], w = p4[q + 256];
      mix(u.x, 4 * q); mix(u.y, 4 * q + 1); mix(u.z, 4 * q + 2); mix(u.w, 4 * q + 3);
      mix(w.x, 4 * (q + 256)); mix(w.y, 4 * (q + 256) + 1); mix(w.z, 4 * (q + 256) + 2); mix(w.w, 4 * (q + 256) + 3);
    }
    if (q < q_hi) {
      const uint4 u = p4[q];
      mix(u.x, 4 * q); mix(u.y, 4 * q + 1); mix(u.z, 4 * q + 2); mix(u.w, 4 * q + 3);
    }
    for (long long pos = 4 * q_hi + threadIdx.x; pos < hi; pos += 256) mix(s.p[pos], pos);
  } else {
    for (long long pos = lo + threadIdx.x; pos < hi; pos += 256) mix(s.p[pos], pos);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hsum += __shfl_xor(hsum, o);
  __shared__ unsigned long long wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = hsum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}
// state[0] = accumulator of the current pass, state[1] = fingerprint of the packed weights; dirty[0] = re-pack?, dirty[1] = force
__global__ void set_word_kernel(int* p, int v) { *p = v; }
__global__ void fingerprint_compare_kernel(unsigned long long* __restrict__ state, int* __restrict__ dirty, int* __restrict__ fp16_stats = nullptr) {
  dirty[0] = (state[0] != state[1]) || dirty[1];
  if (dirty[0] && fp16_stats) fp16_stats[0] = fp16_stats[1] = fp16_stats[2] = 0;   // the pack that follows recounts
  dirty[1] = 0;
  state[1] = state[0];
  state[0] = 0;
}

// ------------------------------------------------------------------------------------------------
// Conditioning.  One workgroup (256 threads = 256 features) per conditioning row.
//   c = t_mlp(sinusoid(t)) + sum_classes emb_c[label or null]    (layers.py:351-364, nnets.py:380-456)
// writes silu(c) (the input of every adaLN projection, layers.py:206,395).
// ------------------------------------------------------------------------------------------------
constexpr int kMaxClasses = 8;
struct CondArgs {
  const float* t;          // (rows) or scalar when t_stride == 0
  int t_stride;
  const float* w0t;        // (256,256) transposed t_embedder.mlp.0.weight  [k][n]
  const float* b0;
  const float* w2t;        // transposed t_embedder.mlp.2.weight
  const float* b2;
  const float* emb;        // concatenated class tables, (sum(vocab+1), 256)
  int n_classes;
  int emb_row0[kMaxClasses];   // first row of class c in `emb`
  int null_tok[kMaxClasses];   // vocab size of class c (= index of its null token)
  int tab_rows[kMaxClasses];   // rows of class c's table (vocab + 1, or vocab when the model has no null row)
  const int64_t* labels[kMaxClasses];  // (rows) or nullptr => null token for every row
  float* silu_c;           // (rows,256)
  int rows;
  int* label_err;          // sticky count of out-of-range labels (clamped instead of reading another class's table)
  const int* gate;         // device-decided plan (scldm_dit_forward_cfg, t_stride 2): run only if gate == nullptr or *gate == gate_want
  int gate_want;
  const int32_t* row_map;  // optional: the label of row u is labels[c][row_map[u]] (per-cell rows out of de-duplicated label rows)
};

// label -> table row; out-of-range labels are clamped and counted (the reference's nn.Embedding raises, nnets.py:420,453)
__device__ __forceinline__ int checked_label(const int64_t* lab, int i, int tab_rows, int* err, bool report) {
  long t = (long)lab[i];
  if (t < 0 || t >= tab_rows) {
    if (report) atomicAdd(err, 1);
    t = t < 0 ? 0 : tab_rows - 1;
  }
  return (int)t;
}

__global__ __launch_bounds__(256) void cond_embed_kernel(const CondArgs a) {
  __shared__ float te[256];
  __shared__ float h1[256];
  if (a.gate && *a.gate != a.gate_want) return;
  const int u = blockIdx.x, n = threadIdx.x;
  const float t = a.t[(size_t)u * a.t_stride];
  {
    const int k = n & 127;
    const float freq = expf(-9.210340371976184f * (float)k / 128.0f);  // exp(-ln(10000) k / half)
    const float arg = t * freq;
    te[n] = (n < 128) ? cosf(arg) : sinf(arg);  // [cos | sin], cos first
  }
  __syncthreads();
  float s = a.b0[n];
#pragma unroll 8
  for (int k = 0; k < 256; ++k) s += a.w0t[k * 256 + n] * te[k];
  h1[n] = silu_f(s);
  __syncthreads();
  float c = a.b2[n];
#pragma unroll 8
  for (int k = 0; k < 256; ++k) c += a.w2t[k * 256 + n] * h1[k];
  for (int ci = 0; ci < a.n_classes; ++ci) {
    int tok = a.null_tok[ci];
    if (a.labels[ci] != nullptr) tok = checked_label(a.labels[ci], a.row_map ? a.row_map[u] : u, a.tab_rows[ci], a.label_err, n == 0);
    c += a.emb[(size_t)(a.emb_row0[ci] + tok) * 256 + n];
  }
  a.silu_c[(size_t)u * 256 + n] = silu_f(c);
}

// Scalar-t fast path of the sampler (the ODE solver broadcasts ONE t, integrators.py:103-104): the timestep MLP is the
// same for every conditioning row, so it runs once per step in a single workgroup ...
__global__ __launch_bounds__(256) void t_embed_kernel(const float* __restrict__ t, const float* __restrict__ w0t,
                                                      const float* __restrict__ b0, const float* __restrict__ w2t,
                                                      const float* __restrict__ b2, float* __restrict__ temb) {
  __shared__ float te[256];
  __shared__ float h1[256];
  const int n = threadIdx.x;
  {
    const int k = n & 127;
    const float freq = expf(-9.210340371976184f * (float)k / 128.0f);
    const float arg = t[0] * freq;
    te[n] = (n < 128) ? cosf(arg) : sinf(arg);
  }
  __syncthreads();
  // (64 weight loads requested together, the additions in k order as before: with 8 in flight the two loops were 64 dependent L2 round
  // trips - 22 us per evaluation of a host-driven solver, profiles/r6_dopri5_before_kernel_stats.txt)
  float s = b0[n];
  for (int k0 = 0; k0 < 256; k0 += 64) {
    float wv[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) wv[j] = w0t[(k0 + j) * 256 + n];
#pragma unroll
    for (int j = 0; j < 64; ++j) s += wv[j] * te[k0 + j];
  }
  h1[n] = silu_f(s);
  __syncthreads();
  float c = b2[n];
  for (int k0 = 0; k0 < 256; k0 += 64) {
    float wv[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) wv[j] = w2t[(k0 + j) * 256 + n];
#pragma unroll
    for (int j = 0; j < 64; ++j) c += wv[j] * h1[k0 + j];
  }
  temb[n] = c;
}

// Whole-trajectory variant: evaluation e of a fixed-grid solve sees t = linspace(0,1,steps)[e] (Euler) or
// linspace[e/2 + e%2] (Heun); all of them are embedded by one launch before the loop starts.
__device__ __forceinline__ float linspace01_dev(int idx, int steps) {  // torch.linspace(0,1,steps), fp32, symmetric fill
  const float step = 1.0f / (float)(steps - 1);
  return (idx < steps / 2) ? step * (float)idx : 1.0f - step * (float)(steps - idx - 1);
}
__global__ __launch_bounds__(256) void t_embed_all_kernel(int steps, int heun, const float* __restrict__ w0t,
                                                          const float* __restrict__ b0, const float* __restrict__ w2t,
                                                          const float* __restrict__ b2, float* __restrict__ temb_all) {
  __shared__ float te[256];
  __shared__ float h1[256];
  const int n = threadIdx.x, e = blockIdx.x;
  const float t = heun ? linspace01_dev(e / 2 + (e & 1), steps) : linspace01_dev(e, steps);
  {
    const int k = n & 127;
    const float freq = expf(-9.210340371976184f * (float)k / 128.0f);
    const float arg = t * freq;
    te[n] = (n < 128) ? cosf(arg) : sinf(arg);
  }
  __syncthreads();
  // (64 weight loads requested together, additions in k order: as t_embed_kernel)
  float s = b0[n];
  for (int k0 = 0; k0 < 256; k0 += 64) {
    float wv[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) wv[j] = w0t[(k0 + j) * 256 + n];
#pragma unroll
    for (int j = 0; j < 64; ++j) s += wv[j] * te[k0 + j];
  }
  h1[n] = silu_f(s);
  __syncthreads();
  float c = b2[n];
  for (int k0 = 0; k0 < 256; k0 += 64) {
    float wv[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) wv[j] = w2t[(k0 + j) * 256 + n];
#pragma unroll
    for (int j = 0; j < 64; ++j) c += wv[j] * h1[k0 + j];
  }
  temb_all[(size_t)e * 256 + n] = c;
}

// ... and every row (row 0 = unconditional, row 1 + p*U + u = pass p / unique label row u) only adds its class embeddings.
struct CondRowsArgs {
  const float* temb;
  const float* emb;
  int n_classes, U, rows;
  int emb_row0[kMaxClasses];
  int null_tok[kMaxClasses];
  int tab_rows[kMaxClasses];
  const int64_t* labels[kMaxClasses];
  uint32_t mask[kMaxClasses];   // per pass: which classes keep their labels
  float* silu_c;
  int* label_err;
  const int* gate;              // device-decided plan: run only if gate == nullptr or *gate == 1 (t is uniform)
};
__global__ __launch_bounds__(256) void cond_rows_kernel(const CondRowsArgs a) {
  if (a.gate && *a.gate != 1) return;
  // blockIdx.y = evaluation of a whole-solve launch (its timestep embedding and its block of output rows); 0 for the per-evaluation launch
  const int r = blockIdx.x, n = threadIdx.x;
  float c = a.temb[(size_t)blockIdx.y * 256 + n];
  const int p = r > 0 ? (r - 1) / a.U : 0, u = r > 0 ? (r - 1) % a.U : 0;
  for (int ci = 0; ci < a.n_classes; ++ci) {
    int tok = a.null_tok[ci];
    if (r > 0 && a.labels[ci] != nullptr && ((a.mask[p] >> ci) & 1u)) tok = checked_label(a.labels[ci], u, a.tab_rows[ci], a.label_err, n == 0);
    c += a.emb[(size_t)(a.emb_row0[ci] + tok) * 256 + n];
  }
  a.silu_c[((size_t)blockIdx.y * a.rows + r) * 256 + n] = silu_f(c);
}

// mod[u][n] = bias[n] + sum_k wt[k][n] * silu_c[u][k] for ALL layers at once (n < mod_w).
// Workgroup = 64 columns x 4 k-quarters (split-K, combined through LDS), kAdaRU rows share each weight read;
// grid = (mod_w / 64, rows / kAdaRU): 400 workgroups for the sampler's 15 rows instead of 100.
constexpr int kAdaRU = 8;
// rows_dev (optional): the number of rows is decided on device (scldm_dit_forward_cfg with t_stride 2): row groups at or beyond
// it exit at once, so the grid may be sized for the larger plan.
__global__ __launch_bounds__(256) void adaln_all_kernel(const float* __restrict__ silu_c, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, float* __restrict__ mod,
                                                        int rows, int mod_w, const int* __restrict__ rows_dev = nullptr) {
  __shared__ float sc[kAdaRU][256];
  __shared__ float part[4][kAdaRU][64];
  const int col = threadIdx.x & 63, kq = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + col;
  const int u0 = blockIdx.y * kAdaRU;
  if (rows_dev) {
    rows = *rows_dev;
    if (u0 >= rows) return;
  }
  for (int i = threadIdx.x; i < kAdaRU * 256; i += 256) {
    const int u = u0 + (i >> 8);
    sc[i >> 8][i & 255] = (u < rows) ? silu_c[(size_t)u * 256 + (i & 255)] : 0.f;
  }
  __syncthreads();
  float acc[kAdaRU];
#pragma unroll
  for (int r = 0; r < kAdaRU; ++r) acc[r] = 0.f;
  if (n < mod_w) {
#pragma unroll 4
    for (int kk = 0; kk < 64; ++kk) {
      const int k = kq * 64 + kk;
      const float w = wt[(size_t)k * mod_w + n];
#pragma unroll
      for (int r = 0; r < kAdaRU; ++r) acc[r] += w * sc[r][k];
    }
  }
#pragma unroll
  for (int r = 0; r < kAdaRU; ++r) part[kq][r][col] = acc[r];
  __syncthreads();
  if (n < mod_w) {
    for (int r = kq; r < kAdaRU; r += 4)
      if (u0 + r < rows) mod[(size_t)(u0 + r) * mod_w + n] = bias[n] + part[0][r][col] + part[1][r][col] + part[2][r][col] + part[3][r][col];
  }
}

// The same product on the matrix pipe, exact fp32 (v_mfma_f32_32x32x2_f32 is bitwise an fmaf chain): one wave = a 32-row x 32-column
// tile over K = 256, lane (row r = lane & 31, half h = lane >> 5) holds the 128 k-values h * 128 .. + 127 of ITS conditioning row in
// registers (32 float4) and streams the matching weight rows (one coalesced 128-byte segment per half-wave and k).  Every output
// element is one fixed-order sum over k whatever the number of rows and wherever its row sits in a tile: a cell's adaLN vectors do
// not depend on the batch it is sampled in (what the sharded sampler's bit-equality self-check relies on).
// Why: joint / multi-class conditioning has hundreds of unique label tuples per batch (parse1m: ~750 rows at 1 024 cells); the VALU
// kernel above re-reads the 13 MB weight matrix once per 8 rows - 100 us per CFG evaluation, 13.6 % of the GPU time of that workload
// (profiles/r4a_parse1m_b1024_kernel_stats.txt).
__global__ __launch_bounds__(256, 2) void adaln_mfma_kernel(const float* __restrict__ silu_c, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ mod,
                                                            int rows, int mod_w, const int* __restrict__ rows_dev = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c32 = lane & 31, hh = lane >> 5;
  const int u0 = blockIdx.y * 32;
  if (rows_dev) rows = *rows_dev;
  if (u0 >= rows) return;
  const int n0 = (blockIdx.x * 4 + wave) * 32;
  if (n0 >= mod_w) return;   // (wave-uniform)
  const int ur = min(u0 + c32, rows - 1);
  const int n = min(n0 + c32, mod_w - 1);
  f32x4 a4[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) a4[j] = *reinterpret_cast<const f32x4*>(silu_c + (size_t)ur * 256 + hh * 128 + 4 * j);
  const float* wcol = wt + (size_t)(hh * 128) * mod_w + n;
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // weight values in chunks of 16 k, the next chunk requested before this chunk's 16 MFMAs (64 cycles each)
  float bcur[16], bnxt[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bcur[i] = wcol[(size_t)i * mod_w];
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    if (ch + 1 < 8) {
#pragma unroll
      for (int i = 0; i < 16; ++i) bnxt[i] = wcol[(size_t)((ch + 1) * 16 + i) * mod_w];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[ch * 4 + (i >> 2)][i & 3], bcur[i], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) bcur[i] = bnxt[i];
  }
  if (n0 + c32 < mod_w) {
    const float bn = bias[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int u = u0 + acc_row(r, hh);
      if (u < rows) mod[(size_t)u * mod_w + n] = acc[r] + bn;
    }
  }
}

// The same product with split-bf16 operands (the arithmetic class of the bf16x3 policy: hi * hi + hi * lo + lo * hi, fp32 accumulate,
// ~2^-17 relative) for the fast precision policies: 48 bf16 MFMAs of 32 cycles per 32 x 32 tile instead of 128 fp32 MFMAs of 64.
// The weights are pre-split at load time into B fragments (kPackAdaX3), a wave splits its 32 conditioning rows once and walks NTW
// column tiles with them.  Every output element is the same fixed sequence of MFMAs over k whatever NTW and the number of rows, so
// the host may pick NTW by row count (more waves for few rows) without changing a bit of the result.
template <int NTW>
__global__ __launch_bounds__(256, 2) void adaln_x3_kernel(const float* __restrict__ silu_c, const bf16x8x2* __restrict__ wfrag,
                                                          const float* __restrict__ bias, float* __restrict__ mod,
                                                          int rows, int mod_w, const int* __restrict__ rows_dev = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c32 = lane & 31, hh = lane >> 5;
  const int u0 = blockIdx.y * 32;
  if (rows_dev) rows = *rows_dev;
  if (u0 >= rows) return;
  const int n_tiles = mod_w >> 5;
  const int nt0 = (blockIdx.x * 4 + wave) * NTW;
  if (nt0 >= n_tiles) return;   // (wave-uniform)
  const float* arow = silu_c + (size_t)min(u0 + c32, rows - 1) * 256 + hh * 8;
  bf16x8x2 af[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(arow + s * 16), hi4 = *reinterpret_cast<const f32x4*>(arow + s * 16 + 4);
    const float t8[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    af[s] = OpBF16x3::pack8(t8);
  }
#pragma unroll 1
  for (int t = 0; t < NTW; ++t) {
    const int nt = nt0 + t;
    if (nt >= n_tiles) break;
    const bf16x8x2* wf = wfrag + (size_t)nt * 16 * 64 + lane;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = OpBF16x3::mma(af[s], wf[s * 64], acc);
    const int n = nt * 32 + c32;
    const float bn = bias[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int u = u0 + acc_row(r, hh);
      if (u < rows) mod[(size_t)u * mod_w + n] = acc[r] + bn;
    }
  }
}

// Many rows (joint conditioning: hundreds of unique label tuples per batch): the one-row-tile kernel above re-reads the 13 MB fragment
// stream once per 32 rows - 315 MB and 70 us per call at 750 rows (profiles/r4e_parse1m_b1024_kernel_stats.txt), bandwidth-bound far above
// its 10 us of MFMAs.  Two kernels instead: (1) the conditioning rows are split ONCE into A fragments [row tile][k step][lane][hi 8 | lo 8];
// (2) a workgroup owns a 128-row x 128-column block, wave (wr, wc) its 64 x 64 quadrant - per k step two A and two B fragments
// (32 bytes per lane each) feed four tiles x three MFMAs, the weights are read once per 128 rows.  Bit-identical to adaln_x3_kernel: an
// output element is the same sequence of MFMAs over k, and the split is the same function of the same fp32 value.
__global__ __launch_bounds__(256) void adaln_split_rows_kernel(const float* __restrict__ silu_c, bf16x8x2* __restrict__ afrag, int rows,
                                                               const int* __restrict__ rows_dev = nullptr) {
  if (rows_dev) rows = *rows_dev;
  const int idx = blockIdx.x * 256 + threadIdx.x;          // (row tile, k step, lane)
  const int lane = idx & 63, s = (idx >> 6) & 15, rt = idx >> 10;
  if (rt * 32 >= rows) return;
  const float* src = silu_c + (size_t)min(rt * 32 + (lane & 31), rows - 1) * 256 + s * 16 + (lane >> 5) * 8;
  const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src), hi4 = *reinterpret_cast<const f32x4*>(src + 4);
  const float t8[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  afrag[idx] = OpBF16x3::pack8(t8);
}
__global__ __launch_bounds__(256, 2) void adaln_x3_block_kernel(const bf16x8x2* __restrict__ afrag, const bf16x8x2* __restrict__ wfrag,
                                                                const float* __restrict__ bias, float* __restrict__ mod, int rows, int mod_w,
                                                                const int* __restrict__ rows_dev = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c32 = lane & 31, hh = lane >> 5;
  if (rows_dev) rows = *rows_dev;
  const int rt0 = blockIdx.y * 4 + (wave >> 1) * 2;       // first of this wave's two row tiles
  if (blockIdx.y * 128 >= rows) return;
  const int n_tiles = mod_w >> 5, n_rt = (rows + 31) >> 5;
  const int nt0 = blockIdx.x * 4 + (wave & 1) * 2;        // first of its two column tiles
  if (nt0 >= n_tiles || rt0 >= n_rt) return;              // (wave-uniform; no barrier in this kernel)
  const bool r1 = rt0 + 1 < n_rt, n1 = nt0 + 1 < n_tiles;
  const bf16x8x2* a0 = afrag + (size_t)rt0 * 16 * 64 + lane;
  const bf16x8x2* a1 = afrag + (size_t)(r1 ? rt0 + 1 : rt0) * 16 * 64 + lane;
  const bf16x8x2* b0 = wfrag + (size_t)nt0 * 16 * 64 + lane;
  const bf16x8x2* b1 = wfrag + (size_t)(n1 ? nt0 + 1 : nt0) * 16 * 64 + lane;
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 acc[2][2] = {{z, z}, {z, z}};
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    const bf16x8x2 fa0 = a0[s * 64], fa1 = a1[s * 64], fb0 = b0[s * 64], fb1 = b1[s * 64];
    acc[0][0] = OpBF16x3::mma(fa0, fb0, acc[0][0]);
    acc[0][1] = OpBF16x3::mma(fa0, fb1, acc[0][1]);
    acc[1][0] = OpBF16x3::mma(fa1, fb0, acc[1][0]);
    acc[1][1] = OpBF16x3::mma(fa1, fb1, acc[1][1]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if ((i == 1 && !r1) || (j == 1 && !n1)) continue;
      const int n = (nt0 + j) * 32 + c32;
      const float bn = bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int u = (rt0 + i) * 32 + acc_row(r, hh);
        if (u < rows) mod[(size_t)u * mod_w + n] = acc[i][j][r] + bn;
      }
    }
}

// plan[0] = 1 if every t[i] == t[0] (the scalar an ODE solver broadcasts, integrators.py:103-104) else 0; plan[1] = the number of
// conditioning rows of the plan that follows (rows_uniform or rows_dense).  One workgroup; NaNs compare unequal (dense plan).
__global__ __launch_bounds__(256) void uniform_t_kernel(const float* __restrict__ t, int n, int rows_uniform, int rows_dense,
                                                        int* __restrict__ plan) {
  __shared__ int diff;
  if (threadIdx.x == 0) diff = 0;
  __syncthreads();
  const float t0 = t[0];
  int d = 0;
  for (int i = threadIdx.x; i < n; i += 256) d |= !(t[i] == t0);
  if (d) diff = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    plan[0] = diff ? 0 : 1;
    plan[1] = diff ? rows_dense : rows_uniform;
  }
}

// ------------------------------------------------------------------------------------------------
// CFG blend (nnets.py:356-378) + explicit ODE state updates.
// v holds n_direct = 2B unconditional outputs followed by P conditional passes of B rows each.
//   dz[i]      = v[i]                                   i <  B
//   dz[B + i]  = v[B+i] + sum_p scale[p] * (v[2B + pB + i] - v[B+i])
// ------------------------------------------------------------------------------------------------
struct CfgArgs {
  const float* v;
  float* dz;       // (2B, e) may alias nothing else
  int B, e, P;     // e = elements per sample (16*din)
  float scale[kMaxClasses];
  float* z;        // optional (Euler step fused into the blend, round 4): z[idx] += hstep * dz[idx] instead of storing dz
  float hstep;
};
__global__ void cfg_blend_kernel(const CfgArgs a) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t half = (size_t)a.B * a.e;
  if (idx >= 2 * half) return;
  float r = a.v[idx];
  if (idx >= half) {
    const float u = r;
    for (int p = 0; p < a.P; ++p) r += a.scale[p] * (a.v[2 * half + (size_t)p * half + (idx - half)] - u);
  }
  if (a.z) a.z[idx] = a.z[idx] + a.hstep * r;   // (the expression of axpy_kernel: same bits as blend + axpy)
  else a.dz[idx] = r;
}

// out = z + h * k
__global__ void axpy_kernel(const float* __restrict__ z, const float* __restrict__ k, float* __restrict__ out, float h, size_t n) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) out[idx] = z[idx] + h * k[idx];
}
// z += hh * (k1 + k2)   (Heun corrector, hh = h/2)
__global__ void heun_kernel(float* __restrict__ z, const float* __restrict__ k1, const float* __restrict__ k2, float hh, size_t n) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) z[idx] = z[idx] + hh * (k1[idx] + k2[idx]);
}

__global__ void fill_row_index_kernel(int32_t* __restrict__ ri, const int32_t* __restrict__ cell_row, int n_direct, int B,
                                      int U, int P) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_direct + P * B) return;
  if (s < n_direct) { ri[s] = 0; return; }
  const int p = (s - n_direct) / B, i = (s - n_direct) % B;
  ri[s] = 1 + p * U + (cell_row ? cell_row[i] : i);
}

__global__ void iota_kernel(int32_t* __restrict__ ri, int n) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) ri[s] = s;
}

}  // namespace scldm
