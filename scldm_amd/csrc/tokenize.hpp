// Encoder input path: tokenize_cells(sample_genes="expressed"),
// reference src/scldm/datamodule.py:660-731.  Per cell: the genes with counts > 0 are compacted, in gene order, to
// the front of a genes_seq_len window; the tail is padded with the mask token / zero counts; library_size = sum of
// counts.  HBM-bound integer/byte work: one workgroup per cell streams the row once (4 B per gene in, 12 B per
// window slot out), positions come from wave ballots + popcounts (no atomics, order preserving, bit exact).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scldm {

constexpr int kTokPer = 4;   // consecutive genes per thread per trip (1024 genes per workgroup trip, one barrier)

__global__ __launch_bounds__(256) void tokenize_expressed_kernel(const float* __restrict__ counts, const int64_t* __restrict__ gene_idx,
                                                                 long gene_row_stride, int G, int S, int64_t mask_idx,
                                                                 int64_t* __restrict__ genes_out, float* __restrict__ counts_out,
                                                                 int32_t* __restrict__ num_expressed, float* __restrict__ library_size) {
  __shared__ int wave_tot[2][4];
  __shared__ float lib_part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row = blockIdx.x;
  const float* __restrict__ c = counts + row * (long)G;
  const int64_t* __restrict__ gi = gene_idx + row * gene_row_stride;
  int64_t* __restrict__ go = genes_out + row * (long)S;
  float* __restrict__ co = counts_out + row * (long)S;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const bool pair_ok = ((row * (long)G) & 1) == 0;   // row start is 8-byte aligned: two-gene loads
  int base = 0, buf = 0;
  float lib = 0.f;
  for (int j0 = 0; j0 < G; j0 += 256 * kTokPer, buf ^= 1) {
    const int j = j0 + tid * kTokPer;
    float v[kTokPer] = {0.f, 0.f, 0.f, 0.f};
    if (j + kTokPer <= G && pair_ok) {
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      const f32x2 a = *reinterpret_cast<const f32x2*>(c + j), b = *reinterpret_cast<const f32x2*>(c + j + 2);
      v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
    } else {
#pragma unroll
      for (int e = 0; e < kTokPer; ++e)
        if (j + e < G) v[e] = c[j + e];
    }
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < kTokPer; ++e) {
      lib += v[e];
      cnt += v[e] > 0.f ? 1 : 0;
    }
    // exclusive prefix of cnt (0..4) over the wave from three ballots, one per bit of cnt
    const unsigned long long b0 = __ballot(cnt & 1), b1 = __ballot(cnt & 2), b2 = __ballot(cnt & 4);
    const int pre = __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt);
    if (lane == 0) wave_tot[buf][wave] = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
    __syncthreads();   // one barrier per trip: the totals live in the buffer the next trip does not touch
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int t = wave_tot[buf][w];
      before += w < wave ? t : 0;
      tot += t;
    }
    int p = base + before + pre;
#pragma unroll
    for (int e = 0; e < kTokPer; ++e)
      if (v[e] > 0.f) {
        if (p < S) {
          go[p] = gi[j + e];
          co[p] = v[e];
        }
        ++p;
      }
    base += tot;
  }
  for (int p = min(base, S) + tid; p < S; p += 256) {
    go[p] = mask_idx;
    co[p] = 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lib += __shfl_xor(lib, o);
  if (lane == 0) lib_part[wave] = lib;
  __syncthreads();
  if (tid == 0) {
    num_expressed[row] = base;
    library_size[row] = (lib_part[0] + lib_part[1]) + (lib_part[2] + lib_part[3]);
  }
}



// ---------------------------------------------------------------------------------------------------------------
// Output assembly: dense generated counts (N,G) -> CSR, replacing
// scipy.sparse.csr_matrix(dense.cpu().numpy()) per batch (reference src/scldm/_utils.py:192-197 after models.py:742).
// Same streaming compaction as the tokenizer; entries != 0 are kept in column order (what scipy keeps).
// Pass 1 counts per row; the caller turns the counts into indptr (exclusive scan over N rows); pass 2 fills.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csr_count_kernel(const float* __restrict__ dense, int G, int32_t* __restrict__ row_nnz) {
  __shared__ int part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ c = dense + blockIdx.x * (long)G;
  int cnt = 0;
  for (int j = tid; j < G; j += 256) cnt += c[j] != 0.f ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) part[wave] = cnt;
  __syncthreads();
  if (tid == 0) row_nnz[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const float* __restrict__ dense, int G, const int64_t* __restrict__ indptr,
                                                       int32_t* __restrict__ indices, float* __restrict__ data) {
  __shared__ int wave_tot[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row = blockIdx.x;
  const float* __restrict__ c = dense + row * (long)G;
  int32_t* __restrict__ io = indices + indptr[row];
  float* __restrict__ vo = data + indptr[row];
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int base = 0, buf = 0;
  for (int j0 = 0; j0 < G; j0 += 256 * kTokPer, buf ^= 1) {
    const int j = j0 + tid * kTokPer;
    float v[kTokPer];
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < kTokPer; ++e) {
      v[e] = j + e < G ? c[j + e] : 0.f;
      cnt += v[e] != 0.f ? 1 : 0;
    }
    const unsigned long long b0 = __ballot(cnt & 1), b1 = __ballot(cnt & 2), b2 = __ballot(cnt & 4);
    const int pre = __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt);
    if (lane == 0) wave_tot[buf][wave] = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int t = wave_tot[buf][w];
      before += w < wave ? t : 0;
      tot += t;
    }
    int p = base + before + pre;
#pragma unroll
    for (int e = 0; e < kTokPer; ++e)
      if (v[e] != 0.f) {
        io[p] = j + e;
        vo[p] = v[e];
        ++p;
      }
    base += tot;
  }
}

}  // namespace scldm
