// What ONE cross-workgroup exchange of a 32 x 256 fp32 partial tile costs on MI355X (feature-split co-operating
// workgroups would need two per layer).  Pairs of workgroups (2 p, 2 p + 1) each publish 32 KB through L2 (global stores, release fence,
// a flag), spin on the partner's flag (bounded: a partner that never arrives ends the loop after ~1 ms instead of hanging the GPU),
// acquire, read the partner's 32 KB and add it to their own - `rounds` times.  Prints the mean time per exchange for a few grid sizes
// (pairs on adjacent workgroup ids land on different XCDs; `stride` 8 puts both on the same XCD).
//   hipcc --offload-arch=gfx950 -O3 tests/perf/xwg_exchange.hip -o tests/perf/xwg_exchange && tests/perf/xwg_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kTile = 32 * 256;   // floats
__global__ __launch_bounds__(256) void xchg_kernel(float* buf, unsigned* flags, int rounds, int stride, float* out, int* timeouts) {
  const int wg = blockIdx.x;
  const int pair = (wg / (2 * stride)) * stride + wg % stride, side = (wg / stride) & 1;   // partner = wg +- stride
  float* mine = buf + ((size_t)pair * 2 + side) * kTile;
  const float* theirs = buf + ((size_t)pair * 2 + (side ^ 1)) * kTile;
  unsigned* fm = flags + pair * 2 + side;
  unsigned* ft = flags + pair * 2 + (side ^ 1);
  float acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = (float)(threadIdx.x + i);
  for (int r = 1; r <= rounds; ++r) {
    for (int i = 0; i < 32; ++i) mine[i * 256 + threadIdx.x] = acc[i];   // publish the partial tile (coalesced rows)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(fm, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(ft, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r && ++spins < 200000) __builtin_amdgcn_s_sleep(1);
      if (spins >= 200000) atomicAdd(timeouts, 1);
    }
    __syncthreads();
    __threadfence();
    for (int i = 0; i < 32; ++i) acc[i] += __builtin_nontemporal_load(theirs + i * 256 + threadIdx.x);
    __syncthreads();   // both sides have read before the next round overwrites (flag r + 1 is only set after the next publish)
    // (the partner may still be reading `mine` of round r when this side publishes round r + 1: a second buffer would be needed in a
    // real kernel; for the latency measurement the race only changes the summed values)
  }
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[wg * 256 + threadIdx.x] = s;
}

int main() {
  const int rounds = 200;
  for (int stride : {1, 8}) {
    for (int wgs : {2, 64, 192, 256, 384}) {
      float *buf, *out; unsigned* flags; int* to;
      hipMalloc(&buf, (size_t)wgs * kTile * 4); hipMalloc(&out, (size_t)wgs * 256 * 4); hipMalloc(&flags, wgs * 4); hipMalloc(&to, 4);
      hipMemset(flags, 0, wgs * 4); hipMemset(to, 0, 4);
      if (wgs % (2 * stride)) { hipFree(buf); hipFree(out); hipFree(flags); hipFree(to); continue; }
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      xchg_kernel<<<wgs, 256>>>(buf, flags, 2, stride, out, to);   // warm-up
      hipMemset(flags, 0, wgs * 4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      xchg_kernel<<<wgs, 256>>>(buf, flags, rounds, stride, out, to);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      int h_to = 0; hipMemcpy(&h_to, to, 4, hipMemcpyDeviceToHost);
      printf("partner stride %d (%s), %3d workgroups: %.2f us per exchange (32 KB out + flag + spin + 32 KB in), %d timeouts\n", stride,
             stride == 8 ? "same XCD" : "neighbouring XCDs", wgs, 1e3 * ms / rounds, h_to);
      hipFree(buf); hipFree(out); hipFree(flags); hipFree(to);
    }
  }
  return 0;
}
