// Micro-benchmark: per-CU LDS-DMA fill rate from L2-resident vs HBM-streamed sources.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int LANES>
__global__ __launch_bounds__(512) void k(const char* src, long long wg_stride, int iters, int bytes_per_iter, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const char* base = src + blockIdx.x * wg_stride;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const char* s = base + (long long)it * bytes_per_iter * (wg_stride ? 1 : 0);
    const int ninst = bytes_per_iter / (LANES * 16);
    for (int j = wave; j < ninst; j += nw) {
      if (lane < LANES)
        __builtin_amdgcn_global_load_lds((gptr_t)(s + (long long)j * LANES * 16 + lane * 16), (lptr_t)(smem + (j * LANES * 16) % 65536), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main(int argc, char** argv) {
  const int iters = 64, bpi = 65536;
  char* src; unsigned long long* out; 
  size_t total = (size_t)256 * iters * bpi;   // 1 GiB for the streaming case
  hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&out, 256 * 8);
  unsigned long long h[256];
  for (int mode = 0; mode < 2; ++mode)          // 0: every WG re-reads its own 64 KB (L2-resident); 1: streams fresh data
    for (int waves : {1, 2, 4, 8})
      for (int lanes : {64, 34}) {
        long long stride = mode ? (long long)iters * bpi : 65536;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (lanes == 64) hipLaunchKernelGGL(k<64>, dim3(256), dim3(waves * 64), 65536, 0, src, mode ? stride : 65536, iters, bpi, out);
          else hipLaunchKernelGGL(k<34>, dim3(256), dim3(waves * 64), 65536, 0, src, mode ? stride : 65536, iters, 34 * 16 * 120, out);
          hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        double bytes = (lanes == 64 ? (double)bpi : 34.0 * 16 * 120) * iters;
        printf("mode %s waves %d lanes %d: %.1f us, %.1f GB/s per CU, %.2f TB/s chip, %.1f B/clk/CU (cycles %llu)\n", mode ? "stream" : "L2", waves, lanes,
               ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * 256 / (ms * 1e-3) / 1e12, bytes / (double)h[0], h[0]);
      }
  return 0;
}
