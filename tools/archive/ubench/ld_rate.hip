// Micro-benchmark: per-CU fill rate, LDS-DMA vs global_load->VGPR->ds_write, vs number of active CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, long long wg_stride, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const char* base = src + blockIdx.x * wg_stride;
  unsigned long long t0 = __builtin_readcyclecounter();
  uint4 sink = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      for (int j = wave; j < 64; j += nw)
        __builtin_amdgcn_global_load_lds((gptr_t)(base + j * 1024 + lane * 16), (lptr_t)(smem + j * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { int j = wave + u * nw; if (j < 64) v[u] = *(const uint4*)(base + j * 1024 + lane * 16); }
#pragma unroll
      for (int u = 0; u < 8; ++u) { int j = wave + u * nw; if (j < 64) { if (MODE == 1) *(uint4*)(smem + j * 1024 + lane * 16) = v[u]; else { sink.x ^= v[u].x; sink.y ^= v[u].y; sink.z ^= v[u].z; sink.w ^= v[u].w; } } }
    }
    __syncthreads();
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (MODE == 2) *(uint4*)(smem + threadIdx.x * 16) = sink;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (MODE == 2 ? smem[5] & 0 : 0);
}
int main() {
  const int iters = 256;
  char* src; unsigned long long* out;
  (void)hipMalloc(&src, 256 * 65536); (void)hipMemset(src, 1, 256 * 65536); (void)hipMalloc(&out, 256 * 8);
  unsigned long long h[256];
  for (int mode = 0; mode < 3; ++mode)
    for (int grid : {256, 64, 8})
      for (int waves : {8}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
          (void)hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(waves * 64), 65536, 0, src, 65536, iters, out);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(waves * 64), 65536, 0, src, 65536, iters, out);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(waves * 64), 65536, 0, src, 65536, iters, out);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        double bytes = 65536.0 * iters;
        printf("%s grid %3d waves %d: %.1f us, %.1f GB/s per WG, %.1f B/clk/CU\n", mode == 0 ? "lds-dma      " : mode == 1 ? "load+ds_write" : "load only    ", grid, waves,
               ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (double)h[0]);
      }
  return 0;
}
