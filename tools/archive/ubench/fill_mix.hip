// Micro-benchmark (round 3): what feeds a CU fastest from L2 -- LDS-DMA, plain global loads, both at once -- and does an L1 (TCP)
// hit or a chip-wide shared source (packed weights every CU reads) change the rate?  One workgroup of 8 waves per CU.
//   mode 0: LDS-DMA, every WG its own 64 KiB (L2-resident after the first pass)
//   mode 1: global_load_dwordx4 -> VGPR, same addresses
//   mode 2: waves 0-3 LDS-DMA, waves 4-7 global loads, disjoint halves of the 64 KiB (do the two paths add up?)
//   mode 3: LDS-DMA of an 8 KiB block that all 8 waves of the WG fetch (L1-resident if LDS-DMA allocates in the TCP)
//   mode 4: global loads of an 8 KiB block that all 8 waves of the WG fetch
//   mode 5: LDS-DMA, ALL workgroups read the SAME 64 KiB (weights)
//   mode 6: global loads, ALL workgroups read the SAME 64 KiB
//   mode 7: global_load_dwordx4 with 64 B contiguous per 4 lanes but rows 128 B apart (channels-last 32-channel voxel rows... strided)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = (MODE == 5 || MODE == 6) ? src : src + (long long)blockIdx.x * 65536;
  uint4 sink = make_uint4(0, 0, 0, 0);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 5) {
#pragma unroll
      for (int u = 0; u < 8; ++u) dma16(base + (wave + 8 * u) * 1024 + lane * 16, smem + (wave + 8 * u) * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1 || MODE == 6) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const uint4*)(base + (wave + 8 * u) * 1024 + lane * 16);
#pragma unroll
      for (int u = 0; u < 8; ++u) { sink.x ^= v[u].x; sink.y ^= v[u].y; sink.z ^= v[u].z; sink.w ^= v[u].w; }
    } else if (MODE == 2) {
      if (wave < 4) {
#pragma unroll
        for (int u = 0; u < 8; ++u) dma16(base + (wave + 4 * u) * 1024 + lane * 16, smem + (wave + 4 * u) * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const uint4*)(base + 32768 + (wave - 4 + 4 * u) * 1024 + lane * 16);
#pragma unroll
        for (int u = 0; u < 8; ++u) { sink.x ^= v[u].x; sink.y ^= v[u].y; sink.z ^= v[u].z; sink.w ^= v[u].w; }
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 8; ++u) dma16(base + u * 1024 + lane * 16, smem + (wave * 8 + u) * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 4) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const uint4*)(base + u * 1024 + lane * 16);
#pragma unroll
      for (int u = 0; u < 8; ++u) { sink.x ^= v[u].x; sink.y ^= v[u].y; sink.z ^= v[u].z; sink.w ^= v[u].w; }
    } else if (MODE == 7) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const uint4*)(base + ((wave + 8 * u) * 16 + (lane >> 2)) * 128 % 65536 + (lane & 3) * 16);
#pragma unroll
      for (int u = 0; u < 8; ++u) { sink.x ^= v[u].x; sink.y ^= v[u].y; sink.z ^= v[u].z; sink.w ^= v[u].w; }
    }
    __syncthreads();
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (sink.x == 0x12345678u && sink.y == 1 && sink.z == 2 && sink.w == 3) *(uint4*)(smem + threadIdx.x * 16) = sink;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const char* src, unsigned long long* out, double bytes_per_iter) {
  const int iters = 256;
  unsigned long long h[256];
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, src, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  double b = bytes_per_iter * iters;
  printf("%-64s %8.1f us  %6.1f B/clk/CU (wg0 cycles)  %6.2f TB/s chip\n", name, ms * 1e3, b / (double)h[0], b * 256 / (ms * 1e-3) / 1e12);
}

int main() {
  char* src; unsigned long long* out;
  (void)hipMalloc(&src, 256 * 65536); (void)hipMemset(src, 1, 256 * 65536); (void)hipMalloc(&out, 256 * 8);
  run<0>("0 LDS-DMA, own 64 KiB per WG (L2)", src, out, 65536);
  run<1>("1 global_load x4 -> VGPR, own 64 KiB per WG (L2)", src, out, 65536);
  run<2>("2 4 waves LDS-DMA + 4 waves global_load, own 64 KiB (L2)", src, out, 65536);
  run<3>("3 LDS-DMA, 8 waves fetch the same 8 KiB (L1?)", src, out, 65536);
  run<4>("4 global_load, 8 waves fetch the same 8 KiB (L1)", src, out, 65536);
  run<5>("5 LDS-DMA, all WGs read the same 64 KiB", src, out, 65536);
  run<6>("6 global_load, all WGs read the same 64 KiB", src, out, 65536);
  run<7>("7 global_load, 64 B pieces at 128 B stride", src, out, 65536);
  return 0;
}
