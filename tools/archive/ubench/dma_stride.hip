// Micro-benchmark (round 3): LDS-DMA fill rate when a lane's 16 bytes are STRIDED in memory -- the access pattern of a
// plane-major LDS halo image filled from a channels-last tensor (one instruction = one 8-channel plane of 64 voxels, voxel stride
// = channels x 2 bytes).  Every workgroup re-reads its own L2-resident 64 KiB; useful bytes are the same for every stride.
//   order 0: the planes of a voxel group back to back (what the conv kernels do); order 1: plane-outer.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// ORDER 2: what one 16-channel STAGE of the generic kernel does today: 2 planes back to back, the other chunks in later passes
// ORDER 3: proposed: a lane pair fetches the 32 contiguous bytes of a voxel's 16-channel chunk (voxel-major LDS image)
template <int S, int ORDER>
__global__ __launch_bounds__(512) void k(const char* src, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (long long)blockIdx.x * 65536;
  constexpr int NP = S / 16, NJ = 65536 / (64 * S);     // planes per voxel, 64-voxel groups
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (ORDER < 2) {
      for (int id = wave; id < NP * NJ; id += 8) {
        const int j = ORDER == 0 ? id / NP : id % NJ, pl = ORDER == 0 ? id % NP : id / NJ;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (long long)(j * 64 + lane) * S + pl * 16), (lptr_t)(smem + id * 1024), 16, 0, 0);
      }
    } else {
      for (int ch = 0; ch < NP / 2; ++ch) {                 // one "stage" per 16-channel chunk
        for (int id = wave; id < 2 * NJ; id += 8) {
          if (ORDER == 2) {
            const int j = id >> 1, pl = id & 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(base + (long long)(j * 64 + lane) * S + ch * 32 + pl * 16), (lptr_t)(smem + ((ch * 2 * NJ + id) % 64) * 1024), 16, 0, 0);
          } else {
            __builtin_amdgcn_global_load_lds((gptr_t)(base + (long long)(id * 32 + (lane >> 1)) * S + ch * 32 + (lane & 1) * 16), (lptr_t)(smem + ((ch * 2 * NJ + id) % 64) * 1024), 16, 0, 0);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int S, int ORDER>
static void run(const char* src, unsigned long long* out) {
  const int iters = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<S, ORDER>), dim3(256), dim3(512), 65536, 0, src, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double b = 65536.0 * iters;
  printf("lane stride %4d B (%2d-channel voxels), %s: %8.1f us  %6.2f TB/s chip  %5.1f B/clk/CU at 2.4 GHz\n", S, S / 2, ORDER == 0 ? "planes back to back" : (ORDER == 1 ? "plane-outer" : (ORDER == 2 ? "16-ch stages, 2 planes x 16 B" : "16-ch stages, lane pairs x 32 B")),
         ms * 1e3, b * 256 / (ms * 1e-3) / 1e12, b / (ms * 1e-3) / 2.4e9);
}
int main() {
  char* src; unsigned long long* out;
  (void)hipMalloc(&src, 256 * 65536); (void)hipMemset(src, 1, 256 * 65536); (void)hipMalloc(&out, 256 * 8);
  run<16, 0>(src, out);
  run<32, 0>(src, out); run<32, 1>(src, out);
  run<64, 0>(src, out); run<64, 1>(src, out);
  run<128, 0>(src, out); run<128, 1>(src, out);
  run<256, 0>(src, out); run<256, 1>(src, out);
  run<512, 0>(src, out);
  run<64, 2>(src, out); run<64, 3>(src, out);
  run<128, 2>(src, out); run<128, 3>(src, out);
  run<256, 2>(src, out); run<256, 3>(src, out);
  run<512, 2>(src, out); run<512, 3>(src, out);
  return 0;
}
