// Micro-benchmark (round 3): MFMA rate of a wave tile of NA A-fragments x NB B-fragments whose operands come from LDS
// (ds_read_b128, software-pipelined one step deep), with and without a loader wave streaming LDS-DMA beside it.
// Answers: how many MFMAs must one fragment read feed before the matrix pipe, not the LDS, is the limit -- for
// v_mfma_f32_32x32x16_f16 (32 cycles) and v_mfma_f32_16x16x32_f16 (16 cycles) -- at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BIG> struct Acc;
template <> struct Acc<1> { typedef f32x16 t; static __device__ __forceinline__ t mfma(f16x8 a, f16x8 b, t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); } };
template <> struct Acc<0> { typedef f32x4 t; static __device__ __forceinline__ t mfma(f16x8 a, f16x8 b, t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };

// NW MFMA waves (+ LW loader waves).  AREG: A fragments stay in registers (no A reads).
template <int BIG, int NA, int NB, int NW, int LW, int AREG, int RD>
__global__ __launch_bounds__((NW + LW) * 64) void k(const char* src, int steps, unsigned long long* out, float* sink, int data) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Acc<BIG>::t acc_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) {
    unsigned v = ((i * 2654435761u) >> 3 & 0x3fff3fffu) | 0x20002000u;   // pseudo-random finite f16 pairs
    const unsigned h = i * 0x9E3779B1u;
    if (data == 1 && i < 32768 / 4) v &= ((h & 0x10000u) ? 0xffffu : 0u) | ((h & 0x20000u) ? 0xffff0000u : 0u);   // B operands: ~half zeros (post-ReLU)
    if (data == 2) v = 0u;
    ((unsigned*)smem)[i] = v;
  }
  __syncthreads();
  if (wave >= NW) {   // loader: stream LDS-DMA into the upper 64 KiB for the whole run
    const char* base = src + (long long)blockIdx.x * 65536;
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        __builtin_amdgcn_global_load_lds((gptr_t)(base + ((it * 4 + u) & 63) * 1024 + lane * 16), (lptr_t)(smem + 65536 + ((it * 4 + u) & 63) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  acc_t acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < (BIG ? 16 : 4); ++j) acc[a][b][j] = 0.f;
  const char* lb = smem + lane * 16;
  f16x8 fa[2][NA], fb[2][NB];
  auto load = [&](int s, int set) {
    if (RD == 2 && s > 0) return;
    if (RD == 3 && s > 1) return;                      // both register sets loaded once, then no reads: operands still alternate
    const int o = (s & 7) * 1024 + wave * 64;          // addresses vary with the step (no hoisting), stay conflict-free
    if (RD == 1 || RD == 4) {                          // inline-asm reads: the compiler does not see them, waits are explicit
      const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(lb + o);
      if (!AREG || s == 0)
#pragma unroll
        for (int a = 0; a < NA; ++a) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[AREG ? 0 : set][a]) : "v"(la), "n"(32768 + a * 2048) : "memory");
#pragma unroll
      for (int b = 0; b < NB; ++b) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][b]) : "v"(la), "n"(b * 4096 % 32768) : "memory");
      return;
    }
    if (!AREG || s == 0)
#pragma unroll
      for (int a = 0; a < NA; ++a) fa[AREG ? 0 : set][a] = *(const f16x8*)(lb + 32768 + o + a * 2048);
#pragma unroll
    for (int b = 0; b < NB; ++b) fb[set][b] = *(const f16x8*)(lb + o + b * 4096 % 32768);
  };
  unsigned long long t0 = __builtin_readcyclecounter();
  load(0, 0);
  for (int s = 0; s < steps; s += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      load(s + h + 1, (h + 1) & 1);
      if (RD == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (RD == 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((AREG ? 0 : NA) + NB) : "memory");   // step s+h landed; s+h+1 in flight
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[a][b] = Acc<BIG>::mfma(fa[AREG ? 0 : h][a], fb[h][b], acc[a][b]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < (BIG ? 16 : 4); ++j) r += acc[a][b][j];
  if (r == 123.456f) sink[threadIdx.x] = r;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

static int g_data = 0, g_reps = 2, g_steps = 4096;
template <int BIG, int NA, int NB, int NW, int LW, int AREG, int RD = 0>
static void run(const char* src, unsigned long long* out, float* sink, int grid = 256) {
  const int steps = g_steps;
  unsigned long long h[256];
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto kern = k<BIG, NA, NB, NW, LW, AREG, RD>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < g_reps; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3((NW + LW) * 64), 131072, 0, src, steps, out, sink, g_data);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  const double flop = (double)steps * NA * NB * NW * (BIG ? 32768.0 : 16384.0) * grid;
  const double mfma_cycles = (double)steps * NA * NB * (BIG ? 32 : 16) * (NW / 4.0);   // per SIMD
  printf("data %s grid %3d %s NA %d NB %d waves %d%s%s: reads/MFMA %.2f  %7.1f us  %6.0f TFLOP/s  MFMA-busy %.2f (wg0: %.2f)\n", g_data == 0 ? "random" : (g_data == 1 ? "relu-like" : "zeros"), grid, BIG ? "32x32x16" : "16x16x32", NA, NB, NW,
         LW ? "+loader" : "", AREG ? " A-in-regs" : (RD == 1 ? " asm-waits" : (RD == 2 ? " NO-READS" : (RD == 3 ? " NO-READS/2 sets" : (RD == 4 ? " asm wait0" : "")))), (double)((AREG ? 0 : NA) + NB) / (NA * NB), ms * 1e3, flop / (ms * 1e-3) / 1e12,
         flop / (ms * 1e-3) / (2.5e15 * grid / 256.0), mfma_cycles / (double)h[0]);
}

int main(int argc, char** argv) {
  // usage: mfma_lds [data: 0 random | 1 relu-like | 2 zeros] [reps] [sustain: 1 = only the generic-kernel tile shape, long]
  g_data = argc > 1 ? atoi(argv[1]) : 0;
  g_reps = argc > 2 ? atoi(argv[2]) : 2;
  const int sustain = argc > 3 ? atoi(argv[3]) : 0;
  char* src; unsigned long long* out; float* sink;
  (void)hipMalloc(&src, 256 * 65536); (void)hipMemset(src, 0, 256 * 65536); (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&sink, 4096);
  if (sustain) {
    run<0, 2, 4, 8, 0, 0, 1>(src, out, sink, 256);      // the generic conv kernel's wave tile, operands from LDS, counted waits
    return 0;
  }
  for (int grid : {256, 64, 8}) {
    run<1, 2, 4, 8, 0, 0, 2>(src, out, sink, grid);
    run<1, 2, 4, 8, 0, 0, 3>(src, out, sink, grid);
    run<1, 2, 4, 8, 0, 0, 1>(src, out, sink, grid);
    run<1, 4, 4, 4, 0, 0, 1>(src, out, sink, grid);
    run<1, 1, 8, 8, 0, 0, 1>(src, out, sink, grid);
    run<0, 2, 4, 8, 0, 0, 3>(src, out, sink, grid);
    run<0, 2, 4, 8, 0, 0, 1>(src, out, sink, grid);
    run<0, 4, 4, 8, 0, 0, 1>(src, out, sink, grid);
  }
  return 0;
}
