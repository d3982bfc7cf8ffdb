// Is s_memtime a shader-clock counter, and what clock does the chip sustain under an MFMA-only / idle-ish load?
// Each wave counts s_memtime ticks across its loop; the host times the launch with events.  ticks / wall = tick rate.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
__global__ __launch_bounds__(256) void mfma_loop(unsigned long long* ticks, float* sink, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));   // (the builtin in this loop compiled to accvgpr shuffles)
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
typedef __attribute__((ext_vector_type(16))) float f16v;
__global__ __launch_bounds__(256) void mfma32_loop(unsigned long long* ticks, float* sink, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
  f16v acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(256) void sleep_loop(unsigned long long* ticks, int iters) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(8);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  unsigned long long* d; float* sink;
  (void)hipMalloc(&d, 2048 * 4 * 8); (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  static unsigned long long h[2048 * 4];
  const int blocks_list[3] = {256, 512, 1024};
  for (int mode = 0; mode < 3; ++mode)
    for (int bi = 0; bi < 3; ++bi) {
      const int blocks = blocks_list[bi];
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, sink, 100000);
        else if (mode == 1) hipLaunchKernelGGL(mfma32_loop, dim3(blocks), dim3(256), 0, 0, d, sink, 100000);
        else hipLaunchKernelGGL(sleep_loop, dim3(blocks), dim3(256), 0, 0, d, 40000);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
      }
      (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
      double mean = 0, mn = 1e30, mx = 0;
      for (int i = 0; i < blocks * 4; ++i) { mean += h[i]; if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; }
      mean /= blocks * 4;
      const double flop_per = mode == 0 ? 8 * 16384.0 : 4 * 32768.0;
      printf("%s blocks %4d (waves/SIMD %.1f): wall %.3f ms, ticks/wave mean %.0f min %.0f max %.0f -> %.0f MHz", mode == 0 ? "mfma16x16x32" : mode == 1 ? "mfma32x32x16" : "sleep", blocks, blocks / 256.0, ms, mean, mn, mx, mean / ms / 1e3);
      if (mode < 2) printf(", %.0f TFLOP/s", blocks * 4.0 * 100000 * flop_per / (ms * 1e-3) / 1e12);
      printf("\n");
    }
  return 0;
}
