// Micro-benchmark (round 3): can v_mfma_f32_16x16x32_f16 and v_exp_f32 / VALU work of the SAME or of ANOTHER wave of a SIMD overlap?
// One loop iteration = the flash-attention kernel's per-key-block instruction mix for one wave: 44 MFMAs (24 score + 20 PV), 32 v_exp_f32,
// 16 v_cvt_pk, ~24 other VALU.  Variants: MFMA only, VALU only, both in program order (MFMA block then VALU block), both interleaved
// (1 MFMA : 2 VALU); at 1 and 2 waves per SIMD (block 256 / 512 threads, one workgroup per CU).  Reports cycles per iteration and wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ void k(int iters, unsigned long long* out, float* sink) {
  f32x4 acc[12];
  float v[32];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.f + i * 0.01f); }
  for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 32; ++i) v[i] = -0.01f * (threadIdx.x & 15) - i * 0.001f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {           // 44 MFMAs on 12 accumulators
#pragma unroll
      for (int m = 0; m < 44; ++m) acc[m % 12] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % 12], 0, 0, 0);
    }
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 || MODE == 2) {           // 32 exp2 + converts + a few adds (dependent on the previous iteration only)
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.5f;
#pragma unroll
      for (int i = 0; i < 32; i += 2) { _Float16 h0 = (_Float16)v[i], h1 = (_Float16)v[i + 1]; v[i] += (float)h0 * 1e-3f; v[i + 1] += (float)h1 * 1e-3f; }
    }
    if (MODE == 3) {                        // the same work, interleaved by the scheduler: 1 MFMA : 2 VALU
#pragma unroll
      for (int m = 0; m < 44; ++m) acc[m % 12] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % 12], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.5f;
#pragma unroll
      for (int i = 0; i < 32; i += 2) { _Float16 h0 = (_Float16)v[i], h1 = (_Float16)v[i + 1]; v[i] += (float)h0 * 1e-3f; v[i + 1] += (float)h1 * 1e-3f; }
#pragma unroll
      for (int m = 0; m < 44; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 12; ++i) r += acc[i][0];
  for (int i = 0; i < 32; ++i) r += v[i];
  if (r == 123.456f) sink[threadIdx.x] = r;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int threads, unsigned long long* out, float* sink) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, 10, out, sink);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, iters, out, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h;
  (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("%-34s waves/SIMD %d: %8.1f us  %7.1f cycles per iteration and wave-slot (counter), %7.1f ns per iteration\n", name, threads / 256, ms * 1e3,
         (double)h / iters, ms * 1e6 / iters);
}

int main() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&sink, 4096);
  for (int threads : {256, 512}) {
    run<0>("44 MFMA", threads, out, sink);
    run<1>("32 exp2 + 16 cvt + 64 VALU", threads, out, sink);
    run<2>("MFMA block, then VALU block", threads, out, sink);
    run<3>("interleaved 1 MFMA : 3 VALU", threads, out, sink);
  }
  return 0;
}
