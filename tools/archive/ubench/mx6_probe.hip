// Micro-benchmark (round 4, groundwork for the next precision step): the block-scaled MFMA with fp6 (e2m3) operands --
//   part 1: operand layout (32 six-bit codes of a lane packed little-endian into 6 VGPRs) and per-lane E8M0 block scales,
//           checked against a host reference;
//   part 2: issue rate of fp6 x fp6, fp6 x fp8, fp4 x fp4 against fp8 x fp8 (v_mfma_scale_f32_16x16x128_f8f6f4), all CUs busy,
//           and of the conv kernel's mix 27 f16 + 14 block-scaled MFMAs with fp6 operands.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mx6_probe tools/ubench/mx6_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__global__ void k_layout(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* d) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 2, 2, 0, sa[l], 0, sb[l]);
  for (int j = 0; j < 4; ++j) d[(4 * (l >> 4) + j) * 16 + (l & 15)] = c[j];
}

// MODE 0: fp8 x fp8, 1: fp6 x fp6, 2: A fp6 x B fp8, 3: fp4 x fp4, 4: 27 f16 + 14 fp6, 5: 27 f16 + 14 fp8
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_rate(int iters, float* sink) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 fa[4], fb[4];
  i32x8 xa[2], xb[2];
  for (int i = 0; i < 4; ++i) {
    unsigned pa[4], pb[4];
    for (int j = 0; j < 4; ++j) {
      const unsigned h = (t * 16 + i * 4 + j) * 2654435761u;
      pa[j] = ((h >> 3) & 0x3fff3fffu) | 0x20002000u;
      pb[j] = (((h * 0x9E3779B1u) >> 3) & 0x3fff3fffu) | 0x20002000u;
    }
    __builtin_memcpy(&fa[i], pa, 16);
    __builtin_memcpy(&fb[i], pb, 16);
  }
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 8; ++j) {
      const unsigned h = (t * 16 + i * 8 + j) * 0x85EBCA6Bu;
      xa[i][j] = (int)(h & 0xbfbfbfbfu);
      xb[i][j] = (int)((h * 2654435761u) & 0xbfbfbfbfu);
    }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sc = 0x7f7f7f7f;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 4) {
#pragma unroll
      for (int s = 0; s < 27; ++s) acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[s & 3], fb[(s >> 2) & 3], acc[s & 3], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      if (MODE == 0 || MODE == 5) acc[s & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xa[s & 1], xb[(s >> 1) & 1], acc[s & 3], 0, 0, 0, sc, 0, sc);
      if (MODE == 1 || MODE == 4) acc[s & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xa[s & 1], xb[(s >> 1) & 1], acc[s & 3], 2, 2, 0, sc, 0, sc);
      if (MODE == 2) acc[s & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xa[s & 1], xb[(s >> 1) & 1], acc[s & 3], 2, 0, 0, sc, 0, sc);
      if (MODE == 3) acc[s & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xa[s & 1], xb[(s >> 1) & 1], acc[s & 3], 4, 4, 0, sc, 0, sc);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (r == 123.456f) sink[threadIdx.x] = r;
}

template <int MODE, int NW>
static void rate(float* sink, double seconds) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<MODE, NW>), dim3(256), dim3(NW * 64), 0, 0, iters, sink);
  (void)hipDeviceSynchronize();
  double total = 0; float ms = 0;
  while (total < seconds * 1e3) {
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k_rate<MODE, NW>), dim3(256), dim3(NW * 64), 0, 0, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    total += ms;
  }
  const double us = ms * 1e3 / 20;
  const double nmx = (double)iters * (NW / 4.0) * 14, nf = MODE >= 4 ? (double)iters * (NW / 4.0) * 27 : 0;   // per SIMD
  const char* names[] = {"fp8 x fp8", "fp6 x fp6", "fp6 x fp8", "fp4 x fp4", "27 f16 + 14 fp6", "27 f16 + 14 fp8"};
  // cycles per block-scaled MFMA at 2.4 GHz nominal, the f16 ones charged 16 cycles each
  printf("mode %d (%-16s) waves/CU %d: %8.1f us/launch  -> %5.1f cycles per K=128 MFMA at 2.4 GHz%s\n", MODE, names[MODE], NW, us,
         (us * 1e-6 * 2.4e9 - nf * 16) / nmx, nf > 0 ? " (f16 MFMAs charged 16)" : "");
  fflush(stdout);
}

static float dec_e2m3(int c) {
  const int s = c >> 5, e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? m / 8.f : (1.f + m / 8.f) * (float)(1 << (e - 1));
  return s ? -v : v;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 1.0;
  // ---- part 1: A[16][128], B[128][16] as e2m3 codes; lane l: row / column l & 15, K = 32 (l >> 4) .. + 31, code i at bit 6 i
  static int ca[16][128], cb[128][16], sa[16][4], sb[16][4];
  srand(1);
  for (int m = 0; m < 16; ++m) for (int k = 0; k < 128; ++k) { ca[m][k] = rand() & 63; cb[k][m] = rand() & 63; }
  for (int m = 0; m < 16; ++m) for (int q = 0; q < 4; ++q) { sa[m][q] = 120 + rand() % 12; sb[m][q] = 122 + rand() % 10; }
  i32x8 ha[64], hb[64]; int hsa[64], hsb[64];
  memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
  for (int l = 0; l < 64; ++l) {
    unsigned char ba[32] = {0}, bb[32] = {0};
    for (int i = 0; i < 32; ++i) {
      const int k = 32 * (l >> 4) + i, bit = 6 * i;
      const unsigned va = (unsigned)ca[l & 15][k] << (bit & 7), vb = (unsigned)cb[k][l & 15] << (bit & 7);
      ba[bit >> 3] |= va & 255; ba[(bit >> 3) + 1] |= va >> 8;
      bb[bit >> 3] |= vb & 255; bb[(bit >> 3) + 1] |= vb >> 8;
    }
    memcpy(&ha[l], ba, 24); memcpy(&hb[l], bb, 24);
    hsa[l] = sa[l & 15][l >> 4] | 0x55aa3300;       // the other bytes must not matter with opsel 0
    hsb[l] = sb[l & 15][l >> 4] | 0x11223300;
  }
  i32x8 *da, *db; int *dsa, *dsb; float* dd;
  (void)hipMalloc(&da, sizeof ha); (void)hipMalloc(&db, sizeof hb); (void)hipMalloc(&dsa, sizeof hsa); (void)hipMalloc(&dsb, sizeof hsb);
  (void)hipMalloc(&dd, 256 * 4);
  (void)hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  (void)hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  float hd[256];
  (void)hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double r = 0;
      for (int k = 0; k < 128; ++k)
        r += (double)dec_e2m3(ca[m][k]) * ldexp(1.0, sa[m][k >> 5] - 127) * dec_e2m3(cb[k][n]) * ldexp(1.0, sb[n][k >> 5] - 127);
      maxerr = fmax(maxerr, fabs(r - hd[m * 16 + n])); maxref = fmax(maxref, fabs(r));
    }
  printf("fp6 (e2m3) layout + per-lane block scales: max |err| %.3e of max |ref| %.3e -> %s\n", maxerr, maxref,
         maxerr <= 1e-5 * maxref ? "MATCH" : "MISMATCH");
  // ---- part 2: rates
  float* sink; (void)hipMalloc(&sink, 4096);
  rate<0, 8>(sink, secs);
  rate<1, 8>(sink, secs);
  rate<2, 8>(sink, secs);
  rate<3, 8>(sink, secs);
  rate<5, 8>(sink, secs);
  rate<4, 8>(sink, secs);
  rate<4, 4>(sink, secs);
  return 0;
}
