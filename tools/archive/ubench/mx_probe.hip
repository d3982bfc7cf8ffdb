// Micro-benchmark (round 4): the block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) as the carrier of the two
// correction products of the split arithmetic (Wh*xl + Wl*xh) beside the f16 main product.
//   part 1: operand layout and scale semantics, checked against a host computation (random e4m3 operands)
//   part 2: sustained rate of f16-only / MX-only / the 27 : 14 mix of the conv kernel, all CUs busy
// usage: mx_probe [reps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

static float e4m3(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}

// one wave, one instruction: a, b = 32 bytes per lane; scale registers per lane
__global__ void k_layout(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* d, int opsel) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 0, sb[l]);
  if (opsel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 1, sa[l], 0, sb[l]);
  if (opsel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 2, sb[l]);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

// MODE 0: f16 only, 1: MX only, 2: 27 f16 + 14 MX per iteration (one 32 -> 16-cout tap sweep of the split conv), 3: 27 x 3 f16 (today's strict)
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_rate(int iters, float* sink, int data) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 fa[4], fb[4];
  i32x8 xa[2], xb[2];
  for (int i = 0; i < 4; ++i) {
    unsigned pa[4], pb[4];
    for (int j = 0; j < 4; ++j) {
      const unsigned h = (t * 16 + i * 4 + j) * 2654435761u;
      pa[j] = data == 2 ? 0u : (((h >> 3) & 0x3fff3fffu) | 0x20002000u);
      pb[j] = data == 2 ? 0u : ((((h * 0x9E3779B1u) >> 3) & 0x3fff3fffu) | 0x20002000u);
      if (data == 1 && ((h >> 20) & 1)) pb[j] = 0u;
    }
    __builtin_memcpy(&fa[i], pa, 16);
    __builtin_memcpy(&fb[i], pb, 16);
  }
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 8; ++j) {
      const unsigned h = (t * 16 + i * 8 + j) * 0x85EBCA6Bu;
      xa[i][j] = data == 2 ? 0 : (int)(h & 0xbfbfbfbfu);
      xb[i][j] = data == 2 ? 0 : (int)((h * 2654435761u) & 0xbfbfbfbfu);
      if (data == 1 && ((h >> 20) & 1)) xb[i][j] = 0;
    }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sc = 0x7f7f7f7f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
      for (int s = 0; s < (MODE == 3 ? 81 : 27); ++s) acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[s & 3], fb[(s >> 2) & 3], acc[s & 3], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int s = 0; s < 14; ++s) acc[s & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xa[s & 1], xb[(s >> 1) & 1], acc[s & 3], 0, 0, 0, sc, 0, sc);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (r == 123.456f) sink[threadIdx.x] = r;
}

template <int MODE, int NW>
static void rate(float* sink, int data, double seconds) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<MODE, NW>), dim3(256), dim3(NW * 64), 0, 0, iters, sink, data);
  (void)hipDeviceSynchronize();
  const double f16 = (MODE == 0 || MODE == 2) ? 27 : (MODE == 3 ? 81 : 0), mx = (MODE == 1 || MODE == 2) ? 14 : 0;
  const double flop = (double)iters * 256 * NW * (f16 * 16384.0 + mx * 65536.0);
  const double cyc = (double)iters * (NW / 4.0) * (f16 * 16 + mx * 32);   // per SIMD at nominal issue
  double total = 0; int n = 0; float ms = 0;
  while (total < seconds * 1e3) {
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k_rate<MODE, NW>), dim3(256), dim3(NW * 64), 0, 0, iters, sink, data);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    total += ms; n += 20;
  }
  const double us = ms * 1e3 / 20;   // last batch = the sustained state
  printf("mode %d (%s) waves/CU %d data %d: %8.1f us/launch  executed %6.0f TFLOP/s  pipe-busy at 2.4 GHz %.2f  [products/s in f16-3x units: %.0f T]\n", MODE,
         MODE == 0 ? "f16 only" : (MODE == 1 ? "MX fp8 only" : (MODE == 2 ? "27 f16 + 14 MX" : "81 f16 (3-product split)")), NW, data, us, flop / (us * 1e-6) / 1e12,
         cyc / (us * 1e-6 * 2.4e9), (double)iters * 256 * NW * 27 * 16384.0 / (us * 1e-6) / 1e12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 1.0;
  // ---- part 1: layout
  unsigned char ha[64][32], hb[64][32];
  int hsa[64], hsb[64];
  srand(7);
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      ha[l][j] = (unsigned char)(rand() & 0xbf);
      hb[l][j] = (unsigned char)(rand() & 0xbf);
    }
  i32x8 *da, *db; int *dsa, *dsb; float* dd; float hd[256];
  (void)hipMalloc(&da, 64 * 32); (void)hipMalloc(&db, 64 * 32); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dd, 1024);
  (void)hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
  for (int test = 0; test < 4; ++test) {
    // test 0: all scales 1.0; 1: scale_a byte0 = 127 + kblock (per-lane block scales); 2: opsel_a = 1 picks byte 1; 3: opsel_b = 2 picks byte 2
    for (int l = 0; l < 64; ++l) {
      hsa[l] = 0x7f7f7f7f; hsb[l] = 0x7f7f7f7f;
      if (test == 1) hsa[l] = 0x7f7f7f00 | (127 + (l >> 4));
      if (test == 2) hsa[l] = 0x7f7f007f | ((127 + (l >> 4)) << 8);
      if (test == 3) hsb[l] = 0x7f007f7f | ((125 + (l >> 4)) << 16);
    }
    (void)hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, test == 2 ? 1 : (test == 3 ? 2 : 0));
    (void)hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane l holds row/col (l & 15), K = 32 * (l >> 4) + byte; block scale of that lane's K block = 2^(byte - 127); D: col = l & 15, row = 4 * (l >> 4) + r
    double maxerr = 0, maxref = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        double ref = 0;
        for (int kb = 0; kb < 4; ++kb) {
          double part = 0;
          for (int j = 0; j < 32; ++j) part += (double)e4m3(ha[kb * 16 + row][j]) * e4m3(hb[kb * 16 + col][j]);
          double s = 1.0;
          if (test == 1 || test == 2) s = ldexp(1.0, kb);
          if (test == 3) s = ldexp(1.0, kb - 2);
          ref += s * part;
        }
        maxerr = fmax(maxerr, fabs(ref - hd[l * 4 + r])); maxref = fmax(maxref, fabs(ref));
      }
    printf("layout test %d: max |err| %.3e of max |ref| %.3e  -> %s\n", test, maxerr, maxref, maxerr <= 1e-5 * maxref ? "hypothesis holds" : "MISMATCH");
  }
  // ---- part 2: rates
  float* sink; (void)hipMalloc(&sink, 4096);
  for (int data : {0, 1}) {
    rate<0, 8>(sink, data, secs);
    rate<1, 8>(sink, data, secs);
    rate<2, 8>(sink, data, secs);
    rate<3, 8>(sink, data, secs);
    rate<2, 4>(sink, data, secs);
  }
  return 0;
}
