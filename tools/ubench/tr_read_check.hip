// ds_read_b64_tr_b16 semantics check: channels-last LDS image [voxel][16 channels] (32 B per voxel); lane (group g, i) asks for
// 8 bytes at voxel (4g + i/4), channels 4(i%4)..+3 and must receive channel i of voxels 4g .. 4g+3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(int* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 4096; i += 64) ((short*)smem)[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int j = i >> 2, q = i & 3;
  __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)(smem + g * 128 + j * 32 + q * 8);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    const int g = l >> 4, i = l & 15, want = (4 * g + e) * 16 + i;
    if (h[l * 4 + e] != want) { if (bad < 8) printf("lane %d e %d: got %d want %d\n", l, e, h[l * 4 + e], want); ++bad; }
  }
  printf("tr_read_check: %d mismatches\n", bad);
  return bad != 0;
}
