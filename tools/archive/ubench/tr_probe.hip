#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short* out) {
  __shared__ unsigned short lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  // candidate addressing: lane t of a 16-lane group supplies row r = t>>2 (32-B rows), column quad q = t&3; group g at +128 B
  const int t = l & 15, g = l >> 4;
  __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + g * 128 + (t >> 2) * 32 + (t & 3) * 8);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l*4+j]); printf("   expect col %d rows %d..: %d %d %d %d\n", l&15, 4*(l>>4), (4*(l>>4))*16+(l&15), (4*(l>>4)+1)*16+(l&15), (4*(l>>4)+2)*16+(l&15), (4*(l>>4)+3)*16+(l&15)); }
  return 0;
}
