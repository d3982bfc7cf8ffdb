// anatomix_amd -- bandwidth kernels around the sliding-window caller
// (monai.inferers.sliding_window_inference as used by
//  /root/reference/anatomix/registration/convex_adam_utils.py:202-219).
#include "amx_common.h"

namespace amx {

// acc[c][v] /= cnt[v]  (the final `output_image / count_map` of sliding_window_inference)
__global__ void sw_normalize_kernel(float* __restrict__ acc, const float* __restrict__ cnt, int channels,
                                    long long voxels) {
  const long long v4n = voxels >> 2;  // float4 path; tail handled scalar
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < v4n;
       i += (long long)gridDim.x * blockDim.x) {
    const float4 c = ((const float4*)cnt)[i];
    const float4 r = make_float4(1.f / c.x, 1.f / c.y, 1.f / c.z, 1.f / c.w);
    for (int ch = 0; ch < channels; ++ch) {
      float4* p = (float4*)(acc + (long long)ch * voxels) + i;
      float4 a = *p;
      a.x *= r.x; a.y *= r.y; a.z *= r.z; a.w *= r.w;
      *p = a;
    }
  }
  if (blockIdx.x == 0)
    for (long long i = (v4n << 2) + threadIdx.x; i < voxels; i += blockDim.x)
      for (int ch = 0; ch < channels; ++ch) acc[(long long)ch * voxels + i] /= cnt[i];
}

// cnt[oz+z][oy+y][ox+x] += wmap[z][y][x]
__global__ void sw_count_kernel(float* __restrict__ cnt, int vh, int vw, int oz, int oy, int ox, int rd, int rh,
                                int rw, const float* __restrict__ wmap) {
  const long long total = (long long)rd * rh * rw;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = i % rw;
    const long long r = i / rw;
    const int y = r % rh;
    const int z = r / rh;
    cnt[((long long)(oz + z) * vh + (oy + y)) * vw + ox + x] += wmap[i];
  }
}

hipError_t launch_sw_normalize(float* acc, const float* cnt, int channels, long long voxels, hipStream_t st) {
  if ((voxels & 3) && ((uintptr_t)acc & 15)) { /* alignment of channel planes is handled by the scalar tail only when voxels%4==0 */ }
  const long long v4n = voxels >> 2;
  int blocks = (int)((v4n + 255) / 256 > 4096 ? 4096 : (v4n + 255) / 256);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sw_normalize_kernel, dim3(blocks), dim3(256), 0, st, acc, cnt, channels, voxels);
  return hipGetLastError();
}

hipError_t launch_sw_count(float* cnt, int vd, int vh, int vw, int oz, int oy, int ox, int rd, int rh, int rw,
                           const float* wmap, hipStream_t st) {
  (void)vd;
  const long long total = (long long)rd * rh * rw;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(sw_count_kernel, dim3(blocks), dim3(256), 0, st, cnt, vh, vw, oz, oy, ox, rd, rh, rw, wmap);
  return hipGetLastError();
}

}  // namespace amx
