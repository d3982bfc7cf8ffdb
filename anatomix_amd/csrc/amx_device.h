// anatomix_amd -- device-side helpers shared by the gfx950 kernels.
#pragma once
#include "amx_common.h"

namespace amx {

template <typename T> struct Ops;
template <> struct Ops<f16> {
  typedef f16x8 vec8;
  static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Ops<bf16> {
  typedef bf16x8 vec8;
  static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

__device__ __forceinline__ int reflect_clamp(int g, int n) {
  g = g < 0 ? -g : g;                 // -1 -> 1
  g = g >= n ? 2 * n - 2 - g : g;     //  n -> n-2
  g = g < 0 ? 0 : g;                  // only reachable for masked (out-of-volume) voxels
  return g >= n ? n - 1 : g;
}

template <typename T>
__device__ __forceinline__ unsigned short to_bits(float v) {
  T t = (T)v;
  return __builtin_bit_cast(unsigned short, t);
}

}  // namespace amx
