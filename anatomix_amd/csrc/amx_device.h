// anatomix_amd -- device-side helpers shared by the gfx950 kernels.
#pragma once
#include "amx_common.h"

namespace amx {

template <typename T> struct Ops;
typedef __attribute__((ext_vector_type(2))) unsigned amx_u32x2;
template <> struct Ops<f16> {
  typedef f16x8 vec8;
  static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  // K = 16 form (lane group g holds k = 4 g .. 4 g + 3): operands as two dwords
  static __device__ __forceinline__ f32x4 mfma16(amx_u32x2 a, amx_u32x2 b, f32x4 c) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, a), __builtin_bit_cast(h4, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned pack2(float a, float b) {      // round-to-nearest pair, a in the low half (v_cvt_pk_f16_f32)
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, h2));
  }
};
template <> struct Ops<bf16> {
  typedef bf16x8 vec8;
  static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(amx_u32x2 a, amx_u32x2 b, f32x4 c) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, b2));
  }
};

// f16 storage overflows at 65504; bf16 shares fp32's exponent range.  Epilogues that store f16 (or f16x2) values test them
// with RangeCheck<T>::bad and raise ConvParams::oflow -- an overflow must never turn into silent Inf (or, behind a ReLU or
// max-pool, silently finite) features.
template <typename T> struct RangeCheck {
  static constexpr bool on = false;
  static __device__ __forceinline__ bool bad(float) { return false; }
};
template <> struct RangeCheck<f16> {
  static constexpr bool on = true;
  static __device__ __forceinline__ bool bad(float v) { return !(__builtin_fabsf(v) <= 65504.f); }   // also true for NaN
};
__device__ __forceinline__ void raise_flag(int* flag, bool bad) {
  if (bad && flag) __atomic_store_n(flag, 1, __ATOMIC_RELAXED);
}

__device__ __forceinline__ int reflect_clamp(int g, int n) {
  g = g < 0 ? -g : g;                 // -1 -> 1
  g = g >= n ? 2 * n - 2 - g : g;     //  n -> n-2
  g = g < 0 ? 0 : g;                  // only reachable for masked (out-of-volume) voxels
  return g >= n ? n - 1 : g;
}

// source coordinate of a halo voxel: reflected, or (raw: the source is the interior of a zero-framed buffer) the neighbour itself,
// kept inside the one-voxel frame for the out-of-volume voxels of partial tiles
__device__ __forceinline__ int halo_coord(int g, int n, int raw) {
  if (raw) return g < -1 ? -1 : (g > n ? n : g);
  return reflect_clamp(g, n);
}

// ---- fp8 side of precision AMX_PREC_F16X2_MX (include/anatomix_amd.h) ---------------------------------------------------------
// A stored value v = hi + lo (f16 pair).  The conv multiplies Wh * hi on the f16 MFMA and forms the two correction products
// Wh * lo + Wl * hi on the block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, twice the f16 rate) from OCP e4m3 copies
//   xh8 = e4m3(hi)          xl8 = e4m3(2^11 lo)          Wh8 = e4m3(2^Sw Wh)          Wl8 = e4m3(2^(Sw+11) Wl)
// (|lo| <= 2^-11 |hi|, so both activation copies share the range of the value itself and both products carry the same power of
// two, which the instruction's block scale removes: ONE uniform scale 2^-(Sw+11)).  Conversions saturate at +-448: a correction term
// of a value beyond that is clipped, i.e. that value is handled with plain f16 accuracy -- never an Inf / NaN.
typedef __attribute__((ext_vector_type(8))) int i32x8;
constexpr float kMxLoScale = 2048.f;        // 2^11
__device__ __forceinline__ unsigned e4m3_pk4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned)v;
}
// the eight values of one 8-channel group -> {xl8 bytes (uint2), xh8 bytes (uint2)}
__device__ __forceinline__ void mx_split8(const float (&f)[8], uint2& xl8, uint2& xh8) {
  float h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (float)(_Float16)f[e];
    l[e] = (float)(_Float16)(f[e] - h[e]) * kMxLoScale;       // from the STORED pair: the copies are a function of the voxel's hi / lo
  }
  xl8 = make_uint2(e4m3_pk4(l[0], l[1], l[2], l[3]), e4m3_pk4(l[4], l[5], l[6], l[7]));
  xh8 = make_uint2(e4m3_pk4(h[0], h[1], h[2], h[3]), e4m3_pk4(h[4], h[5], h[6], h[7]));
}

// The same for a value that is NOT stored as a pair (the normalise-on-load converters): the lo part goes to e4m3 without the detour
// through f16 (two conversions per value less; at least as close to the fp32 value)
__device__ __forceinline__ void mx_split8_direct(const float (&f)[8], uint2& xl8, uint2& xh8) {
  float h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (float)(_Float16)f[e];
    l[e] = (f[e] - h[e]) * kMxLoScale;
  }
  xl8 = make_uint2(e4m3_pk4(l[0], l[1], l[2], l[3]), e4m3_pk4(l[4], l[5], l[6], l[7]));
  xh8 = make_uint2(e4m3_pk4(h[0], h[1], h[2], h[3]), e4m3_pk4(h[4], h[5], h[6], h[7]));
}

// Store the copies of one 8-channel group.  The lanes (2k, 2k + 1) of a wave hold the two groups of ONE 16-channel chunk of one voxel
// (every pass maps consecutive threads to consecutive groups and C / 8 is even): they exchange halves (DPP quad_perm [1,0,3,2]) so
// that the even lane writes the chunk's 16 xl8 bytes and the odd lane its 16 xh8 bytes -- two 16-byte stores per 32-byte chunk
// instead of four 8-byte ones (norm apply at level 0 of anatomix-dev: 670 -> see DESIGN.md).  `chunk` = voxel + 4 C + 32 (c8 >> 1).
__device__ __forceinline__ unsigned dpp_xor1(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ void mx_store_copies(char* chunk, int odd, const float (&f)[8]) {
  uint2 xl8, xh8;
  mx_split8(f, xl8, xh8);
  const uint2 send = odd ? xl8 : xh8;             // what the partner stores
  const unsigned r0 = dpp_xor1(send.x), r1 = dpp_xor1(send.y);
  *(uint4*)(chunk + odd * 16) = odd ? make_uint4(r0, r1, xh8.x, xh8.y) : make_uint4(xl8.x, xl8.y, r0, r1);
}

template <typename T>
__device__ __forceinline__ unsigned short to_bits(float v) {
  T t = (T)v;
  return __builtin_bit_cast(unsigned short, t);
}

// 4x4 transpose across the four lanes of a quad with DPP quad_perm moves (VALU only):
// afterwards register j of quad lane i holds what register i of quad lane j held.
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose4(float (&v)[4], int lane_in_quad_bits) {
  const bool b0 = lane_in_quad_bits & 1, b1 = lane_in_quad_bits & 2;
  // exchange with lane ^ 1 (quad_perm [1,0,3,2] = 0xB1): 2x2 blocks of registers (0,1) and (2,3)
  const float r01 = dpp_quad<0xB1>(b0 ? v[0] : v[1]);
  const float r23 = dpp_quad<0xB1>(b0 ? v[2] : v[3]);
  if (b0) { v[0] = r01; v[2] = r23; } else { v[1] = r01; v[3] = r23; }
  // exchange with lane ^ 2 (quad_perm [2,3,0,1] = 0x4E): registers (0,2) and (1,3)
  const float r02 = dpp_quad<0x4E>(b1 ? v[0] : v[2]);
  const float r13 = dpp_quad<0x4E>(b1 ? v[1] : v[3]);
  if (b1) { v[0] = r02; v[1] = r13; } else { v[2] = r02; v[3] = r13; }
}

// ---- intra-workgroup producer/consumer flags in LDS (loader waves <-> MFMA waves) --------------
// A monotonically increasing 32-bit counter per producer / consumer wave replaces the workgroup
// barrier of the z-marching kernels, so that a wave blocked on a full memory queue (VMEM issue
// back-pressure) delays only the waves that really depend on it.  LDS is a single in-order
// memory per CU: a ds_write issued after `s_waitcnt vmcnt` (LDS-DMA data landed) or after
// `s_waitcnt lgkmcnt(0)` (own ds_reads returned) is observed by other waves after the data.
__device__ __forceinline__ int flag_load(const int* p) {
  return __atomic_load_n(p, __ATOMIC_RELAXED);
}
__device__ __forceinline__ void flag_store(int* p, int v) {
  __atomic_store_n(p, v, __ATOMIC_RELAXED);
}

// Loader-side variants in inline asm: hipcc treats an in-flight LDS-DMA as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of every LDS access of the issuing wave, which would drain the whole
// prefetch queue at each flag access.  The asm forms are invisible to that pass; ordering against the
// DMA data is provided explicitly by the counted vmcnt wait that precedes the store.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// Activation of a wave's accumulators in place, under ONE uniform branch.  `p.act` is a run-time value; testing it per output value
// (as the epilogues did) compiled to a scalar compare-and-branch chain per value -- ~120 scalar instructions per z-march step and
// no scheduling across the values.  (A branch-free form, (f > 0 ? f : f * k) + 0 with k = 0 / slope / 1, was measured too: it
// helps the MFMA-heavy kernels but costs 4 VALU per value, which the VALU-bound stem and the plain 16 -> 16 layer lose again.)
template <int N>
__device__ __forceinline__ void act_inplace(f32x4* a, int act, float slope) {
  if (act == ACT_RELU) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[i][j] = a[i][j] > 0.f ? a[i][j] : 0.f;
  } else if (act == ACT_LRELU) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[i][j] = a[i][j] > 0.f ? a[i][j] : a[i][j] * slope;
  }
}

// Branch-free activation for the bandwidth-bound elementwise kernels (norm apply, BatchNorm forward / backward): k = 0 (relu),
// the slope (leaky relu) or 1 (none); the VALU cost is irrelevant there, the compare-and-branch chains per element were not
// (16 % of in_apply_fast's instructions were branches).  "+ 0" turns the -0 of relu's negative side into +0.
__device__ __forceinline__ float act_k(int act, float slope) { return act == ACT_RELU ? 0.f : (act == ACT_LRELU ? slope : 1.f); }
__device__ __forceinline__ float act_fwd(float v, float k) { return (v > 0.f ? v : v * k) + 0.f; }
__device__ __forceinline__ float act_bwd(float dz, float y, float k) {   // dz * act'(y); an exact 0 where relu masks (also for inf)
  const float m = y > 0.f ? 1.f : k;
  return m == 0.f ? 0.f : dz * m;
}

// LDS-DMA of 16 bytes per lane, written as inline asm: lane l's 16 bytes land at lds_base + 16 l (lds_base wave-uniform).
// Why not the builtin: with a builtin LDS-DMA pending in a wave hipcc falls back to `s_waitcnt lgkmcnt(0)` for every LDS read of
// that wave (no counted waits), which exposes the read latency in an MFMA sweep that runs beside its own prefetch.  The caller
// orders the data with its own `s_waitcnt vmcnt` + barrier / flag, as it has to with the builtin.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // m0 is "reserved": nothing else in these kernels uses it
__device__ __forceinline__ void dma16_asm(const void* g, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ void flag_store_asm(unsigned addr, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ int flag_load_asm(unsigned addr) {
  int r;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  return r;
}

// min over the first N (<= 4) of four consecutive int flags (16-byte aligned) with one ds_read_b128: the consumer waves polled their
// loaders' `ready` flags one relaxed load at a time -- four dependent LDS round trips per poll with four loaders (900-1100 cycles of
// "wait" per z-march step even when the planes had landed, against 250 with two)
typedef __attribute__((ext_vector_type(4))) int flag4_t;
__device__ __forceinline__ flag4_t flag_read4(const int* p) {      // four consecutive flags, one LDS round trip
  flag4_t a;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(lds_addr(p)) : "memory");
  return a;
}
__device__ __forceinline__ int flag_min2(const int* p) {             // min of two consecutive flags (8-byte aligned), one LDS round trip
  typedef __attribute__((ext_vector_type(2))) int i32x2;
  i32x2 a;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(lds_addr(p)) : "memory");
  return __builtin_amdgcn_readfirstlane(a[0] < a[1] ? a[0] : a[1]);
}
template <int N>
__device__ __forceinline__ int flag_min4(const int* p) {
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  i32x4 a;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(lds_addr(p)) : "memory");
  int m = a[0];
#pragma unroll
  for (int i = 1; i < N; ++i) m = a[i] < m ? a[i] : m;
  return __builtin_amdgcn_readfirstlane(m);                  // every lane read the same words: a scalar, so the poll loop is a scalar branch
}

// min over 8 consecutive int flags (32-byte aligned) with two ds_read_b128 and ONE wait
__device__ __forceinline__ int flag_min8_asm(unsigned addr) {
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  i32x4 a, b;
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(a), "=&v"(b)
               : "v"(addr)
               : "memory");
  int m0 = a[0] < a[1] ? a[0] : a[1], m1 = a[2] < a[3] ? a[2] : a[3];
  int m2 = b[0] < b[1] ? b[0] : b[1], m3 = b[2] < b[3] ? b[2] : b[3];
  m0 = m0 < m1 ? m0 : m1;
  m2 = m2 < m3 ? m2 : m3;
  return m0 < m2 ? m0 : m2;
}

// s_waitcnt vmcnt(k * PER) for a run-time k in [0, MAXK]; larger k: no wait needed.
template <int PER, int I>
struct WaitVm {
  static __device__ __forceinline__ void run(int k) {
    if (k == I) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(I * PER) : "memory");
    } else {
      WaitVm<PER, I - 1>::run(k);
    }
  }
};
template <int PER>
struct WaitVm<PER, -1> {
  static __device__ __forceinline__ void run(int) {}
};

// s_waitcnt vmcnt(n) for a wave-uniform run-time n.  The field is 6 bits: n > 63 waits for <= 63 outstanding, which is
// stricter than asked (vector memory operations of one wave complete in order), hence still correct.
template <int I>
struct WaitVmDyn {
  static __device__ __forceinline__ void run(int n) {
    if (n >= I) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(I) : "memory");
    } else {
      WaitVmDyn<I - 1>::run(n);
    }
  }
};
template <>
struct WaitVmDyn<0> {
  static __device__ __forceinline__ void run(int) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};
__device__ __forceinline__ void wait_vmcnt_dyn(int n) { WaitVmDyn<63>::run(n); }

}  // namespace amx
