// anatomix_amd -- stem convolution: fp32 single-channel volume -> 16*Q channels, 3x3x3 reflect
// (network.py module 0: nn.Conv3d(input_nc=1, ngf, 3, padding='same', padding_mode='reflect') +
// folded BatchNorm + ReLU).  25 FLOP/B: purely output-bandwidth bound (4 B in, 32*Q B out per voxel).
//
// Same z-marching structure as amx_conv3d_zmarch.hip: a workgroup owns an in-plane tile and marches
// along z; the fp32 input planes (with halo) stream into an LDS ring by 4-byte LDS-DMA from one
// loader wave.  The whole 27-tap stencil is ONE MFMA per 16 voxels x 16 channels: K = 32 holds the
// 27 taps (k = 8*g + e: lane groups g = 0,1,2 carry the 8 first taps of plane kz = g, group 3
// carries the three (ky,kx) = (2,2) taps and five zero weights).  Each lane gathers its 8 taps
// from the fp32 ring, rounds them to the 16-bit storage type (the same round-to-nearest the
// reference numerics emulation applies to the network input) and feeds them as the B fragment.
// The input may be a strided window of a larger volume (sliding-window caller): nothing is
// gathered or converted in HBM.
#include <stdio.h>
#include <stdlib.h>

#include "amx_device.h"

namespace amx {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kStemF32Offset = 8192;    // byte offset of the fp32 [27][Cout] weight table inside the stem's packed-weight buffer

template <int TY, int TX, int TZ, int NC, int R>
struct StemCfg {
  static constexpr int HY = TY + 2, HX = TX + 2, HVP = HY * HX;
  static constexpr int PLSZ = ((HVP * 4 + 255) / 256) * 256;      // one fp32 z-plane with halo
  static constexpr int FLAGOFF = R * PLSZ;                        // ready at +0, done[8] at +32
  static constexpr int LDS_BYTES = R * PLSZ + 64;
  static constexpr int XT = TX / 16;
  static constexpr int TILES = TZ * TY * XT;
  static constexpr int CTW = TILES / NC;
  static constexpr int WPZ = NC / TZ;
  static constexpr int ROWS_W = TY / WPZ;
  static constexpr int NDMA = (HVP + 63) / 64;                     // 4-byte DMA instructions per plane
  static constexpr int AHEAD = R - (TZ + 2);
  static_assert(TILES % NC == 0 && NC % TZ == 0 && TY % WPZ == 0 && CTW == ROWS_W * XT, "tile/wave decomposition");
  static_assert(AHEAD > TZ && R * NDMA <= 60, "ring / vmcnt range");
};

// SPLIT (strict precision): the gathered fp32 taps are split into hi + lo 16-bit halves in registers, the packed weights
// hold [Wh | Wl], three MFMAs (Wh*xh, Wh*xl, Wl*xh) replace one, and the output voxel holds [hi(Cout) | lo(Cout)].
template <typename T, int Q, int TY, int TX, int TZ, int NC, int R, bool SPLIT>
__global__ __launch_bounds__((NC + 1) * 64) void conv3d_stem_kernel(const ConvParams p, int zseg, int nseg) {
  typedef StemCfg<TY, TX, TZ, NC, R> C;
  typedef typename Ops<T>::vec8 vec8;
  constexpr int HX = C::HX, PLSZ = C::PLSZ, XT = C::XT, CTW = C::CTW, NDMA = C::NDMA;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int b = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) b = (b & 7) * (nb >> 3) + (b >> 3);
  const int bx = b % p.nbx;
  b /= p.nbx;
  const int by = b % p.nby;
  b /= p.nby;
  const int sg = b % nseg;
  const int n = b / nseg;
  const int y0 = by * TY, x0 = bx * TX;
  const int zs = sg * zseg;
  const int ze = (zs + zseg < p.D) ? zs + zseg : p.D;
  const int nplanes = ze - zs + 2;
  const int nsteps = (ze - zs + TZ - 1) / TZ;

  // producer/consumer counters in LDS instead of a workgroup barrier per step (amx_device.h)
  int* ready = (int*)(smem + C::FLAGOFF);
  int* done = (int*)(smem + C::FLAGOFF + 32);
  if (tid < 16) ((int*)(smem + C::FLAGOFF))[tid] = 0;
  __syncthreads();

  if (wave >= NC) {
    // ================================ loader wave ================================
    int off[NDMA];
    bool valid[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      const int hv = j * 64 + lane;
      const int hy = hv / HX, hx = hv - hy * HX;
      valid[j] = hv < C::HVP;
      off[j] = reflect_clamp(y0 + hy - 1, p.H) * (int)p.s0y + reflect_clamp(x0 + hx - 1, p.W) * (int)p.s0x;
    }
    const char* src_n = p.src0 + (long long)n * p.s0n;
    auto issue_plane = [&](int q) {
      const char* plane = src_n + (long long)reflect_clamp(zs - 1 + q, p.D) * p.s0z;
      char* dstp = smem + (q % R) * PLSZ;
#pragma unroll
      for (int j = 0; j < NDMA; ++j)
        if (valid[j]) __builtin_amdgcn_global_load_lds((gptr_t)(plane + off[j]), (lptr_t)(dstp + j * 256), 4, 0, 0);
    };
    int next_issue = 0, next_pub = 0;
    const unsigned a_ready = lds_addr(ready), a_done = lds_addr(done);
    while (next_pub < nplanes) {
      if (next_issue < nplanes) {
        const int md = __builtin_amdgcn_readfirstlane(flag_min8_asm(a_done));
        int lim = R + TZ * md;                              // planes q < TZ*md are dead
        lim = lim < nplanes ? lim : nplanes;
        while (next_issue < lim) issue_plane(next_issue++);
      }
      if (next_issue == next_pub) {
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      WaitVm<NDMA, R - 1>::run(next_issue - next_pub - 1);
      flag_store_asm(a_ready, ++next_pub);
    }
    return;
  }

  // ================================ consumer wave ================================
  const int li = lane & 15, g = lane >> 4;
  vec8 wreg[Q], wlo[SPLIT ? Q : 1];
#pragma unroll
  for (int q = 0; q < Q; ++q) wreg[q] = *(const vec8*)(p.wpk + q * 1024 + lane * 16);
  if (SPLIT)
#pragma unroll
    for (int q = 0; q < Q; ++q) wlo[q] = *(const vec8*)(p.wpk + (Q + q) * 1024 + lane * 16);
  f32x4 bias[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) bias[q] = p.bias ? *(const f32x4*)(p.bias + g * 4 * Q + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int tz = wave / C::WPZ;
  const int wrow = (wave % C::WPZ) * C::ROWS_W;
  const int lanepos = ((wrow * HX) + li) * 4;
  const bool full_xy = (y0 + TY <= p.H) & (x0 + TX <= p.W);
  const int yl = y0 + wrow, xl = x0 + li;
  const int ocs = p.ocs ? p.ocs : 32;                          // bytes between a voxel's 16-channel chunks (amx_common.h)
  char* out_l = p.out + (long long)n * p.on + (long long)yl * p.oy + (long long)xl * p.ox + (long long)((g * 4 * Q) >> 4) * ocs + ((g * 4 * Q) & 15) * 2;

  bool bad = false;
  for (int s = 0; s < nsteps; ++s) {
    {
      int need = TZ * s + TZ + 2;
      need = need < nplanes ? need : nplanes;
      while (flag_load(ready) < need) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    }
    const int zo = zs + s * TZ + tz;
    int sl[3];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) sl[kz] = ((s * TZ + tz + kz) % R) * PLSZ;
    // lane-group bases: groups 0..2 read plane kz = g at taps e = ky*3+kx; group 3 reads tap (2,2) of planes 0,1,2
    const int own = (g == 0 ? sl[0] : (g == 1 ? sl[1] : (g == 2 ? sl[2] : sl[0]))) + lanepos;
    int be[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) be[e] = g < 3 ? own : sl[e] + lanepos + ((2 * HX + 2) - ((e / 3) * HX + e % 3)) * 4;

    // Three phases without control flow between the tiles (the step used to be a chain of tile-after-tile basic blocks: per-value
    // activation branches, per-tile ablation tests -- ~650 instructions per four MFMAs, nothing overlapped): gather all tiles'
    // operands, multiply, activate every accumulator under one branch; only the stores are predicated.
    vec8 bf[CTW], bl[SPLIT ? CTW : 1];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int cx = c % XT, cy = c / XT;
      const int toff = (cy * HX + cx * 16) * 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int imm = ((e / 3) * HX + e % 3) * 4 + toff;
        const float f = *(const float*)(smem + (e < 3 ? be[e] : own) + imm);
        bf[c][e] = (T)f;
        if (SPLIT) bl[SPLIT ? c : 0][e] = (T)(f - (float)bf[c][e]);
      }
    }
    f32x4 acc[CTW][Q];
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int q = 0; q < Q; ++q) acc[c][q] = bias[q];
    if (!(p.dbg & 2)) {
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          acc[c][q] = Ops<T>::mfma(wreg[q], bf[c], acc[c][q]);
          if (SPLIT) {
            acc[c][q] = Ops<T>::mfma(wreg[q], bl[SPLIT ? c : 0], acc[c][q]);
            acc[c][q] = Ops<T>::mfma(wlo[q], bf[c], acc[c][q]);
          }
        }
    }
    act_inplace<CTW * Q>(&acc[0][0], p.act, p.slope);
    if (RangeCheck<T>::on) {
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) bad |= RangeCheck<T>::bad(acc[c][q][j]);   // the values about to be stored
    }
    if (zo < ze && !(p.dbg & 4)) {
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        const int cx = c % XT, cy = c / XT;
        if (!full_xy && !((yl + cy < p.H) & (xl + cx * 16 < p.W))) continue;
        char* dst = out_l + (long long)zo * p.oz + cy * p.oy + (cx * 16) * p.ox;
        float v[4 * Q];
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[q * 4 + j] = acc[c][q][j];
        unsigned w[2 * Q];
#pragma unroll
        for (int j = 0; j < 2 * Q; ++j) w[j] = (unsigned)to_bits<T>(v[2 * j]) | ((unsigned)to_bits<T>(v[2 * j + 1]) << 16);
        if (Q == 1) *(uint2*)dst = make_uint2(w[0], w[1]);
        else *(uint4*)dst = make_uint4(w[0], w[1], w[2], w[3]);
        if (SPLIT) {
#pragma unroll
          for (int j = 0; j < 2 * Q; ++j)
            w[j] = (unsigned)to_bits<T>(v[2 * j] - (float)(T)v[2 * j]) | ((unsigned)to_bits<T>(v[2 * j + 1] - (float)(T)v[2 * j + 1]) << 16);
          char* dlo = dst + (long long)(p.Cout >> 4) * ocs;
          if (Q == 1) *(uint2*)dlo = make_uint2(w[0], w[1]);
          else *(uint4*)dlo = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every ring read of this step has returned
    flag_store(done + wave, s + 1);
  }
  if (RangeCheck<T>::on) raise_flag(p.oflow, bad);
}


// -------------------------------------------------------------------------------------------------------------------------------
// VALU variant (round 3) -- MEASURED, NOT THE DEFAULT (AMX_STEM_VALU=1 selects it).  6 M stem, batch 4 x 128^3, same box: 144 us
// against 131 us for the MFMA kernel (strict, 32 channels: 385 vs 259).  Ablations of both (profiles/r03_stem_ablation.txt): input
// ring + output stores alone take 76-92 us (the 268 MB of output leave at ~3.5 TB/s at best), the arithmetic phase (37 us of
// gather + MFMA there, 54 us of packed fp32 FMAs here -- v_pk_fma_f32 is 4 cycles per wave, the stencil is 46 us of pure VALU at
// 2.4 GHz) adds to that almost serially in BOTH formulations, and more workgroups per CU (512 .. 2048) change nothing.  The stem is
// bound by its store stream plus whatever arithmetic sits in front of it, not by the operand path the verdict suspected.
// The MFMA kernel above spends ~125 instructions per MFMA gathering, converting and packing the 27 taps
// of a voxel into a K = 32 fragment: it is instruction-bound at 2.1x its write floor.  With one input channel the layer is
// 27 x Cout multiply-adds per voxel on an operand that needs no rounding at all -- the fp32 input and the fp32 (norm-folded)
// weights: here a lane owns ONE voxel, reads its 27 fp32 taps from the same LDS ring, and runs 27 x Cout v_fma_f32 whose second
// operand is a scalar register (the weight table [tap][cout] comes through the scalar cache: the constant address space makes
// hipcc use s_load for it).  No conversions, no packing, no MFMA -- and the arithmetic is exact fp32 instead of 16-bit operands.
// Same tile (8 x 32 x 2 planes per step), same loader wave, same flag protocol as above.
typedef const __attribute__((address_space(4))) float* cfloat_ptr;

template <typename T, int Q, int TY, int TX, int TZ, int NC, int R, bool SPLIT>
__global__ __launch_bounds__((NC + 1) * 64) void conv3d_stem_valu_kernel(const ConvParams p, int zseg, int nseg) {
  typedef StemCfg<TY, TX, TZ, NC, R> C;
  constexpr int HX = C::HX, PLSZ = C::PLSZ, NDMA = C::NDMA, CO = 16 * Q;
  static_assert(TX == 32 && NC * 64 == TZ * TY * TX, "one lane per voxel of a step");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) b = (b & 7) * (nb >> 3) + (b >> 3);
  const int bx = b % p.nbx;
  b /= p.nbx;
  const int by = b % p.nby;
  b /= p.nby;
  const int sg = b % nseg;
  const int n = b / nseg;
  const int y0 = by * TY, x0 = bx * TX;
  const int zs = sg * zseg;
  const int ze = (zs + zseg < p.D) ? zs + zseg : p.D;
  const int nplanes = ze - zs + 2;
  const int nsteps = (ze - zs + TZ - 1) / TZ;

  int* ready = (int*)(smem + C::FLAGOFF);
  int* done = (int*)(smem + C::FLAGOFF + 32);
  if (tid < 16) ((int*)(smem + C::FLAGOFF))[tid] = 0;
  __syncthreads();

  if (wave >= NC) {
    // ================================ loader wave (as in the MFMA kernel) ================================
    int off[NDMA];
    bool valid[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      const int hv = j * 64 + lane;
      const int hy = hv / HX, hx = hv - hy * HX;
      valid[j] = hv < C::HVP;
      off[j] = reflect_clamp(y0 + hy - 1, p.H) * (int)p.s0y + reflect_clamp(x0 + hx - 1, p.W) * (int)p.s0x;
    }
    const char* src_n = p.src0 + (long long)n * p.s0n;
    auto issue_plane = [&](int q) {
      const char* plane = src_n + (long long)reflect_clamp(zs - 1 + q, p.D) * p.s0z;
      char* dstp = smem + (q % R) * PLSZ;
#pragma unroll
      for (int j = 0; j < NDMA; ++j)
        if (valid[j]) __builtin_amdgcn_global_load_lds((gptr_t)(plane + off[j]), (lptr_t)(dstp + j * 256), 4, 0, 0);
    };
    int next_issue = 0, next_pub = 0;
    const unsigned a_ready = lds_addr(ready), a_done = lds_addr(done);
    while (next_pub < nplanes) {
      if (next_issue < nplanes) {
        const int md = __builtin_amdgcn_readfirstlane(flag_min8_asm(a_done));
        int lim = R + TZ * md;
        lim = lim < nplanes ? lim : nplanes;
        while (next_issue < lim) issue_plane(next_issue++);
      }
      if (next_issue == next_pub) {
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      WaitVm<NDMA, R - 1>::run(next_issue - next_pub - 1);
      flag_store_asm(a_ready, ++next_pub);
    }
    return;
  }

  // ================================ consumer wave: one voxel per lane ================================
  constexpr int WPZ = NC / TZ;                              // waves per output plane
  const int tz = wave / WPZ;
  const int row = (wave % WPZ) * (TY / WPZ) + (lane >> 5), col = lane & 31;
  const int lanepos = (row * HX + col) * 4;
  const int yl = y0 + row, xl = x0 + col;
  const bool in_xy = (yl < p.H) & (xl < p.W);
  char* out_l = p.out + (long long)n * p.on + (long long)yl * p.oy + (long long)xl * p.ox;
  const cfloat_ptr wt = (cfloat_ptr)(p.wpk + kStemF32Offset);          // [27][CO] fp32, norm gain folded in
  const cfloat_ptr bs = (cfloat_ptr)p.bias;

  bool bad = false;
  for (int s = 0; s < nsteps; ++s) {
    {
      int need = TZ * s + TZ + 2;
      need = need < nplanes ? need : nplanes;
      while (flag_load(ready) < need) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    }
    const int zo = zs + s * TZ + tz;
    float x[27];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      const char* pl = smem + ((s * TZ + tz + kz) % R) * PLSZ + lanepos;
#pragma unroll
      for (int e = 0; e < 9; ++e) x[kz * 9 + e] = *(const float*)(pl + ((e / 3) * HX + e % 3) * 4);
    }
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = p.bias ? bs[c] : 0.f;
    if (!(p.dbg & 2)) {
#pragma unroll
      for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = __builtin_fmaf(x[t], wt[t * CO + c], acc[c]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every ring read of this step has returned
    flag_store(done + wave, s + 1);
    if (p.act == ACT_RELU) {
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] = acc[c] > 0.f ? acc[c] : 0.f;
    } else if (p.act == ACT_LRELU) {
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] = acc[c] > 0.f ? acc[c] : acc[c] * p.slope;
    }
    if (RangeCheck<T>::on) {
#pragma unroll
      for (int c = 0; c < CO; ++c) bad |= RangeCheck<T>::bad(acc[c]);
    }
    if (zo < ze && in_xy && !(p.dbg & 4)) {
      char* dst = out_l + (long long)zo * p.oz;
      unsigned w[CO / 2];
#pragma unroll
      for (int j = 0; j < CO / 2; ++j) w[j] = (unsigned)to_bits<T>(acc[2 * j]) | ((unsigned)to_bits<T>(acc[2 * j + 1]) << 16);
#pragma unroll
      for (int j = 0; j < CO / 8; ++j) *(uint4*)(dst + j * 16) = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
      if (SPLIT) {
#pragma unroll
        for (int j = 0; j < CO / 2; ++j)
          w[j] = (unsigned)to_bits<T>(acc[2 * j] - (float)(T)acc[2 * j]) | ((unsigned)to_bits<T>(acc[2 * j + 1] - (float)(T)acc[2 * j + 1]) << 16);
#pragma unroll
        for (int j = 0; j < CO / 8; ++j) *(uint4*)(dst + p.Cout * 2 + j * 16) = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
      }
    }
  }
  if (RangeCheck<T>::on) raise_flag(p.oflow, bad);
}

// Stem weights: fp32 [Cout][1][3][3][3] (* folded norm gain) -> [q][lane 64][8] A fragments with
// k = 8*g + e <-> tap as described above; row m of tile q is channel (m>>2)*4Q + q*4 + (m&3).
template <typename T>
__global__ void pack_stem_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ wpk,
                                 int Q, int split, int CoutReal) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < 27 * 16 * Q) {                          // fp32 table of the VALU kernel: [tap][cout], gain folded in
    const int t = idx / (16 * Q), c = idx % (16 * Q);
    ((float*)((char*)wpk + kStemF32Offset))[idx] = c < CoutReal ? w[c * 27 + t] * (scale ? scale[c] : 1.f) : 0.f;
  }
  if (idx >= Q * 512 * (split ? 2 : 1)) return;
  const int part = idx / (Q * 512);                 // split: [Wh tiles | Wl tiles]
  const int e = idx & 7, lane = (idx >> 3) & 63, q = (idx >> 9) % Q;
  const int m = lane & 15, g = lane >> 4;
  const int cout = (m >> 2) * 4 * Q + q * 4 + (m & 3);
  int tap = -1;
  if (g < 3) tap = g * 9 + e;
  else if (e < 3) tap = e * 9 + 8;
  float v = 0.f;
  if (tap >= 0 && cout < CoutReal) v = w[cout * 27 + tap] * (scale ? scale[cout] : 1.f);      // padded output channels: zero rows
  wpk[idx] = part ? (T)(v - (float)(T)v) : (T)v;
}

static thread_local char g_kernel_name4[64] = "";
const char* last_conv_stem_kernel_name() { return g_kernel_name4; }

template <typename T, int Q, bool SPLIT>
static hipError_t launch_stem_t(ConvParams p, hipStream_t st) {
  constexpr int TY = 8, TX = 32, TZ = 2, NC = 8, R = 10;
  typedef StemCfg<TY, TX, TZ, NC, R> C;
  snprintf(g_kernel_name4, sizeof g_kernel_name4, "conv3d_stem<%s,q%d,%dx%dx%d,c%d+l1,r%d>",
           __is_same(T, f16) ? (SPLIT ? "f16x2" : "f16") : (SPLIT ? "bf16x2" : "bf16"), Q,
           TZ, TY, TX, NC, R);
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("AMX_DBG");
    dbg = e ? atoi(e) : 0;
  }
  p.dbg = dbg;
  p.nby = (p.H + TY - 1) / TY;
  p.nbx = (p.W + TX - 1) / TX;
  const int tiles = p.nby * p.nbx * p.N;
  static int wgs = -1;
  if (wgs < 0) wgs = getenv("AMX_STEM_WGS") ? atoi(getenv("AMX_STEM_WGS")) : 512;
  int nseg = (wgs + tiles - 1) / tiles;       // tiny LDS footprint: several workgroups per CU
  if (nseg < 1) nseg = 1;
  int zseg = (p.D + nseg - 1) / nseg;
  zseg = (zseg + TZ - 1) / TZ * TZ;
  if (zseg < 8) zseg = 8;
  nseg = (p.D + zseg - 1) / zseg;
  static int valu = -1;
  if (valu < 0) valu = getenv("AMX_STEM_VALU") ? 1 : 0;     // opt-in: measured slower (below)
  if (valu && (p.ocs == 0 || p.ocs == 32)) {      // (the experiment writes channels-last voxels only)
    snprintf(g_kernel_name4, sizeof g_kernel_name4, "conv3d_stem_valu<%s,q%d,%dx%dx%d,c%d+l1,r%d>",
             __is_same(T, f16) ? (SPLIT ? "f16x2" : "f16") : (SPLIT ? "bf16x2" : "bf16"), Q, TZ, TY, TX, NC, R);
    hipLaunchKernelGGL((conv3d_stem_valu_kernel<T, Q, TY, TX, TZ, NC, R, SPLIT>), dim3((unsigned)(tiles * nseg)), dim3((NC + 1) * 64),
                       C::LDS_BYTES, st, p, zseg, nseg);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((conv3d_stem_kernel<T, Q, TY, TX, TZ, NC, R, SPLIT>), dim3((unsigned)(tiles * nseg)), dim3((NC + 1) * 64),
                     C::LDS_BYTES, st, p, zseg, nseg);
  return hipGetLastError();
}

hipError_t launch_conv_stem(const ConvParams& p, int precision, hipStream_t st) {
  const int Q = p.Cout / 16;
  if (Q != 1 && Q != 2) return hipErrorInvalidValue;
  switch (precision) {
    case 0: return Q == 1 ? launch_stem_t<f16, 1, false>(p, st) : launch_stem_t<f16, 2, false>(p, st);
    case 1: return Q == 1 ? launch_stem_t<bf16, 1, false>(p, st) : launch_stem_t<bf16, 2, false>(p, st);
    case 2: return Q == 1 ? launch_stem_t<f16, 1, true>(p, st) : launch_stem_t<f16, 2, true>(p, st);
    case 3: return Q == 1 ? launch_stem_t<bf16, 1, true>(p, st) : launch_stem_t<bf16, 2, true>(p, st);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_pack_stem(const float* w, const float* scale, void* wpk, int Cout, int precision, hipStream_t st, int CoutReal) {
  if (CoutReal <= 0) CoutReal = Cout;
  const int Q = Cout / 16, split = precision >= 2;
  const int n = Q * 512 * (split ? 2 : 1);
  if ((precision & 1) == 0)
    hipLaunchKernelGGL(pack_stem_kernel<f16>, dim3((n + 255) / 256), dim3(256), 0, st, w, scale, (f16*)wpk, Q, split, CoutReal);
  else
    hipLaunchKernelGGL(pack_stem_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, st, w, scale, (bf16*)wpk, Q, split, CoutReal);
  return hipGetLastError();
}

}  // namespace amx
