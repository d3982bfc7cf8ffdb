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
  float s1[4 * Q], s2[4 * Q];                                  // InstanceNorm statistics of the stored values, shifted by the bias
#pragma unroll
  for (int k = 0; k < 4 * Q; ++k) s1[k] = s2[k] = 0.f;
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
        if (p.stats) {
#pragma unroll
          for (int k = 0; k < 4 * Q; ++k) {
            const float d = v[k] - bias[k >> 2][k & 3];
            s1[k] += d;
            s2[k] += d * d;
          }
        }
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
  if (p.stats) {
    // sum over the 16 voxel lanes of the lane group (row_shr 1, 2, 4, 8: lane 15 of the row holds it); slot = (tile, z segment, wave)
#pragma unroll
    for (int k = 0; k < 4 * Q; ++k) {
      s1[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1[k]), 0x111, 0xF, 0xF, true));
      s2[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2[k]), 0x111, 0xF, 0xF, true));
      s1[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1[k]), 0x112, 0xF, 0xF, true));
      s2[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2[k]), 0x112, 0xF, 0xF, true));
      s1[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1[k]), 0x114, 0xF, 0xF, true));
      s2[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2[k]), 0x114, 0xF, 0xF, true));
      s1[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1[k]), 0x118, 0xF, 0xF, true));
      s2[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2[k]), 0x118, 0xF, 0xF, true));
    }
    if (li == 15) {
      const long long nslots = (long long)p.nby * p.nbx * nseg * NC;
      const long long slot = (((long long)by * p.nbx + bx) * nseg + sg) * NC + wave;
      float* o = p.stats + (((long long)n * nslots + slot) * p.Cout + g * 4 * Q) * 2;
#pragma unroll
      for (int k = 0; k < 4 * Q; k += 2) *(float4*)(o + 2 * k) = make_float4(s1[k], s2[k], s1[k + 1], s2[k + 1]);
    }
  }
}


// -------------------------------------------------------------------------------------------------------------------------------
// Row-fragment formulation (round 4) -- the default.  The kernel above is VALU-ISSUE bound, not LDS- or HBM-bound: per 16-voxel
// tile a lane issues 8 ds_read_b32 + 12 conversions / packs to build ONE K = 32 fragment (~125 instructions per MFMA with the
// epilogue; profiles/r03_stem_ablation.txt: 47 of its 130 us are gather + MFMA, the matrix pipe is 3 % busy).  (A VALU-only variant
// -- one voxel per lane, 27 x Cout v_pk_fma_f32 with the weights through the scalar cache, round 3 -- was slower still: 144 us; its
// ablation numbers are in that file.)  Here the matrix pipe pays instead of the VALU:
//   * a converter wave loads the fp32 halo rows into registers, rounds them ONCE per input voxel (instead of once per tap = 27 x)
//     and writes each row to the LDS ring as 16-bit values, twice: copy A as is, copy B shifted by one element, so that the four
//     consecutive values x-1 .. x+2 of ANY x start at a 4-byte aligned address in one of the two copies;
//   * K is laid out as 16 row slots of 4 (dx = -1, 0, +1, zero) over TWO MFMAs: lane group g of MFMA m holds the rows 2p, 2p + 1
//     (p = g + 4m) of the nine (kz, ky) rows -- i.e. a lane's 16-byte B fragment IS two 8-byte LDS reads, no conversion, no packing;
//     MFMA 1 only carries row 8 (one read).  3 reads + 2 MFMAs per tile instead of 8 reads + 12 VALU + 1 MFMA.
// SPLIT: the converter writes the hi and the lo image of every row, the products are Wh*xh + Wh*xl + Wl*xh (6 MFMAs, 6 reads).
template <int TY, int TX, int TZ, int NC, int R, bool SPLIT>
struct Stem2Cfg {
  static constexpr int HY = TY + 2, HX = TX + 2, HVP = HY * HX;
  static constexpr int RS = ((HX + 2) * 2 + 7) / 8 * 8;               // row stride: HX values + 2 zero pads, 8-byte multiple
  static constexpr int CPSZ = HY * RS;                                // one copy of one image of one plane
  static constexpr int IMSZ = 2 * CPSZ;                               // copies A | B
  static constexpr int PLSZ = ((IMSZ * (SPLIT ? 2 : 1) + 8 + 63) / 64) * 64;     // + 8 bytes nobody reads: where the converter's idle lanes write
  static constexpr int DUMMY = PLSZ - 8;
  static constexpr int FLAGOFF = R * PLSZ;                            // ready at +0, done[8] at +32
  static constexpr int NLD = (HVP + 63) / 64;                         // 4-byte DMA instructions (= staged values per lane) per plane
  static constexpr int STGOFF = R * PLSZ + 64;                        // staging ring of fp32 planes as the DMA delivers them
  static constexpr int STGSZ = NLD * 256;
  static constexpr int LDS_BYTES = STGOFF + R * STGSZ;
  static constexpr int XT = TX / 16;
  static constexpr int TILES = TZ * TY * XT;
  static constexpr int CTW = TILES / NC;
  static constexpr int WPZ = NC / TZ;
  static constexpr int ROWS_W = TY / WPZ;
  static_assert(TILES % NC == 0 && NC % TZ == 0 && TY % WPZ == 0 && CTW == ROWS_W * XT, "tile/wave decomposition");
  static_assert(R * NLD <= 60, "vmcnt range");
};

template <typename T, int Q, int TY, int TX, int TZ, int NC, int R, bool SPLIT>
__global__ __launch_bounds__((NC + 1) * 64) void conv3d_stem2_kernel(const ConvParams p, int zseg, int nseg) {
  typedef Stem2Cfg<TY, TX, TZ, NC, R, SPLIT> C;
  typedef typename Ops<T>::vec8 vec8;
  constexpr int HX = C::HX, PLSZ = C::PLSZ, XT = C::XT, CTW = C::CTW, NLD = C::NLD, RS = C::RS, CPSZ = C::CPSZ, IMSZ = C::IMSZ;

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
  // the pad slots of every row (and the flags) start as zeros and are never written again: a fragment's fourth value meets a zero
  // weight, so it must be finite
  for (int i = tid; i < C::LDS_BYTES / 4; i += (NC + 1) * 64) ((int*)smem)[i] = 0;
  __syncthreads();

  if (wave >= NC) {
    // ================================ converter wave ================================
    // The fp32 halo plane arrives by 4-byte LDS-DMA in a staging ring (no register destinations: loads whose results land
    // asynchronously in VGPRs cannot be expressed safely around hipcc's register allocation, and its own wait counting collapsed
    // to one plane of lookahead); this wave then reads ITS six values back, rounds them and writes the two row copies.  Every LDS
    // access of this wave is inline asm: with an LDS-DMA pending hipcc would put vmcnt(0) in front of each one.  Lanes without a
    // voxel write a dummy slot -- no predication, no branches.
    int goff[NLD], la[NLD], lb[NLD];
    bool valid[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int hv = j * 64 + lane;
      valid[j] = hv < C::HVP;
      const int hvc = valid[j] ? hv : C::HVP - 1;
      const int hy = hvc / HX, hx = hvc - hy * HX;
      goff[j] = reflect_clamp(y0 + hy - 1, p.H) * (int)p.s0y + reflect_clamp(x0 + hx - 1, p.W) * (int)p.s0x;
      la[j] = valid[j] ? hy * RS + hx * 2 : C::DUMMY;                            // copy A: value hx at element hx
      lb[j] = (valid[j] && hx > 0) ? CPSZ + hy * RS + (hx - 1) * 2 : C::DUMMY;   // copy B: value hx at element hx - 1
    }
    static_assert(NLD == 6, "the staging read below names six registers");
    const char* src_n = p.src0 + (long long)n * p.s0n;
    char* stg = smem + C::STGOFF;
    auto issue_plane = [&](int q) {
      const char* plane = src_n + (long long)reflect_clamp(zs - 1 + q, p.D) * p.s0z;
      char* dstp = stg + (q % R) * C::STGSZ;
#pragma unroll
      for (int j = 0; j < NLD; ++j)
        if (valid[j]) __builtin_amdgcn_global_load_lds((gptr_t)(plane + goff[j]), (lptr_t)(dstp + j * 256), 4, 0, 0);
    };
    int next_issue = 0, next_pub = 0;
    const unsigned a_ready = lds_addr(ready), a_done = lds_addr(done);
    const unsigned a_stg = lds_addr(stg) + lane * 4, a_ring = lds_addr(smem);
    while (next_pub < nplanes) {
      {
        int lim = next_pub + R;                               // staging slot q % R is free once plane q - R has been converted
        lim = lim < nplanes ? lim : nplanes;
        while (next_issue < lim) issue_plane(next_issue++);
      }
      // the ring slot of plane next_pub is free once the consumers are done with plane next_pub - R: planes < TZ * min(done) are dead
      while (next_pub >= R + TZ * __builtin_amdgcn_readfirstlane(flag_min8_asm(a_done))) __builtin_amdgcn_s_sleep(1);
      WaitVm<NLD, R - 1>::run(next_issue - next_pub - 1);     // the oldest unconverted plane has landed in the staging ring
      float f[NLD];
      asm volatile("ds_read_b32 %0, %6\n\tds_read_b32 %1, %6 offset:256\n\tds_read_b32 %2, %6 offset:512\n\t"
                   "ds_read_b32 %3, %6 offset:768\n\tds_read_b32 %4, %6 offset:1024\n\tds_read_b32 %5, %6 offset:1280\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5])
                   : "v"(a_stg + (unsigned)((next_pub % R) * C::STGSZ))
                   : "memory");
      const unsigned dst = a_ring + (unsigned)((next_pub % R) * PLSZ);
#pragma unroll
      for (int j = 0; j < NLD; ++j) {
        const T h = (T)f[j];
        const unsigned hb = __builtin_bit_cast(unsigned short, h);
        asm volatile("ds_write_b16 %0, %2\n\tds_write_b16 %1, %2" ::"v"(dst + (unsigned)la[j]), "v"(dst + (unsigned)lb[j]), "v"(hb) : "memory");
        if (SPLIT) {
          const T l = (T)(f[j] - (float)h);
          const unsigned lbits = __builtin_bit_cast(unsigned short, l);
          asm volatile("ds_write_b16 %0, %2\n\tds_write_b16 %1, %2" ::"v"(dst + (unsigned)(la[j] == C::DUMMY ? C::DUMMY : IMSZ + la[j])),
                       "v"(dst + (unsigned)(lb[j] == C::DUMMY ? C::DUMMY : IMSZ + lb[j])), "v"(lbits) : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      flag_store_asm(a_ready, ++next_pub);
    }
    return;
  }

  // ================================ consumer wave ================================
  const int li = lane & 15, g = lane >> 4;
  // A fragments: [part (Wh | Wl)][mfma m][q]
  vec8 wreg[2][Q], wlo[SPLIT ? 2 : 1][SPLIT ? Q : 1];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      wreg[m][q] = *(const vec8*)(p.wpk + 4096 + (m * Q + q) * 1024 + lane * 16);
      if (SPLIT) wlo[SPLIT ? m : 0][SPLIT ? q : 0] = *(const vec8*)(p.wpk + 4096 + ((2 + m) * Q + q) * 1024 + lane * 16);
    }
  f32x4 bias[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) bias[q] = p.bias ? *(const f32x4*)(p.bias + g * 4 * Q + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int tz = wave / C::WPZ;
  const int wrow = (wave % C::WPZ) * C::ROWS_W;
  const bool full_xy = (y0 + TY <= p.H) & (x0 + TX <= p.W);
  const int yl = y0 + wrow, xl = x0 + li;
  const int ocs = p.ocs ? p.ocs : 32;
  char* out_l = p.out + (long long)n * p.on + (long long)yl * p.oy + (long long)xl * p.ox + (long long)((g * 4 * Q) >> 4) * ocs + ((g * 4 * Q) & 15) * 2;
  // lane constants of the three reads: rows (2g, 2g + 1) of MFMA 0, row 8 of MFMA 1; row r = 3 kz + ky
  const int r0 = 2 * g, r1 = 2 * g + 1;
  const int kz0 = r0 / 3, kz1 = r1 / 3;
  // odd lanes read copy B (value hx sits at element hx - 1: their four values start 4-byte aligned there)
  const int xpart = (li & 1) * CPSZ + (li - (li & 1)) * 2 + wrow * RS;
  const int ro0 = (r0 % 3) * RS + xpart, ro1 = (r1 % 3) * RS + xpart, ro8 = 2 * RS + xpart;

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
    const int b0 = (kz0 == 0 ? sl[0] : (kz0 == 1 ? sl[1] : sl[2])) + ro0;
    const int b1 = (kz1 == 0 ? sl[0] : (kz1 == 1 ? sl[1] : sl[2])) + ro1;
    const int b8 = sl[2] + ro8;

    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    auto rd = [&](int addr) -> u32x2 {             // two dwords at a 4-byte aligned address (ds_read2_b32)
      const unsigned* q = (const unsigned*)(smem + addr);
      return u32x2{q[0], q[1]};
    };
    vec8 f0[CTW], f1[CTW], l0[SPLIT ? CTW : 1], l1[SPLIT ? CTW : 1];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int cx = c % XT, cy = c / XT;
      const int toff = cy * RS + cx * 32;
      const u32x2 a = rd(b0 + toff), bq = rd(b1 + toff), e = rd(b8 + toff);
      f0[c] = __builtin_bit_cast(vec8, u32x4{a[0], a[1], bq[0], bq[1]});
      f1[c] = __builtin_bit_cast(vec8, u32x4{g == 0 ? e[0] : 0u, g == 0 ? e[1] : 0u, 0u, 0u});
      if (SPLIT) {
        const u32x2 al = rd(b0 + toff + IMSZ), bl = rd(b1 + toff + IMSZ), el = rd(b8 + toff + IMSZ);
        l0[SPLIT ? c : 0] = __builtin_bit_cast(vec8, u32x4{al[0], al[1], bl[0], bl[1]});
        l1[SPLIT ? c : 0] = __builtin_bit_cast(vec8, u32x4{g == 0 ? el[0] : 0u, g == 0 ? el[1] : 0u, 0u, 0u});
      }
    }
    f32x4 acc[CTW][Q];
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        acc[c][q] = Ops<T>::mfma(wreg[0][q], f0[c], bias[q]);
        acc[c][q] = Ops<T>::mfma(wreg[1][q], f1[c], acc[c][q]);
        if (SPLIT) {
          acc[c][q] = Ops<T>::mfma(wreg[0][q], l0[SPLIT ? c : 0], acc[c][q]);
          acc[c][q] = Ops<T>::mfma(wreg[1][q], l1[SPLIT ? c : 0], acc[c][q]);
          acc[c][q] = Ops<T>::mfma(wlo[SPLIT ? 0 : 0][SPLIT ? q : 0], f0[c], acc[c][q]);
          acc[c][q] = Ops<T>::mfma(wlo[SPLIT ? 1 : 0][SPLIT ? q : 0], f1[c], acc[c][q]);
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every ring read of this step has returned (the MFMAs consumed them)
    flag_store(done + wave, s + 1);
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
  }
  if (RangeCheck<T>::on) raise_flag(p.oflow, bad);
}

// Packed stem weights.  Bytes [0, 4096): the tap-gather kernel's tiles [part][q][lane][8], k = 8*g + e <-> tap as described at the top.
// Bytes [4096, 12288): the row-fragment kernel's tiles [part (Wh | Wl)][mfma m][q][lane][8]: k = 8 g + e of MFMA m <-> row
// r = 2 (g + 4 m) + (e >> 2) = 3 kz + ky, dx = e & 3 (3: zero), i.e. tap 3 r + dx; rows >= 9: zero.
// k = 8*g + e <-> tap as described above; row m of tile q is channel (m>>2)*4Q + q*4 + (m&3).
template <typename T>
__global__ void pack_stem_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ wpk,
                                 int Q, int split, int CoutReal) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < 2 * Q * 512 * (split ? 2 : 1)) {       // the row-fragment kernel's tiles
    const int e = idx & 7, lane = (idx >> 3) & 63;
    int r = idx >> 9;
    const int q = r % Q;
    r /= Q;
    const int m = r & 1, part = r >> 1;
    const int mm = lane & 15, g = lane >> 4;
    const int cout = (mm >> 2) * 4 * Q + q * 4 + (mm & 3);
    const int row = 2 * (g + 4 * m) + (e >> 2), dx = e & 3;
    float v = 0.f;
    if (row < 9 && dx < 3 && cout < CoutReal) v = w[cout * 27 + row * 3 + dx] * (scale ? scale[cout] : 1.f);
    ((T*)((char*)wpk + 4096))[idx] = part ? (T)(v - (float)(T)v) : (T)v;
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

// z segments per (n, y, x) tile -- tiny LDS footprint: several workgroups per CU
static void stem_segments(const ConvParams& p, int TY, int TX, int TZ, int* zseg_out, int* nseg_out) {
  const int tiles = ((p.H + TY - 1) / TY) * ((p.W + TX - 1) / TX) * p.N;
  static int wgs = -1;
  if (wgs < 0) wgs = exp_env("AMX_STEM_WGS") ? atoi(exp_env("AMX_STEM_WGS")) : 512;
  int nseg = (wgs + tiles - 1) / tiles;
  if (nseg < 1) nseg = 1;
  int zseg = (p.D + nseg - 1) / nseg;
  zseg = (zseg + TZ - 1) / TZ * TZ;
  if (zseg < 8) zseg = 8;
  *zseg_out = zseg;
  *nseg_out = (p.D + zseg - 1) / zseg;
}

// InstanceNorm statistics in the epilogue (ConvParams::stats): the tap-gather kernel of the split precisions, whole tiles only.
// Returns the slots per sample ([n][slot][Cout][2] partial sums), 0 when the layer must keep its separate statistics pass.
int conv_stem_stats_slots(const ConvParams& p, int precision) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_FUSED_STATS") ? 1 : 0;
  if (precision < 2 || off) return 0;
  constexpr int TY = 8, TX = 32, TZ = 2, NC = 8;
  if (p.H % TY || p.W % TX || p.D % TZ) return 0;
  int zseg, nseg;
  stem_segments(p, TY, TX, TZ, &zseg, &nseg);
  if (p.D % zseg) return 0;
  return (p.H / TY) * (p.W / TX) * nseg * NC;
}

template <typename T, int Q, bool SPLIT>
static hipError_t launch_stem_t(ConvParams p, hipStream_t st) {
  constexpr int TY = 8, TX = 32, TZ = 2, NC = 8, R = 10;
  typedef StemCfg<TY, TX, TZ, NC, R> C;
  snprintf(g_kernel_name4, sizeof g_kernel_name4, "conv3d_stem<%s,q%d,%dx%dx%d,c%d+l1,r%d>",
           __is_same(T, f16) ? (SPLIT ? "f16x2" : "f16") : (SPLIT ? "bf16x2" : "bf16"), Q,
           TZ, TY, TX, NC, R);
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = exp_env("AMX_DBG");
    dbg = e ? atoi(e) : 0;
  }
  p.dbg = dbg;
  p.nby = (p.H + TY - 1) / TY;
  p.nbx = (p.W + TX - 1) / TX;
  const int tiles = p.nby * p.nbx * p.N;
  int zseg, nseg;
  stem_segments(p, TY, TX, TZ, &zseg, &nseg);
  // The row-fragment kernel is the default where it measured faster: single 16-bit storage (6 M stem, batch 4 x 128^3, same box:
  // 135.3 -> 91.5 us).  In the split precisions the layer writes twice the bytes and is store-bound either way (tap-gather
  // 220 / 145 us vs rows 235 / 152 us for 32 / 16 channels): those keep the tap-gather kernel.  AMX_STEM_GATHER=1 / =0 force one.
  static int gather = -1;
  if (gather < 0) gather = exp_env("AMX_STEM_GATHER") ? atoi(exp_env("AMX_STEM_GATHER")) : 2;
  if (!p.stats && (gather == 0 || (gather == 2 && !SPLIT))) {
    if (p.dbg & 2) p.dbg |= 4;
    typedef Stem2Cfg<TY, TX, TZ, NC, R, SPLIT> C2;
    snprintf(g_kernel_name4, sizeof g_kernel_name4, "conv3d_stem<%s,q%d,%dx%dx%d,c%d+cv1,r%d,rows>",
             __is_same(T, f16) ? (SPLIT ? "f16x2" : "f16") : (SPLIT ? "bf16x2" : "bf16"), Q, TZ, TY, TX, NC, R);
    hipLaunchKernelGGL((conv3d_stem2_kernel<T, Q, TY, TX, TZ, NC, R, SPLIT>), dim3((unsigned)(tiles * nseg)), dim3((NC + 1) * 64),
                       C2::LDS_BYTES, st, p, zseg, nseg);
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
  const int n = 2 * Q * 512 * (split ? 2 : 1);     // threads: the larger (row-fragment) packing
  if ((precision & 1) == 0)
    hipLaunchKernelGGL(pack_stem_kernel<f16>, dim3((n + 255) / 256), dim3(256), 0, st, w, scale, (f16*)wpk, Q, split, CoutReal);
  else
    hipLaunchKernelGGL(pack_stem_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, st, w, scale, (bf16*)wpk, Q, split, CoutReal);
  return hipGetLastError();
}

}  // namespace amx
