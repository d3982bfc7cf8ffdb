// anatomix_amd -- conv3d 3x3x3 reflect, 32 -> 32 channels at full resolution in precision AMX_PREC_F16X2_MX, with the
// InstanceNorm + activation of its INPUT applied on the way in: the level-0 layers of `anatomix-dev`
// (/root/reference/anatomix/model/network.py:334-369,403-445 with the kwargs of load_from_hf.py:18-24: modules 3, 6, 76).
//
// Why a kernel of its own.  On the generic kernel these layers are (a) preceded by a norm-apply pass over a 1 GB tensor each (the
// conv stores RAW outputs because InstanceNorm needs the whole plane first; `y = act(a x + b)` then runs as its own read + write:
// 430 us per layer at batch 4, 28 % of the f16x2mx forward in those passes) and (b) bound by LDS traffic, because the packed
// weights come through the LDS with every stage.  Here:
//   * CONVERTER waves replace the LDS-DMA loaders of the z-marching kernels: they load the raw f16 pair of the producing conv,
//     form act(a x + b) with the (n, c) coefficients of the finalized statistics, split it into the f16 operand and the two e4m3
//     correction copies, and write the ring planes.  The normalised tensor never exists in HBM and its pass disappears;
//   * the weights stay in REGISTERS for the whole march, which 2 x (28 f16 + 14 fp8) fragments per cout tile only allow when the
//     f16 product and the fp8 correction products of a tile live in DIFFERENT waves: 4 "main" waves (Wh * xh, v_mfma_f32_16x16x32_f16)
//     and 4 "mx" waves (Wh8 * xl8 + Wl8 * xh8, v_mfma_scale_f32_16x16x128_f8f6f4) -- the same 448 matrix-pipe cycles per tile
//     each -- one pair per SIMD.  The mx wave hands its partial accumulators to its main wave through a 4 KiB LDS mailbox;
//   * the main wave adds them, the bias, accumulates the InstanceNorm statistics of ITS output (shifted sums, one slot per wave,
//     fixed-order fold later: deterministic, no atomics) and stores the raw f16 pair, row-planar like every f16x2mx tensor.
// Tile: 2 output planes x 2 rows x 32 x per step (one 2 x 2 x 16 block per consumer wave), marched along z through a ring of
// R input planes; 12 waves per workgroup, one workgroup per CU.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "amx_device.h"

namespace amx {

// Tile (y, x): 2 x 32 (two 16-voxel column tiles side by side) or 4 x 16 (two row pairs on top of each other) -- the second form halos
// 6 x 18 instead of 4 x 34 voxels per plane for the same 64 outputs (1.69 instead of 2.13 converted voxels per output voxel, two
// converter passes per plane instead of three).  HALF = byte offset of a wave's half tile inside a plane.
// NBOX: mailbox sets (2: an mx wave may run one step ahead of the main waves that take its partial sums).
template <int TY_, int TX_, int R_ = 6, int NBOX_ = 1>
struct ZxCfgT {
  static constexpr int TY = TY_, TX = TX_, TZ = 2, R = R_, NBOX = NBOX_;
  static_assert((TY == 2 && TX == 32) || (TY == 4 && TX == 16), "two half tiles of 2 rows x 16 voxels");
  static constexpr int NCV = 4, NMAIN = 4, NMX = 4, NC = NMAIN + NMX;     // converter / consumer waves
  static constexpr int HY = TY + 2, HX = TX + 2, HVP = HY * HX;           // halo voxels of one z-plane
  static constexpr int PPL = ((HVP * 16 + 255) / 256) * 256;              // one 16-byte-per-voxel plane
  static constexpr int X8OFF = 4 * PPL;                                   // hi: 4 planes of 8 channels; x8: [chunk][xl8 | xh8]
  static constexpr int PLSZ = 8 * PPL;
  static constexpr int FLAGOFF = R * PLSZ;                                // ready[4] +0, done[8] +32, mxdone[4] +64, mainfree[4] +96
  static constexpr int XCHOFF = FLAGOFF + 128;                            // mailboxes [mx wave 4][x half 2][tile 4][1 KiB]
  static constexpr int LDS_BYTES = XCHOFF + NBOX * 4 * 8192;
  static constexpr int NJ = (HVP + 63) / 64;                              // converter passes per plane
  static constexpr int HALF = TX == 32 ? 256 : 2 * HX * 16;
  static_assert(LDS_BYTES <= 160 * 1024, "ring + mailboxes must fit the LDS");
};
typedef ZxCfgT<2, 32> ZxCfg;

typedef __attribute__((ext_vector_type(4))) int i32x4;

// Timing ablations (tools/zx_ablate.sh, env AMX_ZX_DBG) exist only in a build with -DAMX_ZX_ABLATE: even as never-taken run-time branches
// they cost (two further switches of this kind slowed the kernel by 10 %: conditional loads are not scheduled ahead).
#ifdef AMX_ZX_ABLATE
#define ZX_DBG(bits) (p.dbg & (bits))
#else
#define ZX_DBG(bits) false
#endif

// Parameters beyond ConvParams: the pending norm of the input (null: the input is stored already activated)
struct ZxExtra {
  const float* in_ab;     // [N][C0][2] (a, b): y = act(a x + b)
  int in_act;
  float in_slope;
  const char* wx;         // fp8 fragments of pack_weights_zx_kernel
};

template <typename C>
__global__ __launch_bounds__((ZxCfg::NC + ZxCfg::NCV) * 64) void conv3d_k3_zx_kernel(const ConvParams p, const ZxExtra e) {
  constexpr int HX = C::HX, PPL = C::PPL, PLSZ = C::PLSZ, R = C::R, TZ = C::TZ, NJ = C::NJ;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- work item: (n, y tile, x tile); x fastest so that XCD neighbours share halos in L2
  int b = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) b = (b & 7) * (nb >> 3) + (b >> 3);
  const int bx = b % p.nbx;
  b /= p.nbx;
  const int by = b % p.nby;
  const int n = b / p.nby;
  const int y0 = by * C::TY, x0 = bx * C::TX;
  const int nplanes = p.D + 2;                              // input planes q = 0 .. D+1 <-> z = q - 1
  const int nsteps = p.D / TZ;

  int* ready = (int*)(smem + C::FLAGOFF);
  int* done = (int*)(smem + C::FLAGOFF + 32);
  int* mxdone = (int*)(smem + C::FLAGOFF + 64);
  int* mainfree = (int*)(smem + C::FLAGOFF + 96);
  if (tid < 32) ((int*)(smem + C::FLAGOFF))[tid] = 0;
  __syncthreads();

  if (wave >= C::NC) {
    // =========================== converter wave: the 8-channel group cw of every input plane ===========================
    const int cw = wave - C::NC;
    const int cs0 = p.cs0 ? p.cs0 : 32;
    const long long lo_off = (long long)(p.C0 >> 4) * cs0;                // the lo planes follow the hi planes of the row
    // per-lane source offsets of halo voxel hv = 64 j + lane (fixed for the march) and its slot in a ring plane
    int off[NJ], slot[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int hv = j * 64 + lane;
      const int hvc = hv < C::HVP ? hv : C::HVP - 1;
      const int hy = hvc / HX, hx = hvc - hy * HX;
      off[j] = reflect_clamp(y0 + hy - 1, p.H) * (int)p.s0y + reflect_clamp(x0 + hx - 1, p.W) * (int)p.s0x + (cw >> 1) * cs0 + (cw & 1) * 16;
      slot[j] = hv < C::HVP ? hv * 16 : -1;
    }
    float ca[8], cb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ca[k] = e.in_ab ? e.in_ab[((long long)n * p.C0 + cw * 8 + k) * 2] : 1.f;
      cb[k] = e.in_ab ? e.in_ab[((long long)n * p.C0 + cw * 8 + k) * 2 + 1] : 0.f;
    }
    const float ak = act_k(e.in_ab ? e.in_act : ACT_NONE, e.in_slope);
    const char* src_n = p.src0 + (long long)n * p.s0n;
    const unsigned a_done = lds_addr(done);
    constexpr int PF = 1;                                    // planes in flight ahead of the one being converted
    uint4 rh[PF + 1][NJ], rl[PF + 1][NJ];
    float vmax = 0.f;                                        // largest |normalised value| this wave turned into an f16 hi part
    auto load_plane = [&](int q, int set) {                  // unconditional (clamped) loads: PF planes ahead in registers
      const int qq = q < nplanes ? q : nplanes - 1;
      const char* plane = src_n + (long long)reflect_clamp(qq - 1, p.D) * p.s0z;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        rh[set][j] = *(const uint4*)(plane + off[j]);
        rl[set][j] = *(const uint4*)(plane + off[j] + lo_off);
      }
    };
    auto convert_plane = [&](int q, int set) {
      char* dstp = smem + (q % R) * PLSZ;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const unsigned wh[4] = {rh[set][j].x, rh[set][j].y, rh[set][j].z, rh[set][j].w};
        const unsigned wl[4] = {rl[set][j].x, rl[set][j].y, rl[set][j].z, rl[set][j].w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float x = (float)__builtin_bit_cast(f16, (unsigned short)(wh[k >> 1] >> ((k & 1) * 16))) +
                          (float)__builtin_bit_cast(f16, (unsigned short)(wl[k >> 1] >> ((k & 1) * 16)));
          v[k] = act_fwd(x * ca[k] + cb[k], ak);
          vmax = __builtin_fmaxf(vmax, __builtin_fabsf(v[k]));
        }
        unsigned hp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) hp[k] = (unsigned)to_bits<f16>(v[2 * k]) | ((unsigned)to_bits<f16>(v[2 * k + 1]) << 16);
        uint2 xl8, xh8;
        mx_split8_direct(v, xl8, xh8);
        if (slot[j] >= 0) {
          *(uint4*)(dstp + cw * PPL + slot[j]) = make_uint4(hp[0], hp[1], hp[2], hp[3]);
          char* x8 = dstp + C::X8OFF + (cw >> 1) * 2 * PPL + slot[j] + (cw & 1) * 8;       // chunk cw >> 1: [xl8 plane | xh8 plane]
          *(uint2*)x8 = xl8;
          *(uint2*)(x8 + PPL) = xh8;
        }
      }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_plane(i, i);
    for (int q = 0; q < nplanes; q += PF + 1) {
#pragma unroll
      for (int h = 0; h < PF + 1; ++h) {
        const int qq = q + h;
        load_plane(qq + PF, (h + PF) % (PF + 1));
        if (qq < nplanes) {
          // ring slot qq % R is free once every consumer is done with plane qq - R: planes < TZ * min(done) are dead
          while (qq >= R + TZ * __builtin_amdgcn_readfirstlane(flag_min8_asm(a_done))) __builtin_amdgcn_s_sleep(1);
          if (!ZX_DBG(16) || qq < R) convert_plane(qq, h);      // (AMX_ZX_DBG 16: timing ablation, the ring is only filled once)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS is in-order per CU: the flag lands after the plane's data
          flag_store(ready + cw, qq + 1);
        }
      }
    }
    // the f16 range of the operand the normalisation produced (what the separate apply pass checks before it stores; a NaN input
    // surfaces in the main waves' check of the outputs)
    raise_flag(p.oflow, !(vmax <= 65504.f));
    return;
  }

  // ================================= consumer waves =================================
  // main wave w (0..3): cout tile wq = w >> 1, x half wcx = w & 1, all 14 tap steps of the f16 product.
  // mx wave 4 + m: cout tile wq = m >> 1, tap steps 7 h .. 7 h + 6 (h = m & 1) of the fp8 products for BOTH x halves -- half the
  // resident A fragments (56 VGPRs instead of 112: with all 14 the wave spilled 250 registers inside the 170-register budget of three
  // waves per SIMD) for the same 7 x 8 MFMAs per step and wave.
  const bool is_mx = wave >= C::NMAIN;
  const int li = lane & 15, g = lane >> 4, hi = g >> 1;
  const int vbase = (g & 1) * PPL + li * 16;                 // lane base inside a plane pair; + wcx * 256 for the x half

  auto wait_planes = [&](int s) {
    int need = TZ * s + TZ + 2;
    need = need < nplanes ? need : nplanes;
    static_assert(C::NCV <= 4 && C::FLAGOFF % 16 == 0, "the ready flags are polled with one 16-byte read");
    while (flag_min4<C::NCV>(ready) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };

  if (is_mx) {
    const int mw = wave - C::NMAIN, wq = mw >> 1, mh = mw & 1;
    const int mx_sa = __builtin_amdgcn_readfirstlane(*p.mxs);
    const int sb = 0x7f7f7f7f;
    char* box0 = smem + C::XCHOFF + mw * 8192 + lane * 16;
    auto frag = [&](int addr) -> i32x8 {                    // chunk 0 plane | chunk 1 plane (2 PPL further on), same voxel
      const i32x4 a = *(const i32x4*)(smem + addr);
      const i32x4 bq = *(const i32x4*)(smem + addr + 2 * PPL);
      return __builtin_shufflevector(a, bq, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // A fragments: [step 14][q][half r][lane][16 B]; lane group g: kind g & 1 (0: Wh8, meets xl8; 1: Wl8, meets xh8), tap A / B of the
    // step by g >> 1; the 32 K bytes of a lane = chunk 0 (16 channels) | chunk 1 -- the two 16-byte halves r
    auto body = [&](auto HSEL) {
      constexpr int S0 = decltype(HSEL)::value * 7;          // this wave's steps S0 .. S0 + 6
      i32x8 wreg[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const i32x4 a = *(const i32x4*)(e.wx + (((S0 + i) * 2 + wq) * 2 + 0) * 1024 + lane * 16);
        const i32x4 bq = *(const i32x4*)(e.wx + (((S0 + i) * 2 + wq) * 2 + 1) * 1024 + lane * 16);
        wreg[i] = __builtin_shufflevector(a, bq, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      // ... and the f16 A fragments of tap steps 12 / 13 (the (., 2, 2) taps) of channel half mh: those 2 x 8 MFMAs per x half run here
      // instead of in the main waves, whose sweep + epilogue chain is the longer one (ablations: mx sweeps -119 us, main sweeps -254)
      typedef Ops<f16>::vec8 hvec8;
      const hvec8 wf12 = *(const hvec8*)(p.wpk + ((mh * kSteps + 12) * 2 + wq) * 1024 + lane * 16);
      const hvec8 wf13 = *(const hvec8*)(p.wpk + ((mh * kSteps + 13) * 2 + wq) * 1024 + lane * 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      auto mine = [](const int step) { return step >= S0 && step < S0 + 7; };
      for (int s = 0; s < nsteps; ++s) {
        wait_planes(s);
        // the mailbox (set s % NBOX) is free once BOTH main waves of this cout tile have taken step s - NBOX
        char* box = box0 + (C::NBOX == 2 ? (s & 1) * 32768 : 0);
        while (flag_min2(mainfree + wq * 2) < s + 1 - C::NBOX) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        auto mm = [&](const int step, const i32x8& f, f32x4 a) {
          return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wreg[step - S0], f, a, 0, 0, 0, mx_sa, 0, sb);
        };
        // the two x halves as a RUN-TIME loop: unrolled, hipcc kept the whole step's fragments and addresses alive (447 spilled
        // registers inside the 168 of three waves per SIMD; 97 registers, none spilled, this way)
#pragma unroll 1
        for (int xh = 0; xh < (ZX_DBG(2 | 32) ? 0 : 2); ++xh) {   // (AMX_ZX_DBG 2 / 32: timing ablation without the sweeps)
          const int lanebase = vbase + xh * C::HALF;
          f32x4 acc[2][2];
#pragma unroll
          for (int tz = 0; tz < 2; ++tz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) acc[tz][cy] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int pl = 0; pl < 4; ++pl) {
            const int sl = ((s * TZ + pl) % R) * PLSZ + C::X8OFF;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                       // halo row r serves output row cy with tap row ky = r - cy
              bool used = false;
#pragma unroll
              for (int tz = 0; tz < 2; ++tz)
#pragma unroll
                for (int cy = 0; cy < 2; ++cy) used |= (pl - tz >= 0 && pl - tz <= 2 && r - cy >= 0 && r - cy <= 2 && mine((pl - tz) * 3 + r - cy));
              if (!used) continue;
              const i32x8 f = frag(lanebase + hi * 16 + sl + (r * HX) * 16);
#pragma unroll
              for (int tz = 0; tz < 2; ++tz)
#pragma unroll
                for (int cy = 0; cy < 2; ++cy) {
                  const int kz = pl - tz, ky = r - cy;
                  if (kz < 0 || kz > 2 || ky < 0 || ky > 2 || !mine(kz * 3 + ky)) continue;
                  acc[tz][cy] = mm(kz * 3 + ky, f, acc[tz][cy]);
                }
              __builtin_amdgcn_sched_barrier(0);                // keep the next fragment's reads behind this one's MFMAs: hoisted,
            }                                                   // the fully unrolled step held dozens of 8-register fragments (470 spills)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
              bool used = false;
#pragma unroll
              for (int tz = 0; tz < 2; ++tz) used |= (pl - tz >= 0 && pl - tz <= 2 && mine(9 + pl - tz));
              if (!used) continue;
              const i32x8 f = frag(lanebase + hi * 16 * HX + sl + (cy * HX + 2) * 16);
#pragma unroll
              for (int tz = 0; tz < 2; ++tz) {
                const int kz = pl - tz;
                if (kz < 0 || kz > 2 || !mine(9 + kz)) continue;
                acc[tz][cy] = mm(9 + kz, f, acc[tz][cy]);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (mine(12)) {                                       // steps 12, 13 belong to the same (upper) half
#pragma unroll
            for (int tz = 0; tz < 2; ++tz) {
              const int sl0 = ((s * TZ + tz) % R) * PLSZ, sl1 = ((s * TZ + tz + 1) % R) * PLSZ, sl2 = ((s * TZ + tz + 2) % R) * PLSZ;
              const int bz = lanebase + (hi ? sl1 : sl0) + C::X8OFF + (2 * HX + 2) * 16;
              const int b0 = lanebase + sl2 + C::X8OFF + (2 * HX + 2) * 16;
#pragma unroll
              for (int cy = 0; cy < 2; ++cy) {
                acc[tz][cy] = mm(12, frag(bz + (cy * HX) * 16), acc[tz][cy]);
                acc[tz][cy] = mm(13, frag(b0 + (cy * HX) * 16), acc[tz][cy]);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          {
            const int koff = mh * 2 * PPL;                      // hi planes 2 mh, 2 mh + 1
#pragma unroll
            for (int tz = 0; tz < 2; ++tz) {
              const int sl0 = ((s * TZ + tz) % R) * PLSZ, sl1 = ((s * TZ + tz + 1) % R) * PLSZ, sl2 = ((s * TZ + tz + 2) % R) * PLSZ;
              const int bz = lanebase + (hi ? sl1 : sl0) + koff + (2 * HX + 2) * 16;
              const int b0 = lanebase + sl2 + koff + (2 * HX + 2) * 16;
#pragma unroll
              for (int cy = 0; cy < 2; ++cy) {
                acc[tz][cy] = Ops<f16>::mfma(wf12, *(const hvec8*)(smem + bz + (cy * HX) * 16), acc[tz][cy]);
                acc[tz][cy] = Ops<f16>::mfma(wf13, *(const hvec8*)(smem + b0 + (cy * HX) * 16), acc[tz][cy]);
              }
            }
          }
#pragma unroll
          for (int tz = 0; tz < 2; ++tz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) *(f32x4*)(box + (xh * 4 + tz * 2 + cy) * 1024) = acc[tz][cy];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // every ring read of this step has returned
        flag_store(done + wave, s + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flag_store(mxdone + mw, s + 1);
      }
    };
    if (mh == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
    return;
  }

  const int cwv = wave;
  const int wq = cwv >> 1, wcx = cwv & 1;
  const int lanebase = vbase + wcx * C::HALF;
  const int base_d1 = lanebase + hi * 16;
  const int base_dx = lanebase + hi * 16 * HX;
  // ---------------- main wave: Wh * xh of cout tile wq, + the mx partial, bias, statistics, store ----------------
  typedef Ops<f16>::vec8 vec8;
  vec8 wreg[2][12];                                        // tap steps 0 .. 11 (12 / 13: the mx waves)
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int s = 0; s < 12; ++s) wreg[k][s] = *(const vec8*)(p.wpk + ((k * kSteps + s) * 2 + wq) * 1024 + lane * 16);
  const int cbc = g * 8 + wq * 4;                           // lane holds output channels cbc .. cbc + 3 (Q = 2 packing)
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *(const f32x4*)(p.bias + cbc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int ocs = p.ocs ? p.ocs : 32;
  const int yl = y0 + (C::TX == 32 ? 0 : wcx * 2), xl = x0 + (C::TX == 32 ? wcx * 16 : 0) + li;
  char* out_l = p.out + (long long)n * p.on + (long long)yl * p.oy + (long long)xl * p.ox + (long long)(cbc >> 4) * ocs + (cbc & 15) * 2;
  const long long out_lo = (long long)(p.Cout >> 4) * ocs;
  // the network's output conv: fp32 [n][c][z][y][x] -- per channel j of the lane a 64-byte run of the 16 voxel lanes
  float* out32_l = p.out32 ? p.out32 + (long long)n * p.pn + (long long)cbc * p.pc + (long long)yl * p.py + xl : nullptr;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  bool bad = false;

  for (int s = 0; s < nsteps; ++s) {
    wait_planes(s);
    int b1[4], bx3[4];
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      const int sl = ((s * TZ + pl) % R) * PLSZ;
      b1[pl] = base_d1 + sl;
      bx3[pl] = base_dx + sl;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int tz = 0; tz < 2; ++tz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy) acc[tz][cy] = bias;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (ZX_DBG(2 | 64)) break;
      const int koff = k * 2 * PPL;                          // channel planes 2k, 2k + 1
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        vec8 F[4], H[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) F[r] = *(const vec8*)(smem + b1[pl] + koff + (r * HX) * 16);
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) H[cy] = *(const vec8*)(smem + bx3[pl] + koff + (cy * HX + 2) * 16);
#pragma unroll
        for (int tz = 0; tz < 2; ++tz) {
          const int kz = pl - tz;
          if (kz < 0 || kz > 2) continue;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) acc[tz][cy] = Ops<f16>::mfma(wreg[k][kz * 3 + ky], F[cy + ky], acc[tz][cy]);
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) acc[tz][cy] = Ops<f16>::mfma(wreg[k][9 + kz], H[cy], acc[tz][cy]);
        }
      }
      // (tap steps 12 / 13 run in the mx waves)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every ring read of this step has returned
    flag_store(done + wave, s + 1);

    // ---- the correction products: the two mx waves of this cout tile (tap steps 0..6 / 7..13), this wave's x half
    while (flag_min2(mxdone + wq * 2) < s + 1) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        const char* box = smem + C::XCHOFF + (wq * 2 + mh) * 8192 + lane * 16 + wcx * 4096 + (C::NBOX == 2 ? (s & 1) * 32768 : 0);
#pragma unroll
        for (int tz = 0; tz < 2; ++tz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) acc[tz][cy] += *(const f32x4*)(box + (tz * 2 + cy) * 1024);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      flag_store(mainfree + cwv, s + 1);
    }
    // ---- statistics of the values about to be stored (hi + lo = the fp32 value), shifted by the bias; raw store (the norm of THIS
    //      layer is its consumer's business)
#pragma unroll
    for (int tz = 0; tz < 2; ++tz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[tz][cy][j];
          bad |= RangeCheck<f16>::bad(v[j]);
          const float d = v[j] - bias[j];
          s1[j] += d;
          s2[j] += d * d;
        }
        if (ZX_DBG(4)) continue;
        if (out32_l) {
          float* d32 = out32_l + (long long)(s * TZ + tz) * p.pz + cy * p.py;
#pragma unroll
          for (int j = 0; j < 4; ++j) d32[(long long)j * p.pc] = v[j];
          continue;
        }
        char* dst = out_l + (long long)(s * TZ + tz) * p.oz + cy * p.oy;
        *(uint2*)dst = make_uint2((unsigned)to_bits<f16>(v[0]) | ((unsigned)to_bits<f16>(v[1]) << 16),
                                  (unsigned)to_bits<f16>(v[2]) | ((unsigned)to_bits<f16>(v[3]) << 16));
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = v[j] - (float)(f16)v[j];
        *(uint2*)(dst + out_lo) = make_uint2((unsigned)to_bits<f16>(r[0]) | ((unsigned)to_bits<f16>(r[1]) << 16),
                                             (unsigned)to_bits<f16>(r[2]) | ((unsigned)to_bits<f16>(r[3]) << 16));
      }
  }
  if (!p.out32) raise_flag(p.oflow, bad);
  if (p.stats) {
    // sum over the 16 voxel lanes of the lane group (row_shr 1, 2, 4, 8: lane 15 of the row holds it); slot = (tile, x half), this
    // wave's 16 channels of it -- the pair (wq = 0, 1) of an x half fills all 32 channels of the slot
#pragma unroll
    for (int k = 0; k < 4; ++k) {
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
      const long long nslots = (long long)p.nby * p.nbx * 2;
      const long long sl = ((long long)by * p.nbx + bx) * 2 + wcx;
      float* o = p.stats + (((long long)n * nslots + sl) * p.Cout + cbc) * 2;
      *(float4*)o = make_float4(s1[0], s2[0], s1[1], s2[1]);
      *(float4*)(o + 4) = make_float4(s1[2], s2[2], s1[3], s2[3]);
    }
  }
}

// fp8 A fragments of the mx waves: [step 14][q 2][half r 2][lane 64][16 bytes] (56 KiB).  MFMA row m of tile q -> output channel
// (m >> 2) * 8 + q * 4 + (m & 3) (the Q = 2 packing of pack_weights_kernel, which the main waves read); lane group g: kind g & 1
// (0: e4m3(2^Sw Wh), meets xl8; 1: e4m3(2^(Sw+11) Wl), meets xh8), tap = g >> 1 ? tapB(step) : tapA(step); byte e of half r =
// input channel 16 r + e.  Sw: mx_weight_shift of the layer (mxs[1] holds max |w|, written by launch_pack_weights_mx before this).
__device__ __forceinline__ int zx_weight_shift(const int* mxs) {
  const float mx = __builtin_bit_cast(float, (unsigned)mxs[1]);
  if (!(mx > 0.f) || !(mx < 3.0e38f)) return 0;
  int e;
  const float m = __builtin_frexpf(mx, &e);
  int sw = (m > 0.875f ? 8 : 9) - e;
  return sw < -100 ? -100 : (sw > 100 ? 100 : sw);
}
__global__ void pack_weights_zx_kernel(const float* __restrict__ w, const float* __restrict__ scale, unsigned char* __restrict__ wx,
                                       int CoutReal, const int* __restrict__ mxs) {
  const int sw = zx_weight_shift(mxs);
  const float sh = __builtin_ldexpf(1.f, sw), sl = __builtin_ldexpf(1.f, sw + 11);
  const int total = kSteps * 2 * 2 * 64 * 4;                // one thread = 4 bytes
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e4 = idx & 3, lane = (idx >> 2) & 63;
    int r = idx >> 8;
    const int half = r & 1;
    r >>= 1;
    const int q = r & 1, s = r >> 1;
    const int m = lane & 15, g = lane >> 4;
    const int cout = (m >> 2) * 8 + q * 4 + (m & 3);
    const int tap = (g >> 1) ? tapB_index(s) : tapA_index(s);
    float v4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cin = half * 16 + e4 * 4 + k;
      float v = 0.f;
      if (tap >= 0 && cout < CoutReal) {
        v = w[((long long)cout * 32 + cin) * 27 + tap];
        if (scale) v *= scale[cout];
      }
      const float hi = (float)(f16)v;
      v4[k] = (g & 1) ? (v - hi) * sl : hi * sh;
    }
    *(unsigned*)(wx + (((s * 2 + q) * 2 + half) * 64 + lane) * 16 + e4 * 4) = e4m3_pk4(v4[0], v4[1], v4[2], v4[3]);
  }
}

size_t conv_zx_packed_bytes() { return (size_t)kSteps * 2 * 2 * 1024; }

hipError_t launch_pack_weights_zx(const float* w, const float* scale, void* wx, const int* mxs, int CoutReal, hipStream_t st) {
  hipLaunchKernelGGL(pack_weights_zx_kernel, dim3(56), dim3(256), 0, st, w, scale, (unsigned char*)wx, CoutReal, mxs);
  return hipGetLastError();
}

// One full-resolution row-planar f16x2mx segment of 32 channels -> 32 channels, whole tiles; 16-bit row-planar output, or the network's
// fp32 planar output (no importance map, no activation of its own)
bool conv_zx_eligible(const ConvParams& p) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_ZX") ? 1 : 0;
  const bool out_ok = p.out32 ? (!p.wmap && p.act == ACT_NONE && !p.stats) : (p.out && p.ox == 32);
  return !off && !p.src0_f32c1 && p.C0 == 32 && p.C1 == 0 && p.Cout == 32 && out_ok && p.mxs &&
         ((p.W % 32 == 0 && p.H % 2 == 0) || (p.W % 16 == 0 && p.H % 4 == 0)) && p.D % 2 == 0 && p.D >= 4 && p.s0x == 32;
}
int conv_zx_stats_slots(int H, int W) { return H * W / 32; }       // one per (tile, half tile), whichever tile shape runs

static thread_local char g_kernel_name_zx[64] = "";
const char* last_conv_zx_kernel_name() { return g_kernel_name_zx; }

template <typename C>
static hipError_t launch_conv_zx_t(ConvParams p, const float* in_ab, int in_act, float in_slope, const void* wx, hipStream_t st) {
  snprintf(g_kernel_name_zx, sizeof g_kernel_name_zx, "conv3d_k3_zx<f16x2mx,32->32,%dx%dx%d,m4+x4+cv4,r%d%s%s>", C::TZ, C::TY, C::TX, C::R,
           in_ab ? ",norm-in" : "", p.out32 ? ",o1" : "");
  static amx::DeviceOnce attr_once;
  auto kern = conv3d_k3_zx_kernel<C>;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_once.set();
  }
  static int dbg = -1;
  if (dbg < 0) dbg = exp_env("AMX_ZX_DBG") ? atoi(exp_env("AMX_ZX_DBG")) : 0;
  p.dbg = dbg;
  p.nby = p.H / C::TY;
  p.nbx = p.W / C::TX;
  ZxExtra e;
  e.in_ab = in_ab; e.in_act = in_act; e.in_slope = in_slope; e.wx = (const char*)wx;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nby * p.nbx * p.N)), dim3((C::NC + C::NCV) * 64), C::LDS_BYTES, st, p, e);
  return hipGetLastError();
}

hipError_t launch_conv_zx(ConvParams p, const float* in_ab, int in_act, float in_slope, const void* wx, hipStream_t st) {
  static int tile = -1;                 // AMX_ZX_TILE = 0: 2x32 tiles wherever they fit, 1 (default): 4x16 tiles wherever they fit
  if (tile < 0) tile = exp_env("AMX_ZX_TILE") ? atoi(exp_env("AMX_ZX_TILE")) : 1;
  const bool wide_ok = p.W % 32 == 0 && p.H % 2 == 0, tall_ok = p.W % 16 == 0 && p.H % 4 == 0;
  // (a ring of 8 planes instead of 6 for the 4x16 tiles: 1016 / 1067 -> 1015 / 1037 us, inside the noise -- not instantiated)
  // (two mailbox sets -- NBOX = 2, an mx wave one step ahead of its main waves: 1030 / 1014 -> 1036 / 1015 us, nothing -- not instantiated)
  if (tall_ok && (tile == 1 || !wide_ok)) return launch_conv_zx_t<ZxCfgT<4, 16>>(p, in_ab, in_act, in_slope, wx, st);
  return launch_conv_zx_t<ZxCfgT<2, 32>>(p, in_ab, in_act, in_slope, wx, st);
}

}  // namespace amx
