// anatomix_amd -- conv3d 3x3x3 reflect, second-generation kernel: persistent workgroups,
// LDS-DMA halo gather, double-buffered stages, deferred output stores.
//
// Same arithmetic formulation as amx_conv3d.hip (weights = MFMA A operand, activations = B operand,
// plane-major LDS halo image, 14 paired-tap steps per 16 input channels).  What changes is the
// data movement and the per-item overhead, which is what bounded the first kernel:
//   * every workgroup is persistent over a contiguous run of (brick, cout-group) items and over
//     the 16*NCH-channel stages of each item; the flattened (item, stage) sequence is software
//     pipelined with two LDS buffers -- while stage t is multiplied out of buffer t&1, the halo
//     rows (+ packed weights + bias) of stage t+1 stream into the other buffer;
//   * the gather uses LDS-DMA (global_load_lds_dwordx4): one instruction moves one halo row of one
//     8-channel plane (<= 64 voxels x 16 B) HBM/L2 -> LDS with no VGPR round trip.  The row's
//     (z, y) reflect/upsample/concat address is wave-uniform SCALAR math; only the x offset is
//     per lane and it is computed once per stage;
//   * when a layer has a single stage and a single cout group (all 16->16 layers) the packed
//     weights and the bias are loaded once per workgroup and stay resident;
//   * one barrier per stage: [s_waitcnt vmcnt(0) ; s_barrier] -> issue DMA(t+1) -> store the
//     PREVIOUS item's packed outputs -> MFMA sweep(t).  The stores therefore drain under the sweep
//     instead of in front of the next barrier;
//   * item coordinates advance by increment-and-carry on the scalar unit (no integer divisions
//     in the loop); accumulators start from the bias, so the epilogue is activation + pack.
#include <stdio.h>
#include <stdlib.h>

#include "amx_device.h"

namespace amx {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T, int WZ, int WY, int WX, int NWZ, int NWY, int Q, int NCH, int OUTMODE, int NLW = 0, int NBUF = 2>
struct Conv2Cfg {
  static constexpr int NW = NWZ * NWY;
  static constexpr int TZ = WZ * NWZ, TY = WY * NWY, TX = WX;
  static constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  static constexpr int PLANE = ((HV * 16 + 255) / 256) * 256;
  static constexpr int HALO = 2 * PLANE;              // bytes per 16-channel sub-chunk
  static constexpr int WSUB = kSteps * Q * 1024;      // packed weight bytes per sub-chunk
  static constexpr int WOFF = NCH * HALO;
  static constexpr int BIASOFF = WOFF + NCH * WSUB;   // 16*Q fp32 (256 B reserved)
  static constexpr int BUF = BIASOFF + 256;           // one pipeline buffer
  static constexpr int FLAGOFF = NBUF * BUF;           // loader-wave mode: ready[NLW] at +0, done[8] at +32
  static constexpr int LDS_BYTES = NBUF * BUF + (NLW ? 64 : 0);
  static constexpr int LX = WX >= 16 ? 16 : 8;
  static constexpr int LY = 16 / LX;
  static constexpr int XT = WX / LX;
  static constexpr int YT = WY / LY;
  static constexpr int CTW = WZ * YT * XT;
  // low-resolution staging of the upsampled segment (one row per output voxel row pair): instead of
  // gathering the x2-upsampled halo (every low-res voxel fetched up to 8 times) the stage holds the
  // low-res halo itself and the B-fragment addresses divide by two.  Needs 16-voxel column tiles
  // in one row and one 16-channel sub-chunk per stage.
  static constexpr bool LOWUP = WX >= 16 && NCH == 1 && WZ == 1 && (WY % 2) == 0;
  static constexpr int LZH = TZ / 2 + 2, LYH = TY / 2 + 2, LXH = TX / 2 + 2;   // low-res halo extent
  static constexpr int RRL = 64 / LXH;                // low-res halo rows per DMA instruction
  static constexpr int NROWL = LZH * LYH;
  static constexpr int NINSTL = (NROWL + RRL - 1) / RRL;
  static constexpr int RR = 64 / HX;                  // halo rows per DMA instruction
  static constexpr int NROW = HZ * HY;
  static constexpr int NINST = (NROW + RR - 1) / RR;
  static_assert(HX <= 64, "halo row must fit one wave");
  static_assert(WY % LY == 0 && WX % LX == 0, "wave sub-brick must tile into 16-voxel columns");
  static_assert(LDS_BYTES <= 160 * 1024, "the pipeline buffers must fit the 160 KiB LDS");
  static_assert(NLW == 0 ? NBUF == 2 : (NBUF >= 2 && NLW <= 4 && NWZ * NWY <= 8), "pipeline shape");
};

struct ItemCoord {   // all members wave-uniform
  int bx, by, bz, n, cg;
};

__device__ __forceinline__ void advance_item(ItemCoord& c, const ConvParams& p) {
  if (++c.bx < p.nbx) return;
  c.bx = 0;
  if (++c.by < p.nby) return;
  c.by = 0;
  if (++c.bz < p.nbz) return;
  c.bz = 0;
  if (++c.n < p.N) return;
  c.n = 0;
  ++c.cg;
}

// NLW > 0: LOADER-WAVE mode.  NLW extra waves issue every LDS-DMA of the (item, stage) sequence into a ring of NBUF
// buffers and the MFMA waves only multiply and store; the two sides meet through LDS counters (no workgroup barrier).
// Reason (cycle trace of 128 -> 128 @16^3, batch 4): per stage a MFMA wave spent ~2050 cycles ISSUING its share of the
// next stage's DMA (a VMEM instruction blocks its wave while the memory queue is full) next to ~2250 cycles of MFMAs.
//
// SPLIT (strict precision): every activation tensor holds 2*C 16-bit channels per voxel, [hi(C) | lo(C)] with value = hi + lo
// (lo = the rounding residual of hi), and the packed weights hold [Wh | Wl] the same way.  The conv then runs three times the
// stages -- Wh*xh, Wh*xl, Wl*xh, the dropped Wl*xl term is 2^-16 (bf16) / 2^-22 (f16) relative -- into the same fp32
// accumulators, and the epilogue splits the fp32 result into hi / lo again.  Everything else (halo staging, LDS-DMA, tap
// steps) is unchanged: a stage is still 16 channels of one tensor behind one base pointer.
//
// MX (AMX_PREC_F16X2_MX; implies SPLIT storage of the output, T = f16): TWO stages per 16-channel chunk instead of three.  Part 0 is
// Wh * xh as above; part 1 stages the chunk's 32 bytes of e4m3 copies per voxel -- [xl8(16) | xh8(16)], the same halo bytes and the
// same two LDS planes as a 16-bit stage -- and sweeps them in 7 steps of v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128 = 4 taps x 16
// channels x {Wh8 * xl8, Wl8 * xh8}; 32 cycles, twice the f16 rate) against the fp8 fragments of pack_weights_mx_kernel with one
// uniform block scale.  Lane group g: plane g & 1 (xl8 / xh8), taps 4 j + 2 (g >> 1) + {0, 1} = the two 16-byte reads of a fragment.
template <typename T, int WZ, int WY, int WX, int NWZ, int NWY, int Q, int NCH, int OUTMODE, int NLW, int NBUF, bool SPLIT, bool MX = false>
__global__ __launch_bounds__((NWZ * NWY + NLW) * 64) void conv3d_k3_v2_kernel(const ConvParams p) {
  static_assert(!MX || (SPLIT && __is_same(T, f16)), "the fp8 correction stages extend the f16 hi / lo storage");
  constexpr bool LOWUP = Conv2Cfg<T, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE, NLW, NBUF>::LOWUP && !MX;
  typedef Conv2Cfg<T, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE, NLW, NBUF> C;
  typedef typename Ops<T>::vec8 vec8;
  constexpr int HY = C::HY, HX = C::HX, PLANE = C::PLANE, HALO = C::HALO, NW = C::NW;
  constexpr int WOFF = C::WOFF, CTW = C::CTW, LX = C::LX, LY = C::LY, XT = C::XT, YT = C::YT;
  constexpr int NPEND = OUTMODE == 0 ? (SPLIT ? 4 * Q : 2 * Q) : 4 * Q;   // 32-bit words kept per column tile until the store

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;

  // ---- this workgroup's contiguous run of items; XCD b%8 gets a contiguous span of runs
  const int nbricks = p.nbz * p.nby * p.nbx * p.N;
  const int ncg = p.Cout / (16 * Q);
  const long long items = (long long)nbricks * ncg;
  const int G = gridDim.x;
  int jw = blockIdx.x;
  if ((G & 7) == 0) jw = (jw & 7) * (G >> 3) + (jw >> 3);
  const int it0 = (int)(items * jw / G), it1 = (int)(items * (jw + 1) / G);
  const int nchunk = (p.C0 + p.C1) >> 4;                  // 16-channel chunks of the LOGICAL input
  const int nstage = (MX ? 2 : (SPLIT ? 3 : 1)) * nchunk / NCH;
  const int mx_sa = MX ? __builtin_amdgcn_readfirstlane(*p.mxs) : 0;    // E8M0 block scale of the fp8 weights (x4 bytes)
  const int T_total = (it1 - it0) * nstage;
  if (T_total <= 0) return;
  const bool resident = NLW == 0 && nstage == 1 && ncg == 1;     // weights + bias loaded once per workgroup
  // DMA work is split over the issuing waves: all NW waves (classic) or the NLW loader waves
  const int iw = NLW ? wave - NW : wave;
  constexpr int INW = NLW ? NLW : NW;
  int dma_count = 0;                                   // DMA instructions this wave issued in the current issue() call

  // ---- lane-constant LDS read bases (relative to the pipeline buffer)
  const int wz = wave / NWY, wy = wave % NWY;
  const int dy = (LX == 16) ? 0 : (li >> 3);
  const int dx = (LX == 16) ? li : (li & 7);
  const int lanehv = ((wz * WZ) * HY + wy * WY + dy) * HX + dx;
  const int lanebase = (g & 1) * PLANE + lanehv * 16;
  const int hi = g >> 1;
  const int base_d1 = lanebase + hi * 16;
  const int base_dx = lanebase + hi * 16 * HX;
  const int base_dz = lanebase + hi * 16 * HX * HY;
  const int base_d0 = lanebase;

  // ---- lane-constant LDS read bases for LOW-RES staged (upsampled) stages.  Output voxel (z,y,x) of the
  //      brick reads low-res halo voxel ((z+kz-1)>>1, (y+ky-1)>>1, (x+kx-1)>>1) + 1 per axis.
  int U1[3], U2e[3], U2o[3], Uz = 0;
  if (LOWUP) {
    constexpr int LXH = C::LXH, LYH = C::LYH;
    const int Lx0 = ((li - 1) >> 1) + 1, Lx1 = (li >> 1) + 1, Lx2 = ((li + 1) >> 1) + 1;
    const int ub = (g & 1) * PLANE + ((wy * WY / 2) * LXH) * 16;
    int zrow[3];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) zrow[kz] = ((((wz * WZ + kz - 1) >> 1) + 1) * LYH * LXH) * 16;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      U1[kz] = ub + zrow[kz] + (hi ? Lx1 : Lx0) * 16;           // steps 0..8: (kz,ky,0) | (kz,ky,1)
      U2o[kz] = ub + zrow[kz] + Lx2 * 16;                        // kx = 2, both halves in the same low row
      U2e[kz] = U2o[kz] + hi * LXH * 16;                         // kx = 2, (ky=0 | ky=1) one low row apart (even cy)
    }
    Uz = ub + (hi ? zrow[1] : zrow[0]) + Lx2 * 16;               // step 12: (0,2,2) | (1,2,2)
  }

  // ---- DMA lane constants: lane -> (row within instruction, x within halo row)
  const int dl = lane / HX, hxl = lane - dl * HX;
  const bool dma_lane = dl < C::RR;
  // packed mode (RR == 1): instruction j covers halo voxels 64 j .. 64 j + 63; this wave issues j = iw, iw + INW, ...
  constexpr int NPK = (C::HV + 63) / 64, PKS = C::RR == 1 ? (NPK + INW - 1) / INW : 1;
  int pk_pos[PKS], pk_off0[PKS], pk_off1[PKS];           // hz | hy << 8 | hx << 16 (or -1 past the halo); byte offsets per item
  // 32-bit offsets inside one sample: every full-resolution segment must be smaller than 2 GiB
  const bool pk_ok = C::RR == 1 && !(p.dbg & 256) && (long long)p.D * (p.s0z > p.s1z ? p.s0z : p.s1z) < (1ll << 31);
#pragma unroll
  for (int m = 0; m < PKS; ++m) {
    const int hv = (iw + m * INW) * 64 + lane;
    const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX;
    pk_pos[m] = hv < C::HV ? (hz | (hy << 8) | ((rem - hy * HX) << 16)) : -1;
    pk_off0[m] = pk_off1[m] = 0;
  }

  ItemCoord nx;   // next item to issue DMA for (one-time decode; afterwards increment-and-carry)
  {
    nx.cg = it0 / nbricks;
    int b = it0 - nx.cg * nbricks;
    nx.bx = b % p.nbx;
    b /= p.nbx;
    nx.by = b % p.nby;
    b /= p.nby;
    nx.bz = b % p.nbz;
    nx.n = b / p.nbz;
  }
  ItemCoord cu = nx;   // item being multiplied
  int nx_stage = 0, cu_stage = 0;

  // ---- issue the LDS-DMA of one (item, stage) into pipeline buffer `bsel01` (inline asm, see dma16_asm)
#define AMX_DMA16(src, dst) dma16_asm((const void*)(src), (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(dst)))
  auto issue = [&](const ItemCoord& it, int stage, int bsel01, bool with_weights) {
    char* buf = smem + bsel01 * C::BUF;
    const int z0 = it.bz * C::TZ, y0 = it.by * C::TY, x0 = it.bx * C::TX;
    const int gx = reflect_clamp(x0 + hxl - 1, p.W);
    if (C::RR == 1 && pk_ok && stage == 0) {
#pragma unroll
      for (int m = 0; m < PKS; ++m) {
        const int pos = pk_pos[m];
        const int gz = reflect_clamp(z0 + (pos & 255) - 1, p.D), gy = reflect_clamp(y0 + ((pos >> 8) & 255) - 1, p.H);
        const int gxx = reflect_clamp(x0 + ((pos >> 16) & 255) - 1, p.W);
        pk_off0[m] = gz * (int)p.s0z + gy * (int)p.s0y + gxx * (int)p.s0x;
        pk_off1[m] = gz * (int)p.s1z + gy * (int)p.s1y + gxx * (int)p.s1x;
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int vch = stage * NCH + k;                      // wave-uniform; SPLIT: virtual chunk in [0, 3 * nchunk)
      const int part = SPLIT ? vch / nchunk : 0;            // 0: Wh * xh   1: Wh * xl   2: Wl * xh   (MX: 1 = both, in fp8)
      const int ch = (vch - part * nchunk) << 4;
      const bool second = ch >= p.C0;
      const int sh = second ? p.up_shift : 0;
      const int cseg = second ? p.C1 : p.C0, chs = second ? ch - p.C0 : ch;
      // a voxel's 32-byte pieces: hi chunks, then the lo chunks, then (MX) the chunks of e4m3 copies; `cs` bytes apart
      const int sec = part == 1 ? (MX ? 2 : 1) : 0;
      const int cs = second ? (p.cs1 ? p.cs1 : 32) : (p.cs0 ? p.cs0 : 32);
      const long long coff = (long long)(sec * (cseg >> 4) + (chs >> 4)) * cs;
      const char* base = (second ? p.src1 + (long long)it.n * p.s1n : p.src0 + (long long)it.n * p.s0n) + coff;
      const long long sz = second ? p.s1z : p.s0z, sy = second ? p.s1y : p.s0y;
      const int xoff = (gx >> sh) * (int)(second ? p.s1x : p.s0x);
      if (LOWUP && second && p.up_shift) {
        // low-res halo of the upsampled segment: reflect padding at high resolution == replicate (clamp)
        // at low resolution; brick origins are even.
        const int LD = p.D >> 1, LH = p.H >> 1, LW = p.W >> 1;
        const int dl2 = lane / C::LXH, hx2 = lane - dl2 * C::LXH;
        int gx2 = (x0 >> 1) - 1 + hx2;
        gx2 = gx2 < 0 ? 0 : (gx2 >= LW ? LW - 1 : gx2);
        for (int j = iw; j < C::NINSTL; j += INW) {
          const int r = j * C::RRL + dl2;
          const int lz = r / C::LYH, ly = r - lz * C::LYH;
          int gz = (z0 >> 1) - 1 + lz, gy = (y0 >> 1) - 1 + ly;
          gz = gz < 0 ? 0 : (gz >= LD ? LD - 1 : gz);
          gy = gy < 0 ? 0 : (gy >= LH ? LH - 1 : gy);
          const char* src = base + gz * sz + gy * sy + gx2 * (int)p.s1x;
          char* dst = buf + k * HALO + j * (C::RRL * C::LXH * 16);   // uniform
          dma_count += 2;
          if (dl2 < C::RRL && r < C::NROWL) {
            AMX_DMA16(src, dst);
            AMX_DMA16(src + 16, dst + PLANE);
          }
        }
      } else if (C::RR == 1 && sh == 0 && pk_ok) {
        // Halo rows wider than half a wave (HX = 34): one instruction per ROW uses 34 of 64 lanes and lands at unaligned LDS
        // addresses (22 % of the LDS cycles of these kernels were bank conflicts).  The halo image is linear, so an instruction
        // fills 64 consecutive halo voxels = one aligned KiB instead; every lane has its own source voxel, whose byte offset
        // inside the sample was computed once per item (pk_off).
        const int* offs = second ? pk_off1 : pk_off0;
#pragma unroll
        for (int m = 0; m < PKS; ++m) {
          const int j = iw + m * INW;                          // uniform
          if (j >= NPK) break;
          const char* src = base + offs[m];
          char* dst = buf + k * HALO + j * 1024;               // uniform
          dma_count += 2;
          if (pk_pos[m] >= 0) {
            AMX_DMA16(src, dst);
            AMX_DMA16(src + 16, dst + PLANE);
          }
        }
      } else if (C::RR == 1) {
        // one halo row per instruction: (z, y) addressing stays on the scalar unit
        for (int j = iw; j < C::NROW; j += INW) {
          const int hz = j / HY, hy = j - hz * HY;
          const int gz = reflect_clamp(z0 + hz - 1, p.D) >> sh;
          const int gy = reflect_clamp(y0 + hy - 1, p.H) >> sh;
          const char* row = base + gz * sz + gy * sy;         // uniform
          char* dst = buf + k * HALO + j * (HX * 16);          // uniform LDS row base
          dma_count += 2;
          if (dma_lane) {
            AMX_DMA16(row + xoff, dst);
            AMX_DMA16(row + xoff + 16, dst + PLANE);
          }
        }
      } else {
        for (int j = iw; j < C::NINST; j += INW) {
          const int r = j * C::RR + dl;
          const int hz = r / HY, hy = r - hz * HY;
          const int gz = reflect_clamp(z0 + hz - 1, p.D) >> sh;
          const int gy = reflect_clamp(y0 + hy - 1, p.H) >> sh;
          const char* src = base + gz * sz + gy * sy + xoff;
          char* dst = buf + k * HALO + j * (C::RR * HX * 16);   // uniform
          dma_count += 2;
          if (dma_lane && r < C::NROW) {
            AMX_DMA16(src, dst);
            AMX_DMA16(src + 16, dst + PLANE);
          }
        }
      }
    }
    if (with_weights) {
      // packed weights of these sub-chunks: linear copy, 1 KiB per instruction
      // SPLIT: packed as [cg][Wh chunks | Wl chunks]; parts 0 and 1 read Wh, part 2 reads Wl
      const int vch0 = stage * NCH, part0 = SPLIT ? vch0 / nchunk : 0;
      const long long wchunk = SPLIT ? (long long)it.cg * 2 * nchunk + (vch0 - part0 * nchunk) + (part0 == (MX ? 1 : 2) ? nchunk : 0)
                                     : (long long)it.cg * nchunk + vch0;
      const char* ws = p.wpk + wchunk * C::WSUB;
      constexpr int NWI = NCH * C::WSUB / 1024;
      for (int j = iw; j < NWI; j += INW) {
        ++dma_count;
        AMX_DMA16(ws + j * 1024 + lane * 16, buf + WOFF + j * 1024);
      }
      if (iw == INW - 1 && stage == 0 && p.bias) {     // bias rides with the item's first stage
        ++dma_count;
        if (lane < 4 * Q)
          AMX_DMA16((const char*)p.bias + (it.cg * 16 * Q) * 4 + lane * 16, buf + C::BIASOFF);
      }
    }
  };
  auto step_next = [&]() {
    if (++nx_stage == nstage) {
      nx_stage = 0;
      advance_item(nx, p);
    }
  };

  int* ready = (int*)(smem + C::FLAGOFF);
  int* done = (int*)(smem + C::FLAGOFF + 32);
  if (NLW) {
    if (tid < 16) ((int*)(smem + C::FLAGOFF))[tid] = (tid >= 8 + NW) ? 0x7fffffff : 0;   // absent MFMA waves: "done"
    __syncthreads();
    if (wave >= NW) {
      // ================================ loader wave ================================
      const unsigned a_ready = lds_addr(ready + iw), a_done = lds_addr(done);
      for (int t = 0; t < T_total; ++t) {
        if (t >= NBUF)                                    // buffer t % NBUF is free once every MFMA wave finished step t - NBUF
          while (__builtin_amdgcn_readfirstlane(flag_min8_asm(a_done)) < t - NBUF + 1) __builtin_amdgcn_s_sleep(1);
        dma_count = 0;
        issue(nx, nx_stage, t % NBUF, true);
        step_next();
        if (t > 0) {                                      // step t-1 has landed once only step t's DMA is still in flight
          wait_vmcnt_dyn(__builtin_amdgcn_readfirstlane(dma_count));
          flag_store_asm(a_ready, t);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      flag_store_asm(a_ready, T_total);
      return;
    }
  }

  // ---- deferred store of a finished item's packed outputs
  unsigned pend[CTW][NPEND];
  ItemCoord pd = cu;
  bool pending = false;
  auto flush = [&]() {
    const int z0 = pd.bz * C::TZ, y0 = pd.by * C::TY, x0 = pd.bx * C::TX;
    const int cb = pd.cg * 16 * Q + g * 4 * Q;
    const bool full = (z0 + C::TZ <= p.D) & (y0 + C::TY <= p.H) & (x0 + C::TX <= p.W);
    const int zl = z0 + wz * WZ, yl = y0 + wy * WY + dy, xl = x0 + dx;
    if (OUTMODE == 0) {
      const int ocs = p.ocs ? p.ocs : 32;              // the lane's 4Q <= 16 channels sit inside one 16-channel chunk
      char* lbase = p.out + (long long)pd.n * p.on + (long long)zl * p.oz + (long long)yl * p.oy +
                    (long long)xl * p.ox + (long long)(cb >> 4) * ocs + (cb & 15) * 2;
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
        if (!full && !((zl + cz < p.D) & (yl + cy * LY < p.H) & (xl + cx * LX < p.W))) continue;
        char* dst = lbase + cz * p.oz + (cy * LY) * p.oy + (cx * LX) * p.ox;
#pragma unroll
        for (int half = 0; half < (SPLIT ? 2 : 1); ++half) {          // SPLIT: hi channels, then the lo channels Cout further on
          char* d = dst + (long long)half * (p.Cout >> 4) * ocs;
          const int o = half * 2 * Q;
          if (Q == 1) {
            *(uint2*)d = make_uint2(pend[c][o], pend[c][o + 1]);
          } else {
#pragma unroll
            for (int j = 0; j < Q / 2; ++j)
              *(uint4*)(d + j * 16) = make_uint4(pend[c][o + 4 * j], pend[c][o + 4 * j + 1], pend[c][o + 4 * j + 2], pend[c][o + 4 * j + 3]);
          }
        }
      }
    } else {
      float* lbase = p.out32 + (long long)pd.n * p.pn + (long long)cb * p.pc + (long long)zl * p.pz +
                     (long long)yl * p.py + xl;
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
        if (!full && !((zl + cz < p.D) & (yl + cy * LY < p.H) & (xl + cx * LX < p.W))) continue;
        float* dst = lbase + cz * p.pz + (cy * LY) * p.py + cx * LX;
        if (p.wmap) {
          const float wgt = p.wmap[((long long)(zl + cz) * p.H + (yl + cy * LY)) * p.W + xl + cx * LX];
#pragma unroll
          for (int j = 0; j < 4 * Q; ++j) dst[(long long)j * p.pc] += wgt * __builtin_bit_cast(float, pend[c][j]);
        } else {
#pragma unroll
          for (int j = 0; j < 4 * Q; ++j) dst[(long long)j * p.pc] = __builtin_bit_cast(float, pend[c][j]);
        }
      }
    }
  };

  f32x4 acc[CTW][Q];
  f32x4 kshift[Q];                 // InstanceNorm statistics: the conv bias of this lane's channels, reloaded only when the cout group changes
  int kshift_cg = -1;

  // optional cycle trace (AMX_TRACE=1): wave 0 of each workgroup stamps s_memtime per phase
  unsigned long long* trace = (p.dbg & 8) ? (unsigned long long*)p.stats + (long long)blockIdx.x * 128 : nullptr;
  int tcount = 0;
#define AMX_STAMP()                                                               \
  do {                                                                            \
    if (trace && wave == 0 && lane == 0 && tcount < 128) trace[tcount] = __builtin_readcyclecounter(); \
    ++tcount;                                                                     \
  } while (0)

  AMX_STAMP();
  if (!NLW) {
    issue(nx, nx_stage, 0, true);
    step_next();
  }
  const bool late_issuer = !NLW && NW == 8 && wave >= 4 && !(p.dbg & 16);
  bool issue_due = false;
  constexpr int kLateStep = 3;
  auto late_issue = [&](int t) {
    issue(nx, nx_stage, (t + 1) & 1, !resident);
    step_next();
    issue_due = false;
  };
  AMX_STAMP();
  for (int t = 0; t < T_total; ++t) {
    if (NLW) {
      static_assert(NLW <= 4 && C::FLAGOFF % 16 == 0, "the ready flags are polled with one 16-byte read");
      while (flag_min4<(NLW ? NLW : 1)>(ready) < t + 1) __builtin_amdgcn_s_sleep(1);   // every loader's share of step t has landed
      asm volatile("" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA(t) pieces (and older stores) are done
      AMX_STAMP();
      __syncthreads();                                   // every wave's DMA(t) landed; buffer (t+1)&1 is free
      AMX_STAMP();
      if (t + 1 < T_total && !(p.dbg & 1)) {
        // Waves i and i + 4 share a SIMD.  If all eight issued their DMA share here the matrix pipes would idle for the ~1300
        // cycles that takes (a DMA instruction blocks its wave while the memory queue is full); the upper four therefore issue
        // theirs a few steps into the sweep, while their SIMD partners -- done issuing -- multiply.
        if (late_issuer) issue_due = true;
        else {
          issue(nx, nx_stage, (t + 1) & 1, !resident);
          step_next();
        }
      }
    }
    AMX_STAMP();
    if (pending) {                                     // previous item's outputs drain under this sweep
      if (!(p.dbg & 4)) flush();
      pending = false;
    }
    AMX_STAMP();
    const char* buf = smem + (t % NBUF) * C::BUF;
    const char* wbuf = resident ? smem : buf;          // resident weights/bias live in buffer 0
    if (cu_stage == 0) {
      // accumulators start from the bias (folded norm shift): lane holds channels cb .. cb+4Q-1
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const f32x4 bv = p.bias ? *(const f32x4*)(wbuf + C::BIASOFF + (g * Q + q) * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[c][q] = bv;
      }
    }

    // ---- MFMA sweep: 14 paired-tap steps per 16-channel sub-chunk
    const bool up_stage = LOWUP && p.up_shift && (((cu_stage * NCH) % nchunk) << 4) >= p.C0;
    const bool mx_stage = MX && cu_stage * NCH >= nchunk;
    // Both sweeps are software-pipelined one step deep by hand: the fragments of step i+1 are requested before the MFMAs of
    // step i, and scheduling fences keep that order (left alone, hipcc issues a step's reads right before a full
    // `s_waitcnt lgkmcnt(0)` and exposes the LDS latency every four MFMAs -- measured 38 % MFMA-busy with the DMA switched off).
    if (MX && mx_stage) {
      if (!(p.dbg & 2)) {
        // fp8 correction stage: 7 steps x NCH chunks; fragments are 2 x 16 bytes per lane (A: two 1-KiB halves, B: two taps).
        // The (step, column tile) sequence is pipelined one item deep: a 32-cycle MFMA per (tile, q) leaves the LDS far more time per
        // fragment than the f16 sweep has, and whole-step double buffering of the 32-byte B fragments does not fit 256 VGPRs.
        constexpr int TSX = NCH * 7;
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        constexpr int NA = CTW == 1 ? 3 : 2;                    // a step's A fragments are requested two ITEMS ahead: with one item per
        i32x4 fa[NA][Q][2], fb[3][2];                           // step that is two steps, i.e. three sets in flight
        auto tapoff = [&](const int tp) {                       // byte offset of tap tp in the halo image (27 -> 26: zero weights)
          const int t = tp > 26 ? 26 : tp;
          return (((t / 9) * HY + (t / 3) % 3) * HX + t % 3) * 16;
        };
        auto load_a = [&](const int i, const int set) {
          const int k = i / 7, j = i - k * 7;
#pragma unroll
          for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < 2; ++r)
              fa[set][q][r] = *(const i32x4*)(wbuf + WOFF + (((k * 7 + j) * Q + q) * 2 + r) * 1024 + lane * 16);
        };
        auto load_b = [&](const int i, const int c, const int set) {
          const int k = i / 7, j = i - k * 7;
          const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
          const int coff = ((cz * HY + cy * LY) * HX + cx * LX) * 16;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int lo = tapoff(4 * j + r), up = tapoff(4 * j + 2 + r);
            fb[set][r] = *(const i32x4*)(buf + base_d0 + hi * (up - lo) + (lo + k * HALO + coff));   // per lane: tap 4 j + 2 hi + r
          }
        };
        // item u = (step i, column tile c); its fragments are requested two items ahead (a wave issues only Q 32-cycle MFMAs per
        // item: one item of lookahead left the LDS latency exposed), the step's A fragments with its first item
        auto request = [&](const int u2) {
          if (u2 < TSX * CTW) {
            const int i2 = u2 / CTW, c2 = u2 - i2 * CTW;
            load_b(i2, c2, u2 % 3);
            if (c2 == 0) load_a(i2, i2 % NA);
          }
        };
        const int sb = 0x7f7f7f7f;                              // activations: scale 1
        request(0);
        request(1);
#pragma unroll
        for (int i = 0; i < TSX; ++i) {
#pragma unroll
          for (int c = 0; c < CTW; ++c) {
            const int u = i * CTW + c;
            request(u + 2);
            __builtin_amdgcn_sched_barrier(0);
            const i32x8 b8 = __builtin_shufflevector(fb[u % 3][0], fb[u % 3][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
              const i32x8 a8 = __builtin_shufflevector(fa[i % NA][q][0], fa[i % NA][q][1], 0, 1, 2, 3, 4, 5, 6, 7);
              acc[c][q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[c][q], 0, 0, 0, mx_sa, 0, sb);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!NLW && u == kLateStep && issue_due) late_issue(t);
          }
        }
      }
    } else if (!(p.dbg & 2) && up_stage) {
      // the stage buffer holds the LOW-RES halo of 16 upsampled channels
      constexpr int LXH = C::LXH;
      vec8 fa[2][Q], fb[2][CTW];
      auto load_step = [&](const int s, const int set) {
        const int kz = s < 9 ? s / 3 : (s < 12 ? s - 9 : (s == 12 ? 0 : 2));
        const int ky = s < 9 ? s % 3 : (s < 12 ? 0 : 2);
#pragma unroll
        for (int q = 0; q < Q; ++q) fa[set][q] = *(const vec8*)(wbuf + WOFF + (s * Q + q) * 1024 + lane * 16);
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          const int cx = c % XT, cy = (c / XT) % YT;
          const int rowc = ((cy + ky - 1) >> 1) + 1;               // low row of the tile for tap ky (lo lanes)
          const int ubase = s < 9 ? U1[kz] : (s < 12 ? ((cy & 1) ? U2o[kz] : U2e[kz]) : (s == 12 ? Uz : U2o[2]));
          fb[set][c] = *(const vec8*)(buf + ubase + (rowc * LXH + cx * 8) * 16);
        }
      };
      load_step(0, 0);
#pragma unroll
      for (int s = 0; s < kSteps; ++s) {
        if (s + 1 < kSteps) load_step(s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[c][q] = Ops<T>::mfma(fa[s & 1][q], fb[s & 1][c], acc[c][q]);
        __builtin_amdgcn_sched_barrier(0);
        if (!NLW && s == kLateStep && issue_due) late_issue(t);
      }
    } else if (!(p.dbg & 2)) {
      constexpr int TS = NCH * kSteps;
      vec8 fa[2][Q], fb[2][CTW];
      auto load_step = [&](const int i, const int set) {
        const int k = i / kSteps, s = i - k * kSteps;
        const int kz = s < 9 ? s / 3 : (s < 12 ? s - 9 : (s == 12 ? 0 : 2));
        const int ky = s < 9 ? s % 3 : (s < 12 ? 0 : 2);
        const int kx = s < 9 ? 0 : 2;
        const int tapoff = ((kz * HY + ky) * HX + kx) * 16;
        const int bsel = s < 9 ? base_d1 : (s < 12 ? base_dx : (s == 12 ? base_dz : base_d0));
#pragma unroll
        for (int q = 0; q < Q; ++q)
          fa[set][q] = *(const vec8*)(wbuf + WOFF + ((k * kSteps + s) * Q + q) * 1024 + lane * 16);
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
          const int coff = ((cz * HY + cy * LY) * HX + cx * LX) * 16;
          fb[set][c] = *(const vec8*)(buf + bsel + k * HALO + tapoff + coff);
        }
      };
      load_step(0, 0);
#pragma unroll
      for (int i = 0; i < TS; ++i) {
        if (i + 1 < TS) load_step(i + 1, (i + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[c][q] = Ops<T>::mfma(fa[i & 1][q], fb[i & 1][c], acc[c][q]);
        __builtin_amdgcn_sched_barrier(0);
        if (!NLW && i == kLateStep && issue_due) late_issue(t);
      }
    }
    if (!NLW && issue_due) late_issue(t);          // sweep skipped (debug ablation)
    AMX_STAMP();
    if (NLW) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of this stage has returned: the buffer may be refilled
      flag_store(done + wave, t + 1);
    }
    if (++cu_stage < nstage) continue;
    cu_stage = 0;

    // ---- InstanceNorm statistics of this item (ConvParams::stats, not with the cycle trace): shifted sums S1 = sum(x - K),
    //      S2 = sum((x - K)^2) of the values ABOUT TO BE STORED (rounded like the store; strict: hi + lo = the fp32 value), K = the
    //      conv bias of the channel.  Reduced over the 16 voxel lanes of a lane group with DPP row shifts, ONE slot per
    //      (brick, wave) and sample: partial[n][brick * NW + wave][channel][2] -- every entry written exactly once, summed in a
    //      fixed order by in_finalize (deterministic, no atomics).  Saves the separate statistics pass over the tensor.
    if (OUTMODE == 0 && p.stats && !(p.dbg & 8)) {
      const int z0s = cu.bz * C::TZ, y0s = cu.by * C::TY, x0s = cu.bx * C::TX;
      const int zls = z0s + wz * WZ, yls = y0s + wy * WY + dy, xls = x0s + dx;
      const int cbs = cu.cg * 16 * Q + g * 4 * Q;
      if (cu.cg != kshift_cg) {        // (a global load per item stalled the epilogue: +7 % on the 32 -> 32 @128^3 layers)
        kshift_cg = cu.cg;
#pragma unroll
        for (int q = 0; q < Q; ++q) kshift[q] = p.bias ? *(const f32x4*)(p.bias + cbs + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      float s1[4 * Q], s2[4 * Q];
#pragma unroll
      for (int k = 0; k < 4 * Q; ++k) s1[k] = s2[k] = 0.f;
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
        const bool in = (zls + cz < p.D) & (yls + cy * LY < p.H) & (xls + cx * LX < p.W);
        if (in) {
#pragma unroll
          for (int q = 0; q < Q; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float v = SPLIT ? acc[c][q][j] : (float)(T)acc[c][q][j];
              const float d = v - kshift[q][j];
              s1[q * 4 + j] += d;
              s2[q * 4 + j] += d * d;
            }
          }
        }
      }
      // sum over the 16 lanes of the lane group: row_shr 1, 2, 4, 8 (out-of-row lanes contribute 0); lane 15 of the row holds it
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
        const long long brick = ((long long)cu.bz * p.nby + cu.by) * p.nbx + cu.bx;
        const long long nslots = (long long)p.nbz * p.nby * p.nbx * NW;
        float* o = p.stats + (((long long)cu.n * nslots + brick * NW + wave) * p.Cout + cbs) * 2;
#pragma unroll
        for (int k = 0; k < 4 * Q; k += 2)               // (S1, S2) pairs of two channels per 16-byte store
          *(float4*)(o + 2 * k) = make_float4(s1[k], s2[k], s1[k + 1], s2[k + 1]);
      }
    }

    // ---- item finished: activation + pack into the pending registers (stored next iteration)
    act_inplace<CTW * Q>(&acc[0][0], p.act, p.slope);
    bool bad = false;
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      float v[4 * Q];
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float f = acc[c][q][j];
          if (OUTMODE == 0 && RangeCheck<T>::on) bad |= RangeCheck<T>::bad(f);   // the value about to be stored
          v[q * 4 + j] = f;
        }
      if (OUTMODE == 0) {
#pragma unroll
        for (int j = 0; j < 2 * Q; ++j)
          pend[c][j] = (unsigned)to_bits<T>(v[2 * j]) | ((unsigned)to_bits<T>(v[2 * j + 1]) << 16);
        if (SPLIT) {
#pragma unroll
          for (int j = 0; j < 2 * Q; ++j) {
            const float r0 = v[2 * j] - (float)(T)v[2 * j], r1 = v[2 * j + 1] - (float)(T)v[2 * j + 1];
            pend[c][2 * Q + j] = (unsigned)to_bits<T>(r0) | ((unsigned)to_bits<T>(r1) << 16);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4 * Q; ++j) pend[c][j] = __builtin_bit_cast(unsigned, v[j]);
      }
    }
    if (OUTMODE == 0 && RangeCheck<T>::on) raise_flag(p.oflow, bad);
    pd = cu;
    pending = true;
    advance_item(cu, p);
  }
  if (pending && !(p.dbg & 4)) flush();
}

// -------------------------------------------------------------------------------------------
// launcher
// -------------------------------------------------------------------------------------------
static thread_local char g_kernel_name2[64] = "";
const char* last_conv_v2_kernel_name() { return g_kernel_name2; }
static thread_local int g_stats_slots = 0;
int last_conv_v2_stats_slots() { return g_stats_slots; }       // partial-statistics slots per sample written by the last launch
int conv_v2_stats_slots(int D, int H, int W, int Q);

static int g_num_cus = 0;

template <typename T, int SPLITM, int WZ, int WY, int WX, int NWZ, int NWY, int Q, int NCH, int OUTMODE, int NLW = 0, int NBUF = 2>
static hipError_t launch_cfg2(ConvParams p, hipStream_t st) {
  constexpr bool SPLIT = SPLITM >= 1, MX = SPLITM == 2;          // SPLITM: 0 single value, 1 hi / lo split, 2 split + fp8 correction stages
  typedef Conv2Cfg<T, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE, NLW, NBUF> C;
  if (MX && !p.mxs) return hipErrorInvalidValue;
  const char* tn = __is_same(T, f16) ? (MX ? "f16x2mx" : (SPLIT ? "f16x2" : "f16")) : (SPLIT ? "bf16x2" : "bf16");
  if (NLW)
    snprintf(g_kernel_name2, sizeof g_kernel_name2, "conv3d_k3_v2<%s,%dx%dx%d,w%d+l%d,b%d,q%d,nch%d,o%d>",
             tn, C::TZ, C::TY, C::TX, C::NW, NLW, NBUF, Q, NCH, OUTMODE);
  else
    snprintf(g_kernel_name2, sizeof g_kernel_name2, "conv3d_k3_v2<%s,%dx%dx%d,w%d,q%d,nch%d,o%d>",
             tn, C::TZ, C::TY, C::TX, C::NW, Q, NCH, OUTMODE);
  auto kern = conv3d_k3_v2_kernel<T, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE, NLW, NBUF, SPLIT, MX>;
  static amx::DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_once.set();
  }
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  p.nbz = (p.D + C::TZ - 1) / C::TZ;
  p.nby = (p.H + C::TY - 1) / C::TY;
  p.nbx = (p.W + C::TX - 1) / C::TX;
  g_stats_slots = p.nbz * p.nby * p.nbx * C::NW;
  // the forward sizes its statistics scratch with conv_v2_stats_slots(): the two must agree, or the epilogue would write past it
  if (p.stats && !(p.dbg & 8) && g_stats_slots != conv_v2_stats_slots(p.D, p.H, p.W, Q)) return hipErrorInvalidConfiguration;
  static int dbg = -1;
  static unsigned long long* trace_buf = nullptr;
  if (dbg < 0) {
    const char* e = exp_env("AMX_DBG");
    dbg = e ? atoi(e) : 0;
    if (exp_env("AMX_TRACE")) dbg |= 8;
  }
  p.dbg = dbg;
  const long long items = (long long)p.nbz * p.nby * p.nbx * p.N * (p.Cout / (16 * Q));
  const int per_cu = C::LDS_BYTES <= 80 * 1024 ? 2 : 1;
  long long grid = items < (long long)g_num_cus * per_cu ? items : (long long)g_num_cus * per_cu;
  if (dbg & 8) {   // debug only: per-phase cycle stamps of a few workgroups, printed after a sync
    if (!trace_buf && hipMalloc((void**)&trace_buf, 1024 * 128 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(trace_buf, 0, 1024 * 128 * 8, st);
    p.stats = (float*)trace_buf;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((C::NW + NLW) * 64), C::LDS_BYTES, st, p);
  if (dbg & 8) {
    static int printed = 0;
    (void)hipStreamSynchronize(st);
    if (printed++ == 3) {
      static unsigned long long hostbuf[1024 * 128];
      (void)hipMemcpy(hostbuf, trace_buf, sizeof hostbuf, hipMemcpyDeviceToHost);
      const int wgs[4] = {0, 1, (int)grid / 2, (int)grid - 1};
      for (int wi = 0; wi < 4; ++wi) {
        const unsigned long long* tr = hostbuf + (long long)wgs[wi] * 128;
        fprintf(stderr, "[trace %s wg %d] start %llu :", g_kernel_name2, wgs[wi], tr[0]);
        for (int k = 1; k < 128 && tr[k]; ++k) fprintf(stderr, " %llu", tr[k] - tr[k - 1]);
        fprintf(stderr, "\n");
      }
    }
  }
  return hipGetLastError();
}

// Loader-wave mode (4 loader waves, ring of 3 buffers) where three stage buffers fit the LDS -- the deep, small levels,
// whose stages are short: 128 -> 128 @16^3 32 -> 27 us, 384 -> 128 @16^3 66 -> 52 us, 256 -> 256 @8^3 36 -> 30 us at batch 4.
// With only two buffers the loaders cannot run ahead and the classic barrier pipeline is faster (96 -> 32 @64^3: 202 vs 223 us).
template <typename T, int SPLIT, int WZ, int WY, int WX, int NWZ, int NWY, int Q, int NCH, int OUTMODE>
static hipError_t launch_pick(const ConvParams& p, hipStream_t st) {
  static int classic = -1;
  if (classic < 0) classic = exp_env("AMX_V2_CLASSIC") ? 1 : 0;
  typedef Conv2Cfg<T, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE> C0;
  constexpr bool can = 3 * C0::BUF + 64 <= 160 * 1024 && NWZ * NWY <= 8;
  if constexpr (can) {
    if (!classic) return launch_cfg2<T, SPLIT, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE, 4, 3>(p, st);
  }
  return launch_cfg2<T, SPLIT, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE>(p, st);
}

// environment switches of the brick choice, read once (the statistics-slot count below must see the same values as the dispatch)
static bool v2_narrow() {
  static int v = -1;
  if (v < 0) v = exp_env("AMX_V2_NARROW") ? 1 : 0;
  return v != 0;
}
static bool v2_half_brick() {
  static int v = -1;
  if (v < 0) v = exp_env("AMX_V2_NO_HALF_BRICK") ? 0 : 1;
  return v != 0;
}

template <typename T, int OUTMODE, int SPLIT>
static hipError_t launch_conv2_t(const ConvParams& p, int Q, hipStream_t st) {
  const int nch = (p.C0 + p.C1) / 16;
  if (p.W >= 32) {
    if (Q == 1) return launch_pick<T, SPLIT, 1, 4, 32, 4, 2, 1, 1, OUTMODE>(p, st);       // brick 4x8x32, 8 waves
    if (Q == 2) {
      // AMX_V2_NARROW: brick 4x4x16 instead -- three stage buffers fit the LDS, so the loader-wave pipeline applies.  Measured on
      // 96 -> 32 @64^3, batch 4: 206 -> 197 us (+5 %); not the default: the same instantiation also serves the 16^3 layers, and one
      // kernel name per layer class keeps the per-kernel tables of bench.py and rocprofv3 comparable.
      if (OUTMODE == 0 && v2_narrow()) return launch_pick<T, SPLIT, 1, 2, 16, 4, 2, 2, 1, 0>(p, st);
      // (brick 4x8x16 instead -- 12 % less halo, 84 % instead of 53 % of the DMA lanes used -- measured on f16x2mx, batch 4: 64 -> 64 @64^3
      //  506 -> 508 us, 192 -> 64 @64^3 1337 -> 1377, 96 -> 32 @128^3 2815 -> 2854: these layers are not fill-bound)
      return launch_pick<T, SPLIT, 1, 2, 32, 4, 2, 2, 1, OUTMODE>(p, st);       // brick 4x4x32, 8 waves
    }
    if (OUTMODE == 0 && Q == 4) return launch_pick<T, SPLIT, 1, 2, 16, 4, 2, 4, 1, 0>(p, st);  // brick 4x4x16, 8 waves
    return hipErrorInvalidValue;
  }
  if (OUTMODE == 1) return hipErrorInvalidValue;
  if (p.W >= 16) {
    if (Q == 1) return launch_pick<T, SPLIT, 1, 2, 16, 4, 1, 1, 1, 0>(p, st);
    if (Q == 2) return launch_pick<T, SPLIT, 1, 2, 16, 4, 2, 2, 1, 0>(p, st);       // brick 4x4x16, 8 waves
    if (Q == 4) return launch_pick<T, SPLIT, 1, 2, 16, 4, 2, 4, 1, 0>(p, st);
  }
  if (Q == 1) {
    if (nch % 2 == 0) return launch_pick<T, SPLIT, 1, 2, 8, 4, 1, 1, 2, 0>(p, st);
    return launch_pick<T, SPLIT, 1, 2, 8, 4, 1, 1, 1, 0>(p, st);
  }
  // 4-wide levels: half bricks (4x2x8, 4 waves) -- twice the workgroups (1024 -> 1024 @4^3, batch 4: 128 -> 256 on 256 CUs) and half as
  // many waves streaming the same weight fragments from L2
  if (Q == 2 && p.W <= 4 && v2_half_brick()) return launch_pick<T, SPLIT, 1, 2, 8, 4, 1, 2, 1, 0>(p, st);
  if (Q == 2) return launch_pick<T, SPLIT, 1, 2, 8, 4, 2, 2, 1, 0>(p, st);           // brick 4x4x8, 8 waves
  if (Q == 4) return launch_pick<T, SPLIT, 1, 2, 8, 4, 1, 4, 1, 0>(p, st);
  return hipErrorInvalidValue;
}

// Partial-statistics slots per sample a launch of this layer writes (ConvParams::stats): bricks x MFMA waves of the brick shape
// launch_conv2_t picks for (W, Q) -- keep in step with the dispatch above.
int conv_v2_stats_slots(int D, int H, int W, int Q) {
  int tz = 4, ty, tx, nw;
  if (W >= 32) {
    if (Q == 1) { ty = 8; tx = 32; nw = 8; }
    else if (Q == 2) { ty = 4; tx = v2_narrow() ? 16 : 32; nw = 8; }
    else { ty = 4; tx = 16; nw = 8; }
  } else if (W >= 16) {
    if (Q == 1) { ty = 2; tx = 16; nw = 4; }
    else { ty = 4; tx = 16; nw = 8; }
  } else {
    if (Q == 2 && !(W <= 4 && v2_half_brick())) { ty = 4; tx = 8; nw = 8; }
    else { ty = 2; tx = 8; nw = 4; }
  }
  return ((D + tz - 1) / tz) * ((H + ty - 1) / ty) * ((W + tx - 1) / tx) * nw;
}

hipError_t launch_conv_v2(const ConvParams& p, int precision, int Q, hipStream_t st) {
  const bool planar = p.out32 != nullptr;
  switch (precision) {
    case 0: return planar ? launch_conv2_t<f16, 1, false>(p, Q, st) : launch_conv2_t<f16, 0, false>(p, Q, st);
    case 1: return planar ? launch_conv2_t<bf16, 1, false>(p, Q, st) : launch_conv2_t<bf16, 0, false>(p, Q, st);
    case 2: return planar ? launch_conv2_t<f16, 1, true>(p, Q, st) : launch_conv2_t<f16, 0, true>(p, Q, st);
    case 3: return planar ? launch_conv2_t<bf16, 1, true>(p, Q, st) : launch_conv2_t<bf16, 0, true>(p, Q, st);
    case 4: return planar ? launch_conv2_t<f16, 1, 2>(p, Q, st) : launch_conv2_t<f16, 0, 2>(p, Q, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace amx
