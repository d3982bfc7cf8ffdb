// anatomix_amd -- the decoder's last "upsample + concat + conv" at full resolution:
//   y = act(conv3x3x3_reflect(cat(skip[16ch], up2_nearest(low[32ch]))) + shift),  48 -> 16 channels
// (network.py modules 58-61 of the 6M model: nn.Upsample(2,'nearest') -> torch.cat((skip, up), 1)
// -> nn.Conv3d(48,16,3,reflect) -> BatchNorm3d -> ReLU).  25 % of the network's FLOPs.
//
// Two algebraic facts make this layer much cheaper than a 48-channel 27-tap convolution:
//  (1) a 3x3x3 convolution over a NEAREST-upsampled tensor only ever sees 2x2x2 distinct low-res
//      voxels: for an output voxel of parity p = o & 1 along an axis the three taps o-1, o, o+1 read
//      low-res voxels {l-1, l, l} (p = 0) or {l, l, l+1} (p = 1), l = o >> 1.  Pre-summing the
//      weights that hit the same voxel turns the 27-tap/32-channel part into 8 taps with
//      parity-dependent weights: 8 MFMAs instead of 27 per 16 voxels x 16 channels.  Reflect
//      padding of the upsampled tensor becomes REPLICATE padding of the low-res tensor
//      (-1 -> 1 -> low 0; N -> N-2 -> low N/2-1), so the rule holds at the volume border too.
//      The merged weights are summed in fp32 and rounded once, which is closer to the fp32
//      reference than rounding 27 weights separately;
//  (2) the low-res tensor is staged ONCE per workgroup at low resolution (1/8 of the bytes of its
//      upsampled image), the skip tensor streams through a z-marching ring exactly as in
//      amx_conv3d_zmarch.hip.
// Work decomposition: a workgroup marches an 8x32 in-plane tile along z, 2 output planes (= 1
// low-res plane) per step.  Each of the 8 consumer waves owns ONE parity class (pz,py,px): its
// column tiles are 16 voxels x = 2i + px of one row, so all lanes share the class's merged weights,
// which stay in registers (8 fragments) next to the 14 skip fragments.  The skip planes are stored
// parity-split along x in LDS (even voxels, then odd voxels of a row) so that a tile's 16 voxels
// are 16 consecutive 16-byte slots for every tap (conflict-free ds_read_b128).  Three loader waves
// stream the skip planes (one per 8-channel plane) and the low-res planes by LDS-DMA.
#include <stdio.h>
#include <stdlib.h>

#include "amx_device.h"

namespace amx {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct UpcatCfg {
  static constexpr int TY = 8, TX = 32, NC = 8, NL = 3;
  static constexpr int R = 10;                                   // skip ring (z-planes)
  static constexpr int RL = 6;                                   // low-res ring (low-res z-planes)
  static constexpr int HY = TY + 2, HX = TX + 2, HXH = HX / 2;   // 34 voxels per row = 17 even + 17 odd
  static constexpr int HVP = HY * HX;
  static constexpr int PPL = ((HVP * 16 + 255) / 256) * 256;     // one 8-channel plane of a skip z-plane
  static constexpr int PLSZ = 2 * PPL;
  static constexpr int LY = TY / 2 + 2, LX = TX / 2 + 2, LVP = LY * LX;   // low-res halo plane 6 x 18
  static constexpr int LPL = ((LVP * 16 + 255) / 256) * 256;     // one 8-channel plane of a low-res z-plane
  static constexpr int LPSZ = 4 * LPL;                           // 32 channels
  static constexpr int LOFF = R * PLSZ;                          // low-res ring base
  static constexpr int FLAGOFF = LOFF + RL * LPSZ;               // ready[3] at +0, done[8] at +32
  static constexpr int LDS_BYTES = FLAGOFF + 64;
  static constexpr int NDMA = (HVP + 63) / 64;                   // per (skip z-plane, channel plane)
  static constexpr int NDMAL = 4 * ((LVP + 63) / 64);            // per low-res z-plane
  static constexpr int AHEAD = R - 4;                            // skip planes beyond a step's needs (TZ = 2)
  static constexpr int AHEADL = RL - 3;
  static_assert(LDS_BYTES <= 160 * 1024, "rings must fit the LDS");
  static_assert(R * NDMA <= 60 && RL * NDMAL <= 60, "vmcnt range");
};

template <typename T, int OUTMODE>
__global__ __launch_bounds__((UpcatCfg::NC + UpcatCfg::NL) * 64) void conv3d_upcat16_kernel(const ConvParams p, int zseg,
                                                                                           int nseg) {
  typedef UpcatCfg C;
  typedef typename Ops<T>::vec8 vec8;
  constexpr int HX = C::HX, PPL = C::PPL, PLSZ = C::PLSZ, LX = C::LX, LPL = C::LPL, LPSZ = C::LPSZ, R = C::R, RL = C::RL;
  constexpr int NC = C::NC, TY = C::TY, TX = C::TX;

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
  const int zs = sg * zseg;                                  // even
  const int ze = (zs + zseg < p.D) ? zs + zseg : p.D;        // output planes [zs, ze), ze - zs even
  const int nsteps = (ze - zs) / 2;
  const int nplanes = ze - zs + 2;                           // skip planes  q  <-> z  = zs - 1 + q
  const int nlow = nsteps + 2;                               // low planes   ql <-> lz = zs/2 - 1 + ql

  // producer/consumer counters in LDS (no workgroup barrier in the march, see amx_device.h):
  // ready[0], ready[1]: skip planes landed per channel plane; ready[2]: low-res planes landed; done[w]: steps consumed
  int* ready = (int*)(smem + C::FLAGOFF);
  int* done = (int*)(smem + C::FLAGOFF + 32);
  if (tid < 16) ((int*)(smem + C::FLAGOFF))[tid] = 0;
  __syncthreads();

  if (wave >= NC + 2) {
    // ====================== loader: low-res planes (32 channels, replicate-clamped) ======================
    constexpr int NJ = (C::LVP + 63) / 64;
    int off[NJ];
    bool valid[NJ];
    const int LD = p.D >> 1, LH = p.H >> 1, LW = p.W >> 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int hv = j * 64 + lane;
      const int hy = hv / LX, hx = hv - hy * LX;
      valid[j] = hv < C::LVP;
      int gy = (y0 >> 1) - 1 + hy, gxx = (x0 >> 1) - 1 + hx;
      gy = gy < 0 ? 0 : (gy >= LH ? LH - 1 : gy);
      gxx = gxx < 0 ? 0 : (gxx >= LW ? LW - 1 : gxx);
      off[j] = gy * (int)p.s1y + gxx * (int)p.s1x;
    }
    const char* src_n = p.src1 + (long long)n * p.s1n;
    auto issue_plane = [&](int ql) {
      int lz = (zs >> 1) - 1 + ql;
      lz = lz < 0 ? 0 : (lz >= LD ? LD - 1 : lz);
      const char* plane = src_n + (long long)lz * p.s1z;
      char* dstp = smem + C::LOFF + (ql % RL) * LPSZ;
#pragma unroll
      for (int cg = 0; cg < 4; ++cg)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (valid[j])
            __builtin_amdgcn_global_load_lds((gptr_t)(plane + off[j] + cg * 16), (lptr_t)(dstp + cg * LPL + j * 1024), 16, 0, 0);
    };
    int next_issue = 0, next_pub = 0;
    const unsigned a_ready = lds_addr(ready + 2), a_done = lds_addr(done);
    while (next_pub < nlow) {
      if (next_issue < nlow) {
        const int md = __builtin_amdgcn_readfirstlane(flag_min8_asm(a_done));
        int lim = RL + md;                                    // low planes ql < md are dead
        lim = lim < nlow ? lim : nlow;
        while (next_issue < lim) issue_plane(next_issue++);
      }
      if (next_issue == next_pub) {
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      WaitVm<C::NDMAL, RL - 1>::run(next_issue - next_pub - 1);
      flag_store_asm(a_ready, ++next_pub);
    }
    return;
  }
  if (wave >= NC) {
    // ====================== loader: skip planes, channel plane cp, parity-split rows ======================
    const int cp = wave - NC;
    int off[C::NDMA];
    bool valid[C::NDMA];
#pragma unroll
    for (int j = 0; j < C::NDMA; ++j) {
      const int sl = j * 64 + lane;                    // LDS slot within the plane
      const int hy = sl / HX, pos = sl - hy * HX;
      const int hx = pos < C::HXH ? 2 * pos : 2 * (pos - C::HXH) + 1;     // even voxels first, then odd
      valid[j] = sl < C::HVP;
      off[j] = reflect_clamp(y0 + hy - 1, p.H) * (int)p.s0y + reflect_clamp(x0 + hx - 1, p.W) * (int)p.s0x + cp * 16;
    }
    const char* src_n = p.src0 + (long long)n * p.s0n;
    auto issue_plane = [&](int q) {
      const char* plane = src_n + (long long)reflect_clamp(zs - 1 + q, p.D) * p.s0z;
      char* dstp = smem + (q % R) * PLSZ + cp * PPL;
#pragma unroll
      for (int j = 0; j < C::NDMA; ++j)
        if (valid[j]) __builtin_amdgcn_global_load_lds((gptr_t)(plane + off[j]), (lptr_t)(dstp + j * 1024), 16, 0, 0);
    };
    int next_issue = 0, next_pub = 0;
    const unsigned a_ready = lds_addr(ready + cp), a_done = lds_addr(done);
    while (next_pub < nplanes) {
      if (next_issue < nplanes) {
        const int md = __builtin_amdgcn_readfirstlane(flag_min8_asm(a_done));
        int lim = R + 2 * md;                                 // skip planes q < 2*md are dead
        lim = lim < nplanes ? lim : nplanes;
        while (next_issue < lim) issue_plane(next_issue++);
      }
      if (next_issue == next_pub) {
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      WaitVm<C::NDMA, R - 1>::run(next_issue - next_pub - 1);
      flag_store_asm(a_ready, ++next_pub);
    }
    return;
  }

  // ================================= consumer wave: parity class (pz, py, px) =================================
  const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
  const int li = lane & 15, g = lane >> 4, hi = g >> 1;
  vec8 wsk[kSteps], wup[8];
#pragma unroll
  for (int s = 0; s < kSteps; ++s) wsk[s] = *(const vec8*)(p.wpk + s * 1024 + lane * 16);
#pragma unroll
  for (int e = 0; e < 8; ++e) wup[e] = *(const vec8*)(p.wpk + (kSteps + wave * 8 + e) * 1024 + lane * 16);
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *(const f32x4*)(p.bias + g * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // skip image: voxel x = 2*li + px at tap kx sits in row slot ((px+kx)&1)*17 + li + ((px+kx)>>1)
  const int slot0 = px * C::HXH + li;                         // kx = 0
  const int d1 = px ? -(C::HXH - 1) : C::HXH;                 // slot(kx=1) - slot(kx=0)
  const int lanebase = (g & 1) * PPL + (py * HX + slot0) * 16;          // row py + ky + 2*ly
  const int base_d1 = lanebase + hi * d1 * 16;
  const int base_dx = lanebase + hi * HX * 16;
  // low-res image: voxel lx = li + (px - 1 + ex) + 1 (halo), row ly + (py - 1 + ey) + 1, plane lz + (pz - 1 + ez)
  const int lbase = C::LOFF + g * LPL + ((py * LX) + li + px) * 16;

  const bool full_xy = (y0 + TY <= p.H) & (x0 + TX <= p.W);
  const int yl = y0 + py, xl = x0 + 2 * li + px;
  char* out_l = OUTMODE == 0 ? p.out + (long long)n * p.on + (long long)yl * p.oy + (long long)xl * p.ox + g * 8 : nullptr;
  char* out_w = OUTMODE == 0 ? p.out + (long long)n * p.on + (long long)yl * p.oy + (long long)xl * p.ox + (g >> 1) * 16 : nullptr;
  float* out32_l = OUTMODE == 1 ? p.out32 + (long long)n * p.pn + (long long)(g * 4) * p.pc + (long long)yl * p.py + xl
                                : nullptr;

  bool bad = false;
  for (int s = 0; s < nsteps; ++s) {
    {
      int need = 2 * s + 4, needl = s + 3;                   // skip planes q <= 2s+3 and low planes ql <= s+2
      need = need < nplanes ? need : nplanes;
      needl = needl < nlow ? needl : nlow;
      while (true) {
        static_assert(C::FLAGOFF % 16 == 0, "the ready flags are polled with one 16-byte read");
        const flag4_t r = flag_read4(ready);                  // (one 16-byte read instead of three relaxed loads per poll)
        if ((r[0] < r[1] ? r[0] : r[1]) >= need && r[2] >= needl) break;
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
    }
    const int zo = zs + 2 * s + pz;
    // ring offsets of the three input planes and the two low-resolution planes: wave-uniform, kept in scalar registers
    int slo[3], llo[2];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) slo[kz] = __builtin_amdgcn_readfirstlane(((2 * s + pz + kz) % R) * PLSZ);
#pragma unroll
    for (int ez = 0; ez < 2; ++ez) llo[ez] = __builtin_amdgcn_readfirstlane(((s + pz + ez) % RL) * LPSZ);
    const int bzv = lanebase + (hi ? slo[1] : slo[0]);      // taps (0,2,2)|(1,2,2): low lanes plane 0, high lanes plane 1

    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = bias;

    if (!(p.dbg & 2)) {
      // The step's 88 fragment uses -- 14 paired-tap steps of the skip part (tile c = low row ly = c <-> output row y0 + 2c + py), then
      // the 8 merged taps e = (ez, ey, ex) of the upsampled part on the low-resolution ring, 4 row tiles each, ONE MFMA per fragment --
      // as one static sequence with rolling reads: use i sits in slot i mod NSLOT, the read for use i + NSLOT is issued right after the
      // MFMA of use i.  (Left to the compiler the sequence was "2 ds_read, wait, MFMA, wait, MFMA": the LDS round trip in front of
      // nearly every pair of MFMAs, in the largest kernel of the 6 M forward.)
      constexpr int NSK = 4 * kSteps, NU = NSK + 4 * 8, NSLOT = 8;
      auto src = [&](int i) -> const vec8* {                 // (i is a compile-time constant after unrolling)
        if (i < NSK) {
          const int st = i >> 2, c = i & 3;
          const int kz = st < 9 ? st / 3 : (st < 12 ? st - 9 : (st == 12 ? 0 : 2));
          const int ky = st < 9 ? st % 3 : (st < 12 ? 0 : 2);
          const int kx2 = st < 9 ? 0 : 1;                     // kx = 2 is one slot right of kx = 0 in the split row
          const int off = (ky * HX + kx2) * 16 + (2 * c * HX) * 16;
          if (st < 9) return (const vec8*)(smem + base_d1 + (slo[kz] + off));
          if (st < 12) return (const vec8*)(smem + base_dx + (slo[kz] + off));
          if (st == 12) return (const vec8*)(smem + bzv + off);
          return (const vec8*)(smem + lanebase + (slo[2] + off));
        }
        const int e = (i - NSK) >> 2, c = (i - NSK) & 3;
        const int ez = e >> 2, ey = (e >> 1) & 1, ex = e & 1;
        return (const vec8*)(smem + lbase + (llo[ez] + ((c + ey) * LX + ex) * 16));
      };
      vec8 S[NSLOT];
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) S[i] = *src(i);
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int c = i < NSK ? (i & 3) : ((i - NSK) & 3);
        acc[c] = Ops<T>::mfma(i < NSK ? wsk[i >> 2] : wup[(i - NSK) >> 2], S[i % NSLOT], acc[c]);
        if (i + NSLOT < NU) S[i % NSLOT] = *src(i + NSLOT);
        __builtin_amdgcn_sched_barrier(0);                   // (keeps every read behind the MFMA of the use it replaces, and no further)
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every ring read of this step has returned
    flag_store(done + wave, s + 1);

    act_inplace<4>(&acc[0], p.act, p.slope);
    if (OUTMODE == 0 && zo < ze && !(p.dbg & 4) && !(p.dbg & 32)) {
      // 16-bit channels-last output: the lane groups g and g ^ 1 exchange one row each (v_permlane16_swap), so that a lane
      // stores 8 consecutive channels of ONE voxel -- one 16-byte store per lane and row pair instead of two 8-byte ones.
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        unsigned pk[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float f = acc[2 * cp + h][j];
              if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(f);
            v[j] = f;
          }
          pk[h][0] = (unsigned)to_bits<T>(v[0]) | ((unsigned)to_bits<T>(v[1]) << 16);
          pk[h][1] = (unsigned)to_bits<T>(v[2]) | ((unsigned)to_bits<T>(v[3]) << 16);
        }
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        const int row = 2 * (2 * cp + (g & 1));            // even g: row of c = 2 cp, odd g: row of c = 2 cp + 1
        if (!full_xy && !((yl + row < p.H) & (xl < p.W))) continue;
        *(uint4*)(out_w + (long long)zo * p.oz + row * p.oy) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
    } else if (zo < ze && !(p.dbg & 4)) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!full_xy && !((yl + 2 * c < p.H) & (xl < p.W))) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float f = acc[c][j];
          if (OUTMODE == 0 && RangeCheck<T>::on) bad |= RangeCheck<T>::bad(f);   // the value about to be stored
          v[j] = f;
        }
        if (OUTMODE == 0) {
          char* dst = out_l + (long long)zo * p.oz + (2 * c) * p.oy;
          *(uint2*)dst = make_uint2((unsigned)to_bits<T>(v[0]) | ((unsigned)to_bits<T>(v[1]) << 16),
                                    (unsigned)to_bits<T>(v[2]) | ((unsigned)to_bits<T>(v[3]) << 16));
        } else {
          float* dst = out32_l + (long long)zo * p.pz + (2 * c) * p.py;
          if (p.wmap) {
            const float wgt = p.wmap[((long long)zo * p.H + yl + 2 * c) * p.W + xl];
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(long long)j * p.pc] += wgt * v[j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(long long)j * p.pc] = v[j];
          }
        }
      }
    }
  }
  if (OUTMODE == 0 && RangeCheck<T>::on) raise_flag(p.oflow, bad);
}

// Weights for conv3d_upcat16: fp32 w[16][48][27] (* folded gain) ->
//   [14 skip fragments (channels 0..15, the standard paired-tap layout)]
//   [8 parity classes][8 merged taps e = (ez,ey,ex)] fragments with K = 32 up-channels:
//   A[m][g*8+k] = sum over the original taps that land on low-res offset (p-1+e) per axis.
template <typename T>
__global__ void pack_upcat16_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ wpk) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = (kSteps + 64) * 512;
  if (idx >= total) return;
  const int e8 = idx & 7, lane = (idx >> 3) & 63, frag = idx >> 9;
  const int m = lane & 15, g = lane >> 4;
  const float sc = scale ? scale[m] : 1.f;
  float v = 0.f;
  if (frag < kSteps) {
    const int tap = (g >> 1) ? tapB_index(frag) : tapA_index(frag);
    const int cin = (g & 1) * 8 + e8;
    if (tap >= 0) v = w[(m * 48 + cin) * 27 + tap] * sc;
  } else {
    const int f = frag - kSteps;
    const int cls = f >> 3, e = f & 7;
    const int par[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};   // pz, py, px
    const int ee[3] = {e >> 2, (e >> 1) & 1, e & 1};
    const int cin = 16 + g * 8 + e8;
    // per axis: parity 0: e=0 -> {k=0}, e=1 -> {1,2};  parity 1: e=0 -> {0,1}, e=1 -> {2}
    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
      if (par[a] == 0) { lo[a] = ee[a] ? 1 : 0; hi[a] = ee[a] ? 2 : 0; }
      else { lo[a] = ee[a] ? 2 : 0; hi[a] = ee[a] ? 2 : 1; }
    }
    float sum = 0.f;
    for (int kz = lo[0]; kz <= hi[0]; ++kz)
      for (int ky = lo[1]; ky <= hi[1]; ++ky)
        for (int kx = lo[2]; kx <= hi[2]; ++kx) sum += w[(m * 48 + cin) * 27 + (kz * 3 + ky) * 3 + kx];
    v = sum * sc;
  }
  wpk[idx] = (T)v;
}

static thread_local char g_kernel_name5[64] = "";
const char* last_conv_upcat_kernel_name() { return g_kernel_name5; }

size_t conv_upcat16_packed_bytes() { return (size_t)(kSteps + 64) * 1024; }

bool conv_upcat16_eligible(const ConvParams& p) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_UPCAT") ? 1 : 0;
  return !off && !p.src0_f32c1 && p.up_shift == 1 && p.C0 == 16 && p.C1 == 32 && p.Cout == 16 && p.W >= 32 && p.H >= 8 && p.D >= 4 &&
         !(p.D & 1) && !(p.H & 1) && !(p.W & 1);
}

template <typename T, int OUTMODE>
static hipError_t launch_upcat_t(ConvParams p, hipStream_t st) {
  typedef UpcatCfg C;
  snprintf(g_kernel_name5, sizeof g_kernel_name5, "conv3d_upcat16<%s,2x8x32,c8+l3,r%d/%d,o%d>", __is_same(T, f16) ? "f16" : "bf16",
           C::R, C::RL, OUTMODE);
  auto kern = conv3d_upcat16_kernel<T, OUTMODE>;
  static amx::DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_once.set();
  }
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = exp_env("AMX_DBG");
    dbg = e ? atoi(e) : 0;
  }
  p.dbg = dbg;
  p.nby = (p.H + C::TY - 1) / C::TY;
  p.nbx = (p.W + C::TX - 1) / C::TX;
  const int tiles = p.nby * p.nbx * p.N;
  int nseg, zseg;
  pick_z_segments(tiles, p.D, 2, 256 * (C::LDS_BYTES <= 80 * 1024 ? 2 : 1), &zseg, &nseg);
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * nseg)), dim3((C::NC + C::NL) * 64), C::LDS_BYTES, st, p, zseg, nseg);
  return hipGetLastError();
}

// p.wpk must point at the upcat16 packing (pack_upcat16_kernel).
hipError_t launch_conv_upcat16(const ConvParams& p, int precision, hipStream_t st) {
  const bool planar = p.out32 != nullptr;
  if (precision == 0) return planar ? launch_upcat_t<f16, 1>(p, st) : launch_upcat_t<f16, 0>(p, st);
  return planar ? launch_upcat_t<bf16, 1>(p, st) : launch_upcat_t<bf16, 0>(p, st);
}

hipError_t launch_pack_upcat16(const float* w, const float* scale, void* wpk, int precision, hipStream_t st) {
  const int total = (kSteps + 64) * 512;
  if (precision == 0)
    hipLaunchKernelGGL(pack_upcat16_kernel<f16>, dim3((total + 255) / 256), dim3(256), 0, st, w, scale, (f16*)wpk);
  else
    hipLaunchKernelGGL(pack_upcat16_kernel<bf16>, dim3((total + 255) / 256), dim3(256), 0, st, w, scale, (bf16*)wpk);
  return hipGetLastError();
}

}  // namespace amx
