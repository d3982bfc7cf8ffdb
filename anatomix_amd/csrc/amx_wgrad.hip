// anatomix_amd -- weight gradient of nn.Conv3d(k3, stride 1, padding='same', padding_mode='reflect')
// (the wgrad half of the backward of anatomix/model/network.py:310-461, needed by the contrastive step,
// pretraining/models/supcl_model.py:603-661).
//
//   dW[co][ci][kz][ky][kx] = sum over (n, z, y, x) of  dY[n][z][y][x][co] * In[n][r(z+kz-1)][r(y+ky-1)][r(x+kx-1)][ci]
//   (r = reflect; In may be the concatenation of a full-resolution segment and a nearest-upsampled half-resolution
//    segment, exactly as the forward kernels read it).
//
// A GEMM whose reduction index is the VOXEL: D[co][ci] += A[co][k] B[k][ci] per tap with k = 32 voxels, v_mfma_f32_16x16x32.
// A lane of an MFMA operand holds 8 consecutive k of ONE channel, but the tensors are channels-last: the transposition is done by
// the LDS read itself (gfx950 `ds_read_b64_tr_b16`), see below.  Reflect padding and the nearest upsample of the concat convs are
// resolved when a tile is staged (per-lane source addresses of the LDS-DMA).  Partials [chunk][pair][27][16][16] fp32 are added by
// wgrad_reduce_kernel with a fixed summation tree: deterministic, no atomics.
// (History: channels-last tiles and 16-bit gathers, 98 LDS reads + ~140 VALU per 27 MFMAs; then tiles transposed at staging with
// 32-bit LDS writes, a z-ring and a two-deep register prefetch -- 6000 cycles per 512-voxel item whatever the layer, because the
// register copy of the prefetch forced `vmcnt(0)` every item and the transposed writes cost as much LDS time as the sweep.  The form
// below: 1028 -> 685 us over the 13 conv shapes of the 6 M UNet's step, every shape faster, tools/wgrad_layers.py.)
#include <stdlib.h>

#include "amx_device.h"

namespace amx {

// ---------------------------------------------------------------------------------------------------------------------
// Transpose-read form (gfx950 `ds_read_b64_tr_b16`).  The tiles stay CHANNELS-LAST in LDS -- 32 bytes per voxel and channel
// tile, exactly as they lie in memory -- so they are staged by LDS-DMA (global_load_lds_dwordx4, no registers, no VALU, no
// ds_write) by two loader waves, and the MFMA operands (8 consecutive voxels of one channel per lane) are formed by the
// LDS itself: a 16-lane group that points its lanes at the four 8-byte channel quads of four consecutive voxels (128
// contiguous bytes) receives channel (lane & 15) of those four voxels (checked in tools/ubench/tr_read_check.hip).  Per
// K-block (32 voxels x 27 taps): 2 reads for dY, 4 reads per (kz, ky) for the input, 27 MFMAs; a wave's two K-blocks are
// vertically adjacent rows and share their halo reads where the geometry allows (W > 16).
//   * bank conflicts: the two groups served together (lanes 0-31) must lie in different 128-byte halves of the 256-byte
//     bank row; with 8 voxels (256 B) per group they would not, so the 128-byte blocks of a row are stored swizzled,
//     position = block ^ ((block >> 1) & 1) -- free: the DMA lane that fills a position simply fetches the swizzled source;
//   * work: a workgroup owns one (cout tile, cin tile) pair and a contiguous range of the planes of the (n, y tile, x tile,
//     z) sequence; 8 MFMA waves split the K-blocks of an item (G planes), keep their 27 accumulators for the whole range
//     and are summed through LDS at the end (same partial layout / reduce kernels as before);
//   * pipeline: ring of 3G + 2 input planes and 3G dY planes; the loaders run up to two items ahead of the MFMA waves,
//     paced by counted vmcnt waits and two LDS counters (`ready` per loader, `done` per MFMA wave) -- no barrier in the march;
//   * workgroup ids are dealt to the 8 XCDs round-robin: an XCD gets a CONTIGUOUS range of the (chunk, pair) list, so
//     neighbouring tiles' shared halo rows and a chunk's dY (read once per cin tile) meet in one L2, and no XCD is handed
//     more than its 32 workgroups (one per compute unit).
// Geometry by row width: (TY, TX, G) = (8, 64, 1) for W > 32, (16, 32, 1), (16, 16, 2), (8, 8, 4) for W <= 8.
// Measured (tools/wgrad_ab.sh, 16 -> 16 @128^3 x 2, 66-70 us): DMA alone 50 us, MFMA sweep alone 58 us, neither (launch, flags, the
// cross-wave sum, the reduce launch) 22 us; the sweep runs at 64 % MFMA-busy, paced by the per-wave issue rate (~6 instructions per
// MFMA), zero bank conflicts (SQ_LDS_BANK_CONFLICT); counter traffic 1.09x the algorithmic bytes (tools/wgrad_traffic.sh).
template <int TY_, int TX_, int G_>
struct WtCfg {
  static constexpr int TY = TY_, TX = TX_, G = G_;
  static constexpr int RSPAN = TX >= 32 ? 1 : 32 / TX;               // rows a K-block spans
  static constexpr int XBN = TX >= 32 ? TX / 32 : 1;                 // K-blocks side by side in x
  static constexpr int NKB = TY * TX / 32;                           // K-blocks per plane
  static constexpr int KPW = (G * NKB + 7) / 8;                      // K-blocks per MFMA wave and item
  static constexpr bool SHARE = TX >= 32 && G == 1 && NKB == 16;     // a wave's two K-blocks are rows r0, r0 + 1 of one x block: shared halo rows
  static constexpr bool SHAREZ = TX == 16 && G == 2 && NKB == 8;     // ... the same rows of planes z, z + 1: shared input planes
  static constexpr int IRB = TX == 8 ? 5 : TX / 4 + 1;               // 128-byte blocks per halo row (odd for TX = 8: groups in different rows)
  static constexpr int DRB = TX == 8 ? 3 : TX / 4;
  static constexpr int NDI = ((TY + 2) * IRB * 128 + 1023) / 1024;   // DMA instructions (1 KiB each) per input plane
  static constexpr int NDD = (TY * DRB * 128 + 1023) / 1024;
  static constexpr int ISZ = NDI * 1024, DSZ = NDD * 1024;
  static constexpr int RI = 3 * G + 2, RD = 3 * G;
  static constexpr int C = G == 1 ? 2 : 1;                           // load groups ahead of the first item (input planes z-1, z)
  static constexpr int LAG = G == 1 ? 4 : G == 2 ? 3 : 2;            // group j may be issued once item j - LAG is done
  static constexpr int NL = 2;                                       // loader waves
  static constexpr int NTI = (NDI + NL - 1) / NL, NTD = (NDD + NL - 1) / NL;
  static constexpr int FLAGS = RI * ISZ + RD * DSZ;
  static constexpr int LDS = FLAGS + 64;
  static constexpr int RED = 4 * 27 * 64 * 16;
  static_assert(LDS <= 160 * 1024 && RED <= FLAGS, "LDS budget");
  static_assert(G * NKB % 8 == 0 || G * NKB < 8, "K-blocks of an item split evenly over the 8 MFMA waves");
};

typedef __attribute__((ext_vector_type(4))) short amx_s16x4;
__device__ __forceinline__ amx_u32x2 lds_read_tr16(unsigned addr) {
  typedef __attribute__((address_space(3))) amx_s16x4* lp_t;
  return __builtin_bit_cast(amx_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)addr));
}

#ifdef AMX_EXPERIMENT
#define AMX_WT_DBG p.dbg
#else
#define AMX_WT_DBG 0
#endif
template <typename T, typename C>
__global__ __launch_bounds__(640) void conv3d_wgrad_tr_kernel(const WgradParams p) {
  typedef typename Ops<T>::vec8 vec8;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int TY = C::TY, TX = C::TX, G = C::G, IRB = C::IRB, DRB = C::DRB, RI = C::RI, RD = C::RD, ISZ = C::ISZ, DSZ = C::DSZ;
  char* ins = smem;
  char* dys = smem + RI * ISZ;
  int* ready = (int*)(smem + C::FLAGS);                             // [NL]
  int* done = ready + 8;                                            // [8], 32-byte aligned
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int ncit = (p.C0 + p.C1) / 16, npairs = (p.Cout / 16) * ncit;
  const int wq = (blockIdx.x & 7) * p.cpx + (blockIdx.x >> 3);       // XCD x works on entries [x * cpx, (x + 1) * cpx) of the (chunk, pair) list
  if (wq >= p.nchunk * npairs) return;
  const int chunk = wq / npairs, pair = wq - chunk * npairs;
  const int cot = pair / ncit, cit = pair % ncit;
  const bool seg1 = cit * 16 >= p.C0;
  const int ci0 = seg1 ? cit * 16 - p.C0 : cit * 16;
  const int sh = seg1 ? p.up_shift : 0;

  if (tid < 16) ready[tid] = 0;
  __syncthreads();

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int P0 = chunk * p.ppc, P1 = P0 + p.ppc < p.nplanes ? P0 + p.ppc : p.nplanes;
  int Jbase = 0, Sbase = 0;

  if (wave >= 8) {
    // =========================================== loader wave l ===========================================
    const int l = wave - 8;
    const unsigned a_ready = lds_addr(ready + l), a_done = lds_addr(done);
    // every loader wave issues exactly NTI + NTD instructions per plane (an instruction index past the plane repeats the wave's
    // last one: same source, same destination), so the counted waits are compile-time constants and the issue code is branch-free
    constexpr int PER = G * (C::NTI + C::NTD);
    int kki[C::NTI], kkd[C::NTD];
#pragma unroll
    for (int t = 0; t < C::NTI; ++t) kki[t] = l + C::NL * t < C::NDI ? l + C::NL * t : l + C::NL * (t - 1);
#pragma unroll
    for (int t = 0; t < C::NTD; ++t) kkd[t] = l + C::NL * t < C::NDD ? l + C::NL * t : l + C::NL * (t - 1);
    for (int P = P0; P < P1;) {
      const int tile = P / p.D, za = P - tile * p.D;
      const int zb = za + (P1 - P) < p.D ? za + (P1 - P) : p.D;
      const int xt = tile % p.nxt, yt = (tile / p.nxt) % p.nyt, n = tile / (p.nxt * p.nyt);
      const int y0 = yt * TY, x0 = xt * TX;
      const int nz = zb - za, ni = (nz + G - 1) / G, ngr = ni + C::C;
      // per-lane source offsets of this tile (z-independent): unit u = 16 bytes at LDS offset 16 u of the plane image
      int offi[C::NTI], offd[C::NTD];
      const long long sy = seg1 ? p.s1y : p.s0y, sx = seg1 ? p.s1x : p.s0x;
#pragma unroll
      for (int t = 0; t < C::NTI; ++t) {
        const int kk = l + C::NL * t < C::NDI ? l + C::NL * t : l + C::NL * (t - 1);
        const int u = kk * 64 + lane, bp = u >> 3, jv = (u >> 1) & 3, hf = u & 1;
        int rho = bp / IRB;
        const int bs = bp - rho * IRB, b = bs ^ ((bs >> 1) & 1), h = 4 * b + jv;
        rho = rho > TY + 1 ? TY + 1 : rho;
        const int yy = reflect_clamp(y0 + rho - 1, p.H) >> sh, xx = reflect_clamp(x0 + h - 1, p.W) >> sh;
        offi[t] = (int)(yy * sy + xx * sx) + ci0 * 2 + hf * 16;
      }
#pragma unroll
      for (int t = 0; t < C::NTD; ++t) {
        const int kk = l + C::NL * t < C::NDD ? l + C::NL * t : l + C::NL * (t - 1);
        const int u = kk * 64 + lane, bp = u >> 3, jv = (u >> 1) & 3, hf = u & 1;
        int r = bp / DRB;
        const int bs = bp - r * DRB, b = bs ^ ((bs >> 1) & 1);
        r = r > TY - 1 ? TY - 1 : r;
        int yy = y0 + r, xx = x0 + 4 * b + jv;
        yy = yy < p.H ? yy : p.H - 1;
        xx = xx < p.W ? xx : p.W - 1;
        offd[t] = (int)(yy * p.yy + xx * p.yx) + cot * 32 + hf * 16;
      }
      const char* src_n = seg1 ? p.src1 + (long long)n * p.s1n : p.src0 + (long long)n * p.s0n;
      const long long sz = seg1 ? p.s1z : p.s0z;
      const char* dy_n = p.dy + (long long)n * p.yn;
      auto issue_group = [&](int j) {
#pragma unroll
        for (int gz = 0; gz < G; ++gz) {
          const int q = j * G + gz;                                 // input plane sequence number: z = za - 1 + q
          const char* plane = src_n + (long long)(reflect_clamp(za - 1 + q, p.D) >> sh) * sz;
          char* dst = ins + (q % RI) * ISZ;
#pragma unroll
          for (int t = 0; t < C::NTI; ++t)
            if (!(AMX_WT_DBG & 2))
              __builtin_amdgcn_global_load_lds((gptr_t)(plane + offi[t]), (lptr_t)(dst + kki[t] * 1024), 16, 0, 0);
          int d = (j - C::C) * G + gz;                              // dY plane of item j - C (before the first item: plane 0 again)
          d = d < 0 ? 0 : d;
          const int zz = za + d < p.D ? za + d : p.D - 1;
          const char* dpl = dy_n + (long long)zz * p.yz;
          char* ddst = dys + (d % RD) * DSZ;
#pragma unroll
          for (int t = 0; t < C::NTD; ++t)
            if (!(AMX_WT_DBG & 2))
              __builtin_amdgcn_global_load_lds((gptr_t)(dpl + offd[t]), (lptr_t)(ddst + kkd[t] * 1024), 16, 0, 0);
        }
      };
      // the ring restarts with every run of planes: the previous run must be consumed
      while (__builtin_amdgcn_readfirstlane(flag_min8_asm(a_done)) < Sbase) __builtin_amdgcn_s_sleep(2);
      int next_issue = 0, next_pub = 0;
      while (next_pub < ngr) {
        if (next_issue < ngr) {
          const int md = __builtin_amdgcn_readfirstlane(flag_min8_asm(a_done)) - Sbase;
          int lim = md + C::LAG + 1;
          lim = lim < ngr ? lim : ngr;
          while (next_issue < lim) issue_group(next_issue++);
        }
        if (next_issue == next_pub) {
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        if (AMX_WT_DBG & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (next_issue - next_pub - 1 > 63 / PER) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");   // (6-bit field: stricter than needed)
        else WaitVm<PER, 63 / PER>::run(next_issue - next_pub - 1);     // the oldest unpublished group has landed
        flag_store_asm(a_ready, Jbase + ++next_pub);
      }
      Jbase += ngr; Sbase += ni; P += nz;
    }
  } else {
    // ============================================ MFMA wave ============================================
    const int li = lane & 15, g = lane >> 4;
    const int gx = TX >= 32 ? g : (TX == 16 ? (g & 1) : 0), gr = TX >= 32 ? 0 : (TX == 16 ? (g >> 1) : g);
    const int odd = gx & 1;                                         // parity of the group's 8-voxel block: decides the swizzled positions
    const int ib = (li >> 2) * 32 + (li & 3) * 8;
    // input reads of a row: P = halo x0 .. x0+3, Q = x0+4 .. x0+7 (kx = 0 as they are) and the same two voxels further, P2 = x0+2 ..
    // x0+5, Q2 = x0+6 .. x0+9 (kx = 2 as they are; kx = 1 is four v_alignbit of P, Q and Q2's second register).  Operand tuples must
    // be even-aligned, so forming kx = 2 from P, Q and one more read costs four v_mov per row -- and the VALU port, which the MFMAs
    // share, is what paces this kernel (measured: 4.9 VALU per MFMA, 35 % MFMA-busy); a fourth read is free beside that.
    // Blocks 2m, 2m+1, 2m+2 of the lane group sit at positions pos0, pos1, pos2 (swizzle); a lane's voxel of P2 / Q2 lies in the
    // first (j < 2) or the second of the two blocks the shifted quad straddles.
    const int jv_ = li >> 2, q4_ = li & 3;
    const unsigned inb = lds_addr(ins) + (gr * IRB + 2 * gx) * 128;
    const unsigned pos0 = odd ? 128 : 0, pos1 = odd ? 0 : 128, pos2 = odd ? 256 : 384;
    const unsigned inP = inb + pos0 + ib, inQ = inb + pos1 + ib;
    const unsigned sh2 = ((jv_ + 2) & 3) * 32 + q4_ * 8;
    const unsigned inP2 = inb + (jv_ < 2 ? pos0 : pos1) + sh2, inQ2 = inb + (jv_ < 2 ? pos1 : pos2) + sh2;
    const unsigned dyb = lds_addr(dys) + (gr * DRB + 2 * gx) * 128 + ib;
    const unsigned dyP = dyb + (odd ? 128 : 0), dyQ = dyb + (odd ? 0 : 128);
    const bool ragged = (p.H % TY) != 0 || (p.W % TX) != 0;
    for (int P = P0; P < P1;) {
      const int tile = P / p.D, za = P - tile * p.D;
      const int zb = za + (P1 - P) < p.D ? za + (P1 - P) : p.D;
      const int xt = tile % p.nxt, yt = (tile / p.nxt) % p.nyt;
      const int y0 = yt * TY, x0 = xt * TX;
      const int nz = zb - za, ni = (nz + G - 1) / G, ngr = ni + C::C;
      for (int it = 0; it < ni; ++it) {
        const int need = Jbase + it + C::C + 1;
        while (flag_min2(ready) < need) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        // A fragments and ring offsets of the wave's K-blocks first, then their 9 rows each (a row = one (kz, ky): four reads, four
        // v_alignbit, three MFMAs; the order of reads and MFMAs is the scheduler's -- pinning the reads two rows ahead measured slower).
        if constexpr (C::SHARE) {
          // rows r0, r0 + 1 of one x block: the four halo rows r0 .. r0 + 3 of a kz are read ONCE and feed both rows (row r0 takes
          // halo row hr as ky = hr, row r0 + 1 as ky = hr - 1): 16 reads + 16 v_alignbit per 18 MFMAs instead of 24 + 24
          const int xb = wave % C::XBN, r0 = (wave / C::XBN) * 2, zl = it;
          vec8 afr[2];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const unsigned ab = (zl % RD) * DSZ + ((r0 + rr) * DRB + 8 * xb) * 128;
            amx_u32x2 a0 = lds_read_tr16(dyP + ab), a1 = lds_read_tr16(dyQ + ab);
            if (ragged) {
              const bool rowok = y0 + r0 + rr < p.H;
              const int xv = x0 + xb * 32 + gx * 8;
              unsigned m[4];
#pragma unroll
              for (int e = 0; e < 4; ++e)
                m[e] = rowok ? ((xv + 2 * e < p.W ? 0xffffu : 0u) | (xv + 2 * e + 1 < p.W ? 0xffff0000u : 0u)) : 0u;
              a0[0] &= m[0]; a0[1] &= m[1]; a1[0] &= m[2]; a1[1] &= m[3];
            }
            const unsigned av[4] = {a0[0], a0[1], a1[0], a1[1]};
            afr[rr] = __builtin_bit_cast(vec8, av);
          }
          if (!(AMX_WT_DBG & 1)) {
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) {
              const unsigned o = ((zl + kz) % RI) * ISZ + (r0 * IRB + 8 * xb) * 128;
              const unsigned vP = inP + o, vQ = inQ + o, vP2 = inP2 + o, vQ2 = inQ2 + o;
#pragma unroll
              for (int hr = 0; hr < 4; ++hr) {
                const amx_u32x2 P_ = lds_read_tr16(vP + hr * IRB * 128), Q_ = lds_read_tr16(vQ + hr * IRB * 128);
                const amx_u32x2 P2 = lds_read_tr16(vP2 + hr * IRB * 128), Q2 = lds_read_tr16(vQ2 + hr * IRB * 128);
                const unsigned b0[4] = {P_[0], P_[1], Q_[0], Q_[1]};
                const unsigned b1[4] = {__builtin_amdgcn_alignbit(P_[1], P_[0], 16), __builtin_amdgcn_alignbit(Q_[0], P_[1], 16),
                                        __builtin_amdgcn_alignbit(Q_[1], Q_[0], 16), __builtin_amdgcn_alignbit(Q2[1], Q_[1], 16)};
                const unsigned b2[4] = {P2[0], P2[1], Q2[0], Q2[1]};
                if (hr < 3) {
                  const int t0 = (kz * 3 + hr) * 3;
                  acc[t0] = Ops<T>::mfma(afr[0], __builtin_bit_cast(vec8, b0), acc[t0]);
                  acc[t0 + 1] = Ops<T>::mfma(afr[0], __builtin_bit_cast(vec8, b1), acc[t0 + 1]);
                  acc[t0 + 2] = Ops<T>::mfma(afr[0], __builtin_bit_cast(vec8, b2), acc[t0 + 2]);
                }
                if (hr > 0) {
                  const int t0 = (kz * 3 + hr - 1) * 3;
                  acc[t0] = Ops<T>::mfma(afr[1], __builtin_bit_cast(vec8, b0), acc[t0]);
                  acc[t0 + 1] = Ops<T>::mfma(afr[1], __builtin_bit_cast(vec8, b1), acc[t0 + 1]);
                  acc[t0 + 2] = Ops<T>::mfma(afr[1], __builtin_bit_cast(vec8, b2), acc[t0 + 2]);
                }
              }
            }
          }
        } else if constexpr (C::SHAREZ) {
          // the same K-block (two rows of 16) in planes z, z + 1 of the item: input planes z - 1 .. z + 2 are read once and feed both
          // (plane offset u is kz = u for plane z, kz = u - 1 for plane z + 1)
          const int r = wave * C::RSPAN, zl0 = it * G;
          const bool two = zl0 + 1 < nz;                            // (wave-uniform; a last odd plane has no partner)
          vec8 afz[2];
#pragma unroll
          for (int pz = 0; pz < 2; ++pz) {
            const int zl = two ? zl0 + pz : zl0;
            const unsigned ab = (zl % RD) * DSZ + (r * DRB) * 128;
            amx_u32x2 a0 = lds_read_tr16(dyP + ab), a1 = lds_read_tr16(dyQ + ab);
            if (ragged) {
              const bool rowok = y0 + r + gr < p.H;
              const int xv = x0 + gx * 8;
              unsigned m[4];
#pragma unroll
              for (int e = 0; e < 4; ++e)
                m[e] = rowok ? ((xv + 2 * e < p.W ? 0xffffu : 0u) | (xv + 2 * e + 1 < p.W ? 0xffff0000u : 0u)) : 0u;
              a0[0] &= m[0]; a0[1] &= m[1]; a1[0] &= m[2]; a1[1] &= m[3];
            }
            const unsigned vm = (pz == 0 || two) ? 0xffffffffu : 0u;
            const unsigned av[4] = {a0[0] & vm, a0[1] & vm, a1[0] & vm, a1[1] & vm};
            afz[pz] = __builtin_bit_cast(vec8, av);
          }
          if (!(AMX_WT_DBG & 1)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (u == 3 && !two) continue;                         // plane z + 2 has not been loaded for a lone last plane
              const unsigned o = ((zl0 + u) % RI) * ISZ + (r * IRB) * 128;
              const unsigned vP = inP + o, vQ = inQ + o, vP2 = inP2 + o, vQ2 = inQ2 + o;
#pragma unroll
              for (int ky = 0; ky < 3; ++ky) {
                const amx_u32x2 P_ = lds_read_tr16(vP + ky * IRB * 128), Q_ = lds_read_tr16(vQ + ky * IRB * 128);
                const amx_u32x2 P2 = lds_read_tr16(vP2 + ky * IRB * 128), Q2 = lds_read_tr16(vQ2 + ky * IRB * 128);
                const unsigned b0[4] = {P_[0], P_[1], Q_[0], Q_[1]};
                const unsigned b1[4] = {__builtin_amdgcn_alignbit(P_[1], P_[0], 16), __builtin_amdgcn_alignbit(Q_[0], P_[1], 16),
                                        __builtin_amdgcn_alignbit(Q_[1], Q_[0], 16), __builtin_amdgcn_alignbit(Q2[1], Q_[1], 16)};
                const unsigned b2[4] = {P2[0], P2[1], Q2[0], Q2[1]};
                if (u < 3) {
                  const int t0 = (u * 3 + ky) * 3;
                  acc[t0] = Ops<T>::mfma(afz[0], __builtin_bit_cast(vec8, b0), acc[t0]);
                  acc[t0 + 1] = Ops<T>::mfma(afz[0], __builtin_bit_cast(vec8, b1), acc[t0 + 1]);
                  acc[t0 + 2] = Ops<T>::mfma(afz[0], __builtin_bit_cast(vec8, b2), acc[t0 + 2]);
                }
                if (u > 0) {
                  const int t0 = ((u - 1) * 3 + ky) * 3;
                  acc[t0] = Ops<T>::mfma(afz[1], __builtin_bit_cast(vec8, b0), acc[t0]);
                  acc[t0 + 1] = Ops<T>::mfma(afz[1], __builtin_bit_cast(vec8, b1), acc[t0 + 1]);
                  acc[t0 + 2] = Ops<T>::mfma(afz[1], __builtin_bit_cast(vec8, b2), acc[t0 + 2]);
                }
              }
            }
          }
        } else {
        vec8 af[C::KPW];
        unsigned ibs[C::KPW][3];
        bool any = false;
#pragma unroll
        for (int t = 0; t < C::KPW; ++t) {
          const int kb = wave + 8 * t;
          const int gz = kb / C::NKB, kbp = kb - gz * C::NKB;
          const int r = (kbp / C::XBN) * C::RSPAN, xb = kbp % C::XBN;
          int zl = it * G + gz;
          const bool valid = kb < G * C::NKB && zl < nz;            // (wave-uniform)
          any |= valid;
          if (!valid) zl = it * G;                                  // an invalid K-block reads the item's first plane (data that has landed) against zeros
          const unsigned ab = (zl % RD) * DSZ + (r * DRB + 8 * xb) * 128;
          amx_u32x2 a0 = lds_read_tr16(dyP + ab), a1 = lds_read_tr16(dyQ + ab);
          if (ragged) {                                             // voxels outside the volume contribute nothing
            const bool rowok = y0 + r + gr < p.H;
            const int xv = x0 + xb * 32 + gx * 8;
            unsigned m[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              m[e] = rowok ? ((xv + 2 * e < p.W ? 0xffffu : 0u) | (xv + 2 * e + 1 < p.W ? 0xffff0000u : 0u)) : 0u;
            a0[0] &= m[0]; a0[1] &= m[1]; a1[0] &= m[2]; a1[1] &= m[3];
          }
          const unsigned vm = valid ? 0xffffffffu : 0u;
          const unsigned av[4] = {a0[0] & vm, a0[1] & vm, a1[0] & vm, a1[1] & vm};
          af[t] = __builtin_bit_cast(vec8, av);
#pragma unroll
          for (int kz = 0; kz < 3; ++kz) ibs[t][kz] = ((zl + kz) % RI) * ISZ + (r * IRB + 8 * xb) * 128;
        }
        if (any && !(AMX_WT_DBG & 1)) {
#pragma unroll
          for (int t = 0; t < C::KPW; ++t)
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) {
              const unsigned vP = inP + ibs[t][kz], vQ = inQ + ibs[t][kz], vP2 = inP2 + ibs[t][kz], vQ2 = inQ2 + ibs[t][kz];
#pragma unroll
              for (int ky = 0; ky < 3; ++ky) {
                const amx_u32x2 P_ = lds_read_tr16(vP + ky * IRB * 128), Q_ = lds_read_tr16(vQ + ky * IRB * 128);
                const amx_u32x2 P2 = lds_read_tr16(vP2 + ky * IRB * 128), Q2 = lds_read_tr16(vQ2 + ky * IRB * 128);
                const unsigned b0[4] = {P_[0], P_[1], Q_[0], Q_[1]};                   // kx = 0
                const unsigned b1[4] = {__builtin_amdgcn_alignbit(P_[1], P_[0], 16), __builtin_amdgcn_alignbit(Q_[0], P_[1], 16),
                                        __builtin_amdgcn_alignbit(Q_[1], Q_[0], 16), __builtin_amdgcn_alignbit(Q2[1], Q_[1], 16)};
                const unsigned b2[4] = {P2[0], P2[1], Q2[0], Q2[1]};                   // kx = 2
                const int t0 = (kz * 3 + ky) * 3;
                acc[t0] = Ops<T>::mfma(af[t], __builtin_bit_cast(vec8, b0), acc[t0]);
                acc[t0 + 1] = Ops<T>::mfma(af[t], __builtin_bit_cast(vec8, b1), acc[t0 + 1]);
                acc[t0 + 2] = Ops<T>::mfma(af[t], __builtin_bit_cast(vec8, b2), acc[t0 + 2]);
              }
            }
        }
        }
        asm volatile("" ::: "memory");                              // LDS serves a wave's accesses in order: the flag lands behind the reads
        if (lane == 0) flag_store(done + wave, Sbase + it + 1);
      }
      Jbase += ngr; Sbase += ni; P += nz;
    }
  }
  // ---- sum the 8 MFMA waves through LDS (3 rounds), wave 0 writes the partial
  const int m = lane & 15, kg = lane >> 4;
  float* red = (float*)smem;                                        // [4 waves][27][64][4] floats = 108 KiB
  for (int half = 4; half >= 1; half >>= 1) {
    __syncthreads();
    if (wave >= half && wave < 2 * half)
#pragma unroll
      for (int t = 0; t < 27; ++t) *(f32x4*)(red + (((size_t)(wave - half) * 27 + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    if (wave < half)
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const f32x4 o = *(const f32x4*)(red + (((size_t)wave * 27 + t) * 64 + lane) * 4);
        acc[t] = acc[t] + o;
      }
  }
  if (p.nchunk == 1) {
    // the only chunk: dW itself, through LDS so that the stores follow dW's layout (a cout row of the tile is 16 x 27 contiguous floats)
    __syncthreads();
    if (wave == 0)
#pragma unroll
      for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[((kg * 4 + j) * 16 + m) * 27 + t] = acc[t][j];     // [co][ci][tap]
    __syncthreads();
    const int cin = p.cin_real, cib = cit * 16;
    const int nci = cin - cib < 16 ? cin - cib : 16;                // real input channels of this tile (the rest is padding)
    for (int e = tid; e < 16 * nci * 27; e += 640) {
      const int co = e / (nci * 27), rest = e - co * (nci * 27);
      float* o = p.dw + ((size_t)(cot * 16 + co) * cin + cib) * 27 + rest;
      const float v = red[co * 16 * 27 + rest];
      *o = p.accumulate ? *o + v : v;
    }
  } else if (wave == 0) {
    float* out = p.partial + ((size_t)chunk * npairs + pair) * 27 * 256;
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) out[t * 256 + (kg * 4 + j) * 16 + m] = acc[t][j];      // [tap][co][ci]
  }
}

// dW[co][ci][tap] (+)= sum over chunks.  A block owns 8 consecutive elements of the partial layout
// [pair][tap][co16][ci16] (64 contiguous bytes per chunk); its 16 lanes per element add chunks l, l+16, ... in order and the
// 16 lane sums are added in lane order: a fixed summation tree, independent of the launch.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cout,
                                                          int Cin, int CinPad, int nchunk, int accumulate) {
  __shared__ float red[16][17];
  const int ncit = CinPad / 16, npairs = (Cout / 16) * ncit;
  const long long E = (long long)npairs * 27 * 256;
  const int el = threadIdx.x & 15, l = threadIdx.x >> 4;     // 16 consecutive elements (one 64-byte segment) x 16 chunk lanes
  const long long e = (long long)blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < E) {
    int c = l;
    for (; c + 48 < nchunk; c += 64) {                        // four independent loads in flight, added in chunk order
      const float v0 = partial[(size_t)c * E + e], v1 = partial[(size_t)(c + 16) * E + e];
      const float v2 = partial[(size_t)(c + 32) * E + e], v3 = partial[(size_t)(c + 48) * E + e];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; c < nchunk; c += 16) s += partial[(size_t)c * E + e];
  }
  red[el][l] = s;
  __syncthreads();
  if (threadIdx.x < 16 && (long long)blockIdx.x * 16 + threadIdx.x < E) {
    const long long ee = (long long)blockIdx.x * 16 + threadIdx.x;
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[threadIdx.x][k];
    const int ci16 = ee % 16, co16 = (ee / 16) % 16, tap = (ee / 256) % 27, pair = ee / (256 * 27);
    const int co = (pair / ncit) * 16 + co16, ci = (pair % ncit) * 16 + ci16;
    if (ci < Cin) {
      float* o = dw + ((long long)co * Cin + ci) * 27 + tap;
      *o = accumulate ? *o + t : t;
    }
  }
}

// nchunk <= 16 (the wide layers: 16 or more channel-tile pairs share the 256 workgroups): one thread per element walks the chunks in
// order.  The same sums as the kernel above -- there every chunk lane holds at most one chunk and the lane sums are added in lane
// order -- without 27 648 blocks of 16 elements each (75 -> ~10 us on 128 -> 128).
__global__ __launch_bounds__(256) void wgrad_reduce_few_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cout,
                                                              int Cin, int CinPad, int nchunk, int accumulate) {
  const int ncit = CinPad / 16, npairs = (Cout / 16) * ncit;
  const long long E = (long long)npairs * 27 * 256;
  const long long ee = (long long)blockIdx.x * 256 + threadIdx.x;
  if (ee >= E) return;
  float v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = c < nchunk ? partial[(size_t)c * E + ee] : 0.f;
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) t += v[c];
  const int ci16 = ee % 16, co16 = (ee / 16) % 16, tap = (ee / 256) % 27, pair = ee / (256 * 27);
  const int co = (pair / ncit) * 16 + co16, ci = (pair % ncit) * 16 + ci16;
  if (ci < Cin) {
    float* o = dw + ((long long)co * Cin + ci) * 27 + tap;
    *o = accumulate ? *o + t : t;
  }
}

// Transpose-read kernel: geometry class by row width and the split of the (tile, z) plane sequence over the workgroups.
static int wt_class(int W) { return W <= 8 ? 3 : W <= 16 ? 2 : W <= 32 ? 1 : 0; }
static void wt_geometry(int cls, int* ty, int* tx, int* g) {
  static const int T[4][3] = {{8, 64, 1}, {16, 32, 1}, {16, 16, 2}, {8, 8, 4}};
  *ty = T[cls][0]; *tx = T[cls][1]; *g = T[cls][2];
}
struct WtPlan { int cls, nyt, nxt, nplanes, ppc, nchunk, cpx; };
static WtPlan wt_plan(int N, int D, int H, int W, int Cout, int CinPad) {
  WtPlan q;
  int ty, tx, g;
  q.cls = wt_class(W);
  wt_geometry(q.cls, &ty, &tx, &g);
  q.nyt = (H + ty - 1) / ty;
  q.nxt = (W + tx - 1) / tx;
  q.nplanes = N * q.nyt * q.nxt * D;
  const int npairs = (Cout / 16) * (CinPad / 16);
  // one workgroup per compute unit (LDS), ONE round of at most 256 workgroups, every workgroup a contiguous run of planes
  static int target = -1;
  if (target < 0) target = exp_env("AMX_WGRAD_WGS") ? atoi(exp_env("AMX_WGRAD_WGS")) : 256;
  int nc = target / npairs;
  if (nc < 1) nc = 1;
  if (q.nplanes <= 4 * g) nc = 1;              // at most four items in all: one workgroup per pair writes dW itself (no partials, no reduce launch)
  int ppc = (q.nplanes + nc - 1) / nc;
  ppc = (ppc + g - 1) / g * g;                 // whole items
  q.ppc = ppc;
  q.nchunk = (q.nplanes + ppc - 1) / ppc;
  q.cpx = (q.nchunk * npairs + 7) / 8;         // (chunk, pair) entries per XCD: at most 32, one workgroup per compute unit
  return q;
}

size_t wgrad_scratch_bytes(int N, int D, int H, int W, int Cout, int CinPad) {
  const WtPlan q = wt_plan(N, D, H, W, Cout, CinPad);
  return (size_t)q.nchunk * (Cout / 16) * (CinPad / 16) * 27 * 256 * sizeof(float);
}

template <typename T, typename C>
static hipError_t launch_wt(const WgradParams& p, int nwg, hipStream_t st) {
  static DeviceOnce once;
  if (!once.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3d_wgrad_tr_kernel<T, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    once.set();
  }
  hipLaunchKernelGGL((conv3d_wgrad_tr_kernel<T, C>), dim3(nwg), dim3(640), C::LDS, st, p);
  return hipGetLastError();
}

hipError_t launch_wgrad(WgradParams p, int CinReal, float* dw, int accumulate, void* scratch, int precision, hipStream_t st) {
  const int CinPad = p.C0 + p.C1;
  if (p.W > 128) return hipErrorInvalidValue;
  const WtPlan q = wt_plan(p.N, p.D, p.H, p.W, p.Cout, CinPad);
  p.partial = (float*)scratch;
  p.nchunk = q.nchunk; p.nyt = q.nyt; p.nxt = q.nxt; p.ppc = q.ppc; p.cpx = q.cpx; p.nplanes = q.nplanes;
  p.dw = dw; p.cin_real = CinReal; p.accumulate = accumulate;
  p.dbg = exp_env("AMX_WGRAD_DBG") ? atoi(exp_env("AMX_WGRAD_DBG")) : 0;
  const int npairs = (p.Cout / 16) * (CinPad / 16);
  const int nwg = 8 * q.cpx;
  hipError_t e;
#define AMX_WT(T)                                                                                                            \
  (q.cls == 0 ? launch_wt<T, WtCfg<8, 64, 1>>(p, nwg, st) : q.cls == 1 ? launch_wt<T, WtCfg<16, 32, 1>>(p, nwg, st)          \
   : q.cls == 2 ? launch_wt<T, WtCfg<16, 16, 2>>(p, nwg, st) : launch_wt<T, WtCfg<8, 8, 4>>(p, nwg, st))
  e = precision == 0 ? AMX_WT(f16) : AMX_WT(bf16);
#undef AMX_WT
  if (e != hipSuccess) return e;
  const int nchunk = q.nchunk;
  const long long E = (long long)npairs * 27 * 256;
  if (nchunk == 1) return hipGetLastError();                        // written by the kernel
  if (nchunk <= 16)
    hipLaunchKernelGGL(wgrad_reduce_few_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, (const float*)scratch, dw, p.Cout,
                       CinReal, CinPad, nchunk, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((E + 15) / 16)), dim3(256), 0, st, (const float*)scratch, dw, p.Cout,
                       CinReal, CinPad, nchunk, accumulate);
  return hipGetLastError();
}

}  // namespace amx
