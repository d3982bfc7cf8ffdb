// anatomix_amd -- the UPSAMPLED segment of a decoder "upsample + concat + conv" block as its own launch:
//   P[class][l] = sum over the 2x2x2 low-res neighbourhood of merged weights . low[l + e + class - 1]
// (network.py:403-435: nn.Upsample(2,'nearest') -> torch.cat((skip, up), 1) -> nn.Conv3d(.., 3, reflect)).
//
// Why a second kernel.  A 3x3x3 convolution over a NEAREST-upsampled tensor only ever sees 2x2x2 distinct low-res voxels per
// output voxel: along an axis, output parity p = o & 1 reads low-res {l-1, l, l} (p = 0) or {l, l, l+1} (p = 1), l = o >> 1.
// Pre-summing the taps that hit the same voxel leaves 8 taps with parity-dependent weights instead of 27 -- 3.4x fewer MFMAs on
// the 2/3 of the input channels that come from the upsample (amx_conv3d_upcat.hip does this for the 48 -> 16 layer with all its
// weights in registers).  For the wider concat layers (96 -> 32, 192 -> 64, 384 -> 128) the eight weight sets do not fit
// registers next to the 27-tap skip weights, and on this chip the matrix pipes are POWER-limited (profiles/r03_power_ceiling.txt:
// ~1.5 PFLOP/s on random f16 operands with every CU busy, socket at its 1.4 kW cap), so executed MFMAs are what has to go.
// Hence the split:  the ordinary kernels run the 27-tap conv over the SKIP channels only, without bias or activation, into a
// scratch tensor (unchanged kernels, their tuned epilogues included);  this kernel then computes the merged-tap part at LOW
// resolution, all eight parity classes at once, adds the skip partial sums and the bias, activates and stores the layer's output
// (class (pz,py,px) owns the voxels (2lz+pz, 2ly+py, 2lx+px)).  The one extra rounding is that of the skip partial sums -- the
// smaller of the two parts (1/3 of the reduction).  (The other order was measured first: a partial tensor added inside the skip
// conv's epilogue cost the z-march kernel +14 .. +31 us on 96 -> 32 @64^3 -- a wave's loads retire behind its stores -- and
// rounds the larger part.)
//
// Formulation.  One workgroup = 8 waves = the 8 parity classes (pz,py,px); it owns a low-res brick TZ x TY x 16 cells (= TZ*TY
// tiles of 16 consecutive lx) and a group of 16*Q output channels.  MFMA v_mfma_f32_16x16x32: A = merged weights (16 couts x 32
// up-channels of ONE merged tap), B = 16 low-res voxels x those 32 channels.  A wave streams its class's 8 taps x Q A-fragments of
// the current 32-channel stage through a 4-tap window of REGISTERS (loaded straight from L2, four taps ahead of their use); the low-res halo of a stage ((TZ+2) x (TY+2) x 18
// voxels x 32 channels, replicate-clamped = reflect padding of the upsampled tensor) sits in LDS, plane-major like every other
// kernel here, fetched by LDS-DMA two stages ahead into a ring of buffers.  Reflect padding at high resolution is
// replicate padding at low resolution (-1 -> 1 -> low 0; N -> N-2 -> low N/2-1).
#include <stdio.h>
#include <stdlib.h>

#include "amx_device.h"

namespace amx {


template <int Q, int TZ, int TY, int NBUF, int KS, bool SPLIT = false, int LXT = 16>
struct UpmCfg {
  static_assert(LXT == 16 || LXT == 8, "a tile is 16 cells of one low-res row, or 8 cells of two rows (rows shorter than 16)");
  static constexpr int RPT = 16 / LXT;                   // rows per tile
  static constexpr int BY = TY * RPT;                    // brick rows
  static constexpr int NP = SPLIT ? 2 : 1;                // strict precision: hi and lo halves of every operand
  static_assert(!SPLIT || KS == 1, "strict precision: 32-channel stages");
  static constexpr int T = TZ * TY;                       // tiles per wave (= per class)
  static constexpr int HZ = TZ + 2, HY = BY + 2, HX = LXT + 2, HV = HZ * HY * HX;
  static constexpr int PL = ((HV * 16 + 255) / 256) * 256;   // one 8-channel plane of the halo
  static constexpr int BUF = 4 * KS * NP * PL;            // KS x 32 channels per stage (strict: hi planes, then lo planes)
  static constexpr int NJ = (HV + 63) / 64;               // DMA instructions per plane
  static constexpr int NDMA = 4 * KS * NP * NJ;
  static constexpr int PER_WAVE = (NDMA + 7) / 8;
  static constexpr int LDS_BYTES = NBUF * BUF;
  static_assert(LDS_BYTES <= 160 * 1024, "ring must fit the LDS");
};

// SPLIT (strict precision): every tensor holds [hi(C) | lo(C)] 16-bit channels per voxel (value = hi + lo), the packing holds Wh
// and Wl fragments per tap; the sweep multiplies Wh*xh + Wh*xl + Wl*xh into the same fp32 accumulators (amx_conv3d_v2.hip), the
// epilogue adds hi + lo of the skip partial sums and splits the result again.
template <typename T_, int Q, int TZ, int TY, int NBUF, int KS, bool SPLIT, int LXT>
__global__ __launch_bounds__(512) void conv3d_upmerge_kernel(const UpmergeParams p) {
  typedef UpmCfg<Q, TZ, TY, NBUF, KS, SPLIT, LXT> C;
  constexpr int RPT = C::RPT, BY = C::BY;
  constexpr int NP = C::NP;
  typedef typename Ops<T_>::vec8 vec8;
  constexpr int HY = C::HY, HX = C::HX, PL = C::PL, NT = C::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int lix = li & (LXT - 1), liy = LXT == 8 ? li >> 3 : 0;     // cell of the tile this lane multiplies
  const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;

  // ---- this workgroup's contiguous run of items (brick, cout group); XCD b % 8 gets a contiguous span
  const int nbricks = p.nbz * p.nby * p.nbx * p.N;
  const int ncg = p.Cout / (16 * Q);
  const int items = nbricks * ncg;
  const int G = gridDim.x;
  int jw = blockIdx.x;
  if ((G & 7) == 0) jw = (jw & 7) * (G >> 3) + (jw >> 3);
  const int it0 = (int)((long long)items * jw / G), it1 = (int)((long long)items * (jw + 1) / G);
  const int nstage = p.C1 / (32 * KS);
  const int T_total = (it1 - it0) * nstage;
  if (T_total <= 0) return;

  struct Item { int bx, by, bz, n, cg; };
  auto decode = [&](int it) {
    Item r;
    r.cg = it / nbricks;
    int b = it - r.cg * nbricks;
    r.bx = b % p.nbx; b /= p.nbx;
    r.by = b % p.nby; b /= p.nby;
    r.bz = b % p.nbz;
    r.n = b / p.nbz;
    return r;
  };

  // ---- LDS-DMA of one (item, stage): this wave's share of the 4 planes x NJ pieces; per-lane source offsets per item
  int dma_item = -1;
  int off[C::PER_WAVE];
  auto issue = [&](int it, int stage, int bsel) {
    const Item I = decode(it);
    if (it != dma_item) {
      dma_item = it;
#pragma unroll
      for (int m = 0; m < C::PER_WAVE; ++m) {
        const int id = wave + 8 * m;                       // piece id = plane * NJ + j
        const int j = id % C::NJ;
        const int hv = j * 64 + lane;
        const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        int gz = I.bz * TZ - 1 + hz, gy = I.by * BY - 1 + hy, gx = I.bx * LXT - 1 + hx;
        gz = gz < 0 ? 0 : (gz >= p.LD ? p.LD - 1 : gz);
        gy = gy < 0 ? 0 : (gy >= p.LH ? p.LH - 1 : gy);
        gx = gx < 0 ? 0 : (gx >= p.LW ? p.LW - 1 : gx);
        off[m] = hv < C::HV ? gz * (int)p.sz + gy * (int)p.sy + gx * (int)p.sx : -1;
      }
    }
    const char* base = p.src + (long long)I.n * p.sn;
    const int cs = p.cs ? p.cs : 32;                       // a voxel's 32-byte pieces: channels-last (32) or one row plane apart
    char* buf = smem + bsel * C::BUF;
#pragma unroll
    for (int m = 0; m < C::PER_WAVE; ++m) {
      const int id = wave + 8 * m;
      if (id < C::NDMA) {
        const int plane = id / C::NJ, j = id - plane * C::NJ;
        const int cbyte = stage * (64 * KS) + (SPLIT && plane >= 4 ? p.C1 * 2 + (plane - 4) * 16 : plane * 16);
        if (off[m] >= 0)
          dma16_asm(base + off[m] + (long long)(cbyte >> 5) * cs + (cbyte & 31),
                    (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(buf + plane * PL + j * 1024)));
      }
    }
  };

  // ---- this class's A fragments of one (item, stage)
  auto wptr = [&](int it, int stage) -> const char* {
    const int cg = it / nbricks;
    return p.wpk + ((((long long)cg * nstage * KS + stage * KS) * 8 + wave) * 8 * NP * Q) * 1024 + lane * 16;
  };
  // tap index e' = ks * 8 + e of a stage -> byte offset from wptr (the packing is [32-channel block][class][tap][q])
  auto woff = [](const int e2) { return ((e2 >> 3) * 64 + (e2 & 7)) * NP * Q * 1024; };
  constexpr int NTAP = 8 * KS;
  // A fragments stream through a window of 4 taps: tap e of a stage sits in wq[e % WIN]; after its last use the slot is refilled
  // with tap e + WIN (of this stage or the next one) -- WIN taps (>= ~1000 cycles of MFMAs) of cover for an L2 round trip.
  // Nothing is reused across stages, so holding all eight taps would only cost accumulator registers.
  constexpr int WIN = 4;
  vec8 wq[WIN][NP][Q];
  // ---- lane-constant LDS base: halo voxel (lz + pz + ez, ly + py + ey, li + px + ex), plane g
  const int lbase = g * PL + ((pz * HY + py + liy) * HX + px + lix) * 16;

  f32x4 acc[NT][Q];
  int nx_it = it0, nx_stage = 0;                           // next (item, stage) to fetch
  auto step_next = [&]() {
    if (++nx_stage == nstage) { nx_stage = 0; ++nx_it; }
  };
  issue(nx_it, nx_stage, 0);
  step_next();
  if (NBUF > 2 && T_total > 1) { issue(nx_it, nx_stage, 1); step_next(); }
  {
    const char* w0 = wptr(it0, 0);
#pragma unroll
    for (int e = 0; e < WIN; ++e)
#pragma unroll
      for (int h = 0; h < NP; ++h)
#pragma unroll
        for (int q = 0; q < Q; ++q) wq[e][h][q] = *(const vec8*)(w0 + woff(e) + (h * Q + q) * 1024);
  }

  int cu_it = it0, cu_stage = 0;
  bool bad = false;
  // deferred store of a finished item (full-resolution channels-last output, this class's voxels are every second one per axis): issued after the next stage's DMA, drains under its sweep
  unsigned pend[NT][2 * Q * NP];
  int pend_it = -1;
  // the skip conv's partial sums of the item being multiplied: requested in the item's last stage, after the previous item's
  // stores (whose registers they may take), so that the latency hides under the sweep
  unsigned padd[NT][2 * Q * NP];
  auto load_part = [&](const int it) {
    const Item I = decode(it);
    const int lx = I.bx * LXT + lix;
    const int ocs = p.ocs ? p.ocs : 32, cb2 = (I.cg * 16 * Q + g * 4 * Q) * 2;
    const long long sy = (long long)p.Cout * 2 * NP * (2 * p.LW), sz = sy * (2 * p.LH), sx = ocs == 32 ? (long long)p.Cout * 2 * NP : 32;
    const char* pb = p.part + (long long)I.n * sz * (2 * p.LD) + pz * sz + py * sy + (2 * lx + px) * sx + (long long)(cb2 >> 5) * ocs + (cb2 & 31);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int lz = I.bz * TZ + i / TY, ly = I.by * BY + (i % TY) * RPT + liy;
      const bool in = (lz < p.LD) & (ly < p.LH) & (lx < p.LW);
#pragma unroll
      for (int hl = 0; hl < NP; ++hl) {                     // strict: the lo channels follow Cout channels further on in the voxel
        const char* src = pb + 2 * lz * sz + 2 * ly * sy + hl * p.Cout * 2;
        unsigned* dst = &padd[i][hl * 2 * Q];
        if (Q == 1) {
          const uint2 v = in ? *(const uint2*)src : make_uint2(0u, 0u);
          dst[0] = v.x; dst[1] = v.y;
        } else {
#pragma unroll
          for (int h = 0; h < Q / 2; ++h) {
            const uint4 v = in ? *(const uint4*)(src + h * 16) : make_uint4(0u, 0u, 0u, 0u);
            dst[4 * h] = v.x; dst[4 * h + 1] = v.y; dst[4 * h + 2] = v.z; dst[4 * h + 3] = v.w;
          }
        }
      }
    }
  };
  auto flush = [&]() {
    const Item I = decode(pend_it);
    pend_it = -1;
    const int lx = I.bx * LXT + lix;
    const int ocs = p.ocs ? p.ocs : 32, cb2 = (I.cg * 16 * Q + g * 4 * Q) * 2;
    const long long sy = (long long)p.Cout * 2 * NP * (2 * p.LW), sz = sy * (2 * p.LH), sx = ocs == 32 ? (long long)p.Cout * 2 * NP : 32;
    char* pb = p.out + (long long)I.n * sz * (2 * p.LD) + pz * sz + py * sy + (2 * lx + px) * sx + (long long)(cb2 >> 5) * ocs + (cb2 & 31);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int lz = I.bz * TZ + i / TY, ly = I.by * BY + (i % TY) * RPT + liy;
      if (lz >= p.LD || ly >= p.LH || lx >= p.LW) continue;
#pragma unroll
      for (int hl = 0; hl < NP; ++hl) {
        char* dst = pb + 2 * lz * sz + 2 * ly * sy + hl * p.Cout * 2;     // output voxel (2 lz + pz, 2 ly + py, 2 lx + px)
        const unsigned* v = &pend[i][hl * 2 * Q];
        if (Q == 1) *(uint2*)dst = make_uint2(v[0], v[1]);
        else {
#pragma unroll
          for (int h = 0; h < Q / 2; ++h) *(uint4*)(dst + h * 16) = make_uint4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
        }
      }
    }
  };
  for (int t = 0; t < T_total; ++t) {
    // This wave's DMA pieces of stage t must have landed before the barrier.  Vector memory operations complete in order, and
    // tap 0's fragment was requested (during stage t - 1) AFTER those pieces: touching it makes hipcc place exactly the counted
    // wait that covers them, while the requests of taps 1-3 and of stage t + 1's halo stay in flight.
    if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("" ::"v"(wq[0][0][0]) : "memory");
    __syncthreads();                                       // everyone's pieces of stage t landed; buffer (t + 2) % NBUF is free
    if (t + NBUF - 1 < T_total && !(p.dbg & 1)) {
      issue(nx_it, nx_stage, (t + NBUF - 1) % NBUF);
      step_next();
    }
    if (pend_it >= 0 && !(p.dbg & 4)) flush();
    if (cu_stage == nstage - 1 && !(p.dbg & 16)) load_part(cu_it);
    const char* buf = smem + (t % NBUF) * C::BUF + lbase;
    if (cu_stage == 0) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // weights of the NEXT (item, stage) replace a tap's registers right after its last use
    const bool more = t + 1 < T_total;
    const int w_it = cu_stage + 1 < nstage ? cu_it : cu_it + 1, w_stage = cu_stage + 1 < nstage ? cu_stage + 1 : 0;
    const char* wn = more ? wptr(w_it, w_stage) : p.wpk;
    const char* wc = wptr(cu_it, cu_stage);
    if (!(p.dbg & 2)) {
      // B fragments in groups of GT tiles, one group ahead of the MFMAs (scheduling fences keep the order: left alone, hipcc
      // hoists every read of a tap -- 64 registers of fragments next to 128 of accumulators -- and spills)
      constexpr int GT = SPLIT ? 2 : (NT < 4 ? NT : 4), NGT = NT / GT, NG = NTAP * NGT;   // strict: two fragments (hi, lo) per tile
      vec8 fb[2][GT][NP];
      auto load_group = [&](const int gi, const int set) {
        const int e2 = gi / NGT, i0 = (gi % NGT) * GT;
        const int ks = e2 >> 3, ez = (e2 >> 2) & 1, ey = (e2 >> 1) & 1, ex = e2 & 1;
#pragma unroll
        for (int k = 0; k < GT; ++k) {
          const int lz = (i0 + k) / TY, ly = (i0 + k) % TY;
#pragma unroll
          for (int h = 0; h < NP; ++h)
            fb[set][k][h] = *(const vec8*)(buf + (ks + h) * 4 * PL + (((lz + ez) * HY + ly * RPT + ey) * HX + ex) * 16);
        }
      };
      load_group(0, 0);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const int e = gi / NGT, i0 = (gi % NGT) * GT;
        if (gi + 1 < NG) load_group(gi + 1, (gi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < GT; ++k)
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            acc[i0 + k][q] = Ops<T_>::mfma(wq[e % WIN][0][q], fb[gi & 1][k][0], acc[i0 + k][q]);
            if (SPLIT) {
              acc[i0 + k][q] = Ops<T_>::mfma(wq[e % WIN][0][q], fb[gi & 1][k][NP - 1], acc[i0 + k][q]);     // Wh * xl
              acc[i0 + k][q] = Ops<T_>::mfma(wq[e % WIN][NP - 1][q], fb[gi & 1][k][0], acc[i0 + k][q]);     // Wl * xh
            }
          }
        __builtin_amdgcn_sched_barrier(0);
        if ((gi % NGT) == NGT - 1 && (e + WIN < NTAP || more)) {   // last use of tap e: its slot takes tap e + WIN (next stage: - NTAP)
          const char* src = e + WIN < NTAP ? wc + woff(e + WIN) : wn + woff(e + WIN - NTAP);
#pragma unroll
          for (int h = 0; h < NP; ++h)
#pragma unroll
            for (int q = 0; q < Q; ++q) wq[e % WIN][h][q] = *(const vec8*)(src + (h * Q + q) * 1024);
        }
      }
    }
    if (++cu_stage < nstage) continue;
    cu_stage = 0;

    // ---- item finished: + skip partial sums + bias, activation, pack; stored under the NEXT stage's sweep (flush)
    {
      const int cgb = (cu_it / nbricks) * 16 * Q + g * 4 * Q;
      auto cvt = [](const unsigned w, const int half) { return (float)__builtin_bit_cast(T_, (unsigned short)(half ? w >> 16 : w & 0xffffu)); };
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const f32x4 bv = p.bias ? *(const f32x4*)(p.bias + cgb + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float a = cvt(padd[i][2 * q + (j >> 1)], j & 1);
            if (SPLIT) a += cvt(padd[i][2 * Q + 2 * q + (j >> 1)], j & 1);
            acc[i][q][j] += bv[j] + a;
          }
      }
      act_inplace<NT * Q>(&acc[0][0], p.act, p.slope);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (RangeCheck<T_>::on) bad |= RangeCheck<T_>::bad(acc[i][q][j]);
        pend[i][2 * q] = (unsigned)to_bits<T_>(acc[i][q][0]) | ((unsigned)to_bits<T_>(acc[i][q][1]) << 16);
        pend[i][2 * q + 1] = (unsigned)to_bits<T_>(acc[i][q][2]) | ((unsigned)to_bits<T_>(acc[i][q][3]) << 16);
        if (SPLIT) {
          float r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = acc[i][q][j] - (float)(T_)acc[i][q][j];
          pend[i][2 * Q + 2 * q] = (unsigned)to_bits<T_>(r[0]) | ((unsigned)to_bits<T_>(r[1]) << 16);
          pend[i][2 * Q + 2 * q + 1] = (unsigned)to_bits<T_>(r[2]) | ((unsigned)to_bits<T_>(r[3]) << 16);
        }
      }
    pend_it = cu_it;
    ++cu_it;
  }
  if (pend_it >= 0 && !(p.dbg & 4)) flush();
  if (RangeCheck<T_>::on) raise_flag(p.oflow, bad);
}

// Merged weights: fp32 w[Cout][CinTotal][27] (* folded gain), up-channels at [c_off, c_off + C1) ->
//   [cout group][stage = 32 up-channels][class (pz,py,px)][tap e = (ez,ey,ex)][q][lane][8],
//   row m of tile q = output channel cg*16Q + (m>>2)*4Q + q*4 + (m&3) (the layout every epilogue here assumes),
//   k = g*8 + j = up-channel stage*32 + g*8 + j;  value = sum of the original taps landing on low-res offset (class - 1 + e)
//   per axis, formed in fp32 and rounded ONCE.
template <typename T>
__global__ void pack_upmerge_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ wpk, int c_off,
                                    int CinTotal, int C1, int Cout, int Q, int split) {
  const int nstage = C1 / 32, NP = split ? 2 : 1;           // strict: [.. tap e][part: hi, lo][q]
  const long long total = (long long)(Cout / (16 * Q)) * nstage * 64 * NP * Q * 512;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63;
    long long r = idx >> 9;
    const int q = r % Q; r /= Q;
    const int part = r % NP; r /= NP;
    const int e = r % 8; r /= 8;
    const int cls = r % 8; r /= 8;
    const int stage = r % nstage;
    const int cg = (int)(r / nstage);
    const int m = lane & 15, g = lane >> 4;
    const int cout = cg * 16 * Q + (m >> 2) * 4 * Q + q * 4 + (m & 3);
    const int cin = c_off + stage * 32 + g * 8 + j;
    const int par[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};
    const int ee[3] = {e >> 2, (e >> 1) & 1, e & 1};
    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {   // parity 0: e=0 -> {k=0}, e=1 -> {1,2};  parity 1: e=0 -> {0,1}, e=1 -> {2}
      if (par[a] == 0) { lo[a] = ee[a] ? 1 : 0; hi[a] = ee[a] ? 2 : 0; }
      else { lo[a] = ee[a] ? 2 : 0; hi[a] = ee[a] ? 2 : 1; }
    }
    float sum = 0.f;
    for (int kz = lo[0]; kz <= hi[0]; ++kz)
      for (int ky = lo[1]; ky <= hi[1]; ++ky)
        for (int kx = lo[2]; kx <= hi[2]; ++kx) sum += w[((long long)cout * CinTotal + cin) * 27 + (kz * 3 + ky) * 3 + kx];
    const float v = sum * (scale ? scale[cout] : 1.f);
    const T vh = (T)v;
    wpk[idx] = part ? (T)(v - (float)vh) : vh;
  }
}

static thread_local char g_kernel_name6[64] = "";
const char* last_conv_upmerge_kernel_name() { return g_kernel_name6; }

size_t conv_upmerge_packed_bytes(int C1, int Cout, int split) { return (size_t)Cout * C1 * 64 * 2 * (split ? 2 : 1); }   // 8 classes x 8 taps per (cout, cin)
size_t conv_upmerge_partial_bytes(int N, int D, int H, int W, int Cout) { return (size_t)N * D * H * W * Cout * 2; }
int conv_upmerge_q(int Cout, int split) { return (!split && Cout % 32 == 0) ? 2 : 1; }

// the low-res tensor must be at least one tile wide; 32-channel stages
// (the 48 -> 16 layer has its own fused kernel in the 16-bit precisions, amx_conv3d_upcat.hip; in strict precision it comes here)
bool conv_upmerge_eligible(int C0, int C1, int Cout, int D, int H, int W, int up_shift, int split) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_UPMERGE") ? 1 : 0;
  return !off && up_shift == 1 && C0 >= 16 && C1 >= 32 && C1 % 32 == 0 && Cout >= (split ? 16 : 32) && Cout % 16 == 0 && W >= 16 && !(D & 1) && !(H & 1) &&
         !(W & 1) && D >= 4 && H >= 4;
}

static int g_num_cus6 = 0;

template <typename T, int Q, int TZ, int TY, int NBUF, int KS, bool SPLIT = false, int LXT = 16>
static hipError_t launch_upm(UpmergeParams p, hipStream_t st) {
  typedef UpmCfg<Q, TZ, TY, NBUF, KS, SPLIT, LXT> C;
  snprintf(g_kernel_name6, sizeof g_kernel_name6, "conv3d_upmerge<%s,q%d,%dx%dx%d,b%d,k%d>", __is_same(T, f16) ? (SPLIT ? "f16x2" : "f16") : (SPLIT ? "bf16x2" : "bf16"), Q,
           TZ, C::BY, LXT, NBUF, 32 * KS);
  auto kern = conv3d_upmerge_kernel<T, Q, TZ, TY, NBUF, KS, SPLIT, LXT>;
  static amx::DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_once.set();
  }
  if (g_num_cus6 == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    g_num_cus6 = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  p.nbz = (p.LD + TZ - 1) / TZ;
  p.nby = (p.LH + C::BY - 1) / C::BY;
  p.nbx = (p.LW + LXT - 1) / LXT;
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = exp_env("AMX_DBG");
    dbg = e ? atoi(e) : 0;
  }
  p.dbg = dbg;
  const long long items = (long long)p.nbz * p.nby * p.nbx * p.N * (p.Cout / (16 * Q));
  const long long grid = items < g_num_cus6 ? items : g_num_cus6;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), C::LDS_BYTES, st, p);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_upm_split(const UpmergeParams& p, hipStream_t st) {
  if (p.LW < 16) return launch_upm<T, 1, 2, 2, 3, 1, true, 8>(p, st);        // rows shorter than a 16-cell tile: 8 cells x 2 rows
  const long long cells = (long long)p.N * p.LD * p.LH * ((p.LW + 15) / 16);
  if (cells / 8 * (p.Cout / 16) >= 256) return launch_upm<T, 1, 2, 4, 2, 1, true>(p, st);   // two buffers: hi + lo planes are 55 KB
  return launch_upm<T, 1, 2, 2, 3, 1, true>(p, st);
}

template <typename T>
static hipError_t launch_upm_t(const UpmergeParams& p, hipStream_t st) {
  const int Q = conv_upmerge_q(p.Cout, 0);
  if (p.LW < 16) {                                                             // rows shorter than a 16-cell tile: 8 cells x 2 rows
    if (Q == 2) return p.C1 % 64 == 0 ? launch_upm<T, 2, 2, 2, 2, 2, false, 8>(p, st) : launch_upm<T, 2, 2, 2, 3, 1, false, 8>(p, st);
    return launch_upm<T, 1, 2, 2, 3, 1, false, 8>(p, st);
  }
  const long long cells = (long long)p.N * p.LD * p.LH * ((p.LW + 15) / 16);       // tiles of 16 cells
  const long long groups = p.Cout / (16 * Q);
  // bricks of 8 / 4 tiles at Q = 2 (16 tiles = 128 accumulator registers spill next to the fragment buffers), 16 / 8 at Q = 1: the
  // largest one that still gives every compute unit an item.  64-channel stages (two buffers) where the channel count allows:
  // half the barriers per item.
  const bool k64 = p.C1 % 64 == 0;
  if (Q == 2) {
    if (cells / 8 * groups >= 256) return launch_upm<T, 2, 2, 4, 3, 1>(p, st);   // (64-channel stages spill 8 registers here)
    return k64 ? launch_upm<T, 2, 2, 2, 2, 2>(p, st) : launch_upm<T, 2, 2, 2, 3, 1>(p, st);
  }
  if (cells / 16 * groups >= 256) return launch_upm<T, 1, 2, 8, 3, 1>(p, st);
  return k64 ? launch_upm<T, 1, 2, 4, 2, 2>(p, st) : launch_upm<T, 1, 2, 4, 3, 1>(p, st);
}

hipError_t launch_conv_upmerge(const UpmergeParams& p, int precision, hipStream_t st) {
  if (precision == 0) return launch_upm_t<f16>(p, st);
  if (precision == 1) return launch_upm_t<bf16>(p, st);
  if (precision == 2) return launch_upm_split<f16>(p, st);
  if (precision == 3) return launch_upm_split<bf16>(p, st);
  return hipErrorInvalidValue;
}

hipError_t launch_pack_upmerge(const float* w, const float* scale, void* wpk, int c_off, int CinTotal, int C1, int Cout, int precision,
                               hipStream_t st) {
  const int split = precision >= 2;
  const int Q = conv_upmerge_q(Cout, split);
  const long long total = (long long)Cout * C1 * 64 * (split ? 2 : 1);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if ((precision & 1) == 0)
    hipLaunchKernelGGL(pack_upmerge_kernel<f16>, dim3(blocks), dim3(256), 0, st, w, scale, (f16*)wpk, c_off, CinTotal, C1, Cout, Q, split);
  else
    hipLaunchKernelGGL(pack_upmerge_kernel<bf16>, dim3(blocks), dim3(256), 0, st, w, scale, (bf16*)wpk, c_off, CinTotal, C1, Cout, Q, split);
  return hipGetLastError();
}

}  // namespace amx
