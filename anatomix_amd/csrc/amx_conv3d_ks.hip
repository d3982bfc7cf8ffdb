// anatomix_amd -- conv3d 3x3x3 reflect, third-generation kernel for the DEEP levels (<= 32^3): weights stationary in REGISTERS,
// the K dimension split over the waves of a workgroup, fixed-order reduction through LDS.
//
// Why (profiles/r04_layers_6m.txt): at the 32^3 .. 8^3 levels conv3d_k3_v2 spends its time FILLING LDS -- every
// 16-channel stage of every brick re-stages 28 .. 56 KiB of packed weight fragments next to ~12 .. 20 KiB of halo, the waves of a
// workgroup split the brick's voxels and all read the same weight fragments from LDS (0.75 .. 1 LDS read per MFMA), and a stage is
// only 28 .. 112 MFMAs per wave between two workgroup barriers: 7 .. 26 % of the MFMA peak.  This kernel turns the roles around:
//   * a workgroup owns ONE (cout group of 16 Q channels, K slice of 16 CPW KW input channels) and keeps those weights -- the MFMA
//     A operand -- in the REGISTERS of its waves for its whole life (Q * 14 * CPW fragments = up to 224 VGPRs per wave: the
//     register file is the largest on-chip memory, 512 KiB per CU).  Weights never touch LDS and are fetched once per workgroup;
//   * the KW waves of a team split K: wave k owns the input-channel chunks k CPW .. k CPW + CPW - 1 of the slice and sweeps the
//     WHOLE brick (NVT column tiles x Q cout tiles = 16 .. 32 accumulators), so that a B fragment read from LDS feeds Q MFMAs and
//     the LDS carries 1 / Q reads per MFMA;
//   * each wave DMAs the halo planes of ITS chunks into two private LDS buffers (double-buffered over the (brick, chunk) sequence):
//     no wave ever reads bytes another wave staged, so the sweep needs no barrier and no flags -- only the issuing wave's own
//     vmcnt;
//   * the partial sums of the KW waves meet in an LDS scratch tile after the sweep: wave k finalises the column tiles
//     k OWN .. k OWN + OWN - 1 and adds the partials in the FIXED order wave 0, 1, .., KW - 1 (deterministic; no atomics), then
//     applies bias + activation and stores 16-bit channels-last voxels exactly like conv3d_k3_v2;
//   * layers with more (cout group, K slice) pairs than voxels to keep 256 workgroups busy (the 8^3 level) split K across
//     workgroups as well (PART): fp32 partial tensors per slice, summed in slice order by splitk_reduce_kernel (+ bias,
//     activation, range check, 16-bit store).
//   * CW > 1 (the 64-input-channel layers): TWO waves per SIMD.  With one 400-register wave per SIMD nothing hides a stall -- the
//     LDS-DMA issue of the next halo (~100 cycles per KiB instruction, the wave is blocked), the reduction, the LDS latency of a B
//     fragment all idle the matrix pipe (measured: 20 k cycles per brick for 7.2 k of MFMA issue).  The cout tiles of the
//     workgroup are therefore split over CW waves per K position: wave (k, c) keeps Q tiles x one chunk = 112 VGPRs of weights, the
//     CW waves of a K position share that chunk's halo buffers (each issues 1 / CW of its DMA) and reduce independently; the
//     existing reduction barriers order "DMA landed" and "buffer free" for the shared buffers, so still no flags.
// Same arithmetic formulation as the other conv kernels (amx_conv3d.hip): A = packed weights [cout group][chunk][step 14][q][lane][8],
// B = activations from a plane-major halo image, 14 paired-tap steps per 16 channels; reflect padding resolved in the gather.
// Replaces nn.Conv3d(k=3, padding_mode='reflect') + folded eval BatchNorm3d + ReLU of /root/reference/anatomix/model/network.py:334-445
// at the levels below 64^3.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "amx_device.h"

namespace amx {

// Q: cout tiles per WAVE; QP: tiles per cout group of the packed weights (conv_pick_q); CPW: chunks per wave; KW: K positions
// (waves splitting the slice's chunks); CW: waves splitting the workgroup's cout tiles; TEAMS: independent bricks in flight; NH: tile groups
// NBUF: chunk buffers per K position.  1: no halo prefetch, half the LDS -- TWO workgroups per CU (<= 80 KiB, <= 256 registers): their
// phases drift apart, so one sweeps while the other waits for its halo or reduces (what the barrier-coupled waves of ONE workgroup
// cannot do: all of them are in the same phase)
template <int TZ_, int TY_, int TX_, int Q_, int QP_, int CPW_, int KW_, int CW_, int TEAMS_, int NH_, int NBUF_ = 2>
struct KsCfg {
  static constexpr int TZ = TZ_, TY = TY_, TX = TX_, Q = Q_, QP = QP_, CPW = CPW_, KW = KW_, CW = CW_, TEAMS = TEAMS_, NH = NH_, NBUF = NBUF_;
  static constexpr int WGS_PER_CU = NBUF == 1 ? 2 : 1;
  static constexpr int NWAVE = KW * CW * TEAMS;
  // A 16-voxel column tile is one row of 16 voxels (TX >= 16) or, for 8-wide bricks, the rows (z, y, x0 .. x0 + 7) and
  // (z + 2, y, ...): ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md,
  // LDS), i.e. voxels {0-3, 12-15} of one 8-channel plane with voxels {4-11} of the other, and the two planes sit on the same
  // 16-byte slots mod 16.  With the second row at + HX = 10 slots (rows y, y + 1: the first form) voxels 12-15 fall on the slots of
  // voxels 0-1 -- a 2-way conflict in every group, SQ_LDS_BANK_CONFLICT = 46 % of the LDS cycles at the 8^3 level; a second row at
  // 8 slots mod 16 makes all sixteen distinct, and two z planes of the 6 x 10 halo are 120 = 8 (mod 16) slots apart.
  static constexpr int LX = TX >= 16 ? 16 : 8;
  static constexpr bool ZPAIR = LX == 8;
  static constexpr int XT = TX / LX, YT = TY, ZT = ZPAIR ? TZ / 2 : TZ;
  static constexpr int NVT = ZT * YT * XT;                 // 16-voxel column tiles per brick (every wave sweeps all of them)
  __host__ __device__ static constexpr int tile_z(int i) { return ZPAIR ? (i >> 1) * 4 + (i & 1) : i; }   // first z plane of tile row i
  static constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  static constexpr int PLANE = (HV * 16 + 255) / 256 * 256;   // one 8-channel plane of the halo image
  static constexpr int CHBUF = 2 * PLANE;                  // the halo of one 16-channel chunk
  static constexpr int NPK = (HV + 63) / 64;               // LDS-DMA instructions per plane (64 consecutive halo voxels each)
  static constexpr int NPKW = (NPK + CW - 1) / CW;         // ... of which one wave issues every CW-th
  static constexpr int HALO_BYTES = TEAMS * KW * NBUF * CHBUF;   // NBUF chunk buffers per (team, K position), shared by its CW waves
  // NH > 1: the brick's column tiles are swept in NH consecutive groups out of the same staged halo, each group reduced and stored
  // before the next starts (fewer accumulator tiles next to the stationary weights: the 32-tile form of Q = 4 spilled 20-40
  // registers to scratch memory, reloaded in every reduction)
  static constexpr int NVTG = NVT / NH;                    // column tiles per group
  static constexpr int OWN = NVTG / KW;                    // column tiles of a group a wave finalises
  static constexpr int SCR_MAX = 160 * 1024 / WGS_PER_CU - HALO_BYTES;
  static constexpr int pick_rt() {                         // tiles per owner and reduction round that fit the scratch
    for (int r = OWN; r >= 1; --r)
      if (OWN % r == 0 && TEAMS * CW * KW * (KW - 1) * r * Q * 1024 <= SCR_MAX) return r;
    return 0;
  }
  static constexpr int RT = pick_rt();
  static constexpr int ROUNDS = RT ? OWN / RT : 0;
  static constexpr int SCR_GROUP = KW * (KW - 1) * RT * Q * 1024;   // one (team, cout wave) reduction group
  static constexpr int SCR_BYTES = TEAMS * CW * SCR_GROUP;
  static constexpr int LDS_BYTES = HALO_BYTES + SCR_BYTES;
  static_assert(NVT % NH == 0 && NVTG % KW == 0, "every wave finalises the same number of column tiles of every group");
  static_assert(NH == 1 || CPW == 1, "tile groups re-read the stage's halo: one chunk per wave and brick");
  static_assert(CW == 1 || CPW == 1, "shared halo buffers are ordered by the per-brick reduction barriers: one stage per brick");
  static_assert(RT >= 1, "the reduction scratch must fit behind the halo buffers");
  static_assert(TX % LX == 0 && (!ZPAIR || TZ % 4 == 0), "the brick must tile into 16-voxel columns");
  static_assert(Q * kSteps * CPW * 4 <= (NWAVE * WGS_PER_CU > 4 ? 112 : 224), "the stationary A fragments must leave room for accumulators and B fragments");
  static_assert(NBUF == 2 || (NBUF == 1 && CPW == 1 && CW == 1), "the un-prefetched form stages one chunk per wave and brick, privately");
  static_assert((CW * Q) % QP == 0 || QP % Q == 0, "a wave's cout tiles lie inside one packed cout group");
};

struct KsBrick {   // wave-uniform
  int n, z0, y0, x0;
};

template <typename T, typename C, bool PART>
__global__ __launch_bounds__(C::NWAVE * 64, C::NWAVE * C::WGS_PER_CU / 4) void conv3d_k3_ks_kernel(const ConvParams p) {
  typedef typename Ops<T>::vec8 vec8;
  constexpr int Q = C::Q, QP = C::QP, CPW = C::CPW, KW = C::KW, CW = C::CW, TEAMS = C::TEAMS, NVTG = C::NVTG, NH = C::NH;
  constexpr int HY = C::HY, HX = C::HX, PLANE = C::PLANE, CHBUF = C::CHBUF, NPK = C::NPK, NPKW = C::NPKW;
  constexpr int LX = C::LX, XT = C::XT, YT = C::YT, OWN = C::OWN, RT = C::RT, ROUNDS = C::ROUNDS;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave / (KW * CW), k = (wave - team * (KW * CW)) / CW, cw = wave % CW;   // wave = (team, K position, cout wave)
  const int li = lane & 15, g = lane >> 4;

  // ---- this workgroup's (cout group, K slice) and its contiguous run of bricks; XCD b % 8 gets a contiguous span of the
  //      (slice, cout group, run) order, so workgroups that share weights / halos share an L2
  const int ncg = p.Cout / (16 * Q * CW);                    // cout blocks of a workgroup: CW waves x Q tiles
  const int S = p.kslices > 0 ? p.kslices : 1;
  const int combos = ncg * S;
  const int G = gridDim.x, wpc = G / combos;
  int L = blockIdx.x;
  if ((G & 7) == 0) L = (L & 7) * (G >> 3) + (L >> 3);
  const int combo = L / wpc, run = L - combo * wpc;
  const int slice = combo / ncg, cg = combo - slice * ncg;
  const int nbricks = p.nbz * p.nby * p.nbx * p.N;
  const int b0 = (int)((long long)nbricks * run / wpc), b1 = (int)((long long)nbricks * (run + 1) / wpc);
  const int niter = (b1 - b0 + TEAMS - 1) / TEAMS;
  if (niter <= 0) return;
  const int nchunk = p.C0 >> 4;
  const int ch0 = slice * (KW * CPW) + k * CPW;             // this wave's first input-channel chunk
  const int cs = p.cs0 ? p.cs0 : 32;                        // bytes between a voxel's 16-channel chunks
  // this wave's Q cout tiles inside the packed weights [cout group of QP tiles][chunk][step][q][lane][8]
  const int tile0 = (cg * CW + cw) * Q;
  const int pg = tile0 / QP, q0 = tile0 - pg * QP;

  // ---- lane-constant LDS read bases (relative to a chunk buffer), as in conv3d_k3_v2 with the wave at the brick's origin
  const int dzl = C::ZPAIR ? (li >> 3) * 2 : 0;             // the lane's voxel inside its column tile
  const int dx = C::ZPAIR ? (li & 7) : li;
  const int lanebase = (g & 1) * PLANE + (dzl * HY * HX + dx) * 16;
  const int hi = g >> 1;
  const int base_d1 = lanebase + hi * 16;
  const int base_dx = lanebase + hi * 16 * HX;
  const int base_dz = lanebase + hi * 16 * HX * HY;
  const int base_d0 = lanebase;

  // ---- DMA lane constants: instruction j of a plane covers halo voxels 64 j .. 64 j + 63
  //      (the CW waves of a K position issue every CW-th instruction of the shared chunk buffer)
  int pk_off[NPKW];                                         // byte offset of the lane's halo voxel inside the sample; -1: past the halo
#pragma unroll
  for (int m = 0; m < NPKW; ++m) pk_off[m] = 0;
  char* const halo = smem + (team * KW + k) * (C::NBUF * CHBUF);
  char* const scratch = smem + C::HALO_BYTES + (team * CW + cw) * C::SCR_GROUP;

  auto decode = [&](int b) {                                // brick index -> sample and origin
    KsBrick r;
    const int bx = b % p.nbx;
    b /= p.nbx;
    const int by = b % p.nby;
    b /= p.nby;
    const int bz = b % p.nbz;
    r.n = b / p.nbz;
    r.z0 = bz * C::TZ;
    r.y0 = by * C::TY;
    r.x0 = bx * C::TX;
    return r;
  };
  auto offsets = [&](const KsBrick& br) {                   // per-lane source offsets of the brick's halo voxels inside the sample
#pragma unroll
    for (int m = 0; m < NPKW; ++m) {
      const int hv = (m * CW + cw) * 64 + lane;             // (divisions by constants; recomputed per brick instead of kept in registers)
      const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
      const int gz = reflect_clamp(br.z0 + hz - 1, p.D), gy = reflect_clamp(br.y0 + hy - 1, p.H);
      const int gx = reflect_clamp(br.x0 + hx - 1, p.W);
      pk_off[m] = hv < C::HV ? gz * (int)p.s0z + gy * (int)p.s0y + gx * (int)p.s0x : -1;
    }
  };
#define AMX_DMA16(src, dst) dma16_asm((const void*)(src), (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(dst)))
  auto issue = [&](const KsBrick& br, int cp, int bsel) {   // halo of chunk ch0 + cp of brick br -> chunk buffer bsel
    const char* base = p.src0 + (long long)br.n * p.s0n + (long long)(ch0 + cp) * cs;
    char* dst = halo + bsel * CHBUF;
#pragma unroll
    for (int m = 0; m < NPKW; ++m) {
      const int j = m * CW + cw;                            // uniform
      if (j < NPK && pk_off[m] >= 0) {
        AMX_DMA16(base + pk_off[m], dst + j * 1024);
        AMX_DMA16(base + pk_off[m] + 16, dst + PLANE + j * 1024);
      }
    }
  };

  // ablation switches and the optional cycle trace exist in -DAMX_EXPERIMENT builds only (AMX_DBG / AMX_TRACE=1: wave 0 of each
  // workgroup stamps s_memtime per phase); the product kernel carries neither the branches nor their registers
#ifdef AMX_EXPERIMENT
  const int dbg = p.dbg;
  unsigned long long* trace = (dbg & 8) ? (unsigned long long*)p.stats + (long long)blockIdx.x * 64 : nullptr;
  int tcount = 0;
#define AMX_STAMP()                                                                                      \
  do {                                                                                                   \
    if (trace && wave == 0 && lane == 0 && tcount < 64) trace[tcount] = __builtin_readcyclecounter();    \
    ++tcount;                                                                                            \
  } while (0)
#else
  constexpr int dbg = 0;
#define AMX_STAMP() do {} while (0)
#endif
  AMX_STAMP();

  // ---- L2 warm-up of this workgroup's weights.  Every workgroup of a (cout group, slice) streams the SAME packed range, all of
  //      them at kernel start: each line is then a first touch (HBM / Infinity-Cache latency, ~1-2 us) for every CU at once, and a
  //      CU keeps only ~13 KiB in flight -- 110 .. 224 KiB per workgroup arrived at ~5 B/clk/CU (10 us of an 18 .. 25 us layer).
  //      Here the workgroups sharing a range each touch a DIFFERENT 32 KiB part of it first (one dword per 128-byte line and lane,
  //      value discarded), so the whole range is on its way into the XCD's L2 after one round trip and the fragment loads below
  //      find it there (L2 hits: ~45 B/clk/CU).
  int sink = 0;                                             // destination of the touch: must stay allocated until the load has returned
  {
    // the slice's chunks of this wave's packed cout group are one contiguous range; the waves of all workgroups sharing it take
    // its 8 KiB parts (64 lines per wave instruction) round robin
    const long long range_bytes = (long long)KW * CPW * kSteps * QP * 1024;
    const char* wrange = p.wpk + ((long long)pg * nchunk + slice * (KW * CPW)) * (kSteps * QP * 1024);
    const int nparts = (int)((range_bytes + 8191) >> 13);
    const long long off = (long long)((run * C::NWAVE + wave) % nparts) * 8192 + lane * 128;
    if (off < range_bytes && !(dbg & 32)) asm volatile("global_load_dword %0, %1, off" : "=v"(sink) : "v"(wrange + off) : "memory");
  }

  const int cb = pg * 16 * QP + g * 4 * QP + q0 * 4;        // this lane's first output channel (4 Q consecutive ones: packing order)

  // ---- first brick's halo, then the stationary A fragments (in flight together)
  KsBrick cu = decode(b0 + team), nx = cu;
  bool cu_valid = b0 + team < b1;
  if (cu_valid) {
    offsets(cu);
    issue(cu, 0, 0);
  }
  vec8 wa[CPW][kSteps][Q];
  {
    const char* wbase = p.wpk + ((((long long)pg * nchunk + ch0) * kSteps) * QP + q0) * 1024 + lane * 16;
#pragma unroll
    for (int cp = 0; cp < CPW; ++cp)
#pragma unroll
      for (int s = 0; s < kSteps; ++s)
#pragma unroll
        for (int q = 0; q < Q; ++q) wa[cp][s][q] = *(const vec8*)(wbase + ((cp * kSteps + s) * QP + q) * 1024);
  }
  // The first brick's halo was issued BEFORE the fragment loads and vector memory loads return in order: once at most the NA
  // fragment loads are outstanding it has landed.  The fragments themselves are waited for by the compiler's own
  // counted vmcnt as the first sweep reaches them (the LDS-DMA issued meanwhile is invisible to it and only makes those waits
  // stricter), so the first sweep starts while most of the weights are still on their way.
  constexpr int NA = Q * kSteps * CPW;
  if (dbg & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
  static_assert(NA <= 63, "vmcnt is a 6-bit counter");
  asm volatile("" ::"v"(sink));                             // (the touch is older than the halo DMA: returned by now)
  if (CW > 1) __syncthreads();                              // the other cout waves' share of the first halo has landed too
  AMX_STAMP();

  f32x4 acc[NVTG][Q];
  int t = 0;                                                // stage counter of the brick's first chunk: chunk buffer (t + cp) & 1
  for (int it = 0; it < niter; ++it) {
    // the brick after this one (its halo offsets replace the current ones once the current brick's last chunk is on its way)
    const int bn = b0 + (it + 1) * TEAMS + team;
    const bool nx_valid = it + 1 < niter && bn < b1;
    if (nx_valid) nx = decode(bn);
    const bool full = cu_valid && (cu.z0 + C::TZ <= p.D) & (cu.y0 + C::TY <= p.H) & (cu.x0 + C::TX <= p.W);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
      for (int cp = 0; cp < CPW; ++cp) {
        // ---- prefetch the next stage of this wave into the other buffer (its previous reader, stage t + cp - 1, is done)
        if (h == 0 && C::NBUF == 1) {
          // un-prefetched: this brick's halo is requested now (the first one was requested in the prologue) and waited for; the CU's
          // other workgroup has the matrix pipes meanwhile
          if (it > 0 && cu_valid && !(dbg & 1)) {
            offsets(cu);
            issue(cu, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        } else if (h == 0) {
          if (dbg & 1) {                                    // ablation: no halo DMA after the first stage
          } else if (cp + 1 < CPW) {
            if (cu_valid) issue(cu, cp + 1, (t + cp + 1) & 1);
          } else if (nx_valid) {
            offsets(nx);
            issue(nx, 0, (t + cp + 1) & 1);
          }
        }
        AMX_STAMP();
        if (cp == 0) {
#pragma unroll
          for (int c = 0; c < NVTG; ++c)
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- MFMA sweep of chunk cp over the tiles of group h: 14 paired-tap steps
        if (cu_valid && !(dbg & 2)) {
          const char* buf = halo + (C::NBUF == 1 ? 0 : ((t + cp) & 1) * CHBUF);
          // B fragments through a small register ring over the flattened (step, column tile) sequence: item u + P is requested
          // before the Q MFMAs of item u (a fragment feeds Q MFMAs = 16 Q cycles of the matrix pipe; P items cover the LDS latency).
          constexpr int P = Q >= 4 ? 4 : 6, R = P + 1, NU = kSteps * NVTG;
          vec8 fb[R];
          auto load_item = [&](const int u) {
            const int s = u / NVTG, c = h * NVTG + (u - s * NVTG);
            const int kz = s < 9 ? s / 3 : (s < 12 ? s - 9 : (s == 12 ? 0 : 2));
            const int ky = s < 9 ? s % 3 : (s < 12 ? 0 : 2);
            const int kx = s < 9 ? 0 : 2;
            const int tapoff = ((kz * HY + ky) * HX + kx) * 16;
            const int bsel = s < 9 ? base_d1 : (s < 12 ? base_dx : (s == 12 ? base_dz : base_d0));
            const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
            const int coff = ((C::tile_z(cz) * HY + cy) * HX + cx * LX) * 16;
            fb[u % R] = *(const vec8*)(buf + bsel + tapoff + coff);
          };
#pragma unroll
          for (int u = 0; u < P; ++u) load_item(u);
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            if (u + P < NU) load_item(u + P);
            __builtin_amdgcn_sched_barrier(0);
            const int s = u / NVTG, c = u - s * NVTG;
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[c][q] = Ops<T>::mfma(wa[cp][s][q], fb[u % R], acc[c][q]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        AMX_STAMP();
        if (h == NH - 1) {
          // the next stage's halo (issued at least a whole sweep ago) and the previous stores have landed; the reads of this stage
          // have returned (their MFMAs were issued), so its buffer may be refilled by the stage after the next
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          AMX_STAMP();
        }
      }

      // ---- reduction of group h over the KW waves of the team, OWN / RT rounds: in a round wave o finalises its tiles
      //      o OWN + r RT + i of the group.  Bias of this lane's channels: requested per group (L1-resident after the first) and used
      //      behind the barrier -- 4 Q registers that are not carried through the sweeps.
      f32x4 bq[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q)
        bq[q] = (!PART && p.bias) ? *(const f32x4*)(p.bias + cb + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        if (dbg & 16) {                                     // ablation: no reduction, no stores (the barriers order the shared buffers)
          __syncthreads();
          __syncthreads();
          continue;
        }
#pragma unroll
        for (int o = 0; o < KW; ++o) {
          if (o != k && cu_valid) {
            const int slot = k < o ? k : k - 1;
            char* dst = scratch + ((o * (KW - 1) + slot) * RT * Q) * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
              for (int q = 0; q < Q; ++q) *(f32x4*)(dst + (i * Q + q) * 1024) = acc[o * OWN + r * RT + i][q];
          }
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < KW; ++o) {
          if (o == k && cu_valid) {
            bool bad = false;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
              f32x4 v[Q];
              // fixed order wave 0, 1, .., KW - 1; this wave's own partial sits in its registers
#pragma unroll
              for (int w2 = 0; w2 < KW; ++w2) {
                const int slot = w2 < o ? w2 : w2 - 1;
                const char* src = scratch + ((o * (KW - 1) + slot) * RT * Q) * 1024 + lane * 16;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                  const f32x4 x = w2 == o ? acc[o * OWN + r * RT + i][q] : *(const f32x4*)(src + (i * Q + q) * 1024);
                  v[q] = w2 == 0 ? x : v[q] + x;
                }
              }
              if (!PART) {
#pragma unroll
                for (int q = 0; q < Q; ++q) v[q] += bq[q];
                act_inplace<Q>(&v[0], p.act, p.slope);
              }
              // ---- store: lane (li, g) holds the 4 Q consecutive channels cb .. of voxel li of the tile
              const int c = h * NVTG + o * OWN + r * RT + i;
              const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
              const int zl = cu.z0 + C::tile_z(cz) + dzl, yl = cu.y0 + cy, xl = cu.x0 + cx * LX + dx;
              if (!full && !((zl < p.D) & (yl < p.H) & (xl < p.W))) continue;
              if (PART) {
                const long long vox = (((long long)cu.n * p.D + zl) * p.H + yl) * p.W + xl;
                float* d = p.part + ((long long)slice * p.N * p.D * p.H * p.W + vox) * p.Cout + cb;
#pragma unroll
                for (int q = 0; q < Q; ++q) *(f32x4*)(d + q * 4) = v[q];
              } else {
                const int ocs = p.ocs ? p.ocs : 32;          // the lane's 4 Q <= 16 channels sit inside one 16-channel chunk
                char* d = p.out + (long long)cu.n * p.on + (long long)zl * p.oz + (long long)yl * p.oy + (long long)xl * p.ox +
                          (long long)(cb >> 4) * ocs + (cb & 15) * 2;
                unsigned pk[2 * Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(v[q][j]);
                  pk[2 * q] = (unsigned)to_bits<T>(v[q][0]) | ((unsigned)to_bits<T>(v[q][1]) << 16);
                  pk[2 * q + 1] = (unsigned)to_bits<T>(v[q][2]) | ((unsigned)to_bits<T>(v[q][3]) << 16);
                }
                if (Q == 1) {
                  *(uint2*)d = make_uint2(pk[0], pk[1]);
                } else {
#pragma unroll
                  for (int j = 0; j < Q / 2; ++j) *(uint4*)(d + j * 16) = make_uint4(pk[4 * j], pk[4 * j + 1], pk[4 * j + 2], pk[4 * j + 3]);
                }
              }
            }
            if (!PART && RangeCheck<T>::on) raise_flag(p.oflow, bad);
          }
        }
        __syncthreads();                                    // the scratch tiles may be overwritten (next round / group / brick)
      }
      AMX_STAMP();
    }
    t += CPW;
    cu = nx;
    cu_valid = nx_valid;
  }
#undef AMX_STAMP
#undef AMX_DMA16
}

// ---- cross-workgroup split-K: out = act(bias + sum over the slices IN SLICE ORDER of the fp32 partial tensors [slice][voxel][Cout]).
// One thread = 8 channels of one voxel.
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int S, const float* __restrict__ bias, char* __restrict__ out,
                                     long long on, long long oz, long long oy, long long ox, int ocs, int N, int D, int H, int W,
                                     int Cout, int act, float slope, int* oflow) {
  const int c8n = Cout >> 3;
  const long long nvox = (long long)N * D * H * W, total = nvox * c8n;
  const float kact = act_k(act, slope);
  bool bad = false;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % c8n);
    const long long vox = idx / c8n;
    const float* src = part + vox * Cout + c8 * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int s = 0; s < S; ++s) {
      const float4 a = *(const float4*)(src + (long long)s * nvox * Cout), b = *(const float4*)(src + (long long)s * nvox * Cout + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = act_fwd(v[e] + (bias ? bias[c8 * 8 + e] : 0.f), kact);
      if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(v[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_bits<T>(v[2 * e]) | ((unsigned)to_bits<T>(v[2 * e + 1]) << 16);
    long long r = vox;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    r /= H;
    const int z = (int)(r % D);
    const int n = (int)(r / D);
    char* d = out + (long long)n * on + (long long)z * oz + (long long)y * oy + (long long)x * ox + (long long)(c8 >> 1) * (ocs ? ocs : 32) +
              (c8 & 1) * 16;
    *(uint4*)d = make_uint4(o[0], o[1], o[2], o[3]);
  }
  if (RangeCheck<T>::on) raise_flag(oflow, bad);
}

// -------------------------------------------------------------------------------------------
// launcher
// -------------------------------------------------------------------------------------------
static thread_local char g_kernel_name_ks[64] = "";
const char* last_conv_ks_kernel_name() { return g_kernel_name_ks; }
static int g_ks_cus = 0;

template <typename T, typename C, bool PART>
static hipError_t launch_ks_cfg(ConvParams p, hipStream_t st) {
  snprintf(g_kernel_name_ks, sizeof g_kernel_name_ks, "conv3d_k3_ks<%s,%dx%dx%d,q%d,k%dx%d,c%d,t%d,h%d,b%d%s>", __is_same(T, f16) ? "f16" : "bf16",
           C::TZ, C::TY, C::TX, C::Q, C::KW, C::CPW, C::CW, C::TEAMS, C::NH, C::NBUF, PART ? ",part" : "");
  auto kern = conv3d_k3_ks_kernel<T, C, PART>;
  static amx::DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_once.set();
  }
  static int ks_cus_dev = -1;                                 // (the CU count of the device it was read on)
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorUnknown;
    if (g_ks_cus == 0 || dev != ks_cus_dev) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
      g_ks_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      ks_cus_dev = dev;
    }
  }
  p.nbz = (p.D + C::TZ - 1) / C::TZ;
  p.nby = (p.H + C::TY - 1) / C::TY;
  p.nbx = (p.W + C::TX - 1) / C::TX;
  const int nbricks = p.nbz * p.nby * p.nbx * p.N;
  const int S = p.kslices > 0 ? p.kslices : 1;
  const int combos = p.Cout / (16 * C::Q * C::CW) * S;
  int wpc = g_ks_cus * C::WGS_PER_CU / combos;              // workgroups per (cout group, slice): WGS_PER_CU workgroups per CU in total
  if (wpc < 1) wpc = 1;
  const int runs = (nbricks + C::TEAMS - 1) / C::TEAMS;      // a workgroup needs at least one brick per team to be useful
  if (wpc > runs) wpc = runs;
  static int dbg = -1;                                      // (always 0 in the product build: exp_env is a constant there)
  static unsigned long long* trace_buf = nullptr;
  if (dbg < 0) {
    const char* e = exp_env("AMX_DBG");
    dbg = e ? atoi(e) : 0;
    if (exp_env("AMX_TRACE")) dbg |= 8;
  }
  p.dbg = dbg;
  const unsigned grid = (unsigned)(combos * wpc);
  if (dbg & 8) {   // debug only: per-phase cycle stamps, printed after a sync
    if (!trace_buf && hipMalloc((void**)&trace_buf, 1024 * 64 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(trace_buf, 0, 1024 * 64 * 8, st);
    p.stats = (float*)trace_buf;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NWAVE * 64), C::LDS_BYTES, st, p);
  if (dbg & 8) {
    static int printed = 0;
    (void)hipStreamSynchronize(st);
    if (printed++ == 3) {
      static unsigned long long hostbuf[1024 * 64];
      (void)hipMemcpy(hostbuf, trace_buf, sizeof hostbuf, hipMemcpyDeviceToHost);
      const int wgs[4] = {0, 1, (int)grid / 2, (int)grid - 1};
      for (int wi = 0; wi < 4; ++wi) {
        const unsigned long long* tr = hostbuf + (long long)wgs[wi] * 64;
        fprintf(stderr, "[trace %s wg %d] start %llu :", g_kernel_name_ks, wgs[wi], tr[0]);
        for (int k = 1; k < 64 && tr[k]; ++k) fprintf(stderr, " %llu", tr[k] - tr[k - 1]);
        fprintf(stderr, "\n");
      }
    }
  }
  return hipGetLastError();
}

// Shapes this kernel takes: single full-resolution segment, 16-bit channels-last (or row-planar) output, the plain 16-bit
// precisions; (Q, input chunks) among the register-stationary configurations below; 8 <= W.
//   chunks per slice = KW * CPW;  Q * CPW <= 4 (224 VGPRs of weights per wave)
struct KsPlan {
  int ok, kw, cpw, teams, slices, cw;
};
static KsPlan ks_plan(const ConvParams& p, int precision, int Q, bool can_split) {
  KsPlan r = {0, 0, 0, 0, 1, 1};
  if (precision > 1 || p.C1 != 0 || p.src0_f32c1 || p.out32 || p.out2 || p.stats || !p.out || p.W < 8 || p.H < 2 || p.D < 2) return r;
  if ((long long)p.D * p.s0z >= (1ll << 31)) return r;        // 32-bit halo offsets inside one sample
  const int nchunk = p.C0 >> 4;
  if (p.C0 & 15) return r;
  const int tz = p.W >= 16 ? 2 : 4, ty = 4, tx = p.W >= 16 ? 16 : 8;
  const long long nbricks = (long long)((p.D + tz - 1) / tz) * ((p.H + ty - 1) / ty) * ((p.W + tx - 1) / tx) * p.N;
  const int ncg = p.Cout / (16 * Q);
  if (Q == 4) {
    if (p.W < 16) return r;
    // the packed group of 4 tiles is split over two cout waves of 2 tiles: 112 registers of weights, two waves per SIMD
    if (nchunk == 2) { r = {1, 2, 1, 2, 1, 2}; return r; }
    if (nchunk == 4) { r = {1, 4, 1, 1, 1, 2}; return r; }
    return r;
  }
  if (Q != 2) return r;
  // Q = 2: slices of 4 chunks (CPW = 1) or 8 chunks (CPW = 2).  Prefer the fewest slices that still give every CU a workgroup.
  // One-chunk waves pair up as two cout waves per K position (two waves per SIMD) where 64-cout blocks still fill the chip.
  auto cw_of = [&](int slices) { return (p.Cout % 64 == 0 && (long long)(p.Cout / 64) * slices * nbricks >= 200) ? 2 : 1; };
  if (nchunk == 4) { r = {1, 4, 1, 1, 1, cw_of(1)}; return r; }
  if (nchunk % 8 == 0 && can_split) {
    // >= 8 chunks: only as a cross-workgroup K split.  (Un-split, with two chunks per wave -- 224 registers of weights, one wave
    // per SIMD -- 128 -> 128 @16^3 measured 24.9 us per launch against 21.3 for conv3d_k3_v2: profiles/r05_ks_shapes.txt; that
    // shape stays on the generic kernel.)
    const int s8 = nchunk / 8;                                // slices with CPW = 2
    static const int thin = exp_env("AMX_KS_THIN") ? 1 : 0;  // experiment: always one chunk per wave (twice the slices)
    if ((long long)ncg * nbricks >= 192) return r;            // enough (cout group, brick) pairs without a split: partial tensors would only add traffic
    if (!thin && s8 >= 2 && (long long)ncg * s8 * nbricks >= 192 && s8 <= 8) { r = {1, 4, 2, 1, s8, 1}; return r; }
    if (s8 * 2 <= 16) { r = {1, 4, 1, 1, s8 * 2, cw_of(s8 * 2)}; return r; }
  }
  return r;
}

bool conv_ks_eligible(const ConvParams& p, int precision, int Q) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_KS") ? 1 : 0;
  if (off) return false;
  return ks_plan(p, precision, Q, p.part != nullptr).ok != 0;
}

// bytes of fp32 partial tensors a layer of this shape needs when the caller offers split-K scratch (0: no split)
size_t conv_ks_part_bytes(int C0, int Cout, int N, int D, int H, int W, int precision, int Q) {
  ConvParams p;
  memset(&p, 0, sizeof p);
  p.C0 = C0; p.Cout = Cout; p.N = N; p.D = D; p.H = H; p.W = W; p.out = (char*)1;
  p.s0z = (long long)H * W * C0 * 2;
  const KsPlan pl = ks_plan(p, precision, Q, true);
  if (!pl.ok || pl.slices <= 1) return 0;
  return (size_t)pl.slices * N * D * H * W * Cout * sizeof(float);
}

template <typename T>
static hipError_t launch_ks_t(ConvParams p, const KsPlan& pl, int Q, hipStream_t st) {
  const bool wide = p.W >= 16;
  p.kslices = pl.slices;
  const bool part = pl.slices > 1;
  hipError_t e = hipErrorInvalidValue;
#define AMX_KS(TZ, TY, TX, QW, QP, CPW, KW, CW, TEAMS, NH, ...)                              \
  do {                                                                                       \
    typedef KsCfg<TZ, TY, TX, QW, QP, CPW, KW, CW, TEAMS, NH, ##__VA_ARGS__> CC;             \
    e = part ? launch_ks_cfg<T, CC, true>(p, st) : launch_ks_cfg<T, CC, false>(p, st);       \
  } while (0)
  //                                                 brick     Q/wave packed CPW KW CW teams groups
  // Measured and NOT kept (profiles/r05_ks_shapes.txt, r05_ks_flags_trace.txt): NBUF = 1 -- two un-prefetched 4-wave workgroups per CU
  // instead of one 8-wave workgroup -- 64 -> 64 @32^3 41.9 -> 47.8 us (every brick's halo is then staged by two workgroups); LDS
  // counters instead of workgroup barriers for the two reduction groups: 42.3 -> 42.1 us (the groups do drift into opposite phases,
  // but a wave sweeps only 28 % of its time: the reduction round itself is the cost).
  if (Q == 4 && pl.kw == 2) {            // 32 -> 64 (W >= 32: conv_pick_q): 2 chunks, two bricks in flight, two waves per SIMD
    if (wide) AMX_KS(2, 4, 16, 2, 4, 1, 2, 2, 2, 2);
  } else if (Q == 4) {                   // 64 -> 64: 4 chunks x 2 cout waves
    if (wide) AMX_KS(2, 4, 16, 2, 4, 1, 4, 2, 1, 2);
  } else if (pl.cpw == 2) {              // 8 chunks per slice: 224 registers of weights, one wave per SIMD
    if (wide) AMX_KS(2, 4, 16, 2, 2, 2, 4, 1, 1, 1); else AMX_KS(4, 4, 8, 2, 2, 2, 4, 1, 1, 1);
  } else {                               // 4 chunks per slice, Q = 2 packing: 64 couts per workgroup on two cout waves
    if (pl.cw == 2) {
      if (wide) AMX_KS(2, 4, 16, 2, 2, 1, 4, 2, 1, 2); else AMX_KS(4, 4, 8, 2, 2, 1, 4, 2, 1, 2);
    } else {
      if (wide) AMX_KS(2, 4, 16, 2, 2, 1, 4, 1, 1, 1); else AMX_KS(4, 4, 8, 2, 2, 1, 4, 1, 1, 1);
    }
  }
#undef AMX_KS
  if (e != hipSuccess || !part) return e;
  const long long total = (long long)p.N * p.D * p.H * p.W * (p.Cout / 8);
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, st, p.part, pl.slices, p.bias, p.out, p.on, p.oz, p.oy, p.ox,
                     p.ocs, p.N, p.D, p.H, p.W, p.Cout, p.act, p.slope, p.oflow);
  return hipGetLastError();
}

hipError_t launch_conv_ks(const ConvParams& p, int precision, int Q, hipStream_t st) {
  const KsPlan pl = ks_plan(p, precision, Q, p.part != nullptr);
  if (!pl.ok) return hipErrorInvalidValue;
  return precision == 0 ? launch_ks_t<f16>(p, pl, Q, st) : launch_ks_t<bf16>(p, pl, Q, st);
}

}  // namespace amx
