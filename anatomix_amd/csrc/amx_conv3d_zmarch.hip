// anatomix_amd -- conv3d 3x3x3 reflect for the narrow full/half-resolution layers
// (16 -> 16 @128^3: network.py modules 3, 6, 62, 65; 16 -> 32 and 32 -> 32 @64^3: modules 10, 13, 55
// of the 6M model), optionally with the following nn.MaxPool3d(2) fused into the epilogue:
// z-marching streaming kernel with producer / consumer wave specialisation.
//
// Roofline: 216 .. 431 FLOP/B -- these layers sit at or below the ridge point, so
// the design minimises bytes moved per output voxel and keeps the memory pipeline continuously full:
//   * a workgroup owns an in-plane tile TY x TX and marches along z through a segment of the
//     volume, two output planes per step.  Input z-planes (with their 1-voxel in-plane halo) live in
//     an LDS RING of R planes; every input plane is fetched ONCE per workgroup (read amplification
//     (TY+2)(TX+2)/(TY*TX) instead of the ~2x of a brick with a full 3-D halo);
//   * 2*NCK loader waves (one per 8-channel plane) do nothing but LDS-DMA (global_load_lds_dwordx4, 64
//     lanes x 16 B, per-lane reflect offsets precomputed once per march): a VMEM instruction blocks
//     its wave while the memory queue is full, so issuing from the MFMA waves serialised streaming
//     and math.  Loaders run up to R - 4 planes ahead with counted waits (never vmcnt(0));
//   * 8 consumer waves only sweep and store.  Each owns ONE 16-channel output tile q and a 2 (planes) x
//     2 (rows) x 16 (x) block of voxels per step, and keeps that tile's packed weights (14*NCK A
//     fragments) and bias in registers for the whole march.  The four input planes and four input rows
//     the block touches are shared between its four column tiles: 32 ds_read_b128 feed 56 MFMAs;
//   * the 2x2x2 block is exactly one max-pool window per x pair, so the pooled tensor is produced in
//     registers (max over the four accumulators, one DPP exchange for the x neighbour) and written
//     next to the full-resolution output -- the separate pool kernel and its re-read disappear;
//   * no workgroup barrier in the march: loaders publish "planes landed" and consumers "steps done"
//     counters in LDS (amx_device.h), so a wave stalled on VMEM issue only delays the waves that
//     depend on it and the store bursts of different waves de-synchronise.
// Arithmetic is identical to the generic kernel (same packed-weight layout, same 14 paired-tap steps).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "amx_device.h"

namespace amx {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NCK, int QT, int TY, int TX, int R>
struct ZmCfg {
  static constexpr int NC = 8, TZ = 2;
  static constexpr int NL = 2 * NCK;                               // loader waves: one per 8-channel plane
  static constexpr int HY = TY + 2, HX = TX + 2, HVP = HY * HX;    // halo voxels of one z-plane
  static constexpr int PPL = ((HVP * 16 + 255) / 256) * 256;       // one 8-channel plane of one z-plane
  static constexpr int PLSZ = NL * PPL;                            // one z-plane (16*NCK channels)
  static constexpr int FLAGOFF = R * PLSZ;                         // ready[NL] at +0, done[NC] at +32
  static constexpr int LDS_BYTES = R * PLSZ + 64;
  static constexpr int XT = TX / 16;
  static constexpr int WPQ = NC / QT;                              // consumer waves per output tile q
  static constexpr int NDMA = (HVP + 63) / 64;                     // DMA instructions per (z-plane, channel plane)
  static_assert(WPQ == (TY / 2) * XT, "each consumer wave owns a 2 x 2 x 16 voxel block per step");
  static_assert(R - 4 > TZ, "ring must hold more than one step of prefetch");
  static constexpr bool VMCNT_OK = R * NDMA <= 60;                  // a loader wave must not exceed the 6-bit vmcnt range (checked where loaders run)
  static_assert(NL <= 8 && LDS_BYTES <= 160 * 1024, "ring must fit the LDS");
};

// Output staging of the storer waves (NS = 2): one step of the tile (2 planes x TY x TX voxels) in LDS.
//   OUTMODE 0: [tz][y][x][Cout] 16-bit, exactly the global row layout (a row of TX voxels = TX * CB bytes);
//   OUTMODE 1: [c][tz][y][x] fp32 with the channel stride padded to 2112 B (conflict-free b128 writes).
template <int QT, int TY, int TX, int OUTMODE>
struct ZmStage {
  static constexpr int CB = 32 * QT;                                // bytes of one voxel (16-bit, Cout channels)
  static constexpr int CS = 2 * TY * TX * 4 + 64;                   // fp32 planar: bytes of one channel
  static constexpr int BYTES = OUTMODE == 0 ? 2 * TY * TX * CB : 16 * QT * CS;
  static constexpr int NB = OUTMODE == 0 ? 2 : 1;                   // staging tiles (the 32 KiB fp32 tile fits once)
  static constexpr int TOTAL = NB * BYTES;
};

// STEM variant (round 6): the layer's INPUT is the single-channel stem convolution of the network input (network.py module 0 + folded
// BatchNorm + ReLU), computed into the ring by the workgroup itself instead of being read from HBM -- the stem's 268 MB store and this
// layer's 268 MB read (batch 4, 128^3) disappear; the stem costs one K = 32 MFMA pair per 16 voxels (amx_conv3d_stem.hip, row-fragment
// formulation: same operands, same two MFMAs, same rounding -> the ring holds bit for bit what the stem kernel stores).
//   * 16 waves of <= 128 registers: 8 consumers (unchanged sweep, rolling fragment reads) + 7 STEM waves + 1 LOADER wave;
//   * the loader streams the 12 rows x 40 floats of every fp32 input plane into a staging ring -- two 16-byte LDS-DMA instructions per
//     plane (`landed` counts them);
//   * stem wave w owns the staged planes k = w, w + 7, ...: it rounds them ONCE to the storage type and writes the two shifted 16-bit
//     row copies of the stem kernel into the input ring of RC planes -- copy A: value x at element x, copy B: value x + 1 at element
//     x, so that the four consecutive values x - 1 .. x + 2 of ANY x start 4-byte aligned in one of the copies -- reflecting the INPUT
//     through the staged row / column it reads (`inconv[w]` = the next plane it will convert);
//   * and it computes WHOLE ring planes q = w, w + 7, ...: a ring voxel OUTSIDE the volume is the reflection of the stem's OUTPUT, so
//     its lane gathers the taps of the reflected voxel; it publishes ready[w] = the next plane it will publish.
// (the forms this went through and what each showed: section 0.)
struct StemIn {
  const char* src;              // fp32 network input [N][1][D][H][W] through byte strides (x contiguous)
  long long sn, sz, sy;
  long long offs[16];           // use_offs: sample i reads its volume at src + offs[i] bytes (sliding-window batches) instead of i * sn
  int use_offs;
  const char* wpk;              // the stem's packed weights (pack_stem_kernel; the row-fragment tiles start at byte 4096)
  const float* bias;            // [16] folded norm shift / conv bias or null
  int act;                      // ACT_NONE or ACT_RELU
  float slope;
};

template <int TY, int TX>
struct ZmStemCfg {
  static constexpr int HY = TY + 2, HX = TX + 2, HVP = HY * HX;
  static constexpr int IY = HY + 2, IX = HX + 2, IVP = IY * IX;       // input region of a ring plane: one more voxel of halo
  static constexpr int RS = (TX + 8) * 2;                             // bytes of a row copy: the values [x0 - 2, x0 + TX + 6)
  static constexpr int CPSZ = IY * RS;                                // copies A | B of one input plane
  static constexpr int IPLSZ = ((2 * CPSZ + 8 + 63) / 64) * 64;       // + 8 bytes nobody reads (idle lanes of a conversion)
  static constexpr int DUMMY = IPLSZ - 8;
  static constexpr int RC = 16;                                       // input ring (row copies): what the <= 6 ring planes in production read + lookahead
  static constexpr int SW = 40;                                       // staged fp32 row: the floats [x0 - 4, x0 + 36), ten 16-byte DMA pieces
  static constexpr int STGSZ = ((IY * SW * 4 + 255) / 256) * 256;     // one staged fp32 plane (12 rows)
  static constexpr int RSTG = 8;                                      // staging ring (DMA planes in flight)
  static constexpr int NLD = (IVP + 63) / 64;                         // values per lane of a conversion
  static constexpr int BYTES = RC * IPLSZ + RSTG * STGSZ;
  static constexpr int NSW = 7;                                       // stem waves (+ 1 loader wave = 8 producer waves; 16 waves of <= 128 registers)
  static constexpr int LEAD = 3;                                      // a stem wave converts its input planes up to LEAD beyond the ring plane it starts
  static constexpr int NT = 2 * HY + (2 * HY + 15) / 16;              // tiles of one ring plane: (row, x 0..15), (row, x 16..31), the columns 32, 33 of 8 rows
  static constexpr bool OK = TX == 32 && IY == 12 && RSTG * 2 <= 60 && NLD == 7;   // two DMA instructions (6 rows x ten pieces) per input plane; vmcnt range
};

// One z-segment of one in-plane tile per workgroup.  OUTMODE 0: 16-bit NDHWC; 1: fp32 planar.
// NS = 2 adds two STORER waves: the MFMA waves leave each step's outputs in an LDS staging tile and the storers
// copy it out (storer w owns output plane w of every step).  A wave that issues global stores stalls while the
// memory queue is full; with the stores on the MFMA waves, math and write-out serialised (16 -> 16 @128^3, batch
// 4: 91 us without the epilogue, 100 us without the math, 130 us together; fp32 planar output 93 / 165 / 247 us).
// Requires full tiles and 16-byte aligned dense outputs (conv_zmarch_can_stage), else NS = 0.
// POOL: the fused 2x2x2 max-pool output (p.out2) is a compile-time variant -- its own kernel symbol (own row in profiler
// tables: it writes 1/8 more) and no run-time test in the epilogue of the plain one.
// SPLIT (strict precision, 16 -> 16 only): the input voxel is [hi(16) | lo(16)] -- the ring holds four 8-channel planes like a
// 32-channel input -- and the step is swept twice: the hi planes against Wh and Wl (two MFMAs per fragment read), the lo planes
// against Wh (Wh / Wl are the two packed "chunks" and stay in registers like the two chunks of the 32-channel kernel); the
// epilogue splits the fp32 result again.
template <typename T, int NCK, int QT, int TY, int TX, int R, int OUTMODE, int NS, bool POOL, bool SPLIT = false, bool STEM = false>
__global__ __launch_bounds__((8 + (STEM ? ZmStemCfg<8, 32>::NSW + 1 : 2 * (SPLIT ? 2 : NCK)) + NS) * 64) void conv3d_k3_zmarch_kernel(const ConvParams p, int zseg, int nseg, const StemIn si) {
  static_assert(!SPLIT || (NCK == 1 && QT == 1 && NS == 0), "strict z-march: 16 -> 16, direct stores");
  static_assert(!STEM || (NCK == 1 && !SPLIT && NS == 0 && OUTMODE == 0), "stem-fed z-march: 16 input channels, direct 16-bit stores");
  constexpr int NCKP = SPLIT ? 2 : NCK;                             // chunk planes in the ring / weight sets in registers
  typedef ZmCfg<NCKP, QT, TY, TX, R> C;
  typedef ZmStage<QT, TY, TX, OUTMODE> SG;
  typedef typename Ops<T>::vec8 vec8;
  constexpr int HX = C::HX, PPL = C::PPL, PLSZ = C::PLSZ, XT = C::XT, NDMA = C::NDMA, NL = C::NL, NC = C::NC, TZ = C::TZ;
  constexpr int STAGEOFF = C::FLAGOFF + 128;                        // flags: ready +0, done +32, staged +64, stored +96
  static_assert(NS == 0 || NS == 2, "one storer wave per output plane of a step");
  static_assert(NS == 0 || STAGEOFF + SG::TOTAL <= 160 * 1024, "ring + staging tile must fit the LDS");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- work item: (n, z segment, y tile, x tile); x fastest so XCD-neighbours share halos in L2
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
  const int zs = sg * zseg;                                 // even
  const int ze = (zs + zseg < p.D) ? zs + zseg : p.D;      // output planes [zs, ze)
  const int nplanes = ze - zs + 2;                          // input planes q = 0 .. nplanes-1 <-> z = zs-1+q
  const int nsteps = (ze - zs + TZ - 1) / TZ;

  int* ready = (int*)(smem + C::FLAGOFF);
  int* done = (int*)(smem + C::FLAGOFF + 32);
  int* staged = (int*)(smem + C::FLAGOFF + 64);              // per consumer wave: steps whose outputs are in the staging tile
  int* stored = (int*)(smem + C::FLAGOFF + 96);              // per storer wave: steps it has read out of the staging tile
  // (STEM: ready[w] = the next ring plane stem wave w publishes, stored[w] = inconv[w] = the next input plane it converts; [7] belongs to no wave)
  if (tid < ((NS || STEM) ? 32 : 16))
    ((int*)(smem + C::FLAGOFF))[tid] = (STEM && (tid < 8 || tid >= 24)) ? ((tid & 7) == 7 ? 0x7fffffff : (tid & 7)) : 0;
  if constexpr (STEM) {
    // The pad elements of every row copy (36 .. 39; the conversions write the 36 values of a row) are read as a fragment's fourth value
    // against a zero weight: they must be FINITE, so the input ring starts as zeros -- whatever bit patterns the previous kernel left in
    // this CU's LDS would otherwise turn 0 x Inf into NaN at a few voxels, on some boxes, sometimes (seen once in the registration test).
    typedef ZmStemCfg<TY, TX> SC;
    static_assert(SC::OK && STAGEOFF + SC::BYTES <= 160 * 1024, "stem-fed geometry; ring + input rings must fit the LDS");
    for (int i = tid; i < SC::RC * SC::IPLSZ / 4; i += (NC + SC::NSW + 1) * 64) ((int*)(smem + STAGEOFF))[i] = 0;
  }
  __syncthreads();

  if (NS && wave >= NC + NL) {
    // =========================== storer wave: output plane w of every step ===========================
    const int w = wave - NC - NL;
    const unsigned a_staged = lds_addr(staged), a_stored = lds_addr(stored + w);
    if constexpr (OUTMODE == 0) {
      constexpr int NR = TY * QT;                            // 1 KiB pieces of this plane's TY rows
      const char* sbase = smem + STAGEOFF + w * TY * TX * SG::CB + lane * 16;
      char* gbase = p.out + (long long)n * p.on + (long long)y0 * p.oy + (long long)x0 * p.ox + lane * 16;
      for (int s = 0; s < nsteps; ++s) {
        while (__builtin_amdgcn_readfirstlane(flag_min8_asm(a_staged)) < s + 1) __builtin_amdgcn_s_sleep(1);
        uint4 v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = *(const uint4*)(sbase + (s % SG::NB) * SG::BYTES + r * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flag_store_asm(a_stored, s + 1);
        const int zo = zs + s * TZ + w;
        if (zo >= ze || (p.dbg & 4)) continue;
        char* dst = gbase + (long long)zo * p.oz;
#pragma unroll
        for (int r = 0; r < NR; ++r) *(uint4*)(dst + (long long)(r / QT) * p.oy + (r % QT) * 1024) = v[r];
      }
    } else {
      constexpr int NCH = 16 * QT;
      const int y = lane >> 3, xq = lane & 7;                // one instruction = 8 rows x 128 B of one channel plane
      const char* sbase = smem + STAGEOFF + (w * TY + y) * (TX * 4) + xq * 16;
      float* gbase = p.out32 + (long long)n * p.pn + (long long)(y0 + y) * p.py + x0 + xq * 4;
      for (int s = 0; s < nsteps; ++s) {
        while (__builtin_amdgcn_readfirstlane(flag_min8_asm(a_staged)) < s + 1) __builtin_amdgcn_s_sleep(1);
        float4 v[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[c] = *(const float4*)(sbase + c * SG::CS);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flag_store_asm(a_stored, s + 1);
        const int zo = zs + s * TZ + w;
        if (zo >= ze || (p.dbg & 4)) continue;
        float* dst = gbase + (long long)zo * p.pz;
        if (p.wmap) {   // sliding-window accumulate: acc += wmap * feature
          const float4 wg = *(const float4*)(p.wmap + ((long long)zo * p.H + y0 + y) * p.W + x0 + xq * 4);
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const float4 old = *(const float4*)(dst + (long long)c * p.pc);
            v[c] = make_float4(old.x + wg.x * v[c].x, old.y + wg.y * v[c].y, old.z + wg.z * v[c].z, old.w + wg.w * v[c].w);
          }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) *(float4*)(dst + (long long)c * p.pc) = v[c];
      }
    }
    return;
  }

  if constexpr (STEM) {
    if (wave >= NC) {
      typedef ZmStemCfg<TY, TX> SC;
      constexpr int RS = SC::RS, CPSZ = SC::CPSZ, IPLSZ = SC::IPLSZ, RC = SC::RC, RSTG = SC::RSTG, NLD = SC::NLD, IX = SC::IX, SW = SC::SW;
      constexpr int INOFF = STAGEOFF, STGOFF = STAGEOFF + RC * IPLSZ;
      int* landed = staged;                                    // fp32 input planes landed in the staging ring (the storers' flag words are free: NS = 0)
      int* inconv = stored;                                    // per stem wave: the next input plane (index from zlo) it will convert
      // input planes the segment touches: ring plane q <-> z = zs - 1 + q, the stem voxel plane z1 = refl(z) reads z1 - 1 .. z1 + 1
      const int zlo = zs >= 2 ? zs - 2 : 0;
      const int zhi = ze + 1 <= p.D - 1 ? ze + 1 : p.D - 1;
      const int nin = zhi - zlo + 1;
      const char* src_n = si.src + (si.use_offs ? si.offs[n & 15] : (long long)n * si.sn);
      if (wave == NC + SC::NSW) {
        // =========================== loader wave ===========================
        // The 12 rows x 40 floats [y0 - 2, y0 + 10) x [x0 - 4, x0 + 36) of an input plane as 120 16-byte pieces = TWO LDS-DMA instructions
        // (lane -> row pr of six, piece pc of ten).  Rows / pieces outside the volume are clamped into it and never read: the reflection
        // happens when a stem wave reads the staged plane back (conversion).
        const int pr = lane / 10, pc = lane - pr * 10;
        const bool pvalid = lane < 60;
        int goff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int yy = y0 - 2 + pr + 6 * j, xx = x0 - 4 + 4 * pc;
          yy = yy < 0 ? 0 : (yy > p.H - 1 ? p.H - 1 : yy);
          xx = xx < 0 ? 0 : (xx > p.W - 4 ? p.W - 4 : xx);
          goff[j] = pvalid ? yy * (int)si.sy + xx * 4 : 0;
        }
        auto issue_plane = [&](int k) {
          const char* plane = src_n + (long long)(zlo + k) * si.sz;
          char* dstp = smem + STGOFF + (k % RSTG) * SC::STGSZ;
          if (pvalid) {
            __builtin_amdgcn_global_load_lds((gptr_t)(plane + goff[0]), (lptr_t)dstp, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(plane + goff[1]), (lptr_t)(dstp + 960), 16, 0, 0);
          }
        };
        int next_issue = 0, next_pub = 0, seen_conv = 0;
        const unsigned a_landed = lds_addr(landed);
#ifdef AMX_EXPERIMENT
        unsigned long long* ltr = (p.dbg & 8) && lane == 0 && blockIdx.x == 0 ? (unsigned long long*)p.stats + 1001 * 128 : nullptr;   // trace slot (wg 500, half 1)
        int lcount = 0;
#define AMX_LSTAMP() do { if (ltr && lcount < 128) ltr[lcount] = __builtin_readcyclecounter(); ++lcount; } while (0)
        AMX_LSTAMP();
#else
#define AMX_LSTAMP() do {} while (0)
#endif
        while (next_pub < nin) {
          // staging slot k % RSTG held plane k - RSTG: every plane below k - RSTG + 1 must have been converted (min over the waves' inconv)
          while (next_issue < nin) {
            const int need = next_issue - RSTG + 1;
            if (seen_conv < need) seen_conv = __builtin_amdgcn_readfirstlane(flag_min8_asm(lds_addr(inconv)));
            if (seen_conv < need) break;
            issue_plane(next_issue++);
          }
          if (next_issue == next_pub) {                         // staging full, everything published: wait for the conversions
            __builtin_amdgcn_s_sleep(2);
            continue;
          }
          AMX_LSTAMP();                                          // [issued what the staging ring allows]
          WaitVm<2, RSTG - 1>::run(next_issue - next_pub - 1);  // the oldest unpublished plane has landed
          AMX_LSTAMP();                                          // [landed]
          flag_store_asm(a_landed, ++next_pub);
          AMX_LSTAMP();                                          // [published]
        }
        return;
      }
      // =========================== stem wave w: the ring planes q = w, w + NSW, ... ===========================
      // A plane is a chain of LDS round trips (two polls, the operand reads, the dependent MFMA pair, the writes): ~2300 cycles however few
      // tiles a wave computes -- with every stem wave on every plane (first form: tiles t = NSW i + w) the workgroup made one plane per
      // 1.1 us whether it had three or seven stem waves.  So the waves take WHOLE planes, up to R - 4 = 6 of them in production at a time.
      // ready[w] = the next plane wave w will publish (starts at w): min over the waves = every plane below it is in the ring.
      // Ring plane = 10 rows of 34 voxels: tiles (row, x 0..15), (row, x 16..31), and the columns x = 32, 33 of eight rows per tile
      // (lane -> row li / 2).  Row tiles reach their operands from one base register per read kind and x tile through immediate
      // offsets; the reflected rows 0 / 9 (workgroups on the volume's y faces) add a uniform offset.
      constexpr int NSW = SC::NSW;
      const int w = wave - NC;
      const int li = lane & 15, g = lane >> 4;
      // (s_setprio 3 for these waves -- the consumers wait for their planes a third of the time -- measured SLOWER: 187 -> 201 us per launch)
      const vec8 w0 = *(const vec8*)(si.wpk + 4096 + lane * 16);
      // MFMA 1 carries row 8 only (lane group 0, k = 0 .. 2 of its K = 32 tile): run as the K = 16 product v_mfma_f32_16x16x16 (lane group g
      // holds k = 4 g .. 4 g + 3) on the fragment's first two dwords -- half the matrix time, no zero registers in the B operand
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      u32x2 w1 = *(const u32x2*)(si.wpk + 4096 + 1024 + lane * 16);
      if (g) w1 = u32x2{0u, 0u};
      const f32x4 sbias = si.bias ? *(const f32x4*)(si.bias + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float lo_clip = si.act == ACT_RELU ? 0.f : -__builtin_inff();   // relu as max(v, 0); no activation: max(v, -inf)
      // lane constants: rows (2g, 2g + 1) of MFMA 0, row 8 of MFMA 1; row r = 3 kz + ky
      const int r0 = 2 * g, r1 = 2 * g + 1;
      const int kz0 = r0 / 3, kz1 = r1 / 3;
      auto xpart = [&](int hx) -> int {                        // row-copy offset of the four values x - 1 .. x + 2 of ring column hx (reflected)
        const int v0 = reflect_clamp(x0 + hx - 1, p.W) - (x0 - 2) - 1;
        return (v0 & 1) * CPSZ + (v0 & ~1) * 2;
      };
      auto ytop = [&](int hy) -> int { return reflect_clamp(y0 + hy - 1, p.H) - (y0 - 2) - 1; };   // region row of tap ky = 0 of ring row hy
      const int xA = xpart(li), xB = xpart(16 + li);
      const int d0 = __builtin_amdgcn_readfirstlane((ytop(0) - 0) * RS), d9 = __builtin_amdgcn_readfirstlane((ytop(9) - 9) * RS);   // 0 unless reflected
      const int wlane = (g >> 1) * PPL + (g & 1) * 8 + li * 16;
      int croff[2], cwoff[2];                                  // column tiles
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int crow = 8 * c + (li >> 1);
        const bool cvalid = crow < C::HY;
        const int crc = cvalid ? crow : 0;
        croff[c] = ytop(crc) * RS + xpart(32 + (li & 1));
        cwoff[c] = cvalid ? (g >> 1) * PPL + (g & 1) * 8 + (crow * HX + 32 + (li & 1)) * 16 : (g >> 1) * PPL + C::HVP * 16 + (g & 1) * 8;   // idle lanes: the plane's padding
      }
      constexpr int NB = 4;                                    // tiles per batch
      float vmax = 0.f;
      int seen_in = 0, seen_done = 0;
      // ---- conversion duty: this wave rounds the staged fp32 input planes k = w, w + NSW, ... (index from zlo) to the storage type and writes
      // their two shifted row copies (copy A: value ix at element ix, copy B: value ix at element ix - 1; the stem kernel's layout).  The
      // reflection of the input happens here: lane value (iy, ix) of the 12 x 36 region reads the staged row / column of the reflected voxel.
      // (its per-value source / destination offsets are recomputed per conversion -- one plane in seven per wave -- rather than kept
      //  in 21 registers next to the tile pipeline's: those spilled)
      int myk = w, seen_landed = 0, seen_ready = 0;
      bool ibad = false;
      auto convert_plane = [&](int k) {
        while (seen_landed < k + 1) {
          seen_landed = flag_load(landed);
          if (seen_landed < k + 1) __builtin_amdgcn_s_sleep(1);
        }
        // the row-copy slot of input plane zin held plane zin - RC, last read by ring plane z = zin - RC + 1 (q = z - zs + 1): every ring
        // plane below zin - RC + 3 - zs must be done before it is overwritten
        const int need_done = zlo + k - RC + 3 - zs;
        while (seen_ready < need_done) {
          seen_ready = __builtin_amdgcn_readfirstlane(flag_min8_asm(lds_addr(ready)));
          if (seen_ready < need_done) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        const int so = (k % RSTG) * SC::STGSZ, dof = ((zlo + k) % RC) * IPLSZ;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));                       // (opaque: keeps hipcc from hoisting the seven offset sets out of the plane loop)
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
          const int hv = j * 64 + lane_o;
          const bool valid = hv < SC::IVP;
          const int hvc = valid ? hv : SC::IVP - 1;
          const int iy = hvc / IX, ix = hvc - iy * IX;
          const int ry = reflect_clamp(y0 + iy - 2, p.H) - (y0 - 2), rx = reflect_clamp(x0 + ix - 2, p.W) - (x0 - 4);
          const float f = *(const float*)(smem + STGOFF + so + (ry * SW + rx) * 4);
          if (RangeCheck<T>::on) ibad |= RangeCheck<T>::bad(f);       // an input beyond the storage range (or NaN) would reach the taps as Inf
          const unsigned short hb = to_bits<T>(f);
          const int la = valid ? iy * RS + ix * 2 : SC::DUMMY;
          const int lb = (valid && ix > 0) ? CPSZ + iy * RS + (ix - 1) * 2 : SC::DUMMY;
          *(unsigned short*)(smem + INOFF + dof + la) = hb;
          *(unsigned short*)(smem + INOFF + dof + lb) = hb;
        }
        asm volatile("" ::: "memory");                         // (LDS serves a wave's accesses in order: the flag lands behind the row copies)
        flag_store(inconv + w, k + NSW);
      };
#ifdef AMX_EXPERIMENT
      unsigned long long* str = (p.dbg & 8) && lane == 0 && blockIdx.x == 0 && w == 0 ? (unsigned long long*)p.stats + 1000 * 128 : nullptr;   // trace slot (wg 500, half 0)
      int scount = 0;
#define AMX_SSTAMP() do { if (str && scount < 128) str[scount] = __builtin_readcyclecounter(); ++scount; } while (0)
      AMX_SSTAMP();
#else
#define AMX_SSTAMP() do {} while (0)
#endif
      for (int q = w; q < nplanes; q += NSW) {
        const int z1 = reflect_clamp(zs - 1 + q, p.D);
        int zi[3];
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) zi[kz] = reflect_clamp(z1 - 1 + kz, p.D);
        int zmax = zi[0] > zi[1] ? zi[0] : zi[1];
        zmax = zmax > zi[2] ? zmax : zi[2];
        // own conversions first, up to LEAD planes beyond what this ring plane reads: a wave never blocks on a ring slot while it owes an
        // input plane that an older ring plane (another wave's) still waits for
        {
          int kmax = zs + q - zlo + SC::LEAD;
          kmax = kmax < nin - 1 ? kmax : nin - 1;
          for (; myk <= kmax; myk += NSW) convert_plane(myk);
        }
        while (seen_in < zmax - zlo + 1) {                     // (polled only when the last value seen does not already allow it)
          seen_in = __builtin_amdgcn_readfirstlane(flag_min8_asm(lds_addr(inconv)));
          if (seen_in < zmax - zlo + 1) __builtin_amdgcn_s_sleep(1);
        }
        while (q >= R + TZ * seen_done) {
          seen_done = __builtin_amdgcn_readfirstlane(flag_min8_asm(lds_addr(done)));
          if (q >= R + TZ * seen_done) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        AMX_SSTAMP();                                            // [input landed, ring slot free]
#ifdef AMX_EXPERIMENT
        if (p.dbg & 64) { flag_store(ready + w, q + NSW); AMX_SSTAMP(); AMX_SSTAMP(); continue; }
#endif
        int ib[3];
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) ib[kz] = INOFF + (zi[kz] % RC) * IPLSZ;
        const int b0 = (kz0 == 0 ? ib[0] : (kz0 == 1 ? ib[1] : ib[2])) + (r0 % 3) * RS;
        const int b1 = (kz1 == 0 ? ib[0] : (kz1 == 1 ? ib[1] : ib[2])) + (r1 % 3) * RS;
        const int b8 = ib[2] + 2 * RS;
        // bases (opaque to the compiler: the row offsets below then stay instruction immediates)
        int a0 = b0 + xA, a1 = b1 + xA, a8 = b8 + xA, c0 = b0 + xB, c1 = b1 + xB, c8 = b8 + xB;
        int wr = (q % R) * PLSZ + wlane;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a8), "+v"(c0), "+v"(c1), "+v"(c8), "+v"(wr));
        auto rd = [&](int addr) -> u32x2 {                     // two dwords at a 4-byte aligned address (ds_read2_b32)
          const unsigned* qq = (const unsigned*)(smem + addr);
          return u32x2{qq[0], qq[1]};
        };
        // tile TI of the plane: its three read addresses / its write address
        auto raddr = [&](int TI, int& r_0, int& r_1, int& r_8) {
          if (TI < 2 * C::HY) {
            const int row = TI >> 1, xt = TI & 1;
            const int d = (row == 0 ? d0 : (row == C::HY - 1 ? d9 : 0)) + row * RS;
            r_0 = (xt ? c0 : a0) + d; r_1 = (xt ? c1 : a1) + d; r_8 = (xt ? c8 : a8) + d;
          } else {
            const int c = TI - 2 * C::HY;
            r_0 = b0 + croff[c]; r_1 = b1 + croff[c]; r_8 = b8 + croff[c];
          }
        };
        auto waddr = [&](int TI) -> int {
          if (TI < 2 * C::HY) return wr + ((TI >> 1) * HX + 16 * (TI & 1)) * 16;
          return (q % R) * PLSZ + cwoff[TI - 2 * C::HY];
        };
        // Software pipeline over batches of NB tiles: iteration b issues the reads of batch b + 2, the products of batch b + 1 (its reads
        // are an iteration old) and the epilogue of batch b (its products are an iteration old).  One batch at a time -- reads, wait,
        // dependent MFMA pair, wait, epilogue -- took a stem wave ~8000 cycles per plane beside the consumers (6 batches x ~1300).
        constexpr int NBT = (SC::NT + NB - 1) / NB;
        u32x4 fa[2][NB];
        u32x2 fe[2][NB];
        f32x4 sacc[2][NB];
        auto stage_a = [&](int bt) {
#pragma unroll
          for (int t = 0; t < NB; ++t) {
            if (bt * NB + t >= SC::NT) continue;
            int r_0, r_1, r_8;
            raddr(bt * NB + t, r_0, r_1, r_8);
            const u32x2 a = rd(r_0), bq = rd(r_1);
            fa[bt & 1][t] = u32x4{a[0], a[1], bq[0], bq[1]};
            fe[bt & 1][t] = rd(r_8);
          }
        };
        auto stage_b = [&](int bt) {
#pragma unroll
          for (int t = 0; t < NB; ++t)
            if (bt * NB + t < SC::NT) sacc[bt & 1][t] = Ops<T>::mfma(w0, __builtin_bit_cast(vec8, fa[bt & 1][t]), sbias);
#pragma unroll
          for (int t = 0; t < NB; ++t)
            if (bt * NB + t < SC::NT) sacc[bt & 1][t] = Ops<T>::mfma16(w1, fe[bt & 1][t], sacc[bt & 1][t]);
        };
        auto stage_c = [&](int bt) {
#pragma unroll
          for (int t = 0; t < NB; ++t) {
            if (bt * NB + t >= SC::NT) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(sacc[bt & 1][t][j], lo_clip, __builtin_inff());   // max(v, clip) in one VALU; NOT inline
                                                                              // asm: hipcc must see the read of the MFMA result to pad its latency
            if (RangeCheck<T>::on) {
              asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(vmax) : "v"(vmax), "v"(v[0]), "v"(v[1]));
              asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(vmax) : "v"(vmax), "v"(v[2]), "v"(v[3]));
            }
            *(uint2*)(smem + waddr(bt * NB + t)) = make_uint2(Ops<T>::pack2(v[0], v[1]), Ops<T>::pack2(v[2], v[3]));
          }
        };
        stage_a(0);
        if (NBT > 1) stage_a(1);
        __builtin_amdgcn_sched_barrier(0);
        stage_b(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) {
          if (bt + 2 < NBT) stage_a(bt + 2);
          __builtin_amdgcn_sched_barrier(0);
          if (bt + 1 < NBT) stage_b(bt + 1);
          __builtin_amdgcn_sched_barrier(0);
          stage_c(bt);
          __builtin_amdgcn_sched_barrier(0);
        }
        AMX_SSTAMP();                                            // [plane computed]
        asm volatile("" ::: "memory");                         // (LDS serves a wave's accesses in order: the flag lands behind the plane)
        flag_store(ready + w, q + NSW);
        AMX_SSTAMP();                                            // [published]
      }
      for (; myk < nin; myk += NSW) convert_plane(myk);       // input planes that only other waves' last ring planes read
      if (RangeCheck<T>::on) raise_flag(p.oflow, ibad | !(vmax <= 65504.f));
      return;
    }
  }

  if (!STEM && wave >= NC) {
    // =========================== loader wave: channel plane cp = wave - NC ===========================
    static_assert(STEM || C::VMCNT_OK, "loader wave must not exceed the 6-bit vmcnt range");
    const int cp = wave - NC;
    int off[NDMA];                     // per-lane source offset of halo voxel hv = 64*j + lane (fixed for the march)
    bool valid[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      const int hv = j * 64 + lane;
      const int hy = hv / HX, hx = hv - hy * HX;
      valid[j] = hv < C::HVP;
      off[j] = halo_coord(y0 + hy - 1, p.H, p.raw_halo) * (int)p.s0y + halo_coord(x0 + hx - 1, p.W, p.raw_halo) * (int)p.s0x + cp * 16;
    }
    const char* src_n = p.src0 + (long long)n * p.s0n;
    auto issue_plane = [&](int q) {
      const char* plane = src_n + (long long)halo_coord(zs - 1 + q, p.D, p.raw_halo) * p.s0z;   // uniform
      char* dstp = smem + (q % R) * PLSZ + cp * PPL;                                    // uniform
#pragma unroll
      for (int j = 0; j < NDMA; ++j)
        if (valid[j]) __builtin_amdgcn_global_load_lds((gptr_t)(plane + off[j]), (lptr_t)(dstp + j * 1024), 16, 0, 0);
    };
    int next_issue = 0, next_pub = 0;
    const unsigned a_ready = lds_addr(ready + cp), a_done = lds_addr(done);
    while (next_pub < nplanes) {
      if (next_issue < nplanes) {
        const int md = __builtin_amdgcn_readfirstlane(flag_min8_asm(a_done));
        int lim = R + TZ * md;                              // planes q < TZ*md are dead: their slots are free
        lim = lim < nplanes ? lim : nplanes;
        while (next_issue < lim) issue_plane(next_issue++);
      }
      if (next_issue == next_pub) {                         // ring full, everything published: wait for consumers
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      WaitVm<NDMA, R - 1>::run(next_issue - next_pub - 1);  // the oldest unpublished plane has landed
      flag_store_asm(a_ready, ++next_pub);
    }
    return;
  }

  // ================================= consumer wave =================================
  const int li = lane & 15, g = lane >> 4, hi = g >> 1;
  const int wq = wave / C::WPQ;                            // output tile (16 channels) of this wave
  const int wr = wave % C::WPQ;
  const int wrow = (wr / XT) * 2;                          // first of the wave's two rows
  const int wcx = wr % XT;                                 // the wave's x tile
  // resident weights (A fragments of tile wq) and bias
  vec8 wreg[NCKP][kSteps];
#pragma unroll
  for (int k = 0; k < NCKP; ++k)
#pragma unroll
    for (int s = 0; s < kSteps; ++s) wreg[k][s] = *(const vec8*)(p.wpk + ((k * kSteps + s) * QT + wq) * 1024 + lane * 16);
  const int cb = g * 4 * QT + wq * 4;                      // lane holds output channels cb .. cb+3
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *(const f32x4*)(p.bias + cb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int lanebase = (g & 1) * PPL + ((wrow * HX) + wcx * 16 + li) * 16;
  const int base_d1 = lanebase + hi * 16;
  const int base_dx = lanebase + hi * 16 * HX;
  const bool full_xy = (y0 + TY <= p.H) & (x0 + TX <= p.W);

  // per-lane output bases (plane z added per step)
  const int yl = y0 + wrow, xl = x0 + wcx * 16 + li;
  char* out_l = OUTMODE == 0 ? p.out + (long long)n * p.on + (long long)yl * p.oy + (long long)xl * p.ox + cb * 2 : nullptr;
  float* out32_l = OUTMODE == 1 ? p.out32 + (long long)n * p.pn + (long long)cb * p.pc + (long long)yl * p.py + xl : nullptr;
  constexpr bool WIDE = OUTMODE == 0 && NS == 0 && QT == 1;
  // WIDE: lane (g, li) stores channels 8 (g >> 1) .. + 7 of voxel (row yl + (g & 1), xl)
  char* out_w = WIDE ? p.out + (long long)n * p.on + (long long)(yl + (g & 1)) * p.oy + (long long)xl * p.ox + (g >> 1) * 16 : nullptr;
  // fp32 planar output with 16-byte stores needs x-quads that never straddle a row and aligned planes
  const bool vec_planar = OUTMODE == 1 && !(p.W & 3) && !(p.py & 3) && !(p.pz & 3) && !(p.pc & 3) && !(p.pn & 3) &&
                          !((size_t)p.out32 & 15) && !((size_t)p.wmap & 15);
  float* out32_q = OUTMODE == 1 ? p.out32 + (long long)n * p.pn + (long long)(cb + (li & 3)) * p.pc + (long long)yl * p.py +
                                      x0 + wcx * 16 + (li & ~3)
                                : nullptr;
  // fused 2x2x2 max-pool output (dense 16-bit NDHWC at half resolution); even lanes store
  char* out2_l = (OUTMODE == 0 && POOL) ? p.out2 + (long long)n * p.qn + (long long)(yl >> 1) * p.qy +
                                               (long long)(xl >> 1) * p.qx + cb * 2
                                          : nullptr;

  bool bad = false;
  // optional cycle trace (AMX_TRACE=1): consumer waves 0 and 4 of a few workgroups stamp s_memtime per phase
  unsigned long long* trace = (p.dbg & 8) && (wave == 0 || wave == 4) && lane == 0 && blockIdx.x < 500
                                  ? (unsigned long long*)p.stats + ((long long)blockIdx.x * 2 + (wave >> 2)) * 128 : nullptr;
  int tcount = 0;
#define AMX_ZSTAMP()                                                             \
  do {                                                                           \
    if (trace && tcount < 128) trace[tcount] = __builtin_readcyclecounter();     \
    ++tcount;                                                                    \
  } while (0)
  AMX_ZSTAMP();
  for (int s = 0; s < nsteps; ++s) {
    {
      int need = TZ * s + TZ + 2;                          // planes q <= TZ*s + TZ + 1 must have landed
      need = need < nplanes ? need : nplanes;
      static_assert(NL <= 4 && C::FLAGOFF % 16 == 0, "the ready flags are polled with one 16-byte read");
      if constexpr (STEM) {
        while (__builtin_amdgcn_readfirstlane(flag_min8_asm(lds_addr(ready))) < need) __builtin_amdgcn_s_sleep(1);
      } else {
        while (flag_min4<NL>(ready) < need) __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
    }

    AMX_ZSTAMP();                                            // [planes landed]
#ifdef AMX_EXPERIMENT
    // (AMX_DBG bits 16..23 = n: the second consumer wave of every SIMD starts its first sweep n x 512 cycles late)
    if (s == 0 && wave >= 4 && wave < 8)
      for (int k = (p.dbg >> 16) & 255; k > 0; --k) __builtin_amdgcn_s_sleep(8);
#endif
    // ring slots of the four input planes zs+2s-1 .. zs+2s+2  (q = 2s + pl)
    int b1[4], bx3[4];
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      const int sl = ((s * TZ + pl) % R) * PLSZ;
      b1[pl] = base_d1 + sl;
      bx3[pl] = base_dx + sl;
    }
    // accumulators: [output plane tz][row cy]
    f32x4 acc[2][2];
#pragma unroll
    for (int tz = 0; tz < 2; ++tz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy) acc[tz][cy] = bias;

    // TIMING-ONLY experiment (AMX_DBG bits 10..12 = extra sweeps per step; results are garbage): what a layer costs when its matrix
    // work is doubled / tripled at unchanged traffic -- the lower bound of a conv -> conv chain through the ring
    for (int rep = (p.dbg >> 10) & 7; rep >= 0; --rep)
    if (!(p.dbg & 2)) {
      if constexpr ((NCKP > 1 && !SPLIT) || STEM) {   // (STEM: 16 waves per workgroup leave each 128 registers -- no room for the double-buffered plane sets below)
        // Two (or, SPLIT, two passes over) channel chunks: 28 resident weight fragments leave room for ONE set of six ring fragments, and
        // "read a plane's six, then its 16 MFMAs" (the first form) exposed the LDS round trip eight times per step -- a wave's sweep
        // took ~4100 cycles for 1792 cycles of MFMAs, with or without the other wave of its SIMD (profiles/r05_wgrad_probe.txt).  Rolling
        // form: the step's fragment uses are one static sequence (per pass: 4 planes x [4 row fragments, 2 x+2 fragments], then the 8
        // fragments of the taps (0,2,2)|(1,2,2) and (2,2,2)); use i sits in slot i mod NSLOT, and the read for use i + NSLOT is issued
        // right after the MFMAs of use i -- about 9 MFMAs ahead of its first use, in fewer registers than the one-plane set.
        constexpr int PER = 4 * 6 + 8, NU = NCKP * PER, NSLOT = 4;
        // addresses are formed per read from two lane bases and a wave-uniform ring offset (scalar registers): the eight per-plane
        // address registers of the first form, next to 112 weight registers, pushed weight fragments into scratch memory
        int slo[4];                                          // ring offsets of the step's four input planes (uniform)
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) slo[pl] = __builtin_amdgcn_readfirstlane(((s * TZ + pl) % R) * PLSZ);
        const int tail_lo = lanebase + (2 * HX + 2) * 16;   // taps (kz, 2, 2): low lanes plane tz (+ hi: tz + 1), and plane tz + 2
        auto src = [&](int i) -> const vec8* {              // ring address of use i (i is a compile-time constant after unrolling)
          const int ps = i / PER, j = i % PER, koff = ps * 2 * PPL;
          if (j < 24) {
            const int pl = j / 6, f = j % 6;
            return f < 4 ? (const vec8*)(smem + base_d1 + (slo[pl] + koff + (f * HX) * 16))
                         : (const vec8*)(smem + base_dx + (slo[pl] + koff + ((f - 4) * HX + 2) * 16));
          }
          const int t = j - 24, tz = t >> 2, cy = (t >> 1) & 1, which = t & 1;
          if (which) return (const vec8*)(smem + tail_lo + (slo[tz + 2] + koff + (cy * HX) * 16));
          return (const vec8*)(smem + tail_lo + ((hi ? slo[tz + 1] : slo[tz]) + koff + (cy * HX) * 16));
        };
        auto use = [&](int i, const vec8& frag) {           // the MFMAs that consume use i
          const int ps = i / PER, j = i % PER;
          const int k = SPLIT ? 0 : ps;
          const bool both = SPLIT && ps == 0;
          auto mm = [&](const int widx, f32x4 a) {
            a = Ops<T>::mfma(wreg[k][widx], frag, a);
            if (both) a = Ops<T>::mfma(wreg[NCKP - 1][widx], frag, a);
            return a;
          };
          if (j < 24) {
            const int pl = j / 6, f = j % 6;
#pragma unroll
            for (int tz = 0; tz < 2; ++tz) {
              const int kz = pl - tz;
              if (kz < 0 || kz > 2) continue;
              if (f < 4) {
#pragma unroll
                for (int cy = 0; cy < 2; ++cy) {
                  const int ky = f - cy;
                  if (ky >= 0 && ky < 3) acc[tz][cy] = mm(kz * 3 + ky, acc[tz][cy]);
                }
              } else {
                acc[tz][f - 4] = mm(9 + kz, acc[tz][f - 4]);
              }
            }
          } else {
            const int t = j - 24, tz = t >> 2, cy = (t >> 1) & 1, which = t & 1;
            acc[tz][cy] = mm(12 + which, acc[tz][cy]);
          }
        };
        vec8 S[NSLOT];
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) S[i] = *src(i);
#pragma clang loop unroll(full)                              // (a partly unrolled loop indexes wreg[] dynamically: the weights go to scratch memory)
        for (int i = 0; i < NU; ++i) {
          use(i, S[i % NSLOT]);
          if (i + NSLOT < NU) S[i % NSLOT] = *src(i + NSLOT);
          __builtin_amdgcn_sched_barrier(0);                 // (keeps every read behind the MFMAs of the use it replaces, and no further)
        }
      } else {
#pragma unroll
        for (int ps = 0; ps < NCKP; ++ps) {
          // SPLIT: pass 0 reads the hi planes and multiplies every fragment with Wh AND Wl (two MFMAs per fragment read),
          // pass 1 reads the lo planes and multiplies with Wh.  Otherwise pass = chunk.
          const int k = SPLIT ? 0 : ps;                        // (first) weight set of this pass
          const bool both = SPLIT && ps == 0;                  // also the second weight set (Wl)
          const int koff = ps * 2 * PPL;                       // channel planes 2 ps, 2 ps + 1
          auto mm = [&](const int widx, const vec8& frag, f32x4 a) {
            a = Ops<T>::mfma(wreg[k][widx], frag, a);
            if (both) a = Ops<T>::mfma(wreg[NCKP - 1][widx], frag, a);
            return a;
          };
          // Per input plane pl: 4 row fragments of taps (kz,ky,0)|(kz,ky,1) and 2 fragments of taps
          // (kz,0,2)|(kz,1,2); plane pl is tap plane kz = pl - tz of output plane tz.  Batches are double
          // buffered one plane ahead of their MFMAs (sched_barrier pins the phases).
          constexpr bool DB = NCKP == 1;     // two chunks: 28 resident weight fragments leave no room for a 2nd buffer
          vec8 F[DB ? 2 : 1][4], H[DB ? 2 : 1][2];
          auto load_plane = [&](int buf, int pl) {
  #pragma unroll
            for (int r = 0; r < 4; ++r) F[buf][r] = *(const vec8*)(smem + b1[pl] + koff + (r * HX) * 16);
  #pragma unroll
            for (int cy = 0; cy < 2; ++cy) H[buf][cy] = *(const vec8*)(smem + bx3[pl] + koff + (cy * HX + 2) * 16);
          };
          auto mma_plane = [&](int buf, int pl) {
  #pragma unroll
            for (int tz = 0; tz < 2; ++tz) {
              const int kz = pl - tz;
              if (kz < 0 || kz > 2) continue;
  #pragma unroll
              for (int ky = 0; ky < 3; ++ky)
  #pragma unroll
                for (int cy = 0; cy < 2; ++cy) acc[tz][cy] = mm(kz * 3 + ky, F[buf][cy + ky], acc[tz][cy]);
  #pragma unroll
              for (int cy = 0; cy < 2; ++cy) acc[tz][cy] = mm(9 + kz, H[buf][cy], acc[tz][cy]);
            }
          };
          if (DB) {
            load_plane(0, 0);
  #pragma unroll
            for (int pl = 0; pl < 4; ++pl) {
              if (pl + 1 < 4) load_plane((pl + 1) & 1, pl + 1);
              __builtin_amdgcn_sched_barrier(0);
              mma_plane(pl & 1, pl);
            }
          } else {
  #pragma unroll
            for (int pl = 0; pl < 4; ++pl) {
              load_plane(0, pl);
              mma_plane(0, pl);
            }
          }
          // taps (0,2,2)|(1,2,2) (low lanes plane tz, high lanes plane tz+1) and (2,2,2) (plane tz+2)
  #pragma unroll
          for (int tz = 0; tz < 2; ++tz) {
            const int sl0 = ((s * TZ + tz) % R) * PLSZ, sl1 = ((s * TZ + tz + 1) % R) * PLSZ, sl2 = ((s * TZ + tz + 2) % R) * PLSZ;
            const int bz = lanebase + (hi ? sl1 : sl0) + koff + (2 * HX + 2) * 16;
            const int b0 = lanebase + sl2 + koff + (2 * HX + 2) * 16;
  #pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
              const vec8 s0 = *(const vec8*)(smem + bz + (cy * HX) * 16);
              const vec8 s1 = *(const vec8*)(smem + b0 + (cy * HX) * 16);
              acc[tz][cy] = mm(12, s0, acc[tz][cy]);
              acc[tz][cy] = mm(13, s1, acc[tz][cy]);
            }
          }
        }
      }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every ring read of this step has returned
    flag_store(done + wave, s + 1);
    AMX_ZSTAMP();                                            // [sweep done]

    // ---- epilogue: activation + store (never waited for)
    if (p.dbg & 4) {
      if (NS) flag_store(staged + wave, s + 1);
      continue;
    }
    if constexpr (NS > 0) {   // the staging tile still holds step s-1 until both storers have read it into registers
      if (s >= SG::NB)
        while (flag_min2(stored) < s + 1 - SG::NB) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    }
    act_inplace<4>(&acc[0][0], p.act, p.slope);
    float pooled[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int tz = 0; tz < 2; ++tz) {
      const int zo = zs + s * TZ + tz;
      if (WIDE && (SPLIT || !(p.dbg & 32))) {
        // Cout = 16, 16-bit channels-last output: the rows cy = 0 / 1 of a lane quartet (g, g ^ 1) are exchanged with one
        // v_permlane16_swap per dword so that every lane owns 8 consecutive channels of ONE voxel: one 16-byte store per lane
        // (a wave instruction = two 512-byte rows) instead of two 8-byte ones.  SPLIT: the same for the lo halves, which sit
        // Cout channels further on in the voxel.
        unsigned pk[2][2], pl[SPLIT ? 2 : 1][2];
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float f = acc[tz][cy][j];
              if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(f);
            v[j] = f;
            pooled[j] = f > pooled[j] ? f : pooled[j];
          }
          pk[cy][0] = (unsigned)to_bits<T>(v[0]) | ((unsigned)to_bits<T>(v[1]) << 16);
          pk[cy][1] = (unsigned)to_bits<T>(v[2]) | ((unsigned)to_bits<T>(v[3]) << 16);
          if constexpr (SPLIT) {
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = v[j] - (float)(T)v[j];
            pl[cy][0] = (unsigned)to_bits<T>(r[0]) | ((unsigned)to_bits<T>(r[1]) << 16);
            pl[cy][1] = (unsigned)to_bits<T>(r[2]) | ((unsigned)to_bits<T>(r[3]) << 16);
          }
        }
        // odd rows of pk[0] <-> even rows of pk[1]: even g keeps row cy = 0 (its own 4 channels + the 4 of g + 1), odd g row cy = 1
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        unsigned l0a = 0, l0b = 0, l1a = 0, l1b = 0;
        if constexpr (SPLIT) {
          const auto t0 = __builtin_amdgcn_permlane16_swap(pl[0][0], pl[1][0], false, false);
          const auto t1 = __builtin_amdgcn_permlane16_swap(pl[0][1], pl[1][1], false, false);
          l0a = t0[0]; l0b = t0[1]; l1a = t1[0]; l1b = t1[1];
        }
        if (zo >= ze) continue;
        if (!full_xy && !((yl + (g & 1) < p.H) & (xl < p.W))) continue;
        *(uint4*)(out_w + (long long)zo * p.oz) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        if constexpr (SPLIT) *(uint4*)(out_w + (long long)zo * p.oz + p.Cout * 2) = make_uint4(l0a, l1a, l0b, l1b);
        continue;
      }
#pragma unroll
      for (int cy = 0; cy < 2; ++cy) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float f = acc[tz][cy][j];
          if (OUTMODE == 0 && RangeCheck<T>::on) bad |= RangeCheck<T>::bad(f);   // the value about to be stored
          v[j] = f;
          pooled[j] = f > pooled[j] ? f : pooled[j];
        }
        if constexpr (NS > 0) {
          if (OUTMODE == 0) {
            char* dst = smem + STAGEOFF + (s % SG::NB) * SG::BYTES + (((tz * TY + wrow + cy) * TX) + wcx * 16 + li) * SG::CB + cb * 2;
            *(uint2*)dst = make_uint2((unsigned)to_bits<T>(v[0]) | ((unsigned)to_bits<T>(v[1]) << 16),
                                      (unsigned)to_bits<T>(v[2]) | ((unsigned)to_bits<T>(v[3]) << 16));
          } else {
            quad_transpose4(v, li);          // lane: channel cb + (li&3), x = (li&~3) .. +3
            char* dst = smem + STAGEOFF + (cb + (li & 3)) * SG::CS + ((tz * TY + wrow + cy) * TX + wcx * 16 + (li & ~3)) * 4;
            *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
          }
          continue;
        }
        if (zo >= ze) continue;
        if (!full_xy && !((yl + cy < p.H) & (xl < p.W))) continue;
        if (OUTMODE == 0) {
          char* dst = out_l + (long long)zo * p.oz + cy * p.oy;
          *(uint2*)dst = make_uint2((unsigned)to_bits<T>(v[0]) | ((unsigned)to_bits<T>(v[1]) << 16),
                                    (unsigned)to_bits<T>(v[2]) | ((unsigned)to_bits<T>(v[3]) << 16));
        } else if (vec_planar) {
          // 4x4 transpose inside each lane quad (DPP, no LDS): lane (li, g) ends up with channel
          // cb + (li&3) at x = (li&~3) .. +3, i.e. ONE 16-byte store per lane instead of four 4-byte ones.
          quad_transpose4(v, li);
          float* dst = out32_q + (long long)zo * p.pz + cy * p.py;
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (p.wmap) {
            const float4 wg = *(const float4*)(p.wmap + ((long long)zo * p.H + yl + cy) * p.W + x0 + wcx * 16 + (li & ~3));
            const float4 old = *(const float4*)dst;
            o = make_float4(old.x + wg.x * o.x, old.y + wg.y * o.y, old.z + wg.z * o.z, old.w + wg.w * o.w);
          }
          *(float4*)dst = o;     // (a nontemporal store here measured 16 % slower: 282 vs 242 us at batch 4)
        } else {
          float* dst = out32_l + (long long)zo * p.pz + cy * p.py;
          if (p.wmap) {
            const float wgt = p.wmap[((long long)zo * p.H + yl + cy) * p.W + xl];
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(long long)j * p.pc] += wgt * v[j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(long long)j * p.pc] = v[j];
          }
        }
      }
    }
    if constexpr (NS > 0) {
      asm volatile("" ::: "memory");                        // LDS is in-order: the flag lands after the staged data
      flag_store(staged + wave, s + 1);
    }
    if constexpr (OUTMODE == 0 && POOL) {
      // the wave's 2 x 2 x 16 block is one max-pool window per x pair: exchange with lane ^ 1, even lanes store.
      // (rounding is monotonic, so pooling the fp32 values equals pooling the stored 16-bit values.)
      float pm[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float o = dpp_quad<0xB1>(pooled[j]);
        pm[j] = pooled[j] > o ? pooled[j] : o;
      }
      const int zp = (zs + s * TZ) >> 1;
      if (!(li & 1) && zs + s * TZ + 1 < ze && (full_xy || ((yl + 1 < p.H) & (xl + 1 < p.W)))) {
        char* dst = out2_l + (long long)zp * p.qz;
        *(uint2*)dst = make_uint2((unsigned)to_bits<T>(pm[0]) | ((unsigned)to_bits<T>(pm[1]) << 16),
                                  (unsigned)to_bits<T>(pm[2]) | ((unsigned)to_bits<T>(pm[3]) << 16));
        if constexpr (SPLIT) {    // hi + lo is monotonic in the fp32 value: the split of the maximum IS the stored maximum
          float r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = pm[j] - (float)(T)pm[j];
          *(uint2*)(dst + p.Cout * 2) = make_uint2((unsigned)to_bits<T>(r[0]) | ((unsigned)to_bits<T>(r[1]) << 16),
                                                   (unsigned)to_bits<T>(r[2]) | ((unsigned)to_bits<T>(r[3]) << 16));
        }
      }
    }
    AMX_ZSTAMP();                                            // [epilogue issued]
  }
  if (OUTMODE == 0 && RangeCheck<T>::on) raise_flag(p.oflow, bad);
}

static thread_local char g_kernel_name3[64] = "";
const char* last_conv_zm_kernel_name() { return g_kernel_name3; }

template <typename T, int NCK, int QT, int TY, int R, int OUTMODE, int NS, bool POOL = false, bool SPLIT = false, int TX = 32, bool STEM = false>
static hipError_t launch_zm_ns(ConvParams p, hipStream_t st, const StemIn& si = StemIn{}) {
  if constexpr (OUTMODE == 0 && NS == 0 && !POOL && !STEM)
    if (p.out2) return launch_zm_ns<T, NCK, QT, TY, R, OUTMODE, NS, true, SPLIT, TX, STEM>(p, st, si);
  constexpr int TZ = 2;
  typedef ZmCfg<SPLIT ? 2 : NCK, QT, TY, TX, R> C;
  constexpr int LDS = STEM ? C::FLAGOFF + 128 + ZmStemCfg<TY, TX>::BYTES : NS ? C::FLAGOFF + 128 + ZmStage<QT, TY, TX, OUTMODE>::TOTAL : C::LDS_BYTES;
  if (STEM)
    snprintf(g_kernel_name3, sizeof g_kernel_name3, "conv3d_k3_zmarch<%s,stem1->16->16,%dx%dx%d,c8+st%d+ld1,r%d/%d%s>",
             __is_same(T, f16) ? "f16" : "bf16", TZ, TY, TX, ZmStemCfg<TY, TX>::NSW, R, ZmStemCfg<TY, TX>::RC, p.out2 ? ",pool" : "");
  else
  snprintf(g_kernel_name3, sizeof g_kernel_name3, "conv3d_k3_zmarch<%s%s,%d->%d,%dx%dx%d,c8+l%d+s%d,r%d,o%d%s>",
           __is_same(T, f16) ? "f16" : "bf16", SPLIT ? "x2" : "", 16 * NCK, 16 * QT, TZ, TY, TX, C::NL, NS, R, OUTMODE,
           p.out2 ? ",pool" : "");
  auto kern = conv3d_k3_zmarch_kernel<T, NCK, QT, TY, TX, R, OUTMODE, NS, POOL, SPLIT, STEM>;
  static amx::DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_once.set();
  }
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = exp_env("AMX_DBG");
    dbg = e ? atoi(e) : 0;
    if (exp_env("AMX_TRACE")) dbg |= 8;
  }
  p.dbg = dbg;
  p.nby = (p.H + TY - 1) / TY;
  p.nbx = (p.W + TX - 1) / TX;
  // z segments: enough workgroups to fill 256 CUs, each segment a multiple of TZ planes, >= 8 planes
  const int tiles = p.nby * p.nbx * p.N;
  int nseg, zseg;
  pick_z_segments(tiles, p.D, TZ, 256 * (LDS <= 80 * 1024 ? 2 : 1), &zseg, &nseg);
  static unsigned long long* trace_buf = nullptr;
  if (p.dbg & 8) {   // debug only: per-phase cycle stamps of consumer waves 0 and 4, printed after a sync
    if (!trace_buf && hipMalloc((void**)&trace_buf, 1024 * 128 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(trace_buf, 0, 1024 * 128 * 8, st);
    p.stats = (float*)trace_buf;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * nseg)), dim3((8 + (STEM ? ZmStemCfg<TY, TX>::NSW + 1 : C::NL) + NS) * 64), LDS, st, p, zseg, nseg, si);
  if (p.dbg & 8) {
    static int printed = 0;
    (void)hipStreamSynchronize(st);
    if (printed++ == 3) {
      static unsigned long long hostbuf[1024 * 128];
      (void)hipMemcpy(hostbuf, trace_buf, sizeof hostbuf, hipMemcpyDeviceToHost);
      const int wgs[4] = {0, 77, 255, 500};                     // (STEM: the slots of workgroup 500 hold stem wave 0 / the loader of workgroup 0)
      for (int wi = 0; wi < (STEM ? 4 : 3); ++wi)
        for (int half = 0; half < 2; ++half) {
          const unsigned long long* tr = hostbuf + ((long long)wgs[wi] * 2 + half) * 128;
          fprintf(stderr, "[trace %s wg %d wave %d] wait/sweep/epilogue:", g_kernel_name3, wgs[wi], half * 4);
          for (int k = 1; k + 2 < 128 && tr[k + 2]; k += 3)
            fprintf(stderr, " %llu/%llu/%llu", tr[k] - tr[k - 1], tr[k + 1] - tr[k], tr[k + 2] - tr[k + 1]);
          fprintf(stderr, "\n");
        }
    }
  }
  return hipGetLastError();
}

// Storer waves need full tiles and dense, 16-byte aligned outputs (whole rows are copied as 16-byte pieces).
template <int QT, int TY, int OUTMODE>
static bool zm_can_stage(const ConvParams& p) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_STORERS") ? 1 : 0;
  if (off || p.H % TY || p.W % 32) return false;
  if (OUTMODE == 0)
    return p.ox == 32 * QT && !((size_t)p.out & 15) && !(p.oy & 15) && !(p.oz & 15) && !(p.on & 15);
  // fp32 planar rows move as float4 pieces; global dwordx4 accesses only need dword alignment on gfx9, so a window that
  // starts at an odd x of the accumulation volume (sliding-window starts 25, 75, 125 ...) is staged too -- its pieces just
  // straddle 16-byte boundaries.  (Unstaged, those windows took the scattered 4-byte read-modify-write path: 197 vs 83 us.)
  static int aligned_only = -1;
  if (aligned_only < 0) aligned_only = exp_env("AMX_STAGE_ALIGNED_ONLY") ? 1 : 0;
  if (aligned_only) return !(p.py & 3) && !(p.pz & 3) && !(p.pc & 3) && !(p.pn & 3) && !((size_t)p.out32 & 15) && !((size_t)p.wmap & 15);
  return !((size_t)p.out32 & 3) && !((size_t)p.wmap & 3);
}

template <typename T, int NCK, int QT, int TY, int R, int OUTMODE>
static hipError_t launch_zm(const ConvParams& p, hipStream_t st) {
  // 12 waves = 3 per SIMD keep the 170-VGPR budget of the MFMA waves; the 32 -> 32 kernel already has 4 loaders
  // Measured (batch 4, 128^3, 16 -> 16): fp32 planar output 253 -> 177 us with storers (whole 128-byte lines, eight
  // rows per store instruction, instead of 64-byte pieces from the MFMA lanes); 16-bit NDHWC output 130 -> 155 us
  // (its direct stores are already 512-byte runs; the staging round trip only adds LDS traffic) -- so planar only.
  static int stage16 = -1;
  if (stage16 < 0) stage16 = exp_env("AMX_STAGE16") ? 1 : 0;
  if constexpr (NCK == 1)
    if ((OUTMODE == 1 || (stage16 && !p.out2)) && zm_can_stage<QT, TY, OUTMODE>(p)) return launch_zm_ns<T, NCK, QT, TY, R, OUTMODE, 2>(p, st);
  return launch_zm_ns<T, NCK, QT, TY, R, OUTMODE, 0>(p, st);
}

// Eligibility: one full-resolution input segment of 16 or 32 channels, 16 or 32 output channels
// (not 32 -> 16), W >= 32; the packed weights must use Q = Cout/16 tiles per group (conv_pick_q does).
bool conv_zmarch_eligible(const ConvParams& p) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_ZMARCH") ? 1 : 0;
  if (off || p.src0_f32c1 || p.C1 != 0 || p.W < 32 || p.H < 8 || p.D < 8) return false;
  if (p.C0 == 16 && p.Cout == 16) return true;
  if (p.out32) return false;                                // planar epilogue only instantiated for 16 -> 16
  return (p.C0 == 16 || p.C0 == 32) && p.Cout == 32;
}

// The fused max-pool needs even extents (every 2x2x2 window inside one wave's block).
bool conv_zmarch_can_pool(const ConvParams& p) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_POOLFUSE") ? 1 : 0;
  return !off && conv_zmarch_eligible(p) && !p.out32 && !(p.D & 1) && !(p.H & 1) && !(p.W & 1);
}

// Strict precision: the 16 -> 16 layers (all of level 0 in the 6M network) have a z-march variant too.
bool conv_zmarch_eligible_split(const ConvParams& p) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_ZMARCH_SPLIT") ? 1 : 0;
  return !off && conv_zmarch_eligible(p) && p.C0 == 16 && p.Cout == 16;
}
bool conv_zmarch_can_pool_split(const ConvParams& p) { return conv_zmarch_can_pool(p) && conv_zmarch_eligible_split(p); }

// Stem-fed 16 -> 16 layer (network.py modules 0..5 of the 6 M model as ONE launch): `p` describes the 16 -> 16 layer (its src0 is
// ignored), the remaining arguments the stem in front of it.  Whole tiles only; single 16-bit precisions.
bool conv_zmarch_stem_eligible(const ConvParams& p, int precision) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_STEMFUSE") ? 1 : 0;
  return !off && precision < 2 && p.C0 == 16 && p.C1 == 0 && p.Cout == 16 && !p.out32 && !p.raw_halo && p.W >= 32 && !(p.W % 32) && !(p.H % 8) &&
         !(p.D & 1) && p.D >= 8 && p.ox == 32 && (p.ocs == 0 || p.ocs == 32);
}
// x_offs (host array of p.N element offsets, or null; at most 16 samples then): sample i reads its volume at x + x_offs[i]
// (sliding-window batches: every window is reflect-padded as its own input).
hipError_t launch_conv_zmarch_stem(const ConvParams& p, const float* x, long long xs_n, long long xs_z, long long xs_y, const long long* x_offs,
                                   const void* stem_wpk, const float* stem_bias, int stem_act, float stem_slope, int precision, hipStream_t st) {
  if (stem_act != ACT_NONE && stem_act != ACT_RELU) return hipErrorInvalidValue;   // the stem waves clip with one v_med3
  if ((precision != 0 && precision != 1) || (x_offs && p.N > 16)) return hipErrorInvalidValue;
  StemIn si;
  memset(&si, 0, sizeof si);
  si.src = (const char*)x; si.sn = xs_n; si.sz = xs_z; si.sy = xs_y;
  if (x_offs) {
    si.use_offs = 1;
    for (int i = 0; i < p.N; ++i) si.offs[i] = x_offs[i] * 4;
  }
  si.wpk = (const char*)stem_wpk; si.bias = stem_bias; si.act = stem_act; si.slope = stem_slope;
  if (precision == 0) return launch_zm_ns<f16, 1, 1, 8, 10, 0, 0, false, false, 32, true>(p, st, si);
  return launch_zm_ns<bf16, 1, 1, 8, 10, 0, 0, false, false, 32, true>(p, st, si);
}

hipError_t launch_conv_zmarch(const ConvParams& p, int precision, hipStream_t st) {
  const bool planar = p.out32 != nullptr;
  if (precision >= 2) {   // strict: 16 -> 16 only; ring of 7 planes x 4 channel planes, stores from the accumulators
    if (p.C0 != 16 || p.Cout != 16) return hipErrorInvalidValue;
    if (precision == 2) return planar ? launch_zm_ns<f16, 1, 1, 8, 7, 1, 0, false, true>(p, st) : launch_zm_ns<f16, 1, 1, 8, 7, 0, 0, false, true>(p, st);
    return planar ? launch_zm_ns<bf16, 1, 1, 8, 7, 1, 0, false, true>(p, st) : launch_zm_ns<bf16, 1, 1, 8, 7, 0, 0, false, true>(p, st);
  }
  if (p.Cout == 16) {
    if (precision == 0) return planar ? launch_zm<f16, 1, 1, 8, 10, 1>(p, st) : launch_zm<f16, 1, 1, 8, 10, 0>(p, st);
    return planar ? launch_zm<bf16, 1, 1, 8, 10, 1>(p, st) : launch_zm<bf16, 1, 1, 8, 10, 0>(p, st);
  }
  // (8 x 16 instead of 4 x 32 (y, x) tiles for the 32-cout layers -- halo 1.41 instead of 1.59 loaded voxels per output, the re-tiling
  //  that gave the normalise-on-load kernel 15 % -- runs (launch_zm_ns<..., TX = 16>) and was measured at batch 4: 32 -> 32 @64^3
  //  722 -> 730 TF, 16 -> 32 unchanged: these layers do not pay for their halo)
  if (p.C0 == 16) return precision == 0 ? launch_zm<f16, 1, 2, 4, 10, 0>(p, st) : launch_zm<bf16, 1, 2, 4, 10, 0>(p, st);
  return precision == 0 ? launch_zm<f16, 2, 2, 4, 10, 0>(p, st) : launch_zm<bf16, 2, 2, 4, 10, 0>(p, st);
}

}  // namespace amx
