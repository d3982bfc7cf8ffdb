// anatomix_amd -- fused attention core of the 3D ViT variant (`anatomix-dev-vit`, PrimusV2-S; BASELINE configs[4]):
//   per-head LayerNorm of q and k (the reference wrapper's ``qk_norm``, anatomix/model/vit3d/architectures.py:108-115)
//   -> rotary position embedding on the patch tokens (timm apply_rot_embed_cat: x*cos + rot(x)*sin, register tokens untouched)
//   -> softmax(q k^T / sqrt(d)) v, flash-style (scores never leave registers), for 4104 tokens x 6 heads x head_dim 66.
// This is 63 % of the model's FLOPs (2 * 2 * 4104^2 * 396 per layer x 12 layers); the linears around it run on this library's own
// weight-stationary MFMA product kernel (amx_gemm.hip) -- no vendor GEMM anywhere in the forward (amx_vit.hip).
//
// Two kernels:
//   attn_prep   fp32 [b][n][heads*hd] q / k / v -> f16 operands: Qp [b][h][n_pad][104], q pre-multiplied by log2(e) / sqrt(hd);
//               K and V^T per key block of 64 in MFMA FRAGMENT order [b][h][block][fragment][lane][8 halves] (12 + 10 fragments
//               of 1 KiB): attn_fwd reads a fragment with one ds_read_b128 at lane * 16 -- conflict-free.  (Row-major tiles
//               with padded rows, the first layout, measured 46 % of the LDS cycles as bank conflicts: profiles/r03_vit_pmc_sq.)
//               LayerNorm and rotation in fp32, ONE rounding to f16.
//   attn_fwd    one workgroup = 128 queries of one (b, h), 4 waves x 32 queries.  Everything is computed TRANSPOSED so that a
//               lane owns ONE query: S^T = K Q^T (A = K rows from LDS, B = Q^T fragments resident in registers), then each lane
//               holds, for its query, 4 keys of every 16-key tile -- the running max needs two cross-lane exchanges per block,
//               the rescale factor is a per-lane scalar, and the probabilities are already laid out as the B operand of
//               O^T += V^T P^T if the 32 keys of a K-step are taken in the order {tile 2s keys 4g..4g+3, tile 2s+1 keys 4g..4g+3}
//               -- the V^T fragments are read in that same order (two 8-byte LDS reads), so P never goes through LDS.
//               v_mfma_f32_16x16x32_f16, fp32 accumulation, fp32 softmax in the exp2 domain.
#include <math.h>
#include <stdio.h>

#include <type_traits>

#include "amx_device.h"

namespace amx {


constexpr int kAttKRow = 104;         // halves per Qp / Kp row (208 B)
constexpr int kAttDV = 80;            // head_dim padded to 5 output tiles of 16
constexpr int kAttBN = 64;            // keys per block
constexpr int kAttQT = 2;             // 16-query tiles per wave (measured: QT = 1 halves the work per staged K / V tile and is slower at batch 2)
constexpr int kAttBM = 4 * 16 * kAttQT;  // queries per workgroup (4 waves)
constexpr int kAttPad = 128;          // token padding of the operand buffers
constexpr int kAttNLW = 2;            // loader waves
constexpr int kAttNBUF = 3;           // ring depth (key blocks in flight); 5 measured the same: the MFMA waves, not the loaders, set the pace
constexpr int kAttKFrag = 12;         // K fragments per key block: 4 key tiles x 3 K steps (96 padded dims), 1 KiB each
constexpr int kAttVFrag = 10;         // V^T fragments per key block: 5 output tiles x 2 K steps (64 keys)

__host__ __device__ inline int att_npad(int n) { return (n + kAttPad - 1) / kAttPad * kAttPad; }

// grid (n_pad / 64, heads, b), block 256 = 4 waves x 16 tokens: one workgroup = one key block of one head.  One wave handles a
// token at a time, lane d < hd holds channel d (and d + 64).  q rows go straight to global memory (a row is contiguous); the K
// tile (12 KiB) and the V^T tile (10 KiB) of the block are assembled in LDS in fragment order and written out linearly (the
// per-token 2-byte stores at a 144-byte stride of the first version were 2/3 of this kernel's time).
// Row `hd` of the V^T tile is set to ONE for real tokens: the PV product then accumulates the softmax row sum in output row hd
// for free (attn_fwd, ONES) -- head_dim 80 has no spare row and keeps the VALU sum.
__global__ __launch_bounds__(256) void attn_prep_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ qn_w,
                                                        const float* __restrict__ qn_b, const float* __restrict__ kn_w,
                                                        const float* __restrict__ kn_b, float eps, const float* __restrict__ rope,
                                                        int n_prefix, int n, int heads, int hd, int ld, f16* __restrict__ Qp,
                                                        f16* __restrict__ Kp, f16* __restrict__ Vt) {
  // K and V^T tiles of the block in MFMA FRAGMENT order ([fragment][lane][8 halves], the order attn_fwd reads them with one
  // conflict-free ds_read_b128 per fragment): assembled here, written out linearly.
  __shared__ __attribute__((aligned(16))) f16 ktile[kAttKFrag * 512];
  __shared__ __attribute__((aligned(16))) f16 vt[kAttVFrag * 512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y, b = blockIdx.z, npad = att_npad(n);
  for (int i = threadIdx.x; i < kAttKFrag * 64; i += 256) ((uint4*)ktile)[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int i = threadIdx.x; i < kAttVFrag * 64; i += 256) ((uint4*)vt)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const bool a0 = lane < hd, a1 = lane + 64 < hd;
  // wave-wide sum on the VALU (DPP row shifts + row broadcasts, total read from lane 63): the ds_bpermute butterflies of
  // __shfl_xor went through the LDS crossbar, 24 per token, and set this kernel's pace
  auto wave_sum = [](float x) {
    auto dpp = [](float v, auto CTRL, auto ROWS) {
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(CTRL)::value, decltype(ROWS)::value, 0xF, true));
    };
    x += dpp(x, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xF>{});   // row_shr:1
    x += dpp(x, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xF>{});   // row_shr:2
    x += dpp(x, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xF>{});   // row_shr:4
    x += dpp(x, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xF>{});   // row_shr:8 -> lane 15 of a row = row sum
    x += dpp(x, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});   // row_bcast:15 into rows 1, 3
    x += dpp(x, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});   // row_bcast:31 into rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
  };
  const float qpost = 1.4426950408889634f / sqrtf((float)hd);
  float lnw[2][2] = {{1.f, 1.f}, {1.f, 1.f}}, lnb[2][2] = {{0.f, 0.f}, {0.f, 0.f}};     // [q | k][channel lane | lane + 64]: loop invariants
  const bool has_norm[2] = {qn_w != nullptr, kn_w != nullptr};
  if (qn_w) {
    if (a0) { lnw[0][0] = qn_w[lane]; lnb[0][0] = qn_b[lane]; }
    if (a1) { lnw[0][1] = qn_w[lane + 64]; lnb[0][1] = qn_b[lane + 64]; }
  }
  if (kn_w) {
    if (a0) { lnw[1][0] = kn_w[lane]; lnb[1][0] = kn_b[lane]; }
    if (a1) { lnw[1][1] = kn_w[lane + 64]; lnb[1][1] = kn_b[lane + 64]; }
  }
  for (int i0 = 0; i0 < 16; i0 += 4) {
  // four tokens' operands are requested together (24 loads in flight per lane; one at a time exposed the latency 16 times)
  float xin[4][6], rin[4][4];            // rin: rotary table entries {sin, cos} of channels lane, lane + 64 (shared by q and k)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tok = blockIdx.x * kAttBN + wave * 16 + i0 + j;
    const int tc = tok < n ? tok : n - 1;
    const long long src = ((long long)b * n + tc) * ld + (long long)h * hd;     // ld: floats per token row of q / k / v
    xin[j][0] = a0 ? q[src + lane] : 0.f; xin[j][1] = a1 ? q[src + lane + 64] : 0.f;
    xin[j][2] = a0 ? k[src + lane] : 0.f; xin[j][3] = a1 ? k[src + lane + 64] : 0.f;
    xin[j][4] = a0 ? v[src + lane] : 0.f; xin[j][5] = a1 ? v[src + lane + 64] : 0.f;
    rin[j][0] = rin[j][1] = rin[j][2] = rin[j][3] = 0.f;
    if (rope && tc >= n_prefix) {
      const float* tb = rope + (long long)(tc - n_prefix) * 2 * hd;
      if (a0) { rin[j][0] = tb[lane]; rin[j][1] = tb[hd + lane]; }
      if (a1) { rin[j][2] = tb[lane + 64]; rin[j][3] = tb[hd + lane + 64]; }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = i0 + j;
    const int tib = wave * 16 + i, tok = blockIdx.x * kAttBN + tib;          // token in block, token
    const long long row = ((long long)b * heads + h) * npad + tok;
    f16* qo = Qp + row * kAttKRow;
    if (tok >= n) {                      // padding rows: zeros (their scores are masked, their V columns -- incl. the ones row -- add nothing)
      for (int d = lane; d < kAttKRow; d += 64) qo[d] = (f16)0.f;
      continue;
    }
    const float x[2][2] = {{xin[j][0], xin[j][1]}, {xin[j][2], xin[j][3]}};
    const float v0 = xin[j][4], v1 = xin[j][5];
#pragma unroll
    for (int s = 0; s < 2; ++s) {        // s = 0: query, 1: key.  nn.LayerNorm(head_dim): biased variance, eps inside the root
      float y0 = x[s][0], y1 = x[s][1];
      if (has_norm[s]) {
        const float mean = wave_sum(y0 + y1) / hd;
        const float d0 = a0 ? y0 - mean : 0.f, d1 = a1 ? y1 - mean : 0.f;
        const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) / hd + eps);
        y0 = a0 ? d0 * rstd * lnw[s][0] + lnb[s][0] : 0.f;
        y1 = a1 ? d1 * rstd * lnw[s][1] + lnb[s][1] : 0.f;
      }
      if (rope && tok >= n_prefix) {     // x * cos + rot(x) * sin, rot(x)[2i] = -x[2i+1], rot(x)[2i+1] = x[2i]
        const float o0 = dpp_quad<0xB1>(y0), o1 = dpp_quad<0xB1>(y1);          // the rotation partner sits in the neighbouring lane
        const float sgn = (lane & 1) ? 1.f : -1.f;
        if (a0) y0 = y0 * rin[j][1] + sgn * o0 * rin[j][0];
        if (a1) y1 = y1 * rin[j][3] + sgn * o1 * rin[j][2];
      }
      if (s == 0) {
        qo[lane] = (f16)(y0 * qpost);
        if (lane + 64 < kAttKRow) qo[lane + 64] = (f16)(a1 ? y1 * qpost : 0.f);
      } else {                           // key (tile tib / 16, row tib % 16): dim d -> fragment (tile, d / 32), lane ((d % 32) / 8, row), element d % 8
        auto kidx = [&](int d) { return (((tib >> 4) * 3 + (d >> 5)) * 64 + ((d & 31) >> 3) * 16 + (tib & 15)) * 8 + (d & 7); };
        if (a0) ktile[kidx(lane)] = (f16)y0;
        if (a1) ktile[kidx(lane + 64)] = (f16)y1;
      }
    }
    // V^T: dv row d, key tib -> fragment (d / 16, tib / 32), lane (g, d % 16); the 32 keys of a K step are taken in the order
    // {tile 2s keys 4g..4g+3, tile 2s+1 keys 4g..4g+3} (the order the probabilities leave the softmax in, see attn_fwd)
    auto vidx = [&](int d) {
      const int kq = tib & 31, gq = (kq & 15) >> 2, e = (kq >> 4) * 4 + (kq & 3);
      return (((d >> 4) * 2 + (tib >> 5)) * 64 + gq * 16 + (d & 15)) * 8 + e;
    };
    if (a0) vt[vidx(lane)] = (f16)v0;
    if (a1) vt[vidx(lane + 64)] = (f16)v1;
    if (lane == 0 && hd < kAttDV) vt[vidx(hd)] = (f16)1.f;
  }
  }
  __syncthreads();
  const long long blk = ((long long)b * heads + h) * (npad / kAttBN) + blockIdx.x;     // [b][h][key block][fragment][lane][8 halves]
  uint4* kd = (uint4*)(Kp + blk * (kAttKFrag * 512));
  uint4* vd = (uint4*)(Vt + blk * (kAttVFrag * 512));
  for (int i = threadIdx.x; i < kAttKFrag * 64; i += 256) kd[i] = ((const uint4*)ktile)[i];
  for (int i = threadIdx.x; i < kAttVFrag * 64; i += 256) vd[i] = ((const uint4*)vt)[i];
}

typedef __attribute__((address_space(1))) const void* att_gptr_t;
typedef __attribute__((address_space(3))) void* att_lptr_t;

// grid (n_pad / kAttBM, heads, b), block (4 + kAttNLW) * 64 = 4 MFMA waves (32 queries each) + kAttNLW LOADER waves.
// K tile (64 keys x 208 B) and V^T tile (80 rows x 144 B, stored block-major with the padded rows already in place) are each one
// contiguous chunk of global memory: the loader wave moves them with LDS-DMA (1 KiB per instruction, no VGPR round trip) into a
// ring of three buffers, runs up to two key blocks ahead with counted vmcnt waits and meets the MFMA waves through LDS counters
// (amx_device.h) -- no workgroup barrier in the loop.
template <bool ONES>
__global__ __launch_bounds__((4 + kAttNLW) * 64) void attn_fwd_kernel(const f16* __restrict__ Qp, const f16* __restrict__ Kp,
                                                       const f16* __restrict__ Vt, int n, int heads, int hd, float* __restrict__ out) {
  constexpr int QT = kAttQT, KB = kAttKFrag * 1024, VB = kAttVFrag * 1024, BUF = KB + VB, NBUF = kAttNBUF;
  constexpr int NDMA = (BUF + 1023) / 1024, NLW = kAttNLW;  // 1 KiB pieces per key block (the last one partial)
  constexpr int PER = (NDMA + NLW - 1) / NLW;               // pieces per loader wave and block (waves without a last piece pad the count)
  static_assert(BUF % 16 == 0 && PER * 2 <= 60 && NLW <= 8, "tile size / vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF * BUF + 64 B of counters
  int* ready = (int*)(smem + NBUF * BUF);                 // per loader wave: key blocks landed (8 slots, unused = INT_MAX)
  int* done = ready + 8;                                   // per MFMA wave: key blocks finished (8 slots, unused = INT_MAX)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, npad = att_npad(n);
  const long long bh = (long long)b * heads + h;
  const int nblk = (n + kAttBN - 1) / kAttBN, nblk_pad = npad / kAttBN;
  if (tid < 16) ready[tid] = ((tid >= NLW && tid < 8) || tid >= 8 + 4) ? 0x7fffffff : 0;
  __syncthreads();

  if (wave >= 4) {
    // =============================== loader wave lw: pieces j = lw, lw + NLW, ... ===============================
    const int lw = wave - 4;
    const char* kbase = (const char*)(Kp + bh * nblk_pad * (kAttKFrag * 512));
    const char* vbase = (const char*)(Vt + bh * nblk_pad * (kAttVFrag * 512));
    const unsigned a_ready = lds_addr(ready + lw), a_done = lds_addr(done);
    for (int blk = 0; blk < nblk; ++blk) {
      if (blk >= NBUF)                                    // buffer blk % NBUF is free once every MFMA wave finished block blk - NBUF
        while (__builtin_amdgcn_readfirstlane(flag_min8_asm(a_done)) < blk - NBUF + 1) __builtin_amdgcn_s_sleep(1);
      char* buf = smem + (blk % NBUF) * BUF;
      const char* ks = kbase + (long long)blk * KB;
      const char* vs = vbase + (long long)blk * VB;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int j = lw + k * NLW;
        int off = j * 1024 + lane * 16;                    // K tile first, V^T tile behind it -- the same order as in the buffer
        // a wave without a piece in the last round re-fetches its first one: every wave then has exactly PER instructions per
        // block in flight, which is what the counted wait below relies on
        const int jj = j * 1024 < BUF ? j : lw;
        off = jj * 1024 + lane * 16;
        if (off < BUF) {
          const char* src = off < KB ? ks + off : vs + (off - KB);
          __builtin_amdgcn_global_load_lds((att_gptr_t)src, (att_lptr_t)(buf + jj * 1024), 16, 0, 0);
        }
      }
      if (blk > 0) {                                      // block blk - 1 has landed once only this block's pieces are in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        flag_store_asm(a_ready, blk);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flag_store_asm(a_ready, nblk);
    return;
  }

  const int q0 = blockIdx.x * kAttBM + wave * 16 * QT;
  // Q^T B fragments: lane (i, g) holds Q[query q0 + 16 qt + i][32 kk + 8 g .. + 7]
  f16x8 qf[QT][3];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
      qf[qt][kk] = *(const f16x8*)(Qp + (bh * npad + q0 + qt * 16 + li) * kAttKRow + kk * 32 + g * 8);
  f32x4 acc_o[QT][5];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) acc_o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Softmax reference point m_ref per query (exp2 domain), LAZY: the scores leave the MFMA already shifted (the first K step
  // accumulates onto -m_ref), and m_ref only moves when a score exceeds it by more than kLag (then p <= 2^kLag stays far inside
  // f16) -- after the first key blocks that is rare, and the common path has no subtraction, no rescale of the output
  // accumulators and no cross-lane traffic.  ONES: the row sum comes out of the PV product (V^T row hd is all ones).
  constexpr float kLag = 8.f;
  float m_ref[QT], l_run[QT];
  f32x4 negm[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m_ref[qt] = 0.f; l_run[qt] = 0.f; negm[qt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // max over the four lanes that share a query (lane, lane ^ 16, lane ^ 32): row swaps, VALU only
  auto max4 = [](float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = fmaxf(__builtin_bit_cast(float, s16[0]), __builtin_bit_cast(float, s16[1]));
    const unsigned w = __builtin_bit_cast(unsigned, x);
    const auto s32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__builtin_bit_cast(float, s32[0]), __builtin_bit_cast(float, s32[1]));
  };

  for (int blk = 0; blk < nblk; ++blk) {
    while (true) {                     // every loader wave's share of this block has landed
      int m = flag_load(ready);
#pragma unroll
      for (int i = 1; i < NLW; ++i) {
        const int r = flag_load(ready + i);
        m = r < m ? r : m;
      }
      if (m >= blk + 1) break;
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    const char* sK = smem + (blk % NBUF) * BUF;
    const char* sV = sK + KB;

    // ---- S^T = K Q^T : acc_s[qt][kt], lane (i, g) holds scores of query i for keys 16 kt + 4 g + j
    f32x4 acc_s[QT][4];
    f16x8 kfr[4][3];                   // every K fragment of the block is requested before the first MFMA (12 reads in flight, counted waits)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) kfr[kt][kk] = *(const f16x8*)(sK + ((kt * 3 + kk) * 64 + lane) * 16);
    __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise re-interleaves them two at a time with a full wait before each MFMA pair
    // K step outermost: 8 independent accumulators between two MFMAs on the same one (kt-major order left ONE, and a dependent
    // MFMA cannot issue before its predecessor's passes are through)
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          acc_s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[kt][kk], qf[qt][kk], kk == 0 ? negm[qt] : acc_s[qt][kt], 0, 0, 0);
      }
    // V^T fragments of the block: requested now, needed after the softmax arithmetic
    f16x8 vfr[5][2];
#pragma unroll
    for (int dt = 0; dt < 5; ++dt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        vfr[dt][ks] = *(const f16x8*)(sV + ((dt * 2 + ks) * 64 + lane) * 16);
      }
    if ((blk + 1) * kAttBN > n) {      // keys beyond the sequence (last block only)
      asm volatile("" ::: "memory");     // keeps this a BRANCH: if-converted it was 80 select / compare instructions in every block
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (blk * kAttBN + kt * 16 + 4 * g + j >= n) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) acc_s[qt][kt][j] = -3.0e38f;
          }
    }
    // ---- softmax (exp2 domain), probabilities straight into the B fragments of O^T += V^T P^T
    float mx[QT];
    bool over = blk == 0;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      mx[qt] = acc_s[qt][0][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx[qt] = fmaxf(mx[qt], acc_s[qt][kt][j]);
      over |= mx[qt] > kLag;
    }
    if (__builtin_amdgcn_ballot_w64(over) != 0) {          // wave-uniform; rare after the first blocks
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const float mq = max4(mx[qt]);                      // the four lanes of a query move together
        const float delta = (blk == 0 || mq > kLag) ? mq : 0.f;
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_ref[qt] += delta;
        negm[qt] = f32x4{-m_ref[qt], -m_ref[qt], -m_ref[qt], -m_ref[qt]};
        l_run[qt] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc_s[qt][kt][j] -= delta;
#pragma unroll
        for (int dt = 0; dt < 5; ++dt)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc_o[qt][dt][j] *= alpha;
      }
    }
    f16x8 pf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pr = __builtin_amdgcn_exp2f(acc_s[qt][kt][j]);
          if (!ONES) ps += pr;
          pf[qt][kt >> 1][(kt & 1) * 4 + j] = (f16)pr;
        }
      if (!ONES) l_run[qt] += ps;
    }
    // ---- O^T += V^T P^T : A = V^T rows (dv) with the keys of a K-step in the order {tile 2s: 4g..4g+3, tile 2s+1: 4g..4g+3}
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfr[dt][ks], pf[qt][ks], acc_o[qt][dt], 0, 0, 0);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every read of this block's tiles has returned: the buffer may be refilled
    flag_store(done + wave, blk + 1);
  }
  // ---- epilogue: the row sums were kept per lane (4 of every 16 keys): add the four lane groups, normalise, store
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = l_run[qt];
    if (ONES) {                          // output row hd of this query = sum of its (f16-rounded) probabilities: held by lane group (hd % 16) / 4
      float cand = 0.f;
#pragma unroll
      for (int dt = 0; dt < 5; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (dt == (hd >> 4) && j == (hd & 3)) cand = acc_o[qt][dt][j];
      l = __shfl(cand, li + 16 * ((hd & 15) >> 2), 64);
    } else {
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.f / l;
    const int qi = q0 + qt * 16 + li;
    if (qi >= n) continue;
    float* o = out + ((long long)b * n + qi) * heads * hd + (long long)h * hd;
#pragma unroll
    for (int dt = 0; dt < 5; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dv = dt * 16 + 4 * g + j;
        if (dv < hd) o[dv] = acc_o[qt][dt][j] * inv;
      }
  }
}

size_t attention_scratch_bytes(int b, int heads, int n) {
  const size_t npad = att_npad(n);
  return (size_t)b * heads * (npad * kAttKRow + (npad / kAttBN) * (kAttKFrag + kAttVFrag) * 512) * sizeof(f16);
}

// operand buffers inside the scratch allocation (the ViT engine's q | k and v projections write them directly, amx_gemm.hip EPI_QK / EPI_VT)
void attention_operands(void* scratch, int b, int heads, int n, void** Qp, void** Kp, void** Vt, int* npad_out, int* nblk_pad_out) {
  const int npad = att_npad(n);
  f16* q = (f16*)scratch;
  f16* k = q + (size_t)b * heads * npad * kAttKRow;
  f16* v = k + (size_t)b * heads * (npad / kAttBN) * (kAttKFrag * 512);
  *Qp = q; *Kp = k; *Vt = v;
  if (npad_out) *npad_out = npad;
  if (nblk_pad_out) *nblk_pad_out = npad / kAttBN;
}

// softmax(q k^T) v on prepared operands
hipError_t launch_attention_fwd(const void* Qp, const void* Kp, const void* Vt, int b, int n, int heads, int hd, float* out, hipStream_t st) {
  const int npad = att_npad(n);
  constexpr int LDS = kAttNBUF * (kAttKFrag + kAttVFrag) * 1024 + 64;
  static amx::DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_once.set();
  }
  if (hd < kAttDV)
    hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(npad / kAttBM, heads, b), dim3((4 + kAttNLW) * 64), LDS, st, (const f16*)Qp, (const f16*)Kp,
                       (const f16*)Vt, n, heads, hd, out);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(npad / kAttBM, heads, b), dim3((4 + kAttNLW) * 64), LDS, st, (const f16*)Qp, (const f16*)Kp,
                       (const f16*)Vt, n, heads, hd, out);
  return hipGetLastError();
}

hipError_t launch_attention_ld(const float* q, const float* k, const float* v, int ld, const float* qn_w, const float* qn_b, const float* kn_w,
                               const float* kn_b, float eps, const float* rope, int n_prefix, int b, int n, int heads, int hd,
                               float* out, void* scratch, hipStream_t st) {
  void *Qp, *Kp, *Vt;
  int npad;
  attention_operands(scratch, b, heads, n, &Qp, &Kp, &Vt, &npad, nullptr);
  hipLaunchKernelGGL(attn_prep_kernel, dim3(npad / kAttBN, heads, b), dim3(256), 0, st, q, k, v, qn_w, qn_b, kn_w, kn_b, eps, rope,
                     n_prefix, n, heads, hd, ld, (f16*)Qp, (f16*)Kp, (f16*)Vt);
  return launch_attention_fwd(Qp, Kp, Vt, b, n, heads, hd, out, st);
}

hipError_t launch_attention(const float* q, const float* k, const float* v, const float* qn_w, const float* qn_b, const float* kn_w,
                            const float* kn_b, float eps, const float* rope, int n_prefix, int b, int n, int heads, int hd,
                            float* out, void* scratch, hipStream_t st) {
  return launch_attention_ld(q, k, v, heads * hd, qn_w, qn_b, kn_w, kn_b, eps, rope, n_prefix, b, n, heads, hd, out, scratch, st);
}

}  // namespace amx
