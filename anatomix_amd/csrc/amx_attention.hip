// anatomix_amd -- fused attention core of the 3D ViT variant (`anatomix-dev-vit`, PrimusV2-S; BASELINE configs[4]):
//   per-head LayerNorm of q and k (the reference wrapper's ``qk_norm``, anatomix/model/vit3d/architectures.py:108-115)
//   -> rotary position embedding on the patch tokens (timm apply_rot_embed_cat: x*cos + rot(x)*sin, register tokens untouched)
//   -> softmax(q k^T / sqrt(d)) v, flash-style (scores never leave registers), for 4104 tokens x 6 heads x head_dim 66.
// This is 63 % of the model's FLOPs (2 * 2 * 4104^2 * 396 per layer x 12 layers); the plain linears around it stay on the
// vendor GEMM (torch -> hipBLASLt), as the brief allows for plain library GEMMs.
//
// Two kernels:
//   attn_prep   fp32 [b][n][heads*hd] q / k / v -> f16 operands: Qp, Kp [b][h][n_pad][104] (rows padded to 208 B: the 16 lanes of
//               an MFMA fragment then read 16 different bank quads), q pre-multiplied by log2(e) / sqrt(hd); V^T [b][h][80][n_pad].
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

#include "amx_device.h"

namespace amx {


constexpr int kAttKRow = 104;         // halves per Qp / Kp row (208 B)
constexpr int kAttDV = 80;            // head_dim padded to 5 output tiles of 16
constexpr int kAttBN = 64;            // keys per block
constexpr int kAttQT = 2;             // 16-query tiles per wave (measured: QT = 1 halves the work per staged K / V tile and is slower at batch 2)
constexpr int kAttBM = 4 * 16 * kAttQT;  // queries per workgroup (4 waves)
constexpr int kAttPad = 128;          // token padding of the operand buffers
constexpr int kAttVRow = 72;          // halves per V^T row in LDS (144 B)

__host__ __device__ inline int att_npad(int n) { return (n + kAttPad - 1) / kAttPad * kAttPad; }

// grid (ceil(n_pad / 4), heads, b), block 256 = 4 tokens x 64 lanes.  One wave = one token of one head: lane d < hd holds channel d.
__global__ __launch_bounds__(256) void attn_prep_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ qn_w,
                                                        const float* __restrict__ qn_b, const float* __restrict__ kn_w,
                                                        const float* __restrict__ kn_b, float eps, const float* __restrict__ rope,
                                                        int n_prefix, int n, int heads, int hd, f16* __restrict__ Qp,
                                                        f16* __restrict__ Kp, f16* __restrict__ Vt) {
  const int lane = threadIdx.x & 63, tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int h = blockIdx.y, b = blockIdx.z, npad = att_npad(n);
  if (tok >= npad) return;
  const long long row = ((long long)b * heads + h) * npad + tok;
  f16* qo = Qp + row * kAttKRow;
  f16* ko = Kp + row * kAttKRow;
  f16* vo = Vt + ((long long)b * heads + h) * kAttDV * npad + tok;
  if (tok >= n) {                      // padding rows: zeros (their scores are masked, their V columns add nothing)
    for (int d = lane; d < kAttKRow; d += 64) { qo[d] = (f16)0.f; ko[d] = (f16)0.f; }
    for (int d = lane; d < kAttDV; d += 64) vo[(long long)d * npad] = (f16)0.f;
    return;
  }
  const long long src = ((long long)b * n + tok) * heads * hd + (long long)h * hd;
  // lane l holds channels l and l + 64 (head_dim <= 96); a rotation pair (2i, 2i+1) sits in neighbouring lanes of one slot
  const bool a0 = lane < hd, a1 = lane + 64 < hd;
  auto wave_sum = [](float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
  };
  auto norm_rot = [&](const float* x, const float* w, const float* bb, float post, float& r0, float& r1) {
    float v0 = a0 ? x[src + lane] : 0.f, v1 = a1 ? x[src + lane + 64] : 0.f;
    if (w) {                           // nn.LayerNorm(head_dim): biased variance, eps inside the root
      const float mean = wave_sum(v0 + v1) / hd;
      const float d0 = a0 ? v0 - mean : 0.f, d1 = a1 ? v1 - mean : 0.f;
      const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) / hd + eps);
      v0 = a0 ? d0 * rstd * w[lane] + bb[lane] : 0.f;
      v1 = a1 ? d1 * rstd * w[lane + 64] + bb[lane + 64] : 0.f;
    }
    if (rope && tok >= n_prefix) {     // x * cos + rot(x) * sin, rot(x)[2i] = -x[2i+1], rot(x)[2i+1] = x[2i]
      const float o0 = __shfl_xor(v0, 1, 64), o1 = __shfl_xor(v1, 1, 64);
      const float* tb = rope + (long long)(tok - n_prefix) * 2 * hd;
      const float sgn = (lane & 1) ? 1.f : -1.f;
      if (a0) v0 = v0 * tb[hd + lane] + sgn * o0 * tb[lane];
      if (a1) v1 = v1 * tb[hd + lane + 64] + sgn * o1 * tb[lane + 64];
    }
    r0 = v0 * post;
    r1 = v1 * post;
  };
  float q0v, q1v, k0v, k1v;
  norm_rot(q, qn_w, qn_b, 1.4426950408889634f / sqrtf((float)hd), q0v, q1v);
  norm_rot(k, kn_w, kn_b, 1.f, k0v, k1v);
  qo[lane] = (f16)q0v;
  ko[lane] = (f16)k0v;
  if (lane + 64 < kAttKRow) {
    qo[lane + 64] = (f16)(a1 ? q1v : 0.f);
    ko[lane + 64] = (f16)(a1 ? k1v : 0.f);
  }
  vo[(long long)lane * npad] = (f16)(a0 ? v[src + lane] : 0.f);
  if (lane + 64 < kAttDV) vo[(long long)(lane + 64) * npad] = (f16)(a1 ? v[src + lane + 64] : 0.f);
}

// grid (n_pad / kAttBM, heads, b), block 256.  LDS: two (K, V^T) tile buffers -- block t+1 is committed while block t is multiplied,
// one barrier per key block.
__global__ __launch_bounds__(256) void attn_fwd_kernel(const f16* __restrict__ Qp, const f16* __restrict__ Kp,
                                                       const f16* __restrict__ Vt, int n, int heads, int hd, float* __restrict__ out) {
  constexpr int QT = kAttQT, BUF = kAttBN * kAttKRow * 2 + kAttDV * kAttVRow * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, npad = att_npad(n);
  const long long bh = (long long)b * heads + h;
  const int q0 = blockIdx.x * kAttBM + wave * 16 * QT;
  const f16* Kbase = Kp + bh * npad * kAttKRow;
  const f16* Vbase = Vt + bh * kAttDV * npad;

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
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -3.0e38f; l_run[qt] = 0.f; }

  // global -> register prefetch of one key block: K tile = 64 x 208 B contiguous (832 x 16 B), V^T tile = 80 rows x 128 B
  constexpr int KV16 = kAttBN * kAttKRow * 2 / 16, VV16 = kAttDV * 8;
  uint4 pk[4], pv[3];
  auto prefetch = [&](int blk) {
    const char* ks = (const char*)(Kbase + (long long)blk * kAttBN * kAttKRow);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + r * 256;
      if (idx < KV16) pk[r] = *(const uint4*)(ks + idx * 16);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int idx = tid + r * 256;
      if (idx < VV16) pv[r] = *(const uint4*)((const char*)(Vbase + (long long)(idx >> 3) * npad + blk * kAttBN) + (idx & 7) * 16);
    }
  };
  auto commit = [&](int buf) {
    char* sK = smem + buf * BUF;
    char* sV = sK + kAttBN * kAttKRow * 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + r * 256;
      if (idx < KV16) *(uint4*)(sK + idx * 16) = pk[r];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int idx = tid + r * 256;
      if (idx < VV16) *(uint4*)(sV + (idx >> 3) * (kAttVRow * 2) + (idx & 7) * 16) = pv[r];
    }
  };

  const int nblk = (n + kAttBN - 1) / kAttBN;
  prefetch(0);
  commit(0);
  if (nblk > 1) prefetch(1);
  for (int blk = 0; blk < nblk; ++blk) {
    __syncthreads();                   // buffer blk & 1 is complete; every wave has left buffer (blk + 1) & 1 (block blk - 1)
    if (blk + 1 < nblk) {
      commit((blk + 1) & 1);
      if (blk + 2 < nblk) prefetch(blk + 2);
    }
    const char* sK = smem + (blk & 1) * BUF;
    const char* sV = sK + kAttBN * kAttKRow * 2;

    // ---- S^T = K Q^T : acc_s[qt][kt], lane (i, g) holds scores of query i for keys 16 kt + 4 g + j
    f32x4 acc_s[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) acc_s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const f16x8 kf = *(const f16x8*)(sK + (kt * 16 + li) * (kAttKRow * 2) + kk * 64 + g * 16);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc_s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][kk], acc_s[qt][kt], 0, 0, 0);
      }
    if ((blk + 1) * kAttBN > n) {      // keys beyond the sequence (last block only)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (blk * kAttBN + kt * 16 + 4 * g + j >= n) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) acc_s[qt][kt][j] = -3.0e38f;
          }
    }
    // ---- online softmax (exp2 domain), probabilities straight into the B fragments of O^T += V^T P^T
    f16x8 pf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = acc_s[qt][0][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, acc_s[qt][kt][j]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qt], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
      m_run[qt] = m_new;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pr = __builtin_amdgcn_exp2f(acc_s[qt][kt][j] - m_new);
          ps += pr;
          pf[qt][kt >> 1][(kt & 1) * 4 + j] = (f16)pr;
        }
      l_run[qt] = l_run[qt] * alpha + ps;
#pragma unroll
      for (int dt = 0; dt < 5; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_o[qt][dt][j] *= alpha;
    }
    // ---- O^T += V^T P^T : A = V^T rows (dv) with the keys of a K-step in the order {tile 2s: 4g..4g+3, tile 2s+1: 4g..4g+3}
#pragma unroll
    for (int dt = 0; dt < 5; ++dt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const char* vr = sV + (dt * 16 + li) * (kAttVRow * 2) + (ks * 32 + 4 * g) * 2;
        const uint2 lo = *(const uint2*)vr, hi = *(const uint2*)(vr + 32);
        const uint4 raw = make_uint4(lo.x, lo.y, hi.x, hi.y);
        const f16x8 vf = __builtin_bit_cast(f16x8, raw);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt][ks], acc_o[qt][dt], 0, 0, 0);
      }
  }
  // ---- epilogue: the row sums were kept per lane (4 of every 16 keys): add the four lane groups, normalise, store
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = l_run[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
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
  return (size_t)b * heads * npad * (2 * kAttKRow + kAttDV) * sizeof(f16);
}

hipError_t launch_attention(const float* q, const float* k, const float* v, const float* qn_w, const float* qn_b, const float* kn_w,
                            const float* kn_b, float eps, const float* rope, int n_prefix, int b, int n, int heads, int hd,
                            float* out, void* scratch, hipStream_t st) {
  const int npad = att_npad(n);
  f16* Qp = (f16*)scratch;
  f16* Kp = Qp + (size_t)b * heads * npad * kAttKRow;
  f16* Vt = Kp + (size_t)b * heads * npad * kAttKRow;
  hipLaunchKernelGGL(attn_prep_kernel, dim3((npad + 3) / 4, heads, b), dim3(256), 0, st, q, k, v, qn_w, qn_b, kn_w, kn_b, eps, rope,
                     n_prefix, n, heads, hd, Qp, Kp, Vt);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(npad / kAttBM, heads, b), dim3(256), 0, st, Qp, Kp, Vt, n, heads, hd, out);
  return hipGetLastError();
}

}  // namespace amx
