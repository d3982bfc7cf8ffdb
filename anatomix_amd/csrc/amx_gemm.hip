// anatomix_amd -- token-matrix products of the 3D ViT variant (`anatomix-dev-vit`): every nn.Linear of the EVA blocks
// (q/k/v, attention output, SwiGLU fc1_g/fc1_x, fc2), the tokenizer's 1x1x1 embedding projection and the decoder's
// 2x2x2 stride-2 transposed convolutions (a ConvTranspose3d with kernel == stride is a product of the voxel matrix with a
// [Cin][8 Cout] matrix followed by a scatter) -- reference: anatomix/model/vit3d/architectures.py:89-165, 231-260 and the
// upstream blocks restated in oracle/vit_ref.py.
//
//   C^T[n][m] = sum_k W[n][k] X[m][k]      v_mfma_f32_16x16x32_f16, fp32 accumulate
//
// The WEIGHTS are the A operand (rows = output features n), the token rows the B operand (columns = tokens m): a lane then
// owns FOUR CONSECUTIVE output features of one token, so every epilogue (bias, LayerScale residual, SwiGLU, scatter) works
// on 8 / 16 contiguous bytes of a row.
//
//   weights   packed once into fragment order [n tile][k step][lane][8 halves] (1 KiB per fragment, one coalesced load);
//   tokens    row-major f16 [M][lda], lda a multiple of 32 with zero padding columns (the LayerNorm kernels below write that
//             layout); lane (j, g) of a fragment reads 16 bytes of row j -- no LDS, the operands of a K step go straight from
//             L2 / L1 to registers, double buffered one K step ahead.  A workgroup is 4 waves stacked along M, each wave owns
//             MT x NT tiles (64 x 64 / 64 x 80 outputs): 2 (MT + NT) fragment loads feed MT * NT MFMAs.
//   SPLIT     hi + lo f16 pairs for both operands, three MFMAs per product (fp32-grade; the tokenizer / decoder ends of the
//             network need it, the blocks do not -- measured in DESIGN.md section 8).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "amx_device.h"
#include "amx_gemm.h"

namespace amx {

// erf with |error| < 1.5e-7 (Abramowitz-Stegun 7.1.26): one rcp, one exp, a degree-5 polynomial -- GELU in a product kernel's epilogue
// must not cost more VALU time than the MFMAs it follows (libm's erff: ~70 instructions with both branches taken per wave)
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = __builtin_fabsf(x), t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float r = 1.f - poly * __expf(-ax * ax);
  return __builtin_copysignf(r, x);
}
__device__ __forceinline__ float gelu_exact(float y) { return 0.5f * y * (1.f + erf_fast(y * 0.70710678118654752f)); }

// Epilogues of the product kernels: `acc[t][u]` = tile (feature tile nt0 + t, row tile at m0 + 16 u) of this wave.
template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[NT][MT], const int m0, const int nt0, const int li, const int g) {
  if (EPI == EPI_PLANAR) {
    // lane (li, g) holds feature n = 16 tile + li of voxels m .. m + 3 (one x run); features are ordered ((dz, dy), c, dx), so the
    // neighbouring lane holds the other x parity of the same channel: one quad exchange turns 2 x 4 strided values into two float4
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const int m = m0 + u * 16 + 4 * g;
      if (m >= p.M) continue;
      const int x = m % p.gw, r1 = m / p.gw, y = r1 % p.gh, r2 = r1 / p.gh, z = r2 % p.gd, b = r2 / p.gd;
      const bool even = !(li & 1);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (nt0 + t >= p.ntiles) continue;
        const int n = (nt0 + t) * 16 + li, q = n >> 1, c = q % p.Cp, zy = q / p.Cp;
        const float add = c < p.Creal ? p.bias[c] - (p.sub ? p.sub[b * p.Creal + c] : 0.f) : 0.f;
        const float v0 = acc[t][u][0] + add, v1 = acc[t][u][1] + add, v2 = acc[t][u][2] + add, v3 = acc[t][u][3] + add;
        const float o0 = dpp_quad<0xB1>(even ? v2 : v0), o1 = dpp_quad<0xB1>(even ? v3 : v1);
        if (c >= p.Creal) continue;
        const long long row = (((long long)b * p.Creal + c) * 2 * p.gd + 2 * z + (zy >> 1)) * 2 * p.gh + 2 * y + (zy & 1);
        float* o = (float*)p.out + row * 2 * p.gw + 2 * x + (even ? 0 : 4);
        *(float4*)o = even ? make_float4(v0, o0, v1, o1) : make_float4(o0, v2, o1, v3);
      }
    }
    return;
  }
  if (EPI == EPI_QK) {
    // One column block = one (q | k, head): NT = 5 tiles = the head's 80 padded features.  Lane (li, g) holds features
    // d = 16 t + 4 g .. + 3 of token row li: bias, per-head LayerNorm (sum over the lane's 20 values + the 4 lane groups),
    // rotary embedding (the pair (2 i, 2 i + 1) sits in ONE lane), then the f16 operand layouts of attn_fwd directly.
    const int sel = nt0 / (p.heads * NT), h = (nt0 / NT) % p.heads;
    const float* nw = sel ? p.knw : p.qnw;
    const float* nb = sel ? p.knb : p.qnb;
    // every load of the epilogue first, unconditionally (see the generic epilogue): bias, norm parameters, rotary entries
    float4 bias4[NT], w4a[NT], o4a[NT], sc[MT][NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bias4[t] = *(const float4*)(p.bias + (nt0 + t) * 16 + 4 * g);
      w4a[t] = nw ? *(const float4*)(nw + t * 16 + 4 * g) : make_float4(1.f, 1.f, 1.f, 1.f);       // padded to Cp floats, zeros behind hd
      o4a[t] = nw ? *(const float4*)(nb + t * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (p.rope) {
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        int m = m0 + u * 16 + li;
        m = m < p.M ? m : p.M - 1;
        int rt = m % p.T - p.n_prefix;
        rt = rt > 0 ? rt : 0;
        const float4* tb = (const float4*)p.rope + (long long)rt * (p.hd / 2);     // per pair {sin, sin, cos, cos}
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int pr = (t * 16 + 4 * g) / 2 + r;
            sc[u][t][r] = tb[pr < p.hd / 2 ? pr : p.hd / 2 - 1];
          }
      }
    }
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const int m = m0 + u * 16 + li;
      const bool live = m < p.M;
      const int mm = live ? m : p.M - 1, b = mm / p.T, tok = mm % p.T;
      float v[NT][4];
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = t * 16 + 4 * g;
        const float4 b4 = bias4[t];
        v[t][0] = acc[t][u][0] + b4.x; v[t][1] = acc[t][u][1] + b4.y; v[t][2] = acc[t][u][2] + b4.z; v[t][3] = acc[t][u][3] + b4.w;
#pragma unroll
        for (int r = 0; r < 4; ++r) s += c + r < p.hd ? v[t][r] : 0.f;
      }
      if (nw) {                          // nn.LayerNorm(head_dim): biased variance, eps inside the root
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = s / p.hd;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = t * 16 + 4 * g + r < p.hd ? v[t][r] - mean : 0.f;
            v[t][r] = d;
            q += d * d;
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q / p.hd + p.att_eps);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 w4 = w4a[t], o4 = o4a[t];
          v[t][0] = v[t][0] * rstd * w4.x + o4.x; v[t][1] = v[t][1] * rstd * w4.y + o4.y;
          v[t][2] = v[t][2] * rstd * w4.z + o4.z; v[t][3] = v[t][3] * rstd * w4.w + o4.w;
        }
      }
      if (p.rope && tok >= p.n_prefix) {   // x * cos + rot(x) * sin, rot(x)[2i] = -x[2i+1], rot(x)[2i+1] = x[2i]
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const float4 q4 = sc[u][t][r >> 1];
            const float x0 = v[t][r], x1 = v[t][r + 1];          // features behind hd are zero and are written as zero below
            v[t][r] = x0 * q4.z - x1 * q4.x;
            v[t][r + 1] = x1 * q4.w + x0 * q4.y;
          }
      }
      if (!live) continue;
      const float post = sel ? 1.f : p.qscale;
      const long long bh = (long long)b * p.heads + h;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = t * 16 + 4 * g;
        unsigned short o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = to_bits<f16>(c + r < p.hd ? v[t][r] * post : 0.f);
        const uint2 pk = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
        if (sel == 0) {                  // Qp [b][h][n_pad][104 halves]
          *(uint2*)((f16*)p.Qp + (bh * p.npad + tok) * 104 + c) = pk;
        } else {                         // K fragment tile of key block tok / 64: dim d -> fragment (tile, d / 32), lane ((d % 32) / 8, row), element d % 8
          const int kb = tok >> 6, tib = tok & 63;
          const long long idx = (((bh * p.nblk_pad + kb) * 12 + (tib >> 4) * 3 + (c >> 5)) * 64 + ((c & 31) >> 3) * 16 + (tib & 15)) * 8 + (c & 7);
          *(uint2*)((f16*)p.Kp + idx) = pk;
        }
      }
    }
    return;
  }
  if (EPI == EPI_VT) {
    // Token rows are the A operand: lane (li, g) holds feature n = 16 tile + li (head n / Cp, V^T row d = n % Cp) of tokens m .. m + 3.
    // V^T fragment (d / 16, key / 32): lane ((key % 16) / 4, d % 16), element (key % 32 / 16) * 4 + key % 4 -- four consecutive keys
    // of one lane group are 8 contiguous bytes.  Row d == hd is the ONES row (softmax row sum through the PV product).
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const int mb = m0 + u * 16 + 4 * g;
      if (mb >= p.M) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (nt0 + t >= p.ntiles) continue;
        const int n = (nt0 + t) * 16 + li, h = n / p.Cp, d = n % p.Cp;
        if (d > p.hd || h >= p.heads) continue;           // rows behind the ones row stay zero (cleared once per forward)
        const float bn = p.bias[n];
        unsigned short o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = to_bits<f16>(d == p.hd ? 1.f : acc[t][u][r] + bn);
        if ((p.T & 3) == 0) {            // the four tokens share sample, key block and lane group
          const int b = mb / p.T, tok = mb % p.T, kb = tok >> 6, tib = tok & 63, kq = tib & 31;
          const long long idx = (((((long long)b * p.heads + h) * p.nblk_pad + kb) * 10 + (d >> 4) * 2 + (tib >> 5)) * 64 + ((kq & 15) >> 2) * 16 + (d & 15)) * 8 + (kq >> 4) * 4;
          *(uint2*)((f16*)p.Vt + idx) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mb + r;
            if (m >= p.M) continue;
            const int b = m / p.T, tok = m % p.T, kb = tok >> 6, tib = tok & 63, kq = tib & 31;
            const long long idx = (((((long long)b * p.heads + h) * p.nblk_pad + kb) * 10 + (d >> 4) * 2 + (tib >> 5)) * 64 + ((kq & 15) >> 2) * 16 + (d & 15)) * 8 + (kq >> 4) * 4 + (kq & 3);
            ((unsigned short*)p.Vt)[idx] = o[r];
          }
        }
      }
    }
    return;
  }
  if (EPI == EPI_SCATTER_LN) {
    // NT == Cp / 16: the wave holds every channel of its rows for one parity: channel LayerNorm + GELU in registers, the hi / lo
    // operand rows of the next transposed conv are stored directly (the fp32 tensor between the two never exists).
    // Pass 1 (per row tile): bias, mean, deviations in place, rstd.  Pass 2 (per channel tile): the LayerNorm parameters are
    // loaded ONCE per tile as float4 and applied to every row tile (per-value loads were 128 load instructions per wave and tile).
    const int par = nt0 / NT;
    float rstd[MT];
    long long vox[MT];
    bool live[MT];
    float4 bias4[NT], w4a[NT], o4a[NT];    // every load of the epilogue first (see the generic epilogue)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = t * 16 + 4 * g, cc = c + 3 < p.Creal ? c : 0;      // tiles that straddle Creal take the scalar path below
      bias4[t] = *(const float4*)(p.bias + c);
      w4a[t] = *(const float4*)(p.lnw + cc);
      o4a[t] = *(const float4*)(p.lnb + cc);
    }
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const int m = m0 + u * 16 + li;
      live[u] = m < p.M;
      const int mm = live[u] ? m : p.M - 1;
      const int x = mm % p.gw, r1 = mm / p.gw, y = r1 % p.gh, r2 = r1 / p.gh, z = r2 % p.gd, b = r2 / p.gd;
      vox[u] = ((((long long)b * 2 * p.gd + 2 * z + (par >> 2)) * 2 * p.gh + 2 * y + ((par >> 1) & 1)) * 2 * p.gw + 2 * x + (par & 1));
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = t * 16 + 4 * g;
        const float4 b4 = bias4[t];
        acc[t][u][0] += b4.x; acc[t][u][1] += b4.y; acc[t][u][2] += b4.z; acc[t][u][3] += b4.w;
#pragma unroll
        for (int r = 0; r < 4; ++r) s += c + r < p.Creal ? acc[t][u][r] : 0.f;
      }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mean = s / p.Creal;
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = t * 16 + 4 * g + r < p.Creal ? acc[t][u][r] - mean : 0.f;
          acc[t][u][r] = d;
          q += d * d;
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      rstd[u] = rsqrtf(q / p.Creal + p.eps);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = t * 16 + 4 * g;
      if (c >= p.ldo) continue;
      float w4[4] = {0.f, 0.f, 0.f, 0.f}, o4[4] = {0.f, 0.f, 0.f, 0.f};
      if (c + 3 < p.Creal) {
        const float4 a4 = w4a[t], c4 = o4a[t];
        w4[0] = a4.x; w4[1] = a4.y; w4[2] = a4.z; w4[3] = a4.w;
        o4[0] = c4.x; o4[1] = c4.y; o4[2] = c4.z; o4[3] = c4.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c + r < p.Creal) { w4[r] = p.lnw[c + r]; o4[r] = p.lnb[c + r]; }
      }
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        if (!live[u]) continue;
        unsigned short hb[4], lb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float yv = c + r < p.Creal ? gelu_exact(acc[t][u][r] * rstd[u] * w4[r] + o4[r]) : 0.f;
          const f16 yh = (f16)yv;
          hb[r] = __builtin_bit_cast(unsigned short, yh);
          lb[r] = to_bits<f16>(yv - (float)yh);
        }
        *(uint2*)((f16*)p.out + vox[u] * p.ldo + c) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
        if (p.out_lo) *(uint2*)((f16*)p.out_lo + vox[u] * p.ldo + c) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
      }
    }
    return;
  }
  // EVERY global load of the epilogue is issued first, unconditionally (clamped indices), into registers: loads placed under the
  // per-tile conditions below were each followed by a full wait -- ten to twenty exposed memory latencies per row group.
  float4 bias4[NT], gam4[NT], extra[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int nt = nt0 + t < p.ntiles ? nt0 + t : p.ntiles - 1, n = nt * 16 + 4 * g;
    bias4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    gam4[t] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (EPI == EPI_SCATTER) bias4[t] = *(const float4*)(p.bias + n % p.Cp);
    else if (p.bias) bias4[t] = *(const float4*)(p.bias + n);
    if (EPI == EPI_RESID && p.gamma) gam4[t] = *(const float4*)(p.gamma + n);
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      int m = m0 + u * 16 + li;
      m = m < p.M ? m : p.M - 1;
      const int nn = n < p.Nreal ? n : 0;
      if (EPI == EPI_RESID) extra[t][u] = *(const float4*)((const float*)p.out + (long long)m * p.ldo + nn);
      else if (EPI == EPI_TOKENS) extra[t][u] = *(const float4*)(p.pos + (long long)(m % p.V) * p.ldo + nn);
      else extra[t][u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int u = 0; u < MT; ++u) {
    const int m = m0 + u * 16 + li;
    if (m >= p.M) continue;
    long long row_off = 0;
    if (EPI == EPI_SCATTER) {        // m = ((b gd + z) gh + y) gw + x
      const int x = m % p.gw, r1 = m / p.gw, y = r1 % p.gh, r2 = r1 / p.gh, z = r2 % p.gd, b = r2 / p.gd;
      row_off = ((((long long)b * 2 * p.gd + 2 * z) * 2 * p.gh + 2 * y) * 2 * p.gw + 2 * x);    // voxel index of parity (0, 0, 0)
    } else if (EPI == EPI_TOKENS) {
      const int b = m / p.V, v = m % p.V;
      row_off = (long long)b * (p.V + p.nreg) + p.nreg + v;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (nt0 + t >= p.ntiles) continue;
      const int n = (nt0 + t) * 16 + 4 * g;
      f32x4 v = acc[t][u];
      const float4 b4 = bias4[t];
      if (EPI == EPI_SWIGLU) {
        if (t & 1) continue;                                        // tile pair (2s, 2s + 1) = (gate, value) of hidden columns 16 s ..
        if (t + 1 < NT) {
          const f32x4 x = acc[t + 1 < NT ? t + 1 : t][u];
          const float4 bx = bias4[t + 1 < NT ? t + 1 : t];
          const int hcol = (nt0 + t) / 2 * 16 + 4 * g;
          const float gv[4] = {v[0] + b4.x, v[1] + b4.y, v[2] + b4.z, v[3] + b4.w};
          const float xv[4] = {x[0] + bx.x, x[1] + bx.y, x[2] + bx.z, x[3] + bx.w};
          unsigned short o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = to_bits<f16>(gv[j] / (1.f + __expf(-gv[j])) * xv[j]);
          *(uint2*)((f16*)p.out + (long long)m * p.ldo + hcol) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
        }
        continue;
      }
      if (EPI == EPI_SCATTER) {
        const int par = n / p.Cp, c = n % p.Cp;
        if (c >= p.Creal) continue;
        const long long vox = row_off + ((long long)(par >> 2) * 2 * p.gh + ((par >> 1) & 1)) * 2 * p.gw + (par & 1);
        *(float4*)((float*)p.out + vox * p.Creal + c) = make_float4(v[0] + b4.x, v[1] + b4.y, v[2] + b4.z, v[3] + b4.w);
        continue;
      }
      if (n >= p.Nreal) continue;
      float r[4] = {v[0] + b4.x, v[1] + b4.y, v[2] + b4.z, v[3] + b4.w};
      if (EPI == EPI_F32) {
        *(float4*)((float*)p.out + (long long)m * p.ldo + n) = make_float4(r[0], r[1], r[2], r[3]);
      } else if (EPI == EPI_F16) {
        *(uint2*)((f16*)p.out + (long long)m * p.ldo + n) =
            make_uint2(to_bits<f16>(r[0]) | ((unsigned)to_bits<f16>(r[1]) << 16), to_bits<f16>(r[2]) | ((unsigned)to_bits<f16>(r[3]) << 16));
      } else if (EPI == EPI_RESID) {                                // x += gamma * (W h + b): LayerScale residual, fp32 stream
        const float4 g4 = gam4[t];
        float4 cur = extra[t][u];
        cur.x += g4.x * r[0]; cur.y += g4.y * r[1]; cur.z += g4.z * r[2]; cur.w += g4.w * r[3];
        *(float4*)((float*)p.out + (long long)m * p.ldo + n) = cur;
      } else if (EPI == EPI_TOKENS) {                               // token = projection + bias + position embedding, behind the registers
        const float4 pe = extra[t][u];
        *(float4*)((float*)p.out + row_off * p.ldo + n) = make_float4(r[0] + pe.x, r[1] + pe.y, r[2] + pe.z, r[3] + pe.w);
      }
    }
  }
}

template <int MT, int NT, int EPI, bool SPLIT, int PF>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int ncb = (p.ntiles + NT - 1) / NT;                      // column blocks of one row block: ids 8 apart = one XCD, one L2 (see wsgemm_kernel)
  const int xcd = blockIdx.x & 7, cb = (blockIdx.x >> 3) % ncb, rb = (blockIdx.x / (8 * ncb)) * 8 + xcd;
  const int m0 = (rb * 4 + wave) * MT * 16, nt0 = cb * NT;
  if (m0 >= p.M) return;
  const int KS = p.KS;

  const char* wp[NT];
  const char* ap[MT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int nt = nt0 + t < p.ntiles ? nt0 + t : p.ntiles - 1;
    wp[t] = p.w_hi + ((long long)nt * KS * 64 + lane) * 16;
  }
#pragma unroll
  for (int u = 0; u < MT; ++u) {
    int m = m0 + u * 16 + li;
    m = m < p.M ? m : p.M - 1;
    ap[u] = p.a_hi + (long long)m * p.lda * 2 + g * 16;
  }
  const long long wlo = SPLIT ? p.w_lo - p.w_hi : 0, alo = SPLIT ? p.a_lo - p.a_hi : 0;

  f32x4 acc[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < MT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int NB = PF + 1;                                      // operand buffers: PF K steps in flight beside the one being multiplied
  f16x8 wh[NB][NT], ah[NB][MT], wl[NB][SPLIT ? NT : 1], al[NB][SPLIT ? MT : 1];
  auto load = [&](int ks, const int buf) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      wh[buf][t] = *(const f16x8*)(wp[t] + (long long)ks * 1024);
      if (SPLIT) wl[buf][t] = *(const f16x8*)(wp[t] + wlo + (long long)ks * 1024);
    }
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      ah[buf][u] = *(const f16x8*)(ap[u] + ks * 64);
      if (SPLIT) al[buf][u] = *(const f16x8*)(ap[u] + alo + ks * 64);
    }
  };
  auto compute = [&](const int buf) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        if (EPI == EPI_PLANAR || EPI == EPI_VT) {     // token rows as the A operand: a lane then owns 4 consecutive VOXELS / TOKENS of one feature
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[buf][u], wh[buf][t], acc[t][u], 0, 0, 0);
          if (SPLIT) {
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[buf][u], wh[buf][t], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[buf][u], wl[buf][t], acc[t][u], 0, 0, 0);
          }
          continue;
        }
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[buf][t], ah[buf][u], acc[t][u], 0, 0, 0);
        if (SPLIT) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[buf][t], al[buf][u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[buf][t], ah[buf][u], acc[t][u], 0, 0, 0);
        }
      }
  };
  // Prefetches are UNCONDITIONAL (the K index is clamped, the tail re-fetches the last step): a conditional load block makes the
  // waitcnt pass assume the shorter queue at the join and drain it with vmcnt(0) before every buffer -- no overlap left.
#pragma unroll
  for (int i = 0; i < PF; ++i) load(i < KS ? i : KS - 1, i);
  int ks = 0;
  for (; ks + NB <= KS; ks += NB) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int nx = ks + j + PF;
      load(nx < KS ? nx : KS - 1, (j + PF) % NB);
      __builtin_amdgcn_sched_barrier(0);      // and the scheduler must not sink the loads below the MFMAs either
      compute(j);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int j = 0; j < NB - 1; ++j) {          // tail: < NB steps, their operands are already in flight
    if (ks + j < KS) compute(j);
  }

  gemm_epilogue<MT, NT, EPI>(p, acc, m0, nt0, li, g);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-stationary variant.  The direct kernel above streams BOTH operands from L2 for every tile: on the blocks' shapes
// (16416 x 396..1056 by 396..2112) it is bound by L2 / texture-path bandwidth, not by the matrix pipe (profiles/r03_vit_*).
// Here a workgroup owns NT feature tiles with the WHOLE K extent resident in LDS (fragment order, conflict-free ds_read_b128:
// 65 KiB for 80 features x 416), and its four waves stream row groups past it: per K step a wave loads MT row fragments from
// global memory and reads NT weight fragments from LDS for MT * NT MFMAs.  The row fragments of the next unit (row group x
// K chunk of KC steps) are requested before the current unit is multiplied, so a wave always has KC * MT (x2 when split)
// 1-KiB loads in flight; accumulators stay in registers across the chunks of a row group.
static int gemm_env(const char* name, int dflt) {
  const char* e = exp_env(name);
  return e ? atoi(e) : dflt;
}

template <int V> using IC = std::integral_constant<int, V>;

template <int MT, int NT, int KC, int EPI, bool SPLIT, int NW = 4>
__global__ __launch_bounds__(NW * 64) void wsgemm_kernel(GemmParams p, int rows_per_wg, int xcd_order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];          // [hi | lo][NT][KS][64 lanes][16 B]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int ncb = (p.ntiles + NT - 1) / NT;
  // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so the ncb column blocks that
  // share a row block are given ids 8 apart -- they land on ONE XCD and the rows are fetched into one L2 instead of up to 8
  // (the last transposed conv re-read its 0.5 GB operand through 4 L2s: 633 -> ? us, profiles/r03_vit_*).
  // With many column blocks (SwiGLU: 33) the per-XCD workgroup count overshoots the XCD's slots and a second round starts: plain order there.
  const int xcd = blockIdx.x & 7;
  const int cb = xcd_order ? (blockIdx.x >> 3) % ncb : blockIdx.x % ncb;
  const int rb = xcd_order ? (blockIdx.x / (8 * ncb)) * 8 + xcd : blockIdx.x / ncb;
  if (rb * rows_per_wg >= p.M) return;
  const int nt0 = cb * NT, KS = p.KS;
  constexpr int nw = NW;                                                // waves sharing the slice (8 where LDS leaves room for one workgroup per CU only)
  const int plane = NT * KS * 64;                                       // 16-byte units per plane
  // Fill: LDS-DMA, 1 KiB (one fragment) per instruction, every wave queues its whole share before the single wait.  (A load ->
  // ds_write loop is serialised on the memory latency by its data dependence: measured 35-50 us of a 60 us launch.)
  {
    const int nfrag = NT * KS * (SPLIT ? 2 : 1);
    for (int j = wave; j < nfrag; j += nw) {
      const int pl = j / (NT * KS), r = j - pl * NT * KS, t = r / KS, ks = r - t * KS;
      const int nt = nt0 + t < p.ntiles ? nt0 + t : p.ntiles - 1;
      const char* src = (pl ? p.w_lo : p.w_hi) + (((long long)nt * KS + ks) * 64 + lane) * 16;
      dma16_asm(src, lds_addr(smem) + (unsigned)j * 1024u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const int row0 = rb * rows_per_wg;
  const int row1 = row0 + rows_per_wg < p.M ? row0 + rows_per_wg : p.M;
  const int groups = (row1 - row0 + MT * 16 - 1) / (MT * 16), nchunk = (KS + KC - 1) / KC;
  const long long alo = SPLIT ? p.a_lo - p.a_hi : 0;

  f16x8 ah[2][MT][KC], al[2][SPLIT ? MT : 1][SPLIT ? KC : 1];
  f32x4 acc[NT][MT];
  auto issue = [&](auto BUF, int grp, int c) {
    constexpr int B = decltype(BUF)::value;
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      int m = row0 + (grp * MT + u) * 16 + li;
      m = m < p.M ? m : p.M - 1;
      const char* ap = p.a_hi + (long long)m * p.lda * 2 + g * 16;
#pragma unroll
      for (int i = 0; i < KC; ++i) {          // unconditional (clamped) loads: see the direct kernel
        const int ks = c * KC + i < KS ? c * KC + i : KS - 1;
        ah[B][u][i] = *(const f16x8*)(ap + ks * 64);
        if (SPLIT) al[B][u][i] = *(const f16x8*)(ap + alo + ks * 64);
      }
    }
  };
  auto step = [&](auto BUF, int grp, int c) {
    constexpr int B = decltype(BUF)::value;
    if (c == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < MT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < KC; ++i) {
      const int ks = c * KC + i;
      if (ks >= KS) break;
      f16x8 wh[NT], wl[SPLIT ? NT : 1];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        wh[t] = *(const f16x8*)(smem + ((long long)(t * KS + ks) * 64 + lane) * 16);
        if (SPLIT) wl[t] = *(const f16x8*)(smem + ((long long)plane + (t * KS + ks) * 64 + lane) * 16);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < MT; ++u) {
          if (EPI == EPI_PLANAR || EPI == EPI_VT) {
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[B][u][i], wh[t], acc[t][u], 0, 0, 0);
            if (SPLIT) {
              acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[B][u][i], wh[t], acc[t][u], 0, 0, 0);
              acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[B][u][i], wl[t], acc[t][u], 0, 0, 0);
            }
          } else {
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], ah[B][u][i], acc[t][u], 0, 0, 0);
            if (SPLIT) {
              acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], al[B][u][i], acc[t][u], 0, 0, 0);
              acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], ah[B][u][i], acc[t][u], 0, 0, 0);
            }
          }
        }
    }
    if (c == nchunk - 1) gemm_epilogue<MT, NT, EPI>(p, acc, row0 + grp * MT * 16, nt0, li, g);
  };
  auto next = [&](int& gq, int& cq) {
    if (++cq == nchunk) { cq = 0; gq += nw; }
  };
  int g0 = wave, c0 = 0, g1 = wave, c1 = 0;
  next(g1, c1);
  if (g0 >= groups) return;
  issue(IC<0>{}, g0, c0);
  while (true) {                               // the unit after the last one re-fetches rows of the last group (clamped), nothing uses it
    issue(IC<1>{}, g1 < groups ? g1 : g0, g1 < groups ? c1 : c0);
    __builtin_amdgcn_sched_barrier(0);
    step(IC<0>{}, g0, c0);
    __builtin_amdgcn_sched_barrier(0);
    g0 = g1; c0 = c1; next(g1, c1);
    if (g0 >= groups) break;
    issue(IC<0>{}, g1 < groups ? g1 : g0, g1 < groups ? c1 : c0);
    __builtin_amdgcn_sched_barrier(0);
    step(IC<1>{}, g0, c0);
    __builtin_amdgcn_sched_barrier(0);
    g0 = g1; c0 = c1; next(g1, c1);
    if (g0 >= groups) break;
  }
}

template <int MT, int NT, int KC, int EPI, bool SPLIT, int NW = 4>
static hipError_t launch_ws(const GemmParams& p, hipStream_t st) {
  const int ncb = (p.ntiles + NT - 1) / NT;
  const size_t lds = (size_t)NT * p.KS * 1024 * (SPLIT ? 2 : 1);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static size_t attr = 0;                                          // per instantiation
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute((const void*)wsgemm_kernel<MT, NT, KC, EPI, SPLIT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr = lds;
  }
  // Row blocks: whole groups of 4 waves x MT tiles, sized so that ALL workgroups are resident at once (LDS decides how many fit
  // on a CU; a few workgroups more than slots cost a whole second round), in XCD order a multiple of 8 row blocks.
  const int per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3);
  constexpr int nw = NW;
  const int unit = MT * 16 * nw, G = (p.M + unit - 1) / unit;
  const int xcd_order = ncb <= 16;
  int nrb = 256 * per_cu / ncb;
  if (xcd_order) nrb = nrb / 8 * 8;
  nrb = nrb < 1 ? 1 : (nrb > G ? G : nrb);
  int rows_per_wg = (G + nrb - 1) / nrb * unit;
  // The planar-output stage moves exactly its algorithmic bytes (0.54 GB in, 1.07 GB out: profiles/r03_vit_pmc_traffic.json) yet
  // runs at 2.7 TB/s.  Row blocks of 128 / 256 / 1024 / 5504 rows: 696 / 604 / 548 / 602 us -- 1024 it is (AMX_PLANAR_ROWS).
  static const int planar_rows = gemm_env("AMX_PLANAR_ROWS", 1024);
  if (EPI == EPI_PLANAR && planar_rows > 0 && rows_per_wg > planar_rows) rows_per_wg = (planar_rows + unit - 1) / unit * unit;
  nrb = (p.M + rows_per_wg - 1) / rows_per_wg;
  if (xcd_order) nrb = (nrb + 7) / 8 * 8;                          // whole groups of 8 row blocks (XCD-aware order in the kernel)
  hipLaunchKernelGGL((wsgemm_kernel<MT, NT, KC, EPI, SPLIT, NW>), dim3((unsigned)ncb * nrb), dim3(64 * nw), lds, st, p, rows_per_wg, xcd_order);
  return hipGetLastError();
}

template <int MT, int NT, int EPI, bool SPLIT, int PF>
static hipError_t launch_one(const GemmParams& p, hipStream_t st) {
  const int ncb = (p.ntiles + NT - 1) / NT, nrb = ((p.M + MT * 64 - 1) / (MT * 64) + 7) / 8 * 8;
  hipLaunchKernelGGL((gemm_kernel<MT, NT, EPI, SPLIT, PF>), dim3((unsigned)ncb * nrb), dim3(256), 0, st, p);
  return hipGetLastError();
}

template <int EPI, bool SPLIT>
static hipError_t launch_epi(const GemmParams& p, hipStream_t st);

// weight-stationary dispatch; returns hipErrorNotSupported when the slice does not fit LDS (the caller falls back to the direct kernel)
template <int EPI, bool SPLIT>
static hipError_t launch_ws_epi(const GemmParams& p, hipStream_t st) {
  const bool five = EPI != EPI_SWIGLU && p.ntiles % 5 == 0;
  const size_t per_tile = (size_t)p.KS * 1024 * (SPLIT ? 2 : 1);
  if (SPLIT) {
    if (five && 5 * per_tile <= 160 * 1024) return launch_ws<2, 5, 2, EPI, true>(p, st);
    if (!five && 4 * per_tile <= 160 * 1024) {
      // The store-bound planar stage: ONE row tile per wave (112 registers: 4 waves per SIMD).  Loads and stores share a wave's
      // in-order memory counter, so a wave's next operands wait behind the acknowledgement of its previous stores; more waves =
      // more independent queues (2 tiles per wave 547 us, 1 tile 476 us; deeper prefetch 550 -- no help).
      if (EPI == EPI_PLANAR) return launch_ws<1, 4, 2, EPI, true>(p, st);
      if (EPI == EPI_SCATTER && 4 * per_tile > 80 * 1024) return launch_ws<1, 4, 2, EPI, true, 8>(p, st);   // one workgroup per CU: 8 waves
      return launch_ws<2, 4, 2, EPI, true>(p, st);
    }
    return hipErrorNotSupported;
  }
  // measured in the ViT forward (tools/vit_gemm_sweep.sh): 4 row tiles per wave win for the wide fp32-output product (q | k | v),
  // 2 row tiles (3 waves per SIMD) for the residual / SwiGLU epilogues
  static const int wsmt_env = gemm_env("AMX_GEMM_WSMT", 0);
  const int wsmt = wsmt_env ? wsmt_env : (EPI == EPI_F32 ? 4 : 2);
  if (wsmt == 2) {
    if (five && 5 * per_tile <= 160 * 1024) return launch_ws<2, 5, 4, EPI, false>(p, st);
    if (!five && 4 * per_tile <= 160 * 1024) return launch_ws<2, 4, 4, EPI, false>(p, st);
    return hipErrorNotSupported;
  }
  if (five && 5 * per_tile <= 160 * 1024) return launch_ws<4, 5, 3, EPI, false>(p, st);
  if (!five && 4 * per_tile <= 160 * 1024) return launch_ws<4, 4, 4, EPI, false>(p, st);
  return hipErrorNotSupported;
}

template <int EPI, bool SPLIT>
static hipError_t launch_epi(const GemmParams& p, hipStream_t st) {
  static const int use_ws = gemm_env("AMX_GEMM_WS", 1);
  if (use_ws) {
    const hipError_t e = launch_ws_epi<EPI, SPLIT>(p, st);
    if (e != hipErrorNotSupported) return e;
  }
  // NT = 5 where the tile count is a multiple of 5 (396 -> 25 tiles, 3 x 396 -> 75): no idle column tiles.  SwiGLU pairs need an even NT.
  const bool five = EPI != EPI_SWIGLU && p.ntiles % 5 == 0;
  static const int force_mt = gemm_env("AMX_GEMM_MT", 0), pf = gemm_env("AMX_GEMM_PF", 2);
  bool small = (long long)((p.M + 255) / 256) * ((p.ntiles + (five ? 4 : 3)) / (five ? 5 : 4)) < 512;   // < 2 workgroups per CU: halve the row block
  if (force_mt) small = force_mt == 2;
  if (SPLIT) return five ? launch_one<2, 5, EPI, true, 1>(p, st) : launch_one<2, 4, EPI, true, 1>(p, st);
  if (pf == 1) {
    if (five) return small ? launch_one<2, 5, EPI, false, 1>(p, st) : launch_one<4, 5, EPI, false, 1>(p, st);
    return small ? launch_one<2, 4, EPI, false, 1>(p, st) : launch_one<4, 4, EPI, false, 1>(p, st);
  }
  if (five) return small ? launch_one<2, 5, EPI, false, 2>(p, st) : launch_one<4, 5, EPI, false, 2>(p, st);
  return small ? launch_one<2, 4, EPI, false, 2>(p, st) : launch_one<4, 4, EPI, false, 2>(p, st);
}

hipError_t launch_gemm(const GemmParams& p, int epi, hipStream_t st) {
  const bool split = p.a_lo != nullptr;
  if (split != (p.w_lo != nullptr)) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_F32: return split ? launch_epi<EPI_F32, true>(p, st) : launch_epi<EPI_F32, false>(p, st);
    case EPI_F16: return split ? hipErrorInvalidValue : launch_epi<EPI_F16, false>(p, st);
    case EPI_RESID: return split ? hipErrorInvalidValue : launch_epi<EPI_RESID, false>(p, st);
    case EPI_SWIGLU: return split ? hipErrorInvalidValue : launch_epi<EPI_SWIGLU, false>(p, st);
    case EPI_SCATTER: return split ? launch_epi<EPI_SCATTER, true>(p, st) : launch_epi<EPI_SCATTER, false>(p, st);
    case EPI_TOKENS: return split ? launch_epi<EPI_TOKENS, true>(p, st) : launch_epi<EPI_TOKENS, false>(p, st);
    case EPI_PLANAR:
      if (p.gw % 4) return hipErrorInvalidValue;
      return split ? launch_epi<EPI_PLANAR, true>(p, st) : launch_epi<EPI_PLANAR, false>(p, st);
    case EPI_QK:      // a head's 5 tiles with the whole K extent in LDS (embed_dim <= 1024); wider models stream the weights (direct kernel)
      if (split || p.Cp != 80 || p.ntiles != 2 * p.heads * 5) return hipErrorInvalidValue;
      // (4 row tiles per wave measured the same 64 us as 2: of those, ~10 us are the LDS fill, ~19 the K loop, ~34 the epilogue)
      // eight waves of ONE row tile (the epilogue is long: more waves hide it; 64 -> 56 us; the residual / SwiGLU products measured no gain)
      return (size_t)5 * p.KS * 1024 <= 160 * 1024 ? launch_ws<1, 5, 4, EPI_QK, false, 8>(p, st) : launch_one<2, 5, EPI_QK, false, 2>(p, st);
    case EPI_VT:
      if (split || p.Cp != 80 || p.ntiles != p.heads * 5) return hipErrorInvalidValue;
      return (size_t)5 * p.KS * 1024 <= 160 * 1024 ? launch_ws<1, 5, 4, EPI_VT, false, 8>(p, st) : launch_one<2, 5, EPI_VT, false, 2>(p, st);
    case EPI_SCATTER_LN:
      if (p.Cp != 128 || p.ntiles != 64 || p.ldo > 128) return hipErrorInvalidValue;
      if ((size_t)8 * p.KS * 1024 * (split ? 2 : 1) <= 160 * 1024)
        // the 112 KiB slice leaves one workgroup per CU: 8 waves of one row tile each (442 -> 369 us against 4 waves of two)
        return split ? launch_ws<1, 8, 2, EPI_SCATTER_LN, true, 8>(p, st) : launch_ws<2, 8, 4, EPI_SCATTER_LN, false>(p, st);
      return split ? launch_one<2, 8, EPI_SCATTER_LN, true, 1>(p, st) : launch_one<2, 8, EPI_SCATTER_LN, false, 1>(p, st);
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight packing: fp32 parameters -> f16 (hi [+ lo]) fragments [n tile][k step][lane][8].
//   mode 0  rows of up to three [rows_i][K] matrices stacked (q | k | v; a single Linear; a 1x1x1 conv weight [N][K])
//   mode 1  SwiGLU: tile 2 s = rows 16 s .. of src0 (gate), tile 2 s + 1 = the same rows of src1 (value)
//   mode 2  ConvTranspose3d weight [K = Cin][Cout][2][2][2]: n = parity * Cp + c
//   mode 3  the same weight with n = ((dz, dy) * Cp + c) * 2 + dx (EPI_PLANAR)
//   mode 4  q | k (or v alone): every head's Creal rows padded to Cp (EPI_QK / EPI_VT)
__global__ void pack_gemm_kernel(const float* s0, const float* s1, const float* s2, int r0, int r1, int r2, int K, int mode, int Cp,
                                 int Creal, int ntiles, int KS, f16* hi, f16* lo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one (tile, step, lane) = 8 halves
  if (idx >= (long long)ntiles * KS * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) % KS, nt = (idx >> 6) / KS;
  const int i = lane & 15, g = lane >> 4;
  unsigned short h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = ks * 32 + g * 8 + e;
    float w = 0.f;
    if (k < K) {
      if (mode == 0) {
        const int n = nt * 16 + i;
        if (n < r0) w = s0[(long long)n * K + k];
        else if (n < r0 + r1) w = s1[(long long)(n - r0) * K + k];
        else if (n < r0 + r1 + r2) w = s2[(long long)(n - r0 - r1) * K + k];
      } else if (mode == 1) {
        const int n = (nt >> 1) * 16 + i;
        if (n < r0) w = ((nt & 1) ? s1 : s0)[(long long)n * K + k];
      } else if (mode == 2) {
        const int n = nt * 16 + i, par = n / Cp, c = n % Cp;
        if (par < 8 && c < Creal) w = s0[((long long)k * Creal + c) * 8 + par];
      } else if (mode == 4) {         // head-padded rows of up to two [heads * Creal][K] matrices: n = (matrix, head, d < Cp)
        const int n = nt * 16 + i, span = r0 / Creal * Cp, src = n / span, within = n % span, head = within / Cp, d = within % Cp;
        const float* S = src == 0 ? s0 : (src == 1 ? s1 : nullptr);
        if (S && d < Creal) w = S[((long long)head * Creal + d) * K + k];
      } else {                        // mode 3: n = ((dz dy) * Cp + c) * 2 + dx
        const int n = nt * 16 + i, q = n >> 1, c = q % Cp, zy = q / Cp;
        if (zy < 4 && c < Creal) w = s0[((long long)k * Creal + c) * 8 + zy * 2 + (n & 1)];
      }
    }
    const f16 wh = (f16)w;
    h[e] = __builtin_bit_cast(unsigned short, wh);
    l[e] = to_bits<f16>(w - (float)wh);
  }
  ((uint4*)hi)[idx] = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
  if (lo) ((uint4*)lo)[idx] = make_uint4(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16), l[4] | ((unsigned)l[5] << 16), l[6] | ((unsigned)l[7] << 16));
}

hipError_t launch_pack_gemm(const float* s0, const float* s1, const float* s2, int r0, int r1, int r2, int K, int mode, int Cp, int Creal,
                            int ntiles, int KS, void* hi, void* lo, hipStream_t st) {
  const long long n = (long long)ntiles * KS * 64;
  hipLaunchKernelGGL(pack_gemm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s0, s1, s2, r0, r1, r2, K, mode, Cp, Creal, ntiles,
                     KS, (f16*)hi, (f16*)lo);
  return hipGetLastError();
}

// dst[off .. off + n) = src (or `fill` when src is null): assembles padded bias / gain vectors on the device
__global__ void vec_place_kernel(float* dst, const float* src, int n, float fill) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src ? src[i] : fill;
}
hipError_t launch_vec_place(float* dst, const float* src, int n, float fill, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(vec_place_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dst, src, n, fill);
  return hipGetLastError();
}

// rotary table [tokens][2 hd] = [sin | cos] -> [tokens][hd / 2][{sin 2i, sin 2i+1, cos 2i, cos 2i+1}] (one 16-byte load per pair)
__global__ void rope_pairs_kernel(float* dst, const float* src, int tokens, int hd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tokens * hd / 2) return;
  const int tk = i / (hd / 2), pr = i % (hd / 2);
  const float* s = src + (long long)tk * 2 * hd;
  ((float4*)dst)[i] = make_float4(s[2 * pr], s[2 * pr + 1], s[hd + 2 * pr], s[hd + 2 * pr + 1]);
}
hipError_t launch_rope_pairs(float* dst, const float* src, int tokens, int hd, hipStream_t st) {
  hipLaunchKernelGGL(rope_pairs_kernel, dim3((tokens * hd / 2 + 255) / 256), dim3(256), 0, st, dst, src, tokens, hd);
  return hipGetLastError();
}

__global__ void headpad_vec_kernel(float* dst, const float* src, int heads, int hd, int hd_pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= heads * hd_pad) return;
  const int d = i % hd_pad;
  dst[i] = (src && d < hd) ? src[(i / hd_pad) * hd + d] : 0.f;
}
hipError_t launch_headpad_vec(float* dst, const float* src, int heads, int hd, int hd_pad, hipStream_t st) {
  hipLaunchKernelGGL(headpad_vec_kernel, dim3((heads * hd_pad + 255) / 256), dim3(256), 0, st, dst, src, heads, hd, hd_pad);
  return hipGetLastError();
}

// SwiGLU bias in packed feature order: tile 2 s = gate bias of hidden 16 s .., tile 2 s + 1 = value bias of the same columns
__global__ void swiglu_bias_kernel(float* dst, const float* bg, const float* bx, int hidden) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * ((hidden + 15) / 16 * 16)) return;                  // hidden rounded up to whole tiles (zeros behind it)
  const int tile = i >> 4, h = (tile >> 1) * 16 + (i & 15);
  dst[i] = h < hidden ? ((tile & 1) ? bx : bg)[h] : 0.f;
}
hipError_t launch_swiglu_bias(float* dst, const float* bg, const float* bx, int hidden, hipStream_t st) {
  hipLaunchKernelGGL(swiglu_bias_kernel, dim3((2 * ((hidden + 15) / 16 * 16) + 255) / 256), dim3(256), 0, st, dst, bg, bx, hidden);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Row LayerNorm -> f16 operand rows.  One wave per row (C <= 64 * PER): mean, then the variance of the deviations (two
// register passes, fp32), y = (x - mean) * rstd * w + b, optional exact GELU, ONE rounding to f16 (hi) [+ the remainder
// (lo)], zero fill up to ldo.  Row r of the output reads input row (r / rows_out) * rows_in + skip + r % rows_out: the
// final norm drops the register tokens that way.
template <typename TIN, int PER, bool GELU>
__global__ __launch_bounds__(256) void ln_rows_kernel(const TIN* __restrict__ in, long long ldi, int C, const float* __restrict__ w,
                                                      const float* __restrict__ b, float eps, int M, int rows_out, int rows_in, int skip,
                                                      f16* __restrict__ hi, f16* __restrict__ lo, int ldo) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const long long ri = (long long)(r / rows_out) * rows_in + skip + r % rows_out;
  const TIN* x = in + ri * ldi;
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < C ? (float)x[c] : 0.f;
    s += v[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float d = i * 64 + lane < C ? v[i] - mean : 0.f;
    v[i] = d;
    q += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / C + eps);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = i * 64 + lane;
    if (c >= ldo) continue;
    float y = 0.f;
    if (c < C) {
      y = w ? v[i] * rstd * w[c] + b[c] : v[i] + mean;              // w == null: plain conversion (no norm)
      if (GELU) y = gelu_exact(y);
    }
    const f16 yh = (f16)y;
    hi[(long long)r * ldo + c] = yh;
    if (lo) lo[(long long)r * ldo + c] = (f16)(y - (float)yh);
  }
}

// The same with 4 elements per lane and chunk (16-byte fp32 / 8-byte f16 loads, 8-byte stores): C, ldi and ldo multiples of 4.
template <typename TIN, int PERV, bool GELU>
__global__ __launch_bounds__(256) void ln_rows_vec_kernel(const TIN* __restrict__ in, long long ldi, int C, const float* __restrict__ w,
                                                          const float* __restrict__ b, float eps, int M, int rows_out, int rows_in, int skip,
                                                          f16* __restrict__ hi, f16* __restrict__ lo, int ldo) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const long long ri = (long long)(r / rows_out) * rows_in + skip + r % rows_out;
  const TIN* x = in + ri * ldi;
  float v[PERV][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PERV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      if (sizeof(TIN) == 4) {
        const float4 t = *(const float4*)((const float*)x + c);
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
      } else {
        typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
        const f16x4 t = *(const f16x4*)((const f16*)x + c);
        v[i][0] = (float)t[0]; v[i][1] = (float)t[1]; v[i][2] = (float)t[2]; v[i][3] = (float)t[3];
      }
    } else {
      v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
    }
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PERV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = (i * 64 + lane) * 4 < C ? v[i][e] - mean : 0.f;
      v[i][e] = d;
      q += d * d;
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / C + eps);
#pragma unroll
  for (int i = 0; i < PERV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c >= ldo) continue;
    unsigned short hb[4] = {0, 0, 0, 0}, lb[4] = {0, 0, 0, 0};
    if (c < C) {
      float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (w) { w4 = *(const float4*)(w + c); b4 = *(const float4*)(b + c); }
      const float ww[4] = {w4.x, w4.y, w4.z, w4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = w ? v[i][e] * rstd * ww[e] + bb[e] : v[i][e] + mean;
        if (GELU) y = gelu_exact(y);
        const f16 yh = (f16)y;
        hb[e] = __builtin_bit_cast(unsigned short, yh);
        lb[e] = to_bits<f16>(y - (float)yh);
      }
    }
    *(uint2*)(hi + (long long)r * ldo + c) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
    if (lo) *(uint2*)(lo + (long long)r * ldo + c) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
  }
}

hipError_t launch_ln_rows(const void* in, int in_f16, long long ldi, int C, const float* w, const float* b, float eps, int M, int rows_out,
                          int rows_in, int skip, int gelu, void* hi, void* lo, int ldo, hipStream_t st) {
  if (ldo < C) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
  if (C % 4 == 0 && ldi % 4 == 0 && ldo % 4 == 0 && (ldo + 255) / 256 <= 12 && !(in_f16 && gelu)) {
#define AMX_LNV(TIN, PERV, G) \
  hipLaunchKernelGGL((ln_rows_vec_kernel<TIN, PERV, G>), grid, block, 0, st, (const TIN*)in, ldi, C, w, b, eps, M, rows_out, rows_in, skip, (f16*)hi, (f16*)lo, ldo)
    const int perv = (ldo + 255) / 256;
    if (in_f16) { if (perv <= 2) AMX_LNV(f16, 2, false); else if (perv <= 5) AMX_LNV(f16, 5, false); else AMX_LNV(f16, 12, false); }
    else if (gelu) { if (perv <= 1) AMX_LNV(float, 1, true); else if (perv <= 2) AMX_LNV(float, 2, true); else AMX_LNV(float, 5, true); }
    else { if (perv <= 2) AMX_LNV(float, 2, false); else if (perv <= 5) AMX_LNV(float, 5, false); else AMX_LNV(float, 12, false); }
#undef AMX_LNV
    return hipGetLastError();
  }
  if (C > 64 * 17 || ldo > 64 * 17) return hipErrorInvalidValue;      // scalar fallback: rows of up to 1088 columns
#define AMX_LN(TIN, PER, G) \
  hipLaunchKernelGGL((ln_rows_kernel<TIN, PER, G>), grid, block, 0, st, (const TIN*)in, ldi, C, w, b, eps, M, rows_out, rows_in, skip, (f16*)hi, (f16*)lo, ldo)
  const int per = (ldo + 63) / 64;
  if (in_f16) {
    if (gelu) return hipErrorInvalidValue;
    if (per <= 7) AMX_LN(f16, 7, false); else AMX_LN(f16, 17, false);
  } else if (gelu) {
    if (per <= 4) AMX_LN(float, 4, true); else AMX_LN(float, 7, true);
  } else {
    if (per <= 7) AMX_LN(float, 7, false); else AMX_LN(float, 17, false);
  }
#undef AMX_LN
  return hipGetLastError();
}

// register tokens [nreg][E] in front of every sample's patch tokens (architectures.py:117-120: prepended, dropped before decoding)
__global__ void place_registers_kernel(const float* __restrict__ reg, int nreg, int E, int rows_per_b, float* __restrict__ tok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i < nreg * E) tok[(long long)b * rows_per_b * E + i] = reg[i];
}
hipError_t launch_place_registers(const float* reg, int nreg, int E, int rows_per_b, int nb, float* tok, hipStream_t st) {
  if (nreg <= 0) return hipSuccess;
  hipLaunchKernelGGL(place_registers_kernel, dim3((nreg * E + 255) / 256, nb), dim3(256), 0, st, reg, nreg, E, rows_per_b, tok);
  return hipGetLastError();
}

// Column sums of an f16 (hi [+ lo]) operand matrix per sample: out[b][chunk][c] = sum over the chunk's rows (fp32; the
// caller adds the chunks).  Feeds ChannelDemean (architectures.py:28-33) through the linearity of the last transposed conv.
// One thread = 8 columns (16-byte loads), ld / 8 threads per row, 256 / (ld / 8) rows per pass; LDS reduction over the rows.
__global__ __launch_bounds__(256) void colsum_kernel(const char* __restrict__ hi, const char* __restrict__ lo, int ld, int rows_per_b,
                                                     int rows_per_chunk, float* __restrict__ out) {
  __shared__ float red[256][9];
  const int b = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
  const int tpr = ld / 8, rpp = 256 / tpr;                         // threads per row, rows per pass
  const int cg = threadIdx.x % tpr, rr = threadIdx.x / tpr;
  const long long r0 = (long long)b * rows_per_b + (long long)chunk * rows_per_chunk;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rr < rpp)
    for (int r = rr; r < rows_per_chunk; r += rpp) {
      const long long o = ((r0 + r) * ld + cg * 8) * 2;
      const f16x8 h = *(const f16x8*)(hi + o);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)h[e];
      if (lo) {
        const f16x8 l = *(const f16x8*)(lo + o);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)l[e];
      }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  if (threadIdx.x < tpr) {                                          // thread cg adds the rpp partial rows of its column group
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < rpp; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] += red[q * tpr + cg][e];
#pragma unroll
    for (int e = 0; e < 8; ++e) out[((long long)b * nch + chunk) * ld + cg * 8 + e] = t[e];
  }
}
hipError_t launch_colsum(const void* hi, const void* lo, int ld, int C, int nb, int rows_per_b, int nchunk, float* out, hipStream_t st) {
  if (ld > 256 || ld % 8 || C > ld || rows_per_b % nchunk) return hipErrorInvalidValue;
  hipLaunchKernelGGL(colsum_kernel, dim3(nchunk, nb), dim3(256), 0, st, (const char*)hi, (const char*)lo, ld, rows_per_b, rows_per_b / nchunk, out);
  return hipGetLastError();
}

// mean of output channel c of sample b after the last transposed conv: (1/8) sum_par sum_k W[k][c][par] xbar[b][k] + bias[c],
// xbar = column sums / rows.  One workgroup per sample: the chunk sums are combined into LDS, then one thread per channel.
__global__ __launch_bounds__(256) void demean_kernel(const float* __restrict__ colsum, int nchunk, int ld, int K, double inv_rows,
                                                     const float* __restrict__ W, int Creal, const float* __restrict__ bias, float* __restrict__ mean) {
  __shared__ double xbar[256], part[256];
  const int b = blockIdx.x;
  {                                             // 256 / K threads per input channel share the chunks (independent loads, short chains)
    const int tpk = 256 / K, k = threadIdx.x % K, sub = threadIdx.x / K;
    double xs = 0.0;
    if (sub < tpk)
      for (int ch = sub; ch < nchunk; ch += tpk) xs += colsum[((long long)b * nchunk + ch) * ld + k];
    part[threadIdx.x] = sub < tpk ? xs : 0.0;
    __syncthreads();
    if ((int)threadIdx.x < K) {
      double t = 0.0;
      for (int q = 0; q < tpk; ++q) t += part[q * K + threadIdx.x];
      xbar[threadIdx.x] = t * inv_rows;
    }
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= Creal) return;
  double acc = 0.0;
  for (int k = 0; k < K; ++k) {
    const float4 w0 = *(const float4*)(W + ((long long)k * Creal + c) * 8), w1 = *(const float4*)(W + ((long long)k * Creal + c) * 8 + 4);
    acc += xbar[k] * (double)(w0.x + w0.y + w0.z + w0.w + w1.x + w1.y + w1.z + w1.w) * 0.125;
  }
  mean[b * Creal + c] = (float)(acc + (bias ? bias[c] : 0.f));
}
hipError_t launch_demean(const float* colsum, int nchunk, int ld, int K, long long rows_per_b, const float* W, int Creal, const float* bias, int nb,
                         float* mean, hipStream_t st) {
  if (Creal > 256 || K > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(demean_kernel, dim3(nb), dim3(256), 0, st, colsum, nchunk, ld, K, 1.0 / (double)rows_per_b, W, Creal, bias, mean);
  return hipGetLastError();
}

// channels-last fp32 [b][vox][C] -> planar fp32 [b][C][vox], minus a per (b, c) constant: 64 voxels x C through LDS
__global__ __launch_bounds__(256) void export_planar_kernel(const float* __restrict__ in, int C, long long vox, const float* __restrict__ sub,
                                                            float* __restrict__ out) {
  extern __shared__ float tile[];                                   // [64][C + 1]
  const int b = blockIdx.y;
  const long long v0 = (long long)blockIdx.x * 64;
  const float* src = in + ((long long)b * vox + v0) * C;
  for (int i = threadIdx.x; i < 64 * C; i += 256) tile[(i / C) * (C + 1) + i % C] = src[i];
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * C; i += 256) {
    const int c = i >> 6, v = i & 63;
    out[((long long)b * C + c) * vox + v0 + v] = tile[v * (C + 1) + c] - (sub ? sub[b * C + c] : 0.f);
  }
}
hipError_t launch_export_planar(const float* in, int C, long long vox, int nb, const float* sub, float* out, hipStream_t st) {
  if (vox % 64 || C > 128) return hipErrorInvalidValue;
  hipLaunchKernelGGL(export_planar_kernel, dim3((unsigned)(vox / 64), nb), dim3(256), 64 * (C + 1) * sizeof(float), st, in, C, vox, sub, out);
  return hipGetLastError();
}

}  // namespace amx
