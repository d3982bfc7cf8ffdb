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

#include "amx_device.h"
#include "amx_gemm.h"

namespace amx {

template <int MT, int NT, int EPI, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int ncb = (p.ntiles + NT - 1) / NT;                      // column blocks; consecutive workgroups share the token rows
  const int cb = blockIdx.x % ncb, rb = blockIdx.x / ncb;
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

  f16x8 wh[2][NT], ah[2][MT], wl[2][SPLIT ? NT : 1], al[2][SPLIT ? MT : 1];
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
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[buf][t], ah[buf][u], acc[t][u], 0, 0, 0);
        if (SPLIT) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[buf][t], al[buf][u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[buf][t], ah[buf][u], acc[t][u], 0, 0, 0);
        }
      }
  };
  load(0, 0);
  for (int ks = 0; ks < KS; ks += 2) {
    if (ks + 1 < KS) load(ks + 1, 1);
    compute(0);
    if (ks + 2 < KS) load(ks + 2, 0);
    if (ks + 1 < KS) compute(1);
  }

  // ------------------------------------------------ epilogues: lane (li, g) holds features n .. n + 3 of token row m
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
      if (EPI == EPI_SWIGLU) {
        if (t & 1) continue;                                        // tile pair (2s, 2s + 1) = (gate, value) of hidden columns 16 s ..
        if (t + 1 < NT) {
          const f32x4 x = acc[t + 1 < NT ? t + 1 : t][u];
          const float4 bg = *(const float4*)(p.bias + n), bx = *(const float4*)(p.bias + n + 16);
          const int hcol = (nt0 + t) / 2 * 16 + 4 * g;
          const float gv[4] = {v[0] + bg.x, v[1] + bg.y, v[2] + bg.z, v[3] + bg.w};
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
        const float4 b4 = *(const float4*)(p.bias + c);
        const long long vox = row_off + ((long long)(par >> 2) * 2 * p.gh + ((par >> 1) & 1)) * 2 * p.gw + (par & 1);
        *(float4*)((float*)p.out + vox * p.Creal + c) = make_float4(v[0] + b4.x, v[1] + b4.y, v[2] + b4.z, v[3] + b4.w);
        continue;
      }
      if (n >= p.Nreal) continue;
      float4 b4 = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      float r[4] = {v[0] + b4.x, v[1] + b4.y, v[2] + b4.z, v[3] + b4.w};
      if (EPI == EPI_F32) {
        *(float4*)((float*)p.out + (long long)m * p.ldo + n) = make_float4(r[0], r[1], r[2], r[3]);
      } else if (EPI == EPI_F16) {
        *(uint2*)((f16*)p.out + (long long)m * p.ldo + n) =
            make_uint2(to_bits<f16>(r[0]) | ((unsigned)to_bits<f16>(r[1]) << 16), to_bits<f16>(r[2]) | ((unsigned)to_bits<f16>(r[3]) << 16));
      } else if (EPI == EPI_RESID) {                                // x += gamma * (W h + b): LayerScale residual, fp32 stream
        float* o = (float*)p.out + (long long)m * p.ldo + n;
        const float4 g4 = p.gamma ? *(const float4*)(p.gamma + n) : make_float4(1.f, 1.f, 1.f, 1.f);
        float4 cur = *(const float4*)o;
        cur.x += g4.x * r[0]; cur.y += g4.y * r[1]; cur.z += g4.z * r[2]; cur.w += g4.w * r[3];
        *(float4*)o = cur;
      } else if (EPI == EPI_TOKENS) {                               // token = projection + bias + position embedding, behind the registers
        const float4 pe = *(const float4*)(p.pos + (long long)(m % p.V) * p.ldo + n);
        *(float4*)((float*)p.out + row_off * p.ldo + n) = make_float4(r[0] + pe.x, r[1] + pe.y, r[2] + pe.z, r[3] + pe.w);
      }
    }
  }
}

template <int MT, int NT, int EPI, bool SPLIT>
static hipError_t launch_one(const GemmParams& p, hipStream_t st) {
  const int ncb = (p.ntiles + NT - 1) / NT, nrb = (p.M + MT * 64 - 1) / (MT * 64);
  hipLaunchKernelGGL((gemm_kernel<MT, NT, EPI, SPLIT>), dim3((unsigned)ncb * nrb), dim3(256), 0, st, p);
  return hipGetLastError();
}

template <int EPI, bool SPLIT>
static hipError_t launch_epi(const GemmParams& p, hipStream_t st) {
  // NT = 5 where the tile count is a multiple of 5 (396 -> 25 tiles, 3 x 396 -> 75): no idle column tiles.  SwiGLU pairs need an even NT.
  const bool five = EPI != EPI_SWIGLU && p.ntiles % 5 == 0;
  const bool small = (long long)((p.M + 255) / 256) * ((p.ntiles + (five ? 4 : 3)) / (five ? 5 : 4)) < 512;   // < 2 workgroups per CU: halve the row block
  if (SPLIT) return five ? launch_one<2, 5, EPI, true>(p, st) : launch_one<2, 4, EPI, true>(p, st);
  if (five) return small ? launch_one<2, 5, EPI, false>(p, st) : launch_one<4, 5, EPI, false>(p, st);
  return small ? launch_one<2, 4, EPI, false>(p, st) : launch_one<4, 4, EPI, false>(p, st);
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
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight packing: fp32 parameters -> f16 (hi [+ lo]) fragments [n tile][k step][lane][8].
//   mode 0  rows of up to three [rows_i][K] matrices stacked (q | k | v; a single Linear; a 1x1x1 conv weight [N][K])
//   mode 1  SwiGLU: tile 2 s = rows 16 s .. of src0 (gate), tile 2 s + 1 = the same rows of src1 (value)
//   mode 2  ConvTranspose3d weight [K = Cin][Cout][2][2][2]: n = parity * Cp + c
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
      } else {
        const int n = nt * 16 + i, par = n / Cp, c = n % Cp;
        if (par < 8 && c < Creal) w = s0[((long long)k * Creal + c) * 8 + par];
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

// SwiGLU bias in packed feature order: tile 2 s = gate bias of hidden 16 s .., tile 2 s + 1 = value bias of the same columns
__global__ void swiglu_bias_kernel(float* dst, const float* bg, const float* bx, int hidden) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * hidden) return;
  const int tile = i >> 4, h = (tile >> 1) * 16 + (i & 15);
  dst[i] = h < hidden ? ((tile & 1) ? bx : bg)[h] : 0.f;
}
hipError_t launch_swiglu_bias(float* dst, const float* bg, const float* bx, int hidden, hipStream_t st) {
  hipLaunchKernelGGL(swiglu_bias_kernel, dim3((2 * hidden + 255) / 256), dim3(256), 0, st, dst, bg, bx, hidden);
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
      if (GELU) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752f));
    }
    const f16 yh = (f16)y;
    hi[(long long)r * ldo + c] = yh;
    if (lo) lo[(long long)r * ldo + c] = (f16)(y - (float)yh);
  }
}

hipError_t launch_ln_rows(const void* in, int in_f16, long long ldi, int C, const float* w, const float* b, float eps, int M, int rows_out,
                          int rows_in, int skip, int gelu, void* hi, void* lo, int ldo, hipStream_t st) {
  if (C > 64 * 17 || ldo > 64 * 17 || ldo < C) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
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
  __shared__ double xbar[256];
  const int b = blockIdx.x;
  if ((int)threadIdx.x < K) {
    double xs = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) xs += colsum[((long long)b * nchunk + ch) * ld + threadIdx.x];
    xbar[threadIdx.x] = xs * inv_rows;
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
