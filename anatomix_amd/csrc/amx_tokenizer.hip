// anatomix_amd -- convolutional tokenizer of the 3D ViT variant (PatchEmbedDeeper: anatomix/model/vit3d/deep_tokenizer.py:12-68;
// the stages themselves are dynamic-network-architectures' PatchEmbed_deeper, restated in oracle/vit_ref.py::tokenizer):
//   stem  conv3(1 -> 32, zero padding) - InstanceNorm(affine) - LeakyReLU(0.01)
//   3 x   BasicBlockD, stride 2:  conv3 s2 - IN - LReLU - conv3 - IN  +  [AvgPool(2) - conv1 - IN]  -> LReLU
//   proj  conv1(128 -> embed_dim)                     (amx_gemm.hip, EPI_TOKENS)
//
// Precision: ten InstanceNorms in a row amplify 16-bit storage error beyond the 1e-3 budget (measured 1.3e-3 for f16 storage of the
// conv outputs alone), so this end of the network runs "strict": raw conv outputs are kept in fp32,
// normalised activations are stored as hi + lo f16 planes, every product is three MFMAs (Wh xh + Wh xl + Wl xh).
//
// Kernels (all zero-padded -- the UNet kernels are reflect-padded and stride 1, hence not reused):
//   tokstem   pass 0: statistics of the stem conv only (the 1-channel input is 32x smaller than the output, so the conv is simply
//             computed twice); pass 1: conv again, normalise, LeakyReLU, store hi / lo planes AND their 2x2x2 average (the skip
//             branch's AvgPool input) -- the raw stem output is never stored.  K = 27 taps padded to one 32-wide MFMA step.
//   tokconv   3x3x3 (stride 1 / 2) and 1x1x1 conv as implicit GEMM: weights are the A operand (fragment-packed), a lane's B
//             fragment is 8 channels of one input voxel of the current tap, loaded straight from global memory (zero outside the
//             volume) -- no LDS; fp32 raw output + per-wave {sum, sumsq} per channel in the epilogue.
//   finalize / apply / combine: statistics -> (scale, shift); elementwise normalise + LeakyReLU (+ residual sum, + AvgPool).
#include <stdio.h>

#include "amx_device.h"
#include "amx_gemm.h"

namespace amx {

__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
  unsigned short h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const f16 a = (f16)v[e];
    h[e] = __builtin_bit_cast(unsigned short, a);
    l[e] = to_bits<f16>(v[e] - (float)a);
  }
  hi = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
  lo = make_uint4(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16), l[4] | ((unsigned)l[5] << 16), l[6] | ((unsigned)l[7] << 16));
}
__device__ __forceinline__ void split4(const float (&v)[4], uint2& hi, uint2& lo) {
  unsigned short h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f16 a = (f16)v[e];
    h[e] = __builtin_bit_cast(unsigned short, a);
    l[e] = to_bits<f16>(v[e] - (float)a);
  }
  hi = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
  lo = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
}
// sum over the 16 lanes of a lane group (lanes 16 g .. 16 g + 15)
__device__ __forceinline__ float sum16(float x) {
  x += __shfl_xor(x, 1, 64);
  x += __shfl_xor(x, 2, 64);
  x += __shfl_xor(x, 4, 64);
  x += __shfl_xor(x, 8, 64);
  return x;
}

// =====================================================================================================================
// 3x3x3 / 1x1x1 convolution, zero padding, stride 1 or 2.  grid (row blocks, Cout / (16 NT)), block 256 = 4 waves along M;
// a wave owns MT tiles of 16 consecutive output voxels (raster order) x NT tiles of 16 output channels.
template <int TAPS, int STRIDE, int MT, int NT, bool XS>
__global__ __launch_bounds__(256) void tokconv_kernel(TokConvParams p) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each).  Each XCD gets a CONTIGUOUS eighth of the
  // output rows, in order, so the input rows that neighbouring output rows / z-slices share are fetched into one L2 close in time
  // (plain order: the 32 -> 32 stride-2 conv at 128^3 fetched 3.47 GB for a 1.07 GB input, profiles/r03_vit_pmc_traffic.json).
  const int per_xcd = gridDim.x >> 3;
  int blk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  // ... and WALKS Z FIRST: workgroup = 4 waves x MT tiles of 16 raster-order voxels = a few rows of ONE z-plane; consecutive
  // workgroups take the same rows of consecutive z-planes, which share one (stride 2) or two (stride 1) of their three input
  // planes -- re-used while still in L2 -- instead of consecutive rows, which share one input row in nine.
  {
    const int per_plane = p.Ho * p.Wo / (64 * MT), per_sample = per_plane * p.Do;      // workgroups per z-plane / per sample
    if (per_plane >= 1 && per_plane * 64 * MT == p.Ho * p.Wo && blk < per_sample * p.N) {
      const int bs = blk / per_sample, r = blk - bs * per_sample;
      blk = bs * per_sample + (r % p.Do) * per_plane + r / p.Do;
    }
  }
  const int wid = blk * 4 + wave;                                      // wave index over all samples
  const long long tiles = (long long)p.N * p.Do * p.Ho * p.Wo / 16;
  if ((long long)wid * MT >= tiles) return;
  const int nt0 = blockIdx.y * NT;
  const int nch = p.Cin / 32, KS = TAPS * nch;

  int tb[MT], tz[MT], ty[MT], tx[MT];                                   // this lane's output voxel of every tile (16 consecutive voxels in raster order)
#pragma unroll
  for (int u = 0; u < MT; ++u) {
    const long long m = ((long long)wid * MT + u) * 16 + li;
    tx[u] = (int)(m % p.Wo);
    const long long r1 = m / p.Wo;
    ty[u] = (int)(r1 % p.Ho);
    const long long r2 = r1 / p.Ho;
    tz[u] = (int)(r2 % p.Do);
    tb[u] = (int)(r2 / p.Do);
  }
  const long long lo_off = XS ? p.x_lo - p.x_hi : 0, wlo_off = p.w_lo - p.w_hi;
  const char* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = p.w_hi + ((long long)(nt0 + t) * KS * 64 + lane) * 16;

  f32x4 acc[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < MT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  f16x8 wh[2][NT], wl[2][NT], xh[2][MT], xl[2][XS ? MT : 1];   // XS: the input has a remainder plane (three MFMAs per product), else two
  // (valid when the wave's voxels share one z-plane: whole rows per wave)
  const bool zflip = (p.Ho * p.Wo) % (16 * MT) == 0 && (__builtin_amdgcn_readfirstlane(tz[0]) & 1);
  const bool yflip = p.Wo == 16 * MT && (__builtin_amdgcn_readfirstlane(ty[0]) & 1);
  int ks_n = 0;                                                         // the K step the next load() fetches (clamped: the tail re-fetches the last)
  auto load = [&](const int buf) {
    const int ksq = ks_n < KS ? ks_n : KS - 1, tap_q = ksq / nch, ch_n = ksq - tap_q * nch;
    // odd z-planes take their kz taps in REVERSE: the input plane two neighbouring output planes share is then read by both at the
    // same phase of their K loops (the even one's last third, the odd one's last third) -- while it is still in L2
    int tap_n = tap_q;
    if (TAPS == 27) {
      const int kzq = tap_q / 9, kyq = (tap_q / 3) % 3;
      tap_n = (zflip ? 2 - kzq : kzq) * 9 + (yflip ? 2 - kyq : kyq) * 3 + tap_q % 3;    // the same for ky between neighbouring rows (one row per wave)
    }
    const int ksl = tap_n * nch + ch_n;
    const int kz = TAPS == 1 ? 1 : tap_n / 9, ky = TAPS == 1 ? 1 : (tap_n / 3) % 3, kx = TAPS == 1 ? 1 : tap_n % 3;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      wh[buf][t] = *(const f16x8*)(wp[t] + (long long)ksl * 1024);
      wl[buf][t] = *(const f16x8*)(wp[t] + wlo_off + (long long)ksl * 1024);
    }
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const int iz = tz[u] * STRIDE + kz - 1, iy = ty[u] * STRIDE + ky - 1, ix = tx[u] * STRIDE + kx - 1;
      const bool ok = iz >= 0 && iz < p.D && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const long long vox = (((long long)tb[u] * p.D + iz) * p.H + iy) * p.W + ix;
      const char* src = p.x_hi + (vox * p.Cin + ch_n * 32 + g * 8) * 2;
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      xh[buf][u] = ok ? *(const f16x8*)src : z;
      if (XS) xl[buf][u] = ok ? *(const f16x8*)(src + lo_off) : z;
    }
    ++ks_n;
  };
  auto compute = [&](const int buf) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[buf][t], xh[buf][u], acc[t][u], 0, 0, 0);
        if (XS) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[buf][t], xl[buf][XS ? u : 0], acc[t][u], 0, 0, 0);
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[buf][t], xh[buf][u], acc[t][u], 0, 0, 0);
      }
  };
  // unconditional prefetch (a conditional load block makes the waitcnt pass drain the queue with vmcnt(0) at the join)
  load(0);
  int ks = 0;
  for (; ks + 2 <= KS; ks += 2) {
    load(1);
    __builtin_amdgcn_sched_barrier(0);        // and the loads stay in front of the MFMAs they overlap
    compute(0);
    __builtin_amdgcn_sched_barrier(0);
    load(0);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (ks < KS) compute(0);

  // epilogue: lane (li, g) holds channels n .. n + 3 of output voxel (tile u, li)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = (nt0 + t) * 16 + 4 * g;
    const float4 b4 = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const float r[4] = {acc[t][u][0] + b4.x, acc[t][u][1] + b4.y, acc[t][u][2] + b4.z, acc[t][u][3] + b4.w};
      const long long m = ((long long)wid * MT + u) * 16 + li;
      *(float4*)(p.raw + m * p.Cout + n) = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] += r[j]; q[j] += r[j] * r[j]; }
    }
    if (p.stats) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] = sum16(s[j]); q[j] = sum16(q[j]); }
      if (li == 0) {
        float* o = p.stats + ((long long)wid * p.Cout + n) * 2;
        *(float4*)o = make_float4(s[0], q[0], s[1], q[1]);
        *(float4*)(o + 4) = make_float4(s[2], q[2], s[3], q[3]);
      }
    }
  }
}

static int tokconv_mt(const TokConvParams& p) {          // 64 voxels per wave when that still gives >= 8 waves per CU
  const long long vox = (long long)p.N * p.Do * p.Ho * p.Wo;
  return vox / 64 >= 2048 ? 4 : 2;
}
int tokconv_slots(const TokConvParams& p) { return (int)((long long)p.Do * p.Ho * p.Wo / (16 * tokconv_mt(p))); }

template <int TAPS, int STRIDE>
static hipError_t launch_tokconv_ts(const TokConvParams& p, hipStream_t st) {
  const int mt = tokconv_mt(p), nt = p.Cout >= 64 ? 4 : 2;
  const long long waves = (long long)p.N * p.Do * p.Ho * p.Wo / (16 * mt);
  const dim3 grid((unsigned)(((waves + 3) / 4 + 7) / 8 * 8), p.Cout / (16 * nt)), block(256);      // a multiple of 8 workgroups (XCD-aware order)
#define AMX_TC(M_, N_)                                                                                        \
  do {                                                                                                        \
    if (p.x_lo) hipLaunchKernelGGL((tokconv_kernel<TAPS, STRIDE, M_, N_, true>), grid, block, 0, st, p);      \
    else hipLaunchKernelGGL((tokconv_kernel<TAPS, STRIDE, M_, N_, false>), grid, block, 0, st, p);           \
  } while (0)
  if (mt == 4 && nt == 4) AMX_TC(4, 4);
  else if (mt == 4) AMX_TC(4, 2);
  else if (nt == 4) AMX_TC(2, 4);
  else AMX_TC(2, 2);
#undef AMX_TC
  return hipGetLastError();
}

hipError_t launch_tokconv(const TokConvParams& p, int taps, int stride, hipStream_t st) {
  if (p.Cin % 32 || p.Cout % 32 || ((long long)p.Do * p.Ho * p.Wo) % 64) return hipErrorInvalidValue;   // a wave never straddles two samples
  if (taps == 27 && stride == 2) return launch_tokconv_ts<27, 2>(p, st);
  if (taps == 27 && stride == 1) return launch_tokconv_ts<27, 1>(p, st);
  if (taps == 1 && stride == 1) return launch_tokconv_ts<1, 1>(p, st);
  return hipErrorInvalidValue;
}

// conv weight [Cout][Cin][taps] fp32 -> fragments [Cout / 16][tap * (Cin / 32) + chunk][lane][8], hi and lo planes
__global__ void pack_tokconv_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, f16* __restrict__ hi, f16* __restrict__ lo) {
  const int nch = Cin / 32, KS = taps * nch;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)(Cout / 16) * KS * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) % KS, nt = (idx >> 6) / KS;
  const int tap = ks / nch, ch = ks % nch, n = nt * 16 + (lane & 15), c0 = ch * 32 + (lane >> 4) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = w[((long long)n * Cin + c0 + e) * taps + tap];
  uint4 h, l;
  split8(v, h, l);
  ((uint4*)hi)[idx] = h;
  ((uint4*)lo)[idx] = l;
}
hipError_t launch_pack_tokconv(const float* w, int Cout, int Cin, int taps, void* hi, void* lo, hipStream_t st) {
  const long long n = (long long)(Cout / 16) * taps * (Cin / 32) * 64;
  hipLaunchKernelGGL(pack_tokconv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, Cout, Cin, taps, (f16*)hi, (f16*)lo);
  return hipGetLastError();
}

// =====================================================================================================================
// Stem: 1 -> 32 channels.  A wave owns the 2 x 2 rows (z, z + 1) x (y, y + 1) of the volume and walks them in x, 16 voxels at
// a time: 4 tiles x 2 channel tiles, K = 32 (27 taps + 5 zero columns).  Lane (li, g) of a B fragment gathers taps 8 g .. 8 g + 7
// of voxel li (fp32 loads, zero outside the volume) and splits them into hi + lo.
template <int PASS>
__global__ __launch_bounds__(256) void tokstem_kernel(TokStemParams p) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int wid = blockIdx.x * 4 + wave;
  const int hy = p.H / 2, hz = p.D / 2;
  if (wid >= p.N * hz * hy) return;
  const int y0 = (wid % hy) * 2, z0 = (wid / hy % hz) * 2, b = wid / (hy * hz);
  const f16x8 wh0 = *(const f16x8*)(p.w_hi + lane * 16), wh1 = *(const f16x8*)(p.w_hi + 1024 + lane * 16);
  const f16x8 wl0 = *(const f16x8*)(p.w_lo + lane * 16), wl1 = *(const f16x8*)(p.w_lo + 1024 + lane * 16);
  int dz[8], dy[8], dx[8];
  bool tv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int t = 8 * g + e;
    tv[e] = t < 27;
    dz[e] = t / 9 - 1;
    dy[e] = (t / 3) % 3 - 1;
    dx[e] = t % 3 - 1;
  }
  const float* xb = p.x + (long long)b * p.D * p.H * p.W;
  float4 bias[2], sc[2], sh[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bias[t] = *(const float4*)(p.bias + t * 16 + 4 * g);
    if (PASS == 1) {
      sc[t] = *(const float4*)(p.scale + b * 32 + t * 16 + 4 * g);
      sh[t] = *(const float4*)(p.shift + b * 32 + t * 16 + 4 * g);
    }
  }
  float s[2][4] = {}, q[2][4] = {};
  // The taps of the NEXT 16 voxels are gathered before this step's results are stored: loads and stores share the wave's in-order
  // memory counter, so a gather issued after the stores would wait for their acknowledgement (and the gather latency itself
  // overlaps the epilogue).
  auto gather = [&](int x0, float (&v)[4][8]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int z = z0 + (u >> 1), y = y0 + (u & 1), x = x0 + li;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int iz = z + dz[e], iy = y + dy[e], ix = x + dx[e];
        const bool ok = tv[e] && iz >= 0 && iz < p.D && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        v[u][e] = ok ? xb[((long long)iz * p.H + iy) * p.W + ix] : 0.f;
      }
    }
  };
  float vc[4][8];
  gather(0, vc);
  for (int x0 = 0; x0 < p.W; x0 += 16) {
    f32x4 acc[2][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      uint4 h, l;
      split8(vc[u], h, l);
      const f16x8 xh = __builtin_bit_cast(f16x8, h), xl = __builtin_bit_cast(f16x8, l);
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xh, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xl, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, xh, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xh, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xl, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, xh, a1, 0, 0, 0);
      acc[0][u] = a0;
      acc[1][u] = a1;
    }
    gather(x0 + 16 < p.W ? x0 + 16 : x0, vc);          // unconditional (the last step re-gathers its own taps)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float bb[4] = {bias[t].x, bias[t].y, bias[t].z, bias[t].w};
      if (PASS == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float r = acc[t][u][j] + bb[j];
            s[t][j] += r;
            q[t][j] += r * r;
          }
      } else {
        const float sa[4] = {sc[t].x, sc[t].y, sc[t].z, sc[t].w}, ta[4] = {sh[t].x, sh[t].y, sh[t].z, sh[t].w};
        float pool[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float yv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float r = (acc[t][u][j] + bb[j]) * sa[j] + ta[j];
            yv[j] = r > 0.f ? r : r * p.slope;
            pool[j] += yv[j];
          }
          uint2 h, l;
          split4(yv, h, l);
          const long long vox = (((long long)b * p.D + z0 + (u >> 1)) * p.H + y0 + (u & 1)) * p.W + x0 + li;
          *(uint2*)(p.h_hi + (vox * 32 + t * 16 + 4 * g) * 2) = h;
          if (p.h_lo) *(uint2*)(p.h_lo + (vox * 32 + t * 16 + 4 * g) * 2) = l;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) pool[j] = (pool[j] + __shfl_xor(pool[j], 1, 64)) * 0.125f;
        if (!(li & 1)) {
          uint2 h, l;
          split4(pool, h, l);
          const long long pv = (((long long)b * hz + z0 / 2) * hy + y0 / 2) * (p.W / 2) + (x0 + li) / 2;
          *(uint2*)(p.p_hi + (pv * 32 + t * 16 + 4 * g) * 2) = h;
          *(uint2*)(p.p_lo + (pv * 32 + t * 16 + 4 * g) * 2) = l;
        }
      }
    }
  }
  if (PASS == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[t][j] = sum16(s[t][j]); q[t][j] = sum16(q[t][j]); }
      if (li == 0) {
        float* o = p.stats + ((long long)wid * 32 + t * 16 + 4 * g) * 2;
        *(float4*)o = make_float4(s[t][0], q[t][0], s[t][1], q[t][1]);
        *(float4*)(o + 4) = make_float4(s[t][2], q[t][2], s[t][3], q[t][3]);
      }
    }
  }
}

int tokstem_slots(int D, int H, int W) { return (D / 2) * (H / 2); }

hipError_t launch_tokstem(const TokStemParams& p, int pass, hipStream_t st) {
  if (p.W % 16 || (p.D & 1) || (p.H & 1)) return hipErrorInvalidValue;
  const int waves = p.N * (p.D / 2) * (p.H / 2);
  if (pass == 0) hipLaunchKernelGGL(tokstem_kernel<0>, dim3((waves + 3) / 4), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(tokstem_kernel<1>, dim3((waves + 3) / 4), dim3(256), 0, st, p);
  return hipGetLastError();
}

// stem weight [32][1][3][3][3] -> two fragments (channel tiles), K index = tap (27 .. 31 zero)
__global__ void pack_tokstem_kernel(const float* __restrict__ w, f16* __restrict__ hi, f16* __restrict__ lo) {
  const int idx = threadIdx.x;                       // 128 = 2 tiles x 64 lanes
  const int lane = idx & 63, nt = idx >> 6, n = nt * 16 + (lane & 15), k0 = (lane >> 4) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = k0 + e < 27 ? w[n * 27 + k0 + e] : 0.f;
  uint4 h, l;
  split8(v, h, l);
  ((uint4*)hi)[idx] = h;
  ((uint4*)lo)[idx] = l;
}
hipError_t launch_pack_tokstem(const float* w, void* hi, void* lo, hipStream_t st) {
  hipLaunchKernelGGL(pack_tokstem_kernel, dim3(1), dim3(128), 0, st, w, (f16*)hi, (f16*)lo);
  return hipGetLastError();
}

// =====================================================================================================================
// one workgroup per (n, c): slots {sum, sumsq} -> scale, shift (double combine; biased variance, eps inside the root)
__global__ __launch_bounds__(256) void tok_finalize_kernel(const float* __restrict__ stats, int slots, int C, double inv_count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double ss[256], sq[256];
  const int n = blockIdx.x / C, c = blockIdx.x % C;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < slots; i += 256) {
    const float2 v = *(const float2*)(stats + (((long long)n * slots + i) * C + c) * 2);
    s += v.x;
    q += v.y;
  }
  ss[threadIdx.x] = s;
  sq[threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; sq[threadIdx.x] += sq[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = ss[0] * inv_count;
    double var = sq[0] * inv_count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double k = (gamma ? (double)gamma[c] : 1.0) / sqrt(var + (double)eps);
    scale[n * C + c] = (float)k;
    shift[n * C + c] = (float)((beta ? (double)beta[c] : 0.0) - mean * k);
  }
}
hipError_t launch_tok_finalize(const float* stats, int N, int slots, int C, long long count, const float* gamma, const float* beta, float eps,
                               float* scale, float* shift, hipStream_t st) {
  hipLaunchKernelGGL(tok_finalize_kernel, dim3(N * C), dim3(256), 0, st, stats, slots, C, 1.0 / (double)count, gamma, beta, eps, scale, shift);
  return hipGetLastError();
}

// one thread = 8 channels of one voxel
__global__ __launch_bounds__(256) void tok_apply_kernel(const float* __restrict__ raw, long long vox, int C, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float slope, char* __restrict__ hi, char* __restrict__ lo,
                                                        long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;    // index of an 8-channel group
  if (i >= total) return;
  const int cg = C / 8, c0 = (int)(i % cg) * 8;
  const int n = (int)(i / cg / vox);
  const float4 a = *(const float4*)(raw + i * 8), b = *(const float4*)(raw + i * 8 + 4);
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  float y[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float r = x[e] * scale[n * C + c0 + e] + shift[n * C + c0 + e];
    y[e] = r > 0.f ? r : r * slope;
  }
  uint4 h, l;
  split8(y, h, l);
  *(uint4*)(hi + i * 16) = h;
  *(uint4*)(lo + i * 16) = l;
}
hipError_t launch_tok_apply(const float* raw, int N, long long vox, int C, const float* scale, const float* shift, float slope, void* hi, void* lo,
                            hipStream_t st) {
  const long long total = (long long)N * vox * C / 8;
  hipLaunchKernelGGL(tok_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, raw, vox, C, scale, shift, slope, (char*)hi, (char*)lo, total);
  return hipGetLastError();
}

// one thread = 8 channels of one POOLED voxel: the eight voxels below it are combined, stored, and averaged
__global__ __launch_bounds__(256) void tok_combine_kernel(const float* __restrict__ ra, const float* __restrict__ rb, int N, int D, int H, int W, int C,
                                                          const float* __restrict__ sa, const float* __restrict__ ta, const float* __restrict__ sb,
                                                          const float* __restrict__ tb, float slope, char* __restrict__ h_hi, char* __restrict__ h_lo,
                                                          char* __restrict__ p_hi, char* __restrict__ p_lo) {
  const int cg = C / 8, pd = D / 2, ph = H / 2, pw = W / 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)N * pd * ph * pw * cg;
  if (i >= total) return;
  const int c0 = (int)(i % cg) * 8;
  long long r = i / cg;
  const int x = (int)(r % pw); r /= pw;
  const int y = (int)(r % ph); r /= ph;
  const int z = (int)(r % pd);
  const int n = (int)(r / pd);
  float ka[8], oa[8], kb[8], ob[8], pool[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ka[e] = sa[n * C + c0 + e]; oa[e] = ta[n * C + c0 + e];
    kb[e] = sb[n * C + c0 + e]; ob[e] = tb[n * C + c0 + e];
    pool[e] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const long long vox = (((long long)n * D + 2 * z + (s >> 2)) * H + 2 * y + ((s >> 1) & 1)) * W + 2 * x + (s & 1);
    const float4 a0 = *(const float4*)(ra + vox * C + c0), a1 = *(const float4*)(ra + vox * C + c0 + 4);
    const float4 b0 = *(const float4*)(rb + vox * C + c0), b1 = *(const float4*)(rb + vox * C + c0 + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float hv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (av[e] * ka[e] + oa[e]) + (bv[e] * kb[e] + ob[e]);
      hv[e] = v > 0.f ? v : v * slope;
      pool[e] += hv[e];
    }
    uint4 h, l;
    split8(hv, h, l);
    *(uint4*)(h_hi + (vox * C + c0) * 2) = h;
    *(uint4*)(h_lo + (vox * C + c0) * 2) = l;
  }
  if (p_hi) {
#pragma unroll
    for (int e = 0; e < 8; ++e) pool[e] *= 0.125f;
    uint4 h, l;
    split8(pool, h, l);
    *(uint4*)(p_hi + i * 16) = h;
    *(uint4*)(p_lo + i * 16) = l;
  }
}
hipError_t launch_tok_combine(const float* raw_a, const float* raw_b, int N, int D, int H, int W, int C, const float* sa, const float* ta,
                              const float* sb, const float* tb, float slope, void* h_hi, void* h_lo, void* p_hi, void* p_lo, hipStream_t st) {
  if ((D | H | W) & 1 || C % 8) return hipErrorInvalidValue;
  const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(tok_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, raw_a, raw_b, N, D, H, W, C, sa, ta, sb, tb, slope,
                     (char*)h_hi, (char*)h_lo, (char*)p_hi, (char*)p_lo);
  return hipGetLastError();
}

}  // namespace amx
