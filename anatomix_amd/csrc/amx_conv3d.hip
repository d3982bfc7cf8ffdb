// anatomix_amd -- implicit-GEMM 3x3x3 reflect-padded convolution for gfx950 (CDNA4).
//
// Replaces the nn.Conv3d(k=3, padding='same', padding_mode='reflect') (+ folded eval BatchNorm3d
// + ReLU, + nearest Upsample + torch.cat on the input side) modules that
// /root/reference/anatomix/model/network.py:309-465 strings together.
//
// Formulation (per workgroup = one output brick TZ x TY x TX of one sample, 16*Q output channels):
//   D[cout][voxel] += W[cout][k] * X[k][voxel],   k = (tap, cin)
//   A operand  = weights (rows = output channels)  -> per-lane accumulators hold 4 consecutive
//                channels of one voxel, so the epilogue writes 8*Q contiguous bytes per lane.
//   B operand  = activations: lane (i = lane&15, g = lane>>4) reads ONE 16-byte LDS slot = 8
//                channels of voxel i at the tap selected by g (see amx_common.h step table).
//   MFMA       = v_mfma_f32_16x16x32_{f16,bf16}: K = 32 = 2 taps x 16 input channels.
// LDS image of the input halo: plane-major [8-channel plane][halo voxel] x 16 B, so the 16 lanes
// of a B-fragment read hit 16 consecutive 16-B slots (ds_read_b128, conflict-free for every tap).
// Reflect padding, the nearest-x2 upsample of the low-resolution segment and the skip||up
// channel concat are all resolved in the per-lane global address of the halo gather; none of
// them is ever materialised in HBM.
#include <stdio.h>

#include "amx_device.h"

namespace amx {

template <typename T, int WZ, int WY, int WX, int NWZ, int NWY, int Q, int NCH, int OUTMODE>
struct ConvCfg {
  static constexpr int TZ = WZ * NWZ, TY = WY * NWY, TX = WX;
  static constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  static constexpr int PLANE = ((HV * 16 + 255) / 256) * 256;
  static constexpr int HALO = 2 * PLANE;            // bytes per 16-channel sub-chunk
  static constexpr int WOFF = NCH * HALO;
  static constexpr int WSUB = kSteps * Q * 1024;    // weight bytes per 16-channel sub-chunk
  static constexpr int LDS_BYTES = NCH * (HALO + WSUB);
  static constexpr int LX = WX >= 16 ? 16 : 8;      // column tile: LY rows x LX voxels = 16
  static constexpr int LY = 16 / LX;
  static constexpr int XT = WX / LX;
  static constexpr int YT = WY / LY;
  static constexpr int CTW = WZ * YT * XT;          // column tiles per wave
  static_assert(NWZ * NWY == 4, "4 waves per workgroup");
  static_assert(WY % LY == 0 && WX % LX == 0, "wave sub-brick must tile into 16-voxel columns");
};

template <typename T, int WZ, int WY, int WX, int NWZ, int NWY, int Q, int NCH, int OUTMODE>
__global__ __launch_bounds__(256) void conv3d_k3_kernel(const ConvParams p) {
  typedef ConvCfg<T, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE> C;
  typedef typename Ops<T>::vec8 vec8;
  constexpr int HY = C::HY, HX = C::HX, HV = C::HV, PLANE = C::PLANE, HALO = C::HALO;
  constexpr int WOFF = C::WOFF, CTW = C::CTW, LX = C::LX, LY = C::LY, XT = C::XT, YT = C::YT;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;

  // ---- brick decode; blocks of one XCD (b % 8) get a contiguous run of bricks so that
  //      neighbouring halos are re-read from that XCD's own L2.
  const int nb = p.nbz * p.nby * p.nbx * p.N;
  int b = blockIdx.x;
  if ((nb & 7) == 0) b = (b & 7) * (nb >> 3) + (b >> 3);
  const int bx = b % p.nbx;
  int t = b / p.nbx;
  const int by = t % p.nby;
  t /= p.nby;
  const int bz = t % p.nbz;
  const int n = t / p.nbz;
  const int z0 = bz * C::TZ, y0 = by * C::TY, x0 = bx * C::TX;
  const int cgrp = blockIdx.y;

  const int nchunk = (p.C0 + p.C1) >> 4;
  const int nstage = nchunk / NCH;

  f32x4 acc[CTW][Q];
#pragma unroll
  for (int c = 0; c < CTW; ++c)
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- lane-constant LDS read bases
  const int wz = wave / NWY, wy = wave % NWY;
  const int dy = (LX == 16) ? 0 : (li >> 3);
  const int dx = (LX == 16) ? li : (li & 7);
  const int lanehv = ((wz * WZ) * HY + wy * WY + dy) * HX + dx;
  const int lanebase = (g & 1) * PLANE + lanehv * 16;
  const int hi = g >> 1;
  const int base_d1 = lanebase + hi * 16;
  const int base_dx = lanebase + hi * 16 * HX;
  const int base_dz = lanebase + hi * 16 * HX * HY;
  const int base_d0 = lanebase;

  const long long src0_n = (long long)n * p.s0n;
  const long long src1_n = (long long)n * p.s1n;
  const char* wsrc = p.wpk + (long long)cgrp * nchunk * C::WSUB;

  for (int stage = 0; stage < nstage; ++stage) {
    if (stage > 0) __syncthreads();
    // ---- stage the halo of NCH x 16 input channels: unit = (halo voxel, sub-chunk, plane)
    {
      constexpr int UPV = 2 * NCH;
      constexpr int NU = HV * UPV;
#pragma unroll 4
      for (int u = tid; u < NU; u += 256) {
        const int hv = u / UPV;
        const int sub = u % UPV;
        const int k = sub >> 1, pl = sub & 1;
        const int hx = hv % HX;
        const int t2 = hv / HX;
        const int hy = t2 % HY;
        const int hz = t2 / HY;
        const int gz = reflect_clamp(z0 + hz - 1, p.D);
        const int gy = reflect_clamp(y0 + hy - 1, p.H);
        const int gx = reflect_clamp(x0 + hx - 1, p.W);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (p.src0_f32c1) {
          if (pl == 0 && k == 0 && stage == 0) {
            const float f =
                *(const float*)(p.src0 + src0_n + gz * p.s0z + gy * p.s0y + gx * p.s0x);
            v.x = (unsigned)to_bits<T>(f);
          }
        } else {
          const int ch = ((stage * NCH + k) << 4) + (pl << 3);
          const bool second = ch >= p.C0;
          const int sh = second ? 1 : 0;
          const char* base = second ? p.src1 + src1_n : p.src0 + src0_n;
          const long long sz = second ? p.s1z : p.s0z;
          const long long sy = second ? p.s1y : p.s0y;
          const long long sx = second ? p.s1x : p.s0x;
          const int cc = second ? ch - p.C0 : ch;
          v = *(const uint4*)(base + (gz >> sh) * sz + (gy >> sh) * sy + (gx >> sh) * sx + cc * 2);
        }
        *(uint4*)(smem + k * HALO + pl * PLANE + hv * 16) = v;
      }
    }
    // ---- stage the packed weights of these sub-chunks (linear copy)
    {
      const char* ws = wsrc + (long long)stage * NCH * C::WSUB;
      constexpr int NW = NCH * kSteps * Q * 64;
#pragma unroll 4
      for (int u = tid; u < NW; u += 256)
        *(uint4*)(smem + WOFF + u * 16) = *(const uint4*)(ws + (long long)u * 16);
    }
    __syncthreads();

    // ---- MFMA sweep: 14 steps per 16-channel sub-chunk
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
#pragma unroll
      for (int s = 0; s < kSteps; ++s) {
        const int kz = s < 9 ? s / 3 : (s < 12 ? s - 9 : (s == 12 ? 0 : 2));
        const int ky = s < 9 ? s % 3 : (s < 12 ? 0 : 2);
        const int kx = s < 9 ? 0 : 2;
        const int tapoff = ((kz * HY + ky) * HX + kx) * 16;
        const int bsel = s < 9 ? base_d1 : (s < 12 ? base_dx : (s == 12 ? base_dz : base_d0));
        vec8 a[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q)
          a[q] = *(const vec8*)(smem + WOFF + ((k * kSteps + s) * Q + q) * 1024 + lane * 16);
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
          const int coff = ((cz * HY + cy * LY) * HX + cx * LX) * 16;
          const vec8 bf = *(const vec8*)(smem + bsel + k * HALO + tapoff + coff);
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[c][q] = Ops<T>::mfma(a[q], bf, acc[c][q]);
        }
      }
    }
  }

  // ---- epilogue: bias (folded norm) + activation; lane holds channels cb .. cb+4Q-1 of one voxel
  const int cb = cgrp * 16 * Q + g * 4 * Q;
  float bias[4 * Q];
#pragma unroll
  for (int j = 0; j < 4 * Q; ++j) bias[j] = p.bias ? p.bias[cb + j] : 0.f;

#pragma unroll
  for (int c = 0; c < CTW; ++c) {
    const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
    const int z = z0 + wz * WZ + cz;
    const int y = y0 + wy * WY + cy * LY + dy;
    const int x = x0 + cx * LX + dx;
    const bool ok = (z < p.D) & (y < p.H) & (x < p.W);
    float v[4 * Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float f = acc[c][q][j] + bias[q * 4 + j];
        if (p.act == ACT_RELU) f = f > 0.f ? f : 0.f;
        else if (p.act == ACT_LRELU) f = f > 0.f ? f : f * p.slope;
        v[q * 4 + j] = f;
      }
    if (!ok) continue;
    if (OUTMODE == 0) {
      char* dst = p.out + (long long)n * p.on + (long long)z * p.oz + (long long)y * p.oy +
                  (long long)x * p.ox + cb * 2;
      unsigned w[2 * Q];
#pragma unroll
      for (int j = 0; j < 2 * Q; ++j)
        w[j] = (unsigned)to_bits<T>(v[2 * j]) | ((unsigned)to_bits<T>(v[2 * j + 1]) << 16);
      if (Q == 1) {
        *(uint2*)dst = make_uint2(w[0], w[1]);
      } else {
#pragma unroll
        for (int j = 0; j < Q / 2; ++j)
          *(uint4*)(dst + j * 16) = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
      }
    } else {
      float* dst = p.out32 + (long long)n * p.pn + (long long)cb * p.pc + (long long)z * p.pz +
                   (long long)y * p.py + x;
      if (p.wmap) {
        const float wgt = p.wmap[((long long)z * p.H + y) * p.W + x];
#pragma unroll
        for (int j = 0; j < 4 * Q; ++j) dst[(long long)j * p.pc] += wgt * v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4 * Q; ++j) dst[(long long)j * p.pc] = v[j];
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// Weight packer: fp32 [Cout][Cin][3][3][3] (+ per-channel scale = folded norm gain) -> A fragments.
// Packed layout [cout_group = Cout/(16Q)][chunk = CinPad/16][step 14][q][lane 64][8] of T.
// Row m of MFMA tile q in cout group cg is output channel cg*16Q + (m>>2)*4Q + q*4 + (m&3), so
// that after the MFMA every lane owns 4Q CONSECUTIVE output channels (wide epilogue stores).
// -------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                    T* __restrict__ wpk, int CinReal, int CinPad, int Cout, int Q) {
  const long long total = (long long)(Cout / 16) * (CinPad / 16) * kSteps * 64 * 8;
  const int nchunk = CinPad / 16;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int e = idx & 7;
    const int lane = (idx >> 3) & 63;
    long long r = idx >> 9;
    const int q = r % Q;
    r /= Q;
    const int s = r % kSteps;
    r /= kSteps;
    const int chunk = r % nchunk;
    const int cg = r / nchunk;
    const int m = lane & 15, g = lane >> 4;
    const int cout = cg * 16 * Q + (m >> 2) * 4 * Q + q * 4 + (m & 3);
    const int tap = (g >> 1) ? tapB_index(s) : tapA_index(s);
    const int cin = chunk * 16 + (g & 1) * 8 + e;
    float v = 0.f;
    if (tap >= 0 && cin < CinReal) {
      v = w[((long long)cout * CinReal + cin) * 27 + tap];
      if (scale) v *= scale[cout];
    }
    wpk[idx] = (T)v;
  }
}

// Folds eval-mode BatchNorm (network.py:154-155 -> nn.BatchNorm3d) into a per-channel gain and
// shift: s = gamma / sqrt(var + eps), t = beta - mean * s (+ conv_bias * s).  Without a norm the
// gain is 1 and the shift is the conv bias.
__global__ void fold_norm_kernel(const float* gamma, const float* beta, const float* mean,
                                 const float* var, const float* conv_bias, float eps, int C,
                                 float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 1.f, t = 0.f;
  if (var) {
    s = (gamma ? gamma[c] : 1.f) / sqrtf(var[c] + eps);
    t = (beta ? beta[c] : 0.f) - mean[c] * s;
  }
  if (conv_bias) t += conv_bias[c] * s;
  scale[c] = s;
  shift[c] = t;
}

// 2x2x2 stride-2 pooling on channels-last 16-bit tensors (nn.MaxPool3d(2) / nn.AvgPool3d(2),
// network.py:297,368).  One thread = one output voxel x 8 channels (16 B).
template <typename T, int AVG>
__global__ void pool2_kernel(const char* __restrict__ in, char* __restrict__ out, int N, int Do,
                             int Ho, int Wo, int C) {
  const int c8n = C >> 3;
  const long long total = (long long)N * Do * Ho * Wo * c8n;
  const long long sx = (long long)C * 2, sy = sx * (Wo * 2), sz = sy * (Ho * 2);
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = idx % c8n;
    long long r = idx / c8n;
    const int x = r % Wo;
    r /= Wo;
    const int y = r % Ho;
    r /= Ho;
    const int z = r % Do;
    const int n = r / Do;
    const char* base = in + (long long)n * sz * (Do * 2) + (2 * z) * sz + (2 * y) * sy + (2 * x) * sx + c8 * 16;
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint4 raw = *(const uint4*)(base + (k >> 2) * sz + ((k >> 1) & 1) * sy + (k & 1) * sx);
      const unsigned wv[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned short bits = (unsigned short)(wv[e >> 1] >> ((e & 1) * 16));
        const float f = (float)__builtin_bit_cast(T, bits);
        if (k == 0) m[e] = f;
        else m[e] = AVG ? m[e] + f : (f > m[e] ? f : m[e]);
      }
    }
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = AVG ? m[2 * e] * 0.125f : m[2 * e];
      const float bq = AVG ? m[2 * e + 1] * 0.125f : m[2 * e + 1];
      o[e] = (unsigned)to_bits<T>(a) | ((unsigned)to_bits<T>(bq) << 16);
    }
    *(uint4*)(out + idx * 16) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// -------------------------------------------------------------------------------------------
// host-side launchers
// -------------------------------------------------------------------------------------------
static thread_local char g_kernel_name[64] = "";
const char* last_conv_kernel_name() { return g_kernel_name; }

template <typename T, int WZ, int WY, int WX, int NWZ, int NWY, int Q, int NCH, int OUTMODE>
static hipError_t launch_cfg(ConvParams p, hipStream_t st) {
  typedef ConvCfg<T, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE> C;
  snprintf(g_kernel_name, sizeof g_kernel_name, "conv3d_k3<%s,%dx%dx%d,q%d,nch%d,o%d>", __is_same(T, f16) ? "f16" : "bf16",
           C::TZ, C::TY, C::TX, Q, NCH, OUTMODE);
  auto kern = conv3d_k3_kernel<T, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  p.nbz = (p.D + C::TZ - 1) / C::TZ;
  p.nby = (p.H + C::TY - 1) / C::TY;
  p.nbx = (p.W + C::TX - 1) / C::TX;
  dim3 grid((unsigned)(p.nbz * p.nby * p.nbx * p.N), (unsigned)(p.Cout / (16 * Q)));
  hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, st, p);
  return hipGetLastError();
}

// Tile-shape heuristic.  Returns the Q (16-channel MFMA tiles per workgroup) a layer will be
// launched with; the packed-weight layout depends on it.
int conv_pick_q(int Cout, int W) {
  if (Cout == 16) return 1;
  if (Cout == 32) return 2;
  if (W >= 32) return 4;      // Cout >= 64 at 32^3 and larger
  if (W >= 16) return 2;      // 16^3: more cout groups to fill the chip
  return 1;
}

template <typename T, int OUTMODE>
static hipError_t launch_conv_t(const ConvParams& p, int Q, hipStream_t st) {
  const int nch = (p.C0 + p.C1) / 16;
  if (p.W >= 32 && Q == 1) return launch_cfg<T, 1, 8, 32, 4, 1, 1, 1, OUTMODE>(p, st);
  if (p.W >= 32 && Q == 2) return launch_cfg<T, 1, 4, 32, 4, 1, 2, 1, OUTMODE>(p, st);
  if (p.W >= 32 && Q == 4) return launch_cfg<T, 1, 2, 16, 4, 1, 4, 1, OUTMODE>(p, st);
  if (p.W >= 16) {
    if (Q == 1) return launch_cfg<T, 1, 2, 16, 4, 1, 1, 1, OUTMODE>(p, st);
    if (Q == 2) {
      if (nch % 2 == 0) return launch_cfg<T, 1, 2, 16, 4, 1, 2, 2, OUTMODE>(p, st);
      return launch_cfg<T, 1, 2, 16, 4, 1, 2, 1, OUTMODE>(p, st);
    }
    if (Q == 4) return launch_cfg<T, 1, 2, 16, 4, 1, 4, 1, OUTMODE>(p, st);
  }
  // W <= 8 (8^3 bottleneck and the tiny levels of small test volumes): 2x8 column tiles
  if (Q == 1) {
    if (nch % 4 == 0) return launch_cfg<T, 1, 2, 8, 4, 1, 1, 4, OUTMODE>(p, st);
    return launch_cfg<T, 1, 2, 8, 4, 1, 1, 1, OUTMODE>(p, st);
  }
  if (Q == 2) return launch_cfg<T, 1, 2, 8, 4, 1, 2, 1, OUTMODE>(p, st);
  if (Q == 4) return launch_cfg<T, 1, 2, 8, 4, 1, 4, 1, OUTMODE>(p, st);
  return hipErrorInvalidValue;
}

hipError_t launch_conv_v2(const ConvParams& p, int precision, int Q, hipStream_t st);
const char* last_conv_v2_kernel_name();
bool conv_zmarch_eligible(const ConvParams& p);
hipError_t launch_conv_zmarch(const ConvParams& p, int precision, hipStream_t st);
const char* last_conv_zm_kernel_name();

hipError_t launch_conv(const ConvParams& p, int precision, int Q, hipStream_t st) {
  const bool planar = p.out32 != nullptr;
  if (conv_zmarch_eligible(p) && Q == 1) {   // HBM-bound 16->16 full-resolution layers: z-marching ring kernel
    hipError_t e = launch_conv_zmarch(p, precision, st);
    snprintf(g_kernel_name, sizeof g_kernel_name, "%s", last_conv_zm_kernel_name());
    return e;
  }
  if (!p.src0_f32c1) {   // everything but the fp32 single-channel stem runs on the persistent DMA kernel
    hipError_t e = launch_conv_v2(p, precision, Q, st);
    snprintf(g_kernel_name, sizeof g_kernel_name, "%s", last_conv_v2_kernel_name());
    return e;
  }
  if (precision == 0) {
    return planar ? launch_conv_t<f16, 1>(p, Q, st) : launch_conv_t<f16, 0>(p, Q, st);
  }
  return planar ? launch_conv_t<bf16, 1>(p, Q, st) : launch_conv_t<bf16, 0>(p, Q, st);
}

hipError_t launch_pack_weights(const float* w, const float* scale, void* wpk, int CinReal, int CinPad,
                               int Cout, int Q, int precision, hipStream_t st) {
  const long long total = (long long)(Cout / 16) * (CinPad / 16) * kSteps * 64 * 8;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (precision == 0)
    hipLaunchKernelGGL(pack_weights_kernel<f16>, dim3(blocks), dim3(256), 0, st, w, scale, (f16*)wpk,
                       CinReal, CinPad, Cout, Q);
  else
    hipLaunchKernelGGL(pack_weights_kernel<bf16>, dim3(blocks), dim3(256), 0, st, w, scale,
                       (bf16*)wpk, CinReal, CinPad, Cout, Q);
  return hipGetLastError();
}

hipError_t launch_fold_norm(const float* gamma, const float* beta, const float* mean, const float* var,
                            const float* conv_bias, float eps, int C, float* scale, float* shift,
                            hipStream_t st) {
  hipLaunchKernelGGL(fold_norm_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, mean, var,
                     conv_bias, eps, C, scale, shift);
  return hipGetLastError();
}

hipError_t launch_pool2(const void* in, void* out, int N, int Do, int Ho, int Wo, int C, int avg,
                        int precision, hipStream_t st) {
  const long long total = (long long)N * Do * Ho * Wo * (C / 8);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
#define AMX_POOL(T, A)                                                                            \
  hipLaunchKernelGGL((pool2_kernel<T, A>), dim3(blocks), dim3(256), 0, st, (const char*)in, (char*)out, \
                     N, Do, Ho, Wo, C)
  if (precision == 0) { if (avg) AMX_POOL(f16, 1); else AMX_POOL(f16, 0); }
  else { if (avg) AMX_POOL(bf16, 1); else AMX_POOL(bf16, 0); }
#undef AMX_POOL
  return hipGetLastError();
}

}  // namespace amx
