// anatomix_amd -- shared pieces of the conv3d path: weight packing (A fragments), BatchNorm folding,
// 2x2x2 pooling, tile-shape heuristic and the dispatch between the conv kernels.
//
// The convolution kernels themselves live in
//   amx_conv3d_stem.hip    fp32 single-channel stem (27 taps in one MFMA)
//   amx_conv3d_zmarch.hip  z-marching ring kernel for the narrow HBM-bound layers
//   amx_conv3d_upcat.hip   merged-tap upsample+concat conv (48 -> 16 at full resolution)
//   amx_conv3d_v2.hip      generic persistent double-buffered kernel (every other layer)
// They replace the nn.Conv3d(k=3, padding='same', padding_mode='reflect') (+ folded eval BatchNorm3d
// + ReLU, + nearest Upsample + torch.cat on the input side) modules that
// /root/reference/anatomix/model/network.py:309-465 strings together.  Common formulation:
//   D[cout][voxel] += W[cout][k] * X[k][voxel],   k = (tap, cin)
//   A operand  = weights (rows = output channels)  -> per-lane accumulators hold 4 consecutive
//                channels of one voxel, so the epilogue writes 8*Q contiguous bytes per lane.
//   B operand  = activations: lane (i = lane&15, g = lane>>4) reads ONE 16-byte LDS slot = 8
//                channels of voxel i at the tap selected by g (see amx_common.h step table).
//   MFMA       = v_mfma_f32_16x16x32_{f16,bf16}: K = 32 = 2 taps x 16 input channels.
#include <stdio.h>
#include <stdlib.h>

#include "amx_device.h"

namespace amx {

// -------------------------------------------------------------------------------------------
// Weight packer: fp32 [Cout][Cin][3][3][3] (+ per-channel scale = folded norm gain) -> A fragments.
// Packed layout [cout_group = Cout/(16Q)][chunk = CinPad/16][step 14][q][lane 64][8] of T.
// Row m of MFMA tile q in cout group cg is output channel cg*16Q + (m>>2)*4Q + q*4 + (m&3), so
// that after the MFMA every lane owns 4Q CONSECUTIVE output channels (wide epilogue stores).
// -------------------------------------------------------------------------------------------
// split (strict precision): per cout group the chunk axis is doubled, [Wh chunks | Wl chunks], Wl = the residual of the
// rounding of Wh (conv3d_k3_v2's SPLIT mode).
template <typename T>
__device__ __forceinline__ void pack_weights_range(const float* __restrict__ w, const float* __restrict__ scale,
                                                   T* __restrict__ wpk, int CinReal, int CinPad, int Cout, int Q, int mode, int CoutReal,
                                                   int split, int CinStride, int C0Real, int C0Phys, long long first, long long stride) {
  const int nchunk_phys = CinPad / 16 * (split ? 2 : 1);
  const long long total = (long long)(Cout / 16) * nchunk_phys * kSteps * 64 * 8;
  const int nchunk = CinPad / 16;
  for (long long idx = first; idx < total; idx += stride) {
    const int e = idx & 7;
    const int lane = (idx >> 3) & 63;
    long long r = idx >> 9;
    const int q = r % Q;
    r /= Q;
    const int s = r % kSteps;
    r /= kSteps;
    const int chunk2 = r % nchunk_phys;
    const int cg = r / nchunk_phys;
    const int part = chunk2 >= nchunk;
    const int chunk = chunk2 - part * nchunk;
    const int m = lane & 15, g = lane >> 4;
    const int cout = cg * 16 * Q + (m >> 2) * 4 * Q + q * 4 + (m & 3);
    const int tap = (g >> 1) ? tapB_index(s) : tapA_index(s);
    // physical input channel -> channel of the weight tensor.  C0Phys > C0Real: the first (skip) segment of a concat input is
    // stored padded to a multiple of 16 channels (ngf = 24: 24 -> 32), the second segment starts behind the padding.
    const int cphys = chunk * 16 + (g & 1) * 8 + e;
    int cin = cphys;
    if (C0Phys > C0Real) cin = cphys < C0Phys ? (cphys < C0Real ? cphys : CinReal) : C0Real + (cphys - C0Phys);
    float v = 0.f;
    if (tap >= 0 && cin < CinReal && (mode != 0 || CoutReal <= 0 || cout < CoutReal)) {    // (padded output channels: zero rows)
      if (mode == 0) {
        v = w[((long long)cout * CinStride + cin) * 27 + tap];   // CinStride > CinReal: the leading channels of a wider tensor
      } else if (cout < CoutReal) {
        // data-gradient weights straight from the forward tensor w[CinReal][CoutReal][27]: taps flipped, channels transposed
        v = w[((long long)cin * CoutReal + cout) * 27 + (26 - tap)];
      }
      if (scale) v *= scale[cout];
    }
    wpk[idx] = part ? (T)(v - (float)(T)v) : (T)v;
  }
}

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                    T* __restrict__ wpk, int CinReal, int CinPad, int Cout, int Q, int mode, int CoutReal,
                                    int split, int CinStride, int C0Real, int C0Phys) {
  pack_weights_range<T>(w, scale, wpk, CinReal, CinPad, Cout, Q, mode, CoutReal, split, CinStride, C0Real, C0Phys,
                        blockIdx.x * (long long)blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

// Many layers in ONE launch (the training step packs every conv's forward weights, then every data-gradient packing, once per step:
// 40 launches of 5-14 us each on the critical path became two).  The descriptors travel in the kernel arguments; a block finds its
// layer from the block prefix table.
struct PackBatch {
  enum { kMax = 40 };
  int count;
  int first_block[kMax + 1];
  const float* w[kMax];
  void* wpk[kMax];
  int CinReal[kMax], CinPad[kMax], Cout[kMax], Q[kMax], mode[kMax], CoutReal[kMax];
};
template <typename T>
__global__ void pack_weights_batch_kernel(const PackBatch b) {
  int l = 0;
  while (l + 1 < b.count && (int)blockIdx.x >= b.first_block[l + 1]) ++l;
  const int nb = b.first_block[l + 1] - b.first_block[l];
  pack_weights_range<T>(b.w[l], nullptr, (T*)b.wpk[l], b.CinReal[l], b.CinPad[l], b.Cout[l], b.Q[l], b.mode[l], b.CoutReal[l], 0,
                        b.CinReal[l], 0, 0, (blockIdx.x - b.first_block[l]) * (long long)blockDim.x + threadIdx.x,
                        (long long)nb * blockDim.x);
}

// ---- AMX_PREC_F16X2_MX: the correction operands of a layer as fp8 A fragments of v_mfma_scale_f32_16x16x128_f8f6f4 -------------
// mxs[1] <- max |w * scale| of the layer (bit pattern of a non-negative float: atomicMax on the unsigned image is monotonic).
__global__ void mx_absmax_kernel(const float* __restrict__ w, const float* __restrict__ scale, long long per_cout, long long total,
                                 int* __restrict__ mxs) {
  float m = 0.f;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    float v = w[idx];
    if (scale) v *= scale[idx / per_cout];
    v = __builtin_fabsf(v);
    m = v > m ? v : m;            // (a NaN weight never wins: the packed values carry it anyway)
  }
  for (int o = 32; o; o >>= 1) {
    const float t = __shfl_xor(m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned*)(mxs + 1), __builtin_bit_cast(unsigned, m));
}
// Sw = floor(log2(448 / max)): the largest weight lands in e4m3's top binade
__device__ __forceinline__ int mx_weight_shift(const int* mxs) {
  const float mx = __builtin_bit_cast(float, (unsigned)mxs[1]);
  if (!(mx > 0.f) || !(mx < 3.0e38f)) return 0;
  int e;
  const float m = __builtin_frexpf(mx, &e);                  // mx = m 2^e, m in [0.5, 1)
  int sw = (m > 0.875f ? 8 : 9) - e;
  return sw < -100 ? -100 : (sw > 100 ? 100 : sw);
}
// Second half of a cout group's packing: [chunk][step j 0..6][q][r 0..1][lane 64][16 bytes].  MFMA row m / tile q -> output channel
// as in pack_weights_kernel; lane group g = lane >> 4: taps 4j + 2 (g >> 1) + r (r = the two 16-byte halves of the lane's 32 K bytes;
// tap 27 = zero); g & 1 = 0: Wh8 (meets xl8), 1: Wl8 (meets xh8); byte e = input channel chunk * 16 + e.
__global__ void pack_weights_mx_kernel(const float* __restrict__ w, const float* __restrict__ scale, unsigned char* __restrict__ wpk,
                                       int CinReal, int CinPad, int Cout, int Q, int CoutReal, int CinStride, int C0Real, int C0Phys,
                                       int* __restrict__ mxs) {
  const int nchunk = CinPad / 16;
  const int sw = mx_weight_shift(mxs);
  if (blockIdx.x == 0 && threadIdx.x == 0) mxs[0] = ((127 - (sw + 11)) & 255) * 0x01010101;     // E8M0 byte of 2^-(Sw + 11)
  const float sh = __builtin_ldexpf(1.f, sw), sl = __builtin_ldexpf(1.f, sw + 11);
  const long long total = (long long)(Cout / 16) * nchunk * 7 * 2 * 64 * 4;                      // one thread = 4 bytes
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int e4 = idx & 3;
    const int lane = (idx >> 2) & 63;
    long long r = idx >> 8;
    const int half = r & 1;
    r >>= 1;
    const int q = r % Q;
    r /= Q;
    const int j = r % 7;
    r /= 7;
    const int chunk = r % nchunk;
    const int cg = r / nchunk;
    const int m = lane & 15, g = lane >> 4;
    const int cout = cg * 16 * Q + (m >> 2) * 4 * Q + q * 4 + (m & 3);
    const int tap = 4 * j + 2 * (g >> 1) + half;
    float v4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cphys = chunk * 16 + e4 * 4 + k;
      int cin = cphys;
      if (C0Phys > C0Real) cin = cphys < C0Phys ? (cphys < C0Real ? cphys : CinReal) : C0Real + (cphys - C0Phys);
      float v = 0.f;
      if (tap < 27 && cin < CinReal && (CoutReal <= 0 || cout < CoutReal)) {
        v = w[((long long)cout * CinStride + cin) * 27 + tap];
        if (scale) v *= scale[cout];
      }
      const float hi = (float)(f16)v;
      v4[k] = (g & 1) ? (v - hi) * sl : hi * sh;
    }
    // position inside the cout group: Wh chunks first (nchunk * 14 Q KiB), then these
    const long long off = ((long long)cg * 2 * nchunk + nchunk + chunk) * (kSteps * Q * 1024) + ((long long)(j * Q + q) * 2 + half) * 1024 +
                          lane * 16 + e4 * 4;
    *(unsigned*)(wpk + off) = e4m3_pk4(v4[0], v4[1], v4[2], v4[3]);
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
// FMT: voxel layout 0 / 1 / 2 (amx_common.h); FMT 2 also writes the e4m3 copies of the pooled values (AMX_PREC_F16X2_MX).
template <typename T, int AVG, int FMT>
__global__ void pool2_kernel(const char* __restrict__ in, char* __restrict__ out, int N, int Do,
                             int Ho, int Wo, int C, int skip_lo) {
  constexpr bool SPLIT = FMT == 1 || FMT == 2, PLANAR = FMT >= 2;      // FMT 3: single values, row-planar (amx_common.h)
  const int c8n = C >> 3;
  const long long total = (long long)N * Do * Ho * Wo * c8n;
  // FMT 2 (row-planar, amx_common.h): 32 bytes per voxel inside plane c8 >> 1 of its row; rows are C * 6 * W bytes in every layout
  constexpr int EB = FMT == 3 ? 2 : fmt_elem_bytes(FMT);
  const long long sx = PLANAR ? 32 : (long long)C * EB, sy = (long long)C * EB * (Wo * 2), sz = sy * (Ho * 2);
  const long long lo_in = FMT == 2 ? 2ll * C * (Wo * 2) : 2 * C, lo_out = FMT == 2 ? 2ll * C * Wo : 2 * C;
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
    const char* base = in + (long long)n * sz * (Do * 2) + (2 * z) * sz + (2 * y) * sy + (2 * x) * sx +
                       (PLANAR ? (long long)(c8 >> 1) * (Wo * 2 * 32) + (c8 & 1) * 16 : c8 * 16);
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const char* vp = base + (k >> 2) * sz + ((k >> 1) & 1) * sy + (k & 1) * sx;
      const uint4 raw = *(const uint4*)vp;
      const unsigned wv[4] = {raw.x, raw.y, raw.z, raw.w};
      uint4 rawl = make_uint4(0, 0, 0, 0);
      if (SPLIT) rawl = *(const uint4*)(vp + lo_in);
      const unsigned wl[4] = {rawl.x, rawl.y, rawl.z, rawl.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned short bits = (unsigned short)(wv[e >> 1] >> ((e & 1) * 16));
        float f = (float)__builtin_bit_cast(T, bits);
        if (SPLIT) f += (float)__builtin_bit_cast(T, (unsigned short)(wl[e >> 1] >> ((e & 1) * 16)));   // exact: hi + lo fits fp32
        if (k == 0) m[e] = f;
        else m[e] = AVG ? m[e] + f : (f > m[e] ? f : m[e]);
      }
    }
    unsigned o[4], ol[4];
    if (AVG) {
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] *= 0.125f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = m[2 * e];
      const float bq = m[2 * e + 1];
      o[e] = (unsigned)to_bits<T>(a) | ((unsigned)to_bits<T>(bq) << 16);
      if (SPLIT) ol[e] = (unsigned)to_bits<T>(a - (float)(T)a) | ((unsigned)to_bits<T>(bq - (float)(T)bq) << 16);
    }
    if (SPLIT) {
      const long long vlin = idx / c8n;                     // output voxel; FMT 2: row vlin / Wo, plane c8 >> 1
      char* op = FMT == 2 ? out + (vlin / Wo) * (6ll * C * Wo) + (long long)(c8 >> 1) * (Wo * 32) + x * 32 + (c8 & 1) * 16
                          : out + vlin * (long long)(C * fmt_elem_bytes(FMT)) + c8 * 16;
      *(uint4*)op = make_uint4(o[0], o[1], o[2], o[3]);
      if (!(FMT == 2 && skip_lo)) *(uint4*)(op + lo_out) = make_uint4(ol[0], ol[1], ol[2], ol[3]);   // (conv-only readers: hi + copies)
      if (FMT == 2) mx_store_copies(op - (c8 & 1) * 16 + 2 * lo_out, c8 & 1, m);
    } else if (FMT == 3) {
      const long long vlin = idx / c8n;
      *(uint4*)(out + (vlin / Wo) * (2ll * C * Wo) + (long long)(c8 >> 1) * (Wo * 32) + x * 32 + (c8 & 1) * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
      *(uint4*)(out + idx * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

// -------------------------------------------------------------------------------------------
// host-side launchers
// -------------------------------------------------------------------------------------------
static thread_local char g_kernel_name[64] = "";
const char* last_conv_kernel_name() { return g_kernel_name; }

// Tile-shape heuristic.  Returns the Q (16-channel MFMA tiles per workgroup) a layer will be
// launched with; the packed-weight layout depends on it.
int conv_pick_q(int Cout, int W, int precision) {
  if (Cout == 16) return 1;
  if (Cout == 32) return 2;
  int q = W >= 32 ? 4 : 2;    // Cout >= 64 at 32^3 and larger: 4; 16^3 and below: more cout groups to fill the chip
  // f16x2mx keeps Q = 2 (brick 4x4x32) at every width: its sweep holds the f16 AND the fp8 fragments, and four column tiles left
  // the 64^3 / 32^3 layers of anatomix-dev at 310-430 TF against 400-550 with two (batch 4, same box: 64 -> 64 @64^3 629 -> 482 us,
  // 192 -> 64 @64^3 1664 -> 1264, 384 -> 128 @32^3 809 -> 631; Q = 1: 536 / 1479 / 733).  The single 16-bit and the bf16x2 / f16x2
  // kernels measured 0-7 % SLOWER with two (6 M forward 64 -> 64 @32^3 45.5 -> 46.9 us; strict 192 -> 64 @64^3 2045 -> 2197).
  if (precision == 4) q = 2;
  // experiment switches (-DAMX_EXPERIMENT builds only): read ONCE -- the packing at create time and the launch must see the same Q --
  // and only the values the kernels are instantiated for
  static const int q_wide = [] { const char* e = exp_env("AMX_Q_WIDE"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
  static const int q_deep = [] { const char* e = exp_env("AMX_Q_DEEP"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
  if (W >= 32 && q_wide) q = q_wide;
  if (W <= 8 && q_deep) q = q_deep;
  // (round 4, same box, batch 4: Q = 1 or 4 instead of 2 at the 8^3 level: 128 -> 256 22.2 -> 22.8 / 27.9 us, 256 -> 256 32.8 -> 34.7 / 44.9;
  //  at the 16^3 level Q = 1 / 4: 128 -> 128 28.1 -> 52.1 / 39.2 us -- more or fewer cout groups do not help: profiles/r04_deep_level_q.txt)
  while (Cout % (16 * q)) q >>= 1;   // 48 / 96 / 192 output channels (data gradients of the concat convs): 1 / 2 / 4
  return q;
}

hipError_t launch_conv_v2(const ConvParams& p, int precision, int Q, hipStream_t st);
const char* last_conv_v2_kernel_name();
int last_conv_v2_stats_slots();
bool conv_ks_eligible(const ConvParams& p, int precision, int Q);
hipError_t launch_conv_ks(const ConvParams& p, int precision, int Q, hipStream_t st);
const char* last_conv_ks_kernel_name();
bool conv_zmarch_eligible(const ConvParams& p);
bool conv_zmarch_eligible_split(const ConvParams& p);
hipError_t launch_conv_zmarch(const ConvParams& p, int precision, hipStream_t st);
const char* last_conv_zm_kernel_name();

// true when launch_conv runs the generic kernel for this layer -- the one whose epilogue can write InstanceNorm partial sums
bool conv_fuses_stats(const ConvParams& p, int precision, int Q) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_FUSED_STATS") ? 1 : 0;
  if (off || p.src0_f32c1 || p.out32) return false;
  return !((((precision < 2 && conv_zmarch_eligible(p)) || ((precision == 2 || precision == 3) && conv_zmarch_eligible_split(p))) && Q == p.Cout / 16));
}
int last_conv_stats_slots() { return last_conv_v2_stats_slots(); }

hipError_t launch_conv(const ConvParams& p, int precision, int Q, hipStream_t st) {
  const bool planar = p.out32 != nullptr;
  if (((precision < 2 && conv_zmarch_eligible(p)) || ((precision == 2 || precision == 3) && conv_zmarch_eligible_split(p))) && Q == p.Cout / 16) {
    // narrow full/half-resolution layers: z-marching ring kernel
    hipError_t e = launch_conv_zmarch(p, precision, st);
    snprintf(g_kernel_name, sizeof g_kernel_name, "%s", last_conv_zm_kernel_name());
    return e;
  }
  if (p.src0_f32c1) return hipErrorInvalidValue;   // the fp32 stem has its own kernel (amx_conv3d_stem.hip)
  if (p.raw_halo) return hipErrorInvalidValue;     // frame-reading sources: the z-march kernels only
  (void)planar;
  if (conv_ks_eligible(p, precision, Q)) {           // deep levels: register-stationary weights, K split over the waves
    hipError_t e = launch_conv_ks(p, precision, Q, st);
    snprintf(g_kernel_name, sizeof g_kernel_name, "%s", last_conv_ks_kernel_name());
    return e;
  }
  hipError_t e = launch_conv_v2(p, precision, Q, st);   // generic path: persistent double-buffered DMA kernel
  snprintf(g_kernel_name, sizeof g_kernel_name, "%s", last_conv_v2_kernel_name());
  return e;
}

hipError_t launch_pack_weights(const float* w, const float* scale, void* wpk, int CinReal, int CinPad,
                               int Cout, int Q, int precision, hipStream_t st, int mode, int CoutReal, int CinStride, int C0Real,
                               int C0Phys) {
  if (CinStride <= 0) CinStride = CinReal;
  if (precision == 4) return hipErrorInvalidValue;     // launch_pack_weights_mx
  const int split = precision >= 2;
  const long long total = (long long)(Cout / 16) * (CinPad / 16) * kSteps * 64 * 8 * (split ? 2 : 1);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if ((precision & 1) == 0)
    hipLaunchKernelGGL(pack_weights_kernel<f16>, dim3(blocks), dim3(256), 0, st, w, scale, (f16*)wpk,
                       CinReal, CinPad, Cout, Q, mode, CoutReal, split, CinStride, C0Real, C0Phys);
  else
    hipLaunchKernelGGL(pack_weights_kernel<bf16>, dim3(blocks), dim3(256), 0, st, w, scale,
                       (bf16*)wpk, CinReal, CinPad, Cout, Q, mode, CoutReal, split, CinStride, C0Real, C0Phys);
  return hipGetLastError();
}

// Batched packing of plain (un-split, un-scaled) layers: w[i] fp32, CinStride = CinReal; single 16-bit precisions only.
hipError_t launch_pack_weights_batch(int count, const float* const* w, void* const* wpk, const int* CinReal, const int* CinPad, const int* Cout,
                                     const int* Q, const int* mode, const int* CoutReal, int precision, hipStream_t st) {
  if (precision > 1) return hipErrorInvalidValue;
  for (int i0 = 0; i0 < count; i0 += PackBatch::kMax) {
    PackBatch b;
    b.count = count - i0 < PackBatch::kMax ? count - i0 : PackBatch::kMax;
    int blocks = 0;
    for (int i = 0; i < b.count; ++i) {
      const int j = i0 + i;
      const long long total = (long long)(Cout[j] / 16) * (CinPad[j] / 16) * kSteps * 64 * 8;
      long long nb = (total + 255) / 256;
      if (nb > 512) nb = 512;
      b.first_block[i] = blocks;
      blocks += (int)nb;
      b.w[i] = w[j]; b.wpk[i] = wpk[j];
      b.CinReal[i] = CinReal[j]; b.CinPad[i] = CinPad[j]; b.Cout[i] = Cout[j]; b.Q[i] = Q[j]; b.mode[i] = mode[j]; b.CoutReal[i] = CoutReal[j];
    }
    b.first_block[b.count] = blocks;
    if ((precision & 1) == 0) hipLaunchKernelGGL(pack_weights_batch_kernel<f16>, dim3(blocks), dim3(256), 0, st, b);
    else hipLaunchKernelGGL(pack_weights_batch_kernel<bf16>, dim3(blocks), dim3(256), 0, st, b);
  }
  return hipGetLastError();
}

// AMX_PREC_F16X2_MX: [Wh f16 chunks | fp8 correction chunks] per cout group (same bytes as the [Wh | Wl] packing of the strict
// precisions) + the layer's block-scale word mxs[0] (mxs[1]: scratch for the weight maximum).  Forward weights only.
hipError_t launch_pack_weights_mx(const float* w, const float* scale, void* wpk, int* mxs, int CinReal, int CinPad, int Cout, int Q,
                                  hipStream_t st, int CoutReal, int CinStride, int C0Real, int C0Phys) {
  if (CinStride <= 0) CinStride = CinReal;
  hipError_t e = launch_pack_weights(w, scale, wpk, CinReal, CinPad, Cout, Q, 2, st, 0, CoutReal, CinStride, C0Real, C0Phys);   // Wh (| Wl, overwritten below)
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(mxs, 0, 2 * sizeof(int), st);
  if (e != hipSuccess) return e;
  const int couts = CoutReal > 0 ? CoutReal : Cout;
  const long long per_cout = (long long)CinStride * 27, total_w = per_cout * couts;
  hipLaunchKernelGGL(mx_absmax_kernel, dim3((unsigned)((total_w + 255) / 256 > 1024 ? 1024 : (total_w + 255) / 256)), dim3(256), 0, st, w,
                     scale, per_cout, total_w, mxs);
  const long long total = (long long)(Cout / 16) * (CinPad / 16) * 7 * 2 * 64 * 4;
  hipLaunchKernelGGL(pack_weights_mx_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, st, w,
                     scale, (unsigned char*)wpk, CinReal, CinPad, Cout, Q, CoutReal, CinStride, C0Real, C0Phys, mxs);
  return hipGetLastError();
}

hipError_t launch_fold_norm(const float* gamma, const float* beta, const float* mean, const float* var,
                            const float* conv_bias, float eps, int C, float* scale, float* shift,
                            hipStream_t st) {
  hipLaunchKernelGGL(fold_norm_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, mean, var,
                     conv_bias, eps, C, scale, shift);
  return hipGetLastError();
}

// planar (precisions 0 / 1 only): input AND output are row-planar (layout FMT 3)
hipError_t launch_pool2(const void* in, void* out, int N, int Do, int Ho, int Wo, int C, int avg,
                        int precision, hipStream_t st, int skip_lo, int planar) {
  if (planar && (precision > 1 || C % 16)) return hipErrorInvalidValue;
  const long long total = (long long)N * Do * Ho * Wo * (C / 8);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
#define AMX_POOL(T, A, S)                                                                            \
  hipLaunchKernelGGL((pool2_kernel<T, A, S>), dim3(blocks), dim3(256), 0, st, (const char*)in, (char*)out, \
                     N, Do, Ho, Wo, C, skip_lo)
  switch (precision + (planar ? 10 : 0)) {
    case 10: if (avg) AMX_POOL(f16, 1, 3); else AMX_POOL(f16, 0, 3); break;
    case 11: if (avg) AMX_POOL(bf16, 1, 3); else AMX_POOL(bf16, 0, 3); break;
    case 0: if (avg) AMX_POOL(f16, 1, false); else AMX_POOL(f16, 0, false); break;
    case 1: if (avg) AMX_POOL(bf16, 1, false); else AMX_POOL(bf16, 0, false); break;
    case 2: if (avg) AMX_POOL(f16, 1, true); else AMX_POOL(f16, 0, true); break;
    case 3: if (avg) AMX_POOL(bf16, 1, true); else AMX_POOL(bf16, 0, true); break;
    case 4: if (avg) AMX_POOL(f16, 1, 2); else AMX_POOL(f16, 0, 2); break;
    default: return hipErrorInvalidValue;
  }
#undef AMX_POOL
  return hipGetLastError();
}

}  // namespace amx
