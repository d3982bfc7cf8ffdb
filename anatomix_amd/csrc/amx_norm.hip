// anatomix_amd -- bandwidth kernels of the InstanceNorm / trilinear variant (`anatomix-dev`:
// norm='instance', interp='trilinear', pooling='Avg'; anatomix/model/load_from_hf.py:18-24).
//
//   nn.InstanceNorm3d(affine=False|True, eps)   network.py:157-163
//       per (n, c): y = (x - mean) / sqrt(var_biased + eps) [* gamma + beta], followed by the activation.
//       Two-pass, deterministic: (1) every block reduces a slab of voxels to per-channel shifted sums
//       S1 = sum(x - K), S2 = sum((x - K)^2) in fp32 (K = the channel's first voxel, which removes the
//       cancellation of E[x^2] - E[x]^2) and writes them to its own slot; (2) one block per sample
//       adds the slots in fixed order and emits the affine pair (a, b) with y = a*x + b;
//       (3) an elementwise pass applies a*x + b and the activation in place.
//   nn.Upsample(scale_factor=2, mode='trilinear')   network.py:407 (align_corners=False)
//       per axis out[2i] = .25*in[i-1] + .75*in[i], out[2i+1] = .75*in[i] + .25*in[i+1] with the
//       neighbour index clamped (PyTorch's source-index clamp puts the whole weight on the edge voxel).
// All tensors are channels-last 16-bit; one thread moves 8 channels (16 B) of one voxel.
#include "amx_device.h"

namespace amx {

template <typename T>
__device__ __forceinline__ void unpack8(const uint4& raw, float (&f)[8]) {
  const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)__builtin_bit_cast(T, (unsigned short)(w[e >> 1] >> ((e & 1) * 16)));
}
template <typename T>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  unsigned o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_bits<T>(f[2 * e]) | ((unsigned)to_bits<T>(f[2 * e + 1]) << 16);
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// Voxel layouts FMT 0 / 1 / 2 (amx_common.h).  Strict precision (FMT >= 1): a voxel holds [hi(C) | lo(C)] 16-bit channels, value =
// hi + lo.  load8 / store8 move one 8-channel group of one voxel: `p` points at the group's hi half, the lo half sits `lo_off` =
// 2*C bytes further on.  FMT 2 (AMX_PREC_F16X2_MX): store8 also writes the e4m3 copies the convolutions multiply with -- group c8
// of the voxel: xl8 at voxel + 4C + 32 (c8 >> 1) + 8 (c8 & 1), xh8 16 bytes behind it; nothing reads them but the conv kernels.
template <int FMT> struct Fmt {
  static constexpr bool SPLIT = FMT == 1 || FMT == 2;
  static constexpr int M = (FMT == 0 || FMT == 3) ? 1 : (FMT == 1 ? 2 : 3);      // voxel bytes = 2 C M
  static constexpr bool PLANAR = FMT >= 2;                          // FMT 3: single values, row-planar (amx_common.h)
  // byte offset of 8-channel group c8 (its hi half) of voxel v -- a linear index whose rows are W voxels long -- and the distance to
  // the lo half.  FMT 2 is row-planar (amx_common.h): plane c8 >> 1 of row v / W, 32 bytes per voxel.
  static __device__ __forceinline__ long long group(long long v, int C, int W, int c8) {
    if (FMT < 2) return v * (2 * C * M) + c8 * 16;
    const long long row = v / W;
    const int x = (int)(v - row * W);
    return row * (2ll * M * C * W) + (long long)(c8 >> 1) * (W * 32) + x * 32 + (c8 & 1) * 16;
  }
  static __device__ __forceinline__ long long group_rx(long long row, int x, int C, int W, int c8) {     // the same from (row, x)
    if (FMT < 2) return (row * W + x) * (2ll * C * M) + c8 * 16;
    return row * (2ll * M * C * W) + (long long)(c8 >> 1) * (W * 32) + x * 32 + (c8 & 1) * 16;
  }
  static __device__ __forceinline__ int lo_off(int C, int W) { return FMT < 2 ? 2 * C : 2 * C * W; }
};
template <typename T, int FMT>
__device__ __forceinline__ void load8(const char* p, int lo_off, float (&f)[8]) {
  unpack8<T>(*(const uint4*)p, f);
  if (FMT == 1 || FMT == 2) {
    float g[8];
    unpack8<T>(*(const uint4*)(p + lo_off), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += g[e];
  }
}
template <typename T, int FMT>
// skip_lo (FMT 2, wave-uniform): the tensor's only readers are convolutions, which read hi and the copies -- the lo plane is not written
__device__ __forceinline__ void store8(char* p, int lo_off, const float (&f)[8], int c8 = 0, bool skip_lo = false) {
  *(uint4*)p = pack8<T>(f);
  if ((FMT == 1 || FMT == 2) && !(FMT == 2 && skip_lo)) {
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = f[e] - (float)(T)f[e];
    *(uint4*)(p + lo_off) = pack8<T>(r);
  }
  if (FMT == 2) mx_store_copies(p - (c8 & 1) * 16 + 2 * lo_off, c8 & 1, f);   // the chunk's plane of copies; with the lane of group c8 ^ 1
}

// slabs per sample in the statistics pass: enough blocks to fill the chip on the big planes, >= 8 loads per
// thread on the small ones, and nblk * C <= 65536 so the partial-sum scratch stays at 512 KiB per sample
__host__ __device__ inline int in_num_blocks(long long vox, int C) {
  long long nb = vox * C / (8 * 256 * 8);
  if (nb > 1024) nb = 1024;
  if (nb * C > 65536) nb = 65536 / C;
  return nb < 1 ? 1 : (int)nb;
}

// grid (kInBlocks, N), block 256.  partial[n][blk][c][2]
template <typename T, int FMT>
__global__ __launch_bounds__(256) void in_stats_kernel(const char* __restrict__ x, float* __restrict__ partial, long long vox,
                                                      int C, int W) {
  constexpr int M = Fmt<FMT>::M;
  extern __shared__ float red[];                       // [256/c8n rows][C][2]
  const int c8n = C >> 3;
  const int n = blockIdx.y, blk = blockIdx.x;
  const int c8 = threadIdx.x % c8n, vrow = threadIdx.x / c8n, nrow = 256 / c8n;
  const char* xs = x + (long long)n * vox * C * 2 * M;
  float K[8];
  unpack8<T>(*(const uint4*)(xs + Fmt<FMT>::group(0, C, W, c8)), K);       // the channel's first voxel (its hi half): shift of the sums
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long per = (vox + gridDim.x - 1) / gridDim.x;
  const long long v0 = (long long)blk * per, v1 = v0 + per < vox ? v0 + per : vox;
  if (vrow < nrow)
    for (long long v = v0 + vrow; v < v1; v += nrow) {
      float f[8];
      load8<T, FMT>(xs + Fmt<FMT>::group(v, C, W, c8), Fmt<FMT>::lo_off(C, W), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = f[e] - K[e];
        s1[e] += d;
        s2[e] += d * d;
      }
    }
  // block reduction over the voxel rows (fixed order)
  if (vrow < nrow)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[((vrow * C) + c8 * 8 + e) * 2] = s1[e];
      red[((vrow * C) + c8 * 8 + e) * 2 + 1] = s2[e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < nrow; ++r) {
      a += red[(r * C + c) * 2];
      b += red[(r * C + c) * 2 + 1];
    }
    float* o = partial + (((long long)n * gridDim.x + blk) * C + c) * 2;
    o[0] = a;
    o[1] = b;
  }
}

// grid (C/8, N), block 256 = 8 channels x 32 partial lanes.  ab[n][c][2] = (a, b) with y = a*x + b.
// Lane l adds partials l, l+32, ... in order, then the 32 lane sums are added in lane order: fixed order, no atomics.
template <typename T, int FMT>
__global__ __launch_bounds__(256) void in_finalize_kernel(const char* __restrict__ x, const float* __restrict__ partial,
                                                         const float* gamma, const float* beta, float eps, long long vox,
                                                         int C, int nblk, float* __restrict__ ab, const float* kshift, int use_kshift, int W) {
  __shared__ double red[2][8][32];
  const int n = blockIdx.y, c = blockIdx.x * 8 + (threadIdx.x & 7), l = threadIdx.x >> 3;
  double s1 = 0.0, s2 = 0.0;
  for (int b = l; b < nblk; b += 32) {
    const float* q = partial + (((long long)n * nblk + b) * C + c) * 2;
    s1 += q[0];
    s2 += q[1];
  }
  red[0][threadIdx.x & 7][l] = s1;
  red[1][threadIdx.x & 7][l] = s2;
  __syncthreads();
  if (threadIdx.x < 8) {
    const int cc = blockIdx.x * 8 + threadIdx.x;
    s1 = 0.0; s2 = 0.0;
    for (int k = 0; k < 32; ++k) {
      s1 += red[0][threadIdx.x][k];
      s2 += red[1][threadIdx.x][k];
    }
    // the shift the partial sums were taken about: the channel's first voxel (in_stats), or -- sums written by a conv epilogue --
    // the conv bias (zero without one)
    float K;
    if (use_kshift) K = kshift ? kshift[cc] : 0.f;
    else {
      const unsigned short kb = *(const unsigned short*)(x + (long long)n * vox * C * Fmt<FMT>::M * 2 + Fmt<FMT>::group(0, C, W, cc >> 3) + (cc & 7) * 2);
      K = (float)__builtin_bit_cast(T, kb);
    }
    const double m1 = s1 / (double)vox;
    double var = s2 / (double)vox - m1 * m1;              // biased variance, shift invariant
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mean = K + (float)m1;
    const float g = gamma ? gamma[cc] : 1.f;
    ab[((long long)n * C + cc) * 2] = rstd * g;
    ab[((long long)n * C + cc) * 2 + 1] = (beta ? beta[cc] : 0.f) - mean * rstd * g;
  }
}

// Many slots (a conv epilogue writes one per brick and wave: 32768 per sample on a 128^3 level): grid (C/8, N, kPreChunks) folds
// the slots of one chunk, in order, into partial2[n][chunk][c][2]; in_finalize then adds the kPreChunks chunks.  Fixed order.
constexpr int kPreChunks = 64;
__global__ __launch_bounds__(256) void in_prereduce_kernel(const float* __restrict__ partial, float* __restrict__ partial2, int C,
                                                          int nblk) {
  __shared__ float red[2][8][32];
  const int n = blockIdx.y, c = blockIdx.x * 8 + (threadIdx.x & 7), l = threadIdx.x >> 3, ch = blockIdx.z;
  const int per = (nblk + kPreChunks - 1) / kPreChunks, b0 = ch * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  float s1 = 0.f, s2 = 0.f;
  for (int b = b0 + l; b < b1; b += 32) {
    const float* q = partial + (((long long)n * nblk + b) * C + c) * 2;
    s1 += q[0];
    s2 += q[1];
  }
  red[0][threadIdx.x & 7][l] = s1;
  red[1][threadIdx.x & 7][l] = s2;
  __syncthreads();
  if (threadIdx.x < 8) {
    s1 = 0.f; s2 = 0.f;
    for (int k = 0; k < 32; ++k) {
      s1 += red[0][threadIdx.x][k];
      s2 += red[1][threadIdx.x][k];
    }
    float* o = partial2 + (((long long)n * kPreChunks + ch) * C + blockIdx.x * 8 + threadIdx.x) * 2;
    o[0] = s1;
    o[1] = s2;
  }
}

template <typename T, int FMT>
__global__ void in_apply_kernel(char* __restrict__ x, const float* __restrict__ ab, long long vox, int C, int N, int act,
                                float slope, int* oflow, int W, int skip_lo) {
  bool bad = false;
  const int c8n = C >> 3;
  const long long total = (long long)N * vox * c8n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = idx % c8n;
    const long long nv = idx / c8n;
    const int n = nv / vox;
    float f[8];
    char* xp = x + Fmt<FMT>::group(nv, C, W, c8);          // (a sample is a whole number of rows: nv / W is the global row)
    load8<T, FMT>(xp, Fmt<FMT>::lo_off(C, W), f);
    const float* q = ab + ((long long)n * C + c8 * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = f[e] * q[2 * e] + q[2 * e + 1];
      v = act_fwd(v, act_k(act, slope));
      if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(v);   // the value about to be stored
      f[e] = v;
    }
    store8<T, FMT>(xp, Fmt<FMT>::lo_off(C, W), f, c8, skip_lo != 0);
  }
  if (RangeCheck<T>::on) raise_flag(oflow, bad);
}

// Same pass when 256 % (C / 8) == 0 (every power-of-two channel count): gridDim.y = sample, a thread keeps ONE 8-channel
// group for its whole grid-stride walk (the stride is a multiple of C / 8), so its 16 coefficients sit in registers and the
// loop has no 64-bit division -- the generic kernel above spends more time on index arithmetic and coefficient loads than on
// memory (3.2 vs 4.6 TB/s on the 268 MB level-0 tensors of anatomix-dev).
template <typename T, int FMT>
__global__ __launch_bounds__(256) void in_apply_fast_kernel(char* __restrict__ x, const float* __restrict__ ab, long long vox,
                                                            int C, int act, float slope, int* oflow, int W, int skip_lo) {
  bool bad = false;
  const int c8n = C >> 3, c8 = threadIdx.x % c8n, n = blockIdx.y;
  const float* q = ab + ((long long)n * C + c8 * 8) * 2;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = q[2 * e];
    b[e] = q[2 * e + 1];
  }
  const long long total = vox * c8n;
  constexpr bool SPLIT = Fmt<FMT>::SPLIT;
  char* xs = x + (long long)n * total * 16 * Fmt<FMT>::M;
  // SPLIT: a thread's group index c8 is fixed, its voxel advances by (gridDim.x * 256) / c8n per iteration
  const long long vstep = (long long)gridDim.x * 256 / c8n;
  long long vv = (blockIdx.x * 256ll + threadIdx.x) / c8n;
  // FMT 2 (row-planar): the thread's voxel as (row, x), advanced by increment-and-carry -- no division in the loop
  long long row = FMT == 2 ? vv / W : 0;
  int xr = FMT == 2 ? (int)(vv - row * W) : 0;
  const long long rstep = FMT == 2 ? vstep / W : 0;
  const int xstep = FMT == 2 ? (int)(vstep - rstep * W) : 0;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256, vv += vstep) {
    float f[8];
    char* xp = FMT == 2 ? xs + Fmt<FMT>::group_rx(row, xr, C, W, c8) : (SPLIT ? xs + vv * (C * 2 * Fmt<FMT>::M) + c8 * 16 : xs + idx * 16);
    if (FMT == 2) {
      row += rstep;
      xr += xstep;
      if (xr >= W) { xr -= W; ++row; }
    }
    load8<T, FMT>(xp, Fmt<FMT>::lo_off(C, W), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = f[e] * a[e] + b[e];
      v = act_fwd(v, act_k(act, slope));
      if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(v);   // the value about to be stored
      f[e] = v;
    }
    store8<T, FMT>(xp, Fmt<FMT>::lo_off(C, W), f, c8, skip_lo != 0);
  }
  if (RangeCheck<T>::on) raise_flag(oflow, bad);
}

// f16x2mx (row-planar, FMT 2): the apply pass of a layer whose output is pooled next -- y = act(a x + b) in place AND the 2x2x2
// pooled tensor in one read of the raw values (network.py:368: conv -> norm -> act -> pool; the in-place tensor stays alive as the
// skip connection).  One block per pooled row (n, zp, yp); a thread owns the 8-channel group c8 of full-resolution column xr of the
// four source rows: lane bit 0 = c8 & 1 (the lane pair mx_store_copies exchanges with), lane bit 1 = xr & 1 (the pooling partner
// along x, one quad_perm away), so every load and store of a wavefront is one contiguous 1 KiB run of a plane.
template <typename T>
__global__ __launch_bounds__(256) void in_apply_pool_kernel(char* __restrict__ x, const float* __restrict__ ab, char* __restrict__ pooled,
                                                            int D2, int H2, int W, int C, int act, float slope, int avg, int skip_lo,
                                                            int pool_skip_lo, int* oflow) {
  bool bad = false;
  int r = blockIdx.x;
  const int yp = r % H2;
  r /= H2;
  const int zp = r % D2, n = r / D2;
  const int c8n = C >> 3, W2 = W >> 1;
  const long long rowb = 6ll * C * W, prowb = 6ll * C * W2;
  const int lo = 2 * C * W, plo = 2 * C * W2;
  const float ak = act_k(act, slope);
  for (int t = threadIdx.x; t < W * c8n; t += 256) {
    const int c8 = ((t >> 1) / W) * 2 + (t & 1), xr = (t >> 1) % W;
    const float* q = ab + ((long long)n * C + c8 * 8) * 2;
    float a[8], b[8], acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] = q[2 * e];
      b[e] = q[2 * e + 1];
      acc[e] = avg ? 0.f : -3.0e38f;
    }
    const long long goff = (long long)(c8 >> 1) * (W * 32) + xr * 32 + (c8 & 1) * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      char* xp = x + (((long long)n * 2 * D2 + 2 * zp + (k >> 1)) * 2 * H2 + 2 * yp + (k & 1)) * rowb + goff;
      float f[8];
      load8<T, 2>(xp, lo, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = act_fwd(f[e] * a[e] + b[e], ak);
        if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(v);
        f[e] = v;
        acc[e] = avg ? acc[e] + v : fmaxf(acc[e], v);
      }
      store8<T, 2>(xp, lo, f, c8, skip_lo != 0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[e]), 0x4E, 0xF, 0xF, true));   // lane ^ 2
      acc[e] = avg ? (acc[e] + o) * 0.125f : fmaxf(acc[e], o);
    }
    if (!(xr & 1)) {
      char* pp = pooled + (((long long)n * D2 + zp) * H2 + yp) * prowb + (long long)(c8 >> 1) * (W2 * 32) + (xr >> 1) * 32 + (c8 & 1) * 16;
      store8<T, 2>(pp, plo, acc, c8, pool_skip_lo != 0);
    }
  }
  if (RangeCheck<T>::on) raise_flag(oflow, bad);
}

// out [N][2D][2H][2W][C] <- in [N][D][H][W][C].  Cell formulation: the 2 x 2 x 2 outputs (2z+1..2z+2, 2y+1..2y+2,
// 2x+1..2x+2) all interpolate the SAME eight inputs (z..z+1, y..y+1, x..x+1), so a thread loads those eight 8-channel
// vectors once and writes eight outputs -- cache reads equal the output bytes instead of 8x (the per-output version was
// L1-bound at 2.0 TB/s on the 537 MB level-0 tensor of anatomix-dev).  Cells run from -1 to L-1 per axis with clamped
// inputs, which reproduces the border rows (output 0 and 2L-1) of align_corners=False.  One block per (n, cz, cy).
template <typename T, int FMT>
// ab != null: `in` is a RAW conv output whose InstanceNorm + activation is pending -- act(a x + b) is applied to the eight inputs on
// the way in (the separate apply pass over the low-resolution tensor disappears; nothing else reads that tensor).
__global__ __launch_bounds__(256) void upsample2_trilinear_kernel(const char* __restrict__ in, char* __restrict__ out, int N,
                                                                  int D, int H, int W, int C, int skip_lo,
                                                                  const float* __restrict__ ab, int act, float slope, int* oflow) {
  constexpr int M = Fmt<FMT>::M;
  const int c8n = C >> 3;
  int r = blockIdx.x;
  const int cy = r % (H + 1) - 1;
  r /= H + 1;
  const int cz = r % (D + 1) - 1;
  const int n = r / (D + 1);
  const int z0 = cz < 0 ? 0 : cz, z1 = cz + 1 < D ? cz + 1 : D - 1;
  const int y0 = cy < 0 ? 0 : cy, y1 = cy + 1 < H ? cy + 1 : H - 1;
  const long long rowb = (long long)W * C * 2 * M;
  const char* rows[4] = {in + (((long long)n * D + z0) * H + y0) * rowb, in + (((long long)n * D + z0) * H + y1) * rowb,
                         in + (((long long)n * D + z1) * H + y0) * rowb, in + (((long long)n * D + z1) * H + y1) * rowb};
  const long long orowb = 2 * rowb;                        // bytes of one output row
  for (int t = threadIdx.x; t < (W + 1) * c8n; t += 256) {
    // channels-last: consecutive threads walk the groups of one cell; row-planar (FMT 2): the two groups of a chunk, then the cells
    // along x, then the chunks -- a wavefront then covers 32 consecutive voxels of ONE plane (1 KiB runs instead of 32-byte pieces;
    // the groups-first order ran this pass at 3.2 TB/s against 4.9 in the channels-last strict mode)
    const int c8 = FMT == 2 ? ((t >> 1) / (W + 1)) * 2 + (t & 1) : t % c8n;
    const int cx = (FMT == 2 ? (t >> 1) % (W + 1) : t / c8n) - 1;
    const int x0 = cx < 0 ? 0 : cx, x1 = cx + 1 < W ? cx + 1 : W - 1;
    float v[8][8];                                         // [(zi, yi, xi)][channel]
#pragma unroll
    for (int k = 0; k < 8; ++k)
      load8<T, FMT>(rows[k >> 1] + Fmt<FMT>::group_rx(0, (k & 1 ? x1 : x0), C, W, c8), Fmt<FMT>::lo_off(C, W), v[k]);
    if (ab) {
      const float* q = ab + ((long long)n * C + c8 * 8) * 2;
      const float ak = act_k(act, slope);
      bool bad = false;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = q[2 * e], b = q[2 * e + 1];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[k][e] = act_fwd(v[k][e] * a + b, ak);
          if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(v[k][e]);
        }
      }
      if (RangeCheck<T>::on) raise_flag(oflow, bad);
    }
#pragma unroll
    // (A separable form -- x, y in place, then z: 36 instead of 64 operations per channel -- was measured SLOWER on the same box:
    //  64 -> 64 into 128^3 646 -> 726 us, 128 -> 128 into 64^3 162 -> 182; the eight independent weighted sums schedule better.)
    for (int o = 0; o < 8; ++o) {                          // output (pz, py, px): 0 = odd position 2c+1, 1 = even position 2c+2
      const int pz = o >> 2, py = (o >> 1) & 1, px = o & 1;
      const int oz = 2 * cz + 1 + pz, oy = 2 * cy + 1 + py, ox = 2 * cx + 1 + px;
      if (oz < 0 || oz >= 2 * D || oy < 0 || oy >= 2 * H || ox < 0 || ox >= 2 * W) continue;
      // weight of the far neighbour (index c+1): 0.25 at the odd position, 0.75 at the even one
      const float wz = pz ? 0.75f : 0.25f, wy = py ? 0.75f : 0.25f, wx = px ? 0.75f : 0.25f;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float wt = ((k & 4) ? wz : 1.f - wz) * ((k & 2) ? wy : 1.f - wy) * ((k & 1) ? wx : 1.f - wx);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += wt * v[k][e];
      }
      store8<T, FMT>(out + (((long long)n * 2 * D + oz) * 2 * H + oy) * orowb + Fmt<FMT>::group_rx(0, ox, C, 2 * W, c8), Fmt<FMT>::lo_off(C, 2 * W), acc, c8, skip_lo != 0);
    }
  }
}

// adjoint of upsample2_trilinear: gin [N][D][H][W][C] <- gout [N][2D][2H][2W][C].  Per axis input j collects the outputs
// 2j-1, 2j, 2j+1, 2j+2 with weights 0.25, 0.75, 0.75, 0.25; indices -1 and 2L fold onto 0 and 2L-1 (the forward's clamps).
template <typename T>
__global__ void upsample2_trilinear_bwd_kernel(const char* __restrict__ gout, char* __restrict__ gin, int N, int D, int H,
                                               int W, int C) {
  const int c8n = C >> 3;
  const long long total = (long long)N * D * H * W * c8n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = idx % c8n;
    long long r = idx / c8n;
    const int x = r % W;
    r /= W;
    const int y = r % H;
    r /= H;
    const int z = r % D;
    const int n = r / D;
    const float wt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < 4; ++a) {
      int oz = 2 * z - 1 + a;
      oz = oz < 0 ? 0 : (oz > 2 * D - 1 ? 2 * D - 1 : oz);
      for (int b = 0; b < 4; ++b) {
        int oy = 2 * y - 1 + b;
        oy = oy < 0 ? 0 : (oy > 2 * H - 1 ? 2 * H - 1 : oy);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int ox = 2 * x - 1 + c;
          ox = ox < 0 ? 0 : (ox > 2 * W - 1 ? 2 * W - 1 : ox);
          float f[8];
          unpack8<T>(*(const uint4*)(gout + ((((long long)n * 2 * D + oz) * 2 * H + oy) * 2 * W + ox) * C * 2 + c8 * 16), f);
          const float w = wt[a] * wt[b] * wt[c];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w * f[e];
        }
      }
    }
    *(uint4*)(gin + idx * 16) = pack8<T>(acc);
  }
}

// y = scale[c] * x + shift[c], then the activation, in place (eval-mode BatchNorm applied as its own pass: only
// used when a feature tap asks for the pre-norm convolution output, network.py:475-529).
template <typename T, int FMT>
__global__ void affine_act_kernel(char* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                  long long nvox, int C, int act, float slope, int* oflow) {
  bool bad = false;
  const int c8n = C >> 3;
  const long long total = nvox * c8n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = idx % c8n;
    float f[8];
    static_assert(FMT < 2, "eval-BatchNorm networks do not run in the row-planar layout");
    char* xp = x + (idx / c8n) * (C * 2 * Fmt<FMT>::M) + c8 * 16;
    load8<T, FMT>(xp, C * 2, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = f[e] * scale[c8 * 8 + e] + shift[c8 * 8 + e];
      v = act_fwd(v, act_k(act, slope));
      if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(v);   // the value about to be stored
      f[e] = v;
    }
    store8<T, FMT>(xp, C * 2, f, c8);
  }
  if (RangeCheck<T>::on) raise_flag(oflow, bad);
}

// Feature tap: 16-bit channels-last -> fp32 NCDHW [N][C0+C1][D][H][W], the layout the reference hands to its
// callers.  Channels [0,C0) come from src0 (full resolution), [C0,C0+C1) from src1, read through >> up_shift
// (the tap at an nn.Upsample id is taken after torch.cat((skip, up), 1), network.py:500-502).
// One thread = one voxel x 8 channels; a wavefront writes 64 consecutive floats of each of its 8 planes.
template <typename T, int FMT>
__global__ void export_ncdhw_kernel(const char* __restrict__ src0, int C0, const char* __restrict__ src1, int C1, int up_shift,
                                    int N, int D, int H, int W, float* __restrict__ out, int S0, int S1) {     // S: stored channels per voxel (>= C)
  const int C = C0 + C1, c8n = (C + 7) >> 3;         // C1 == 0 may leave a ragged last group (stored channels cover it)
  const long long vox = (long long)D * H * W;
  const long long total = (long long)N * c8n * vox;
  const int lw = W >> up_shift, lh = H >> up_shift, ld = D >> up_shift;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long v = idx % vox;
    const long long r = idx / vox;
    const int c8 = r % c8n, n = r / c8n;
    const int c = c8 * 8;
    const char* sp;
    int lo_off = Fmt<FMT>::lo_off(S0, W);
    if (c < C0) {
      sp = src0 + Fmt<FMT>::group((long long)n * vox + v, S0, W, c8);
    } else {
      lo_off = Fmt<FMT>::lo_off(S1, lw);
      const int x = v % W, y = (v / W) % H, z = v / ((long long)W * H);
      const long long lv = (((long long)n * ld + (z >> up_shift)) * lh + (y >> up_shift)) * lw + (x >> up_shift);
      sp = src1 + Fmt<FMT>::group(lv, S1, lw, (c - C0) >> 3);
    }
    float f[8];
    load8<T, FMT>(sp, lo_off, f);
    float* o = out + ((long long)n * C + c) * vox + v;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c + e < C) o[e * vox] = f[e];
  }
}

// The reverse copy for the training path: fp32 planar [N][C][D][H][W] -> 16-bit channels-last, written through BYTE strides
// (so the interior of a zero-framed gradient buffer can be the destination), optionally accumulating into what is there.
// One thread = one voxel x 8 channels: eight coalesced plane reads, one 16-byte store.  Cs = channels of the source tensor,
// Cd = channel pitch of the destination voxel (>= Cs: e.g. one real channel padded to 16 is not needed here, Cs % 8 == 0).
template <typename T>
__global__ void import_ncdhw_kernel(const float* __restrict__ src, char* __restrict__ dst, int N, int C, int D, int H, int W,
                                    long long dn, long long dz, long long dy, long long dx, int accumulate) {
  const int c8n = C >> 3;
  const long long vox = (long long)D * H * W;
  const long long total = (long long)N * c8n * vox;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long v = idx % vox;
    const long long r = idx / vox;
    const int c8 = r % c8n, n = r / c8n;
    const int x = v % W, y = (v / W) % H, z = v / ((long long)W * H);
    const float* sp = src + ((long long)n * C + c8 * 8) * vox + v;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = sp[e * vox];
    char* o = dst + n * dn + z * dz + y * dy + x * dx + c8 * 16;
    if (accumulate) {
      float g[8];
      unpack8<T>(*(const uint4*)o, g);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += g[e];
    }
    *(uint4*)o = pack8<T>(f);
  }
}

// Network input with MORE THAN ONE channel (the reference's constructor takes any input_nc; its pretraining options default to 2,
// pretraining/options/base_options.py:69-73): fp32 [N][Cin][D][H][W] -> 16-bit channels-last with 16 stored channels (Cin real, the
// rest zero), after which the first conv is an ordinary 16 -> ngf layer (the single-channel stem kernel is not involved).
// One thread per voxel; SPLIT: [hi(16) | lo(16)].
template <typename T, int FMT>
__global__ void import_input_kernel(const float* __restrict__ src, char* __restrict__ dst, int N, int Cin, long long vox) {
  const long long total = (long long)N * vox;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long n = idx / vox, v = idx - n * vox;
    float f0[8], f1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      f0[c] = c < Cin ? src[(n * Cin + c) * vox + v] : 0.f;
      f1[c] = c + 8 < Cin ? src[(n * Cin + c + 8) * vox + v] : 0.f;
    }
    static_assert(FMT < 2, "the multi-channel input import writes channels-last voxels");
    char* o = dst + idx * (32 * Fmt<FMT>::M);
    store8<T, FMT>(o, 32, f0, 0);
    store8<T, FMT>(o + 16, 32, f1, 1);
  }
}

hipError_t launch_import_input(const float* src, void* dst, int N, int Cin, long long vox, int precision, hipStream_t st) {
  if (Cin < 1 || Cin > 16) return hipErrorInvalidValue;
  const long long total = (long long)N * vox;
  const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
#define AMX_II(T, S) hipLaunchKernelGGL((import_input_kernel<T, S>), dim3(blocks), dim3(256), 0, st, src, (char*)dst, N, Cin, vox)
  switch (precision) {
    case 0: AMX_II(f16, false); break;
    case 1: AMX_II(bf16, false); break;
    case 2: AMX_II(f16, true); break;
    case 3: AMX_II(bf16, true); break;
    default: return hipErrorInvalidValue;
  }
#undef AMX_II
  return hipGetLastError();
}

// Last launch of a forward: if any epilogue raised the range flag, the network output is overwritten with NaN -- the
// features a caller holds after an f16 overflow are unmistakably invalid, whatever the caller does with status codes.
__global__ void poison_if_flag_kernel(const int* __restrict__ flag, int* host_flag, float* __restrict__ y, long long count) {
  if (*flag == 0) return;
  // the host-visible mirror (pinned, mapped memory) is written from here: no device-to-host copy per forward on the stream
  if (host_flag && blockIdx.x == 0 && threadIdx.x == 0) {
    __atomic_store_n(host_flag, 1, __ATOMIC_RELAXED);
    __threadfence_system();
  }
  const float nan = __builtin_nanf("");
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) y[i] = nan;
}

// y == nullptr / count == 0: only mirror the flag (paths whose output extent the library does not know)
hipError_t launch_poison_if_flag(const int* flag, int* host_flag, float* y, long long count, hipStream_t st) {
  hipLaunchKernelGGL(poison_if_flag_kernel, dim3(count > 0 ? 512 : 1), dim3(256), 0, st, flag, host_flag, y, count);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------- launchers
// scratch: [N][slots][C][2] partial sums + [N][C][2] coefficients.  slots * C <= 65536 for the separate statistics pass; a conv
// epilogue writes one slot per (brick, wave): the caller passes the largest slots * C of its layers (conv_v2_stats_slots)
size_t instnorm_scratch_bytes(int N, int C, long long max_slots_x_C) {
  const size_t per = (size_t)max_slots_x_C > 65536 ? (size_t)max_slots_x_C : 65536;
  return ((size_t)N * per * 2 + (size_t)N * C * 2 + (size_t)N * 64 * C * 2) * sizeof(float);      // + chunk sums of in_prereduce
}
static size_t in_coeff_offset(int N, long long vox, int C, int fused_slots) {      // floats
  (void)vox;
  const size_t per = (size_t)fused_slots * C > 65536 ? (size_t)fused_slots * C : 65536;
  return (size_t)N * per * 2;
}

// fused_slots > 0: the partial sums [N][fused_slots][C][2] were written by the producing conv's epilogue (shift = kshift, the conv
// bias, or 0): only finalize + apply run here.
// W: row length of the row-planar layout (precision 4 only; amx_common.h FMT 2)
hipError_t launch_instnorm(void* x, const float* gamma, const float* beta, float eps, int N, long long vox, int C, int act,
                           float slope, void* scratch, int precision, hipStream_t st, int* oflow, int fused_slots = 0,
                           const float* kshift = nullptr, int W = 0, int skip_lo = 0, int apply = 1, float* ab_out = nullptr) {
  // apply = 0: statistics and finalize only -- the (a, b) pairs go to ab_out [N][C][2] and the CONSUMER normalises on the way in
  // (amx_conv3d_zx.hip); the tensor stays raw
  if (C % 8) return hipErrorInvalidValue;
  if (precision == 4 && (W <= 0 || vox % W || C % 16)) return hipErrorInvalidValue;
  float* partial = (float*)scratch;
  float* ab_in = partial + in_coeff_offset(N, vox, C, fused_slots);
  float* ab = ab_out ? ab_out : ab_in;
  int nblk = fused_slots > 0 ? fused_slots : in_num_blocks(vox, C);
  const bool prereduce = nblk > 512;      // (in_finalize walks the slots 32 at a time with dependent adds: 4096 slots took 55 us, 64 chunks take 5)
  // prereduce: fold the slots into kPreChunks chunks first (the chunk sums live behind the coefficients: N * kPreChunks * C * 2 floats,
  // instnorm_scratch_bytes reserves them)
  float* partial2 = ab_in + (size_t)N * C * 2;
  const int c8n = C / 8;
  if (c8n > 256) return hipErrorInvalidValue;          // C <= 2048
  const int nrow = 256 / c8n;
  const size_t lds = (size_t)nrow * C * 2 * sizeof(float);
  const long long total = (long long)N * vox * c8n;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
#define AMX_IN(T, S)                                                                                                   \
  if (fused_slots <= 0)                                                                                                \
    hipLaunchKernelGGL((in_stats_kernel<T, S>), dim3(nblk, N), dim3(256), lds, st, (const char*)x, partial, vox, C, W);   \
  if (prereduce) {                                                                                                     \
    hipLaunchKernelGGL(in_prereduce_kernel, dim3(C / 8, N, kPreChunks), dim3(256), 0, st, partial, partial2, C, nblk);    \
    partial = partial2;                                                                                                \
    nblk = kPreChunks;                                                                                                 \
  }                                                                                                                    \
  hipLaunchKernelGGL((in_finalize_kernel<T, S>), dim3(C / 8, N), dim3(256), 0, st, (const char*)x, partial, gamma, beta, eps, vox, C, \
                     nblk, ab, kshift, fused_slots > 0 ? 1 : 0, W);                                               \
  if (!apply) {                                                                                                  \
  } else if (256 % c8n == 0) {                                                                                   \
    const long long per = vox * c8n;                                                                             \
    const int bx = (int)((per + 255) / 256 > 4096 ? 4096 : (per + 255) / 256);                                   \
    hipLaunchKernelGGL((in_apply_fast_kernel<T, S>), dim3(bx, N), dim3(256), 0, st, (char*)x, ab, vox, C, act, slope, oflow, W, skip_lo); \
  } else                                                                                                         \
    hipLaunchKernelGGL((in_apply_kernel<T, S>), dim3(blocks), dim3(256), 0, st, (char*)x, ab, vox, C, N, act, slope, oflow, W, skip_lo)
  switch (precision) {
    case 0: { AMX_IN(f16, false); } break;
    case 1: { AMX_IN(bf16, false); } break;
    case 2: { AMX_IN(f16, true); } break;
    case 3: { AMX_IN(bf16, true); } break;
    case 4: { AMX_IN(f16, 2); } break;
    default: return hipErrorInvalidValue;
  }
#undef AMX_IN
  return hipGetLastError();
}

// f16x2mx only: in-place norm apply + activation of x [N][D][H][W][C] (row-planar) and its 2x2x2 avg / max pooled copy
bool in_apply_pool_eligible(int precision, int D, int H, int W, int C) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_APPLY_POOL") ? 1 : 0;
  return !off && precision == 4 && !(D & 1) && !(H & 1) && !(W & 1) && C % 16 == 0 && ((long long)W * (C / 8)) % 64 == 0;
}
hipError_t launch_in_apply_pool(void* x, const float* ab, void* pooled, int N, int D, int H, int W, int C, int act, float slope, int avg,
                                int skip_lo, int pool_skip_lo, int* oflow, hipStream_t st) {
  hipLaunchKernelGGL(in_apply_pool_kernel<f16>, dim3((unsigned)((long long)N * (D / 2) * (H / 2))), dim3(256), 0, st, (char*)x, ab,
                     (char*)pooled, D / 2, H / 2, W, C, act, slope, avg, skip_lo, pool_skip_lo, oflow);
  return hipGetLastError();
}

hipError_t launch_upsample2_trilinear(const void* in, void* out, int N, int D, int H, int W, int C, int precision,
                                      hipStream_t st, int skip_lo = 0, const float* ab = nullptr, int act = 0, float slope = 0.f,
                                      int* oflow = nullptr) {
  const unsigned blocks = (unsigned)((long long)N * (D + 1) * (H + 1));
#define AMX_UP(T, S) hipLaunchKernelGGL((upsample2_trilinear_kernel<T, S>), dim3(blocks), dim3(256), 0, st, (const char*)in, (char*)out, N, D, H, W, C, skip_lo, ab, act, slope, oflow)
  switch (precision) {
    case 0: AMX_UP(f16, false); break;
    case 1: AMX_UP(bf16, false); break;
    case 2: AMX_UP(f16, true); break;
    case 3: AMX_UP(bf16, true); break;
    case 4: AMX_UP(f16, 2); break;
    default: return hipErrorInvalidValue;
  }
#undef AMX_UP
  return hipGetLastError();
}

hipError_t launch_upsample2_trilinear_backward(const void* gout, void* gin, int N, int D, int H, int W, int C, int precision,
                                               hipStream_t st) {
  if (C % 8) return hipErrorInvalidValue;
  const long long total = (long long)N * D * H * W * (C / 8);
  const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  if (precision == 0)
    hipLaunchKernelGGL(upsample2_trilinear_bwd_kernel<f16>, dim3(blocks), dim3(256), 0, st, (const char*)gout, (char*)gin, N, D, H, W, C);
  else
    hipLaunchKernelGGL(upsample2_trilinear_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const char*)gout, (char*)gin, N, D, H, W, C);
  return hipGetLastError();
}

hipError_t launch_affine_act(void* x, const float* scale, const float* shift, int N, long long vox, int C, int act,
                             float slope, int precision, hipStream_t st, int* oflow) {
  if (C % 8) return hipErrorInvalidValue;
  const long long total = (long long)N * vox * (C / 8);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
#define AMX_AA(T, S) hipLaunchKernelGGL((affine_act_kernel<T, S>), dim3(blocks), dim3(256), 0, st, (char*)x, scale, shift, (long long)N * vox, C, act, slope, oflow)
  switch (precision) {
    case 0: AMX_AA(f16, false); break;
    case 1: AMX_AA(bf16, false); break;
    case 2: AMX_AA(f16, true); break;
    case 3: AMX_AA(bf16, true); break;
    default: return hipErrorInvalidValue;
  }
#undef AMX_AA
  return hipGetLastError();
}

// planar0 / planar1 (precisions 0 / 1): that segment is stored row-planar (layout FMT 3); both or neither when both are given
hipError_t launch_export_ncdhw(const void* src0, int C0, const void* src1, int C1, int up_shift, int N, int D, int H, int W,
                               float* out, int precision, hipStream_t st, int S0, int S1, int planar) {
  if (planar && precision > 1) return hipErrorInvalidValue;
  if (S0 <= 0) S0 = C0;
  if (S1 <= 0) S1 = C1;
  // a ragged channel count is only possible for a single segment whose storage is padded (the output conv of a network whose
  // output_nc is not a multiple of 16)
  if (C0 + C1 < 1 || (C1 > 0 && (C0 % 8 || C1 % 8)) || (C1 == 0 && S0 < (C0 + 7) / 8 * 8)) return hipErrorInvalidValue;
  const long long total = (long long)N * ((C0 + C1 + 7) / 8) * D * H * W;
  const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
#define AMX_EX(T, S) hipLaunchKernelGGL((export_ncdhw_kernel<T, S>), dim3(blocks), dim3(256), 0, st, (const char*)src0, C0, (const char*)src1, C1, up_shift, N, D, H, W, out, S0, S1)
  switch (precision + (planar ? 10 : 0)) {
    case 10: AMX_EX(f16, 3); break;
    case 11: AMX_EX(bf16, 3); break;
    case 0: AMX_EX(f16, false); break;
    case 1: AMX_EX(bf16, false); break;
    case 2: AMX_EX(f16, true); break;
    case 3: AMX_EX(bf16, true); break;
    case 4: AMX_EX(f16, 2); break;
    default: return hipErrorInvalidValue;
  }
#undef AMX_EX
  return hipGetLastError();
}

hipError_t launch_import_ncdhw(const float* src, void* dst, int N, int C, int D, int H, int W, long long dn, long long dz,
                               long long dy, long long dx, int accumulate, int precision, hipStream_t st) {
  if (C % 8 || C < 8) return hipErrorInvalidValue;
  const long long total = (long long)N * (C / 8) * D * H * W;
  const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  if (precision == 0)
    hipLaunchKernelGGL(import_ncdhw_kernel<f16>, dim3(blocks), dim3(256), 0, st, src, (char*)dst, N, C, D, H, W, dn, dz, dy, dx, accumulate);
  else
    hipLaunchKernelGGL(import_ncdhw_kernel<bf16>, dim3(blocks), dim3(256), 0, st, src, (char*)dst, N, C, D, H, W, dn, dz, dy, dx, accumulate);
  return hipGetLastError();
}

}  // namespace amx
