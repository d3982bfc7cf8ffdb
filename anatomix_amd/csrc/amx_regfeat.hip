// anatomix_amd -- registration feature post-processing that follows feature extraction:
//   MIND-SSC descriptor            anatomix/registration/convex_adam_utils.py:311-406 (MINDSSC)
//   cat(mind, 0.1 * features) -> avg_pool3d(grid_sp)      run_convex_adam_with_network_feats.py:164-205,
//                                                          instance_optimization.py:105-117 (merge_features)
//   avg_pool3d(k, stride 1, zero padding) xN               convex_adam_utils.py:105-131 (apply_avg_pool3d)
//   SSD correlation volume + argmin                        convex_adam_utils.py:409-491 (correlate)
// All fp32, planar [C][H][W][D] like the reference's tensors (batch 1 everywhere in that pipeline).  Every one of these is
// HBM/cache-bound stencil work on tensors that already sit in HBM; next to 343 UNet windows per volume they cost well
// under 1 % of a registration, so they are written for coalesced streaming and fixed summation order, not for peak.
#include "amx_device.h"

namespace amx {

struct MindPairs {
  int a[12][3], b[12][3];   // sample offsets (already multiplied by the dilation) of the two voxels channel c compares
  int slot[12];             // output channel that shows pair c
};

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// One 8 x 8 x 16 brick per block, 4 voxels per thread.  Per pair c: the squared difference field on the brick + R halo
// goes to LDS (sampled with replication at the volume border, twice: once for the dilated samples, once for the box
// filter -- exactly the two ReplicationPad3d of the reference), then every thread box-sums its voxels.  The 12 sums wait
// in LDS; the epilogue subtracts the per-voxel minimum, writes the unnormalised descriptor and reduces the
// per-voxel channel mean for the global mean the reference clamps against.
template <int R>
__global__ __launch_bounds__(256) void mind_ssd_kernel(const float* __restrict__ img, int H, int W, int D, MindPairs mp,
                                                       float* __restrict__ out, double* __restrict__ partials) {
  constexpr int TZ = 8, TY = 8, TX = 16, HZ = TZ + 2 * R, HY = TY + 2 * R, HX = TX + 2 * R, HXP = HX + 1;
  constexpr int K = 2 * R + 1;
  __shared__ float d2[HZ * HY * HXP];
  __shared__ double red[256];
  const int bx = blockIdx.x * TX, by = blockIdx.y * TY, bz = blockIdx.z * TZ;
  const int tx = threadIdx.x & 15, ty = (threadIdx.x >> 4) & 7, tz = threadIdx.x >> 7;
  __shared__ float ssd[12][4 * 256];   // [pair][voxel of the brick]: LDS rather than a dynamically indexed register array
#pragma unroll 1
  for (int c = 0; c < 12; ++c) {
    for (int t = threadIdx.x; t < HZ * HY * HX; t += 256) {
      const int hx = t % HX, hy = (t / HX) % HY, hz = t / (HX * HY);
      const int z = clampi(bz + hz - R, H - 1), y = clampi(by + hy - R, W - 1), x = clampi(bx + hx - R, D - 1);
      const float va = img[((long long)clampi(z + mp.a[c][0], H - 1) * W + clampi(y + mp.a[c][1], W - 1)) * D + clampi(x + mp.a[c][2], D - 1)];
      const float vb = img[((long long)clampi(z + mp.b[c][0], H - 1) * W + clampi(y + mp.b[c][1], W - 1)) * D + clampi(x + mp.b[c][2], D - 1)];
      const float df = va - vb;
      d2[(hz * HY + hy) * HXP + hx] = df * df;
    }
    __syncthreads();
#pragma unroll 1
    for (int v = 0; v < 4; ++v) {
      const int lz = tz + 2 * v;
      float s = 0.f;
#pragma unroll 1
      for (int kz = 0; kz < K; ++kz)
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
          for (int kx = 0; kx < K; ++kx) s += d2[((lz + kz) * HY + ty + ky) * HXP + tx + kx];
      ssd[c][v * 256 + threadIdx.x] = s / (float)(K * K * K);
    }
    __syncthreads();
  }
  const long long plane = (long long)H * W * D;
  double vsum = 0.0;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int z = bz + tz + 2 * v, y = by + ty, x = bx + tx;
    if (z >= H || y >= W || x >= D) continue;
    float sv[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) sv[c] = ssd[c][v * 256 + threadIdx.x];
    float mn = sv[0];
#pragma unroll
    for (int c = 1; c < 12; ++c) mn = fminf(mn, sv[c]);
    float sum = 0.f;
    const long long o = ((long long)z * W + y) * D + x;
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      const float m = sv[c] - mn;
      sum += m;
      out[(long long)mp.slot[c] * plane + o] = m;
    }
    vsum += (double)(sum / 12.f);
  }
  red[threadIdx.x] = vsum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0];
}

// fixed-order tree over the block partials -> global mean of the per-voxel channel mean (one float)
__global__ __launch_bounds__(256) void mind_mean_kernel(const double* __restrict__ partials, int n, double inv_count,
                                                        float* __restrict__ gm) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) *gm = (float)(red[0] * inv_count);
}

// in place: var = mean_c mind_c clamped to [gm * 1e-3, gm * 1e3]; mind_c = exp(-mind_c / var)
__global__ __launch_bounds__(256) void mind_finish_kernel(float* __restrict__ out, long long plane, MindPairs mp,
                                                          const float* __restrict__ gm) {
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= plane) return;
  const float g = *gm;
  const float lo = (float)((double)g * 0.001), hi = (float)((double)g * 1000.0);
  float m[12], sum = 0.f;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    m[c] = out[(long long)mp.slot[c] * plane + o];
    sum += m[c];
  }
  float var = sum / 12.f;
  var = fminf(fmaxf(var, lo), hi);
#pragma unroll
  for (int c = 0; c < 12; ++c) out[(long long)mp.slot[c] * plane + o] = expf(-(m[c] / var));
}

// out[c] = mean over g^3 of (c < ca ? sa * a[c] : sb * b[c - ca]);  one thread per output voxel, all channels
__global__ __launch_bounds__(256) void pool_cat_kernel(const float* __restrict__ a, int ca, float sa,
                                                       const float* __restrict__ b, int cb, float sb, int H, int W, int D,
                                                       int g, float* __restrict__ out) {
  const int h = H / g, w = W / g, d = D / g;
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long oplane = (long long)h * w * d, iplane = (long long)H * W * D;
  if (o >= oplane) return;
  const int x = (int)(o % d), y = (int)((o / d) % w), z = (int)(o / ((long long)d * w));
  for (int c = 0; c < ca + cb; ++c) {
    const float* src = c < ca ? a + (long long)c * iplane : b + (long long)(c - ca) * iplane;
    const float sc = c < ca ? sa : sb;
    float s = 0.f;
    for (int kz = 0; kz < g; ++kz)
      for (int ky = 0; ky < g; ++ky)
        for (int kx = 0; kx < g; ++kx) s += __fmul_rn(sc, src[((long long)(z * g + kz) * W + y * g + ky) * D + x * g + kx]);   // the product is rounded first, as in `pred * 0.1`
    out[(long long)c * oplane + o] = s / (float)(g * g * g);
  }
}

// avg_pool3d(k, stride 1, padding k/2), zero padding counted in the divisor
__global__ __launch_bounds__(256) void box_filter_kernel(const float* __restrict__ in, float* __restrict__ out, int H,
                                                         int W, int D, int k) {
  const long long plane = (long long)H * W * D;
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= plane) return;
  const int x = (int)(o % D), y = (int)((o / D) % W), z = (int)(o / ((long long)D * W));
  const float* src = in + (long long)blockIdx.y * plane;
  const int r = k / 2;
  float s = 0.f;
  for (int kz = -r; kz <= r; ++kz) {
    const int zz = z + kz;
    if (zz < 0 || zz >= H) continue;
    for (int ky = -r; ky <= r; ++ky) {
      const int yy = y + ky;
      if (yy < 0 || yy >= W) continue;
      for (int kx = -r; kx <= r; ++kx) {
        const int xx = x + kx;
        if (xx >= 0 && xx < D) s += src[((long long)zz * W + yy) * D + xx];
      }
    }
  }
  out[(long long)blockIdx.y * plane + o] = s / (float)(k * k * k);
}

// raw SSD: blockIdx.y = dz.  ssd[(dx*K + dy)*K + dz][p] = sum_c (fix[c][p] - mov0[c][p + (dz,dy,dx) - hw])^2, mov zero outside
template <int K>
__global__ __launch_bounds__(256) void ssd_raw_kernel(const float* __restrict__ fix, const float* __restrict__ mov, int C,
                                                      int h, int w, int d, float* __restrict__ ssd) {
  constexpr int HW = K / 2;
  const long long plane = (long long)h * w * d;
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= plane) return;
  const int x = (int)(o % d), y = (int)((o / d) % w), z = (int)(o / ((long long)d * w));
  const int dz = blockIdx.y, zz = z + dz - HW;
  const bool zin = zz >= 0 && zz < h;
  float acc[K][K];
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) acc[i][j] = 0.f;
  for (int c = 0; c < C; ++c) {
    const float f = fix[(long long)c * plane + o];
    const float* m = mov + (long long)c * plane;
#pragma unroll
    for (int dy = 0; dy < K; ++dy) {
      const int yy = y + dy - HW;
      const bool yin = zin && yy >= 0 && yy < w;
#pragma unroll
      for (int dx = 0; dx < K; ++dx) {
        const int xx = x + dx - HW;
        const float v = (yin && xx >= 0 && xx < d) ? m[((long long)zz * w + yy) * d + xx] : 0.f;
        const float df = f - v;
        acc[dy][dx] += df * df;
      }
    }
  }
#pragma unroll
  for (int dy = 0; dy < K; ++dy)
#pragma unroll
    for (int dx = 0; dx < K; ++dx) ssd[(long long)((dx * K + dy) * K + dz) * plane + o] = acc[dy][dx];
}

// first index of the minimum over the leading dimension
__global__ __launch_bounds__(256) void argmin_kernel(const float* __restrict__ ssd, int n, long long plane,
                                                     long long* __restrict__ idx) {
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= plane) return;
  float best = ssd[o];
  int bi = 0;
  for (int m = 1; m < n; ++m) {
    const float v = ssd[(long long)m * plane + o];
    if (v < best) {
      best = v;
      bi = m;
    }
  }
  idx[o] = bi;
}

static MindPairs make_pairs(int dilation) {
  // the six face neighbours in the reference's order (convex_adam_utils.py:333-340) and its pair selection rule
  // (:343-355): i > j, squared distance 2, row-major over (i, j); output channel k shows pair perm[k] (:395-402)
  static const int six[6][3] = {{-1, 0, 0}, {0, 0, -1}, {0, -1, 0}, {0, 0, 1}, {1, 0, 0}, {0, 1, 0}};
  static const int perm[12] = {6, 8, 1, 11, 2, 10, 0, 7, 9, 4, 5, 3};
  MindPairs mp;
  int c = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      int dist = 0;
      for (int e = 0; e < 3; ++e) dist += (six[i][e] - six[j][e]) * (six[i][e] - six[j][e]);
      if (i > j && dist == 2 && c < 12) {
        for (int e = 0; e < 3; ++e) {
          mp.a[c][e] = six[i][e] * dilation;
          mp.b[c][e] = six[j][e] * dilation;
        }
        ++c;
      }
    }
  for (int k = 0; k < 12; ++k) mp.slot[perm[k]] = k;
  return mp;
}

static inline int cdiv_i(long long a, long long b) { return (int)((a + b - 1) / b); }

size_t mindssc_scratch_bytes(int H, int W, int D) {
  const size_t blocks = (size_t)cdiv_i(D, 16) * cdiv_i(W, 8) * cdiv_i(H, 8);
  return blocks * sizeof(double) + 256;
}

hipError_t launch_mindssc(const float* img, int H, int W, int D, int radius, int dilation, float* out, void* scratch,
                          hipStream_t st) {
  const MindPairs mp = make_pairs(dilation);
  dim3 grid(cdiv_i(D, 16), cdiv_i(W, 8), cdiv_i(H, 8));
  const int nb = grid.x * grid.y * grid.z;
  double* partials = (double*)scratch;
  float* gm = (float*)((char*)scratch + (size_t)nb * sizeof(double));
  if (radius == 1) mind_ssd_kernel<1><<<grid, 256, 0, st>>>(img, H, W, D, mp, out, partials);
  else mind_ssd_kernel<2><<<grid, 256, 0, st>>>(img, H, W, D, mp, out, partials);
  const long long plane = (long long)H * W * D;
  mind_mean_kernel<<<1, 256, 0, st>>>(partials, nb, 1.0 / (double)plane, gm);
  mind_finish_kernel<<<cdiv_i(plane, 256), 256, 0, st>>>(out, plane, mp, gm);
  return hipGetLastError();
}

hipError_t launch_pool_cat(const float* a, int ca, float sa, const float* b, int cb, float sb, int H, int W, int D, int g,
                           float* out, hipStream_t st) {
  const long long oplane = (long long)(H / g) * (W / g) * (D / g);
  pool_cat_kernel<<<cdiv_i(oplane, 256), 256, 0, st>>>(a, ca, sa, b, cb, sb, H, W, D, g, out);
  return hipGetLastError();
}

hipError_t launch_box_filter(const float* in, float* out, int C, int H, int W, int D, int k, hipStream_t st) {
  const long long plane = (long long)H * W * D;
  box_filter_kernel<<<dim3(cdiv_i(plane, 256), C), 256, 0, st>>>(in, out, H, W, D, k);
  return hipGetLastError();
}

size_t correlate_scratch_bytes(int h, int w, int d, int disp_hw) {
  const size_t k = 2 * disp_hw + 1;
  return k * k * k * (size_t)h * w * d * sizeof(float);
}

hipError_t launch_correlate(const float* fix, const float* mov, int C, int h, int w, int d, int disp_hw, float* ssd,
                            long long* argmin, void* scratch, hipStream_t st) {
  const long long plane = (long long)h * w * d;
  const int k = 2 * disp_hw + 1, n = k * k * k;
  float* tmp = (float*)scratch;
  dim3 grid(cdiv_i(plane, 256), k);
  if (disp_hw == 1) ssd_raw_kernel<3><<<grid, 256, 0, st>>>(fix, mov, C, h, w, d, ssd);
  else if (disp_hw == 2) ssd_raw_kernel<5><<<grid, 256, 0, st>>>(fix, mov, C, h, w, d, ssd);
  else ssd_raw_kernel<7><<<grid, 256, 0, st>>>(fix, mov, C, h, w, d, ssd);
  box_filter_kernel<<<dim3(cdiv_i(plane, 256), n), 256, 0, st>>>(ssd, tmp, h, w, d, 3);
  box_filter_kernel<<<dim3(cdiv_i(plane, 256), n), 256, 0, st>>>(tmp, ssd, h, w, d, 3);
  if (argmin) argmin_kernel<<<cdiv_i(plane, 256), 256, 0, st>>>(ssd, n, plane, argmin);
  return hipGetLastError();
}

}  // namespace amx
