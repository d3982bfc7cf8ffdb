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
// filter -- exactly the two ReplicationPad3d of the reference), then every thread box-sums its voxels.  The 12 sums stay
// in registers; the epilogue subtracts the per-voxel minimum, writes the unnormalised descriptor and reduces the
// per-voxel channel mean for the global mean the reference clamps against.
template <int R>
__global__ __launch_bounds__(256) void mind_ssd_kernel(const float* __restrict__ img, int H, int W, int D, int dil, MindPairs mp,
                                                       float* __restrict__ out, double* __restrict__ partials) {
  constexpr int TZ = 8, TY = 8, TX = 16, HZ = TZ + 2 * R, HY = TY + 2 * R, HX = TX + 2 * R, HXP = HX + 1;
  constexpr int K = 2 * R + 1;
  extern __shared__ float lds[];
  // image brick with an (R + dil) halo, holding I(clamp(coordinate)): one global load + clamp per point, once
  const int G = R + dil, IZ = TZ + 2 * G, IY = TY + 2 * G, IX = TX + 2 * G, IXP = IX | 1;
  float* it = lds;                                   // [IZ][IY][IXP]
  float* d2 = it + IZ * IY * IXP;                    // [HZ][HY][HXP]
  __shared__ double red[256];
  const int bx = blockIdx.x * TX, by = blockIdx.y * TY, bz = blockIdx.z * TZ;
  const int tx = threadIdx.x & 15, ty = (threadIdx.x >> 4) & 7, tz = (threadIdx.x >> 7) * 4;   // 4 consecutive z per thread
  for (int t = threadIdx.x; t < IZ * IY * IX; t += 256) {
    const int ix = t % IX, iy = (t / IX) % IY, iz = t / (IX * IY);
    it[(iz * IY + iy) * IXP + ix] =
        img[((long long)clampi(bz + iz - G, H - 1) * W + clampi(by + iy - G, W - 1)) * D + clampi(bx + ix - G, D - 1)];
  }
  __syncthreads();
  float ssd[12][4];                                  // box sums of this thread's voxels (c is unrolled: static indices)
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const int oa = (mp.a[c][0] * IY + mp.a[c][1]) * IXP + mp.a[c][2], ob = (mp.b[c][0] * IY + mp.b[c][1]) * IXP + mp.b[c][2];
#pragma unroll 1
    for (int t = threadIdx.x; t < HZ * HY * HX; t += 256) {
      const int hx = t % HX, hy = (t / HX) % HY, hz = t / (HX * HY);
      // the box filter replicates the DIFFERENCE field at the volume border: evaluate it at the clamped position
      const int z = clampi(bz + hz - R, H - 1) - bz + G, y = clampi(by + hy - R, W - 1) - by + G, x = clampi(bx + hx - R, D - 1) - bx + G;
      const int base = (z * IY + y) * IXP + x;
      const float df = it[base + oa] - it[base + ob];
      d2[(hz * HY + hy) * HXP + hx] = df * df;
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int hz = 0; hz < 4 + 2 * R; ++hz) {          // K x K plane sums, each shared by the K outputs it belongs to
      float ps = 0.f;
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) ps += d2[((tz + hz) * HY + ty + ky) * HXP + tx + kx];
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (hz >= o && hz < o + K) acc[o] += ps;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) ssd[c][o] = acc[o] / (float)(K * K * K);
    __syncthreads();
  }
  const long long plane = (long long)H * W * D;
  double vsum = 0.0;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int z = bz + tz + v, y = by + ty, x = bx + tx;
    if (z >= H || y >= W || x >= D) continue;
    float mn = ssd[0][v];
#pragma unroll
    for (int c = 1; c < 12; ++c) mn = fminf(mn, ssd[c][v]);
    float sum = 0.f;
    const long long o = ((long long)z * W + y) * D + x;
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      const float m = ssd[c][v] - mn;
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

// avg_pool3d(k, stride 1, padding k/2), zero padding counted in the divisor.  4 x 8 x 64 outputs per block from an LDS
// brick with an r halo (zeros outside the volume); each thread owns two (y, x) columns and walks z with k x k plane sums
// shared by the k outputs they contribute to.
__global__ __launch_bounds__(256) void box_filter_kernel(const float* __restrict__ in, float* __restrict__ out, int H,
                                                         int W, int D, int k) {
  constexpr int TZ = 4, TY = 8, TX = 64;
  extern __shared__ float lds[];
  const int r = k / 2, HZ = TZ + 2 * r, HY = TY + 2 * r, HX = TX + 2 * r, HXP = HX | 1;
  const long long plane = (long long)H * W * D;
  const int nbx = (D + TX - 1) / TX, nby = (W + TY - 1) / TY;
  const int bx = (blockIdx.x % nbx) * TX, by = ((blockIdx.x / nbx) % nby) * TY, bz = (blockIdx.x / (nbx * nby)) * TZ;
  const float* src = in + (long long)blockIdx.y * plane;
  for (int t = threadIdx.x; t < HZ * HY * HX; t += 256) {
    const int hx = t % HX, hy = (t / HX) % HY, hz = t / (HX * HY);
    const int z = bz + hz - r, y = by + hy - r, x = bx + hx - r;
    lds[(hz * HY + hy) * HXP + hx] = (z >= 0 && z < H && y >= 0 && y < W && x >= 0 && x < D) ? src[((long long)z * W + y) * D + x] : 0.f;
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
  const float div = (float)(k * k * k);
  for (int yy = 0; yy < 2; ++yy) {
    const int ty = ty0 + 4 * yy;
    float acc[TZ] = {0.f, 0.f, 0.f, 0.f};
    for (int hz = 0; hz < HZ; ++hz) {
      float ps = 0.f;
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) ps += lds[(hz * HY + ty + ky) * HXP + tx + kx];
#pragma unroll
      for (int o = 0; o < TZ; ++o)
        if (hz >= o && hz < o + k) acc[o] += ps;
    }
    const int y = by + ty, x = bx + tx;
    if (y < W && x < D)
#pragma unroll
      for (int o = 0; o < TZ; ++o)
        if (bz + o < H) out[(long long)blockIdx.y * plane + ((long long)(bz + o) * W + y) * D + x] = acc[o] / div;
  }
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
  const int G = radius + dilation, HR = 2 * radius;
  const size_t lds = ((size_t)(8 + 2 * G) * (8 + 2 * G) * ((16 + 2 * G) | 1) + (size_t)(8 + HR) * (8 + HR) * (16 + HR + 1)) * 4;
  if (lds > 150 * 1024) return hipErrorInvalidValue;   // dilation too large for the LDS brick (checked by the caller)
  auto kern = radius == 1 ? mind_ssd_kernel<1> : mind_ssd_kernel<2>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  kern<<<grid, 256, lds, st>>>(img, H, W, D, dilation, mp, out, partials);
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

static inline dim3 box_grid(int C, int H, int W, int D) { return dim3(cdiv_i(D, 64) * cdiv_i(W, 8) * cdiv_i(H, 4), C); }
static inline size_t box_lds(int k) {
  const int r = k / 2;
  return (size_t)(4 + 2 * r) * (8 + 2 * r) * ((64 + 2 * r) | 1) * 4;
}

hipError_t launch_box_filter(const float* in, float* out, int C, int H, int W, int D, int k, hipStream_t st) {
  box_filter_kernel<<<box_grid(C, H, W, D), 256, box_lds(k), st>>>(in, out, H, W, D, k);
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
  box_filter_kernel<<<box_grid(n, h, w, d), 256, box_lds(3), st>>>(ssd, tmp, h, w, d, 3);
  box_filter_kernel<<<box_grid(n, h, w, d), 256, box_lds(3), st>>>(tmp, ssd, h, w, d, 3);
  if (argmin) argmin_kernel<<<cdiv_i(plane, 256), 256, 0, st>>>(ssd, n, plane, argmin);
  return hipGetLastError();
}

}  // namespace amx
