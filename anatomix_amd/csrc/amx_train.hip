// anatomix_amd -- bandwidth kernels of the TRAINING path (contrastive step, pretraining/models/supcl_model.py:603-661:
// the UNet runs in train mode, BatchNorm3d uses batch statistics over the two views, everything is differentiated).
// All tensors are dense 16-bit channels-last [N][voxels][C]; statistics and gradients of parameters are fp32;
// every reduction is two-stage with a fixed order (slab partials, then one thread per channel) -- no atomics.
//
//   bn_train_forward  : nn.BatchNorm3d(train) + activation                           network.py:148-152, 171-196
//                       y = act(gamma (x - mean) rstd + beta); saves mean / rstd; updates the running statistics
//                       (momentum 0.1, UNBIASED variance into running_var, as torch does)
//   bn_act_backward   : adjoint of the above: dz = dy * act'(y); dbeta = sum dz; dgamma = sum dz xhat;
//                       dx = gamma rstd (dz - dbeta / M - xhat dgamma / M), written into the interior of a zero-FRAMED
//                       buffer [N][D+4][H+4][W+4][C] -- the input layout of the data-gradient convolution below
//   pad_fold          : adjoint of the reflect padding of nn.Conv3d(padding='same', padding_mode='reflect'): the
//                       data gradient is first computed on the padded domain [-1, D] as a plain 3x3x3 correlation
//                       of the framed output gradient with the flipped, transposed weights (the forward kernel on a
//                       (D+4)^3 volume: its reflect gather only ever touches the zero frame there), then the border
//                       planes are folded back: dIn[1] += g[-1], dIn[D-2] += g[D] per axis
//   pool2_max_backward: adjoint of nn.MaxPool3d(2): the gradient goes to the FIRST maximum of each window in (z,y,x)
//                       order (torch's tie rule)
#include "amx_device.h"

namespace amx {

template <typename T>
__device__ __forceinline__ void t_unpack8(const uint4& raw, float (&f)[8]) {
  const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)__builtin_bit_cast(T, (unsigned short)(w[e >> 1] >> ((e & 1) * 16)));
}
template <typename T>
__device__ __forceinline__ uint4 t_pack8(const float (&f)[8]) {
  unsigned o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_bits<T>(f[2 * e]) | ((unsigned)to_bits<T>(f[2 * e + 1]) << 16);
  return make_uint4(o[0], o[1], o[2], o[3]);
}

__host__ __device__ inline int tr_num_blocks(long long rows, int C) {   // slabs of the statistics passes
  long long nb = rows * C / (8 * 256 * 8);
  if (nb > 1024) nb = 1024;
  if (nb * C > 65536) nb = 65536 / C;
  return nb < 1 ? 1 : (int)nb;
}

// ---------------------------------------------------------------- forward statistics
// grid nblk, block 256: partial[blk][c] = (count, sum(x - K_blk), sum (x - K_blk)^2) with K_blk = the slab's first row
// (shifted sums: no E[x^2] - E[x]^2 cancellation); rows = N * voxels.
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const char* __restrict__ x, float* __restrict__ partial, long long rows,
                                                      int C) {
  extern __shared__ float red[];                       // [nrow][C][2]
  const int c8n = C >> 3;
  const int c8 = threadIdx.x % c8n, vrow = threadIdx.x / c8n, nrow = 256 / c8n;
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long v0 = (long long)blockIdx.x * per, v1 = v0 + per < rows ? v0 + per : rows;
  float K[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (v0 < rows) t_unpack8<T>(*(const uint4*)(x + (v0 * C + c8 * 8) * 2), K);
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (vrow < nrow) {
    long long v = v0 + vrow;
    for (; v + 3ll * nrow < v1; v += 4ll * nrow) {          // four rows in flight; accumulated in row order (same sums as one by one)
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) raw[u] = *(const uint4*)(x + ((v + (long long)u * nrow) * C + c8 * 8) * 2);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        t_unpack8<T>(raw[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = f[e] - K[e];
          s1[e] += d;
          s2[e] += d * d;
        }
      }
    }
    for (; v < v1; v += nrow) {
      float f[8];
      t_unpack8<T>(*(const uint4*)(x + (v * C + c8 * 8) * 2), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = f[e] - K[e];
        s1[e] += d;
        s2[e] += d * d;
      }
    }
  }
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
    float* o = partial + ((long long)blockIdx.x * C + c) * 2;
    o[0] = a;
    o[1] = b;
  }
}

// grid C/2, block 256 = 2 channels x 128 lanes: two fp64 passes over the slab sums (mean, then centred second moment),
// lane l takes slabs l, l+128, ..., the 128 lane results are added in lane order -> mean, rstd, (a, b), running statistics.
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const char* __restrict__ x, const float* __restrict__ partial,
                                                         const float* gamma, const float* beta, float eps, long long rows, int C,
                                                         int nblk, float* __restrict__ ab, float* save_mean, float* save_rstd,
                                                         float* running_mean, float* running_var, float momentum) {
  constexpr int FC = 2, FL = 128;                  // channels per block x lanes per channel
  __shared__ double sh[2][FC][FL];
  const int cl = threadIdx.x % FC, l = threadIdx.x / FC;
  const int c = blockIdx.x * FC + cl;
  const long long per = (rows + nblk - 1) / nblk;
  const double inv_per = 1.0 / (double)per;
  // Slab b holds sums shifted by its own first element K_b: mean_b = K_b + s1 / n_b, M2_b = s2 - s1^2 / n_b.
  // Pass 1: total mean = sum n_b mean_b / N.  Pass 2: M2 = sum M2_b + n_b (mean_b - mean)^2.  Plain fp64 sums in a fixed
  // order (lane l takes slabs l, l + 128, ...; the 128 lane sums are folded by a fixed pairwise tree) -- no division per slab, and
  // few enough slabs per lane that the scattered K_b loads do not serialise.
  double sn = 0.0, sm = 0.0;
  if (c < C)
    for (int b = l; b < nblk; b += FL) {
      const long long v0 = (long long)b * per, v1 = v0 + per < rows ? v0 + per : rows;
      if (v0 >= rows) break;
      const double nb = (double)(v1 - v0);
      const float K = (float)__builtin_bit_cast(T, *(const unsigned short*)(x + (v0 * C + c) * 2));
      const double s1 = partial[((long long)b * C + c) * 2];
      sn += nb;
      sm += nb * (double)K + s1;
    }
  sh[0][cl][l] = sn;
  sh[1][cl][l] = sm;
  __syncthreads();
  // fixed-order pairwise tree over the 128 lane sums (a serial walk by one thread was 256 dependent LDS reads = most of this kernel's time)
  for (int s = FL / 2; s >= 1; s >>= 1) {
    if (l < s) {
      sh[0][cl][l] += sh[0][cl][l + s];
      sh[1][cl][l] += sh[1][cl][l + s];
    }
    __syncthreads();
  }
  const double cnt = sh[0][cl][0];
  const double mean_all = cnt > 0.0 ? sh[1][cl][0] / cnt : 0.0;
  __syncthreads();
  double m2 = 0.0;
  if (c < C)
    for (int b = l; b < nblk; b += FL) {
      const long long v0 = (long long)b * per, v1 = v0 + per < rows ? v0 + per : rows;
      if (v0 >= rows) break;
      const double nb = (double)(v1 - v0), inb = (v1 - v0) == per ? inv_per : 1.0 / nb;
      const float K = (float)__builtin_bit_cast(T, *(const unsigned short*)(x + (v0 * C + c) * 2));
      const double s1 = partial[((long long)b * C + c) * 2], s2 = partial[((long long)b * C + c) * 2 + 1];
      const double dm = (double)K + s1 * inb - mean_all;
      m2 += (s2 - s1 * s1 * inb) + nb * dm * dm;
    }
  sh[1][cl][l] = m2;
  __syncthreads();
  for (int s = FL / 2; s >= 1; s >>= 1) {
    if (l < s) sh[1][cl][l] += sh[1][cl][l + s];
    __syncthreads();
  }
  if (threadIdx.x >= FC || c >= C) return;
  m2 = sh[1][cl][0];
  const double mean = mean_all;
  double var = m2 / cnt;
  var = var < 0.0 ? 0.0 : var;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f;
  ab[2 * c] = rstd * g;
  ab[2 * c + 1] = (beta ? beta[c] : 0.f) - (float)mean * rstd * g;
  if (save_mean) save_mean[c] = (float)mean;
  if (save_rstd) save_rstd[c] = rstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(cnt > 1.0 ? m2 / (cnt - 1.0) : var);
}

template <typename T>
__global__ void bn_apply_kernel(const char* __restrict__ x, char* __restrict__ y, const float* __restrict__ ab, long long rows,
                                int C, int act, float slope) {
  const int c8n = C >> 3;
  const long long total = rows * c8n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = idx % c8n;
    float f[8];
    t_unpack8<T>(*(const uint4*)(x + idx * 16), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = f[e] * ab[2 * (c8 * 8 + e)] + ab[2 * (c8 * 8 + e) + 1];
      v = act_fwd(v, act_k(act, slope));
      f[e] = v;
    }
    *(uint4*)(y + idx * 16) = t_pack8<T>(f);
  }
}

// 256 % (C / 8) == 0: a thread keeps one 8-channel group for its whole walk -> coefficients in registers, no division
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fast_kernel(const char* __restrict__ x, char* __restrict__ y,
                                                            const float* __restrict__ ab, long long rows, int C, int act,
                                                            float slope) {
  const int c8n = C >> 3, c8 = threadIdx.x % c8n;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = ab[2 * (c8 * 8 + e)];
    b[e] = ab[2 * (c8 * 8 + e) + 1];
  }
  const long long total = rows * c8n;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    float f[8];
    t_unpack8<T>(*(const uint4*)(x + idx * 16), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = f[e] * a[e] + b[e];
      v = act_fwd(v, act_k(act, slope));
      f[e] = v;
    }
    *(uint4*)(y + idx * 16) = t_pack8<T>(f);
  }
}

// ---------------------------------------------------------------- backward of norm + activation
// partial[blk][c] = (sum dz, sum dz * xhat)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const char* __restrict__ dy, const char* __restrict__ y,
                                                          const char* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ partial,
                                                          long long rows, int C, int act, float slope) {
  extern __shared__ float red[];
  const int c8n = C >> 3;
  const int c8 = threadIdx.x % c8n, vrow = threadIdx.x / c8n, nrow = 256 / c8n;
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long v0 = (long long)blockIdx.x * per, v1 = v0 + per < rows ? v0 + per : rows;
  // y == nullptr: the activation's argument is RECOMPUTED from x, z = a x + b with the forward's coefficients (a = rstd gamma,
  // b = beta - mean rstd gamma, the expressions of bn_finalize_kernel) -- act' only needs its sign, and sign(y) = sign(z) -- so the
  // pass reads two tensors instead of three
  float mu[8], rs[8], af[8], bf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean ? mean[c8 * 8 + e] : 0.f;
    rs[e] = rstd ? rstd[c8 * 8 + e] : 1.f;
    const float gm = gamma ? gamma[c8 * 8 + e] : 1.f;
    af[e] = rs[e] * gm;
    bf[e] = (beta ? beta[c8 * 8 + e] : 0.f) - mu[e] * rs[e] * gm;
  }
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto one_row = [&](const uint4& rg, const uint4& rx, const uint4& ry) {
    float g[8], yy[8], xx[8];
    t_unpack8<T>(rg, g);
    t_unpack8<T>(rx, xx);
    if (y) {
      t_unpack8<T>(ry, yy);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) yy[e] = xx[e] * af[e] + bf[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float dz = g[e];
      dz = act_bwd(dz, yy[e], act_k(act, slope));
      s1[e] += dz;
      s2[e] += dz * (xx[e] - mu[e]) * rs[e];
    }
  };
  if (vrow < nrow) {
    long long v = v0 + vrow;
    for (; v + 3ll * nrow < v1; v += 4ll * nrow) {          // four rows in flight; accumulated in row order
      uint4 rg[4], rx[4], ry[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long o = ((v + (long long)u * nrow) * C + c8 * 8) * 2;
        rg[u] = *(const uint4*)(dy + o);
        rx[u] = *(const uint4*)(x + o);
        ry[u] = y ? *(const uint4*)(y + o) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one_row(rg[u], rx[u], ry[u]);
    }
    for (; v < v1; v += nrow) {
      const long long o = (v * C + c8 * 8) * 2;
      one_row(*(const uint4*)(dy + o), *(const uint4*)(x + o), y ? *(const uint4*)(y + o) : make_uint4(0, 0, 0, 0));
    }
  }
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
    float* o = partial + ((long long)blockIdx.x * C + c) * 2;
    o[0] = a;
    o[1] = b;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int C, int nblk,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double sh[2][8][32];
  const int cl = threadIdx.x & 7, l = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int k = l; k < nblk; k += 32) {
      a += partial[((long long)k * C + c) * 2];
      b += partial[((long long)k * C + c) * 2 + 1];
    }
  sh[0][cl][l] = a;
  sh[1][cl][l] = b;
  __syncthreads();
  for (int s = 16; s >= 1; s >>= 1) {                       // fixed pairwise tree over the 32 lane sums
    if (l < s) {
      sh[0][cl][l] += sh[0][cl][l + s];
      sh[1][cl][l] += sh[1][cl][l + s];
    }
    __syncthreads();
  }
  if (threadIdx.x >= 8 || c >= C) return;
  dbeta[c] = (float)sh[0][cl][0];
  dgamma[c] = (float)sh[1][cl][0];
}

// dx into the interior of the framed buffer [N][D+4][H+4][W+4][C].  One block per (n, z, y) row: the row decomposition
// is scalar, the per-channel coefficients dx = a dz + b x + k (a = gamma rstd, b = -gamma rstd^2 dgamma / M,
// k = -a dbeta / M - b mean) are formed once per block in LDS, a thread only splits its index into (x, 8-channel group).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const char* __restrict__ dy, const char* __restrict__ y,
                                                          const char* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ dgamma,
                                                          const float* __restrict__ dbeta, char* __restrict__ dxf, int N, int D, int H,
                                                          int W, int C, int act, float slope) {
  extern __shared__ float coef[];                            // [C][5]: a, b, k of dx and the forward's (a_f, b_f) for y == nullptr
  const float invM = 1.f / ((float)N * D * H * W);
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 1.f, b = 0.f, k = 0.f, a_f = 1.f, b_f = 0.f;
    if (mean) {
      const float gm = gamma ? gamma[c] : 1.f;
      a = gm * rstd[c];
      b = -a * rstd[c] * dgamma[c] * invM;
      k = -a * dbeta[c] * invM - b * mean[c];
      a_f = rstd[c] * gm;
      b_f = (beta ? beta[c] : 0.f) - mean[c] * rstd[c] * gm;
    }
    coef[5 * c] = a; coef[5 * c + 1] = b; coef[5 * c + 2] = k; coef[5 * c + 3] = a_f; coef[5 * c + 4] = b_f;
  }
  __syncthreads();
  const int row = blockIdx.x;                                // (n * D + z) * H + y
  const int yy0 = row % H, zz0 = (row / H) % D, n = row / (H * D);
  const int c8n = C >> 3, per_row = W * c8n;
  const size_t in_off = (size_t)row * W * C * 2;
  const size_t out_off = ((((size_t)n * (D + 4) + zz0 + 2) * (H + 4) + yy0 + 2) * (W + 4) + 2) * C * 2;
  for (int i = threadIdx.x; i < per_row; i += 256) {
    const int c8 = i % c8n;
    float g[8], yv[8], xv[8];
    t_unpack8<T>(*(const uint4*)(dy + in_off + (size_t)i * 16), g);
    if (mean) t_unpack8<T>(*(const uint4*)(x + in_off + (size_t)i * 16), xv);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = 0.f;              // bare activation adjoint: coefficients are (1, 0, 0)
    }
    if (y) {
      t_unpack8<T>(*(const uint4*)(y + in_off + (size_t)i * 16), yv);
    } else {                                                 // recomputed argument of the activation (see bn_bwd_stats_kernel)
#pragma unroll
      for (int e = 0; e < 8; ++e) yv[e] = xv[e] * coef[5 * (c8 * 8 + e) + 3] + coef[5 * (c8 * 8 + e) + 4];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dz = act_bwd(g[e], yv[e], act_k(act, slope));
      const float* q = coef + 5 * (c8 * 8 + e);
      g[e] = q[0] * dz + q[1] * xv[e] + q[2];               // no per-element test of `mean`
    }
    *(uint4*)(dxf + out_off + (size_t)i * 16) = t_pack8<T>(g);     // i = x * c8n + c8: the row is contiguous in the frame too
  }
}

// ---------------------------------------------------------------- reflect-padding adjoint
// one block per output row (n, z, y): the z / y source lists are scalar, only the x list depends on the thread
template <typename T>
__global__ __launch_bounds__(256) void pad_fold_kernel(const char* __restrict__ gf, char* __restrict__ din, int N, int D, int H,
                                                      int W, int C, int accumulate) {
  const int row = blockIdx.x;
  const int y = row % H, z = (row / H) % D, n = row / (H * D);
  // framed coordinate e = padded coordinate j + 2; voxel i collects j = i, j = -1 (if i == 1), j = L (if i == L-2)
  int ez[3], ey[3], nz = 0, ny = 0;
  ez[nz++] = z + 2; if (z == 1) ez[nz++] = 1; if (z == D - 2) ez[nz++] = D + 2;
  ey[ny++] = y + 2; if (y == 1) ey[ny++] = 1; if (y == H - 2) ey[ny++] = H + 2;
  const int c8n = C >> 3, per_row = W * c8n;
  char* orow = din + (size_t)row * W * C * 2;
  for (int i = threadIdx.x; i < per_row; i += 256) {
    const int c8 = i % c8n, x = i / c8n;
    int ex[3], nx = 0;
    ex[nx++] = x + 2; if (x == 1) ex[nx++] = 1; if (x == W - 2) ex[nx++] = W + 2;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (accumulate) t_unpack8<T>(*(const uint4*)(orow + (size_t)i * 16), acc);
    for (int a = 0; a < nz; ++a)
      for (int b = 0; b < ny; ++b) {
        const char* frow = gf + ((((size_t)n * (D + 4) + ez[a]) * (H + 4) + ey[b]) * (W + 4)) * C * 2 + c8 * 16;
        for (int c = 0; c < nx; ++c) {
          float f[8];
          t_unpack8<T>(*(const uint4*)(frow + (size_t)ex[c] * C * 2), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
      }
    *(uint4*)(orow + (size_t)i * 16) = t_pack8<T>(acc);
  }
}

// ---------------------------------------------------------------- adjoint of cat(skip, nearest_up2(low)) in one pass
// dcat [N][D][H][W][c0 + c1] (the data gradient of a concat conv) -> dskip [N][D][H][W][c0] (= or += the first c0 channels)
// and dlow [N][D/2][H/2][W/2][c1] = sum of the 8 children of every low-resolution voxel (fp32 sum, one rounding).
// One thread = one low-resolution voxel x one 8-channel group of EITHER part: the 8 children are read once.
// FRAMED: dcat is the data-gradient conv's raw result on the padded domain, [N][D+4][H+4][W+4][c0 + c1]; the reflect-padding adjoint
// (pad_fold above) is applied while reading -- a child collects its own position and, next to a face, the mirrored frame
// positions -- so the folded full-resolution tensor is never written.
template <typename T, bool FRAMED>
__global__ void upcat_split_kernel(const char* __restrict__ dcat, char* __restrict__ dskip, char* __restrict__ dlow, int N,
                                   int Dl, int Hl, int Wl, int c0, int c1, int acc_skip) {
  const int C = c0 + c1, g0 = c0 >> 3, g1 = c1 >> 3, G = g0 + g1;
  const long long total = (long long)N * Dl * Hl * Wl * G;
  const int W = 2 * Wl, H = 2 * Hl, D = 2 * Dl;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int g = idx % G;
    long long r = idx / G;
    const int xl = r % Wl;
    r /= Wl;
    const int yl = r % Hl;
    r /= Hl;
    const int zl = r % Dl;
    const int n = r / Dl;
    float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = 2 * zl + (k >> 2), y = 2 * yl + ((k >> 1) & 1), x = 2 * xl + (k & 1);
      const long long v = (((long long)n * D + z) * H + y) * W + x;
      float f[8];
      if (!FRAMED) {
        t_unpack8<T>(*(const uint4*)(dcat + (v * C + g * 8) * 2), f);
      } else {
        // framed coordinate e = padded coordinate j + 2; voxel i collects j = i, j = -1 (if i == 1), j = L (if i == L-2)
        int ez[3], ey[3], ex[3], nz = 0, ny = 0, nx = 0;
        ez[nz++] = z + 2; if (z == 1) ez[nz++] = 1; if (z == D - 2) ez[nz++] = D + 2;
        ey[ny++] = y + 2; if (y == 1) ey[ny++] = 1; if (y == H - 2) ey[ny++] = H + 2;
        ex[nx++] = x + 2; if (x == 1) ex[nx++] = 1; if (x == W - 2) ex[nx++] = W + 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
        for (int a = 0; a < nz; ++a)
          for (int b = 0; b < ny; ++b)
            for (int c = 0; c < nx; ++c) {
              const long long vf = (((long long)n * (D + 4) + ez[a]) * (H + 4) + ey[b]) * (W + 4) + ex[c];
              float t[8];
              t_unpack8<T>(*(const uint4*)(dcat + (vf * C + g * 8) * 2), t);
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] += t[e];
            }
      }
      if (g < g0) {                                           // skip part: plain copy (or accumulate) per full-resolution voxel
        char* o = dskip + (v * c0 + g * 8) * 2;
        if (acc_skip) {
          float old[8];
          t_unpack8<T>(*(const uint4*)o, old);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += old[e];
        }
        *(uint4*)o = t_pack8<T>(f);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] += f[e];
      }
    }
    if (g >= g0) {
      const long long vl = (((long long)n * Dl + zl) * Hl + yl) * Wl + xl;
      *(uint4*)(dlow + (vl * c1 + (g - g0) * 8) * 2) = t_pack8<T>(sum);
    }
  }
}

// ---------------------------------------------------------------- max-pool adjoint
// one thread = one pooled voxel x 8 channels; writes all 8 children of the window (dp at the first maximum, else 0)
template <typename T>
__global__ void pool2_max_bwd_kernel(const char* __restrict__ dp, const char* __restrict__ in, char* __restrict__ din, int N,
                                     int Do, int Ho, int Wo, int C, int accumulate) {
  const int c8n = C >> 3;
  const long long total = (long long)N * Do * Ho * Wo * c8n;
  const int H = 2 * Ho, W = 2 * Wo, D = 2 * Do;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = idx % c8n;
    long long r = idx / c8n;
    const int xo = r % Wo;
    r /= Wo;
    const int yo = r % Ho;
    r /= Ho;
    const int zo = r % Do, n = r / Do;
    float g[8];
    t_unpack8<T>(*(const uint4*)(dp + idx * 16), g);
    float best[8];
    int arg[8];
    float val[8][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long long o = ((((long long)n * D + 2 * zo + (k >> 2)) * H + 2 * yo + ((k >> 1) & 1)) * W + 2 * xo + (k & 1)) * C + c8 * 8;
      t_unpack8<T>(*(const uint4*)(in + o * 2), val[k]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = val[0][e];
      arg[e] = 0;
#pragma unroll
      for (int k = 1; k < 8; ++k)
        if (val[k][e] > best[e]) { best[e] = val[k][e]; arg[e] = k; }    // strict >: the first maximum wins
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long long o = ((((long long)n * D + 2 * zo + (k >> 2)) * H + 2 * yo + ((k >> 1) & 1)) * W + 2 * xo + (k & 1)) * C + c8 * 8;
      float out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (accumulate) t_unpack8<T>(*(const uint4*)(din + o * 2), out);
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] += arg[e] == k ? g[e] : 0.f;
      *(uint4*)(din + o * 2) = t_pack8<T>(out);
    }
  }
}

// ---------------------------------------------------------------- data gradient without the padded-domain detour: the shell terms
// The adjoint of y = conv_valid(reflect_pad(x)) is dx = P^T g with g = the zero-padded correlation of dy with the flipped, transposed
// weights on [-1, n]^3 and P^T the fold dx[1] += g[-1], dx[n-2] += g[n] per axis.  Computing g on the zero-FRAMED (n+4)^3 domain and
// folding it afterwards (pad_fold above) costs, for the level-0 layers of a two-view step, a 132^3 convolution whose 170 tiles run as
// two rounds (97 us against 64 us for the 128^3 forward) plus a 70 us fold pass.  Instead:
//   * g on the INTERIOR [0, n)^3 is an ordinary launch of the forward kernel that reads its halo from the zero frame instead of
//     reflecting (ConvParams::raw_halo) -- the forward's tiles, the forward's time -- written straight into dx;
//   * the SHELL values only ever reach the voxels with a coordinate equal to 1 or n-2, and a shell value has few taps (dy is zero
//     outside the volume).  Per destination m and shell source s (m with one or more coordinates 1 -> -1, n-2 -> n):
//         dx[m][ci] += sum over taps t with s - t inside the volume, over co, of  w[co][ci][t] dy[s - t][co]      (w: the FORWARD weights)
//     A pure face voxel (one such coordinate) has one source with one tap along its axis -- a 3x3 correlation of the dy face plane --
//     the rim voxels of the six near-face planes (edges / corners of the inner box) up to 7 sources of 9, 3 or 1 taps.  All of it runs
//     on the MFMA in dgrad_shell_kernel: a wave owns a tile of 16 voxels of one plane x 16 (32) channels, K = 2 taps x 16 or 1 tap x 32
//     channels per step, the B fragment is one 16-byte global load per lane (no LDS), the A fragments of the plane's own face source
//     stay in registers, those of the rare other sources are gathered from the fp32 weights when a tile needs them.  (A VALU version of
//     the shell took 65 us per 16 -> 16 layer at 128^3 x 2 -- 12 M wave instructions for 0.9 GFLOP; rim voxels as chains of dependent
//     gathers another 38 us.)
//   Every destination is owned by exactly one lane (a voxel belongs to the FIRST of the six planes that contains it), sums in
//   fp32, one read-modify-write of the 16-bit dx: deterministic, no atomics.
// dy: interior view of the framed gradient (byte strides); w fp32 [co_real][ci_real][27]; dx dense [N][D][H][W][CDX] 16-bit.
// Shell sources are numbered in ABSOLUTE axes, combo = 9 oz + 3 oy + ox in 1 .. 26 (per axis 0 keep: taps -1, 0, +1; 1 low: the source
// sits at -1 and only tap -1 reaches the volume; 2 high: only tap +1); tap k of a source enumerates the kept axes' taps, x fastest.
// Its MFMA A fragments -- rows = output channels, K = TPS taps x CDY channels per step -- live in one table per layer,
// [step][m tile][lane][8], built by dgrad_shell_pack_kernel (gathering them from the fp32 [co][ci][27] weights inside the main kernel
// cost 8 strided loads per lane and step: 80 us for a 16 -> 16 layer).
__host__ __device__ constexpr int shell_ntap(int combo) { return (combo / 9 ? 1 : 3) * ((combo / 3) % 3 ? 1 : 3) * (combo % 3 ? 1 : 3); }
__host__ __device__ constexpr int shell_steps(int combo, int tps) { return (shell_ntap(combo) + tps - 1) / tps; }
__host__ __device__ constexpr int shell_base(int combo, int tps) {      // steps of the combos before this one
  int b = 0;
  for (int c = 1; c < combo; ++c) b += shell_steps(c, tps);
  return b;
}
__host__ __device__ inline int shell_tap(int combo, int k) {            // -> (tz << 8) | (ty << 4) | tx, each in 0 .. 2
  // divisions by the constant 3 only: with run-time divisors (1 or 3 taps per axis) the three divisions per step were most of the
  // shell kernel's VALU work
  const int oz = combo / 9, oy = (combo / 3) % 3, ox = combo % 3;
  const int k3 = k / 3;
  const int ix = ox ? 0 : k - 3 * k3, k1 = ox ? k : k3;
  const int k13 = k1 / 3;
  const int iy = oy ? 0 : k1 - 3 * k13, iz = oy ? k1 : k13;
  return ((oz == 0 ? iz : (oz == 1 ? 0 : 2)) << 8) | ((oy == 0 ? iy : (oy == 1 ? 0 : 2)) << 4) | (ox == 0 ? ix : (ox == 1 ? 0 : 2));
}
struct ShellBaseTab {                      // shell_base for both step widths, built at compile time
  unsigned char v[2][28];
  constexpr ShellBaseTab() : v() {
    for (int t = 0; t < 2; ++t)
      for (int c = 0; c < 28; ++c) v[t][c] = (unsigned char)shell_base(c, t + 1);
  }
};
__device__ constexpr ShellBaseTab kShellBase{};

template <typename T, int CDY, int CDX>
__global__ __launch_bounds__(64) void dgrad_shell_pack_kernel(const float* __restrict__ w, int co_real, int ci_real, T* __restrict__ tab) {
  constexpr int TPS = 32 / CDY, MT = CDX / 16;
  const int step = blockIdx.x / MT, mt = blockIdx.x % MT;
  int combo = 1, base = 0;
  while (combo < 26 && base + shell_steps(combo, TPS) <= step) base += shell_steps(combo++, TPS);
  const int j = step - base, lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  const int k = TPS == 2 ? 2 * j + (g >> 1) : j, ntap = shell_ntap(combo);
  const int co0 = TPS == 2 ? (g & 1) * 8 : g * 8, ci = mt * 16 + li;
  const int tp = shell_tap(combo, k < ntap ? k : 0);
  const int tap = ((tp >> 8) * 3 + ((tp >> 4) & 15)) * 3 + (tp & 15);
  T* o = tab + ((long long)blockIdx.x * 64 + lane) * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int co = co0 + e;
    o[e] = (T)((k < ntap && co < co_real && ci < ci_real) ? w[((long long)co * ci_real + ci) * 27 + tap] : 0.f);
  }
}

template <typename T, int CDY, int CDX>
__global__ __launch_bounds__(256) void dgrad_shell_kernel(const char* __restrict__ dy, long long yn, long long yz, long long yy, long long yx,
                                                          const T* __restrict__ tab, char* __restrict__ dx, int D, int H, int W) {
  typedef typename Ops<T>::vec8 vec8;
  constexpr int TPS = 32 / CDY;                       // taps per MFMA step: K = 32 = TPS taps x CDY channels
  constexpr int NSTEP = (9 + TPS - 1) / TPS;          // steps of a nine-tap source
  constexpr int MT = CDX / 16;
  // grid: x = chunks of the plane's 16-voxel tiles, y = near-face plane (axis a = y >> 1, coordinate 1 or n_a - 2), z = sample.
  // A voxel is owned by the FIRST plane of that order containing it.
  const int pl = blockIdx.y, a = pl >> 1, side = pl & 1, n = blockIdx.z;
  const int dims[3] = {D, H, W};
  const int b0 = a == 0 ? 1 : 0, b1 = a == 2 ? 1 : 2;
  const int n0 = dims[b0], n1 = dims[b1], na = dims[a];
  const int ma = side ? na - 2 : 1;
  if (ma < 0 || ma >= na || (side && na - 2 == 1)) return;       // (n_a = 3: both sides are the same plane, visited once)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int co0 = TPS == 2 ? (g & 1) * 8 : g * 8;
  const long long ystr[3] = {yz, yy, yx};
  const long long dstr[3] = {(long long)H * W * CDX * 2, (long long)W * CDX * 2, (long long)CDX * 2};
  const vec8* ftab = (const vec8*)tab + lane;
  auto combo_of = [&](int oa, int o0, int o1) {
    int o3[3];
    o3[a] = oa; o3[b0] = o0; o3[b1] = o1;
    return o3[0] * 9 + o3[1] * 3 + o3[2];
  };
  // the plane's own face source (s_a = -1 or n_a, the other axes kept): the only source of all but its rim voxels -- kept in registers
  const int oa_face = side ? 2 : 1;
  const int face_combo = combo_of(oa_face, 0, 0), face_base = kShellBase.v[TPS - 1][face_combo];
  vec8 fa_face[NSTEP][MT];
#pragma unroll
  for (int j = 0; j < NSTEP; ++j)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fa_face[j][mt] = ftab[((face_base + j) * MT + mt) * 64];

  const int tpr = (n1 + 15) >> 4, ntile = n0 * tpr;
  for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
    const int m0 = tile / tpr, m1 = (tile - m0 * tpr) * 16 + li;
    int m[3];
    m[a] = ma; m[b0] = m0; m[b1] = m1;
    bool own = m1 < n1;
    for (int q = 0; q < pl; ++q) own &= !(m[q >> 1] == ((q & 1) ? dims[q >> 1] - 2 : 1));   // an earlier plane owns it
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the destination's current value is requested now, next to the fragments (a tile is two memory round trips otherwise)
    char* d = dx + (((long long)n * D * H * W) * CDX * 2) + ma * dstr[a] + m0 * dstr[b0] + (own ? m1 : 0) * dstr[b1] + g * 8;   // channels 4 g .. 4 g + 3
    uint2 oldv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) oldv[mt] = own ? *(const uint2*)(d + mt * 32) : make_uint2(0, 0);
    // every shell source: per axis keep / low (m == 1) / high (m == n - 2), not all three "keep"
    for (int oa = 0; oa < 3; ++oa) {
      if ((oa == 1 && ma != 1) || (oa == 2 && ma != na - 2)) continue;
      for (int o0 = 0; o0 < 3; ++o0) {
        if ((o0 == 1 && m0 != 1) || (o0 == 2 && m0 != n0 - 2)) continue;
        for (int o1 = 0; o1 < 3; ++o1) {
          if (oa == 0 && o0 == 0 && o1 == 0) continue;
          const bool lv = own && (o1 == 0 || (o1 == 1 ? m1 == 1 : m1 == n1 - 2));
          if (!__any(lv)) continue;                            // wave-uniform
          const int combo = combo_of(oa, o0, o1), ntap = shell_ntap(combo);
          const int sa = oa == 0 ? ma : (oa == 1 ? -1 : na), s0 = o0 == 0 ? m0 : (o0 == 1 ? -1 : n0), s1 = o1 == 0 ? m1 : (o1 == 1 ? -1 : n1);
          const bool face = combo == face_combo;
          const int nstep = (ntap + TPS - 1) / TPS, base = kShellBase.v[TPS - 1][combo];
          // all of the source's fragments are requested before the first MFMA (a load -> MFMA chain per step made every tile a
          // string of memory latencies)
          vec8 fbs[NSTEP], fas[NSTEP][MT];
#pragma unroll
          for (int j = 0; j < NSTEP; ++j) {
            const int k = TPS == 2 ? 2 * j + (g >> 1) : j;
            const int tp = shell_tap(combo, k < ntap ? k : 0);
            const int t3[3] = {tp >> 8, (tp >> 4) & 15, tp & 15};
            const int qa = sa - (t3[a] - 1), q0 = s0 - (t3[b0] - 1), q1 = s1 - (t3[b1] - 1);
            const bool ok = lv && j < nstep && k < ntap && qa >= 0 && qa < na && q0 >= 0 && q0 < n0 && q1 >= 0 && q1 < n1;
            const char* src = dy + (long long)n * yn + (ok ? qa : 0) * ystr[a] + (ok ? q0 : 0) * ystr[b0] + (ok ? q1 : 0) * ystr[b1] + co0 * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) fbs[j][e] = (T)0.f;
            if (ok) fbs[j] = *(const vec8*)src;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fas[j][mt] = face ? fa_face[j][mt] : ftab[((base + (j < nstep ? j : 0)) * MT + mt) * 64];
          }
#pragma unroll
          for (int j = 0; j < NSTEP; ++j) {          // (steps past the source's last one multiply zero B fragments)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = Ops<T>::mfma(fas[j][mt], fbs[j], acc[mt]);
          }
        }
      }
    }
    if (!own) continue;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      uint2* p2 = (uint2*)(d + mt * 32);
      const uint2 old = oldv[mt];
      const float a0 = (float)__builtin_bit_cast(T, (unsigned short)(old.x & 0xffff)) + acc[mt][0];
      const float a1 = (float)__builtin_bit_cast(T, (unsigned short)(old.x >> 16)) + acc[mt][1];
      const float a2 = (float)__builtin_bit_cast(T, (unsigned short)(old.y & 0xffff)) + acc[mt][2];
      const float a3 = (float)__builtin_bit_cast(T, (unsigned short)(old.y >> 16)) + acc[mt][3];
      *p2 = make_uint2((unsigned)to_bits<T>(a0) | ((unsigned)to_bits<T>(a1) << 16), (unsigned)to_bits<T>(a2) | ((unsigned)to_bits<T>(a3) << 16));
    }
  }
}

size_t dgrad_shell_scratch_bytes() { return (size_t)shell_base(27, 1) * 2 * 1024; }     // the largest fragment table (one tap per step, two m tiles)

hipError_t launch_dgrad_fold_shell(const void* dy, long long yn, long long yz, long long yy, long long yx, int cdy, const float* w,
                                   int co_real, int ci_real, void* dx, int cdx, int N, int D, int H, int W, int precision, void* scratch,
                                   hipStream_t st) {
  if ((cdy != 16 && cdy != 32) || (cdx != 16 && cdx != 32) || precision > 1 || !scratch) return hipErrorInvalidValue;
  long long mx = (long long)H * W;
  if ((long long)D * W > mx) mx = (long long)D * W;
  if ((long long)D * H > mx) mx = (long long)D * H;
  long long bx = ((mx + 15) / 16 + 3) / 4;                   // four waves per block, ONE 16-voxel tile per wave: a tile is a latency chain
  if (bx < 1) bx = 1;                                        // (fragments + old value -> MFMAs -> store), so they all go in flight at once
  if (bx > 1024) bx = 1024;
  const dim3 gf((unsigned)bx, 6, (unsigned)N);
  const int nsteps = shell_base(27, 32 / cdy), mt = cdx / 16;
#define AMX_SHELL(T, CDY, CDX)                                                                                                \
  {                                                                                                                           \
    hipLaunchKernelGGL((dgrad_shell_pack_kernel<T, CDY, CDX>), dim3(nsteps * mt), dim3(64), 0, st, w, co_real, ci_real, (T*)scratch);        \
    hipLaunchKernelGGL((dgrad_shell_kernel<T, CDY, CDX>), gf, dim3(256), 0, st, (const char*)dy, yn, yz, yy, yx, (const T*)scratch, (char*)dx, \
                       D, H, W);                                                                                              \
  }
#define AMX_SHELL_T(T)                                                                                                        \
  {                                                                                                                           \
    if (cdy == 16 && cdx == 16) AMX_SHELL(T, 16, 16)                                                                          \
    else if (cdy == 16) AMX_SHELL(T, 16, 32)                                                                                  \
    else if (cdx == 16) AMX_SHELL(T, 32, 16)                                                                                  \
    else AMX_SHELL(T, 32, 32)                                                                                                 \
  }
  if (precision == 0) AMX_SHELL_T(f16) else AMX_SHELL_T(bf16)
#undef AMX_SHELL_T
#undef AMX_SHELL
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------- launchers
size_t train_scratch_bytes(int C) { return ((size_t)65536 * 2 + (size_t)C * 2) * sizeof(float); }

static int grid_for(long long total) { return (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256); }

hipError_t launch_bn_train_forward(const void* x, void* y, const float* gamma, const float* beta, float eps, long long rows, int C,
                                   int act, float slope, void* scratch, float* save_mean, float* save_rstd, float* running_mean,
                                   float* running_var, float momentum, int precision, hipStream_t st) {
  if (C % 8 || C > 2048 || rows < 1) return hipErrorInvalidValue;
  float* partial = (float*)scratch;
  float* ab = partial + (size_t)65536 * 2;
  const int nblk = tr_num_blocks(rows, C), c8n = C / 8, nrow = 256 / c8n;
  const size_t lds = (size_t)nrow * C * 2 * sizeof(float);
  const int blocks = grid_for(rows * c8n);
#define AMX_BN(T)                                                                                                        \
  hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(nblk), dim3(256), lds, st, (const char*)x, partial, rows, C);              \
  hipLaunchKernelGGL(bn_finalize_kernel<T>, dim3((C + 1) / 2), dim3(256), 0, st, (const char*)x, partial, gamma, beta, eps, \
                     rows, C, nblk, ab, save_mean, save_rstd, running_mean, running_var, momentum);                      \
  if (256 % c8n == 0)                                                                                                    \
    hipLaunchKernelGGL(bn_apply_fast_kernel<T>, dim3(blocks), dim3(256), 0, st, (const char*)x, (char*)y, ab, rows, C, act, slope); \
  else                                                                                                                   \
    hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(blocks), dim3(256), 0, st, (const char*)x, (char*)y, ab, rows, C, act, slope)
  if (precision == 0) { AMX_BN(f16); } else { AMX_BN(bf16); }
#undef AMX_BN
  return hipGetLastError();
}

hipError_t launch_bn_act_backward(const void* dy, const void* y, const void* x, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, float* dgamma, float* dbeta, void* dx_framed, int N, int D,
                                  int H, int W, int C, int act, float slope, void* scratch, int precision, hipStream_t st) {
  if (!y && !mean) return hipErrorInvalidValue;             // the activation's argument is either read (y) or recomputed from x
  if (C % 8 || C > 2048 || (long long)N * D * H * W * (C / 8) >= (1ll << 31)) return hipErrorInvalidValue;
  const long long rows = (long long)N * D * H * W;
  float* partial = (float*)scratch;
  const int nblk = tr_num_blocks(rows, C), c8n = C / 8, nrow = 256 / c8n;
  const size_t lds = (size_t)nrow * C * 2 * sizeof(float);
#define AMX_BNB(T)                                                                                                       \
  if (mean) {                                                                                                            \
    hipLaunchKernelGGL(bn_bwd_stats_kernel<T>, dim3(nblk), dim3(256), lds, st, (const char*)dy, (const char*)y,          \
                       (const char*)x, mean, rstd, gamma, beta, partial, rows, C, act, slope);                           \
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, st, partial, C, nblk, dgamma, dbeta);    \
  }                                                                                                                      \
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3((unsigned)(N * D * H)), dim3(256), (size_t)C * 5 * sizeof(float), st,  \
                     (const char*)dy, (const char*)y, (const char*)x, mean, rstd, gamma, beta, dgamma, dbeta, (char*)dx_framed, N, D, \
                     H, W, C, act, slope)
  if (precision == 0) { AMX_BNB(f16); } else { AMX_BNB(bf16); }
#undef AMX_BNB
  return hipGetLastError();
}

hipError_t launch_pad_fold(const void* g_framed, void* din, int N, int D, int H, int W, int C, int accumulate, int precision,
                           hipStream_t st) {
  if (C % 8 || D < 2 || H < 2 || W < 2 || (long long)N * D * H * W * (C / 8) >= (1ll << 31)) return hipErrorInvalidValue;
  const unsigned blocks = (unsigned)(N * D * H);
  if (precision == 0)
    hipLaunchKernelGGL(pad_fold_kernel<f16>, dim3(blocks), dim3(256), 0, st, (const char*)g_framed, (char*)din, N, D, H, W, C, accumulate);
  else
    hipLaunchKernelGGL(pad_fold_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const char*)g_framed, (char*)din, N, D, H, W, C, accumulate);
  return hipGetLastError();
}

hipError_t launch_pool2_max_backward(const void* dp, const void* in, void* din, int N, int Do, int Ho, int Wo, int C,
                                     int accumulate, int precision, hipStream_t st) {
  if (C % 8) return hipErrorInvalidValue;
  const int blocks = grid_for((long long)N * Do * Ho * Wo * (C / 8));
  if (precision == 0)
    hipLaunchKernelGGL(pool2_max_bwd_kernel<f16>, dim3(blocks), dim3(256), 0, st, (const char*)dp, (const char*)in, (char*)din, N,
                       Do, Ho, Wo, C, accumulate);
  else
    hipLaunchKernelGGL(pool2_max_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const char*)dp, (const char*)in, (char*)din, N,
                       Do, Ho, Wo, C, accumulate);
  return hipGetLastError();
}

hipError_t launch_upcat_split(const void* dcat, void* dskip, void* dlow, int N, int Dl, int Hl, int Wl, int c0, int c1,
                              int acc_skip, int framed, int precision, hipStream_t st) {
  if (c0 % 8 || c1 % 8 || c0 < 0 || c1 < 8) return hipErrorInvalidValue;       // c0 == 0: no skip part, everything is summed over children
  const int blocks = grid_for((long long)N * Dl * Hl * Wl * ((c0 + c1) / 8));
#define AMX_UPS(T, F)                                                                                                            \
  hipLaunchKernelGGL((upcat_split_kernel<T, F>), dim3(blocks), dim3(256), 0, st, (const char*)dcat, (char*)dskip, (char*)dlow, N, Dl, \
                     Hl, Wl, c0, c1, acc_skip)
  if (precision == 0) {
    if (framed) { AMX_UPS(f16, true); } else { AMX_UPS(f16, false); }
  } else {
    if (framed) { AMX_UPS(bf16, true); } else { AMX_UPS(bf16, false); }
  }
#undef AMX_UPS
  return hipGetLastError();
}

}  // namespace amx
