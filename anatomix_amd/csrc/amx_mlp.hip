// anatomix_amd -- projection head of the patch sampler, forward and backward
// (reference pretraining/models/pretraining_networks.py:338-350: mlp_k = Linear(C, nc, bias=False) - BatchNorm1d - act
//  [- Linear - BatchNorm1d - act] - Linear - BatchNorm1d(affine=False), applied to [views * patches, C] samples in train
//  mode, :505-511).  fp32 throughout, every reduction in a fixed order.
//
// The arithmetic is tiny (3 GEMMs of 1024 x 256 x <=256 per head, six heads per step); what costs the reference -- and the
// stock-module path here -- is ~50 small launches per head and direction, each paced by the host.  One layer forward is a
// small register-tiled fp32 GEMM plus ONE column-owned kernel (a block owns 8 output columns for ALL rows) that does the
// batch statistics of BatchNorm1d, the normalisation, the activation and the running-stat update from registers.
// Backward per layer: the activation + BatchNorm adjoint (column-owned, one pass), dW = dZ^T X (row-split + ordered sum)
// and dX = dZ W.  (A first version fused the forward GEMM into the column-owned kernel: 32 blocks, each thread walking
// its own rows -> 8x cache-line over-fetch and 50 us per layer; the split version is ~4x faster.)
#include "amx_device.h"

namespace amx {

constexpr int MLP_CB = 8;       // output columns per block of the column-owned kernels

// Batches: the six heads of a contrastive step (and their six losses) are the same chains of small kernels on different tensors; a
// replayed graph pays ~8 us per DEPENDENT node whatever its size, so one launch does a chain step for every head (blockIdx.y / z picks
// the head), 6x fewer nodes on the step's critical path.  Same kernels, same arithmetic per head: bit-identical to head-by-head launches.
struct GemmBatch {
  int nb, splits;
  float scale;
  const float* A[MLP_MAXB];
  const float* B[MLP_MAXB];
  float* C[MLP_MAXB];
  int M[MLP_MAXB], N[MLP_MAXB], R[MLP_MAXB], rps[MLP_MAXB];
};
struct BnFwdBatch {
  const float* z[MLP_MAXB];
  const float* gamma[MLP_MAXB];
  const float* beta[MLP_MAXB];
  float* y[MLP_MAXB];
  float* mean[MLP_MAXB];
  float* rstd[MLP_MAXB];
  float* rmean[MLP_MAXB];
  float* rvar[MLP_MAXB];
};
struct BnBwdBatch {
  const float* dy[MLP_MAXB];
  const float* y[MLP_MAXB];
  const float* z[MLP_MAXB];
  const float* mean[MLP_MAXB];
  const float* rstd[MLP_MAXB];
  const float* gamma[MLP_MAXB];
  float* dz[MLP_MAXB];
  float* dgamma[MLP_MAXB];
  float* dbeta[MLP_MAXB];
};
struct ReduceBatch {
  const float* part[MLP_MAXB];
  float* out[MLP_MAXB];
  int count[MLP_MAXB];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// block-wide sums of CB values per thread, fixed order (wave butterflies, then the four waves in order)
template <int NV>
__device__ __forceinline__ void block_sums(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const float s = wave_sum(v[c]);
    if (lane == 0) red[wv * NV + c] = s;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NV; ++c) v[c] = ((red[c] + red[NV + c]) + red[2 * NV + c]) + red[3 * NV + c];
}

__device__ __forceinline__ float act_grad(float y, int act, float slope) {   // branch-free: one select on a per-kernel constant
  return y > 0.f ? 1.f : act_k(act, slope);
}

// BatchNorm1d with batch statistics over the n rows of z [n][m] + activation; a block owns 8 columns for all rows, so the
// two-pass statistics (mean, then centred sum of squares) come from registers.
template <int RPT>
__global__ __launch_bounds__(256) void mlp_bn_fwd_kernel(const BnFwdBatch bt, int n, int m, float eps, int act, float slope,
                                                         float momentum) {
  __shared__ float red[4 * MLP_CB];
  const int hb = blockIdx.y;
  const float* __restrict__ z = bt.z[hb];
  const float* __restrict__ gamma = bt.gamma[hb];
  const float* __restrict__ beta = bt.beta[hb];
  float* __restrict__ y = bt.y[hb];
  float* __restrict__ mean_out = bt.mean[hb];
  float* __restrict__ rstd_out = bt.rstd[hb];
  float* __restrict__ rmean = bt.rmean[hb];
  float* __restrict__ rvar = bt.rvar[hb];
  const int j0 = blockIdx.x * MLP_CB;
  float v[RPT][MLP_CB];
  float s[MLP_CB];
#pragma unroll
  for (int c = 0; c < MLP_CB; ++c) s[c] = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + 256 * i;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (r < n) {
      const float4* p = (const float4*)(z + (size_t)r * m + j0);
      a0 = p[0];
      a1 = p[1];
    }
    v[i][0] = a0.x; v[i][1] = a0.y; v[i][2] = a0.z; v[i][3] = a0.w; v[i][4] = a1.x; v[i][5] = a1.y; v[i][6] = a1.z; v[i][7] = a1.w;
#pragma unroll
    for (int c = 0; c < MLP_CB; ++c) s[c] += v[i][c];        // rows >= n hold zeros
  }
  block_sums<MLP_CB>(s, red);
  float mu[MLP_CB], q[MLP_CB];
#pragma unroll
  for (int c = 0; c < MLP_CB; ++c) {
    mu[c] = s[c] / (float)n;
    q[c] = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const float dlt = v[i][c] - mu[c];
      q[c] += (threadIdx.x + 256 * i < n) ? dlt * dlt : 0.f;
    }
  }
  block_sums<MLP_CB>(q, red);
#pragma unroll
  for (int c = 0; c < MLP_CB; ++c) {
    const float var = q[c] / (float)n;                 // biased variance normalises (BatchNorm, training)
    const float rs = rsqrtf(var + eps);
    if ((int)threadIdx.x == c) {
      mean_out[j0 + c] = mu[c];
      rstd_out[j0 + c] = rs;
      if (rmean) {                                      // running stats: unbiased variance, momentum as nn.BatchNorm1d
        rmean[j0 + c] = (1.f - momentum) * rmean[j0 + c] + momentum * mu[c];
        rvar[j0 + c] = (1.f - momentum) * rvar[j0 + c] + momentum * (n > 1 ? q[c] / (float)(n - 1) : var);
      }
    }
    q[c] = rs;
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + 256 * i;
    if (r >= n) continue;
    float yo[MLP_CB];
#pragma unroll
    for (int c = 0; c < MLP_CB; ++c) {
      float o = (v[i][c] - mu[c]) * q[c];
      if (gamma) o = o * gamma[j0 + c] + beta[j0 + c];
      yo[c] = act_fwd(o, act_k(act, slope));
    }
    float4* yp = (float4*)(y + (size_t)r * m + j0);
    yp[0] = make_float4(yo[0], yo[1], yo[2], yo[3]);
    yp[1] = make_float4(yo[4], yo[5], yo[6], yo[7]);
  }
}

// adjoint of activation + BatchNorm1d(train):  g = dy * act'(y);  dbeta = sum g;  dgamma = sum g xh;
// dz = rstd * gamma * (g - dbeta / n - xh * dgamma / n),  xh = (z - mean) * rstd
template <int RPT>
__global__ __launch_bounds__(256) void mlp_bwd_norm_kernel(const BnBwdBatch bt, int act, float slope, int n, int m) {
  __shared__ float red[4 * 2 * MLP_CB];
  const int hb = blockIdx.y;
  const float* __restrict__ dy = bt.dy[hb];
  const float* __restrict__ y = bt.y[hb];
  const float* __restrict__ z = bt.z[hb];
  const float* __restrict__ mean = bt.mean[hb];
  const float* __restrict__ rstd = bt.rstd[hb];
  const float* __restrict__ gamma = bt.gamma[hb];
  float* __restrict__ dz = bt.dz[hb];
  float* __restrict__ dgamma = bt.dgamma[hb];
  float* __restrict__ dbeta = bt.dbeta[hb];
  const int j0 = blockIdx.x * MLP_CB;
  float mu[MLP_CB], rs[MLP_CB];
#pragma unroll
  for (int c = 0; c < MLP_CB; ++c) {
    mu[c] = mean[j0 + c];
    rs[c] = rstd[j0 + c];
  }
  float g[RPT][MLP_CB], xh[RPT][MLP_CB];
  float sums[2 * MLP_CB];
#pragma unroll
  for (int c = 0; c < 2 * MLP_CB; ++c) sums[c] = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + 256 * i;
    float dv[MLP_CB], yv[MLP_CB], zv[MLP_CB];
    if (r < n) {
      const float4* a = (const float4*)(dy + (size_t)r * m + j0);
      const float4* b = (const float4*)(y + (size_t)r * m + j0);
      const float4* cc = (const float4*)(z + (size_t)r * m + j0);
      const float4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1], c0 = cc[0], c1 = cc[1];
      dv[0] = a0.x; dv[1] = a0.y; dv[2] = a0.z; dv[3] = a0.w; dv[4] = a1.x; dv[5] = a1.y; dv[6] = a1.z; dv[7] = a1.w;
      yv[0] = b0.x; yv[1] = b0.y; yv[2] = b0.z; yv[3] = b0.w; yv[4] = b1.x; yv[5] = b1.y; yv[6] = b1.z; yv[7] = b1.w;
      zv[0] = c0.x; zv[1] = c0.y; zv[2] = c0.z; zv[3] = c0.w; zv[4] = c1.x; zv[5] = c1.y; zv[6] = c1.z; zv[7] = c1.w;
    }
#pragma unroll
    for (int c = 0; c < MLP_CB; ++c) {
      g[i][c] = r < n ? dv[c] * act_grad(yv[c], act, slope) : 0.f;
      xh[i][c] = r < n ? (zv[c] - mu[c]) * rs[c] : 0.f;
      sums[c] += g[i][c];
      sums[MLP_CB + c] += g[i][c] * xh[i][c];
    }
  }
  block_sums<2 * MLP_CB>(sums, red);
  if (dgamma) {
#pragma unroll
    for (int c = 0; c < MLP_CB; ++c)
      if ((int)threadIdx.x == c) {
        dbeta[j0 + c] = sums[c];
        dgamma[j0 + c] = sums[MLP_CB + c];
      }
  }
  const float inv = 1.f / (float)n;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + 256 * i;
    if (r >= n) continue;
    float o[MLP_CB];
#pragma unroll
    for (int c = 0; c < MLP_CB; ++c) {
      const float gm = gamma ? gamma[j0 + c] : 1.f;
      o[c] = rs[c] * gm * (g[i][c] - sums[c] * inv - xh[i][c] * sums[MLP_CB + c] * inv);
    }
    float4* p = (float4*)(dz + (size_t)r * m + j0);
    p[0] = make_float4(o[0], o[1], o[2], o[3]);
    p[1] = make_float4(o[4], o[5], o[6], o[7]);
  }
}

// C[M][N] (+ blockIdx.z * M * N) = sum over this block's r range of A(m, r) B(r, n).
//   TA = false: A stored [R][M];  TA = true: A stored [M][R].   TB = false: B stored [R][N];  TB = true: B stored [N][R].
// 64 x 64 output tile per block, 4 x 4 per thread, r in chunks of 16 staged through LDS with the next chunk prefetched into
// registers while the current one is multiplied.  The contiguous storage dimension of every operand is a feature count
// (multiple of 4), so all global accesses are aligned float4.  blockIdx.z splits R (partials summed by mlp_reduce_kernel).
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void mlp_gemm_kernel(const GemmBatch g) {
  __shared__ float As[16][68], Bs[16][68];
  const int hb = blockIdx.z / g.splits, sp = blockIdx.z - hb * g.splits;
  const float* __restrict__ A = g.A[hb];
  const float* __restrict__ B = g.B[hb];
  float* __restrict__ Cm = g.C[hb];
  const int M = g.M[hb], N = g.N[hb], R = g.R[hb], r_per_split = g.rps[hb];
  const float scale = g.scale;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  if (m0 >= M || n0 >= N) return;                              // (the grid covers the largest problem of the batch)
  const int rbeg = sp * r_per_split, rend = min(R, rbeg + r_per_split);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  auto fetch = [&](const float* P, bool T, int o0, int O, int r0) -> float4 {
    // T == false: storage [R][O]: thread -> (r = t / 16, o = (t % 16) * 4);  T == true: storage [O][R]: (o = t / 4, r = (t % 4) * 4)
    if (!T) {
      const int r = r0 + (threadIdx.x >> 4), o = o0 + (threadIdx.x & 15) * 4;
      return (r < rend && o < O) ? *(const float4*)(P + (size_t)r * O + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int o = o0 + (threadIdx.x >> 2), r = r0 + (threadIdx.x & 3) * 4;
    if (o >= O || r >= rend) return make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v = *(const float4*)(P + (size_t)o * R + r);       // R % 4 == 0: either the whole float4 is inside R ...
    if (r + 1 >= rend) v.y = 0.f;                             // ... or this block's split ends inside it
    if (r + 2 >= rend) v.z = 0.f;
    if (r + 3 >= rend) v.w = 0.f;
    return v;
  };
  auto stash = [&](float (*S)[68], bool T, float4 v) {
    if (!T) {
      *(float4*)&S[threadIdx.x >> 4][(threadIdx.x & 15) * 4] = v;
    } else {
      const int o = threadIdx.x >> 2, r = (threadIdx.x & 3) * 4;
      S[r][o] = v.x;
      S[r + 1][o] = v.y;
      S[r + 2][o] = v.z;
      S[r + 3][o] = v.w;
    }
  };
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  float4 pa = fetch(A, TA, m0, M, rbeg), pb = fetch(B, TB, n0, N, rbeg);
  for (int r0 = rbeg; r0 < rend; r0 += 16) {
    __syncthreads();
    stash(As, TA, pa);
    stash(Bs, TB, pb);
    __syncthreads();
    if (r0 + 16 < rend) {
      pa = fetch(A, TA, m0, M, r0 + 16);
      pb = fetch(B, TB, n0, N, r0 + 16);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 a = *(const float4*)&As[r][ty * 4];
      const float4 b = *(const float4*)&Bs[r][tx * 4];
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
  }
  float* Cz = Cm + (size_t)sp * M * N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mm = m0 + ty * 4 + i, nn = n0 + tx * 4;
    if (mm < M && nn < N)
      *(float4*)(Cz + (size_t)mm * N + nn) = make_float4(scale * acc[i][0], scale * acc[i][1], scale * acc[i][2], scale * acc[i][3]);
  }
}

// The same product on 32 x 32 tiles, 2 x 2 outputs per thread: four times the workgroups for the shapes where the 64 x 64 kernel
// leaves most of the 256 CUs idle (1024 x 256 x 256: 64 workgroups, each serialising 16 chunks of 256 FMAs per thread).
// Threads 0..127 stage the A chunk, 128..255 the B chunk (one float4 each).
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void mlp_gemm32_kernel(const GemmBatch g) {
  // KC = 64 reduction rows per stage (four 16-row sub-chunks): a stage is one global fetch -> LDS -> two barriers round trip, and
  // with 16-row stages a 1024 x 256 x 256 product was 16 such round trips of ~1 us around 64 FMAs per thread (18 us per launch,
  // 72 launches per contrastive step: profiles/r05 step kernel stats).  Same products added in the same order: bit-identical.
  constexpr int KC = 64;
  __shared__ float As[KC][36], Bs[KC][36];
  const int hb = blockIdx.z / g.splits, sp = blockIdx.z - hb * g.splits;
  const float* __restrict__ A = g.A[hb];
  const float* __restrict__ B = g.B[hb];
  float* __restrict__ Cm = g.C[hb];
  const int M = g.M[hb], N = g.N[hb], R = g.R[hb], r_per_split = g.rps[hb];
  const float scale = g.scale;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  if (m0 >= M || n0 >= N) return;
  const int rbeg = sp * r_per_split, rend = min(R, rbeg + r_per_split);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const bool isb = threadIdx.x >= 128;
  const int t = threadIdx.x & 127;
  const float* P = isb ? B : A;
  const bool T = isb ? TB : TA;
  const int o0 = isb ? n0 : m0, O = isb ? N : M;
  auto fetch = [&](int r0) -> float4 {                        // one 16-row sub-chunk
    if (!T) {                                                 // storage [R][O]
      const int r = r0 + (t >> 3), o = o0 + (t & 7) * 4;
      return (r < rend && o < O) ? *(const float4*)(P + (size_t)r * O + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int o = o0 + (t >> 2), r = r0 + (t & 3) * 4;         // storage [O][R]
    if (o >= O || r >= rend) return make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v = *(const float4*)(P + (size_t)o * R + r);
    if (r + 1 >= rend) v.y = 0.f;
    if (r + 2 >= rend) v.z = 0.f;
    if (r + 3 >= rend) v.w = 0.f;
    return v;
  };
  auto stash = [&](int sub, float4 v) {
    float(*S)[36] = (isb ? Bs : As) + sub * 16;
    if (!T) {
      *(float4*)&S[t >> 3][(t & 7) * 4] = v;
    } else {
      const int o = t >> 2, r = (t & 3) * 4;
      S[r][o] = v.x;
      S[r + 1][o] = v.y;
      S[r + 2][o] = v.z;
      S[r + 3][o] = v.w;
    }
  };
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  float4 pv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) pv[u] = fetch(rbeg + 16 * u);
  for (int r0 = rbeg; r0 < rend; r0 += KC) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) stash(u, pv[u]);
    __syncthreads();
    if (r0 + KC < rend) {
#pragma unroll
      for (int u = 0; u < 4; ++u) pv[u] = fetch(r0 + KC + 16 * u);
    }
    const int nr = min(KC, rend - r0);                        // (rows past rend were stashed as zeros: adding them changes nothing,
#pragma unroll 16                                             //  but a short tail need not be walked)
    for (int r = 0; r < nr; ++r) {
      const float2 a = *(const float2*)&As[r][ty * 2];
      const float2 b = *(const float2*)&Bs[r][tx * 2];
      acc[0][0] += a.x * b.x;
      acc[0][1] += a.x * b.y;
      acc[1][0] += a.y * b.x;
      acc[1][1] += a.y * b.y;
    }
  }
  float* Cz = Cm + (size_t)sp * M * N;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int mm = m0 + ty * 2 + i, nn = n0 + tx * 2;
    if (mm < M && nn < N) *(float2*)(Cz + (size_t)mm * N + nn) = make_float2(scale * acc[i][0], scale * acc[i][1]);
  }
}

// out[i] = part[0][i] + part[1][i] + ... in that order
__global__ __launch_bounds__(256) void mlp_reduce_kernel(const ReduceBatch rb, int splits) {
  const float* __restrict__ part = rb.part[blockIdx.y];
  float* __restrict__ out = rb.out[blockIdx.y];
  const int count = rb.count[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float s = part[i];
  for (int z = 1; z < splits; ++z) s += part[(size_t)z * count + i];
  out[i] = s;
}

constexpr int MLP_WSPLIT = 8;   // row splits of the weight-gradient GEMM (its reduction runs over the n rows)

template <bool TA, bool TB>
static void gemm_dispatch_batch(const GemmBatch& g, hipStream_t st) {
  int maxM = 0, maxN = 0;
  long long wg64 = 0;
  for (int b = 0; b < g.nb; ++b) {
    maxM = g.M[b] > maxM ? g.M[b] : maxM;
    maxN = g.N[b] > maxN ? g.N[b] : maxN;
    wg64 += (long long)((g.N[b] + 63) / 64) * ((g.M[b] + 63) / 64) * g.splits;
  }
  // (both kernels add the same products in the same order: the choice moves time, not values)
  if (wg64 < 200)
    mlp_gemm32_kernel<TA, TB><<<dim3((maxN + 31) / 32, (maxM + 31) / 32, g.nb * g.splits), 256, 0, st>>>(g);
  else
    mlp_gemm_kernel<TA, TB><<<dim3((maxN + 63) / 64, (maxM + 63) / 64, g.nb * g.splits), 256, 0, st>>>(g);
}

template <bool TA, bool TB>
static void gemm_dispatch(const float* A, const float* B, float* Cm, int M, int N, int R, int splits, int rps, float scale,
                          hipStream_t st) {
  GemmBatch g;
  g.nb = 1; g.splits = splits; g.scale = scale;
  g.A[0] = A; g.B[0] = B; g.C[0] = Cm; g.M[0] = M; g.N[0] = N; g.R[0] = R; g.rps[0] = rps;
  gemm_dispatch_batch<TA, TB>(g, st);
}

static inline int rows_per_thread(int n) { return n <= 256 ? 1 : n <= 512 ? 2 : n <= 1024 ? 4 : 8; }

// One layer of nb heads (same rows n and width m, own input widths k[b]): z = x w^T, then BatchNorm1d(train) + activation.
hipError_t launch_mlp_heads_layer_forward(int nb, const float* const* x, int n, const int* k, const float* const* w, int m,
                                          const float* const* gamma, const float* const* beta, float eps, int act, float slope,
                                          float* const* z, float* const* y, float* const* mean, float* const* rstd,
                                          float* const* rmean, float* const* rvar, float momentum, hipStream_t st) {
  GemmBatch g;
  g.nb = nb; g.splits = 1; g.scale = 1.f;
  BnFwdBatch bt;
  for (int b = 0; b < nb; ++b) {
    g.A[b] = x[b]; g.B[b] = w[b]; g.C[b] = z[b]; g.M[b] = n; g.N[b] = m; g.R[b] = k[b]; g.rps[b] = k[b];
    bt.z[b] = z[b]; bt.gamma[b] = gamma[b]; bt.beta[b] = beta[b]; bt.y[b] = y[b]; bt.mean[b] = mean[b]; bt.rstd[b] = rstd[b];
    bt.rmean[b] = rmean[b]; bt.rvar[b] = rvar[b];
  }
  gemm_dispatch_batch<true, true>(g, st);                      // z [n][m] = x [n][k] w^T [k][m]
  const dim3 grid(m / MLP_CB, nb);
#define AMX_MLP_FWD(RPT) mlp_bn_fwd_kernel<RPT><<<grid, 256, 0, st>>>(bt, n, m, eps, act, slope, momentum)
  switch (rows_per_thread(n)) {
    case 1: AMX_MLP_FWD(1); break;
    case 2: AMX_MLP_FWD(2); break;
    case 4: AMX_MLP_FWD(4); break;
    default: AMX_MLP_FWD(8); break;
  }
#undef AMX_MLP_FWD
  return hipGetLastError();
}

hipError_t launch_mlp_layer_forward(const float* x, int n, int k, const float* w, int m, const float* gamma, const float* beta,
                                    float eps, int act, float slope, float* z, float* y, float* mean, float* rstd, float* rmean,
                                    float* rvar, float momentum, hipStream_t st) {
  return launch_mlp_heads_layer_forward(1, &x, n, &k, &w, m, &gamma, &beta, eps, act, slope, &z, &y, &mean, &rstd, &rmean, &rvar,
                                        momentum, st);
}

size_t mlp_backward_scratch_floats(int n, int cin, int width) {
  return 2 * (size_t)n * width + (size_t)MLP_WSPLIT * width * (cin > width ? cin : width);
}

// Adjoint of one layer of nb heads: dz (activation + BatchNorm adjoint), dW = dZ^T X (row-split + ordered sum), dX = dZ W (where asked).
hipError_t launch_mlp_heads_layer_backward(int nb, const float* const* dy, const float* const* y, const float* const* z,
                                           const float* const* mean, const float* const* rstd, const float* const* gamma, int act,
                                           float slope, const float* const* x, const float* const* w, int n, const int* k, int m,
                                           float* const* dz, float* const* dgamma, float* const* dbeta, float* const* dw,
                                           float* const* dx, float* const* wpart, hipStream_t st) {
  BnBwdBatch bt;
  for (int b = 0; b < nb; ++b) {
    bt.dy[b] = dy[b]; bt.y[b] = y[b]; bt.z[b] = z[b]; bt.mean[b] = mean[b]; bt.rstd[b] = rstd[b]; bt.gamma[b] = gamma[b];
    bt.dz[b] = dz[b]; bt.dgamma[b] = dgamma[b]; bt.dbeta[b] = dbeta[b];
  }
  const dim3 grid(m / MLP_CB, nb);
#define AMX_MLP_BWD(RPT) mlp_bwd_norm_kernel<RPT><<<grid, 256, 0, st>>>(bt, act, slope, n, m)
  switch (rows_per_thread(n)) {
    case 1: AMX_MLP_BWD(1); break;
    case 2: AMX_MLP_BWD(2); break;
    case 4: AMX_MLP_BWD(4); break;
    default: AMX_MLP_BWD(8); break;
  }
#undef AMX_MLP_BWD
  // dW [m][k] = dZ^T [m][n] X [n][k]: the reduction runs over the rows, split MLP_WSPLIT ways, partials summed in order
  const int splits = n >= 16 * MLP_WSPLIT ? MLP_WSPLIT : 1;
  const int rps = ((n + splits - 1) / splits + 15) / 16 * 16;
  GemmBatch g;
  g.nb = nb; g.splits = splits; g.scale = 1.f;
  ReduceBatch rb;
  int maxcount = 0;
  for (int b = 0; b < nb; ++b) {
    g.A[b] = dz[b]; g.B[b] = x[b]; g.C[b] = splits > 1 ? wpart[b] : dw[b]; g.M[b] = m; g.N[b] = k[b]; g.R[b] = n; g.rps[b] = rps;
    rb.part[b] = wpart[b]; rb.out[b] = dw[b]; rb.count[b] = m * k[b];
    maxcount = m * k[b] > maxcount ? m * k[b] : maxcount;
  }
  gemm_dispatch_batch<false, false>(g, st);
  if (splits > 1) mlp_reduce_kernel<<<dim3((maxcount + 255) / 256, nb), 256, 0, st>>>(rb, splits);
  // dX [n][k] = dZ [n][m] W [m][k]  (the heads that want it)
  GemmBatch gx;
  gx.nb = 0; gx.splits = 1; gx.scale = 1.f;
  for (int b = 0; b < nb; ++b)
    if (dx[b]) {
      const int j = gx.nb++;
      gx.A[j] = dz[b]; gx.B[j] = w[b]; gx.C[j] = dx[b]; gx.M[j] = n; gx.N[j] = k[b]; gx.R[j] = m; gx.rps[j] = m;
    }
  if (gx.nb) gemm_dispatch_batch<true, false>(gx, st);
  return hipGetLastError();
}

hipError_t launch_mlp_layer_backward(const float* dy, const float* y, const float* z, const float* mean, const float* rstd,
                                     const float* gamma, int act, float slope, const float* x, const float* w, int n, int k,
                                     int m, float* dz, float* dgamma, float* dbeta, float* dw, float* dx, float* wpart,
                                     hipStream_t st) {
  return launch_mlp_heads_layer_backward(1, &dy, &y, &z, &mean, &rstd, &gamma, act, slope, &x, &w, n, &k, m, &dz, &dgamma, &dbeta,
                                         &dw, &dx, &wpart, st);
}

// C (+ z * M * N for split z) = scale * op(A) op(B) over R split `splits` ways -- the same register-tiled kernel, for the two
// GEMMs of the contrastive loss (amx_supcon.hip), nb problems of one shape per launch.  Every contiguous storage dimension must be a
// multiple of 4.
hipError_t launch_small_gemm_batch(int nb, bool ta, bool tb, const float* const* A, const float* const* B, float* const* Cm, int M,
                                   int N, int R, int splits, float scale, hipStream_t st) {
  const int rps = ((R + splits - 1) / splits + 15) / 16 * 16;
  GemmBatch g;
  g.nb = nb; g.splits = splits; g.scale = scale;
  for (int b = 0; b < nb; ++b) {
    g.A[b] = A[b]; g.B[b] = B[b]; g.C[b] = Cm[b]; g.M[b] = M; g.N[b] = N; g.R[b] = R; g.rps[b] = rps;
  }
  if (ta && tb) gemm_dispatch_batch<true, true>(g, st);
  else if (ta) gemm_dispatch_batch<true, false>(g, st);
  else if (tb) gemm_dispatch_batch<false, true>(g, st);
  else gemm_dispatch_batch<false, false>(g, st);
  return hipGetLastError();
}

hipError_t launch_small_gemm(bool ta, bool tb, const float* A, const float* B, float* Cm, int M, int N, int R, int splits,
                             float scale, hipStream_t st) {
  return launch_small_gemm_batch(1, ta, tb, &A, &B, &Cm, M, N, R, splits, scale, st);
}

}  // namespace amx
