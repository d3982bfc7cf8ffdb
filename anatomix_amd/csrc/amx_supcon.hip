// anatomix_amd -- supervised patch contrastive loss, forward + backward in one call
// (reference pretraining/models/supcl_model.py:16-226, SupPatchNCELoss; called once per nce layer per step on
// [2 views x 512 patches, 256] projected features).  fp32 throughout; every reduction has a fixed order (no atomics).
//
//   xn      = x / max(|x|, 1e-8)                                   F.normalize(eps=1e-8)          (:67-70)
//   S       = xn xn^T / T ;  z_ij = S_ij - max_j S_ij                                             (:144-151)
//   plain   : logZ_i = log sum_{j != i} exp(z_ij)                                                 (:203-204)
//   balanced: logZ_i = log sum_{j != i} exp(z_ij) / n_ij,  n_ij = count(class j) - [class j == class i] (sqrt opt.)  (:172-196)
//   loss_i  = -( sum_{j != i, same class} z_ij / npos_i - logZ_i ),  npos_i = count(class i) - 1  (:209-212)
//   loss    = mean_i loss_i   or   sum_i r_i loss_i / sum_i r_i,  r_i = 1 / count(class i) (sqrt opt.)           (:213-224)
// Backward (the max shift is detached and cancels): dL/dS_ij = a_i (softmax_ij - [same, j != i] / npos_i), a_i the
// anchor weight above; dxn = (G + G^T) xn / T; dx = (dxn - xn (xn . dxn)) / |x|.
// The Gram matrix is 1024 x 1024 x 256 = 0.5 GFLOP -- negligible next to the UNet; the point of the kernel is to
// replace ~40 small launches (and a 1024^2 autograd graph) per layer with six.
#include "amx_device.h"

namespace amx {

constexpr int BM = 64, BK = 16;

// C[i][j] = scale * sum_k A[i][k] * B[j][k]   (A: M x K, B: N x K, row-major, K contiguous)
__global__ __launch_bounds__(256) void sc_gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                        float* __restrict__ Cm, int M, int N, int K, float scale) {
  __shared__ float As[BK][BM + 4], Bs[BK][BM + 4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BM;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += BK) {
    for (int t = threadIdx.x; t < BM * BK; t += 256) {
      const int m = t / BK, kk = t % BK;
      As[kk][m] = (i0 + m < M && k0 + kk < K) ? A[(long long)(i0 + m) * K + k0 + kk] : 0.f;
      Bs[kk][m] = (j0 + m < N && k0 + kk < K) ? B[(long long)(j0 + m) * K + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = As[kk][ty * 4 + r];
        b[r] = Bs[kk][tx * 4 + r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] += a[r] * b[c];
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + ty * 4 + r, j = j0 + tx * 4 + c;
      if (i < M && j < N) Cm[(long long)i * N + j] = acc[r][c] * scale;
    }
}

// D[z][i][c] = scale * sum over the z-th quarter of k of (G[i][k] + G[k][i]) * X[k][c]   (G: N x N, X: N x Cc).
// The output is only N x Cc (1024 x 256 = 64 tiles): the reduction is split four ways over blockIdx.z so that 256
// workgroups run; sc_dnorm_kernel adds the four partials in order.
constexpr int SC_KSPLIT = 4;
__global__ __launch_bounds__(256) void sc_gemm_sym_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                         float* __restrict__ Dm, int N, int Cc, float scale) {
  __shared__ float As[BK][BM + 4], Bs[BK][BM + 4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * BM, c0 = blockIdx.x * BM;
  const int kper = ((N + SC_KSPLIT - 1) / SC_KSPLIT + BK - 1) / BK * BK;
  const int kbeg = blockIdx.z * kper, kend = kbeg + kper < N ? kbeg + kper : N;
  Dm += (size_t)blockIdx.z * N * Cc;
  float acc[4][4] = {};
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    // two coalesced passes over the A tile: G[i][k] (k fastest) and G[k][i] (i fastest) -- a single pass with one of the
    // two index orders reads the other operand with stride N (147 us instead of ~30 for N = 1024)
    for (int t = threadIdx.x; t < BM * BK; t += 256) {
      const int m = t / BK, kk = t % BK;
      const int i = i0 + m, k = k0 + kk;
      As[kk][m] = (i < N && k < kend) ? G[(long long)i * N + k] : 0.f;
      const int cc = t % BM, kk2 = t / BM;                 // X tile: row k0 + kk2, columns c0 + cc (coalesced)
      Bs[kk2][cc] = (k0 + kk2 < kend && c0 + cc < Cc) ? X[(long long)(k0 + kk2) * Cc + c0 + cc] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < BM * BK; t += 256) {
      const int m = t % BM, kk = t / BM;
      const int i = i0 + m, k = k0 + kk;
      if (i < N && k < kend) As[kk][m] += G[(long long)k * N + i];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = As[kk][ty * 4 + r];
        b[r] = Bs[kk][tx * 4 + r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] += a[r] * b[c];
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + ty * 4 + r, j = c0 + tx * 4 + c;
      if (i < N && j < Cc) Dm[(long long)i * Cc + j] = acc[r][c] * scale;
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = red[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) s = fmaxf(s, red[k]);
  return s;
}

// ---- batches of losses of one shape (the six layers of a contrastive step: N = views * patches rows, C = the head width): every
// kernel below serves problem blockIdx.y; the scratch of problem b is the b-th slice of one buffer, laid out as launch_supcon lays it out.
struct ScBatch {
  const float* feat[MLP_MAXB];
  const int* lab[MLP_MAXB];
  float* loss[MLP_MAXB];
  float* grad[MLP_MAXB];
  char* scratch;
  size_t stride;                                            // bytes per problem
};
struct ScPtrs { float *xn, *dxn, *S, *inv, *cnt, *rowloss, *wsum; };
__host__ __device__ __forceinline__ ScPtrs sc_ptrs(char* base, int N, int C) {
  ScPtrs p;
  p.xn = (float*)base;
  p.dxn = p.xn + (size_t)N * C;
  p.S = p.dxn + (size_t)SC_KSPLIT * N * C;
  p.inv = p.S + (size_t)N * N;
  p.cnt = p.inv + N;
  p.rowloss = p.cnt + N;
  p.wsum = p.rowloss + N;
  return p;
}

// one block per row: xn = x / max(|x|, eps); inv[i] = 1 / max(|x|, eps)
__global__ __launch_bounds__(256) void sc_normalize_kernel(const ScBatch bt, int N, int C) {
  __shared__ float red[4];
  const ScPtrs q = sc_ptrs(bt.scratch + blockIdx.y * bt.stride, N, C);
  const float* __restrict__ x = bt.feat[blockIdx.y];
  float* __restrict__ xn = q.xn;
  float* __restrict__ inv = q.inv;
  const int i = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float v = x[(long long)i * C + c];
    s += v * v;
  }
  s = block_sum(s, red);
  const float nrm = fmaxf(sqrtf(s), 1e-8f);
  const float r = 1.f / nrm;
  for (int c = threadIdx.x; c < C; c += 256) xn[(long long)i * C + c] = x[(long long)i * C + c] * r;
  if (threadIdx.x == 0) inv[i] = r;
}

// one block per anchor: cnt[i] = #{j : label_j == label_i} (incl. i)
__global__ __launch_bounds__(256) void sc_count_kernel(const ScBatch bt, int N, int C) {
  __shared__ float red[4];
  const int* __restrict__ lab = bt.lab[blockIdx.y];
  float* __restrict__ cnt = sc_ptrs(bt.scratch + blockIdx.y * bt.stride, N, C).cnt;
  const int i = blockIdx.x, li = lab[i];
  float s = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) s += lab[j] == li ? 1.f : 0.f;
  s = block_sum(s, red);
  if (threadIdx.x == 0) cnt[i] = s;
}

// single block: wsum = sum_i r_i (rarity weights), fixed order
__global__ __launch_bounds__(256) void sc_wsum_kernel(const ScBatch bt, int N, int C, int rarity, int sqrt_mode) {
  __shared__ float red[4];
  const ScPtrs q = sc_ptrs(bt.scratch + blockIdx.y * bt.stride, N, C);
  const float* __restrict__ cnt = q.cnt;
  float* __restrict__ wsum = q.wsum;
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) s += rarity ? 1.f / (sqrt_mode ? sqrtf(cnt[i]) : cnt[i]) : 1.f;
  s = block_sum(s, red);
  if (threadIdx.x == 0) wsum[0] = s;
}

// one block per anchor i: row statistics, loss_i, and G_i. = dL/dS_i. written over S_i.
__global__ __launch_bounds__(256) void sc_rows_kernel(const ScBatch bt, int N, int C, int rarity, int balance, int sqrt_mode) {
  __shared__ float red[4];
  const ScPtrs q = sc_ptrs(bt.scratch + blockIdx.y * bt.stride, N, C);
  float* __restrict__ S = q.S;
  const int* __restrict__ lab = bt.lab[blockIdx.y];
  const float* __restrict__ cnt = q.cnt;
  const float* __restrict__ wsum = q.wsum;
  float* __restrict__ rowloss = q.rowloss;
  const int i = blockIdx.x, li = lab[i];
  float* row = S + (long long)i * N;
  float m = -3.0e38f;
  for (int j = threadIdx.x; j < N; j += 256) m = fmaxf(m, row[j]);
  m = block_max(m, red);
  float z = 0.f, ps = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    if (j == i) continue;
    const float zz = row[j] - m;
    const bool same = lab[j] == li;
    float w = 1.f;
    if (balance) {
      const float npc = cnt[j] - (same ? 1.f : 0.f);
      w = 1.f / (sqrt_mode ? sqrtf(npc) : npc);
    }
    z += w * expf(zz);
    if (same) ps += zz;
  }
  z = block_sum(z, red);
  ps = block_sum(ps, red);
  const float npos = cnt[i] - 1.f;
  const float logz = logf(z);
  const float li_loss = -(ps / npos - logz);
  const float ri = rarity ? 1.f / (sqrt_mode ? sqrtf(cnt[i]) : cnt[i]) : 1.f;
  const float ai = ri / wsum[0];
  if (threadIdx.x == 0) rowloss[i] = ai * li_loss;
  for (int j = threadIdx.x; j < N; j += 256) {
    float g = 0.f;
    if (j != i) {
      const bool same = lab[j] == li;
      float w = 1.f;
      if (balance) {
        const float npc = cnt[j] - (same ? 1.f : 0.f);
        w = 1.f / (sqrt_mode ? sqrtf(npc) : npc);
      }
      g = ai * (w * expf(row[j] - m) / z - (same ? 1.f / npos : 0.f));
    }
    row[j] = g;
  }
}

// single block: loss = sum_i rowloss[i] in index order (double)
__global__ __launch_bounds__(256) void sc_reduce_kernel(const ScBatch bt, int N, int C) {
  __shared__ double part[256];
  const float* __restrict__ rowloss = sc_ptrs(bt.scratch + blockIdx.y * bt.stride, N, C).rowloss;
  float* __restrict__ loss = bt.loss[blockIdx.y];
  double s = 0.0;
  const int per = (N + 255) / 256;
  for (int k = 0; k < per; ++k) {
    const int i = threadIdx.x * per + k;
    if (i < N) s += rowloss[i];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 256; ++k) t += part[k];
    loss[0] = (float)t;
  }
}

// one block per row: dx = (dxn - xn (xn . dxn)) * inv   (the clamp max(|x|, eps) is inactive for any non-degenerate row;
// for |x| < eps PyTorch's normalize has zero gradient through the clamp: dx = dxn / eps)
__global__ __launch_bounds__(256) void sc_dnorm_kernel(const ScBatch bt, int C, int N) {
  __shared__ float red[4];
  const ScPtrs q = sc_ptrs(bt.scratch + blockIdx.y * bt.stride, N, C);
  const float* __restrict__ xn = q.xn;
  const float* __restrict__ dxn = q.dxn;
  const float* __restrict__ inv = q.inv;
  float* __restrict__ dx = bt.grad[blockIdx.y];
  const int i = blockIdx.x;
  auto dval = [&](int c) {                                 // the four k-quarters of sc_gemm_sym_kernel, in order
    float d = 0.f;
#pragma unroll
    for (int z = 0; z < SC_KSPLIT; ++z) d += dxn[((size_t)z * N + i) * C + c];
    return d;
  };
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += xn[(long long)i * C + c] * dval(c);
  s = block_sum(s, red);
  const float r = inv[i];
  const bool clamped = r >= 0.99e8f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float d = dval(c);
    dx[(long long)i * C + c] = clamped ? d * r : (d - xn[(long long)i * C + c] * s) * r;
  }
}

size_t supcon_scratch_bytes(int N, int C) {
  return ((size_t)(1 + SC_KSPLIT) * N * C + (size_t)N * N + (size_t)4 * N + 64) * sizeof(float);
}

hipError_t launch_small_gemm_batch(int nb, bool ta, bool tb, const float* const* A, const float* const* B, float* const* Cm, int M,
                                   int N, int R, int splits, float scale, hipStream_t st);   // amx_mlp.hip

// nb losses of one shape: feat[b] [N][C], labels[b] [N] -> loss[b], grad[b] (nullptr for every b: no gradients); scratch: nb slices of
// supcon_scratch_bytes(N, C).
hipError_t launch_supcon_batch(int nb, const float* const* feat, const int* const* labels, int N, int C, float temperature, int rarity,
                               int balance, int sqrt_mode, float* const* loss, float* const* grad, void* scratch, hipStream_t st) {
  ScBatch bt;
  bt.scratch = (char*)scratch;
  bt.stride = supcon_scratch_bytes(N, C);
  const float* xn[MLP_MAXB];
  const float* Sc[MLP_MAXB];
  float* S[MLP_MAXB];
  float* dxa[MLP_MAXB];
  float* dxb[MLP_MAXB];
  for (int b = 0; b < nb; ++b) {
    bt.feat[b] = feat[b]; bt.lab[b] = labels[b]; bt.loss[b] = loss[b]; bt.grad[b] = grad ? grad[b] : nullptr;
    const ScPtrs q = sc_ptrs(bt.scratch + b * bt.stride, N, C);
    xn[b] = q.xn; S[b] = q.S; Sc[b] = q.S; dxa[b] = q.dxn; dxb[b] = q.dxn + (size_t)2 * N * C;
  }
  const bool want_grad = grad && grad[0];
  const dim3 blk(256);
  hipLaunchKernelGGL(sc_normalize_kernel, dim3(N, nb), blk, 0, st, bt, N, C);
  hipLaunchKernelGGL(sc_count_kernel, dim3(N, nb), blk, 0, st, bt, N, C);
  hipLaunchKernelGGL(sc_wsum_kernel, dim3(1, nb), blk, 0, st, bt, N, C, rarity, sqrt_mode);
  const bool fast = !(N & 3) && !(C & 3);          // the register-tiled GEMM of amx_mlp.hip moves aligned float4s
  if (fast) {
    hipError_t e = launch_small_gemm_batch(nb, true, true, xn, xn, S, N, N, C, 1, 1.f / temperature, st);   // S = xn xn^T / T
    if (e != hipSuccess) return e;
  } else {
    for (int b = 0; b < nb; ++b)
      hipLaunchKernelGGL(sc_gemm_nt_kernel, dim3((N + BM - 1) / BM, (N + BM - 1) / BM), blk, 0, st, xn[b], xn[b], S[b], N, N, C,
                         1.f / temperature);
  }
  hipLaunchKernelGGL(sc_rows_kernel, dim3(N, nb), blk, 0, st, bt, N, C, rarity, balance, sqrt_mode);
  hipLaunchKernelGGL(sc_reduce_kernel, dim3(1, nb), blk, 0, st, bt, N, C);
  if (want_grad) {
    if (fast) {
      // dxn = (G + G^T) xn / T as four partial products (two row halves of G xn, two of G^T xn) that sc_dnorm adds in order
      static_assert(SC_KSPLIT == 4, "sc_dnorm sums SC_KSPLIT partials");
      hipError_t e = launch_small_gemm_batch(nb, true, false, Sc, xn, dxa, N, C, N, 2, 1.f / temperature, st);
      if (e != hipSuccess) return e;
      e = launch_small_gemm_batch(nb, false, false, Sc, xn, dxb, N, C, N, 2, 1.f / temperature, st);
      if (e != hipSuccess) return e;
    } else {
      for (int b = 0; b < nb; ++b)
        hipLaunchKernelGGL(sc_gemm_sym_kernel, dim3((C + BM - 1) / BM, (N + BM - 1) / BM, SC_KSPLIT), blk, 0, st, Sc[b], xn[b], dxa[b], N,
                           C, 1.f / temperature);
    }
    hipLaunchKernelGGL(sc_dnorm_kernel, dim3(N, nb), blk, 0, st, bt, C, N);
  }
  return hipGetLastError();
}

hipError_t launch_supcon(const float* feat, const int* labels, int N, int C, float temperature, int rarity, int balance,
                         int sqrt_mode, float* loss, float* grad, void* scratch, hipStream_t st) {
  return launch_supcon_batch(1, &feat, &labels, N, C, temperature, rarity, balance, sqrt_mode, &loss, &grad, scratch, st);
}

}  // namespace amx
