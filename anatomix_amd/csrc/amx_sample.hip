// anatomix_amd -- patch coordinate sampling for PatchSampleF's no-mask branch
// (reference pretraining/models/pretraining_networks.py:443-470: every voxel is a candidate, `randperm(n)[:num_patches]`
// picks a uniform subset without replacement and the flat ids are unravelled to (x, y, z) coordinates).
// randperm sorts ALL n voxels (2 M keys for a 128^3 tap: ~25 launches, ~350 us) to keep 512 of them.  Drawing with
// replacement and keeping the first `num` DISTINCT values in draw order is the same distribution (sequential sampling
// without replacement) and needs only the draws (one torch.randint launch, so torch's generator still seeds it) and this
// single-block kernel, which also does the unravelling.
#include "amx_device.h"

namespace amx {

// draws [n] (values in [0, d0*d1*d2)) -> coords [num][3] int64, C-order unravel.  If fewer than `num` distinct values were
// drawn (practically unreachable for n >= 2 num and >= 8 num voxels) the tail repeats the kept ones cyclically.
// VT: the draws as 32-bit values in LDS when the volume has fewer than 2^31 voxels (every anatomix shape): the all-pairs comparison
// is 524 k compares for 1024 draws and was paced by the VALU at three instructions per 64-bit compare (30 us per launch, five
// launches on the main stream of a training step); one v_cmp per 32-bit value and 16-byte LDS reads bring it under 10 us.
template <typename VT>
__global__ __launch_bounds__(1024) void sample_coords_kernel(const long long* __restrict__ draws, int n, int num, int d1, int d2,
                                                             long long* __restrict__ coords) {
  extern __shared__ long long sm[];
  VT* v = (VT*)sm;                        // [n (+ pad)]   the draws
  VT* kept = v + ((n + 15) & ~15);        // [n]           distinct values in draw order
  __shared__ int wsum[16], total;
  for (int i = threadIdx.x; i < n; i += 1024) v[i] = (VT)draws[i];
  __syncthreads();
  const int per = (n + 1023) / 1024;      // consecutive items per thread, so the prefix sum follows draw order
  const int i0 = threadIdx.x * per;
  int keep_mask = 0, cnt = 0;
  for (int k = 0; k < per; ++k) {
    const int i = i0 + k;
    if (i >= n) break;
    const VT mine = v[i];
    // (one value per iteration was a chain of dependent LDS round trips: 25 us for 1024 draws; independent broadcast reads of 16 / 8
    //  values per iteration are paced by the LDS pipe and the compares instead)
    bool dup = false;
    int j = 0;
    constexpr int UN = sizeof(VT) == 4 ? 16 : 8;
    for (; j + UN <= i; j += UN) {
      VT t[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) t[u] = v[j + u];
#pragma unroll
      for (int u = 0; u < UN; ++u) dup |= (t[u] == mine);
    }
    for (; j < i; ++j) dup |= (v[j] == mine);
    if (!dup) {
      keep_mask |= 1 << k;
      ++cnt;
    }
  }
  // block exclusive scan of cnt: wave scan + wave totals
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < 16; ++w) {
      const int t = wsum[w];
      wsum[w] = run;
      run += t;
    }
    total = run;
  }
  __syncthreads();
  int p = wsum[wv] + inc - cnt;
  for (int k = 0; k < per; ++k) {
    const int i = i0 + k;
    if (i >= n) break;
    if (keep_mask & (1 << k)) kept[p++] = v[i];
  }
  __syncthreads();
  const int have = total;
  for (int r = threadIdx.x; r < num; r += 1024) {
    const long long f = (long long)kept[r < have ? r : (r - have) % have];
    coords[3 * r] = f / ((long long)d1 * d2);
    coords[3 * r + 1] = (f / d2) % d1;
    coords[3 * r + 2] = f % d2;
  }
}

// ---- sampled feature taps (the contrastive step reads 512 voxels of each tapped feature map: supcl_model.py:801-843 calls
// netF(feat_k, num_patches, ids), pretraining_networks.py:472-480 gathers feat[:, :, x, y, z]).  Going through a dense fp32 NCDHW
// copy of every tapped tensor costs an export pass forward and, backward, a dense zero tensor + index_put + an import pass per
// layer -- 268 MB each way at the 128^3 tap -- for 1024 rows.  These two kernels read / write the rows in place instead.
//
// gather : rows[n][p][c] (fp32) = src[n][coords[p]][c];  src is a 16-bit channels-last tensor (dtype 0 f16 / 1 bf16) or fp32
//          (dtype 2: the network output, NCDHW) through element strides (sn, sz, sy, sx, sc)
// scatter: dst[n][coords[p]][c] (16-bit, byte strides) = or += rows[n][p][c], fp32 add then ONE rounding -- the arithmetic of
//          import_ncdhw(accumulate) restricted to the sampled voxels.  The coordinates of a layer are distinct (sampling without
//          replacement), so no two threads touch one element: deterministic.
template <int DT>
__global__ void gather_rows_kernel(const char* __restrict__ src, long long sn, long long sz, long long sy, long long sx, long long sc,
                                   const long long* __restrict__ coords, int N, int P, int C, float* __restrict__ rows) {
  const long long total = (long long)N * P * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const int pi = (int)(r % P), n = (int)(r / P);
    const long long e = n * sn + coords[3 * pi] * sz + coords[3 * pi + 1] * sy + coords[3 * pi + 2] * sx + c * sc;
    float v;
    if (DT == 2) v = ((const float*)src)[e];
    else if (DT == 0) v = (float)((const f16*)src)[e];
    else v = (float)((const bf16*)src)[e];
    rows[i] = v;
  }
}

template <typename T>
__global__ void scatter_rows_kernel(const float* __restrict__ rows, const long long* __restrict__ coords, char* __restrict__ dst,
                                    long long dn, long long dz, long long dy, long long dx, int N, int P, int C, int accumulate) {
  const long long total = (long long)N * P * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const int pi = (int)(r % P), n = (int)(r / P);
    T* d = (T*)(dst + n * dn + coords[3 * pi] * dz + coords[3 * pi + 1] * dy + coords[3 * pi + 2] * dx) + c;
    const float v = rows[i];
    *d = accumulate ? (T)((float)*d + v) : (T)v;
  }
}

hipError_t launch_gather_rows(const void* src, int dtype, long long sn, long long sz, long long sy, long long sx, long long sc,
                              const long long* coords, int N, int P, int C, float* rows, hipStream_t st) {
  const long long total = (long long)N * P * C;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dtype == 0) gather_rows_kernel<0><<<blocks, 256, 0, st>>>((const char*)src, sn, sz, sy, sx, sc, coords, N, P, C, rows);
  else if (dtype == 1) gather_rows_kernel<1><<<blocks, 256, 0, st>>>((const char*)src, sn, sz, sy, sx, sc, coords, N, P, C, rows);
  else if (dtype == 2) gather_rows_kernel<2><<<blocks, 256, 0, st>>>((const char*)src, sn, sz, sy, sx, sc, coords, N, P, C, rows);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_scatter_rows(const float* rows, const long long* coords, void* dst, int dtype, long long dn, long long dz, long long dy,
                               long long dx, int N, int P, int C, int accumulate, hipStream_t st) {
  const long long total = (long long)N * P * C;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dtype == 0) scatter_rows_kernel<f16><<<blocks, 256, 0, st>>>(rows, coords, (char*)dst, dn, dz, dy, dx, N, P, C, accumulate);
  else if (dtype == 1) scatter_rows_kernel<bf16><<<blocks, 256, 0, st>>>(rows, coords, (char*)dst, dn, dz, dy, dx, N, P, C, accumulate);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_sample_coords(const long long* draws, int n, int num, int d0, int d1, int d2, long long* coords, hipStream_t st) {
  const size_t lds = (size_t)(((n + 15) & ~15) + n) * sizeof(long long);
  if ((long long)d0 * d1 * d2 < (1ll << 31))
    sample_coords_kernel<int><<<1, 1024, lds, st>>>(draws, n, num, d1, d2, coords);
  else
    sample_coords_kernel<long long><<<1, 1024, lds, st>>>(draws, n, num, d1, d2, coords);
  return hipGetLastError();
}

}  // namespace amx
