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
__global__ __launch_bounds__(1024) void sample_coords_kernel(const long long* __restrict__ draws, int n, int num, int d1, int d2,
                                                             long long* __restrict__ coords) {
  extern __shared__ long long sm[];
  long long* v = sm;                      // [n]   the draws
  long long* kept = sm + n;               // [n]   distinct values in draw order
  __shared__ int wsum[16], total;
  for (int i = threadIdx.x; i < n; i += 1024) v[i] = draws[i];
  __syncthreads();
  const int per = (n + 1023) / 1024;      // consecutive items per thread, so the prefix sum follows draw order
  const int i0 = threadIdx.x * per;
  int keep_mask = 0, cnt = 0;
  for (int k = 0; k < per; ++k) {
    const int i = i0 + k;
    if (i >= n) break;
    const long long mine = v[i];
    bool dup = false;
    for (int j = 0; j < i; ++j) dup |= (v[j] == mine);
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
    const long long f = kept[r < have ? r : (r - have) % have];
    coords[3 * r] = f / ((long long)d1 * d2);
    coords[3 * r + 1] = (f / d2) % d1;
    coords[3 * r + 2] = f % d2;
  }
}

hipError_t launch_sample_coords(const long long* draws, int n, int num, int d0, int d1, int d2, long long* coords, hipStream_t st) {
  (void)d0;
  const size_t lds = (size_t)n * 2 * sizeof(long long);
  sample_coords_kernel<<<1, 1024, lds, st>>>(draws, n, num, d1, d2, coords);
  return hipGetLastError();
}

}  // namespace amx
