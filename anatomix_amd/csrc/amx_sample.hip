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
// "The first occurrence of every value" through a hash table in LDS: every draw claims the slot of its VALUE (open addressing,
// compare-and-swap on the key) and leaves the minimum of the draw indices there; a draw is kept when that minimum is its own index.
// The result does not depend on the order in which the lanes get there.  (The first two forms compared every draw with all earlier
// ones -- 524 k compares for 1024 draws, 25-30 us on the one compute unit this block runs on, five launches on the main stream of a
// training step; 32-bit values did not change that.)  VT: 32-bit keys when the volume has fewer than 2^31 voxels.
template <typename VT>
__global__ __launch_bounds__(1024) void sample_coords_kernel(const long long* __restrict__ draws, int n, int num, int d1, int d2,
                                                             long long* __restrict__ coords, int tsize) {
  extern __shared__ long long sm[];
  VT* v = (VT*)sm;                        // [n (+ pad)]   the draws
  VT* tkey = v + ((n + 15) & ~15);        // [tsize]       hash table: value of the slot (all ones: empty)
  int* tidx = (int*)(tkey + tsize);       // [tsize]       ... and the smallest draw index that holds it
  VT* kept = tkey;                        // [n]           distinct values in draw order -- the table's memory, once it is done with
  __shared__ int wsum[16], total;
  for (int i = threadIdx.x; i < n; i += 1024) v[i] = (VT)draws[i];
  for (int i = threadIdx.x; i < tsize; i += 1024) {
    tkey[i] = (VT)-1;
    tidx[i] = 0x7fffffff;
  }
  __syncthreads();
  const int per = (n + 1023) / 1024;      // consecutive items per thread, so the prefix sum follows draw order
  const int i0 = threadIdx.x * per;
  auto slot_of = [&](VT key, bool claim) {                 // the slot that holds `key` (claiming an empty one on the way when asked)
    unsigned h = ((unsigned)key * 2654435761u) >> 7;
    for (;;) {
      h &= (unsigned)(tsize - 1);
      VT cur = tkey[h];
      if (cur == key) return (int)h;
      if (claim && cur == (VT)-1) {
        const VT old = atomicCAS(&tkey[h], (VT)-1, key);
        if (old == (VT)-1 || old == key) return (int)h;
      }
      ++h;
    }
  };
  for (int k = 0; k < per; ++k) {
    const int i = i0 + k;
    if (i < n) atomicMin(&tidx[slot_of(v[i], true)], i);
  }
  __syncthreads();
  int keep_mask = 0, cnt = 0;
  for (int k = 0; k < per; ++k) {
    const int i = i0 + k;
    if (i >= n) break;
    if (tidx[slot_of(v[i], false)] == i) {
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
  int p = wsum[wv] + inc - cnt;            // (the barriers of the scan separate the last table read from the first write of `kept`)
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

// ---- small grids: a uniformly random PERMUTATION prefix.  When the grid has fewer than 8 x num voxels (the 8^3 tap of a 128^3 step:
// 512 voxels for 512 patches) the draw-and-keep-distinct kernel above cannot be used and PatchSampleF fell back to torch.randperm --
// about a dozen small launches in the middle of the forward (~150 us of the main stream per training step).  Here: one random 62-bit key
// per voxel (a single torch.randint launch, so torch's generator still seeds it), the voxel indices sorted by (key, index) with a
// bitonic network in LDS by ONE block, the first `num` unravelled.  Every order is equally likely (ties between 62-bit keys: < 1e-12
// for 4096 voxels, and then broken by index -- deterministic either way).
__global__ __launch_bounds__(1024) void sample_perm_kernel(const long long* __restrict__ keys, int nvox, int num, int d1, int d2,
                                                           long long* __restrict__ coords) {
  __shared__ unsigned long long k[4096];                    // (key << 12) | index: one compare orders both
  int np2 = 1;
  while (np2 < nvox) np2 <<= 1;
  for (int i = threadIdx.x; i < np2; i += 1024)
    k[i] = i < nvox ? (((unsigned long long)keys[i] & ((1ull << 50) - 1)) << 12) | (unsigned)i : ~0ull;   // padding sorts last
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (np2 >> 1); t += 1024) {
        const int lo = ((t / stride) * stride << 1) + (t % stride), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = k[lo], b = k[hi];
        if ((a > b) == up) {
          k[lo] = b;
          k[hi] = a;
        }
      }
      __syncthreads();
    }
  for (int r = threadIdx.x; r < num; r += 1024) {
    const long long f = (long long)(k[r] & 4095);
    coords[3 * r] = f / ((long long)d1 * d2);
    coords[3 * r + 1] = (f / d2) % d1;
    coords[3 * r + 2] = f % d2;
  }
}

hipError_t launch_sample_perm(const long long* keys, int nvox, int num, int d1, int d2, long long* coords, hipStream_t st) {
  sample_perm_kernel<<<1, 1024, 0, st>>>(keys, nvox, num, d1, d2, coords);
  return hipGetLastError();
}

// ---- class labels of the sampled patches (SupPatchNCELoss, supcl_model.py:100-123): F.interpolate(seg, size = feature map, 'nearest')
// gathered at the patch coordinates, rounded to integer class ids and tiled over the views.  Six small torch launches per layer
// (resize of the whole map, index, round, cast, repeat, contiguous); here the `views x P` ids are read straight from the
// full-resolution segmentation: source index = min(floor(dst * float(in) / out), in - 1), ATen's nearest rule.
__global__ void gather_labels_kernel(const float* __restrict__ seg, int D, int H, int W, const long long* __restrict__ coords, int P,
                                     int d, int h, int w, int views, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float sz = (float)D / (float)d, sy = (float)H / (float)h, sx = (float)W / (float)w;
  int z = (int)floorf((float)coords[3 * i] * sz), y = (int)floorf((float)coords[3 * i + 1] * sy), x = (int)floorf((float)coords[3 * i + 2] * sx);
  z = z < D - 1 ? z : D - 1;
  y = y < H - 1 ? y : H - 1;
  x = x < W - 1 ? x : W - 1;
  const int lab = (int)rintf(seg[((long long)z * H + y) * W + x]);   // torch.round: to nearest, ties to even
  for (int v = 0; v < views; ++v) out[v * P + i] = lab;
}

hipError_t launch_gather_labels(const float* seg, int D, int H, int W, const long long* coords, int P, int d, int h, int w, int views,
                                int* out, hipStream_t st) {
  gather_labels_kernel<<<(P + 255) / 256, 256, 0, st>>>(seg, D, H, W, coords, P, d, h, w, views, out);
  return hipGetLastError();
}

// the same for nb feature maps of one segmentation in one launch (blockIdx.y = map): dims[3 b ..] = (d, h, w) of map b
struct GatherLabelsBatch {
  const long long* coords[MLP_MAXB];
  int* out[MLP_MAXB];
  int d[MLP_MAXB], h[MLP_MAXB], w[MLP_MAXB];
};
__global__ void gather_labels_batch_kernel(const float* __restrict__ seg, int D, int H, int W, const GatherLabelsBatch bt, int P, int views) {
  const int b = blockIdx.y;
  const long long* __restrict__ coords = bt.coords[b];
  int* __restrict__ out = bt.out[b];
  const int d = bt.d[b], h = bt.h[b], w = bt.w[b];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float sz = (float)D / (float)d, sy = (float)H / (float)h, sx = (float)W / (float)w;
  int z = (int)floorf((float)coords[3 * i] * sz), y = (int)floorf((float)coords[3 * i + 1] * sy), x = (int)floorf((float)coords[3 * i + 2] * sx);
  z = z < D - 1 ? z : D - 1;
  y = y < H - 1 ? y : H - 1;
  x = x < W - 1 ? x : W - 1;
  const int lab = (int)rintf(seg[((long long)z * H + y) * W + x]);
  for (int v = 0; v < views; ++v) out[v * P + i] = lab;
}

hipError_t launch_gather_labels_batch(const float* seg, int D, int H, int W, int nb, const long long* const* coords, int P, const int* dims,
                                      int views, int* const* out, hipStream_t st) {
  GatherLabelsBatch bt;
  for (int b = 0; b < nb; ++b) {
    bt.coords[b] = coords[b]; bt.out[b] = out[b]; bt.d[b] = dims[3 * b]; bt.h[b] = dims[3 * b + 1]; bt.w[b] = dims[3 * b + 2];
  }
  gather_labels_batch_kernel<<<dim3((P + 255) / 256, nb), 256, 0, st>>>(seg, D, H, W, bt, P, views);
  return hipGetLastError();
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
  int tsize = 1024;
  while (tsize < 2 * n) tsize <<= 1;                         // load factor <= 1/2 (n <= 4096: at most 8192 slots)
  const bool small = (long long)d0 * d1 * d2 < (1ll << 31);
  const size_t vb = small ? sizeof(int) : sizeof(long long);
  const size_t lds = (size_t)(((n + 15) & ~15) + tsize) * vb + (size_t)tsize * sizeof(int);   // draws + table (the kept list re-uses the table)
  static amx::DeviceOnce attr_once;                          // per DEVICE: the attribute belongs to the device's copy of the function
  const bool attr = attr_once.done();
  if (lds > 159 * 1024) return hipErrorInvalidValue;
  if (!attr) {                                               // (the kernel also holds 68 bytes of static LDS)
    hipError_t e = hipFuncSetAttribute((const void*)sample_coords_kernel<int>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)sample_coords_kernel<unsigned long long>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    if (e != hipSuccess) return e;
    attr_once.set();
  }
  if (small)
    sample_coords_kernel<int><<<1, 1024, lds, st>>>(draws, n, num, d1, d2, coords, tsize);
  else
    sample_coords_kernel<unsigned long long><<<1, 1024, lds, st>>>(draws, n, num, d1, d2, coords, tsize);
  return hipGetLastError();
}

}  // namespace amx
