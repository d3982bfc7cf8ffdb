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

// ---- backward of a k3 reflect conv whose OUTPUT gradient is nonzero only at P sampled voxels per sample (the contrastive step taps
// the network's output conv at 512 voxels, supcl_model.py:801-843 / pretraining_networks.py:472-480; nothing else uses that output):
// the dense route imports the rows into a zero gradient volume and runs the full weight / data gradient over 2 x 128^3 voxels of zeros
// (~250 us of a 6.5 ms step).  Same rounding points as the dense route: the row gradients are rounded to the storage type (what
// scatter_rows writes), weights of the data gradient are the 16-bit packed values, sums in fp32, the data gradient rounded once.
//   dW[co][ci][t]   = sum over (n, p) in index order of  g16[n][p][co] * x[n][r(c_p + t - 1)][ci]
//   din[n][u][ci]   = sum over p (index order), taps t (index order) with r(c_p + t - 1) = u  of  sum_co g16[n][p][co] * w16[co][ci][t]
// (r = reflect; din must be zero-filled by the caller -- only the <= 27 P voxels per sample that receive something are written, every
// one of them by whoever finds it, all with the same value: deterministic).  Cout, Cin <= 16; x has xc >= Cin channels per voxel.
constexpr int SWG = 8;                                        // sample chunks of the sampled weight gradient
template <typename T>
__global__ __launch_bounds__(256) void sampled_conv_wgrad_kernel(const float* __restrict__ g, const long long* __restrict__ coords,
                                                                 const char* __restrict__ x, int xc, int N, int P, int D, int H, int W,
                                                                 int Cout, int Cin, float* __restrict__ part, int per) {
  // block (tap t, sample chunk k): partial[k][t][co][ci] over samples [k per, (k + 1) per) in index order
  constexpr int CH = 128;                                      // samples staged per round (one exposed gather latency per round)
  __shared__ float gs[CH][16], xs[CH][16];
  const int t = blockIdx.x, kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
  const int co = threadIdx.x >> 4, ci = threadIdx.x & 15;
  float acc = 0.f;
  const int total = N * P;
  const int sbeg = blockIdx.y * per, send = sbeg + per < total ? sbeg + per : total;
  for (int s0 = sbeg; s0 < send; s0 += CH) {
    __syncthreads();
#pragma unroll 8
    for (int e = threadIdx.x; e < CH * 16; e += 256) {
      const int sl = e >> 4, c = e & 15, sidx = s0 + sl;
      float gv = 0.f, xv = 0.f;
      if (sidx < send) {
        const int n = sidx / P, p = sidx - n * P;
        if (c < Cout) gv = (float)(T)g[(size_t)sidx * Cout + c];
        if (c < Cin) {
          const int z = reflect_clamp((int)coords[3 * p] + kz - 1, D), y = reflect_clamp((int)coords[3 * p + 1] + ky - 1, H),
                    xx = reflect_clamp((int)coords[3 * p + 2] + kx - 1, W);
          xv = (float)((const T*)x)[((((size_t)n * D + z) * H + y) * W + xx) * xc + c];
        }
      }
      gs[sl][c] = gv;
      xs[sl][c] = xv;
    }
    __syncthreads();
#pragma unroll 16
    for (int sl = 0; sl < CH; ++sl) acc += gs[sl][co] * xs[sl][ci];          // (rows past the chunk hold zeros)
  }
  part[((size_t)blockIdx.y * 27 + t) * 256 + threadIdx.x] = acc;
}

// dW[co][ci][t] = partial[0] + partial[1] + ... in chunk order
__global__ __launch_bounds__(256) void sampled_conv_wgrad_sum_kernel(const float* __restrict__ part, int Cout, int Cin, float* __restrict__ dw) {
  const int t = blockIdx.x, co = threadIdx.x >> 4, ci = threadIdx.x & 15;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < SWG; ++k) s += part[((size_t)k * 27 + t) * 256 + threadIdx.x];
  if (co < Cout && ci < Cin) dw[((size_t)co * Cin + ci) * 27 + t] = s;
}

// neighbour masks of the samples: a target voxel of sample p (one of its 27 neighbours) can only receive from samples within Chebyshev
// distance 2 of p.  mask[p][c] bit b = sample 32 c + b is that near; one thread per (p, c).  (Searching per TARGET and per lane instead --
// 27 P x 16 lanes, each scanning P samples -- cost 110 us at P = 512; one thread per sample scanning all others, 42 us on eight waves.)
__global__ __launch_bounds__(256) void sampled_neighbours_kernel(const long long* __restrict__ coords, int P, unsigned* __restrict__ mask) {
  const int nchunk = (P + 31) / 32;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P * nchunk) return;
  const int p = i / nchunk, c = i - p * nchunk;
  const int cz = (int)coords[3 * p], cy = (int)coords[3 * p + 1], cx = (int)coords[3 * p + 2];
  unsigned m = 0;
#pragma unroll 8
  for (int b = 0; b < 32; ++b) {
    const int q = 32 * c + b;
    if (q >= P) break;
    const int dz = (int)coords[3 * q] - cz, dy = (int)coords[3 * q + 1] - cy, dx = (int)coords[3 * q + 2] - cx;
    if (!(dz < -2 || dz > 2 || dy < -2 || dy > 2 || dx < -2 || dx > 2)) m |= 1u << b;
  }
  mask[i] = m;
}

template <typename T>
__global__ __launch_bounds__(256) void sampled_conv_dgrad_kernel(const float* __restrict__ g, const long long* __restrict__ coords,
                                                                 const float* __restrict__ w, const unsigned* __restrict__ mask, int N, int P,
                                                                 int D, int H, int W, int Cout, int Cin, char* __restrict__ din, int dc) {
  // 16 lanes per target voxel (one input channel each), 16 targets per block.  (One thread per target with all 16 channels ran 180 us:
  // the lanes of a wave match 27 different taps, so the product block was walked 27 times per hit with two or three lanes active.)
  const int n = blockIdx.y;
  const int ci = threadIdx.x & 15;
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (e >= P * 27) return;
  const int t0 = e % 27, p0 = e / 27;
  const int uz = reflect_clamp((int)coords[3 * p0] + t0 / 9 - 1, D), uy = reflect_clamp((int)coords[3 * p0 + 1] + (t0 / 3) % 3 - 1, H),
            ux = reflect_clamp((int)coords[3 * p0 + 2] + t0 % 3 - 1, W);
  const int cic = ci < Cin ? ci : 0;
  const int nchunk = (P + 31) / 32;
  float acc = 0.f;
  for (int c = 0; c < nchunk; ++c) {                           // samples in index order: one summation order
    unsigned m = mask[(size_t)p0 * nchunk + c];
    while (m) {
      const int b = __builtin_ctz(m);
      m &= m - 1;
      const int p = 32 * c + b;
      const int cz = (int)coords[3 * p], cy = (int)coords[3 * p + 1], cx = (int)coords[3 * p + 2];
      const int dz = cz - uz, dy = cy - uy, dx = cx - ux;
      if (dz < -1 || dz > 1 || dy < -1 || dy > 1 || dx < -1 || dx > 1) continue;
      const float* gr = g + ((size_t)n * P + p) * Cout;
      for (int t = 0; t < 27; ++t) {
        if (reflect_clamp(cz + t / 9 - 1, D) != uz || reflect_clamp(cy + (t / 3) % 3 - 1, H) != uy || reflect_clamp(cx + t % 3 - 1, W) != ux)
          continue;
        const float* wr = w + (size_t)cic * 27 + t;
#pragma unroll 16
        for (int co = 0; co < Cout; ++co) acc += (float)(T)gr[co] * (float)(T)wr[(size_t)co * Cin * 27];
      }
    }
  }
  if (ci < Cin) ((T*)din)[((((size_t)n * D + uz) * H + uy) * W + ux) * dc + ci] = (T)acc;
}

size_t sampled_conv_backward_scratch_bytes(int P) { return (size_t)P * ((P + 31) / 32) * sizeof(unsigned) + (size_t)SWG * 27 * 256 * sizeof(float); }

hipError_t launch_sampled_conv_backward(const float* g, const long long* coords, const void* x, int xc, const float* w, int N, int P, int D,
                                        int H, int W, int Cout, int Cin, float* dw, void* din, int dc, void* scratch, int precision,
                                        hipStream_t st) {
  const dim3 dgrid((P * 27 + 15) / 16, N);
  const int nchunk = (P + 31) / 32;
  unsigned* mask = (unsigned*)scratch;
  float* part = (float*)(mask + (size_t)P * nchunk);
  const int total = N * P, per = (total + SWG - 1) / SWG;
  if (din) sampled_neighbours_kernel<<<(P * nchunk + 255) / 256, 256, 0, st>>>(coords, P, mask);
  if (precision == 0) {
    sampled_conv_wgrad_kernel<f16><<<dim3(27, SWG), 256, 0, st>>>(g, coords, (const char*)x, xc, N, P, D, H, W, Cout, Cin, part, per);
    if (din) sampled_conv_dgrad_kernel<f16><<<dgrid, 256, 0, st>>>(g, coords, w, mask, N, P, D, H, W, Cout, Cin, (char*)din, dc);
  } else {
    sampled_conv_wgrad_kernel<bf16><<<dim3(27, SWG), 256, 0, st>>>(g, coords, (const char*)x, xc, N, P, D, H, W, Cout, Cin, part, per);
    if (din) sampled_conv_dgrad_kernel<bf16><<<dgrid, 256, 0, st>>>(g, coords, w, mask, N, P, D, H, W, Cout, Cin, (char*)din, dc);
  }
  sampled_conv_wgrad_sum_kernel<<<27, 256, 0, st>>>(part, Cout, Cin, dw);
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
