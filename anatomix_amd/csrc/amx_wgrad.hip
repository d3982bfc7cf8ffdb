// anatomix_amd -- weight gradient of nn.Conv3d(k3, stride 1, padding='same', padding_mode='reflect')
// (the wgrad half of the backward of anatomix/model/network.py:310-461, needed by the contrastive step,
// pretraining/models/supcl_model.py:603-661).
//
//   dW[co][ci][kz][ky][kx] = sum over (n, z, y, x) of  dY[n][z][y][x][co] * In[n][r(z+kz-1)][r(y+ky-1)][r(x+kx-1)][ci]
//   (r = reflect; In may be the concatenation of a full-resolution segment and a nearest-upsampled half-resolution
//    segment, exactly as the forward kernels read it).
//
// A GEMM whose reduction index is the VOXEL: D[co][ci] += A[co][k] B[k][ci] per tap with k = 32 consecutive x of
// one row, v_mfma_f32_16x16x32.  A lane of an MFMA operand holds 8 consecutive k of ONE channel, but the tensors are
// channels-last; the tiles are therefore TRANSPOSED when they are staged into LDS ([row][channel][x], two voxels per
// 32-bit write), after which
//   * an A fragment (dY) is one ds_read_b128;
//   * the three x taps of one (kz, ky) come from TWO ds_read_b128 of the halo row (x-1+0 .. +15 in halo coordinates):
//     kx = 0 is the first block itself, kx = 2 is a register renaming, kx = 1 four v_alignbit;
//   * a row stride of 136 halves (272 B = 256 + 16) puts the 16 channel rows of a fragment read on 16 different
//     16-byte bank groups: conflict-free.
// (First version: channels-last tiles and 16-bit gathers, 98 LDS reads + ~140 VALU per 27 MFMAs -- LDS/VALU bound.)
// Reflect padding and the nearest upsample of the concat convs are resolved when the tile is staged.
// Work decomposition: grid (co-tile x ci-tile pairs, spatial chunks).  A workgroup walks its chunk of (n, 4-row tile, z)
// items (z fastest): the 3 input z-planes of an item live in a ring of 4 LDS slots (slot = plane & 3), so a step stages
// ONE new plane (+ the next dY rows), and those global loads are issued into registers BEFORE the item's MFMA sweep and
// written to LDS after it (software prefetch; one barrier per item).  Its 8 waves split the item's K-blocks, keep 27
// accumulators each for the whole chunk, and are summed through LDS at the end.  Partials [chunk][pair][27][16][16]
// fp32 are then added by wgrad_reduce_kernel with a fixed summation tree: deterministic, no atomics.
#include <stdlib.h>

#include "amx_device.h"

namespace amx {

// Tile geometry by row width: TY rows of the padded width Wp per item, LDS row stride RS halves (>= Wp + 8; 2 * RS is an odd
// multiple of 16 bytes mod 256, so the 16 channel rows of a fragment read hit 16 different 16-byte bank groups).  The item
// always holds TY * Wp / 32 = 16 K-blocks, two per wave: narrow (deep) layers used to run with 4 rows like the wide ones and
// kept only 4 (W <= 32) of the 8 waves busy.
constexpr int wg_ty(int Wp) { return Wp <= 32 ? 16 : Wp <= 64 ? 8 : 4; }
constexpr int wg_rs(int Wp) { return Wp <= 32 ? 40 : Wp <= 64 ? 72 : 136; }

struct WgUnit {                       // staging unit: two adjacent voxels x 8 channels
  uint4 a, b;
};

template <typename T, bool RING, int WG_TY, int WG_RS>
__global__ __launch_bounds__(512) void conv3d_wgrad_kernel(const WgradParams p, int Wp) {
  typedef typename Ops<T>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NDB = RING ? 2 : 1;
  constexpr int RS = WG_RS;
  constexpr int DY_BYTES = WG_TY * 16 * RS * 2, PLANE_BYTES = (WG_TY + 2) * 16 * RS * 2;
  constexpr int NPL = 2, NDY = 1;                                   // prefetch units per thread (Wp <= 128)
  char* dys = smem;                                                 // [NDB][WG_TY][16 co][RS]
  char* ins = smem + NDB * DY_BYTES;                                // [RING ? 4 : 3][WG_TY + 2][16 ci][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;

  const int ncit = (p.C0 + p.C1) / 16;
  const int pair = blockIdx.x, cot = pair / ncit, cit = pair % ncit;
  const int chunk = blockIdx.y;
  const bool seg1 = cit * 16 >= p.C0;
  const int ci0 = seg1 ? cit * 16 - p.C0 : cit * 16;
  const int sh = seg1 ? p.up_shift : 0;

  // two 16-byte global loads per unit, eight 32-bit LDS writes (transposed: channel rows, x along the row).
  // A thread's units (u = tid + 512 k) keep their tile coordinates for the whole chunk: the decomposition of u (divisions by the
  // run-time row width) and everything else that does not depend on the item is done ONCE here -- inside the item loop those
  // divisions were ~500 VALU instructions per thread and item, more than the item's MFMA time.
  const int hwp = Wp / 2, hwh = (Wp + 2) / 2;
  const int ndy = WG_TY * hwp * 2, npl = (WG_TY + 2) * hwh * 2;
  struct UnitDy { int row, lds; long long ga, gb; bool ok, oka, okb; } udy[NDY];
  struct UnitPl { int hr, lds; long long ga, gb; bool ok, oka, okb; } upl[NPL];
#pragma unroll
  for (int k = 0; k < NDY; ++k) {
    const int u = tid + k * 512, half = u & 1, xp = (u >> 1) % hwp, row = (u >> 1) / hwp, x = 2 * xp;
    udy[k].ok = u < ndy; udy[k].row = row;
    udy[k].lds = ((row * 16 + half * 8) * RS + x) * 2;
    udy[k].ga = (long long)x * p.yx + cot * 32 + half * 16; udy[k].gb = udy[k].ga + p.yx;
    udy[k].oka = x < p.W; udy[k].okb = x + 1 < p.W;
  }
  const long long sxs = seg1 ? p.s1x : p.s0x;
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int u = tid + k * 512, half = u & 1, xp = (u >> 1) % hwh, hr = (u >> 1) / hwh, xh = 2 * xp;
    upl[k].ok = u < npl; upl[k].hr = hr;
    upl[k].lds = ((hr * 16 + half * 8) * RS + xh) * 2;
    upl[k].ga = (long long)(reflect_clamp(xh - 1, p.W) >> sh) * sxs + ci0 * 2 + half * 16;
    upl[k].gb = (long long)(reflect_clamp(xh, p.W) >> sh) * sxs + ci0 * 2 + half * 16;
    upl[k].oka = xh <= p.W + 1; upl[k].okb = xh + 1 <= p.W + 1;
  }
  auto load_dy = [&](int k, int n, int z, int y0) -> WgUnit {
    WgUnit r{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (udy[k].ok && y0 + udy[k].row < p.H) {
      const char* b = p.dy + (long long)n * p.yn + (long long)z * p.yz + (long long)(y0 + udy[k].row) * p.yy;
      if (udy[k].oka) r.a = *(const uint4*)(b + udy[k].ga);
      if (udy[k].okb) r.b = *(const uint4*)(b + udy[k].gb);
    }
    return r;
  };
  auto load_pl = [&](int k, int n, int zz, int y0) -> WgUnit {      // zz: full-resolution plane index (already reflected)
    WgUnit r{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (upl[k].ok) {
      const int z2 = zz >> sh, yy = reflect_clamp(y0 + upl[k].hr - 1, p.H) >> sh;
      const char* b = seg1 ? p.src1 + (long long)n * p.s1n + (long long)z2 * p.s1z + (long long)yy * p.s1y
                           : p.src0 + (long long)n * p.s0n + (long long)z2 * p.s0z + (long long)yy * p.s0y;
      if (upl[k].oka) r.a = *(const uint4*)(b + upl[k].ga);
      if (upl[k].okb) r.b = *(const uint4*)(b + upl[k].gb);
    }
    return r;
  };
  auto store_unit = [&](char* at, const WgUnit& v) {                // at: LDS address of (channel row e = 0, even x)
    const unsigned a[4] = {v.a.x, v.a.y, v.a.z, v.a.w}, b[4] = {v.b.x, v.b.y, v.b.z, v.b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned lo = (a[e >> 1] >> ((e & 1) * 16)) & 0xffffu, hi = (b[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
      *(unsigned*)(at + e * RS * 2) = lo | (hi << 16);
    }
  };
  auto store_dy = [&](int k, int buf, const WgUnit& v) {
    if (udy[k].ok) store_unit(dys + (size_t)buf * DY_BYTES + udy[k].lds, v);
  };
  auto store_pl = [&](int k, int slot, const WgUnit& v) {
    if (upl[k].ok) store_unit(ins + (size_t)slot * PLANE_BYTES + upl[k].lds, v);
  };

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nxb = Wp / 32;
  const int item0 = chunk * p.items_per_chunk;
  const int item1 = item0 + p.items_per_chunk < p.nitems ? item0 + p.items_per_chunk : p.nitems;
  int db = 0;                                                       // dY buffer of the current item
  WgUnit ra_dy[NDY], ra_pl[NPL];                                    // requested one item ago, stored after this item's sweep
  bool ra_valid = false;
  for (int item = item0; item < item1; ++item) {
    const int z = item % p.D;                                       // z fastest: consecutive items march along z
    const int r = item / p.D;
    const int yt = r % p.nyt, n = r / p.nyt;
    const int y0 = yt * WG_TY;
    int slot[3];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) slot[kz] = RING ? (reflect_clamp(z + kz - 1, p.D) & 3) : kz;
    const bool cold = !RING || item == item0 || z == 0;
    if (cold) {
      __syncthreads();                                              // previous item's fragments are consumed
#pragma unroll
      for (int k = 0; k < NDY; ++k) store_dy(k, db, load_dy(k, n, z, y0));
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        if (RING && kz == 2 && slot[2] == slot[0]) continue;        // z = 0 or D-1: the reflected plane is already staged
        const int zz = reflect_clamp(z + kz - 1, p.D);
#pragma unroll
        for (int k = 0; k < NPL; ++k) store_pl(k, slot[kz], load_pl(k, n, zz, y0));
      }
      __syncthreads();
    }
    // ---- register prefetch, TWO items deep (RING; the next items are z + 1, z + 2 of the same tile): what item z + 1 needs
    //      (its dY rows and plane z + 2) was requested during item z - 1 and is written to LDS after this item's sweep; what
    //      item z + 2 needs is requested now.  One item deep, the ~2 us HBM latency of the request was longer than a sweep and
    //      every item ended waiting for its loads (8800 cycles per item against 1700 of MFMAs).
    const bool has_next = RING && item + 1 < item1 && z + 1 < p.D;
    const bool has_next2 = has_next && item + 2 < item1 && z + 2 < p.D;
    if (cold) ra_valid = false;
    if (has_next && !ra_valid) {                                    // first item of a run: request item z + 1's data now (exposed once)
#pragma unroll
      for (int k = 0; k < NDY; ++k) ra_dy[k] = load_dy(k, n, z + 1, y0);
      if (z + 2 < p.D)
#pragma unroll
        for (int k = 0; k < NPL; ++k) ra_pl[k] = load_pl(k, n, z + 2, y0);
    }
    WgUnit rb_dy[NDY], rb_pl[NPL];
    if (has_next2) {
#pragma unroll
      for (int k = 0; k < NDY; ++k) rb_dy[k] = load_dy(k, n, z + 2, y0);
      if (z + 3 < p.D)
#pragma unroll
        for (int k = 0; k < NPL; ++k) rb_pl[k] = load_pl(k, n, z + 3, y0);
    }
    // ---- K-blocks of this item: (row, xb) -> 32 voxels.  A wave owns the two K-blocks (rows r0, r0 + 1) of one x block: the four
    //      halo rows r0 .. r0 + 3 they touch are read ONCE per kz and feed both rows (row r uses halo rows r + ky), i.e. 8 + 2 / 3
    //      fragment reads per 18 MFMAs instead of 12 + 2 / 3 -- a third of the LDS traffic, which (with its bank conflicts) paced
    //      this kernel.
    const char* dyb = dys + (size_t)db * DY_BYTES;
    const int vrows = p.H - y0 < WG_TY ? p.H - y0 : WG_TY;         // rows of this tile inside the volume (the rest hold zeros)
    {
      const int xb = wave % nxb, r0 = (wave / nxb) * 2;
      if (r0 < vrows) {
        const int X0 = xb * 32 + kg * 8;                           // multiple of 8 halves = 16 bytes
        const vec8 af0 = *(const vec8*)(dyb + ((size_t)(r0 * 16 + m) * RS + X0) * 2);
        const vec8 af1 = *(const vec8*)(dyb + ((size_t)((r0 + 1) * 16 + m) * RS + X0) * 2);   // zeros when r0 + 1 is outside
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
          for (int hr = 0; hr < 4; ++hr) {
            const char* bp = ins + (size_t)slot[kz] * PLANE_BYTES + ((size_t)((r0 + hr) * 16 + m) * RS + X0) * 2;
            const uint4 B0 = *(const uint4*)bp, B1 = *(const uint4*)(bp + 16);   // halo x = X0 .. X0 + 15  (= x - 1 ..)
            const unsigned b0[4] = {B0.x, B0.y, B0.z, B0.w};                       // kx = 0
            const unsigned b1[4] = {__builtin_amdgcn_alignbit(B0.y, B0.x, 16), __builtin_amdgcn_alignbit(B0.z, B0.y, 16),
                                    __builtin_amdgcn_alignbit(B0.w, B0.z, 16), __builtin_amdgcn_alignbit(B1.x, B0.w, 16)};   // kx = 1
            const unsigned b2[4] = {B0.y, B0.z, B0.w, B1.x};                       // kx = 2
            if (hr < 3) {                                          // row r0, ky = hr
              const int t0 = (kz * 3 + hr) * 3;
              acc[t0] = Ops<T>::mfma(af0, __builtin_bit_cast(vec8, b0), acc[t0]);
              acc[t0 + 1] = Ops<T>::mfma(af0, __builtin_bit_cast(vec8, b1), acc[t0 + 1]);
              acc[t0 + 2] = Ops<T>::mfma(af0, __builtin_bit_cast(vec8, b2), acc[t0 + 2]);
            }
            if (hr > 0) {                                          // row r0 + 1, ky = hr - 1
              const int t0 = (kz * 3 + hr - 1) * 3;
              acc[t0] = Ops<T>::mfma(af1, __builtin_bit_cast(vec8, b0), acc[t0]);
              acc[t0 + 1] = Ops<T>::mfma(af1, __builtin_bit_cast(vec8, b1), acc[t0 + 1]);
              acc[t0 + 2] = Ops<T>::mfma(af1, __builtin_bit_cast(vec8, b2), acc[t0 + 2]);
            }
          }
      }
    }
    if (has_next) {
      // slot (z + 2) & 3 and the other dY buffer are not read by this item, and the item that read them last ended with
      // the barrier below
#pragma unroll
      for (int k = 0; k < NDY; ++k) store_dy(k, db ^ 1, ra_dy[k]);
      if (z + 2 < p.D)
#pragma unroll
        for (int k = 0; k < NPL; ++k) store_pl(k, (z + 2) & 3, ra_pl[k]);
      db ^= 1;
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NDY; ++k) ra_dy[k] = rb_dy[k];
#pragma unroll
    for (int k = 0; k < NPL; ++k) ra_pl[k] = rb_pl[k];
    ra_valid = has_next2;
  }
  // ---- sum the 8 waves through LDS (3 rounds), wave 0 writes the partial
  float* red = (float*)smem;                                        // [4 waves][27][64][4] floats = 108 KiB
  for (int half = 4; half >= 1; half >>= 1) {
    __syncthreads();
    if (wave >= half && wave < 2 * half)
#pragma unroll
      for (int t = 0; t < 27; ++t) *(f32x4*)(red + (((size_t)(wave - half) * 27 + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    if (wave < half)
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const f32x4 o = *(const f32x4*)(red + (((size_t)wave * 27 + t) * 64 + lane) * 4);
        acc[t] = acc[t] + o;
      }
  }
  if (wave == 0) {
    float* out = p.partial + ((size_t)chunk * gridDim.x + pair) * 27 * 256;
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) out[t * 256 + (kg * 4 + j) * 16 + m] = acc[t][j];      // [tap][co][ci]
  }
}

// dW[co][ci][tap] (+)= sum over chunks.  A block owns 8 consecutive elements of the partial layout
// [pair][tap][co16][ci16] (64 contiguous bytes per chunk); its 16 lanes per element add chunks l, l+16, ... in order and the
// 16 lane sums are added in lane order: a fixed summation tree, independent of the launch.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cout,
                                                          int Cin, int CinPad, int nchunk, int accumulate) {
  __shared__ float red[16][17];
  const int ncit = CinPad / 16, npairs = (Cout / 16) * ncit;
  const long long E = (long long)npairs * 27 * 256;
  const int el = threadIdx.x & 15, l = threadIdx.x >> 4;     // 16 consecutive elements (one 64-byte segment) x 16 chunk lanes
  const long long e = (long long)blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < E) {
    int c = l;
    for (; c + 48 < nchunk; c += 64) {                        // four independent loads in flight, added in chunk order
      const float v0 = partial[(size_t)c * E + e], v1 = partial[(size_t)(c + 16) * E + e];
      const float v2 = partial[(size_t)(c + 32) * E + e], v3 = partial[(size_t)(c + 48) * E + e];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; c < nchunk; c += 16) s += partial[(size_t)c * E + e];
  }
  red[el][l] = s;
  __syncthreads();
  if (threadIdx.x < 16 && (long long)blockIdx.x * 16 + threadIdx.x < E) {
    const long long ee = (long long)blockIdx.x * 16 + threadIdx.x;
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[threadIdx.x][k];
    const int ci16 = ee % 16, co16 = (ee / 16) % 16, tap = (ee / 256) % 27, pair = ee / (256 * 27);
    const int co = (pair / ncit) * 16 + co16, ci = (pair % ncit) * 16 + ci16;
    if (ci < Cin) {
      float* o = dw + ((long long)co * Cin + ci) * 27 + tap;
      *o = accumulate ? *o + t : t;
    }
  }
}

// nchunk <= 16 (the wide layers: 16 or more channel-tile pairs share the 256 workgroups): one thread per element walks the chunks in
// order.  The same sums as the kernel above -- there every chunk lane holds at most one chunk and the lane sums are added in lane
// order -- without 27 648 blocks of 16 elements each (75 -> ~10 us on 128 -> 128).
__global__ __launch_bounds__(256) void wgrad_reduce_few_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cout,
                                                              int Cin, int CinPad, int nchunk, int accumulate) {
  const int ncit = CinPad / 16, npairs = (Cout / 16) * ncit;
  const long long E = (long long)npairs * 27 * 256;
  const long long ee = (long long)blockIdx.x * 256 + threadIdx.x;
  if (ee >= E) return;
  float v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = c < nchunk ? partial[(size_t)c * E + ee] : 0.f;
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) t += v[c];
  const int ci16 = ee % 16, co16 = (ee / 16) % 16, tap = (ee / 256) % 27, pair = ee / (256 * 27);
  const int co = (pair / ncit) * 16 + co16, ci = (pair % ncit) * 16 + ci16;
  if (ci < Cin) {
    float* o = dw + ((long long)co * Cin + ci) * 27 + tap;
    *o = accumulate ? *o + t : t;
  }
}

static void wgrad_plan(int N, int D, int H, int W, int Cout, int CinPad, int* nitems, int* nchunk, int* ipc, int* nyt) {
  const int ty = wg_ty((W + 31) / 32 * 32);
  *nyt = (H + ty - 1) / ty;
  *nitems = N * D * *nyt;
  const int npairs = (Cout / 16) * (CinPad / 16);
  // One workgroup per compute unit (139 KB of LDS each), so the grid should be ONE round of at most 256 workgroups with as many
  // items each as that allows: 1024 workgroups (four rounds, each with its own ring fill, cross-wave reduction and 27 KB of
  // partial sums) measured 135 us on 16 -> 16 @128^3 x 2 views against 99 us for 256, 99 -> 58 us on 32 -> 32 @64^3, 68 -> 38 us
  // on 64 -> 64 @32^3; 257 .. 320 workgroups (a second, nearly empty round) are the worst case (tools/wgrad_time.py).
  static int target = -1;
  if (target < 0) target = exp_env("AMX_WGRAD_WGS") ? atoi(exp_env("AMX_WGRAD_WGS")) : 256;
  int nc = target / npairs;                    // floor: never spill into a second round
  if (nc > *nitems) nc = *nitems;
  if (nc < 1) nc = 1;
  *ipc = (*nitems + nc - 1) / nc;
  *nchunk = (*nitems + *ipc - 1) / *ipc;
}

size_t wgrad_scratch_bytes(int N, int D, int H, int W, int Cout, int CinPad) {
  int nitems, nchunk, ipc, nyt;
  wgrad_plan(N, D, H, W, Cout, CinPad, &nitems, &nchunk, &ipc, &nyt);
  return (size_t)nchunk * (Cout / 16) * (CinPad / 16) * 27 * 256 * sizeof(float);
}

hipError_t launch_wgrad(WgradParams p, int CinReal, float* dw, int accumulate, void* scratch, int precision, hipStream_t st) {
  const int CinPad = p.C0 + p.C1;
  int nitems, nchunk, ipc, nyt;
  wgrad_plan(p.N, p.D, p.H, p.W, p.Cout, CinPad, &nitems, &nchunk, &ipc, &nyt);
  p.partial = (float*)scratch;
  p.nchunk = nchunk; p.items_per_chunk = ipc; p.nitems = nitems; p.nyt = nyt;
  const int Wp = (p.W + 31) / 32 * 32;
  if (Wp > 128) return hipErrorInvalidValue;                       // WG_RS covers W <= 128
  const int ty = wg_ty(Wp), rs = wg_rs(Wp);
  const size_t dyb = (size_t)ty * 16 * rs * 2, plb = (size_t)(ty + 2) * 16 * rs * 2;
  const size_t red = (size_t)4 * 27 * 64 * 4 * sizeof(float);
  const bool ring = p.D >= 3;
  size_t lds = ring ? 2 * dyb + 4 * plb : dyb + 3 * plb;
  if (lds < red) lds = red;
  const int npairs = (p.Cout / 16) * (CinPad / 16);
#define AMX_WG2(T, R, TY, RS)                                                                                                \
  {                                                                                                                          \
    static bool done = false;                                                                                                \
    if (!done) {                                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)conv3d_wgrad_kernel<T, R, TY, RS>,                                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                            \
      if (e != hipSuccess) return e;                                                                                         \
      done = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL((conv3d_wgrad_kernel<T, R, TY, RS>), dim3(npairs, nchunk), dim3(512), lds, st, p, Wp);                \
  }
#define AMX_WG(T, R)                                                                                                         \
  {                                                                                                                          \
    if (Wp <= 32) AMX_WG2(T, R, 16, 40) else if (Wp <= 64) AMX_WG2(T, R, 8, 72) else AMX_WG2(T, R, 4, 136)                   \
  }
  if (precision == 0) {
    if (ring) AMX_WG(f16, true) else AMX_WG(f16, false)
  } else {
    if (ring) AMX_WG(bf16, true) else AMX_WG(bf16, false)
  }
#undef AMX_WG2
#undef AMX_WG
  const long long E = (long long)npairs * 27 * 256;
  if (nchunk <= 16)
    hipLaunchKernelGGL(wgrad_reduce_few_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, (const float*)scratch, dw, p.Cout,
                       CinReal, CinPad, nchunk, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((E + 15) / 16)), dim3(256), 0, st, (const float*)scratch, dw, p.Cout,
                       CinReal, CinPad, nchunk, accumulate);
  return hipGetLastError();
}

}  // namespace amx
