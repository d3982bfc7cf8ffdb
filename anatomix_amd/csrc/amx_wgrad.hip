// anatomix_amd -- weight gradient of nn.Conv3d(k3, stride 1, padding='same', padding_mode='reflect')
// (the wgrad half of the backward of anatomix/model/network.py:310-461, needed by the contrastive step,
// pretraining/models/supcl_model.py:603-661).
//
//   dW[co][ci][kz][ky][kx] = sum over (n, z, y, x) of  dY[n][z][y][x][co] * In[n][r(z+kz-1)][r(y+ky-1)][r(x+kx-1)][ci]
//   (r = reflect; In may be the concatenation of a full-resolution segment and a nearest-upsampled half-resolution
//    segment, exactly as the forward kernels read it).
//
// A GEMM whose reduction index is the VOXEL: D[co][ci] += A[co][k] B[k][ci] per tap with k = 32 consecutive x of
// one row, v_mfma_f32_16x16x32.  Activations are channels-last, so the 8 consecutive k a lane needs are 8 different
// voxels: fragments are gathered from an LDS tile with 16-bit reads.  Two tricks keep that affordable:
//   * the three x taps of one (kz, ky) read overlapping windows x-1 .. x+8: 10 values are gathered once and the kx = 1
//     fragment is formed with four v_alignbit (34 % of the reads of three separate gathers);
//   * one 32-byte pad slot after every 8 voxels of an LDS row shifts the four k-groups of a fragment onto different
//     banks (8 voxels x 32 B = 256 B would otherwise alias them).
// Work decomposition: grid (co-tile x ci-tile pairs, spatial chunks).  A workgroup walks its chunk of (n, 4-row tile, z)
// items (z fastest), staging per item the dY rows and the 3 x 6 halo rows of In (reflect / upsample resolved at staging);
// its 8 waves split the item's K-blocks, keep 27 accumulators each for the whole chunk, and are summed through LDS
// at the end.  Partials [chunk][pair][27][16][16] fp32 are then added in chunk order by wgrad_reduce_kernel:
// deterministic, no atomics.
#include "amx_device.h"

namespace amx {

constexpr int WG_TY = 4;

__device__ __forceinline__ int padx(int x) { return x + (x >> 3); }

// RING: the chunk's items march along z at a fixed (n, row tile); the 3 input z-planes of an item live in a ring of 4
// LDS slots (slot = plane & 3), so a step stages ONE new plane (+ the next dY rows) instead of three, and those global
// loads are issued into registers BEFORE the item's MFMA sweep and written to LDS after it (software prefetch; one
// barrier per item).  Needs 4 planes + 2 dY buffers in LDS (W <= 128); otherwise every item is staged cold.
template <typename T, bool RING>
__global__ __launch_bounds__(512) void conv3d_wgrad_kernel(const WgradParams p, int Wp) {
  typedef typename Ops<T>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NDB = RING ? 2 : 1;
  constexpr int NPL = 4, NDY = 2;                                   // prefetch registers per thread (Wp <= 128 when RING)
  const int PW = padx(Wp) + 1, PWH = padx(Wp + 2) + 2;              // padded row lengths (voxel slots) of dY / In rows
  const int dy_bytes = WG_TY * PW * 32, plane_bytes = (WG_TY + 2) * PWH * 32;
  char* dys = smem;                                                 // [NDB][WG_TY][PW][32 B]
  char* ins = smem + (size_t)NDB * dy_bytes;                        // [RING ? 4 : 3][WG_TY + 2][PWH][32 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;

  const int ncit = (p.C0 + p.C1) / 16;
  const int pair = blockIdx.x, cot = pair / ncit, cit = pair % ncit;
  const int chunk = blockIdx.y;
  const bool seg1 = cit * 16 >= p.C0;
  const int ci0 = seg1 ? cit * 16 - p.C0 : cit * 16;
  const int sh = seg1 ? p.up_shift : 0;

  // one 16-byte piece of the dY tile / of one halo plane, by flat index t (rows beyond the volume and x >= W: zero)
  auto load_dy = [&](int t, int n, int z, int y0) -> uint4 {
    const int half = t & 1, x = (t >> 1) % Wp, row = (t >> 1) / Wp;
    if (row < WG_TY && x < p.W && y0 + row < p.H)
      return *(const uint4*)(p.dy + (long long)n * p.yn + (long long)z * p.yz + (long long)(y0 + row) * p.yy + (long long)x * p.yx +
                             cot * 32 + half * 16);
    return make_uint4(0, 0, 0, 0);
  };
  auto store_dy = [&](int t, int buf, uint4 v) {
    const int half = t & 1, x = (t >> 1) % Wp, row = (t >> 1) / Wp;
    if (row < WG_TY) *(uint4*)(dys + (size_t)buf * dy_bytes + ((size_t)row * PW + padx(x)) * 32 + half * 16) = v;
  };
  auto load_pl = [&](int t, int n, int zz, int y0) -> uint4 {       // zz: full-resolution plane index (already reflected)
    const int half = t & 1, xh = (t >> 1) % (Wp + 2), hr = (t >> 1) / (Wp + 2);
    if (hr < WG_TY + 2 && xh <= p.W + 1) {
      const int z2 = zz >> sh, yy = reflect_clamp(y0 + hr - 1, p.H) >> sh, xx = reflect_clamp(xh - 1, p.W) >> sh;
      const char* src = seg1 ? p.src1 + (long long)n * p.s1n + (long long)z2 * p.s1z + (long long)yy * p.s1y + (long long)xx * p.s1x
                             : p.src0 + (long long)n * p.s0n + (long long)z2 * p.s0z + (long long)yy * p.s0y + (long long)xx * p.s0x;
      return *(const uint4*)(src + ci0 * 2 + half * 16);
    }
    return make_uint4(0, 0, 0, 0);
  };
  auto store_pl = [&](int t, int slot, uint4 v) {
    const int half = t & 1, xh = (t >> 1) % (Wp + 2), hr = (t >> 1) / (Wp + 2);
    if (hr < WG_TY + 2) *(uint4*)(ins + (size_t)slot * plane_bytes + ((size_t)hr * PWH + padx(xh)) * 32 + half * 16) = v;
  };
  const int ndy = WG_TY * Wp * 2, npl = (WG_TY + 2) * (Wp + 2) * 2;

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nxb = Wp / 32;
  const int item0 = chunk * p.items_per_chunk;
  const int item1 = item0 + p.items_per_chunk < p.nitems ? item0 + p.items_per_chunk : p.nitems;
  int db = 0;                                                       // dY buffer of the current item
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
      for (int t = tid; t < ndy; t += 512) store_dy(t, db, load_dy(t, n, z, y0));
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        if (RING && kz == 2 && slot[2] == slot[0]) continue;        // z = 0 or D-1: the reflected plane is already staged
        const int zz = reflect_clamp(z + kz - 1, p.D);
        for (int t = tid; t < npl; t += 512) store_pl(t, slot[kz], load_pl(t, n, zz, y0));
      }
      __syncthreads();
    }
    // ---- prefetch the next item's new data into registers (RING, next item = z + 1 of the same tile)
    const bool has_next = RING && item + 1 < item1 && z + 1 < p.D;
    const bool next_plane = has_next && z + 2 < p.D;
    uint4 rdy[NDY], rpl[NPL];
    if (has_next) {
#pragma unroll
      for (int k = 0; k < NDY; ++k) rdy[k] = tid + k * 512 < ndy ? load_dy(tid + k * 512, n, z + 1, y0) : make_uint4(0, 0, 0, 0);
      if (next_plane)
#pragma unroll
        for (int k = 0; k < NPL; ++k) rpl[k] = tid + k * 512 < npl ? load_pl(tid + k * 512, n, z + 2, y0) : make_uint4(0, 0, 0, 0);
    }
    // ---- K-blocks of this item: (row, xb) -> 32 voxels
    const char* dyb = dys + (size_t)db * dy_bytes;
    for (int kb = wave; kb < WG_TY * nxb; kb += 8) {
      const int row = kb / nxb, xb = kb % nxb;
      const int X0 = xb * 32 + kg * 8;                             // multiple of 8
      const unsigned short* ap = (const unsigned short*)(dyb + ((size_t)row * PW + padx(X0)) * 32) + m;
      unsigned a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = (unsigned)ap[(2 * j) * 16] | ((unsigned)ap[(2 * j + 1) * 16] << 16);
      const vec8 af = __builtin_bit_cast(vec8, a);
#pragma unroll
      for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const unsigned short* bp =
              (const unsigned short*)(ins + (size_t)slot[kz] * plane_bytes + ((size_t)(row + ky) * PWH + padx(X0)) * 32) + m;
          // window of 10 halo voxels X0 .. X0 + 9 (= x - 1 .. x + 8); slot of element i: i + (i >> 3)
          unsigned pk[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            const int e0 = 2 * i, e1 = 2 * i + 1;
            pk[i] = (unsigned)bp[(e0 + (e0 >> 3)) * 16] | ((unsigned)bp[(e1 + (e1 >> 3)) * 16] << 16);
          }
          unsigned b0[4] = {pk[0], pk[1], pk[2], pk[3]};                                   // kx = 0: x - 1 ..
          unsigned b2[4] = {pk[1], pk[2], pk[3], pk[4]};                                   // kx = 2: x + 1 ..
          unsigned b1[4];                                                                  // kx = 1: x ..
#pragma unroll
          for (int i = 0; i < 4; ++i) b1[i] = __builtin_amdgcn_alignbit(pk[i + 1], pk[i], 16);
          const int t0 = (kz * 3 + ky) * 3;
          acc[t0] = Ops<T>::mfma(af, __builtin_bit_cast(vec8, b0), acc[t0]);
          acc[t0 + 1] = Ops<T>::mfma(af, __builtin_bit_cast(vec8, b1), acc[t0 + 1]);
          acc[t0 + 2] = Ops<T>::mfma(af, __builtin_bit_cast(vec8, b2), acc[t0 + 2]);
        }
    }
    if (has_next) {
      // slot (z + 2) & 3 and the other dY buffer are not read by this item, and the item that read them last ended with
      // the barrier below
#pragma unroll
      for (int k = 0; k < NDY; ++k)
        if (tid + k * 512 < ndy) store_dy(tid + k * 512, db ^ 1, rdy[k]);
      if (next_plane)
#pragma unroll
        for (int k = 0; k < NPL; ++k)
          if (tid + k * 512 < npl) store_pl(tid + k * 512, (z + 2) & 3, rpl[k]);
      db ^= 1;
      __syncthreads();
    }
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
// [pair][tap][co16][ci16] (32 contiguous bytes per chunk); its 32 lanes per element add chunks l, l+32, ... in order and the
// 32 lane sums are added in lane order: a fixed summation tree, independent of the launch.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cout,
                                                          int Cin, int CinPad, int nchunk, int accumulate) {
  __shared__ float red[8][33];
  const int ncit = CinPad / 16, npairs = (Cout / 16) * ncit;
  const long long E = (long long)npairs * 27 * 256;
  const int el = threadIdx.x & 7, l = threadIdx.x >> 3;
  const long long e = (long long)blockIdx.x * 8 + el;
  float s = 0.f;
  if (e < E)
    for (int c = l; c < nchunk; c += 32) s += partial[(size_t)c * E + e];
  red[el][l] = s;
  __syncthreads();
  if (threadIdx.x < 8 && (long long)blockIdx.x * 8 + threadIdx.x < E) {
    const long long ee = (long long)blockIdx.x * 8 + threadIdx.x;
    float t = 0.f;
    for (int k = 0; k < 32; ++k) t += red[threadIdx.x][k];
    const int ci16 = ee % 16, co16 = (ee / 16) % 16, tap = (ee / 256) % 27, pair = ee / (256 * 27);
    const int co = (pair / ncit) * 16 + co16, ci = (pair % ncit) * 16 + ci16;
    if (ci < Cin) {
      float* o = dw + ((long long)co * Cin + ci) * 27 + tap;
      *o = accumulate ? *o + t : t;
    }
  }
}

static void wgrad_plan(int N, int D, int H, int W, int Cout, int CinPad, int* nitems, int* nchunk, int* ipc, int* nyt) {
  *nyt = (H + WG_TY - 1) / WG_TY;
  *nitems = N * D * *nyt;
  const int npairs = (Cout / 16) * (CinPad / 16);
  int nc = (1024 + npairs - 1) / npairs;
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
  const int PW = Wp + (Wp >> 3) + 1, PWH = (Wp + 2) + ((Wp + 2) >> 3) + 2;
  const size_t red = (size_t)4 * 27 * 64 * 4 * sizeof(float);
  const size_t lds_ring = ((size_t)2 * WG_TY * PW + (size_t)4 * (WG_TY + 2) * PWH) * 32;
  const size_t lds_cold = ((size_t)WG_TY * PW + (size_t)3 * (WG_TY + 2) * PWH) * 32;
  const bool ring = Wp <= 128 && lds_ring <= 160 * 1024 && p.D >= 3;
  size_t lds = ring ? lds_ring : lds_cold;
  if (lds < red) lds = red;
  if (lds > 160 * 1024) return hipErrorInvalidValue;              // W <= ~160
  const int npairs = (p.Cout / 16) * (CinPad / 16);
#define AMX_WG(T, R)                                                                                                         \
  {                                                                                                                          \
    static bool done = false;                                                                                                \
    if (!done) {                                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)conv3d_wgrad_kernel<T, R>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         160 * 1024);                                                                        \
      if (e != hipSuccess) return e;                                                                                         \
      done = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL((conv3d_wgrad_kernel<T, R>), dim3(npairs, nchunk), dim3(512), lds, st, p, Wp);                        \
  }
  if (precision == 0) {
    if (ring) AMX_WG(f16, true) else AMX_WG(f16, false)
  } else {
    if (ring) AMX_WG(bf16, true) else AMX_WG(bf16, false)
  }
#undef AMX_WG
  const long long E = (long long)npairs * 27 * 256;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((E + 7) / 8)), dim3(256), 0, st, (const float*)scratch, dw, p.Cout,
                     CinReal, CinPad, nchunk, accumulate);
  return hipGetLastError();
}

}  // namespace amx
