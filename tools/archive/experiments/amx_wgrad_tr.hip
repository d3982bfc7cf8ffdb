// anatomix_amd -- weight gradient of nn.Conv3d(k3, reflect) for the wide levels (W a multiple of 64: the 128^3 and 64^3 levels,
// 90 % of the weight-gradient FLOPs of the contrastive step): gfx950 TRANSPOSE-READ formulation.
//
//   dW[co][ci][kz][ky][kx] = sum over voxels of dY[vox][co] * In[vox + tap][ci]          (amx_wgrad.hip has the full statement)
//
// The reduction index of this GEMM is the voxel, while the tensors are channels-last -- an MFMA lane needs 8 consecutive
// VOXELS of one channel.  amx_wgrad.hip transposes the tiles while staging them (global -> registers -> eight 16-bit-pair LDS
// writes per 16 bytes, ~150 VALU + 24 ds_write per thread and item, all on the MFMA waves, whose own global loads block them while
// the memory queue is full): 55-60 % of its LDS cycles are bank conflicts and the waves wait 59 % of the time.  Here
//   * the tiles are staged AS THEY ARE (channels-last, 32 bytes per voxel) by LDS-DMA from LOADER waves -- no VGPR round trip,
//     no VALU, reflect padding / the nearest upsample of the concat segment resolved in the per-lane source address;
//   * the MFMA waves read their operands with ds_read_b64_tr_b16: the 16 lanes of a group fetch a [4 voxels][16 channels]
//     block and each lane receives one CHANNEL's four voxels.  Two such reads make an 8-voxel fragment; taking the 32 voxels
//     of a K-block in the order {4g..4g+3, 16+4g..16+4g+3} for lane group g (the same order for both operands -- the sum over
//     voxels does not care) makes every instruction read 16 consecutive voxels = 512 contiguous bytes: conflict-free;
//   * the three x taps are three base addresses 32 bytes apart (no register shuffling), the three y taps share halo rows
//     between the two output rows a wave owns;
//   * loader and MFMA waves meet through LDS counters (amx_device.h), bounded spins.
// Workgroup: 4 MFMA waves + 2 loader waves on a 4-row x 64-voxel tile marching along z (ring of 4 input planes, 2 dY tiles:
// 68 KiB of LDS, two workgroups per CU so that one's load latency hides behind the other's MFMAs).  Same partial-sum layout as
// amx_wgrad.hip, same fixed-order reduce kernel: deterministic, no atomics.
#include <stdlib.h>

#include "amx_device.h"

namespace amx {

typedef __attribute__((address_space(1))) const void* wt_gptr_t;
typedef __attribute__((address_space(3))) void* wt_lptr_t;
typedef __attribute__((ext_vector_type(4))) short wt_s16x4;

template <int TX_>
struct WgTrT {
  static constexpr int TY = 4, TX = TX_, HY = TY + 2, HX = TX + 2;
  static constexpr int NCW = TX / 32 * 2, NLW = TX / 32, NPS = 4, NDB = 2;
  static constexpr int DYB = TY * TX * 32;                                  // 8 KiB
  static constexpr int NDY = DYB / 1024;                                    // DMA instructions per dY tile
  static constexpr int NPL = (HY * HX * 32 + 1023) / 1024;                  // ... per input plane (the last one partial)
  static constexpr int PLB = NPL * 1024;                                    // 13 KiB
  static constexpr int FLAGOFF = NDB * DYB + NPS * PLB;
  static constexpr int RED = (NCW / 2) * 27 * 64 * 16;                       // final reduction over the MFMA waves
  static constexpr int LDS = (FLAGOFF + 64) > RED ? (FLAGOFF + 64) : RED;
  static constexpr int MAXJ = (NDY + 3 * NPL + NLW - 1) / NLW;              // most DMA instructions one loader issues for one item
};
// 32-wide tiles: 2 MFMA waves + 1 loader, 36 KiB of LDS -> four workgroups per CU.  The kernel is bound by the latency of its
// tile requests (the ring holds one item ahead), so what counts is how many independent request streams a CU runs: measured on
// 16||up32 -> 16 @128^3 x 2: 64-wide tiles, two workgroups per CU 434 us; 32-wide, four per CU: see DESIGN.md.
typedef WgTrT<32> WgTr;

template <typename T>
__global__ __launch_bounds__((WgTr::NCW + WgTr::NLW) * 64) void conv3d_wgrad_tr_kernel(const WgradParams p, int nxt) {
  typedef WgTr C;
  typedef typename Ops<T>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dys = smem;                                                 // [NDB][TY][TX][16 co]
  char* ins = smem + C::NDB * C::DYB;                               // [NPS][HY][HX][16 ci]
  int* ready = (int*)(smem + C::FLAGOFF);                           // per loader wave: items landed (8 slots, unused = INT_MAX)
  int* done = ready + 8;                                            // per MFMA wave: items finished
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int ncit = (p.C0 + p.C1) / 16;
  const int pair = blockIdx.x, cot = pair / ncit, cit = pair % ncit;
  const int chunk = blockIdx.y;
  const bool seg1 = cit * 16 >= p.C0;
  const int ci0 = seg1 ? cit * 16 - p.C0 : cit * 16;
  const int sh = seg1 ? p.up_shift : 0;
  const int item0 = chunk * p.items_per_chunk;
  const int item1 = item0 + p.items_per_chunk < p.nitems ? item0 + p.items_per_chunk : p.nitems;
  // item = ((n * nyt + yt) * nxt + xt) * D + z   (z fastest: consecutive items march along z through one tile column)
  auto decode = [&](int item, int& z, int& xt, int& yt, int& n) {
    z = item % p.D;
    int r = item / p.D;
    xt = r % nxt;
    r /= nxt;
    yt = r % p.nyt;
    n = r / p.nyt;
  };
  if (tid < 16) ready[tid] = ((tid >= C::NLW && tid < 8) || tid >= 8 + C::NCW) ? 0x7fffffff : 0;
  __syncthreads();

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave >= C::NCW) {
    // =============================== loader wave lw ===============================
    // The DMA instructions of an item form one list: [dY tile: NDY] then [each plane that is not resident yet: NPL]; loader lw
    // issues the entries j = lw, lw + NLW, ...  One instruction = 32 voxels x 32 bytes (lane = voxel * 2 + half).
    const int lw = wave - C::NCW;
    const int half = lane & 1, vl = lane >> 1;
    const unsigned a_ready = lds_addr(ready + lw), a_done = lds_addr(done);
    const long long sxs = seg1 ? p.s1x : p.s0x, sys_ = seg1 ? p.s1y : p.s0y, szs = seg1 ? p.s1z : p.s0z, sns = seg1 ? p.s1n : p.s0n;
    const char* srcb = (seg1 ? p.src1 : p.src0) + ci0 * 2 + half * 16;
    int resident[C::NPS] = {-1, -1, -1, -1};                        // plane index held by each ring slot (wave-uniform)
    int cur_tile = -1;
    int plx[C::NPL];                                                // per plane instruction: this lane's x offset in bytes (tile constant)
    for (int item = item0; item < item1; ++item) {
      int z, xt, yt, n;
      decode(item, z, xt, yt, n);
      const int y0 = yt * C::TY, x0 = xt * C::TX;
      const int k = item - item0;
      const bool cold = item / p.D != cur_tile;
      if (cold) {                                                   // new tile column: nothing of the ring is reusable
        cur_tile = item / p.D;
#pragma unroll
        for (int s = 0; s < C::NPS; ++s) resident[s] = -1;
#pragma unroll
        for (int j = 0; j < C::NPL; ++j) {
          const int hv = j * 32 + vl, hx = hv % C::HX;
          plx[j] = (reflect_clamp(x0 + hx - 1, p.W) >> sh) * (int)sxs;
        }
      }
      // steady state: the slots written now (dY buffer k & 1, the slot of plane z - 3) were last read by item k - 2; a new
      // tile column rewrites every plane slot, including those item k - 1 is reading
      const int need_done = cold ? k : k - 1;
      if (need_done > 0) {
        int polls = 0;
        while (__builtin_amdgcn_readfirstlane(flag_min8_asm(a_done)) < need_done) {
          __builtin_amdgcn_s_sleep(1);
          if (++polls > (1 << 18)) break;                           // bounded spin: a protocol error must not hang the GPU
        }
      }
      int issued = 0, j = 0;                                        // j: running index in the item's instruction list
      {
        const char* b = p.dy + (long long)n * p.yn + (long long)z * p.yz + cot * 32 + half * 16;
        char* dst = dys + (k & 1) * C::DYB;
#pragma unroll
        for (int i = 0; i < C::NDY; ++i, ++j) {
          if ((j % C::NLW) != lw) continue;
          const int v = i * 32 + vl, row = v / C::TX, x = v % C::TX;
          __builtin_amdgcn_global_load_lds((wt_gptr_t)(b + (long long)(y0 + row) * p.yy + (long long)(x0 + x) * p.yx),
                                           (wt_lptr_t)(dst + i * 1024), 16, 0, 0);
          ++issued;
        }
      }
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        const int zz = reflect_clamp(z + kz - 1, p.D), slot = zz & 3;
        if (resident[slot] == zz) continue;
        resident[slot] = zz;
        const char* b = srcb + (long long)n * sns + (long long)(zz >> sh) * szs;
        char* dst = ins + slot * C::PLB;
#pragma unroll
        for (int i = 0; i < C::NPL; ++i, ++j) {
          if ((j % C::NLW) != lw) continue;
          const int hv = i * 32 + vl, hr = hv / C::HX;
          ++issued;                                                 // (a fully masked instruction is still counted: see the wait below)
          if (hv < C::HY * C::HX)
            __builtin_amdgcn_global_load_lds((wt_gptr_t)(b + (long long)(reflect_clamp(y0 + hr - 1, p.H) >> sh) * sys_ + plx[i]),
                                             (wt_lptr_t)(dst + i * 1024), 16, 0, 0);
        }
      }
      // Item k can only be requested once the MFMA waves are inside item k - 1 (see need_done), so there is never a second
      // request to overlap with: wait for this one and publish it at once.
      (void)issued;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      flag_store_asm(a_ready, k + 1);
    }
  } else {
    // =============================== MFMA wave: x block xb, rows r0, r0 + 1 ===============================
    const int t = lane & 15, g = lane >> 4;
    constexpr int NXB = C::TX / 32;
    const int xb = wave % NXB, r0 = (wave / NXB) * 2;
    // per-lane part of every transpose read: voxel 4 g + (t >> 2) of the 16-voxel group, channel quad t & 3
    const int lanepart = (4 * g + (t >> 2)) * 32 + (t & 3) * 8;
    const int a_off = (r0 * C::TX + xb * 32) * 32 + lanepart;                // dY tile: row r0 (row r0 + 1: + TX * 32)
    const int b_off = (r0 * C::HX + xb * 32) * 32 + lanepart;                // plane: halo row r0, tap kx = 0
    auto frag = [&](const char* base) -> vec8 {                               // voxels {4g.., 16 + 4g..} of the K-block at `base`
      const wt_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wt_s16x4*)(wt_lptr_t)base);
      const wt_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wt_s16x4*)(wt_lptr_t)(base + 512));
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(vec8, v);
    };
    for (int item = item0; item < item1; ++item) {
      const int z = item % p.D, k = item - item0;
      {
        int polls = 0;
        while (true) {
          int m = flag_load(ready);
#pragma unroll
          for (int i = 1; i < C::NLW; ++i) {
            const int r = flag_load(ready + i);
            m = r < m ? r : m;
          }
          if (m >= k + 1) break;
          __builtin_amdgcn_s_sleep(1);
          if (++polls > (1 << 18)) break;
        }
        asm volatile("" ::: "memory");
      }
      const char* dyb = dys + (k & 1) * C::DYB;
      const vec8 af0 = frag(dyb + a_off), af1 = frag(dyb + a_off + C::TX * 32);
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        const char* pl = ins + (reflect_clamp(z + kz - 1, p.D) & 3) * C::PLB + b_off;
#pragma unroll
        for (int hr = 0; hr < 4; ++hr)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const vec8 bf = frag(pl + (hr * C::HX + kx) * 32);
            if (hr < 3) acc[(kz * 3 + hr) * 3 + kx] = Ops<T>::mfma(af0, bf, acc[(kz * 3 + hr) * 3 + kx]);
            if (hr > 0) acc[(kz * 3 + hr - 1) * 3 + kx] = Ops<T>::mfma(af1, bf, acc[(kz * 3 + hr - 1) * 3 + kx]);
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // every read of this item has returned: its buffers may be refilled
      flag_store(done + wave, k + 1);
    }
  }
  // ---- sum the 4 MFMA waves through LDS (2 rounds; the loaders only take part in the barriers), wave 0 writes the partial
  float* red = (float*)smem;                                        // [NCW / 2 waves][27][64][4] floats
  for (int hw = C::NCW / 2; hw >= 1; hw >>= 1) {
    __syncthreads();
    if (wave >= hw && wave < 2 * hw)
#pragma unroll
      for (int t2 = 0; t2 < 27; ++t2) *(f32x4*)(red + (((size_t)(wave - hw) * 27 + t2) * 64 + lane) * 4) = acc[t2];
    __syncthreads();
    if (wave < hw)
#pragma unroll
      for (int t2 = 0; t2 < 27; ++t2) {
        const f32x4 o = *(const f32x4*)(red + (((size_t)wave * 27 + t2) * 64 + lane) * 4);
        acc[t2] = acc[t2] + o;
      }
  }
  if (wave == 0) {
    const int m = lane & 15, kg = lane >> 4;
    float* out = p.partial + ((size_t)chunk * gridDim.x + pair) * 27 * 256;
#pragma unroll
    for (int t2 = 0; t2 < 27; ++t2)
#pragma unroll
      for (int j = 0; j < 4; ++j) out[t2 * 256 + (kg * 4 + j) * 16 + m] = acc[t2][j];      // [tap][co][ci]
  }
}

bool wgrad_tr_eligible(const WgradParams& p) {
  static int off = -1;
  if (off < 0) off = getenv("AMX_NO_WGRAD_TR") ? 1 : 0;
  return !off && p.W % WgTr::TX == 0 && p.H % WgTr::TY == 0 && p.D >= 3;
}

void wgrad_tr_plan(int N, int D, int H, int W, int Cout, int CinPad, int* nitems, int* nchunk, int* ipc, int* nyt, int* nxt) {
  *nyt = H / WgTr::TY;
  *nxt = W / WgTr::TX;
  *nitems = N * D * *nyt * *nxt;
  const int npairs = (Cout / 16) * (CinPad / 16);
  int nc = (4096 + npairs - 1) / npairs;                            // several workgroups per CU, a few rounds
  // chunks are whole tile columns (multiples of D items) when there are enough of them: the ring then never restarts mid-column
  const int cols = *nitems / D;
  if (nc > cols) nc = cols;
  if (nc < 1) nc = 1;
  *ipc = (cols + nc - 1) / nc * D;
  *nchunk = (*nitems + *ipc - 1) / *ipc;
}

hipError_t launch_wgrad_tr(WgradParams p, int nxt, int npairs, int precision, hipStream_t st) {
#define AMX_WT(T)                                                                                                            \
  {                                                                                                                          \
    static bool done = false;                                                                                                \
    if (!done) {                                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)conv3d_wgrad_tr_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, WgTr::LDS); \
      if (e != hipSuccess) return e;                                                                                         \
      done = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL((conv3d_wgrad_tr_kernel<T>), dim3(npairs, p.nchunk), dim3((WgTr::NCW + WgTr::NLW) * 64), WgTr::LDS, st, p, nxt); \
  }
  if (precision == 0) AMX_WT(f16) else AMX_WT(bf16)
#undef AMX_WT
  return hipGetLastError();
}

}  // namespace amx
