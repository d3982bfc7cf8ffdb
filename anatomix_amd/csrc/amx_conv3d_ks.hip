// anatomix_amd -- conv3d 3x3x3 reflect, third-generation kernel for the DEEP levels (<= 32^3): weights stationary in REGISTERS,
// the K dimension split over the waves of a workgroup, fixed-order reduction through LDS.
//
// Why (profiles/r04_layers_6m.txt): at the 32^3 .. 8^3 levels conv3d_k3_v2 spends its time FILLING LDS -- every
// 16-channel stage of every brick re-stages 28 .. 56 KiB of packed weight fragments next to ~12 .. 20 KiB of halo, the waves of a
// workgroup split the brick's voxels and all read the same weight fragments from LDS (0.75 .. 1 LDS read per MFMA), and a stage is
// only 28 .. 112 MFMAs per wave between two workgroup barriers: 7 .. 26 % of the MFMA peak.  This kernel turns the roles around:
//   * a workgroup owns ONE (cout group of 16 Q channels, K slice of 16 CPW KW input channels) and keeps those weights -- the MFMA
//     A operand -- in the REGISTERS of its waves for its whole life (Q * 14 * CPW fragments = up to 224 VGPRs per wave: the
//     register file is the largest on-chip memory, 512 KiB per CU).  Weights never touch LDS and are fetched once per workgroup;
//   * the KW waves of a team split K: wave k owns the input-channel chunks k CPW .. k CPW + CPW - 1 of the slice and sweeps the
//     WHOLE brick (NVT column tiles x Q cout tiles = 16 .. 32 accumulators), so that a B fragment read from LDS feeds Q MFMAs and
//     the LDS carries 1 / Q reads per MFMA;
//   * each wave DMAs the halo planes of ITS chunks into two private LDS buffers (double-buffered over the (brick, chunk) sequence):
//     no wave ever reads bytes another wave staged, so the sweep needs no barrier and no flags -- only the issuing wave's own
//     vmcnt;
//   * the partial sums of the KW waves meet in an LDS scratch tile after the sweep: wave k finalises the column tiles
//     k OWN .. k OWN + OWN - 1 and adds the partials in the FIXED order wave 0, 1, .., KW - 1 (deterministic; no atomics), then
//     applies bias + activation and stores 16-bit channels-last voxels exactly like conv3d_k3_v2;
//   * layers with more (cout group, K slice) pairs than voxels to keep 256 workgroups busy (the 8^3 level) split K across
//     workgroups as well (PART): fp32 partial tensors per slice, summed in slice order by splitk_reduce_kernel (+ bias,
//     activation, range check, 16-bit store).
// Same arithmetic formulation as the other conv kernels (amx_conv3d.hip): A = packed weights [cout group][chunk][step 14][q][lane][8],
// B = activations from a plane-major halo image, 14 paired-tap steps per 16 channels; reflect padding resolved in the gather.
// Replaces nn.Conv3d(k=3, padding_mode='reflect') + folded eval BatchNorm3d + ReLU of /root/reference/anatomix/model/network.py:334-445
// at the levels below 64^3.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "amx_device.h"

namespace amx {

template <int TZ_, int TY_, int TX_, int Q_, int CPW_, int KW_, int TEAMS_>
struct KsCfg {
  static constexpr int TZ = TZ_, TY = TY_, TX = TX_, Q = Q_, CPW = CPW_, KW = KW_, TEAMS = TEAMS_;
  static constexpr int NWAVE = KW * TEAMS;
  static constexpr int LX = TX >= 16 ? 16 : 8, LY = 16 / LX;
  static constexpr int XT = TX / LX, YT = TY / LY;
  static constexpr int NVT = TZ * YT * XT;                 // 16-voxel column tiles per brick (every wave sweeps all of them)
  static constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  static constexpr int PLANE = (HV * 16 + 255) / 256 * 256;   // one 8-channel plane of the halo image
  static constexpr int CHBUF = 2 * PLANE;                  // the halo of one 16-channel chunk
  static constexpr int NPK = (HV + 63) / 64;               // LDS-DMA instructions per plane (64 consecutive halo voxels each)
  static constexpr int HALO_BYTES = NWAVE * 2 * CHBUF;     // two private chunk buffers per wave
  static constexpr int OWN = NVT / KW;                     // column tiles a wave finalises
  static constexpr int SCR_MAX = 160 * 1024 - HALO_BYTES;
  static constexpr int pick_rt() {                         // tiles per owner and reduction round that fit the scratch
    for (int r = OWN; r >= 1; --r)
      if (OWN % r == 0 && TEAMS * KW * (KW - 1) * r * Q * 1024 <= SCR_MAX) return r;
    return 0;
  }
  static constexpr int RT = pick_rt();
  static constexpr int ROUNDS = RT ? OWN / RT : 0;
  static constexpr int SCR_BYTES = TEAMS * KW * (KW - 1) * RT * Q * 1024;
  static constexpr int LDS_BYTES = HALO_BYTES + SCR_BYTES;
  static_assert(NVT % KW == 0, "every wave finalises the same number of column tiles");
  static_assert(RT >= 1, "the reduction scratch must fit behind the halo buffers");
  static_assert(TY % LY == 0 && TX % LX == 0, "the brick must tile into 16-voxel columns");
  static_assert(Q * kSteps * CPW * 4 <= 224, "the stationary A fragments must leave room for accumulators and B fragments");
};

struct KsBrick {   // wave-uniform
  int n, z0, y0, x0;
};

template <typename T, typename C, bool PART>
__global__ __launch_bounds__(C::NWAVE * 64) void conv3d_k3_ks_kernel(const ConvParams p) {
  typedef typename Ops<T>::vec8 vec8;
  constexpr int Q = C::Q, CPW = C::CPW, KW = C::KW, TEAMS = C::TEAMS, NVT = C::NVT;
  constexpr int HY = C::HY, HX = C::HX, PLANE = C::PLANE, CHBUF = C::CHBUF, NPK = C::NPK;
  constexpr int LX = C::LX, LY = C::LY, XT = C::XT, YT = C::YT, OWN = C::OWN, RT = C::RT, ROUNDS = C::ROUNDS;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave / KW, k = wave - team * KW;
  const int li = lane & 15, g = lane >> 4;

  // ---- this workgroup's (cout group, K slice) and its contiguous run of bricks; XCD b % 8 gets a contiguous span of the
  //      (slice, cout group, run) order, so workgroups that share weights / halos share an L2
  const int ncg = p.Cout / (16 * Q);
  const int S = p.kslices > 0 ? p.kslices : 1;
  const int combos = ncg * S;
  const int G = gridDim.x, wpc = G / combos;
  int L = blockIdx.x;
  if ((G & 7) == 0) L = (L & 7) * (G >> 3) + (L >> 3);
  const int combo = L / wpc, run = L - combo * wpc;
  const int slice = combo / ncg, cg = combo - slice * ncg;
  const int nbricks = p.nbz * p.nby * p.nbx * p.N;
  const int b0 = (int)((long long)nbricks * run / wpc), b1 = (int)((long long)nbricks * (run + 1) / wpc);
  const int niter = (b1 - b0 + TEAMS - 1) / TEAMS;
  if (niter <= 0) return;
  const int nchunk = p.C0 >> 4;
  const int ch0 = slice * (KW * CPW) + k * CPW;             // this wave's first input-channel chunk
  const int cs = p.cs0 ? p.cs0 : 32;                        // bytes between a voxel's 16-channel chunks

  // ---- lane-constant LDS read bases (relative to a chunk buffer), as in conv3d_k3_v2 with the wave at the brick's origin
  const int dy = (LX == 16) ? 0 : (li >> 3);
  const int dx = (LX == 16) ? li : (li & 7);
  const int lanebase = (g & 1) * PLANE + (dy * HX + dx) * 16;
  const int hi = g >> 1;
  const int base_d1 = lanebase + hi * 16;
  const int base_dx = lanebase + hi * 16 * HX;
  const int base_dz = lanebase + hi * 16 * HX * HY;
  const int base_d0 = lanebase;

  // ---- DMA lane constants: instruction j of a plane covers halo voxels 64 j .. 64 j + 63
  int pk_pos[NPK], pk_off[NPK];
#pragma unroll
  for (int j = 0; j < NPK; ++j) {
    const int hv = j * 64 + lane;
    const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX;
    pk_pos[j] = hv < C::HV ? (hz | (hy << 8) | ((rem - hy * HX) << 16)) : -1;
    pk_off[j] = 0;
  }
  char* const halo = smem + wave * (2 * CHBUF);
  char* const scratch = smem + C::HALO_BYTES + team * (KW * (KW - 1) * RT * Q * 1024);

  auto decode = [&](int b) {                                // brick index -> sample and origin
    KsBrick r;
    const int bx = b % p.nbx;
    b /= p.nbx;
    const int by = b % p.nby;
    b /= p.nby;
    const int bz = b % p.nbz;
    r.n = b / p.nbz;
    r.z0 = bz * C::TZ;
    r.y0 = by * C::TY;
    r.x0 = bx * C::TX;
    return r;
  };
  auto offsets = [&](const KsBrick& br) {                   // per-lane source offsets of the brick's halo voxels inside the sample
#pragma unroll
    for (int j = 0; j < NPK; ++j) {
      const int pos = pk_pos[j];
      const int gz = reflect_clamp(br.z0 + (pos & 255) - 1, p.D), gy = reflect_clamp(br.y0 + ((pos >> 8) & 255) - 1, p.H);
      const int gx = reflect_clamp(br.x0 + ((pos >> 16) & 255) - 1, p.W);
      pk_off[j] = gz * (int)p.s0z + gy * (int)p.s0y + gx * (int)p.s0x;
    }
  };
#define AMX_DMA16(src, dst) dma16_asm((const void*)(src), (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(dst)))
  auto issue = [&](const KsBrick& br, int cp, int bsel) {   // halo of chunk ch0 + cp of brick br -> chunk buffer bsel
    const char* base = p.src0 + (long long)br.n * p.s0n + (long long)(ch0 + cp) * cs;
    char* dst = halo + bsel * CHBUF;
#pragma unroll
    for (int j = 0; j < NPK; ++j) {
      if (pk_pos[j] >= 0) {
        AMX_DMA16(base + pk_off[j], dst + j * 1024);
        AMX_DMA16(base + pk_off[j] + 16, dst + PLANE + j * 1024);
      }
    }
  };

  // ---- first brick's halo, then the stationary A fragments (in flight together)
  KsBrick cu = decode(b0 + team), nx = cu;
  bool cu_valid = b0 + team < b1;
  if (cu_valid) {
    offsets(cu);
    issue(cu, 0, 0);
  }
  vec8 wa[CPW][kSteps][Q];
  {
    const char* wbase = p.wpk + (((long long)cg * nchunk + ch0) * kSteps * Q) * 1024 + lane * 16;
#pragma unroll
    for (int cp = 0; cp < CPW; ++cp)
#pragma unroll
      for (int s = 0; s < kSteps; ++s)
#pragma unroll
        for (int q = 0; q < Q; ++q) wa[cp][s][q] = *(const vec8*)(wbase + ((cp * kSteps + s) * Q + q) * 1024);
  }
  f32x4 bq[Q];
  const int cb = cg * 16 * Q + g * 4 * Q;                   // this lane's first output channel
#pragma unroll
  for (int q = 0; q < Q; ++q)
    bq[q] = (!PART && p.bias) ? *(const f32x4*)(p.bias + cb + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  f32x4 acc[NVT][Q];
  int t = 0;                                                // stage counter: chunk buffer t & 1
  for (int it = 0; it < niter; ++it) {
    // the brick after this one (its halo offsets replace the current ones once the current brick's last chunk is on its way)
    const int bn = b0 + (it + 1) * TEAMS + team;
    const bool nx_valid = it + 1 < niter && bn < b1;
    if (nx_valid) nx = decode(bn);
#pragma unroll
    for (int cp = 0; cp < CPW; ++cp) {
      // ---- prefetch the next stage of this wave into the other buffer (its previous reader, stage t - 1, is done)
      if (cp + 1 < CPW) {
        if (cu_valid) issue(cu, cp + 1, (t + 1) & 1);
      } else if (nx_valid) {
        offsets(nx);
        issue(nx, 0, (t + 1) & 1);
      }
      if (cp == 0) {
#pragma unroll
        for (int c = 0; c < NVT; ++c)
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // ---- MFMA sweep of chunk cp: 14 paired-tap steps, software-pipelined one step deep (see conv3d_k3_v2)
      if (cu_valid) {
        const char* buf = halo + (t & 1) * CHBUF;
        // B fragments through a small register ring over the flattened (step, column tile) sequence: item u + P is requested
        // before the Q MFMAs of item u (a fragment feeds Q MFMAs = 16 Q cycles of the matrix pipe; P items cover the LDS latency).
        // Whole-step double buffering (2 x NVT fragments = 64 VGPRs) does not fit next to 224 VGPRs of weights and 128 of accumulators.
        constexpr int P = Q >= 4 ? 3 : 6, R = P + 1, NU = kSteps * NVT;
        vec8 fb[R];
        auto load_item = [&](const int u) {
          const int s = u / NVT, c = u - s * NVT;
          const int kz = s < 9 ? s / 3 : (s < 12 ? s - 9 : (s == 12 ? 0 : 2));
          const int ky = s < 9 ? s % 3 : (s < 12 ? 0 : 2);
          const int kx = s < 9 ? 0 : 2;
          const int tapoff = ((kz * HY + ky) * HX + kx) * 16;
          const int bsel = s < 9 ? base_d1 : (s < 12 ? base_dx : (s == 12 ? base_dz : base_d0));
          const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
          const int coff = ((cz * HY + cy * LY) * HX + cx * LX) * 16;
          fb[u % R] = *(const vec8*)(buf + bsel + tapoff + coff);
        };
#pragma unroll
        for (int u = 0; u < P; ++u) load_item(u);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (u + P < NU) load_item(u + P);
          __builtin_amdgcn_sched_barrier(0);
          const int s = u / NVT, c = u - s * NVT;
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[c][q] = Ops<T>::mfma(wa[cp][s][q], fb[u % R], acc[c][q]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the next stage's halo (issued a whole sweep ago) and the previous brick's stores have landed; the reads of this stage
      // have returned (their MFMAs were issued), so the buffer may be refilled
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ++t;
    }

    // ---- reduction over the KW waves of the team, OWN / RT rounds: in a round wave o finalises its tiles o OWN + r RT + i
    const bool full = cu_valid && (cu.z0 + C::TZ <= p.D) & (cu.y0 + C::TY <= p.H) & (cu.x0 + C::TX <= p.W);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
      for (int o = 0; o < KW; ++o) {
        if (o != k && cu_valid) {
          const int slot = k < o ? k : k - 1;
          char* dst = scratch + ((o * (KW - 1) + slot) * RT * Q) * 1024 + lane * 16;
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int q = 0; q < Q; ++q) *(f32x4*)(dst + (i * Q + q) * 1024) = acc[o * OWN + r * RT + i][q];
        }
      }
      __syncthreads();
#pragma unroll
      for (int o = 0; o < KW; ++o) {
        if (o == k && cu_valid) {
          bool bad = false;
#pragma unroll
          for (int i = 0; i < RT; ++i) {
            f32x4 v[Q];
            // fixed order wave 0, 1, .., KW - 1; this wave's own partial sits in its registers
#pragma unroll
            for (int w2 = 0; w2 < KW; ++w2) {
              const int slot = w2 < o ? w2 : w2 - 1;
              const char* src = scratch + ((o * (KW - 1) + slot) * RT * Q) * 1024 + lane * 16;
#pragma unroll
              for (int q = 0; q < Q; ++q) {
                const f32x4 x = w2 == o ? acc[o * OWN + r * RT + i][q] : *(const f32x4*)(src + (i * Q + q) * 1024);
                v[q] = w2 == 0 ? x : v[q] + x;
              }
            }
            if (!PART) {
#pragma unroll
              for (int q = 0; q < Q; ++q) v[q] += bq[q];
              act_inplace<Q>(&v[0], p.act, p.slope);
            }
            // ---- store: lane (li, g) holds the 4 Q consecutive channels cb .. of voxel li of the tile
            const int c = o * OWN + r * RT + i;
            const int cx = c % XT, cy = (c / XT) % YT, cz = c / (XT * YT);
            const int zl = cu.z0 + cz, yl = cu.y0 + cy * LY + dy, xl = cu.x0 + cx * LX + dx;
            if (!full && !((zl < p.D) & (yl < p.H) & (xl < p.W))) continue;
            if (PART) {
              const long long vox = (((long long)cu.n * p.D + zl) * p.H + yl) * p.W + xl;
              float* d = p.part + ((long long)slice * p.N * p.D * p.H * p.W + vox) * p.Cout + cb;
#pragma unroll
              for (int q = 0; q < Q; ++q) *(f32x4*)(d + q * 4) = v[q];
            } else {
              const int ocs = p.ocs ? p.ocs : 32;            // the lane's 4 Q <= 16 channels sit inside one 16-channel chunk
              char* d = p.out + (long long)cu.n * p.on + (long long)zl * p.oz + (long long)yl * p.oy + (long long)xl * p.ox +
                        (long long)(cb >> 4) * ocs + (cb & 15) * 2;
              unsigned pk[2 * Q];
#pragma unroll
              for (int q = 0; q < Q; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(v[q][j]);
                pk[2 * q] = (unsigned)to_bits<T>(v[q][0]) | ((unsigned)to_bits<T>(v[q][1]) << 16);
                pk[2 * q + 1] = (unsigned)to_bits<T>(v[q][2]) | ((unsigned)to_bits<T>(v[q][3]) << 16);
              }
              if (Q == 1) {
                *(uint2*)d = make_uint2(pk[0], pk[1]);
              } else {
#pragma unroll
                for (int j = 0; j < Q / 2; ++j) *(uint4*)(d + j * 16) = make_uint4(pk[4 * j], pk[4 * j + 1], pk[4 * j + 2], pk[4 * j + 3]);
              }
            }
          }
          if (!PART && RangeCheck<T>::on) raise_flag(p.oflow, bad);
        }
      }
      __syncthreads();                                      // the scratch tiles may be overwritten (next round / next brick)
    }
    cu = nx;
    cu_valid = nx_valid;
  }
#undef AMX_DMA16
}

// ---- cross-workgroup split-K: out = act(bias + sum over the slices IN SLICE ORDER of the fp32 partial tensors [slice][voxel][Cout]).
// One thread = 8 channels of one voxel.
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int S, const float* __restrict__ bias, char* __restrict__ out,
                                     long long on, long long oz, long long oy, long long ox, int ocs, int N, int D, int H, int W,
                                     int Cout, int act, float slope, int* oflow) {
  const int c8n = Cout >> 3;
  const long long nvox = (long long)N * D * H * W, total = nvox * c8n;
  const float kact = act_k(act, slope);
  bool bad = false;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % c8n);
    const long long vox = idx / c8n;
    const float* src = part + vox * Cout + c8 * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int s = 0; s < S; ++s) {
      const float4 a = *(const float4*)(src + (long long)s * nvox * Cout), b = *(const float4*)(src + (long long)s * nvox * Cout + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = act_fwd(v[e] + (bias ? bias[c8 * 8 + e] : 0.f), kact);
      if (RangeCheck<T>::on) bad |= RangeCheck<T>::bad(v[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_bits<T>(v[2 * e]) | ((unsigned)to_bits<T>(v[2 * e + 1]) << 16);
    long long r = vox;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    r /= H;
    const int z = (int)(r % D);
    const int n = (int)(r / D);
    char* d = out + (long long)n * on + (long long)z * oz + (long long)y * oy + (long long)x * ox + (long long)(c8 >> 1) * (ocs ? ocs : 32) +
              (c8 & 1) * 16;
    *(uint4*)d = make_uint4(o[0], o[1], o[2], o[3]);
  }
  if (RangeCheck<T>::on) raise_flag(oflow, bad);
}

// -------------------------------------------------------------------------------------------
// launcher
// -------------------------------------------------------------------------------------------
static thread_local char g_kernel_name_ks[64] = "";
const char* last_conv_ks_kernel_name() { return g_kernel_name_ks; }
static int g_ks_cus = 0;

template <typename T, typename C, bool PART>
static hipError_t launch_ks_cfg(ConvParams p, hipStream_t st) {
  snprintf(g_kernel_name_ks, sizeof g_kernel_name_ks, "conv3d_k3_ks<%s,%dx%dx%d,q%d,k%dx%d,t%d%s>", __is_same(T, f16) ? "f16" : "bf16",
           C::TZ, C::TY, C::TX, C::Q, C::KW, C::CPW, C::TEAMS, PART ? ",part" : "");
  auto kern = conv3d_k3_ks_kernel<T, C, PART>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (g_ks_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    g_ks_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  p.nbz = (p.D + C::TZ - 1) / C::TZ;
  p.nby = (p.H + C::TY - 1) / C::TY;
  p.nbx = (p.W + C::TX - 1) / C::TX;
  const int nbricks = p.nbz * p.nby * p.nbx * p.N;
  const int S = p.kslices > 0 ? p.kslices : 1;
  const int combos = p.Cout / (16 * C::Q) * S;
  int wpc = g_ks_cus / combos;                              // workgroups per (cout group, slice): one workgroup per CU in total
  if (wpc < 1) wpc = 1;
  const int runs = (nbricks + C::TEAMS - 1) / C::TEAMS;      // a workgroup needs at least one brick per team to be useful
  if (wpc > runs) wpc = runs;
  hipLaunchKernelGGL(kern, dim3((unsigned)(combos * wpc)), dim3(C::NWAVE * 64), C::LDS_BYTES, st, p);
  return hipGetLastError();
}

// Shapes this kernel takes: single full-resolution segment, 16-bit channels-last (or row-planar) output, the plain 16-bit
// precisions; (Q, input chunks) among the register-stationary configurations below; 8 <= W.
//   chunks per slice = KW * CPW;  Q * CPW <= 4 (224 VGPRs of weights per wave)
struct KsPlan {
  int ok, kw, cpw, teams, slices;
};
static KsPlan ks_plan(const ConvParams& p, int precision, int Q, bool can_split) {
  KsPlan r = {0, 0, 0, 0, 1};
  if (precision > 1 || p.C1 != 0 || p.src0_f32c1 || p.out32 || p.out2 || p.stats || !p.out || p.W < 8 || p.H < 2 || p.D < 2) return r;
  if ((long long)p.D * p.s0z >= (1ll << 31)) return r;        // 32-bit halo offsets inside one sample
  const int nchunk = p.C0 >> 4;
  if (p.C0 & 15) return r;
  const int tz = p.W >= 16 ? 2 : 4, ty = 4, tx = p.W >= 16 ? 16 : 8;
  const long long nbricks = (long long)((p.D + tz - 1) / tz) * ((p.H + ty - 1) / ty) * ((p.W + tx - 1) / tx) * p.N;
  const int ncg = p.Cout / (16 * Q);
  if (Q == 4) {
    if (p.W < 16) return r;
    if (nchunk == 2) { r = {1, 2, 1, 2, 1}; return r; }
    if (nchunk == 4) { r = {1, 4, 1, 1, 1}; return r; }
    return r;
  }
  if (Q != 2) return r;
  // Q = 2: slices of 4 chunks (CPW = 1) or 8 chunks (CPW = 2).  Prefer the fewest slices that still give every CU a workgroup.
  if (nchunk == 4) { r = {1, 4, 1, 1, 1}; return r; }
  if (nchunk % 8 == 0) {
    const int s8 = nchunk / 8;                                // slices with CPW = 2
    if (s8 == 1 && ncg * nbricks >= 192) { r = {1, 4, 2, 1, 1}; return r; }
    if (!can_split) {
      if (s8 == 1) { r = {1, 4, 2, 1, 1}; return r; }
      return r;
    }
    if ((long long)ncg * s8 * nbricks >= 192 && s8 <= 8) { r = {1, 4, 2, 1, s8}; return r; }
    if (s8 * 2 <= 16) { r = {1, 4, 1, 1, s8 * 2}; return r; }
  }
  return r;
}

bool conv_ks_eligible(const ConvParams& p, int precision, int Q) {
  static int off = -1;
  if (off < 0) off = exp_env("AMX_NO_KS") ? 1 : 0;
  if (off) return false;
  return ks_plan(p, precision, Q, p.part != nullptr).ok != 0;
}

// bytes of fp32 partial tensors a layer of this shape needs when the caller offers split-K scratch (0: no split)
size_t conv_ks_part_bytes(int C0, int Cout, int N, int D, int H, int W, int precision, int Q) {
  ConvParams p;
  memset(&p, 0, sizeof p);
  p.C0 = C0; p.Cout = Cout; p.N = N; p.D = D; p.H = H; p.W = W; p.out = (char*)1;
  p.s0z = (long long)H * W * C0 * 2;
  const KsPlan pl = ks_plan(p, precision, Q, true);
  if (!pl.ok || pl.slices <= 1) return 0;
  return (size_t)pl.slices * N * D * H * W * Cout * sizeof(float);
}

template <typename T>
static hipError_t launch_ks_t(ConvParams p, const KsPlan& pl, int Q, hipStream_t st) {
  const bool wide = p.W >= 16;
  p.kslices = pl.slices;
  const bool part = pl.slices > 1;
  hipError_t e = hipErrorInvalidValue;
#define AMX_KS(TZ, TY, TX, QQ, CPW, KW, TEAMS)                                               \
  do {                                                                                       \
    typedef KsCfg<TZ, TY, TX, QQ, CPW, KW, TEAMS> CC;                                        \
    e = part ? launch_ks_cfg<T, CC, true>(p, st) : launch_ks_cfg<T, CC, false>(p, st);       \
  } while (0)
  if (Q == 4 && pl.kw == 2) {            // (Q = 4 is only ever picked for W >= 32: conv_pick_q)
    if (wide) AMX_KS(2, 4, 16, 4, 1, 2, 2);
  } else if (Q == 4) {
    if (wide) AMX_KS(2, 4, 16, 4, 1, 4, 1);
  } else if (pl.cpw == 2) {
    if (wide) AMX_KS(2, 4, 16, 2, 2, 4, 1); else AMX_KS(4, 4, 8, 2, 2, 4, 1);
  } else {
    if (wide) AMX_KS(2, 4, 16, 2, 1, 4, 1); else AMX_KS(4, 4, 8, 2, 1, 4, 1);
  }
#undef AMX_KS
  if (e != hipSuccess || !part) return e;
  const long long total = (long long)p.N * p.D * p.H * p.W * (p.Cout / 8);
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, st, p.part, pl.slices, p.bias, p.out, p.on, p.oz, p.oy, p.ox,
                     p.ocs, p.N, p.D, p.H, p.W, p.Cout, p.act, p.slope, p.oflow);
  return hipGetLastError();
}

hipError_t launch_conv_ks(const ConvParams& p, int precision, int Q, hipStream_t st) {
  const KsPlan pl = ks_plan(p, precision, Q, p.part != nullptr);
  if (!pl.ok) return hipErrorInvalidValue;
  return precision == 0 ? launch_ks_t<f16>(p, pl, Q, st) : launch_ks_t<bf16>(p, pl, Q, st);
}

}  // namespace amx
