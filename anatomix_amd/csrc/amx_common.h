// anatomix_amd -- internal shared declarations (device + host) for the gfx950 UNet path.
// Not part of the public C ABI (that is include/anatomix_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace amx {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// 3x3x3 taps are consumed two at a time (K = 32 = 2 taps x 16 channels per MFMA step).
// Step s pairs tap A (lane groups 0,1) with tap B (lane groups 2,3); the pairing is chosen so
// that B's LDS address is A's plus a per-step-class constant (1 voxel, one halo row, one halo
// slab), which lets every ds_read use a lane-constant base + immediate offset.
//   s 0..8  : (kz,ky,0) + (kz,ky,1)        kz = s/3, ky = s%3     delta = 1 voxel
//   s 9..11 : (kz,0,2)  + (kz,1,2)         kz = s-9               delta = halo row
//   s 12    : (0,2,2)   + (1,2,2)                                 delta = halo slab
//   s 13    : (2,2,2)   + <zero weights>                          delta = 0
constexpr int kSteps = 14;

__host__ __device__ constexpr int tapA_index(int s) {           // index into [kz][ky][kx]
  return s < 9 ? s * 3 : (s < 12 ? (s - 9) * 9 + 2 : (s == 12 ? 8 : 26));
}
__host__ __device__ constexpr int tapB_index(int s) {           // -1: no tap (zero weights)
  return s < 9 ? s * 3 + 1 : (s < 12 ? (s - 9) * 9 + 5 : (s == 12 ? 17 : -1));
}

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };

// "Set once" flags of launchers (hipFuncAttributeMaxDynamicSharedMemorySize, cached device properties) are per DEVICE: the attribute
// belongs to the device's copy of the function, so a process that drives a second GPU must set it there too (advisor, round 5).
// `mask`: one bit per device ordinal (devices >= 64 set the attribute on every launch).
struct DeviceOnce {
  unsigned long long mask = 0;
  int dev = 0;
  bool done() {
    if (hipGetDevice(&dev) != hipSuccess) dev = 64;
    return dev < 64 && ((mask >> dev) & 1);
  }
  void set() {
    if (dev < 64) mask |= 1ull << dev;
  }
};

// Experiment / ablation switches are environment variables ONLY in a -DAMX_EXPERIMENT build (make EXTRA=-DAMX_EXPERIMENT); the
// product library reads no environment on any path: every switch is a compile-time "unset".
#ifdef AMX_EXPERIMENT
inline const char* exp_env(const char* name) { return getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif

// Parameters of one 3x3x3 reflect-padded convolution launch.  All tensors are channels-last
// (N, D, H, W, C) 16-bit unless noted.  Strides are in BYTES.
struct ConvParams {
  const char* src0;            // first Cin segment (skip / sole input), full resolution
  const char* src1;            // second Cin segment, HALF resolution, read through nearest x2
  long long s0n, s0z, s0y, s0x;
  long long s1n, s1z, s1y, s1x;
  int C0, C1;                  // channels per segment (multiples of 16); Cin = C0 + C1
  int src0_f32c1;              // 1: src0 is a single-channel fp32 volume (network input)
  int up_shift;                // 1: src1 is HALF resolution and read through nearest x2 (coordinates >> 1);
                               // 0: src1 is a full-resolution tensor (materialised trilinear upsample)
  const char* wpk;             // packed A fragments [cout_group][chunk][step][q][lane][8]
  const float* bias;           // [Cout] fp32 (folded norm shift / conv bias)
  char* out;                   // 16-bit NDHWC output (OUTMODE 0)
  long long on, oz, oy, ox;
  float* out32;                // fp32 planar output (OUTMODE 1): [n][c][z][y][x], x contiguous
  long long pn, pc, pz, py;    // element strides of out32
  const float* wmap;           // optional importance map [D][H][W]: out32 += wmap * value
  int N, D, H, W, Cout;
  int act;
  float slope;
  int nbz, nby, nbx;           // bricks per axis
  float* stats;                // optional per-(n,c) {sum, sumsq} fp32 accumulators (instance norm)
  char* out2;                  // optional second output: 2x2x2 MAX-POOLED copy of `out` (16-bit NDHWC, dense), fused
  long long qn, qz, qy, qx;    //   into the epilogue of the z-marching kernel (nn.MaxPool3d(2), network.py:368)
  int* oflow;                  // optional device flag: set to 1 when a value that is about to be stored in f16 is out of the
                               //   f16 range (|v| > 65504) or NaN -- see amx_unet_numerics_status (include/anatomix_amd.h)
  int dbg;                     // ablation switches (env AMX_DBG; 0 in production): 1 no DMA after the first, 2 no MFMA sweep, 4 no stores
  const int* mxs;              // AMX_PREC_F16X2_MX: device word holding the E8M0 block-scale byte (x4) of this layer's fp8 weights
  int raw_halo;                // 1: src0 is the INTERIOR of a zero-framed buffer (frame >= 1 voxel): halo voxels are read from the frame at
                               //   coordinates -1 / n instead of being reflected (the data gradient's interior part, amx_train.hip)
  float* part;                 // conv3d_k3_ks: fp32 partial tensors [slice][voxel][Cout] of a cross-workgroup K split (scratch offered by the caller; null: no split)
  int kslices;                 //   ... and the number of K slices of this launch (set by the launcher)
  int cs0, cs1, ocs;           // byte stride between the 32-byte pieces (16-channel chunks) of one voxel in src0 / src1 / out; 0 = 32
                               //   (channels-last voxels).  Voxel layout FMT 2 keeps a ROW's pieces of one chunk together: W * 32.
};

// Activation layouts (C stored channels; amx_norm.hip load8 / store8):
//   FMT 0  channels-last voxels [v(C)]                one 16-bit value per channel                    2C bytes per voxel
//   FMT 1  channels-last voxels [hi(C) | lo(C)]       strict: value = hi + lo                         4C bytes
//   FMT 2  AMX_PREC_F16X2_MX: the f16 pair + e4m3 copies of hi and of 2^11 lo, 6C bytes per voxel, stored ROW-PLANAR: a row (n, z, y)
//          of W voxels is 3 C/16 planes of W x 32 bytes -- plane k: hi of channels 16k .. 16k+15; C/16 + k: their lo; 2 C/16 + k: the
//          copies [xl8(16) | xh8(16)] -- so that the 32 bytes per voxel a conv stage gathers are CONTIGUOUS along x.  (In channels-last
//          voxels of 128 .. 6144 bytes a stage used 32 bytes of every cache line it touched: LDS-DMA ran at 11-15 B/clk/CU instead
//          of 44-48, profiles/r03_dma_stride_ubench.txt, and bounded the generic kernel.)
//          byte(n, z, y, x, plane P, b) = ((n D + z) H + y) * 6 C W + P * 32 W + 32 x + b
//   FMT 3  single 16-bit values stored ROW-PLANAR like FMT 2 (2C bytes per voxel, C/16 planes of W x 32 bytes per row): the wide
//          (>= 64-channel) tensors of the f16 / bf16 forward, for the same reason -- byte(n, z, y, x, P, b) = ((n D + z) H + y) * 2 C W + P * 32 W + 32 x + b
__host__ __device__ constexpr int fmt_of_precision(int precision) { return precision < 2 ? 0 : (precision == 4 ? 2 : 1); }
__host__ __device__ constexpr int fmt_elem_bytes(int fmt) { return fmt == 0 ? 2 : (fmt == 1 ? 4 : 6); }

// z segments of a z-marching launch.  Its workgroups are resident `slots` at a time (one per CU for the ring kernels), so a grid of
// tiles * nseg workgroups runs in ceil(tiles * nseg / slots) ROUNDS of about (zseg + fill) plane-times each.  "Enough segments to reach
// `slots` workgroups" (the rule until round 4) is right when tiles divides slots -- the inference shapes -- and bad otherwise: the data
// gradient of a level-0 layer runs on the zero-framed 132^3 domain of two views = 170 tiles -> 2 segments = 340 workgroups = two
// rounds of 66 planes, the second a third full (the time of ONE segment of 132), where 3 segments run as two full rounds of 44.
// Picks the count that minimises rounds * (zseg + fill); `unit` = planes per march step (segments are multiples of it, >= 8 planes).
inline void pick_z_segments(int tiles, int D, int unit, int slots, int* zseg_out, int* nseg_out) {
  const int fill = 6;                                   // ring fill + first-step latency, in plane-times
  long long best = -1;
  int bz = D, bn = 1;
  static const bool old_rule = exp_env("AMX_OLD_ZSEG") != nullptr;     // experiment builds: the rule until round 4, for A/B
  if (old_rule) {
    int n = (slots + tiles - 1) / tiles;
    if (n < 1) n = 1;
    int zs = ((D + n - 1) / n + unit - 1) / unit * unit;
    if (zs < 8) zs = 8;
    *zseg_out = zs;
    *nseg_out = (D + zs - 1) / zs;
    return;
  }
  for (int n = 1; n <= 32; ++n) {
    int zs = (D + n - 1) / n;
    zs = (zs + unit - 1) / unit * unit;
    if (zs < 8) zs = 8;
    const int ns = (D + zs - 1) / zs;
    const long long rounds = ((long long)tiles * ns + slots - 1) / slots;
    const long long score = rounds * (zs + fill);
    if (best < 0 || score < best) {
      best = score;
      bz = zs;
      bn = ns;
    }
  }
  *zseg_out = bz;
  *nseg_out = bn;
}

// Merged-tap convolution over the upsampled segment of a concat layer (amx_conv3d_upmerge.hip).
struct UpmergeParams {
  const char* src;              // low-res tensor [N][LD][LH][LW][C1 (x2 when split)] 16-bit, byte strides below
  long long sn, sz, sy, sx;
  int C1;                       // up-channels, a multiple of 32
  int N, LD, LH, LW, Cout;
  const char* wpk;              // pack_upmerge: [cout group][stage][class][tap e][q][lane][8]
  const char* part;             // partial sums of the skip-channel conv (no bias, no activation): [N][2 LD][2 LH][2 LW][Cout] 16-bit
  char* out;                    // the layer's output, same layout: act(part + merged-tap sum + bias)
  const float* bias;            // [Cout] fp32 or null
  int act;
  float slope;
  int nbz, nby, nbx;
  int* oflow;
  int dbg;
  int cs, ocs;                  // bytes between a voxel's 16-channel chunks in src / in part and out (0 = 32: channels-last voxels;
                                //   row-planar tensors: row length x 32, see ConvParams::cs0)
};

// Weight-gradient launch (amx_wgrad.hip).
constexpr int MLP_MAXB = 8;              // heads / losses per batched launch (amx_mlp.hip, amx_supcon.hip)

struct WgradParams {
  const char* dy;                       // 16-bit [N][D][H][W][Cout] through byte strides (may be a framed view)
  long long yn, yz, yy, yx;
  const char* src0;                     // full-resolution input segment, C0 channels
  long long s0n, s0z, s0y, s0x;
  const char* src1;                     // second segment, C1 channels, read through >> up_shift
  long long s1n, s1z, s1y, s1x;
  int C0, C1, up_shift;
  int N, D, H, W, Cout;
  float* partial;                       // [nchunk][npairs][27][16][16]
  int nchunk, items_per_chunk, nitems, nyt;
  int nxt, ppc, cpx, nplanes;            // transpose-read kernel: x tiles, planes per chunk, (chunk, pair) entries per XCD, tiles * D
  int dbg;                               // experiment builds: 1 = no MFMA sweep, 2 = no DMA
  float* dw;                             // nchunk == 1: the workgroup writes dW[Cout][cin_real][27] itself (no partials, no reduce launch)
  int cin_real, accumulate;
};

}  // namespace amx
