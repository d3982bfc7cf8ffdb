// anatomix_amd -- C ABI (include/anatomix_amd.h) over the gfx950 kernels: layer plan, parameter
// folding/packing, activation arena and the launch schedule of one UNet forward.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/anatomix_amd.h"
#include "amx_common.h"

namespace amx {
hipError_t launch_conv(const ConvParams& p, int precision, int Q, hipStream_t st);
hipError_t launch_pack_weights(const float* w, const float* scale, void* wpk, int CinReal, int CinPad,
                               int Cout, int Q, int precision, hipStream_t st, int mode = 0, int CoutReal = 0, int CinStride = 0,
                               int C0Real = 0, int C0Phys = 0);
hipError_t launch_pack_weights_batch(int count, const float* const* w, void* const* wpk, const int* CinReal, const int* CinPad, const int* Cout,
                                     const int* Q, const int* mode, const int* CoutReal, int precision, hipStream_t st);
hipError_t launch_pack_weights_mx(const float* w, const float* scale, void* wpk, int* mxs, int CinReal, int CinPad, int Cout, int Q,
                                  hipStream_t st, int CoutReal = 0, int CinStride = 0, int C0Real = 0, int C0Phys = 0);
size_t conv_upmerge_packed_bytes(int C1, int Cout, int split);
bool conv_upmerge_eligible(int C0, int C1, int Cout, int D, int H, int W, int up_shift, int split);
hipError_t launch_conv_upmerge(const UpmergeParams& p, int precision, hipStream_t st);
hipError_t launch_pack_upmerge(const float* w, const float* scale, void* wpk, int c_off, int CinTotal, int C1, int Cout, int precision,
                               hipStream_t st);
const char* last_conv_upmerge_kernel_name();
hipError_t launch_fold_norm(const float* gamma, const float* beta, const float* mean, const float* var,
                            const float* conv_bias, float eps, int C, float* scale, float* shift,
                            hipStream_t st);
hipError_t launch_pool2(const void* in, void* out, int N, int Do, int Ho, int Wo, int C, int avg,
                        int precision, hipStream_t st, int skip_lo = 0, int planar = 0);
hipError_t launch_sw_normalize(float* acc, const float* cnt, int channels, long long voxels, hipStream_t st);
hipError_t launch_sw_count(float* cnt, int vd, int vh, int vw, int oz, int oy, int ox, int rd, int rh,
                           int rw, const float* wmap, hipStream_t st);
int conv_pick_q(int Cout, int W, int precision);
size_t conv_ks_part_bytes(int C0, int Cout, int N, int D, int H, int W, int precision, int Q);
hipError_t launch_conv_stem(const ConvParams& p, int precision, hipStream_t st);
hipError_t launch_pack_stem(const float* w, const float* scale, void* wpk, int Cout, int precision, hipStream_t st, int CoutReal = 0);
const char* last_conv_stem_kernel_name();
size_t conv_upcat16_packed_bytes();
bool conv_upcat16_eligible(const ConvParams& p);
hipError_t launch_conv_upcat16(const ConvParams& p, int precision, hipStream_t st);
hipError_t launch_pack_upcat16(const float* w, const float* scale, void* wpk, int precision, hipStream_t st);
const char* last_conv_upcat_kernel_name();
bool conv_zmarch_can_pool(const ConvParams& p);
bool conv_zmarch_can_pool_split(const ConvParams& p);
size_t instnorm_scratch_bytes(int N, int C, long long max_slots_x_C);
int conv_v2_stats_slots(int D, int H, int W, int Q);
bool conv_fuses_stats(const ConvParams& p, int precision, int Q);
int last_conv_stats_slots();
int conv_stem_stats_slots(const ConvParams& p, int precision);
bool in_apply_pool_eligible(int precision, int D, int H, int W, int C);
hipError_t launch_in_apply_pool(void* x, const float* ab, void* pooled, int N, int D, int H, int W, int C, int act, float slope, int avg,
                                int skip_lo, int pool_skip_lo, int* oflow, hipStream_t st);
hipError_t launch_instnorm(void* x, const float* gamma, const float* beta, float eps, int N, long long vox, int C, int act,
                           float slope, void* scratch, int precision, hipStream_t st, int* oflow = nullptr, int fused_slots = 0,
                           const float* kshift = nullptr, int W = 0, int skip_lo = 0, int apply = 1, float* ab_out = nullptr);
bool conv_zx_eligible(const ConvParams& p);
int conv_zx_stats_slots(int H, int W);
size_t conv_zx_packed_bytes();
hipError_t launch_pack_weights_zx(const float* w, const float* scale, void* wx, const int* mxs, int CoutReal, hipStream_t st);
hipError_t launch_conv_zx(ConvParams p, const float* in_ab, int in_act, float in_slope, const void* wx, hipStream_t st);
const char* last_conv_zx_kernel_name();
size_t attention_scratch_bytes(int b, int heads, int n);
void attention_operands(void* scratch, int b, int heads, int n, void** Qp, void** Kp, void** Vt, int* npad_out, int* nblk_pad_out);
hipError_t launch_attention_fwd(const void* Qp, const void* Kp, const void* Vt, int b, int n, int heads, int hd, float* out, hipStream_t st);
hipError_t launch_attention(const float* q, const float* k, const float* v, const float* qn_w, const float* qn_b, const float* kn_w,
                            const float* kn_b, float eps, const float* rope, int n_prefix, int b, int n, int heads, int hd,
                            float* out, void* scratch, hipStream_t st);
hipError_t launch_poison_if_flag(const int* flag, int* host_flag, float* y, long long count, hipStream_t st);
hipError_t launch_upsample2_trilinear(const void* in, void* out, int N, int D, int H, int W, int C, int precision,
                                      hipStream_t st, int skip_lo = 0, const float* ab = nullptr, int act = 0, float slope = 0.f,
                                      int* oflow = nullptr);
hipError_t launch_affine_act(void* x, const float* scale, const float* shift, int N, long long vox, int C, int act,
                             float slope, int precision, hipStream_t st, int* oflow = nullptr);
hipError_t launch_export_ncdhw(const void* src0, int C0, const void* src1, int C1, int up_shift, int N, int D, int H, int W,
                               float* out, int precision, hipStream_t st, int S0 = 0, int S1 = 0, int planar = 0);
const char* last_conv_kernel_name();
size_t train_scratch_bytes(int C);
hipError_t launch_bn_train_forward(const void* x, void* y, const float* gamma, const float* beta, float eps, long long rows, int C,
                                   int act, float slope, void* scratch, float* save_mean, float* save_rstd, float* running_mean,
                                   float* running_var, float momentum, int precision, hipStream_t st);
hipError_t launch_bn_act_backward(const void* dy, const void* y, const void* x, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, float* dgamma, float* dbeta, void* dx_framed, int N, int D,
                                  int H, int W, int C, int act, float slope, void* scratch, int precision, hipStream_t st);
hipError_t launch_adamw(const long long* table, int count, double lr, double b1, double b2, double eps, double wd, int maximize,
                        hipStream_t st, const double* d_hyper = nullptr);
hipError_t launch_pad_fold(const void* g_framed, void* din, int N, int D, int H, int W, int C, int accumulate, int precision,
                           hipStream_t st);
hipError_t launch_dgrad_fold_shell(const void* dy, long long yn, long long yz, long long yy, long long yx, int cdy, const float* w,
                                   int co_real, int ci_real, void* dx, int cdx, int N, int D, int H, int W, int precision, void* scratch,
                                   hipStream_t st);
size_t dgrad_shell_scratch_bytes();
bool conv_zmarch_eligible(const ConvParams& p);
bool conv_zmarch_stem_eligible(const ConvParams& p, int precision);
const char* last_conv_zm_kernel_name();
hipError_t launch_conv_zmarch_stem(const ConvParams& p, const float* x, long long xs_n, long long xs_z, long long xs_y, const long long* x_offs,
                                   const void* stem_wpk, const float* stem_bias, int stem_act, float stem_slope, int precision, hipStream_t st);
hipError_t launch_pool2_max_backward(const void* dp, const void* in, void* din, int N, int Do, int Ho, int Wo, int C,
                                     int accumulate, int precision, hipStream_t st);
size_t wgrad_scratch_bytes(int N, int D, int H, int W, int Cout, int CinPad);
hipError_t launch_wgrad(WgradParams p, int CinReal, float* dw, int accumulate, void* scratch, int precision, hipStream_t st);
size_t supcon_scratch_bytes(int N, int C);
hipError_t launch_supcon(const float* feat, const int* labels, int N, int C, float temperature, int rarity, int balance,
                         int sqrt_mode, float* loss, float* grad, void* scratch, hipStream_t st);
hipError_t launch_mlp_layer_forward(const float* x, int n, int k, const float* w, int m, const float* gamma, const float* beta,
                                    float eps, int act, float slope, float* z, float* y, float* mean, float* rstd, float* rmean,
                                    float* rvar, float momentum, hipStream_t st);
hipError_t launch_mlp_layer_backward(const float* dy, const float* y, const float* z, const float* mean, const float* rstd,
                                     const float* gamma, int act, float slope, const float* x, const float* w, int n, int k,
                                     int m, float* dz, float* dgamma, float* dbeta, float* dw, float* dx, float* wpart,
                                     hipStream_t st);
size_t mlp_backward_scratch_floats(int n, int cin, int width);
hipError_t launch_mlp_heads_layer_forward(int nb, const float* const* x, int n, const int* k, const float* const* w, int m,
                                          const float* const* gamma, const float* const* beta, float eps, int act, float slope,
                                          float* const* z, float* const* y, float* const* mean, float* const* rstd,
                                          float* const* rmean, float* const* rvar, float momentum, hipStream_t st);
hipError_t launch_mlp_heads_layer_backward(int nb, const float* const* dy, const float* const* y, const float* const* z,
                                           const float* const* mean, const float* const* rstd, const float* const* gamma, int act,
                                           float slope, const float* const* x, const float* const* w, int n, const int* k, int m,
                                           float* const* dz, float* const* dgamma, float* const* dbeta, float* const* dw,
                                           float* const* dx, float* const* wpart, hipStream_t st);
hipError_t launch_supcon_batch(int nb, const float* const* feat, const int* const* labels, int N, int C, float temperature, int rarity,
                               int balance, int sqrt_mode, float* const* loss, float* const* grad, void* scratch, hipStream_t st);
hipError_t launch_gather_labels_batch(const float* seg, int D, int H, int W, int nb, const long long* const* coords, int P, const int* dims,
                                      int views, int* const* out, hipStream_t st);
hipError_t launch_upcat_split(const void* dcat, void* dskip, void* dlow, int N, int Dl, int Hl, int Wl, int c0, int c1,
                              int acc_skip, int framed, int precision, hipStream_t st);
hipError_t launch_import_input(const float* src, void* dst, int N, int Cin, long long vox, int precision, hipStream_t st);
hipError_t launch_import_ncdhw(const float* src, void* dst, int N, int C, int D, int H, int W, long long dn, long long dz,
                               long long dy, long long dx, int accumulate, int precision, hipStream_t st);
hipError_t launch_upsample2_trilinear_backward(const void* gout, void* gin, int N, int D, int H, int W, int C, int precision,
                                               hipStream_t st);
hipError_t launch_sample_coords(const long long* draws, int n, int num, int d0, int d1, int d2, long long* coords, hipStream_t st);
hipError_t launch_sample_perm(const long long* keys, int nvox, int num, int d1, int d2, long long* coords, hipStream_t st);
hipError_t launch_gather_labels(const float* seg, int D, int H, int W, const long long* coords, int P, int d, int h, int w, int views,
                                int* out, hipStream_t st);
hipError_t launch_gather_rows(const void* src, int dtype, long long sn, long long sz, long long sy, long long sx, long long sc,
                              const long long* coords, int N, int P, int C, float* rows, hipStream_t st);
hipError_t launch_sampled_conv_backward(const float* g, const long long* coords, const void* x, int xc, const float* w, int N, int P, int D,
                                        int H, int W, int Cout, int Cin, float* dw, void* din, int dc, void* scratch, int precision,
                                        hipStream_t st);
size_t sampled_conv_backward_scratch_bytes(int P);
hipError_t launch_scatter_rows(const float* rows, const long long* coords, void* dst, int dtype, long long dn, long long dz, long long dy,
                               long long dx, int N, int P, int C, int accumulate, hipStream_t st);
size_t mindssc_scratch_bytes(int H, int W, int D);
hipError_t launch_mindssc(const float* img, int H, int W, int D, int radius, int dilation, float* out, void* scratch,
                          hipStream_t st);
hipError_t launch_pool_cat(const float* a, int ca, float sa, const float* b, int cb, float sb, int H, int W, int D, int g,
                           float* out, hipStream_t st);
hipError_t launch_box_filter(const float* in, float* out, int C, int H, int W, int D, int k, hipStream_t st);
size_t correlate_scratch_bytes(int h, int w, int d, int disp_hw);
hipError_t launch_correlate(const float* fix, const float* mov, int C, int h, int w, int d, int disp_hw, float* ssd,
                            long long* argmin, void* scratch, hipStream_t st);
}  // namespace amx

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define AMX_HIP(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return fail(AMX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

}  // namespace
namespace amx {
int set_error(int code, const char* msg) {      // for the other translation units of the C ABI (amx_vit.hip)
  g_err = msg;
  return code;
}
}  // namespace amx
namespace {

enum Kind { K_CONV, K_NORM, K_ACT, K_POOL, K_UP, K_FINAL_ACT };

struct ConvLayer {
  int module_idx = 0, cin = 0, cout = 0, norm_idx = -1;
  bool has_act = false, is_final = false;
  int level = 0;          // resolution level the conv runs at (0 = full)
  int q = 1;              // MFMA tiles per workgroup the weights are packed for
  int cin_pad = 0;        // STORED input channels: every segment padded to a multiple of 16 (ngf = 24: 24 -> 32)
  int cout_p = 0;         // stored output channels (cout rounded up to 16; the extra channels are exact zeros)
  int c0_real = 0, c0_p = 0;   // first conv of a decoder block: real / stored channels of the skip segment
  void* wpk = nullptr;    // packed A fragments
  void* wpk_up = nullptr; // second packing for the 16+32 -> 16 merged-tap kernel (amx_conv3d_upcat.hip)
  bool after_up = false;  // first conv of a decoder block: its input is cat(skip [cout channels], upsample(low [cin - cout]))
  void* wpk_skip = nullptr;   // wider concat layers, nearest upsample: 27-tap packing of the skip channels only ...
  void* wpk_merge = nullptr;  // ... and the merged-tap packing of the upsampled channels (amx_conv3d_upmerge.hip)
  float* in_gamma = nullptr;  // InstanceNorm3d(affine=True) weight / bias of the norm that follows (else null)
  float* in_beta = nullptr;
  float* scale = nullptr; // folded norm gain (applied to the weights at pack time)
  float* shift = nullptr; // epilogue bias
  // eval-BatchNorm layers only: UNFOLDED weights + the norm's own shift, for forwards that tap the pre-norm output
  void* wpk_raw = nullptr;
  int* mxs = nullptr;     // AMX_PREC_F16X2_MX: {E8M0 block-scale word of the fp8 weights, scratch for their maximum}
  void* wx = nullptr;     // AMX_PREC_F16X2_MX, 32 -> 32 layers: fp8 fragments of the normalise-on-load z-march kernel (amx_conv3d_zx.hip)
  bool raw_has_bias = false;  // conv bias under BatchNorm (the reference never builds that: use_bias == (norm=='instance'))
  bool loaded = false;
};

}  // namespace

struct amx_unet {
  amx_unet_cfg cfg;
  std::vector<int> kinds;
  std::vector<ConvLayer> convs;
  std::vector<int> encoder_idx, decoder_idx;
  std::vector<int> mod_c, mod_level;  // per module: channels / resolution level of `feat` after it (post-concat for Upsample)
  int pack_w = 0;  // spatial W the packing heuristic assumed (reference window: 128)
  // device flags raised by any epilogue that was about to store a value outside the f16 range (or NaN): a ring of kFlagSlots,
  // ONE PER FORWARD.  A forward clears its slot on its own stream before its first kernel and its last kernel mirrors the slot
  // into the (sticky) host flag -- so forwards of one handle that overlap on different streams (chunks in flight, pipelined window
  // batches) neither erase nor inherit each other's flag, and nothing is ever reset from another stream.
  static constexpr int kFlagSlots = 16;
  int* d_flags = nullptr;
  int* d_flag = nullptr;    // the slot of the forward being enqueued
  unsigned flag_next = 0;
  int* h_flag = nullptr;    // pinned host mirror, written by the last kernel of a forward when the device flag is up
  int* h_flag_dev = nullptr;   // the same memory as the device sees it
  hipEvent_t acc_done[2] = {nullptr, nullptr};   // amx_unet_forward_windows_pipelined: "slot s has finished accumulating"
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Mirrors the list the reference constructor builds (anatomix/model/network.py:309-465): same
// module order, hence the same integer indices in state_dict keys and encoder/decoder ids.
void build_plan(amx_unet* h) {
  const amx_unet_cfg& c = h->cfg;
  const bool has_norm = c.norm != AMX_NORM_NONE, has_act = c.activation != AMX_ACT_NONE;
  auto add_block = [&](int cin, int cout, int level) {
    ConvLayer L;
    L.module_idx = (int)h->kinds.size();
    L.cin = cin;
    L.cout = cout;
    L.level = level;
    h->kinds.push_back(K_CONV);
    if (has_norm) {
      L.norm_idx = (int)h->kinds.size();
      h->kinds.push_back(K_NORM);
    }
    if (has_act) {
      L.has_act = true;
      h->kinds.push_back(K_ACT);
    }
    h->convs.push_back(L);
  };
  add_block(c.input_nc, c.ngf, 0);
  int in_ngf = c.ngf;
  for (int i = 0; i < c.num_downs; ++i) {
    const int mult = i == 0 ? 1 : 2;
    add_block(in_ngf, in_ngf * mult, i);
    if (c.doubleconv) add_block(in_ngf * mult, in_ngf * mult, i);
    h->encoder_idx.push_back((int)h->kinds.size() - 1);
    h->kinds.push_back(K_POOL);
    in_ngf *= mult;
  }
  add_block(in_ngf, in_ngf * 2, c.num_downs);
  if (c.doubleconv) add_block(in_ngf * 2, in_ngf * 2, c.num_downs);
  int mult = 1 << c.num_downs;
  for (int i = 0; i < c.num_downs; ++i) {
    h->decoder_idx.push_back((int)h->kinds.size());
    h->kinds.push_back(K_UP);
    const int m = c.use_skip ? mult + mult / 2 : mult;
    const int level = c.num_downs - 1 - i;
    add_block(c.ngf * m, c.ngf * (mult / 2), level);
    h->convs.back().after_up = c.use_skip != 0;
    if (c.use_skip) h->convs.back().c0_real = c.ngf * (mult / 2);
    if (c.doubleconv) add_block(c.ngf * (mult / 2), c.ngf * (mult / 2), level);
    mult /= 2;
  }
  ConvLayer F;
  F.module_idx = (int)h->kinds.size();
  F.cin = c.ngf * mult;
  F.cout = c.output_nc;
  F.level = 0;
  F.is_final = true;
  h->kinds.push_back(K_CONV);
  h->convs.push_back(F);
  if (c.final_act != AMX_ACT_NONE) h->kinds.push_back(K_FINAL_ACT);
  // channels / level of `feat` after every module, as Unet.forward sees it (network.py:479-502)
  int ch = c.input_nc, lvl = 0;
  size_t ci = 0;
  std::vector<int> skip_c;
  for (size_t i = 0; i < h->kinds.size(); ++i) {
    switch (h->kinds[i]) {
      case K_CONV: ch = h->convs[ci].cout; lvl = h->convs[ci].level; ++ci; break;
      case K_POOL: lvl += 1; break;
      case K_UP:
        lvl -= 1;
        if (c.use_skip) { ch += skip_c.back(); skip_c.pop_back(); }
        break;
      default: break;
    }
    for (int e : h->encoder_idx)
      if (e == (int)i && c.use_skip) skip_c.push_back(ch);
    h->mod_c.push_back(ch);
    h->mod_level.push_back(lvl);
  }
}

// widest tensor materialised at a level: its own width, or (trilinear) the upsampled image of the level below it
int level_channels(const amx_unet* h, int level) {
  const int own = ((h->cfg.ngf + 15) / 16 * 16) << level;      // stored width: ngf padded to 16 channels
  int c = (h->cfg.interp == AMX_INTERP_TRILINEAR && level < h->cfg.num_downs) ? 2 * own : own;
  // the output conv's result is staged in a level-0 slot when it leaves through the export pass (W < 32 or output_nc > 32):
  // output_nc may exceed ngf.  (Sizing this by ngf alone overran the slot for output_nc = 64 -- silent while the bytes behind
  // the workspace were unused, wrong results / faults once the allocator had neighbours there.)
  if (level == 0 && (h->cfg.output_nc + 15) / 16 * 16 > c) c = (h->cfg.output_nc + 15) / 16 * 16;
  return c;
}

// set by amx_unet_forward_windows_pipelined around its run_forward call (per host thread)
static thread_local hipEvent_t g_acc_gate = nullptr, g_acc_done = nullptr;

struct Profiler {
  std::vector<hipEvent_t> ev;
  std::vector<amx_launch_record> rec;
  hipStream_t st;
  int mark(const amx_launch_record& r) {   // call BEFORE the launch it describes
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess || hipEventRecord(e, st) != hipSuccess) return -1;
    ev.push_back(e);
    rec.push_back(r);
    return 0;
  }
};

struct Arena {
  char* base;
  size_t bytes;
  std::vector<std::vector<char*>> slot;        // [level][3]
  std::vector<std::vector<bool>> used;
};

// strict precision (AMX_PREC_F16X2 / AMX_PREC_BF16X2): every stored voxel holds [hi(C) | lo(C)] 16-bit channels
// AMX_PREC_F16X2_MX: the pair plus 2 bytes of e4m3 copies per channel (amx_common.h, voxel layout FMT 2)
inline bool is_split(int precision) { return precision >= AMX_PREC_F16X2; }
inline bool is_mx(int precision) { return precision == AMX_PREC_F16X2_MX; }
inline long long elem_bytes(int precision) { return amx::fmt_elem_bytes(amx::fmt_of_precision(precision)); }
inline bool f16_stored(int precision) { return precision == AMX_PREC_F16 || precision == AMX_PREC_F16X2 || precision == AMX_PREC_F16X2_MX; }
// kernels without an fp8 stage of their own (the single-channel stem: its operand is the fp32 input) run their f16x2 variant
inline int stem_precision(int precision) { return is_mx(precision) ? AMX_PREC_F16X2 : precision; }

size_t level_bytes(const amx_unet* h, int level, int n, int d, int hh, int w) {
  const size_t vox = (size_t)(d >> level) * (hh >> level) * (w >> level);
  return align_up((size_t)n * vox * level_channels(h, level) * (size_t)elem_bytes(h->cfg.precision), 256);
}

// InstanceNorm scratch of a forward: the separate statistics pass needs 65536 entries per sample; a conv epilogue that writes the
// partial sums itself needs one slot per (brick, wave) of that layer
// (a, b) pairs of a norm whose apply pass is left to the consuming conv (f16x2mx, amx_conv3d_zx.hip): two buffers, used alternately,
// behind the statistics scratch -- the consumer writes ITS statistics into that scratch while it still reads its input's pairs
inline size_t kPendingAbBytes(int n) { return (size_t)n * 2048 * 2 * sizeof(float); }
size_t in_scratch_bytes(const amx_unet* h, int n, int d, int hh, int w) {
  long long worst = 0;
  if (h->cfg.norm == AMX_NORM_INSTANCE || h->cfg.norm == AMX_NORM_INSTANCE_AFFINE)
    for (const ConvLayer& L : h->convs) {
      if (L.norm_idx < 0) continue;
      const long long s = (long long)amx::conv_v2_stats_slots(d >> L.level, hh >> L.level, w >> L.level, L.q) * L.cout_p;
      worst = s > worst ? s : worst;
    }
  return align_up(amx::instnorm_scratch_bytes(n, h->cfg.ngf << h->cfg.num_downs, worst), 256) + 2 * kPendingAbBytes(n);
}

// fp32 partial tensors of the layers that split K across workgroups (conv3d_k3_ks, the deepest levels): the largest one
size_t ks_scratch_bytes(const amx_unet* h, int n, int d, int hh, int w) {
  size_t worst = 0;
  for (const ConvLayer& L : h->convs) {
    if (L.after_up || L.is_final || L.level == 0) continue;
    const size_t b = amx::conv_ks_part_bytes(L.cin_pad, L.cout_p, n, d >> L.level, hh >> L.level, w >> L.level, h->cfg.precision, L.q);
    worst = b > worst ? b : worst;
  }
  return align_up(worst, 256);
}

int check_shape(const amx_unet* h, int n, int d, int hh, int w) {
  const int L = h->cfg.num_downs;
  if (n < 1 || d < 1 || hh < 1 || w < 1) return fail(AMX_ERR_SHAPE, "non-positive shape");
  const int m = 1 << L;
  if (d % m || hh % m || w % m)
    return fail(AMX_ERR_SHAPE, "spatial dims (%d,%d,%d) must be divisible by 2^num_downs = %d", d, hh, w, m);
  if ((d >> L) < 2 || (hh >> L) < 2 || (w >> L) < 2)
    return fail(AMX_ERR_SHAPE, "bottleneck would be smaller than 2 voxels: reflect padding undefined");
  return AMX_OK;
}

// Feature taps of Unet.forward(input, layers, encode_only) (network.py:475-529): module ids in ascending order,
// one fp32 NCDHW device buffer per id; `stop` >= 0 ends the forward after that module (encode_only).
struct TapReq {
  const int* modules;
  int n;
  float* const* out;
  int stop;
};

// One forward.  in_*: fp32 single-channel input view (byte strides); out: fp32 planar output view.
int run_forward_impl(amx_unet* h, const float* x, long long xs_n, long long xs_z, long long xs_y,
                     float* y, long long ys_n, long long ys_c, long long ys_z, long long ys_y,
                     const float* wmap, int n, int d, int hh, int w, void* ws, size_t ws_bytes,
                     hipStream_t st, Profiler* prof, const long long* x_offs,
                     const long long* y_offs, const TapReq* taps) {
  // x_offs / y_offs (host arrays, element offsets, sliding-window mode): sample i reads its window at
  // x + x_offs[i] and accumulates into y + y_offs[i].  Only the stem and the output conv see the volume, so those
  // two run once per window (windows overlap: their accumulations must stay ordered on the stream); every layer
  // in between runs on the whole batch of windows.
  const amx_unet_cfg& c = h->cfg;
  const bool split = is_split(c.precision), mx = is_mx(c.precision);
  const long long eb = elem_bytes(c.precision);   // bytes per stored channel value (hi + lo halves in strict precision, + 2 of e4m3 copies)
  if (int e = check_shape(h, n, d, hh, w)) return e;
  for (const ConvLayer& L : h->convs)
    if (!L.loaded) return fail(AMX_ERR_NOT_LOADED, "conv model.%d has no parameters", L.module_idx);
  const int NL = c.num_downs + 1;
  size_t need = 0;
  for (int l = 0; l < NL; ++l) need += 3 * level_bytes(h, l, n, d, hh, w);
  void* in_scratch = (char*)ws + need;                 // instance-norm partial sums + (a, b) pairs
  need += in_scratch_bytes(h, n, d, hh, w);
  const size_t ks_bytes = ks_scratch_bytes(h, n, d, hh, w);
  float* ks_scratch = ks_bytes ? (float*)((char*)ws + need) : nullptr;
  need += ks_bytes;
  if (ws_bytes < need || ((uintptr_t)ws & 255))
    return fail(AMX_ERR_WORKSPACE, "workspace needs %zu bytes, 256-byte aligned (got %zu)", need, ws_bytes);

  Arena A;
  A.base = (char*)ws;
  A.slot.resize(NL);
  A.used.assign(NL, std::vector<bool>(3, false));
  {
    char* pcur = A.base;
    for (int l = 0; l < NL; ++l)
      for (int s = 0; s < 3; ++s) {
        A.slot[l].push_back(pcur);
        pcur += level_bytes(h, l, n, d, hh, w);
      }
  }
  auto grab = [&](int level) -> int {
    for (int s = 0; s < 3; ++s)
      if (!A.used[level][s]) {
        A.used[level][s] = true;
        return s;
      }
    return -1;
  };

  struct Tensor {                                   // C: stored channels per voxel, Cr: the reference's channel count
    int level = 0, slot = -1, C = 0, Cr = 0;
    bool planar = false;
    const float* ab = nullptr;                      // non-null: the tensor is RAW, its norm + activation pending: y = act(a x + b)
    int ab_act = AMX_ACT_NONE;                      //   ... with this activation
  };
  float* ab_buf[2] = {(float*)((char*)in_scratch + in_scratch_bytes(h, n, d, hh, w) - 2 * kPendingAbBytes(n)),
                      (float*)((char*)in_scratch + in_scratch_bytes(h, n, d, hh, w) - kPendingAbBytes(n))};
  int ab_next = 0;
  static int zx_env = -1;
  if (zx_env < 0) zx_env = amx::exp_env("AMX_NO_ZX") ? 0 : 1;
  // Row-planar storage (amx_common.h, layouts FMT 2 / 3) of the tensors the generic kernel gathers 32 bytes per voxel from: every
  // tensor of f16x2mx; in the single 16-bit precisions the WIDE ones (>= 64 channels: the 32^3 .. 8^3 levels of the 6 M network),
  // whose channels-last voxels of 128 .. 512 bytes left the LDS-DMA at 11-15 B/clk/CU (profiles/r03_dma_stride_ubench.txt).  Their
  // producers and consumers are the generic conv, the merged-tap kernel and the pool kernel; the passes that only know
  // channels-last voxels (norm apply, trilinear upsample, feature-tap export) keep the whole forward channels-last.
  // MEASURED for the single 16-bit precisions and left OFF (AMX_PLANAR16=1 enables it): the wide tensors of the 6 M forward sit at
  // 32^3 .. 8^3, where a stage's time is weight streaming and latency, not the halo gather -- 64 -> 64 @32^3 45.9 -> 44.0 us,
  // 128 -> 128 @16^3 27.6 -> 26.7, 8^3 unchanged; 2743 -> 2772 volumes/s (+1 %, inside the box-to-box noise).  f16x2mx, whose
  // 192-byte voxels sit at 128^3, gained 28 % per layer from the same layout and always uses it.
  static int planar_env = -1;
  if (planar_env < 0) planar_env = amx::exp_env("AMX_PLANAR16") ? 1 : 0;
  const bool planar16 = planar_env && !split && !taps && c.interp == AMX_INTERP_NEAREST &&
                        (c.norm == AMX_NORM_NONE || c.norm == AMX_NORM_BATCH_EVAL);
  Tensor cur;              // current activation (slot -1: the fp32 network input)
  bool have_cur_up = false;  // cur is to be read through a x2 upsample by the next conv
  std::vector<Tensor> skips;
  Tensor pend_skip;
  bool have_skip = false;
  Tensor fused_pool;         // pooled tensor written by the preceding conv's epilogue
  bool have_fused_pool = false;
  bool cur_is_full_up = false;   // cur is a materialised (trilinear) upsample at the consumer's resolution
  size_t conv_i = 0;
  auto tap_of = [&](int module) -> float* {
    if (taps)
      for (int t = 0; t < taps->n; ++t)
        if (taps->modules[t] == module) return taps->out[t];
    return nullptr;
  };
  const int stop = taps ? taps->stop : -1;
  // f16x2mx: a tensor whose only reader is the convolution at module `nxt` needs no lo plane (convolutions read hi and the e4m3 copies);
  // feature taps read the pair, so any tap request keeps every plane
  auto conv_only = [&](size_t nxt) -> int { return mx && !taps && nxt < h->kinds.size() && h->kinds[nxt] == K_CONV; };
  // tap = fp32 NCDHW copy of a stored 16-bit tensor
  auto export_slot = [&](const Tensor& t, float* dst) -> hipError_t {
    return amx::launch_export_ncdhw(A.slot[t.level][t.slot], t.Cr, nullptr, 0, 0, n, d >> t.level, hh >> t.level, w >> t.level,
                                    dst, c.precision, st, t.C, 0);
  };

  if (c.input_nc > 1) {
    if (x_offs || wmap) return fail(AMX_ERR_INVALID, "the fused sliding-window path needs input_nc == 1");
    cur.level = 0; cur.C = 16; cur.Cr = c.input_nc; cur.slot = grab(0);
    if (cur.slot < 0) return fail(AMX_ERR_INVALID, "internal: arena exhausted at level 0");
    AMX_HIP(amx::launch_import_input(x, A.slot[0][cur.slot], n, c.input_nc, (long long)d * hh * w, c.precision, st));
  }
  for (size_t i = 0; i < h->kinds.size(); ++i) {
    const int kind = h->kinds[i];
    if (kind == K_CONV) {
      const ConvLayer& L = h->convs[conv_i++];
      const int lv = L.level;
      const int dd = d >> lv, dh = hh >> lv, dw = w >> lv;
      // ---- stem + the 16 -> 16 layer behind it as ONE launch (amx_conv3d_zmarch.hip, STEM): the stem's output never reaches HBM.
      // Plain forward only: no taps, folded (or no) norm on both layers, nothing else reads the stem's tensor.
      if (cur.slot < 0 && !split && (!x_offs || n <= 16) && L.cout_p == 16 && L.cout == 16 && !L.is_final && conv_i < h->convs.size() &&
          (!L.has_act || c.activation == AMX_ACT_RELU || c.activation == AMX_ACT_NONE) &&
          !(L.norm_idx >= 0 && (c.norm == AMX_NORM_INSTANCE || c.norm == AMX_NORM_INSTANCE_AFFINE))) {
        const ConvLayer& Nx = h->convs[conv_i];
        const size_t g0 = i + 1 + (L.norm_idx >= 0 ? 1 : 0) + (L.has_act ? 1 : 0);       // module index of the next group
        const size_t g1 = g0 + 1 + (Nx.norm_idx >= 0 ? 1 : 0) + (Nx.has_act ? 1 : 0);    // ... and of the one after it
        bool ok = g0 < h->kinds.size() && h->kinds[g0] == K_CONV && Nx.level == 0 && !Nx.is_final && !Nx.after_up && Nx.cin_pad == 16 &&
                  Nx.cout_p == 16 && Nx.cout == 16 && Nx.q == 1 && Nx.loaded;
        for (int e : h->encoder_idx)
          if (e >= (int)i && e < (int)g0) ok = false;                                     // the stem's tensor would be a skip connection
        if (taps) {   // feature taps: only behind the pair (its activated output is module g1 - 1); the stem's tensor is never stored
          for (int t = 0; t < taps->n; ++t)
            if (taps->modules[t] + 1 < (int)g1) ok = false;
          if (taps->stop >= 0 && taps->stop + 1 < (int)g1) ok = false;
        }
        amx::ConvParams p;
        memset(&p, 0, sizeof p);
        p.N = n; p.D = dd; p.H = dh; p.W = dw; p.Cout = Nx.cout_p; p.C0 = 16; p.C1 = 0;
        p.wpk = (const char*)Nx.wpk; p.bias = Nx.shift; p.oflow = h->d_flag;
        p.act = Nx.has_act ? c.activation : AMX_ACT_NONE; p.slope = c.act_slope;
        p.ox = 16 * eb; p.oy = p.ox * dw; p.oz = p.oy * dh; p.on = p.oz * dd;
        if (ok && amx::conv_zmarch_stem_eligible(p, c.precision)) {
          Tensor out;
          out.level = 0; out.C = Nx.cout_p; out.Cr = Nx.cout; out.slot = grab(0);
          if (out.slot < 0) return fail(AMX_ERR_INVALID, "internal: arena exhausted at level 0");
          p.out = A.slot[0][out.slot];
          if (prof) {
            amx_launch_record r;
            memset(&r, 0, sizeof r);
            r.module_idx = L.module_idx; r.cin = L.cin; r.cout = Nx.cout; r.n = n; r.d = dd; r.h = dh; r.w = dw;
            const double vox = (double)n * dd * dh * dw;
            r.flops = 2.0 * 27.0 * (L.cin * L.cout + Nx.cin * Nx.cout) * vox;
            r.bytes = 4.0 * vox + 2.0 * Nx.cout * vox + 2.0 * 27.0 * (L.cin * L.cout + Nx.cin * Nx.cout);   // fp32 input once, 16-bit output once
            if (prof->mark(r)) return fail(AMX_ERR_HIP, "hipEventRecord failed");
          }
          AMX_HIP(amx::launch_conv_zmarch_stem(p, x, xs_n, xs_z, xs_y, x_offs, L.wpk, L.shift, L.has_act ? c.activation : AMX_ACT_NONE,
                                               c.act_slope, c.precision, st));
          if (prof) snprintf(prof->rec.back().kernel, sizeof prof->rec.back().kernel, "%s", amx::last_conv_zm_kernel_name());
          ++conv_i;
          cur = out;
          i = g1 - 1;
          if (float* t = tap_of((int)i)) AMX_HIP(export_slot(out, t));
          for (int e : h->encoder_idx)
            if (e == (int)i && c.use_skip) skips.push_back(cur);
          if (stop == (int)i) return AMX_OK;
          continue;
        }
      }
      amx::ConvParams p;
      memset(&p, 0, sizeof p);
      p.N = n; p.D = dd; p.H = dh; p.W = dw; p.Cout = L.cout_p;
      if (cur.slot < 0) {  // stem: fp32 single-channel input
        p.src0 = (const char*)x;
        p.s0n = xs_n; p.s0z = xs_z; p.s0y = xs_y; p.s0x = 4;
        p.C0 = 16; p.C1 = 0; p.src0_f32c1 = 1;
      } else if (have_cur_up) {
        const Tensor& lo = cur;
        // nearest: `lo` is the half-resolution tensor, read through >> 1; trilinear: already materialised at this level
        const int lw = cur_is_full_up ? dw : dw / 2, lh = cur_is_full_up ? dh : dh / 2, ld = cur_is_full_up ? dd : dd / 2;
        // row-planar layout of f16x2mx (amx_common.h FMT 2): a voxel's 32-byte pieces are 32 bytes apart along x and one row
        // plane (W * 32 bytes) apart per 16-channel chunk; rows, planes and samples keep their channels-last sizes
        const long long lx = (long long)lo.C * eb, ly = lx * lw, lz = ly * lh;
        p.up_shift = cur_is_full_up ? 0 : 1;
        if (have_skip) {
          const long long sx = (long long)pend_skip.C * eb, sy = sx * dw, sz = sy * dh;
          p.src0 = A.slot[pend_skip.level][pend_skip.slot];
          const bool pl0 = mx || pend_skip.planar, pl1 = mx || lo.planar;
          p.s0n = sz * dd; p.s0z = sz; p.s0y = sy; p.s0x = pl0 ? 32 : sx; p.C0 = pend_skip.C; p.cs0 = pl0 ? dw * 32 : 32;
          p.src1 = A.slot[lo.level][lo.slot];
          p.s1n = lz * ld; p.s1z = lz; p.s1y = ly; p.s1x = pl1 ? 32 : lx; p.C1 = lo.C; p.cs1 = pl1 ? lw * 32 : 32;
        } else {  // no skip connection: the whole input is the upsampled tensor
          p.src0 = A.slot[lo.level][lo.slot];  // unused segment of zero channels
          p.C0 = 0;
          p.src1 = A.slot[lo.level][lo.slot];
          const bool pl1 = mx || lo.planar;
          p.s1n = lz * ld; p.s1z = lz; p.s1y = ly; p.s1x = pl1 ? 32 : lx; p.C1 = lo.C; p.cs1 = pl1 ? lw * 32 : 32;
        }
      } else {
        const long long sx = (long long)cur.C * eb, sy = sx * dw, sz = sy * dh;
        p.src0 = A.slot[cur.level][cur.slot];
        const bool pl0 = mx || cur.planar;
        p.s0n = sz * dd; p.s0z = sz; p.s0y = sy; p.s0x = pl0 ? 32 : sx; p.C0 = cur.C; p.C1 = 0; p.cs0 = pl0 ? dw * 32 : 32;
      }
      if (p.C0 + p.C1 != L.cin_pad)
        return fail(AMX_ERR_INVALID, "internal: conv model.%d expects %d channels, schedule has %d",
                    L.module_idx, L.cin_pad, p.C0 + p.C1);
      p.wpk = (const char*)L.wpk;
      p.mxs = L.mxs;
      p.bias = L.shift;
      p.oflow = h->d_flag;
      const bool inorm = L.norm_idx >= 0 && (c.norm == AMX_NORM_INSTANCE || c.norm == AMX_NORM_INSTANCE_AFFINE);
      // InstanceNorm needs the whole (n, c) plane of RAW conv outputs first: the conv stores un-activated values
      // and amx::launch_instnorm normalises + activates them in place afterwards
      // encode_only whose last layer is this group's conv / norm id: the modules after it never run in the reference,
      // so the (in-place) activation must not touch the tapped tensor
      const int idx_act = L.has_act ? L.module_idx + 1 + (L.norm_idx >= 0 ? 1 : 0) : -1;
      const int stop_at = taps ? taps->stop : -1;
      const bool act_on = L.has_act && !(stop_at >= L.module_idx && stop_at < idx_act);
      p.act = (act_on && !inorm) ? c.activation : AMX_ACT_NONE;
      p.slope = c.act_slope;
      // ---- feature taps inside this conv -> norm -> act group.  The reference's activations are in-place modules
      // (network.py:188-196), so what the caller holds for a tap at the NORM id (or at the conv id when there is no
      // norm) is the activated tensor; only a conv followed by a norm yields a distinct, pre-norm tensor.
      float* tap_conv = tap_of(L.module_idx);
      float* tap_norm = L.norm_idx >= 0 ? tap_of(L.norm_idx) : nullptr;
      float* tap_act = idx_act >= 0 ? tap_of(idx_act) : nullptr;
      // pre-norm tap of a FOLDED eval-BatchNorm layer: run the conv with the unfolded weights, export, apply the norm
      const bool raw_bn = tap_conv && L.norm_idx >= 0 && !inorm && !L.is_final;
      if (raw_bn) {
        if (!L.wpk_raw) return fail(AMX_ERR_INVALID, "internal: model.%d has no unfolded packing", L.module_idx);
        p.wpk = (const char*)L.wpk_raw;
        if (L.raw_has_bias)
          return fail(AMX_ERR_INVALID, "pre-norm tap of model.%d: conv bias under BatchNorm is not supported", L.module_idx);
        p.bias = nullptr;   // s * conv + L.shift is the whole folded norm (applied by launch_affine_act below)
        p.act = AMX_ACT_NONE;
      }
      Tensor out;
      out.level = lv; out.C = L.cout_p; out.Cr = L.cout;
      // the fp32 planar epilogues need W >= 32 and <= 32 output channels; outside that the output conv stores 16-bit
      // channels-last like any other layer and one export pass produces the fp32 NCDHW tensor
      const bool final_via_export = L.is_final && (dw < 32 || L.cout_p > 32 || L.cout_p != L.cout);
      if (final_via_export && (wmap || x_offs))
        return fail(AMX_ERR_SHAPE, "sliding-window accumulation needs roi width >= 32 and output_nc <= 32");
      if (L.is_final && c.final_act != AMX_ACT_NONE && stop_at != L.module_idx) p.act = c.final_act;
      if (L.is_final && !final_via_export) {
        p.out32 = y;
        p.pn = ys_n; p.pc = ys_c; p.pz = ys_z; p.py = ys_y;
        p.wmap = wmap;
      } else {
        out.slot = grab(lv);
        if (out.slot < 0) return fail(AMX_ERR_INVALID, "internal: arena exhausted at level %d", lv);
        p.out = A.slot[lv][out.slot];
        p.ox = (long long)L.cout_p * eb; p.oy = p.ox * dw; p.oz = p.oy * dh; p.on = p.oz * dd;
        // >= 64 output channels never take the z-marching / merged 48 -> 16 kernels: the generic conv or the merged-tap pair writes them
        out.planar = planar16 && L.cout_p >= 64 && !L.is_final;
        if (mx || out.planar) { p.ox = 32; p.ocs = dw * 32; }
      }
      if (prof) {
        amx_launch_record r;
        memset(&r, 0, sizeof r);
        r.module_idx = L.module_idx; r.cin = L.cin; r.cout = L.cout; r.n = n; r.d = dd; r.h = dh; r.w = dw;
        const double vox = (double)n * dd * dh * dw;
        r.flops = 2.0 * 27.0 * L.cin * L.cout * vox;
        // ALGORITHMIC bytes: 16-bit activations read once and written once + the weights, the upsampled
        // segment counted at its LOW resolution for either interpolation (a materialised trilinear tensor and the hi / lo / e4m3
        // planes of the split precisions are this build's storage choices, not the layer's traffic requirement)
        const double in_b = cur.slot < 0 ? 4.0 * vox : (p.C0 * vox + p.C1 * vox / 8.0) * 2.0;
        const double out_b = L.is_final ? 4.0 * L.cout * vox : 2.0 * L.cout * vox;
        r.bytes = in_b + out_b + 2.0 * 27.0 * L.cin * L.cout;
        if (prof->mark(r)) return fail(AMX_ERR_HIP, "hipEventRecord failed");
      }
      // nn.MaxPool3d(2) right after this block (network.py:368): fuse it into the z-marching epilogue
      {
        size_t nxt = i + 1 + (L.norm_idx >= 0 ? 1 : 0) + (L.has_act ? 1 : 0);
        if (!L.is_final && !inorm && !raw_bn && nxt < h->kinds.size() && h->kinds[nxt] == K_POOL && c.pooling == AMX_POOL_MAX &&
            cur.slot >= 0 && !have_cur_up && (split ? amx::conv_zmarch_can_pool_split(p) : amx::conv_zmarch_can_pool(p)) &&
            L.q == L.cout_p / 16) {
          fused_pool.level = lv + 1; fused_pool.C = L.cout_p; fused_pool.Cr = L.cout; fused_pool.slot = grab(lv + 1);
          if (fused_pool.slot < 0) return fail(AMX_ERR_INVALID, "internal: arena exhausted at level %d", lv + 1);
          p.out2 = A.slot[lv + 1][fused_pool.slot];
          p.qx = (long long)L.cout_p * eb; p.qy = p.qx * (dw / 2); p.qz = p.qy * (dh / 2); p.qn = p.qz * (dd / 2);
          have_fused_pool = true;
        }
      }
      if (p.src0_f32c1 && (L.is_final || L.cout_p > 32))
        return fail(AMX_ERR_INVALID, "stem kernel supports ngf in {16, 32} and a following layer (ngf=%d)", L.cout);
      const bool use_upcat = !split && !p.src0_f32c1 && !raw_bn && L.wpk_up && have_cur_up && have_skip && amx::conv_upcat16_eligible(p);
      if (use_upcat) p.wpk = (const char*)L.wpk_up;
      // wider concat layers: the ordinary convolution over the skip channels first (raw partial sums into a free slot of this
      // level), then the merged-tap convolution over the upsampled channels, which adds them, the bias and the activation
      const bool use_merge = !use_upcat && !raw_bn && L.wpk_merge && have_cur_up && have_skip && !cur_is_full_up && !L.is_final &&
                             L.cout_p == L.cout && p.C0 == L.cout && amx::conv_upmerge_eligible(p.C0, p.C1, L.cout, dd, dh, dw, p.up_shift, split);
      int p_slot = -1;
      amx::UpmergeParams u;
      memset(&u, 0, sizeof u);
      if (use_merge) {
        p_slot = grab(lv);
        if (p_slot < 0) return fail(AMX_ERR_INVALID, "internal: arena exhausted at level %d", lv);
        u.src = p.src1; u.sn = p.s1n; u.sz = p.s1z; u.sy = p.s1y; u.sx = p.s1x; u.C1 = p.C1;
        u.N = n; u.LD = dd / 2; u.LH = dh / 2; u.LW = dw / 2; u.Cout = L.cout;
        u.wpk = (const char*)L.wpk_merge;
        u.part = A.slot[lv][p_slot]; u.out = p.out;
        u.bias = p.bias; u.act = p.act; u.slope = p.slope;
        u.oflow = h->d_flag;
        u.cs = p.cs1; u.ocs = p.ocs;                 // the low-resolution source / the partial sums and the output (same layout)
        p.out = A.slot[lv][p_slot];                  // same strides as the layer's output
        p.bias = nullptr; p.act = AMX_ACT_NONE;
        p.src1 = nullptr; p.C1 = 0; p.up_shift = 0;
        p.wpk = (const char*)L.wpk_skip;
      }
      // InstanceNorm layers on the generic kernel: the conv epilogue writes the partial sums of the statistics pass itself
      const bool zx_shape = mx && zx_env && L.wx && !have_cur_up && !x_offs && cur.slot >= 0 && (inorm || (L.is_final && !final_via_export)) &&
                            amx::conv_zx_eligible(p);
      // (the stem of the split precisions likewise: amx_conv3d_stem.hip)
      const int stem_slots = (p.src0_f32c1 && inorm && !x_offs) ? amx::conv_stem_stats_slots(p, stem_precision(c.precision)) : 0;
      const bool fuse_stats = inorm && !use_merge && !use_upcat && !L.is_final && !x_offs &&
                              (p.src0_f32c1 ? stem_slots > 0 : (zx_shape || amx::conv_fuses_stats(p, c.precision, L.q)));
      if (fuse_stats) p.stats = (float*)in_scratch;
      if (ks_scratch && !have_cur_up && !L.is_final && cur.slot >= 0 && !raw_bn && !use_merge &&
          amx::conv_ks_part_bytes(p.C0, p.Cout, n, dd, dh, dw, c.precision, L.q) <= ks_bytes)
        p.part = ks_scratch;
      // f16x2mx 32 -> 32 at whole tiles: the normalise-on-load z-march kernel (amx_conv3d_zx.hip).  It is the ONLY consumer of a
      // tensor whose norm was left pending (below), and takes already-normalised inputs too.
      const bool use_zx = zx_shape;
      if (cur.ab && !use_zx) return fail(AMX_ERR_INVALID, "internal: model.%d got an input whose norm is pending but cannot run the fused kernel", L.module_idx);
      auto launch_one = [&](const amx::ConvParams& q) -> hipError_t {
        if (q.src0_f32c1) return amx::launch_conv_stem(q, stem_precision(c.precision), st);
        if (use_upcat) return amx::launch_conv_upcat16(q, c.precision, st);
        if (use_zx) return amx::launch_conv_zx(q, cur.ab, cur.ab_act, c.act_slope, L.wx, st);
        return amx::launch_conv(q, c.precision, L.q, st);
      };
      // pipelined windows (two batches in flight on two streams): the accumulating launches of this batch wait for the other
      // slot's accumulations, so that overlapping windows still add up in window order
      if (x_offs && L.is_final && g_acc_gate) AMX_HIP(hipStreamWaitEvent(st, g_acc_gate, 0));
      if (x_offs && (p.src0_f32c1 || L.is_final)) {
        for (int wi = 0; wi < n; ++wi) {
          amx::ConvParams q = p;
          q.N = 1;
          if (p.src0_f32c1) {
            q.src0 = (const char*)(x + x_offs[wi]);
            q.out = p.out + (long long)wi * p.on;
            if (p.out2) q.out2 = p.out2 + (long long)wi * p.qn;
          } else {
            q.src0 = p.src0 + (long long)wi * p.s0n;
            if (p.src1) q.src1 = p.src1 + (long long)wi * p.s1n;
            q.out32 = y + y_offs[wi];
          }
          AMX_HIP(launch_one(q));
        }
      } else {
        AMX_HIP(launch_one(p));
      }
      if (x_offs && L.is_final && g_acc_done) AMX_HIP(hipEventRecord(g_acc_done, st));
      if (use_merge) {
        AMX_HIP(amx::launch_conv_upmerge(u, c.precision, st));
        A.used[lv][p_slot] = false;                  // the partial sums are dead once their consumer is enqueued (stream order)
      }
      if (prof) {
        if (use_merge)
          snprintf(prof->rec.back().kernel, sizeof prof->rec.back().kernel, "%.34s + %.26s", amx::last_conv_kernel_name(),
                   amx::last_conv_upmerge_kernel_name() + 7);
        else
          snprintf(prof->rec.back().kernel, sizeof prof->rec.back().kernel, "%s",
                   p.src0_f32c1 ? amx::last_conv_stem_kernel_name()
                                : use_upcat ? amx::last_conv_upcat_kernel_name()
                                            : (use_zx ? amx::last_conv_zx_kernel_name() : amx::last_conv_kernel_name()));
      }
      if (raw_bn) {
        AMX_HIP(export_slot(out, tap_conv));
        AMX_HIP(amx::launch_affine_act(A.slot[lv][out.slot], L.scale, L.shift, n, (long long)dd * dh * dw, L.cout_p,
                                       act_on ? c.activation : AMX_ACT_NONE, c.act_slope, c.precision, st, h->d_flag));
      } else if (tap_conv && !L.is_final && inorm) {
        AMX_HIP(export_slot(out, tap_conv));     // the stored raw convolution output, before the instance norm below
      }
      if (inorm) {
        if (L.is_final) return fail(AMX_ERR_INVALID, "internal: instance norm after the output conv");
        if (prof) {
          amx_launch_record r;
          memset(&r, 0, sizeof r);
          snprintf(r.kernel, sizeof r.kernel, "instnorm+act");
          r.module_idx = L.norm_idx; r.cin = r.cout = L.cout; r.n = n; r.d = dd; r.h = dh; r.w = dw;
          r.bytes = (double)eb * L.cout * (double)n * dd * dh * dw * 3.0;
          if (prof->mark(r)) return fail(AMX_ERR_HIP, "hipEventRecord failed");
        }
        // Leave the apply pass to the consumer when that is the fused 32 -> 32 kernel: the module after this conv -> norm -> act group is
        // a conv of that shape at this resolution (so this tensor is no skip connection, no pool / upsample input, no tap)
        bool defer = false;
        {
          const size_t nxt = i + 1 + (L.norm_idx >= 0 ? 1 : 0) + (L.has_act ? 1 : 0);
          if (mx && zx_env && !taps && !x_offs && act_on == L.has_act && nxt < h->kinds.size() && h->kinds[nxt] == K_CONV && conv_i < h->convs.size()) {
            const ConvLayer& Nx = h->convs[conv_i];
            amx::ConvParams t;
            memset(&t, 0, sizeof t);
            t.N = n; t.D = dd; t.H = dh; t.W = dw; t.C0 = L.cout_p; t.C1 = 0; t.Cout = Nx.cout_p; t.out = (char*)1; t.mxs = Nx.mxs; t.s0x = 32; t.ox = 32;
            // (the output conv takes a pending norm too: fp32 planar epilogue, no importance map, no activation of its own)
            const bool nx_kind = Nx.is_final ? (!wmap && c.final_act == AMX_ACT_NONE && dw >= 32 && Nx.cout_p == Nx.cout) : Nx.norm_idx >= 0;
            defer = Nx.wx && Nx.level == lv && nx_kind && Nx.cin_pad == L.cout_p && amx::conv_zx_eligible(t);
          }
        }
        // f16x2mx, pool right after this group: the apply pass also writes the pooled tensor (amx_norm.hip in_apply_pool_kernel); the
        // in-place tensor then has no reader left but the decoder's convolution, which takes hi and the copies
        const size_t nxt_mod = i + 1 + (L.norm_idx >= 0 ? 1 : 0) + (L.has_act ? 1 : 0);
        const bool apply_pool = !defer && mx && !taps && !x_offs && act_on == L.has_act && nxt_mod < h->kinds.size() && h->kinds[nxt_mod] == K_POOL &&
                                !have_fused_pool && amx::in_apply_pool_eligible(c.precision, dd, dh, dw, L.cout_p);
        // trilinear upsample right after this group (decoder): the upsample pass normalises its eight inputs on the way in
        static int up_env = -1;
        if (up_env < 0) up_env = amx::exp_env("AMX_NO_APPLY_UP") ? 0 : 1;
        const bool defer_up = up_env && !defer && !apply_pool && !taps && !x_offs && act_on == L.has_act && nxt_mod < h->kinds.size() &&
                              h->kinds[nxt_mod] == K_UP && c.interp == AMX_INTERP_TRILINEAR;
        if (defer_up) defer = true;
        const int slots = !fuse_stats ? 0 : p.src0_f32c1 ? stem_slots : use_zx ? amx::conv_zx_stats_slots(dh, dw) : amx::last_conv_stats_slots();
        float* abo = (defer || apply_pool) ? ab_buf[ab_next] : nullptr;
        AMX_HIP(amx::launch_instnorm(A.slot[lv][out.slot], L.in_gamma, L.in_beta, c.norm_eps, n, (long long)dd * dh * dw,
                                     L.cout_p, act_on ? c.activation : AMX_ACT_NONE, c.act_slope, in_scratch, c.precision, st, h->d_flag,
                                     slots, fuse_stats ? L.shift : nullptr, dw,
                                     conv_only(i + 1 + (L.norm_idx >= 0 ? 1 : 0) + (L.has_act ? 1 : 0)), (defer || apply_pool) ? 0 : 1, abo));
        if (apply_pool) {
          fused_pool.level = lv + 1; fused_pool.C = L.cout_p; fused_pool.Cr = L.cout; fused_pool.slot = grab(lv + 1);
          if (fused_pool.slot < 0) return fail(AMX_ERR_INVALID, "internal: arena exhausted at level %d", lv + 1);
          AMX_HIP(amx::launch_in_apply_pool(A.slot[lv][out.slot], abo, A.slot[lv + 1][fused_pool.slot], n, dd, dh, dw, L.cout_p,
                                            act_on ? c.activation : AMX_ACT_NONE, c.act_slope, c.pooling == AMX_POOL_AVG, 1,
                                            conv_only(nxt_mod + 1), h->d_flag, st));
          have_fused_pool = true;
          if (prof) snprintf(prof->rec.back().kernel, sizeof prof->rec.back().kernel, "instnorm+act+pool2<%s>", c.pooling == AMX_POOL_AVG ? "avg" : "max");
        }
        if (defer) {
          out.ab = abo;
          out.ab_act = act_on ? c.activation : AMX_ACT_NONE;
          ab_next ^= 1;
          if (prof) snprintf(prof->rec.back().kernel, sizeof prof->rec.back().kernel, "instnorm statistics only (apply fused into the next %s)", defer_up ? "upsample" : "conv");
        }
      }
      if (final_via_export)
        AMX_HIP(amx::launch_export_ncdhw(A.slot[lv][out.slot], L.cout, nullptr, 0, 0, n, dd, dh, dw, y, c.precision, st, L.cout_p, 0));
      if (L.is_final) {
        if (tap_conv)   // contiguous [n][Cout][d][h][w] output (taps are only offered by the plain forward)
          AMX_HIP(hipMemcpyAsync(tap_conv, y, (size_t)n * L.cout * dd * dh * dw * sizeof(float), hipMemcpyDeviceToDevice, st));
      } else {
        if (tap_conv && L.norm_idx < 0) AMX_HIP(export_slot(out, tap_conv));   // aliased by the in-place activation
        if (tap_norm) AMX_HIP(export_slot(out, tap_norm));
        if (tap_act) AMX_HIP(export_slot(out, tap_act));
      }
      // inputs are dead once their consumer is enqueued (stream order)
      if (cur.slot >= 0) A.used[cur.level][cur.slot] = false;
      if (have_skip) A.used[pend_skip.level][pend_skip.slot] = false;
      have_skip = false;
      have_cur_up = false;
      cur_is_full_up = false;
      cur = out;
      // skip to past the fused norm / activation modules
      if (L.norm_idx >= 0) ++i;
      if (L.has_act) ++i;
      // encoder_idx marks the module AFTER which the skip is pushed (network.py:546-547)
      for (int e : h->encoder_idx)
        if (e == (int)i && c.use_skip) {
          skips.push_back(cur);
          // keep it alive: mark as used by skip (cur release below must not free it)
        }
      if (stop >= L.module_idx && stop <= (int)i) return AMX_OK;   // encode_only: layers[-1] lies in this group
    } else if (kind == K_POOL && have_fused_pool) {
      // already produced by the previous conv's epilogue
      bool is_skip = false;
      for (const Tensor& s : skips)
        if (s.level == cur.level && s.slot == cur.slot) is_skip = true;
      if (!is_skip) A.used[cur.level][cur.slot] = false;
      cur = fused_pool;
      have_fused_pool = false;
      if (float* t = tap_of((int)i)) AMX_HIP(export_slot(cur, t));
      if (stop == (int)i) return AMX_OK;
    } else if (kind == K_POOL) {
      const int lv = cur.level + 1;
      Tensor out;
      out.level = lv; out.C = cur.C; out.Cr = cur.Cr; out.slot = grab(lv); out.planar = cur.planar;
      if (out.slot < 0) return fail(AMX_ERR_INVALID, "internal: arena exhausted at level %d", lv);
      if (prof) {
        amx_launch_record r;
        memset(&r, 0, sizeof r);
        snprintf(r.kernel, sizeof r.kernel, "pool2<%s>", c.pooling == AMX_POOL_AVG ? "avg" : "max");
        r.module_idx = (int)i; r.cin = r.cout = cur.C; r.n = n; r.d = d >> lv; r.h = hh >> lv; r.w = w >> lv;
        r.bytes = (double)eb * cur.C * (double)n * (d >> lv) * (hh >> lv) * (w >> lv) * 9.0;
        if (prof->mark(r)) return fail(AMX_ERR_HIP, "hipEventRecord failed");
      }
      AMX_HIP(amx::launch_pool2(A.slot[cur.level][cur.slot], A.slot[lv][out.slot], n, d >> lv, hh >> lv,
                                w >> lv, cur.C, c.pooling == AMX_POOL_AVG, c.precision, st, conv_only(i + 1), cur.planar));
      // the pooled-from tensor stays alive only if it was pushed as a skip
      bool is_skip = false;
      for (const Tensor& s : skips)
        if (s.level == cur.level && s.slot == cur.slot) is_skip = true;
      if (!is_skip) A.used[cur.level][cur.slot] = false;
      cur = out;
      if (float* t = tap_of((int)i)) AMX_HIP(export_slot(cur, t));
      if (stop == (int)i) return AMX_OK;
    } else if (kind == K_UP) {
      if (c.interp == AMX_INTERP_TRILINEAR) {
        // nn.Upsample(2,'trilinear') is materialised (16-bit NDHWC at the finer level); the conv that follows
        // then reads two full-resolution segments (up_shift = 0)
        const int lv = cur.level - 1;
        Tensor up;
        up.level = lv; up.C = cur.C; up.Cr = cur.Cr; up.slot = grab(lv);
        if (up.slot < 0) return fail(AMX_ERR_INVALID, "internal: arena exhausted at level %d", lv);
        if (prof) {
          amx_launch_record r;
          memset(&r, 0, sizeof r);
          snprintf(r.kernel, sizeof r.kernel, "upsample2<trilinear>");
          r.module_idx = (int)i; r.cin = r.cout = cur.C; r.n = n; r.d = d >> lv; r.h = hh >> lv; r.w = w >> lv;
          r.bytes = (double)eb * cur.C * (double)n * (d >> lv) * (hh >> lv) * (w >> lv) * 1.125;
          if (prof->mark(r)) return fail(AMX_ERR_HIP, "hipEventRecord failed");
        }
        AMX_HIP(amx::launch_upsample2_trilinear(A.slot[cur.level][cur.slot], A.slot[lv][up.slot], n, d >> cur.level,
                                                hh >> cur.level, w >> cur.level, cur.C, c.precision, st, conv_only(i + 1), cur.ab,
                                                cur.ab_act, c.act_slope, h->d_flag));
        A.used[cur.level][cur.slot] = false;
        cur = up;
        cur_is_full_up = true;
      }
      have_cur_up = true;
      if (c.use_skip) {
        pend_skip = skips.back();
        skips.pop_back();
        have_skip = true;
      }
      if (float* t = tap_of((int)i)) {   // taken after torch.cat((skip, up), 1) -- network.py:500-502
        const int lv = cur_is_full_up ? cur.level : cur.level - 1;
        AMX_HIP(amx::launch_export_ncdhw(have_skip ? A.slot[pend_skip.level][pend_skip.slot] : nullptr,
                                         have_skip ? pend_skip.Cr : 0, A.slot[cur.level][cur.slot], cur.Cr,
                                         cur_is_full_up ? 0 : 1, n, d >> lv, hh >> lv, w >> lv, t, c.precision, st,
                                         have_skip ? pend_skip.C : 0, cur.C));
      }
      if (stop == (int)i) return AMX_OK;
    } else if (kind == K_FINAL_ACT) {
      // fused into the last conv's epilogue; the caller's tensor at this id IS the network output
      if (float* t = tap_of((int)i))
        AMX_HIP(hipMemcpyAsync(t, y, (size_t)n * c.output_nc * d * hh * w * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
  }
  return AMX_OK;
}

// An f16 overflow (or a NaN) seen by an earlier forward of this handle is reported by the NEXT call; the forward that
// produced it has already had its output overwritten with NaN on the device (poison_if_flag).
int pending_numerics_error(amx_unet* h) {
  if (h->h_flag && *(volatile int*)h->h_flag) {
    *(volatile int*)h->h_flag = 0;      // the device slots are per forward and cleared by their own forwards
    return fail(AMX_ERR_OVERFLOW, "a previous forward of this network produced values outside the f16 range (or NaN) in %s storage; "
                "its output was overwritten with NaN.  Use precision bf16 or strict (bf16x2), which keep fp32's exponent range",
                h->cfg.precision == AMX_PREC_F16X2 ? "f16x2" : (is_mx(h->cfg.precision) ? "f16x2mx" : "f16"));
  }
  return AMX_OK;
}

int run_forward(amx_unet* h, const float* x, long long xs_n, long long xs_z, long long xs_y,
                float* y, long long ys_n, long long ys_c, long long ys_z, long long ys_y,
                const float* wmap, int n, int d, int hh, int w, void* ws, size_t ws_bytes,
                hipStream_t st, Profiler* prof = nullptr, const long long* x_offs = nullptr,
                const long long* y_offs = nullptr, const TapReq* taps = nullptr) {
  if (int e = pending_numerics_error(h)) return e;
  const bool f16_store = f16_stored(h->cfg.precision);
  if (h->d_flags) {
    h->d_flag = h->d_flags + (h->flag_next++ % amx_unet::kFlagSlots);
    if (f16_store) AMX_HIP(hipMemsetAsync(h->d_flag, 0, sizeof(int), st));
  }
  int rc = run_forward_impl(h, x, xs_n, xs_z, xs_y, y, ys_n, ys_c, ys_z, ys_y, wmap, n, d, hh, w, ws, ws_bytes, st, prof, x_offs,
                            y_offs, taps);
  const bool f16_storage = f16_stored(h->cfg.precision);
  if (rc == AMX_OK && f16_storage && h->d_flag) {
    // the output tensor (plain forward: dense [n][Cout][d][hh][w]; windows: the accumulation volume is the caller's, its
    // extent is not known here -- the flag and the status call cover that path) is poisoned when the flag is up
    const bool poison = !wmap && !x_offs && !(taps && taps->stop >= 0);
    AMX_HIP(amx::launch_poison_if_flag(h->d_flag, h->h_flag_dev, poison ? y : nullptr,
                                       poison ? (long long)n * h->cfg.output_nc * d * hh * w : 0, st));
    if (!h->h_flag_dev) AMX_HIP(hipMemcpyAsync(h->h_flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));   // unmapped host memory
  }
  return rc;
}

}  // namespace

extern "C" {

int amx_version(void) { return AMX_VERSION; }

namespace {
__global__ __launch_bounds__(256) void fill_lds_kernel(unsigned pattern, unsigned* sink) {
  extern __shared__ unsigned lds_words[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds_words[i] = pattern;
  __syncthreads();
  if (sink && lds_words[(threadIdx.x * 97) % (160 * 1024 / 4)] != pattern) *sink = 1;     // (keeps the stores alive)
}
}  // namespace

int amx_debug_fill_lds(unsigned pattern, void* stream) {
  static amx::DeviceOnce attr_once;
  if (!attr_once.done()) {
    AMX_HIP(hipFuncSetAttribute((const void*)fill_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.set();
  }
  hipLaunchKernelGGL(fill_lds_kernel, dim3(512), dim3(256), 160 * 1024, (hipStream_t)stream, pattern, (unsigned*)nullptr);
  AMX_HIP(hipGetLastError());
  return AMX_OK;
}

int amx_unet_numerics_status(amx_unet_t* h, int synchronize, void* stream) {
  if (!h) return fail(AMX_ERR_INVALID, "null handle");
  if (synchronize) AMX_HIP(hipStreamSynchronize((hipStream_t)stream));
  return pending_numerics_error(h);
}
const char* amx_last_error(void) { return g_err.c_str(); }

int amx_unet_create(amx_unet_t** out, const amx_unet_cfg* cfg) {
  if (!out || !cfg) return fail(AMX_ERR_INVALID, "null argument");
  *out = nullptr;
  // ngf = 8 mod 16 (the reference's default width 24, network.py:268): the ngf-wide tensors are stored with 16-channel padding
  if (cfg->num_downs < 1 || cfg->num_downs > 7 || cfg->ngf < 8 || cfg->ngf % 8 || (cfg->ngf + 15) / 16 * 16 > 32)
    return fail(AMX_ERR_INVALID, "ngf must be 8, 16, 24 or 32 (the stem kernel stores 16 or 32 channels) and 1 <= num_downs <= 7 (got ngf=%d "
                "num_downs=%d)", cfg->ngf, cfg->num_downs);
  // input_nc > 1: the input is imported into a 16-channel tensor and the first conv is an ordinary layer; output_nc that is not a
  // multiple of 16: the output conv stores padded channels and an export pass writes the fp32 NCDHW tensor
  if (cfg->input_nc < 1 || cfg->input_nc > 16) return fail(AMX_ERR_INVALID, "HIP path supports 1 <= input_nc <= 16 (got %d)", cfg->input_nc);
  if (cfg->output_nc < 1 || cfg->output_nc > 2048) return fail(AMX_ERR_INVALID, "output_nc out of range (got %d)", cfg->output_nc);
  if (cfg->norm < AMX_NORM_NONE || cfg->norm > AMX_NORM_INSTANCE_AFFINE)
    return fail(AMX_ERR_INVALID, "unknown norm mode %d", cfg->norm);
  if (cfg->interp != AMX_INTERP_NEAREST && cfg->interp != AMX_INTERP_TRILINEAR)
    return fail(AMX_ERR_INVALID, "unknown interp mode %d", cfg->interp);
  if ((cfg->ngf << cfg->num_downs) > 2048)
    return fail(AMX_ERR_INVALID, "widest layer has %d channels; the instance-norm kernels handle <= 2048", cfg->ngf << cfg->num_downs);
  if (cfg->activation < AMX_ACT_NONE || cfg->activation > AMX_ACT_LRELU || cfg->final_act < AMX_ACT_NONE ||
      cfg->final_act > AMX_ACT_LRELU)
    return fail(AMX_ERR_INVALID, "unsupported activation");
  if (cfg->precision < AMX_PREC_F16 || cfg->precision > AMX_PREC_F16X2_MX)
    return fail(AMX_ERR_INVALID, "unsupported precision %d", cfg->precision);
  // the fp8 correction stages exist in the generic kernel only; the consumers of a conv's output must be passes that write the e4m3
  // copies (norm apply, pool, upsample) -- i.e. networks that normalise with live statistics, the ones that need a strict mode at all
  if (is_mx(cfg->precision) && cfg->norm != AMX_NORM_INSTANCE && cfg->norm != AMX_NORM_INSTANCE_AFFINE)
    return fail(AMX_ERR_INVALID, "precision f16x2mx is implemented for the InstanceNorm configurations (norm='instance' / 'instance_affine'); "
                "use 'strict' (bf16x2) for this network");
  if (is_mx(cfg->precision) && cfg->input_nc != 1)
    return fail(AMX_ERR_INVALID, "precision f16x2mx needs input_nc == 1 (got %d)", cfg->input_nc);
  amx_unet* h = new amx_unet();
  h->cfg = *cfg;
  build_plan(h);
  h->pack_w = 128;
  for (ConvLayer& L : h->convs) {
    L.cout_p = (L.cout + 15) / 16 * 16;
    if (L.after_up && L.c0_real) {            // cat(skip, up): each segment padded on its own
      L.c0_p = (L.c0_real + 15) / 16 * 16;
      L.cin_pad = L.c0_p + (L.cin - L.c0_real + 15) / 16 * 16;
    } else {
      L.cin_pad = (L.cin + 15) / 16 * 16;
    }
    // Q is chosen for the reference operating point (128^3 windows): level l runs at W = 128>>l.
    const int w_at = h->pack_w >> L.level;
    L.q = amx::conv_pick_q(L.cout_p, w_at > 0 ? w_at : 1, cfg->precision);
    const size_t wbytes = (size_t)L.cout_p * L.cin_pad * 28 * 2 * (is_split(cfg->precision) ? 2 : 1);   // strict: [Wh | Wl]
    hipError_t e = hipMalloc(&L.wpk, wbytes);
    if (e == hipSuccess && !is_split(cfg->precision) && L.cin == 48 && L.cout == 16 && cfg->use_skip && cfg->interp == AMX_INTERP_NEAREST)
      e = hipMalloc(&L.wpk_up, amx::conv_upcat16_packed_bytes());
    // wider concat layers (nearest upsample): split into skip conv + merged-tap conv over the upsampled channels, at the levels
    // that are at least 32 voxels wide at the reference operating point
    if (e == hipSuccess && is_mx(cfg->precision)) e = hipMalloc((void**)&L.mxs, 2 * sizeof(int));
    if (e == hipSuccess && is_mx(cfg->precision) && L.cin_pad == 32 && L.cout_p == 32 && L.cin == 32 && L.cout == 32)
      e = hipMalloc(&L.wx, amx::conv_zx_packed_bytes());
    if (e == hipSuccess && !is_mx(cfg->precision) && L.after_up && cfg->interp == AMX_INTERP_NEAREST && L.wpk_up == nullptr && L.cout_p == L.cout &&
        amx::conv_upmerge_eligible(L.cout, L.cin - L.cout, L.cout, w_at, w_at, w_at, 1, is_split(cfg->precision))) {
      e = hipMalloc(&L.wpk_skip, (size_t)L.cout * L.cout * 28 * 2 * (is_split(cfg->precision) ? 2 : 1));
      if (e == hipSuccess) e = hipMalloc(&L.wpk_merge, amx::conv_upmerge_packed_bytes(L.cin - L.cout, L.cout, is_split(cfg->precision)));
    }
    if (e == hipSuccess && cfg->norm == AMX_NORM_BATCH_EVAL && L.norm_idx >= 0) e = hipMalloc(&L.wpk_raw, wbytes);
    if (e == hipSuccess) e = hipMalloc((void**)&L.scale, L.cout_p * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&L.shift, L.cout_p * sizeof(float));
    if (e == hipSuccess) e = hipMemset(L.scale, 0, L.cout_p * sizeof(float));      // padded channels: gain 0, shift 0 -> exact zeros
    if (e == hipSuccess) e = hipMemset(L.shift, 0, L.cout_p * sizeof(float));
    if (e == hipSuccess && cfg->norm == AMX_NORM_INSTANCE_AFFINE && L.norm_idx >= 0) {
      e = hipMalloc((void**)&L.in_gamma, L.cout_p * sizeof(float));
      if (e == hipSuccess) e = hipMalloc((void**)&L.in_beta, L.cout_p * sizeof(float));
      if (e == hipSuccess) e = hipMemset(L.in_gamma, 0, L.cout_p * sizeof(float));
      if (e == hipSuccess) e = hipMemset(L.in_beta, 0, L.cout_p * sizeof(float));
    }
    if (e != hipSuccess) {
      amx_unet_destroy(h);
      return fail(AMX_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
    }
  }
  hipError_t e = hipMalloc((void**)&h->d_flags, amx_unet::kFlagSlots * sizeof(int));
  if (e == hipSuccess) e = hipMemset(h->d_flags, 0, amx_unet::kFlagSlots * sizeof(int));
  h->d_flag = h->d_flags;
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_flag, sizeof(int), hipHostMallocDefault);
  if (e != hipSuccess) {
    amx_unet_destroy(h);
    return fail(AMX_ERR_HIP, "hipMalloc (status flag): %s", hipGetErrorString(e));
  }
  *h->h_flag = 0;
  if (hipHostGetDevicePointer((void**)&h->h_flag_dev, h->h_flag, 0) != hipSuccess) h->h_flag_dev = nullptr;
  *out = h;
  return AMX_OK;
}

void amx_unet_destroy(amx_unet_t* h) {
  if (!h) return;
  if (h->d_flags) (void)hipFree(h->d_flags);
  if (h->h_flag) (void)hipHostFree(h->h_flag);
  for (int i = 0; i < 2; ++i)
    if (h->acc_done[i]) (void)hipEventDestroy(h->acc_done[i]);
  for (ConvLayer& L : h->convs) {
    if (L.wpk) (void)hipFree(L.wpk);
    if (L.wpk_up) (void)hipFree(L.wpk_up);
    if (L.wpk_raw) (void)hipFree(L.wpk_raw);
    if (L.mxs) (void)hipFree(L.mxs);
    if (L.wx) (void)hipFree(L.wx);
    if (L.wpk_skip) (void)hipFree(L.wpk_skip);
    if (L.wpk_merge) (void)hipFree(L.wpk_merge);
    if (L.scale) (void)hipFree(L.scale);
    if (L.in_gamma) (void)hipFree(L.in_gamma);
    if (L.in_beta) (void)hipFree(L.in_beta);
    if (L.shift) (void)hipFree(L.shift);
  }
  delete h;
}

int amx_unet_num_modules(const amx_unet_t* h) { return h ? (int)h->kinds.size() : fail(AMX_ERR_INVALID, "null handle"); }
int amx_unet_num_convs(const amx_unet_t* h) { return h ? (int)h->convs.size() : fail(AMX_ERR_INVALID, "null handle"); }

int amx_unet_conv_info(const amx_unet_t* h, int conv, int* module_idx, int* cin, int* cout, int* norm_module_idx) {
  if (!h || conv < 0 || conv >= (int)h->convs.size()) return fail(AMX_ERR_INVALID, "bad conv index %d", conv);
  const ConvLayer& L = h->convs[conv];
  if (module_idx) *module_idx = L.module_idx;
  if (cin) *cin = L.cin;
  if (cout) *cout = L.cout;
  if (norm_module_idx) *norm_module_idx = L.norm_idx;
  return AMX_OK;
}

int amx_unet_load_conv(amx_unet_t* h, int module_idx, const float* d_weight, const float* d_bias,
                       const float* d_gamma, const float* d_beta, const float* d_mean, const float* d_var,
                       void* stream) {
  if (!h || !d_weight) return fail(AMX_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  for (ConvLayer& L : h->convs) {
    if (L.module_idx != module_idx) continue;
    const bool bn = L.norm_idx >= 0 && h->cfg.norm == AMX_NORM_BATCH_EVAL;
    if (bn && (!d_mean || !d_var)) return fail(AMX_ERR_INVALID, "model.%d: BatchNorm running stats required", module_idx);
    L.raw_has_bias = bn && d_bias != nullptr;
    AMX_HIP(amx::launch_fold_norm(bn ? d_gamma : nullptr, bn ? d_beta : nullptr, bn ? d_mean : nullptr,
                                  bn ? d_var : nullptr, d_bias, h->cfg.norm_eps, L.cout, L.scale, L.shift, st));
    if (L.in_gamma) {   // InstanceNorm3d(affine=True): keep its weight / bias for the normalisation pass
      if (!d_gamma || !d_beta) return fail(AMX_ERR_INVALID, "model.%d: instance_affine needs the norm weight and bias", module_idx);
      AMX_HIP(hipMemcpyAsync(L.in_gamma, d_gamma, L.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
      AMX_HIP(hipMemcpyAsync(L.in_beta, d_beta, L.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (L.cin == 1 && &L == &h->convs[0]) {   // stem: 27 taps packed into one K = 32 MFMA step
      AMX_HIP(amx::launch_pack_stem(d_weight, L.scale, L.wpk, L.cout_p, stem_precision(h->cfg.precision), st, L.cout));
      if (L.wpk_raw) AMX_HIP(amx::launch_pack_stem(d_weight, nullptr, L.wpk_raw, L.cout_p, stem_precision(h->cfg.precision), st, L.cout));
    } else if (is_mx(h->cfg.precision)) {
      AMX_HIP(amx::launch_pack_weights_mx(d_weight, L.scale, L.wpk, L.mxs, L.cin, L.cin_pad, L.cout_p, L.q, st, L.cout, 0, L.c0_real, L.c0_p));
      if (L.wx) AMX_HIP(amx::launch_pack_weights_zx(d_weight, L.scale, L.wx, L.mxs, L.cout, st));      // (after: it reads the layer's max |w|)
    } else {
      if (L.wpk_raw)
        AMX_HIP(amx::launch_pack_weights(d_weight, nullptr, L.wpk_raw, L.cin, L.cin_pad, L.cout_p, L.q, h->cfg.precision, st, 0, L.cout,
                                         0, L.c0_real, L.c0_p));
      AMX_HIP(amx::launch_pack_weights(d_weight, L.scale, L.wpk, L.cin, L.cin_pad, L.cout_p, L.q,
                                       h->cfg.precision, st, 0, L.cout, 0, L.c0_real, L.c0_p));
      if (L.wpk_up) AMX_HIP(amx::launch_pack_upcat16(d_weight, L.scale, L.wpk_up, h->cfg.precision, st));
      if (L.wpk_merge) {   // skip channels [0, cout) as an ordinary 27-tap packing, upsampled channels [cout, cin) merged
        AMX_HIP(amx::launch_pack_weights(d_weight, L.scale, L.wpk_skip, L.cout, L.cout, L.cout, L.q, h->cfg.precision, st, 0, 0, L.cin));
        AMX_HIP(amx::launch_pack_upmerge(d_weight, L.scale, L.wpk_merge, L.cout, L.cin, L.cin - L.cout, L.cout, h->cfg.precision, st));
      }
    }
    L.loaded = true;
    return AMX_OK;
  }
  return fail(AMX_ERR_INVALID, "model.%d is not a convolution of this network", module_idx);
}

size_t amx_unet_workspace_bytes(const amx_unet_t* h, int n, int d, int hh, int w) {
  if (!h) return 0;
  size_t need = 0;
  for (int l = 0; l <= h->cfg.num_downs; ++l) need += 3 * level_bytes(h, l, n, d, hh, w);
  need += in_scratch_bytes(h, n, d, hh, w);
  need += ks_scratch_bytes(h, n, d, hh, w);
  return need;
}

int amx_unet_forward(amx_unet_t* h, const float* d_x, float* d_y, int n, int d, int hh, int w,
                     void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!h || !d_x || !d_y || !d_workspace) return fail(AMX_ERR_INVALID, "null argument");
  const long long vox = (long long)d * hh * w;
  return run_forward(h, d_x, vox * 4, (long long)hh * w * 4, (long long)w * 4, d_y,
                     vox * h->cfg.output_nc, vox, (long long)hh * w, w, nullptr, n, d, hh, w, d_workspace,
                     workspace_bytes, (hipStream_t)stream);
}

int amx_unet_module_info(const amx_unet_t* h, int module_idx, int* channels, int* level) {
  if (!h || module_idx < 0 || module_idx >= (int)h->kinds.size()) return fail(AMX_ERR_INVALID, "bad module index %d", module_idx);
  if (channels) *channels = h->mod_c[module_idx];
  if (level) *level = h->mod_level[module_idx];
  return AMX_OK;
}

int amx_unet_forward_taps(amx_unet_t* h, const float* d_x, float* d_y, int n, int d, int hh, int w,
                          void* d_workspace, size_t workspace_bytes, const int* tap_modules, int n_taps,
                          float* const* d_tap_out, int stop_module, void* stream) {
  if (!h || !d_x || !d_y || !d_workspace || n_taps < 0 || (n_taps && (!tap_modules || !d_tap_out)))
    return fail(AMX_ERR_INVALID, "null argument");
  const int nmod = (int)h->kinds.size();
  for (int t = 0; t < n_taps; ++t) {
    if (tap_modules[t] < 0 || tap_modules[t] >= nmod || !d_tap_out[t] || (t && tap_modules[t] <= tap_modules[t - 1]))
      return fail(AMX_ERR_INVALID, "tap modules must be strictly ascending ids in [0,%d) with non-null buffers", nmod);
  }
  if (stop_module >= nmod) return fail(AMX_ERR_INVALID, "stop_module %d out of range", stop_module);
  TapReq req{tap_modules, n_taps, d_tap_out, stop_module < 0 ? -1 : stop_module};
  const long long vox = (long long)d * hh * w;
  return run_forward(h, d_x, vox * 4, (long long)hh * w * 4, (long long)w * 4, d_y, vox * h->cfg.output_nc, vox,
                     (long long)hh * w, w, nullptr, n, d, hh, w, d_workspace, workspace_bytes, (hipStream_t)stream,
                     nullptr, nullptr, nullptr, &req);
}

int amx_unet_forward_profiled(amx_unet_t* h, const float* d_x, float* d_y, int n, int d, int hh, int w,
                              void* d_workspace, size_t workspace_bytes, void* stream,
                              amx_launch_record* records, int max_records, int* n_records) {
  if (!h || !d_x || !d_y || !d_workspace || !records || !n_records) return fail(AMX_ERR_INVALID, "null argument");
  Profiler prof;
  prof.st = (hipStream_t)stream;
  const long long vox = (long long)d * hh * w;
  int rc = run_forward(h, d_x, vox * 4, (long long)hh * w * 4, (long long)w * 4, d_y, vox * h->cfg.output_nc, vox,
                       (long long)hh * w, w, nullptr, n, d, hh, w, d_workspace, workspace_bytes, prof.st, &prof);
  if (rc == AMX_OK) {
    amx_launch_record endr;
    memset(&endr, 0, sizeof endr);
    if (prof.mark(endr)) rc = fail(AMX_ERR_HIP, "hipEventRecord failed");
  }
  if (hipStreamSynchronize(prof.st) != hipSuccess && rc == AMX_OK) rc = fail(AMX_ERR_HIP, "hipStreamSynchronize failed");
  int cnt = 0;
  if (rc == AMX_OK) {
    for (size_t k = 0; k + 1 < prof.ev.size() && cnt < max_records; ++k) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, prof.ev[k], prof.ev[k + 1]);
      prof.rec[k].ms = ms;
      records[cnt++] = prof.rec[k];
    }
  }
  for (hipEvent_t e : prof.ev) (void)hipEventDestroy(e);
  *n_records = cnt;
  return rc;
}

int amx_unet_forward_window(amx_unet_t* h, const float* d_vol, int vd, int vh, int vw, int oz, int oy,
                            int ox, int rd, int rh, int rw, const float* d_wmap, float* d_acc,
                            void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!h || !d_vol || !d_acc || !d_wmap || !d_workspace) return fail(AMX_ERR_INVALID, "null argument");
  if (oz < 0 || oy < 0 || ox < 0 || oz + rd > vd || oy + rh > vh || ox + rw > vw)
    return fail(AMX_ERR_SHAPE, "window (%d,%d,%d)+(%d,%d,%d) outside volume (%d,%d,%d)", oz, oy, ox, rd, rh, rw, vd, vh, vw);
  const long long vvox = (long long)vd * vh * vw;
  const long long off = ((long long)oz * vh + oy) * vw + ox;
  return run_forward(h, d_vol + off, vvox * 4, (long long)vh * vw * 4, (long long)vw * 4, d_acc + off,
                     vvox * h->cfg.output_nc, vvox, (long long)vh * vw, vw, d_wmap, 1, rd, rh, rw,
                     d_workspace, workspace_bytes, (hipStream_t)stream);
}

int amx_unet_forward_windows(amx_unet_t* h, const float* d_vol, int vd, int vh, int vw, int n_windows,
                             const int* offsets_zyx, int rd, int rh, int rw, const float* d_wmap, float* d_acc,
                             void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!h || !d_vol || !d_acc || !d_wmap || !d_workspace || !offsets_zyx) return fail(AMX_ERR_INVALID, "null argument");
  if (n_windows < 1 || n_windows > 64) return fail(AMX_ERR_INVALID, "1 <= n_windows <= 64 (got %d)", n_windows);
  long long offs[64];
  for (int i = 0; i < n_windows; ++i) {
    const int oz = offsets_zyx[3 * i], oy = offsets_zyx[3 * i + 1], ox = offsets_zyx[3 * i + 2];
    if (oz < 0 || oy < 0 || ox < 0 || oz + rd > vd || oy + rh > vh || ox + rw > vw)
      return fail(AMX_ERR_SHAPE, "window (%d,%d,%d)+(%d,%d,%d) outside volume (%d,%d,%d)", oz, oy, ox, rd, rh, rw, vd, vh, vw);
    offs[i] = ((long long)oz * vh + oy) * vw + ox;
  }
  const long long vvox = (long long)vd * vh * vw;
  return run_forward(h, d_vol, vvox * 4, (long long)vh * vw * 4, (long long)vw * 4, d_acc, vvox * h->cfg.output_nc, vvox,
                     (long long)vh * vw, vw, d_wmap, n_windows, rd, rh, rw, d_workspace, workspace_bytes,
                     (hipStream_t)stream, nullptr, offs, offs);
}

int amx_unet_forward_windows_pipelined(amx_unet_t* h, const float* d_vol, int vd, int vh, int vw, int n_windows,
                                       const int* offsets_zyx, int rd, int rh, int rw, const float* d_wmap, float* d_acc,
                                       void* d_workspace, size_t workspace_bytes, int slot, void* stream) {
  if (!h) return fail(AMX_ERR_INVALID, "null handle");
  if (slot != 0 && slot != 1) return fail(AMX_ERR_INVALID, "slot must be 0 or 1 (got %d)", slot);
  for (int i = 0; i < 2; ++i)
    if (!h->acc_done[i]) AMX_HIP(hipEventCreateWithFlags(&h->acc_done[i], hipEventDisableTiming));
  g_acc_gate = h->acc_done[slot ^ 1];       // never recorded yet: the wait is a no-op
  g_acc_done = h->acc_done[slot];
  const int rc = amx_unet_forward_windows(h, d_vol, vd, vh, vw, n_windows, offsets_zyx, rd, rh, rw, d_wmap, d_acc, d_workspace,
                                          workspace_bytes, stream);
  g_acc_gate = g_acc_done = nullptr;
  return rc;
}

int amx_sw_normalize(float* d_acc, const float* d_cnt, int channels, long long voxels, void* stream) {
  if (!d_acc || !d_cnt || channels < 1 || voxels < 1) return fail(AMX_ERR_INVALID, "bad argument");
  AMX_HIP(amx::launch_sw_normalize(d_acc, d_cnt, channels, voxels, (hipStream_t)stream));
  return AMX_OK;
}

int amx_sw_count(float* d_cnt, int vd, int vh, int vw, int oz, int oy, int ox, int rd, int rh, int rw,
                 const float* d_wmap, void* stream) {
  if (!d_cnt || !d_wmap) return fail(AMX_ERR_INVALID, "null argument");
  if (oz < 0 || oy < 0 || ox < 0 || oz + rd > vd || oy + rh > vh || ox + rw > vw)
    return fail(AMX_ERR_SHAPE, "window outside volume");
  AMX_HIP(amx::launch_sw_count(d_cnt, vd, vh, vw, oz, oy, ox, rd, rh, rw, d_wmap, (hipStream_t)stream));
  return AMX_OK;
}

size_t amx_conv3d_packed_bytes(int cin, int cout) {
  const int cin_pad = (cin + 15) / 16 * 16;
  size_t bytes = (size_t)cout * cin_pad * 28 * 2;
  if (cin == 48 && cout == 16) bytes = (bytes + 255) / 256 * 256 + amx::conv_upcat16_packed_bytes();
  return 2 * bytes + 256;     // room for the [Wh | Wl] packing of the strict precisions (+ the block-scale words of f16x2mx)
}

static int conv3d_single(const void* d_x0, int c0, const void* d_x1, int c1, const float* d_weight, int weight_mode,
                         int cin_real, int cout_real, const float* d_scale, const float* d_shift, int cout, int n, int d,
                         int hh, int w, int act, float slope, int precision, void* d_wpk, void* d_out16, float* d_out32,
                         void* stream, void* d_scratch = nullptr, size_t scratch_bytes = 0) {
  if (!d_x0 || (!d_weight && !(weight_mode & AMX_WEIGHTS_PREPACKED)) || !d_wpk || (!d_out16 == !d_out32)) return fail(AMX_ERR_INVALID, "bad pointer arguments");
  if (c0 % 16 || c1 % 16 || c0 + c1 < 16 || cout % 16 || cout < 16)
    return fail(AMX_ERR_INVALID, "channel counts must be multiples of 16 (c0=%d c1=%d cout=%d)", c0, c1, cout);
  if (c1 && (!d_x1 || (d & 1) || (hh & 1) || (w & 1))) return fail(AMX_ERR_SHAPE, "upsampled segment needs even dims");
  if (d < 2 || hh < 2 || w < 2) return fail(AMX_ERR_SHAPE, "reflect padding needs >= 2 voxels per axis");
  hipStream_t st = (hipStream_t)stream;
  if (precision < AMX_PREC_F16 || precision > AMX_PREC_F16X2_MX) return fail(AMX_ERR_INVALID, "unsupported precision %d", precision);
  // weight_mode | AMX_WEIGHTS_PREPACKED: d_wpk already holds this layer's packing (amx_conv3d_pack_batch), nothing is packed here
  const bool prepacked = (weight_mode & AMX_WEIGHTS_PREPACKED) != 0;
  weight_mode &= ~AMX_WEIGHTS_PREPACKED;
  if (prepacked && (is_split(precision) || d_scale)) return fail(AMX_ERR_INVALID, "prepacked weights: plain 16-bit precisions, no scale");
  if (is_mx(precision) && weight_mode != 0) return fail(AMX_ERR_INVALID, "f16x2mx packs forward weights only");
  // strict precision: x0 / x1 / out16 voxels hold [hi(C) | lo(C)]; f16x2mx: [hi(C) | lo(C) | e4m3 copies (2C bytes)] -- the INPUT
  // voxels must carry valid copies (tests/_util.py to_ndhwc_mx), the output's copy section is left untouched
  const long long eb = elem_bytes(precision);
  const int q = amx::conv_pick_q(cout, w, precision);
  if (d_out32 && (q > 2 || w < 32)) return fail(AMX_ERR_INVALID, "fp32 planar output needs cout <= 32 and w >= 32");
  if (cin_real < 1 || cin_real > c0 + c1 || cout_real < 1 || cout_real > cout || (weight_mode != 0 && weight_mode != 1))
    return fail(AMX_ERR_INVALID, "bad weight description (mode %d, cin_real %d, cout_real %d)", weight_mode, cin_real, cout_real);
  amx::ConvParams p;
  memset(&p, 0, sizeof p);
  if (is_mx(precision)) {
    int* mxs = (int*)((char*)d_wpk + align_up((size_t)cout * (c0 + c1) * 28 * 2 * 2, 256));
    AMX_HIP(amx::launch_pack_weights_mx(d_weight, d_scale, d_wpk, mxs, cin_real, c0 + c1, cout, q, st, cout_real));
    p.mxs = mxs;
  } else if (!prepacked) {
    AMX_HIP(amx::launch_pack_weights(d_weight, d_scale, d_wpk, cin_real, c0 + c1, cout, q, precision, st, weight_mode, cout_real));
  }
  p.N = n; p.D = d; p.H = hh; p.W = w; p.Cout = cout;
  p.src0 = (const char*)d_x0; p.C0 = c0;
  p.s0x = (long long)c0 * eb; p.s0y = p.s0x * w; p.s0z = p.s0y * hh; p.s0n = p.s0z * d;
  if (c1) {
    p.src1 = (const char*)d_x1; p.C1 = c1; p.up_shift = 1;
    p.s1x = (long long)c1 * eb; p.s1y = p.s1x * (w / 2); p.s1z = p.s1y * (hh / 2); p.s1n = p.s1z * (d / 2);
  }
  if (is_mx(precision)) {      // row-planar tensors (amx_common.h FMT 2)
    p.s0x = 32; p.cs0 = w * 32;
    p.s1x = 32; p.cs1 = (w / 2) * 32;
  }
  p.wpk = (const char*)d_wpk;
  p.bias = d_shift;
  p.act = act; p.slope = slope;
  if (d_out16) {
    p.out = (char*)d_out16;
    p.ox = (long long)cout * eb; p.oy = p.ox * w; p.oz = p.oy * hh; p.on = p.oz * d;
    if (is_mx(precision)) { p.ox = 32; p.ocs = w * 32; }
  } else {
    p.out32 = d_out32;
    p.py = w; p.pz = (long long)hh * w; p.pc = p.pz * d; p.pn = p.pc * cout;
  }
  if (prepacked && !is_split(precision) && weight_mode == 0 && cin_real == 48 && c0 == 16 && c1 == 32 && cout == 16 && amx::conv_upcat16_eligible(p))
    return fail(AMX_ERR_INVALID, "prepacked weights: the 16 + up32 -> 16 merged-tap layer packs its own format (call without the flag)");
  if (!is_split(precision) && weight_mode == 0 && cin_real == 48 && c0 == 16 && c1 == 32 && cout == 16 && amx::conv_upcat16_eligible(p)) {
    char* up = (char*)d_wpk + ((size_t)cout * (c0 + c1) * 28 * 2 + 255) / 256 * 256;
    AMX_HIP(amx::launch_pack_upcat16(d_weight, d_scale, up, precision, st));
    p.wpk = up;
    AMX_HIP(amx::launch_conv_upcat16(p, precision, st));
    return AMX_OK;
  }
  // split-K scratch offered by the caller (amx_conv3d_k3_reflect_ws): deep layers with few voxels split K across workgroups
  if (d_scratch && !c1 && d_out16) {
    const size_t need = amx::conv_ks_part_bytes(c0, cout, n, d, hh, w, precision, q);
    if (need && (scratch_bytes < need || ((uintptr_t)d_scratch & 15)))
      return fail(AMX_ERR_WORKSPACE, "split-K scratch needs %zu bytes, 16-byte aligned (got %zu)", need, scratch_bytes);
    if (need) p.part = (float*)d_scratch;
  }
  AMX_HIP(amx::launch_conv(p, precision, q, st));
  return AMX_OK;
}

// ---- data gradient = interior launch of the forward kernel on the zero-framed gradient + shell terms (amx_train.hip)
static bool dgrad_interior_shape_ok(int c_dy, int cout, int d, int hh, int w, int precision) {
  if (precision != AMX_PREC_F16 && precision != AMX_PREC_BF16) return false;
  amx::ConvParams p;
  memset(&p, 0, sizeof p);
  p.C0 = c_dy; p.Cout = cout; p.D = d; p.H = hh; p.W = w; p.out = (char*)1;
  return amx::conv_zmarch_eligible(p) && amx::conv_pick_q(cout, w, precision) == cout / 16 && (size_t)27 * c_dy * cout * 4 <= 150 * 1024;
}

int amx_conv3d_dgrad_interior_supported(int c_dy, int cout, int d, int hh, int w, int precision) {
  return dgrad_interior_shape_ok(c_dy, cout, d, hh, w, precision) ? 1 : 0;
}

int amx_conv3d_dgrad_interior(const void* d_dy_framed, int c_dy, const float* d_weight, int weight_flags, int cin_real, int cout_real, int cout,
                              int n, int d, int hh, int w, int precision, void* d_wpk, void* d_out16, void* stream) {
  if (!d_dy_framed || !d_wpk || !d_out16 || (!d_weight && !(weight_flags & AMX_WEIGHTS_PREPACKED)))
    return fail(AMX_ERR_INVALID, "dgrad_interior: bad pointer arguments");
  if (!dgrad_interior_shape_ok(c_dy, cout, d, hh, w, precision))
    return fail(AMX_ERR_INVALID, "dgrad_interior: shape outside the z-march kernels (ask amx_conv3d_dgrad_interior_supported)");
  if (cin_real < 1 || cin_real > c_dy || cout_real < 1 || cout_real > cout) return fail(AMX_ERR_INVALID, "dgrad_interior: bad weight description");
  hipStream_t st = (hipStream_t)stream;
  const int q = cout / 16;
  if (!(weight_flags & AMX_WEIGHTS_PREPACKED))
    AMX_HIP(amx::launch_pack_weights(d_weight, nullptr, d_wpk, cin_real, c_dy, cout, q, precision, st, 1, cout_real));
  amx::ConvParams p;
  memset(&p, 0, sizeof p);
  p.N = n; p.D = d; p.H = hh; p.W = w; p.Cout = cout; p.C0 = c_dy;
  p.s0x = (long long)c_dy * 2; p.s0y = p.s0x * (w + 4); p.s0z = p.s0y * (hh + 4); p.s0n = p.s0z * (d + 4);
  p.src0 = (const char*)d_dy_framed + 2 * (p.s0z + p.s0y + p.s0x);       // the interior of [n][d+4][hh+4][w+4][c_dy]
  p.raw_halo = 1;
  p.wpk = (const char*)d_wpk;
  p.act = AMX_ACT_NONE;
  p.out = (char*)d_out16;
  p.ox = (long long)cout * 2; p.oy = p.ox * w; p.oz = p.oy * hh; p.on = p.oz * d;
  AMX_HIP(amx::launch_conv(p, precision, q, st));
  return AMX_OK;
}

size_t amx_conv3d_dgrad_shell_scratch_bytes(void) { return amx::dgrad_shell_scratch_bytes(); }

int amx_conv3d_dgrad_fold_shell(const void* d_dy_framed, int c_dy, const float* d_weight, int co_real, int ci_real, void* d_dx, int c_dx, int n,
                                int d, int hh, int w, int precision, void* d_scratch, void* stream) {
  if (!d_dy_framed || !d_weight || !d_dx || !d_scratch || ((uintptr_t)d_scratch & 15) || co_real < 1 || co_real > c_dy || ci_real < 1 || ci_real > c_dx || d < 2 || hh < 2 || w < 2)
    return fail(AMX_ERR_INVALID, "dgrad_fold_shell: bad arguments");
  const long long fx = (long long)c_dy * 2, fy = fx * (w + 4), fz = fy * (hh + 4), fn = fz * (d + 4);
  AMX_HIP(amx::launch_dgrad_fold_shell((const char*)d_dy_framed + 2 * (fz + fy + fx), fn, fz, fy, fx, c_dy, d_weight, co_real, ci_real, d_dx, c_dx,
                                       n, d, hh, w, precision, d_scratch, (hipStream_t)stream));
  return AMX_OK;
}

int amx_conv3d_pack_batch(const amx_pack_req* reqs, int count, int precision, void* stream) {
  if (!reqs || count < 1 || count > 256 || (precision != AMX_PREC_F16 && precision != AMX_PREC_BF16))
    return fail(AMX_ERR_INVALID, "pack_batch: bad arguments");
  std::vector<const float*> w(count);
  std::vector<void*> wpk(count);
  std::vector<int> cr(count), cp(count), co(count), q(count), md(count), cor(count);
  for (int i = 0; i < count; ++i) {
    const amx_pack_req& r = reqs[i];
    if (!r.d_weight || !r.d_wpk || r.cin_pad % 16 || r.cin_pad < 16 || r.cout % 16 || r.cout < 16 || r.cin_real < 1 || r.cin_real > r.cin_pad ||
        r.cout_real < 1 || r.cout_real > r.cout || (r.weight_mode != 0 && r.weight_mode != 1))
      return fail(AMX_ERR_INVALID, "pack_batch: bad request %d", i);
    w[i] = r.d_weight; wpk[i] = r.d_wpk; cr[i] = r.cin_real; cp[i] = r.cin_pad; co[i] = r.cout; md[i] = r.weight_mode; cor[i] = r.cout_real;
    q[i] = amx::conv_pick_q(r.cout, r.w, precision);
  }
  AMX_HIP(amx::launch_pack_weights_batch(count, w.data(), wpk.data(), cr.data(), cp.data(), co.data(), q.data(), md.data(), cor.data(), precision,
                                         (hipStream_t)stream));
  return AMX_OK;
}

size_t amx_conv3d_scratch_bytes(int c0, int c1, int cout, int n, int d, int hh, int w, int precision) {
  if (c1 || c0 % 16 || cout % 16 || precision < AMX_PREC_F16 || precision > AMX_PREC_F16X2_MX) return 0;
  return amx::conv_ks_part_bytes(c0, cout, n, d, hh, w, precision, amx::conv_pick_q(cout, w, precision));
}

int amx_conv3d_k3_reflect_ws(const void* d_x0, int c0, const void* d_x1, int c1, const float* d_weight,
                             const float* d_scale, const float* d_shift, int cout, int n, int d, int hh, int w,
                             int act, float slope, int precision, void* d_wpk, void* d_out16, float* d_out32,
                             void* d_scratch, size_t scratch_bytes, void* stream) {
  return conv3d_single(d_x0, c0, d_x1, c1, d_weight, 0, c0 + c1, cout, d_scale, d_shift, cout, n, d, hh, w, act, slope,
                       precision, d_wpk, d_out16, d_out32, stream, d_scratch, scratch_bytes);
}

int amx_conv3d_k3_reflect(const void* d_x0, int c0, const void* d_x1, int c1, const float* d_weight,
                          const float* d_scale, const float* d_shift, int cout, int n, int d, int hh, int w,
                          int act, float slope, int precision, void* d_wpk, void* d_out16, float* d_out32,
                          void* stream) {
  return conv3d_single(d_x0, c0, d_x1, c1, d_weight, 0, c0 + c1, cout, d_scale, d_shift, cout, n, d, hh, w, act, slope,
                       precision, d_wpk, d_out16, d_out32, stream);
}

int amx_conv3d_k3_reflect_ex(const void* d_x0, int c0, const void* d_x1, int c1, const float* d_weight, int weight_mode,
                             int cin_real, int cout_real, const float* d_scale, const float* d_shift, int cout, int n,
                             int d, int hh, int w, int act, float slope, int precision, void* d_wpk, void* d_out16,
                             float* d_out32, void* stream) {
  return conv3d_single(d_x0, c0, d_x1, c1, d_weight, weight_mode, cin_real, cout_real, d_scale, d_shift, cout, n, d, hh, w,
                       act, slope, precision, d_wpk, d_out16, d_out32, stream);
}

size_t amx_conv3d_upcat_merged_packed_bytes(int c0, int c1, int cout) {
  return 2 * (align_up((size_t)cout * c0 * 28 * 2, 256) + amx::conv_upmerge_packed_bytes(c1, cout, 0));   // room for [Wh | Wl]
}

int amx_conv3d_upcat_merged(const void* d_x0, int c0, const void* d_x1, int c1, const float* d_weight, const float* d_scale,
                            const float* d_shift, int cout, int n, int d, int hh, int w, int act, float slope, int precision,
                            void* d_wpk, void* d_partial, void* d_out16, void* stream) {
  if (!d_x0 || !d_x1 || !d_weight || !d_wpk || !d_partial || !d_out16) return fail(AMX_ERR_INVALID, "null argument");
  if (precision < AMX_PREC_F16 || precision > AMX_PREC_BF16X2) return fail(AMX_ERR_INVALID, "unsupported precision %d", precision);
  const bool split = is_split(precision);
  const long long eb = split ? 4 : 2;
  if (c0 != cout || !amx::conv_upmerge_eligible(c0, c1, cout, d, hh, w, 1, split))
    return fail(AMX_ERR_INVALID, "merged concat conv needs c0 == cout >= 32 (16 in the strict precisions), c1 %% 32 == 0, w >= 16, even dims "
                "(c0=%d c1=%d cout=%d dims %d,%d,%d)", c0, c1, cout, d, hh, w);
  hipStream_t st = (hipStream_t)stream;
  const int q = amx::conv_pick_q(cout, w, precision);
  char* wmerge = (char*)d_wpk + align_up((size_t)cout * c0 * 28 * 2 * (split ? 2 : 1), 256);
  AMX_HIP(amx::launch_pack_weights(d_weight, d_scale, d_wpk, c0, c0, cout, q, precision, st, 0, 0, c0 + c1));
  AMX_HIP(amx::launch_pack_upmerge(d_weight, d_scale, wmerge, c0, c0 + c1, c1, cout, precision, st));
  amx::ConvParams p;
  memset(&p, 0, sizeof p);
  p.N = n; p.D = d; p.H = hh; p.W = w; p.Cout = cout;
  p.src0 = (const char*)d_x0; p.C0 = c0;
  p.s0x = (long long)c0 * eb; p.s0y = p.s0x * w; p.s0z = p.s0y * hh; p.s0n = p.s0z * d;
  p.wpk = (const char*)d_wpk; p.act = AMX_ACT_NONE;
  p.out = (char*)d_partial;
  p.ox = (long long)cout * eb; p.oy = p.ox * w; p.oz = p.oy * hh; p.on = p.oz * d;
  AMX_HIP(amx::launch_conv(p, precision, q, st));
  amx::UpmergeParams u;
  memset(&u, 0, sizeof u);
  u.src = (const char*)d_x1; u.C1 = c1;
  u.sx = (long long)c1 * eb; u.sy = u.sx * (w / 2); u.sz = u.sy * (hh / 2); u.sn = u.sz * (d / 2);
  u.N = n; u.LD = d / 2; u.LH = hh / 2; u.LW = w / 2; u.Cout = cout;
  u.wpk = wmerge; u.part = (const char*)d_partial; u.out = (char*)d_out16;
  u.bias = d_shift; u.act = act; u.slope = slope;
  AMX_HIP(amx::launch_conv_upmerge(u, precision, st));
  return AMX_OK;
}

int amx_pool2(const void* d_in, void* d_out, int n, int dout, int hout, int wout, int c, int avg, int precision,
              void* stream) {
  if (!d_in || !d_out || c % 8) return fail(AMX_ERR_INVALID, "bad argument");
  AMX_HIP(amx::launch_pool2(d_in, d_out, n, dout, hout, wout, c, avg, precision, (hipStream_t)stream));
  return AMX_OK;
}

// ---------------------------------------------------------------- training-path operators
size_t amx_train_scratch_bytes(int c) { return amx::train_scratch_bytes(c); }

int amx_bn_train_forward(const void* d_x, void* d_y, const float* d_gamma, const float* d_beta, float eps, int n,
                         long long voxels, int c, int act, float slope, void* d_scratch, float* d_save_mean,
                         float* d_save_rstd, float* d_running_mean, float* d_running_var, float momentum, int precision,
                         void* stream) {
  if (!d_x || !d_y || !d_scratch || n < 1 || voxels < 1 || c % 8 || c > 2048) return fail(AMX_ERR_INVALID, "bad argument");
  AMX_HIP(amx::launch_bn_train_forward(d_x, d_y, d_gamma, d_beta, eps, (long long)n * voxels, c, act, slope, d_scratch, d_save_mean,
                                       d_save_rstd, d_running_mean, d_running_var, momentum, precision, (hipStream_t)stream));
  return AMX_OK;
}

int amx_bn_act_backward(const void* d_dy, const void* d_y, const void* d_x, const float* d_mean, const float* d_rstd,
                        const float* d_gamma, float* d_dgamma, float* d_dbeta, void* d_dx_framed, int n, int d, int hh, int w,
                        int c, int act, float slope, void* d_scratch, int precision, void* stream) {
  if (!d_dy || !d_y || !d_dx_framed || !d_scratch || c % 8 || c > 2048) return fail(AMX_ERR_INVALID, "bad argument");
  if (d_mean && (!d_x || !d_rstd || !d_dgamma || !d_dbeta)) return fail(AMX_ERR_INVALID, "norm backward needs x, rstd, dgamma, dbeta");
  AMX_HIP(amx::launch_bn_act_backward(d_dy, d_y, d_x ? d_x : d_y, d_mean, d_rstd, d_gamma, nullptr, d_dgamma, d_dbeta, d_dx_framed, n,
                                      d, hh, w, c, act, slope, d_scratch, precision, (hipStream_t)stream));
  return AMX_OK;
}

int amx_bn_act_backward_recompute(const void* d_dy, const void* d_x, const float* d_mean, const float* d_rstd, const float* d_gamma,
                                  const float* d_beta, float* d_dgamma, float* d_dbeta, void* d_dx_framed, int n, int d, int hh, int w,
                                  int c, int act, float slope, void* d_scratch, int precision, void* stream) {
  if (!d_dy || !d_x || !d_mean || !d_rstd || !d_dgamma || !d_dbeta || !d_dx_framed || !d_scratch || c % 8 || c > 2048)
    return fail(AMX_ERR_INVALID, "bad argument");
  AMX_HIP(amx::launch_bn_act_backward(d_dy, nullptr, d_x, d_mean, d_rstd, d_gamma, d_beta, d_dgamma, d_dbeta, d_dx_framed, n, d, hh, w,
                                      c, act, slope, d_scratch, precision, (hipStream_t)stream));
  return AMX_OK;
}

int amx_pad_fold(const void* d_g_framed, void* d_din, int n, int d, int hh, int w, int c, int accumulate, int precision,
                 void* stream) {
  if (!d_g_framed || !d_din || c % 8 || d < 2 || hh < 2 || w < 2) return fail(AMX_ERR_INVALID, "bad argument");
  AMX_HIP(amx::launch_pad_fold(d_g_framed, d_din, n, d, hh, w, c, accumulate, precision, (hipStream_t)stream));
  return AMX_OK;
}

int amx_adamw_step(const amx_adamw_tensor* tensors, int count, double lr, double beta1, double beta2, double eps, double weight_decay,
                   int maximize, void* stream) {
  static_assert(sizeof(amx_adamw_tensor) == 48, "six 64-bit fields");
  if (count < 0 || (count && !tensors)) return fail(AMX_ERR_INVALID, "bad argument");
  if (!(lr >= 0.0) || !(eps >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(weight_decay >= 0.0))
    return fail(AMX_ERR_INVALID, "adamw: lr %g, betas (%g, %g), eps %g, weight_decay %g out of range", lr, beta1, beta2, eps, weight_decay);
  for (int t = 0; t < count; ++t) {
    const amx_adamw_tensor& r = tensors[t];
    if (!r.param || !r.grad || !r.exp_avg || !r.exp_avg_sq || !r.step || r.numel < 0)
      return fail(AMX_ERR_INVALID, "adamw: tensor %d has a null pointer or a negative size", t);
  }
  AMX_HIP(amx::launch_adamw((const long long*)tensors, count, lr, beta1, beta2, eps, weight_decay, maximize, (hipStream_t)stream));
  return AMX_OK;
}

int amx_adamw_step_dev(const amx_adamw_tensor* tensors, int count, const double* d_hyper, int maximize, void* stream) {
  if (count < 0 || (count && !tensors) || !d_hyper) return fail(AMX_ERR_INVALID, "bad argument");
  for (int t = 0; t < count; ++t) {
    const amx_adamw_tensor& r = tensors[t];
    if (!r.param || !r.grad || !r.exp_avg || !r.exp_avg_sq || !r.step || r.numel < 0)
      return fail(AMX_ERR_INVALID, "adamw: tensor %d has a null pointer or a negative size", t);
  }
  AMX_HIP(amx::launch_adamw((const long long*)tensors, count, 0.0, 0.0, 0.0, 0.0, 0.0, maximize, (hipStream_t)stream, d_hyper));
  return AMX_OK;
}

int amx_pool2_max_backward(const void* d_dp, const void* d_in, void* d_din, int n, int dout, int hout, int wout, int c,
                           int accumulate, int precision, void* stream) {
  if (!d_dp || !d_in || !d_din || c % 8) return fail(AMX_ERR_INVALID, "bad argument");
  AMX_HIP(amx::launch_pool2_max_backward(d_dp, d_in, d_din, n, dout, hout, wout, c, accumulate, precision, (hipStream_t)stream));
  return AMX_OK;
}

size_t amx_conv3d_wgrad_scratch_bytes(int n, int d, int hh, int w, int cout, int cin_pad) {
  return amx::wgrad_scratch_bytes(n, d, hh, w, cout, cin_pad);
}

int amx_conv3d_wgrad(const void* d_dy, long long dy_sn, long long dy_sz, long long dy_sy, long long dy_sx, const void* d_x0,
                     int c0, const void* d_x1, int c1, int cin_real, int cout, int n, int d, int hh, int w, float* d_dw,
                     int accumulate, void* d_scratch, size_t scratch_bytes, int precision, void* stream) {
  if (!d_dy || !d_x0 || !d_dw || !d_scratch) return fail(AMX_ERR_INVALID, "null argument");
  if (c0 % 16 || c1 % 16 || c0 + c1 < 16 || cout % 16 || cout < 16 || cin_real < 1 || cin_real > c0 + c1)
    return fail(AMX_ERR_INVALID, "channel counts must be multiples of 16 (c0=%d c1=%d cout=%d)", c0, c1, cout);
  if (c1 && (!d_x1 || (d & 1) || (hh & 1) || (w & 1))) return fail(AMX_ERR_SHAPE, "upsampled segment needs even dims");
  if (d < 2 || hh < 2 || w < 2 || w > 128) return fail(AMX_ERR_SHAPE, "wgrad supports 2 <= w <= 128 (got %d)", w);
  if (scratch_bytes < amx::wgrad_scratch_bytes(n, d, hh, w, cout, c0 + c1))
    return fail(AMX_ERR_WORKSPACE, "scratch needs %zu bytes", amx::wgrad_scratch_bytes(n, d, hh, w, cout, c0 + c1));
  amx::WgradParams p;
  memset(&p, 0, sizeof p);
  p.dy = (const char*)d_dy; p.yn = dy_sn; p.yz = dy_sz; p.yy = dy_sy; p.yx = dy_sx;
  p.src0 = (const char*)d_x0; p.C0 = c0;
  p.s0x = (long long)c0 * 2; p.s0y = p.s0x * w; p.s0z = p.s0y * hh; p.s0n = p.s0z * d;
  if (c1) {
    p.src1 = (const char*)d_x1; p.C1 = c1; p.up_shift = 1;
    p.s1x = (long long)c1 * 2; p.s1y = p.s1x * (w / 2); p.s1z = p.s1y * (hh / 2); p.s1n = p.s1z * (d / 2);
  }
  p.N = n; p.D = d; p.H = hh; p.W = w; p.Cout = cout;
  AMX_HIP(amx::launch_wgrad(p, cin_real, d_dw, accumulate, d_scratch, precision, (hipStream_t)stream));
  return AMX_OK;
}

size_t amx_attention_scratch_bytes(int b, int heads, int n, int head_dim) {
  if (b < 1 || heads < 1 || n < 1 || head_dim < 2 || head_dim > 80) return 0;
  return amx::attention_scratch_bytes(b, heads, n);
}

int amx_attention_qknorm_rope(const float* d_q, const float* d_k, const float* d_v, const float* d_qn_w, const float* d_qn_b,
                              const float* d_kn_w, const float* d_kn_b, float norm_eps, const float* d_rope, int n_prefix, int b,
                              int n, int heads, int head_dim, float* d_out, void* d_scratch, size_t scratch_bytes, void* stream) {
  if (!d_q || !d_k || !d_v || !d_out || !d_scratch) return fail(AMX_ERR_INVALID, "null argument");
  if (b < 1 || n < 1 || heads < 1 || head_dim < 2 || head_dim > 80 || (head_dim & 1))
    return fail(AMX_ERR_INVALID, "attention: head_dim must be even and <= 80 (got %d), b, n, heads >= 1", head_dim);
  if ((!d_qn_w) != (!d_qn_b) || (!d_kn_w) != (!d_kn_b)) return fail(AMX_ERR_INVALID, "attention: norm weight and bias go together");
  if (n_prefix < 0 || n_prefix > n) return fail(AMX_ERR_INVALID, "attention: 0 <= n_prefix <= n");
  if (scratch_bytes < amx::attention_scratch_bytes(b, heads, n) || ((uintptr_t)d_scratch & 15))
    return fail(AMX_ERR_WORKSPACE, "attention scratch needs %zu bytes, 16-byte aligned", amx::attention_scratch_bytes(b, heads, n));
  AMX_HIP(amx::launch_attention(d_q, d_k, d_v, d_qn_w, d_qn_b, d_kn_w, d_kn_b, norm_eps, d_rope, n_prefix, b, n, heads, head_dim, d_out,
                                d_scratch, (hipStream_t)stream));
  return AMX_OK;
}

int amx_attention_prepared(const void* d_scratch, size_t scratch_bytes, int b, int n, int heads, int head_dim, float* d_out, void* stream) {
  if (!d_scratch || !d_out) return fail(AMX_ERR_INVALID, "null argument");
  if (b < 1 || n < 1 || heads < 1 || head_dim < 2 || head_dim > 80 || (head_dim & 1))
    return fail(AMX_ERR_INVALID, "attention: head_dim must be even and <= 80 (got %d), b, n, heads >= 1", head_dim);
  if (scratch_bytes < amx::attention_scratch_bytes(b, heads, n) || ((uintptr_t)d_scratch & 15))
    return fail(AMX_ERR_WORKSPACE, "attention scratch needs %zu bytes, 16-byte aligned", amx::attention_scratch_bytes(b, heads, n));
  void *Qp, *Kp, *Vt;
  amx::attention_operands((void*)d_scratch, b, heads, n, &Qp, &Kp, &Vt, nullptr, nullptr);
  AMX_HIP(amx::launch_attention_fwd(Qp, Kp, Vt, b, n, heads, head_dim, d_out, (hipStream_t)stream));
  return AMX_OK;
}

size_t amx_supcon_scratch_bytes(int n, int c) { return amx::supcon_scratch_bytes(n, c); }

int amx_supcon_loss(const float* d_feat, const int* d_labels, int n, int c, float temperature, int weigh_rarity,
                    int balance_denominator, int sqrt_mode, float* d_loss, float* d_grad, void* d_scratch,
                    size_t scratch_bytes, void* stream) {
  if (!d_feat || !d_labels || !d_loss || !d_scratch) return fail(AMX_ERR_INVALID, "null argument");
  if (n < 2 || n > 16384 || c < 1 || !(temperature > 0.f)) return fail(AMX_ERR_INVALID, "bad sizes (n=%d c=%d T=%g)", n, c, temperature);
  if (scratch_bytes < amx::supcon_scratch_bytes(n, c))
    return fail(AMX_ERR_WORKSPACE, "scratch needs %zu bytes (got %zu)", amx::supcon_scratch_bytes(n, c), scratch_bytes);
  AMX_HIP(amx::launch_supcon(d_feat, d_labels, n, c, temperature, weigh_rarity, balance_denominator, sqrt_mode, d_loss,
                             d_grad, d_scratch, (hipStream_t)stream));
  return AMX_OK;
}

size_t amx_instance_norm_scratch_bytes(int n, int c) { return amx::instnorm_scratch_bytes(n, c, 0); }

int amx_instance_norm(void* d_x, const float* d_gamma, const float* d_beta, float eps, int n, long long voxels, int c,
                      int act, float slope, void* d_scratch, int precision, void* stream) {
  if (!d_x || !d_scratch || c % 8 || c > 2048 || n < 1 || voxels < 1) return fail(AMX_ERR_INVALID, "bad argument");
  if (!d_gamma != !d_beta) return fail(AMX_ERR_INVALID, "gamma and beta come together");
  // f16x2mx (row-planar layout): this entry has no row length -- a sample is taken as ONE row of `voxels` voxels
  if (precision == AMX_PREC_F16X2_MX && voxels > (1 << 24)) return fail(AMX_ERR_INVALID, "f16x2mx: at most 2^24 voxels per sample here");
  AMX_HIP(amx::launch_instnorm(d_x, d_gamma, d_beta, eps, n, voxels, c, act, slope, d_scratch, precision,
                               (hipStream_t)stream, nullptr, 0, nullptr, (int)voxels));
  return AMX_OK;
}

int amx_upsample2_trilinear(const void* d_in, void* d_out, int n, int din, int hin, int win, int c, int precision,
                            void* stream) {
  if (!d_in || !d_out || c % 8 || n < 1 || din < 1 || hin < 1 || win < 1) return fail(AMX_ERR_INVALID, "bad argument");
  AMX_HIP(amx::launch_upsample2_trilinear(d_in, d_out, n, din, hin, win, c, precision, (hipStream_t)stream));
  return AMX_OK;
}

size_t amx_mindssc_scratch_bytes(int H, int W, int D) { return amx::mindssc_scratch_bytes(H, W, D); }

int amx_mindssc(const float* d_img, int H, int W, int D, int radius, int dilation, float* d_out, void* d_scratch,
                size_t scratch_bytes, void* stream) {
  if (!d_img || !d_out || !d_scratch) return fail(AMX_ERR_INVALID, "null argument");
  if (H < 1 || W < 1 || D < 1) return fail(AMX_ERR_SHAPE, "non-positive shape");
  if (radius < 1 || radius > 2 || dilation < 1 || dilation > 4)
    return fail(AMX_ERR_INVALID, "radius in {1, 2}, dilation in [1, 4] (got %d, %d)", radius, dilation);
  if (scratch_bytes < amx::mindssc_scratch_bytes(H, W, D))
    return fail(AMX_ERR_WORKSPACE, "scratch needs %zu bytes (got %zu)", amx::mindssc_scratch_bytes(H, W, D), scratch_bytes);
  AMX_HIP(amx::launch_mindssc(d_img, H, W, D, radius, dilation, d_out, d_scratch, (hipStream_t)stream));
  return AMX_OK;
}

int amx_avg_pool3d_cat(const float* d_a, int ca, float scale_a, const float* d_b, int cb, float scale_b, int H, int W, int D,
                       int g, float* d_out, void* stream) {
  if (!d_out || (ca > 0 && !d_a) || (cb > 0 && !d_b) || ca < 0 || cb < 0 || ca + cb < 1) return fail(AMX_ERR_INVALID, "bad argument");
  if (g < 1 || H < g || W < g || D < g) return fail(AMX_ERR_SHAPE, "pool size %d does not fit (%d,%d,%d)", g, H, W, D);
  AMX_HIP(amx::launch_pool_cat(d_a, ca, scale_a, d_b, cb, scale_b, H, W, D, g, d_out, (hipStream_t)stream));
  return AMX_OK;
}

int amx_box_filter3d(const float* d_in, float* d_out, int c, int H, int W, int D, int k, void* stream) {
  if (!d_in || !d_out || d_in == d_out || c < 1 || c > 65535) return fail(AMX_ERR_INVALID, "bad argument");
  if (H < 1 || W < 1 || D < 1) return fail(AMX_ERR_SHAPE, "non-positive shape");
  if (k < 3 || k > 9 || !(k & 1)) return fail(AMX_ERR_INVALID, "kernel size must be odd in [3, 9] (got %d)", k);
  AMX_HIP(amx::launch_box_filter(d_in, d_out, c, H, W, D, k, (hipStream_t)stream));
  return AMX_OK;
}

size_t amx_correlate_scratch_bytes(int h, int w, int d, int disp_hw) { return amx::correlate_scratch_bytes(h, w, d, disp_hw); }

int amx_correlate_ssd(const float* d_fix, const float* d_mov, int c, int h, int w, int d, int disp_hw, float* d_ssd,
                      long long* d_argmin, void* d_scratch, size_t scratch_bytes, void* stream) {
  if (!d_fix || !d_mov || !d_ssd || !d_scratch || c < 1) return fail(AMX_ERR_INVALID, "bad argument");
  if (h < 1 || w < 1 || d < 1) return fail(AMX_ERR_SHAPE, "non-positive shape");
  if (disp_hw < 1 || disp_hw > 3) return fail(AMX_ERR_INVALID, "disp_hw in {1, 2, 3} (got %d)", disp_hw);
  if (scratch_bytes < amx::correlate_scratch_bytes(h, w, d, disp_hw))
    return fail(AMX_ERR_WORKSPACE, "scratch needs %zu bytes (got %zu)", amx::correlate_scratch_bytes(h, w, d, disp_hw), scratch_bytes);
  AMX_HIP(amx::launch_correlate(d_fix, d_mov, c, h, w, d, disp_hw, d_ssd, d_argmin, d_scratch, (hipStream_t)stream));
  return AMX_OK;
}

static int mlp_check(int n, int cin, int width, int n_layers) {
  if (n < 1 || n > 2048) return fail(AMX_ERR_SHAPE, "mlp head: 1 <= n <= 2048 rows (got %d)", n);
  if (cin < 4 || cin % 4 || width < 8 || width % 8) return fail(AMX_ERR_SHAPE, "mlp head: cin %% 4 == 0 and width %% 8 == 0 (got %d, %d)", cin, width);
  if (cin > 4096 || width > 4096) return fail(AMX_ERR_SHAPE, "mlp head: at most 4096 features per layer (got %d, %d)", cin, width);
  if (n_layers < 1 || n_layers > 8) return fail(AMX_ERR_INVALID, "mlp head: 1 <= n_layers <= 8 (got %d)", n_layers);
  return AMX_OK;
}

int amx_mlp_head_forward(const float* d_x, int n, int cin, int width, int n_layers, const float* const* w,
                         const float* const* gamma, const float* const* beta, float* const* running_mean,
                         float* const* running_var, float eps, float momentum, int act, float slope, float* d_z, float* d_y,
                         float* d_mean, float* d_rstd, void* stream) {
  if (!d_x || !w || !gamma || !beta || !running_mean || !running_var || !d_z || !d_y || !d_mean || !d_rstd)
    return fail(AMX_ERR_INVALID, "null argument");
  if (int rc = mlp_check(n, cin, width, n_layers)) return rc;
  if (act != AMX_ACT_NONE && act != AMX_ACT_RELU && act != AMX_ACT_LRELU) return fail(AMX_ERR_INVALID, "unsupported activation");
  const size_t plane = (size_t)n * width;
  for (int l = 0; l < n_layers; ++l) {
    if (!w[l] || (!gamma[l] != !beta[l]) || (!running_mean[l] != !running_var[l])) return fail(AMX_ERR_INVALID, "layer %d: bad parameter pointers", l);
    AMX_HIP(amx::launch_mlp_layer_forward(l ? d_y + (l - 1) * plane : d_x, n, l ? width : cin, w[l], width, gamma[l], beta[l], eps,
                                          l + 1 < n_layers ? act : AMX_ACT_NONE, slope, d_z + l * plane, d_y + l * plane,
                                          d_mean + (size_t)l * width, d_rstd + (size_t)l * width, running_mean[l], running_var[l],
                                          momentum, (hipStream_t)stream));
  }
  return AMX_OK;
}

size_t amx_mlp_head_scratch_bytes(int n, int cin, int width) { return amx::mlp_backward_scratch_floats(n, cin, width) * sizeof(float); }

int amx_mlp_head_backward(const float* d_dy, const float* d_x, int n, int cin, int width, int n_layers,
                          const float* const* w, const float* const* gamma, int act, float slope, const float* d_z,
                          const float* d_y, const float* d_mean, const float* d_rstd, float* const* dw, float* const* dgamma,
                          float* const* dbeta, float* d_dx, void* d_scratch, size_t scratch_bytes, void* stream) {
  if (!d_dy || !d_x || !w || !gamma || !d_z || !d_y || !d_mean || !d_rstd || !dw || !dgamma || !dbeta || !d_scratch)
    return fail(AMX_ERR_INVALID, "null argument");
  if (int rc = mlp_check(n, cin, width, n_layers)) return rc;
  if (scratch_bytes < amx_mlp_head_scratch_bytes(n, cin, width))
    return fail(AMX_ERR_WORKSPACE, "scratch needs %zu bytes (got %zu)", amx_mlp_head_scratch_bytes(n, cin, width), scratch_bytes);
  const size_t plane = (size_t)n * width;
  float* dz = (float*)d_scratch;
  float* dprev = dz + plane;
  for (int l = n_layers - 1; l >= 0; --l) {
    if (!w[l] || !dw[l] || (gamma[l] && (!dgamma[l] || !dbeta[l]))) return fail(AMX_ERR_INVALID, "layer %d: bad parameter pointers", l);
    AMX_HIP(amx::launch_mlp_layer_backward(l + 1 < n_layers ? dprev : d_dy, d_y + l * plane, d_z + l * plane,
                                           d_mean + (size_t)l * width, d_rstd + (size_t)l * width, gamma[l],
                                           l + 1 < n_layers ? act : AMX_ACT_NONE, slope, l ? d_y + (l - 1) * plane : d_x, w[l], n,
                                           l ? width : cin, width, dz, gamma[l] ? dgamma[l] : nullptr, gamma[l] ? dbeta[l] : nullptr,
                                           dw[l], l ? dprev : d_dx, dprev + plane, (hipStream_t)stream));
  }
  return AMX_OK;
}

// ---- the heads / losses of ONE contrastive step as batches: n_heads chains of the same length run as one chain of launches
int amx_mlp_heads_forward(int n_heads, const float* const* d_x, int n, const int* cin, int width, int n_layers, const float* const* w,
                          const float* const* gamma, const float* const* beta, float* const* running_mean, float* const* running_var,
                          float eps, float momentum, int act, float slope, float* const* d_z, float* const* d_y, float* const* d_mean,
                          float* const* d_rstd, void* stream) {
  if (!d_x || !cin || !w || !gamma || !beta || !running_mean || !running_var || !d_z || !d_y || !d_mean || !d_rstd)
    return fail(AMX_ERR_INVALID, "null argument");
  if (n_heads < 1 || n_heads > amx::MLP_MAXB) return fail(AMX_ERR_INVALID, "1 <= heads <= %d (got %d)", amx::MLP_MAXB, n_heads);
  if (act != AMX_ACT_NONE && act != AMX_ACT_RELU && act != AMX_ACT_LRELU) return fail(AMX_ERR_INVALID, "unsupported activation");
  const size_t plane = (size_t)n * width;
  for (int h = 0; h < n_heads; ++h) {
    if (int rc = mlp_check(n, cin[h], width, n_layers)) return rc;
    if (!d_x[h] || !d_z[h] || !d_y[h] || !d_mean[h] || !d_rstd[h]) return fail(AMX_ERR_INVALID, "head %d: null buffer", h);
    for (int l = 0; l < n_layers; ++l) {
      const int i = h * n_layers + l;
      if (!w[i] || (!gamma[i] != !beta[i]) || (!running_mean[i] != !running_var[i]))
        return fail(AMX_ERR_INVALID, "head %d layer %d: bad parameter pointers", h, l);
    }
  }
  for (int l = 0; l < n_layers; ++l) {
    const float *x[amx::MLP_MAXB], *wl[amx::MLP_MAXB], *gl[amx::MLP_MAXB], *bl[amx::MLP_MAXB];
    float *z[amx::MLP_MAXB], *y[amx::MLP_MAXB], *mu[amx::MLP_MAXB], *rs[amx::MLP_MAXB], *rm[amx::MLP_MAXB], *rv[amx::MLP_MAXB];
    int k[amx::MLP_MAXB];
    for (int h = 0; h < n_heads; ++h) {
      const int i = h * n_layers + l;
      x[h] = l ? d_y[h] + (l - 1) * plane : d_x[h];
      k[h] = l ? width : cin[h];
      wl[h] = w[i]; gl[h] = gamma[i]; bl[h] = beta[i]; rm[h] = running_mean[i]; rv[h] = running_var[i];
      z[h] = d_z[h] + l * plane; y[h] = d_y[h] + l * plane; mu[h] = d_mean[h] + (size_t)l * width; rs[h] = d_rstd[h] + (size_t)l * width;
    }
    AMX_HIP(amx::launch_mlp_heads_layer_forward(n_heads, x, n, k, wl, width, gl, bl, eps, l + 1 < n_layers ? act : AMX_ACT_NONE, slope,
                                                z, y, mu, rs, rm, rv, momentum, (hipStream_t)stream));
  }
  return AMX_OK;
}

int amx_mlp_heads_backward(int n_heads, const float* const* d_dy, const float* const* d_x, int n, const int* cin, int width, int n_layers,
                           const float* const* w, const float* const* gamma, int act, float slope, const float* const* d_z,
                           const float* const* d_y, const float* const* d_mean, const float* const* d_rstd, float* const* dw,
                           float* const* dgamma, float* const* dbeta, float* const* d_dx, void* const* d_scratch, size_t scratch_bytes,
                           void* stream) {
  if (!d_dy || !d_x || !cin || !w || !gamma || !d_z || !d_y || !d_mean || !d_rstd || !dw || !dgamma || !dbeta || !d_dx || !d_scratch)
    return fail(AMX_ERR_INVALID, "null argument");
  if (n_heads < 1 || n_heads > amx::MLP_MAXB) return fail(AMX_ERR_INVALID, "1 <= heads <= %d (got %d)", amx::MLP_MAXB, n_heads);
  const size_t plane = (size_t)n * width;
  for (int h = 0; h < n_heads; ++h) {
    if (int rc = mlp_check(n, cin[h], width, n_layers)) return rc;
    if (scratch_bytes < amx_mlp_head_scratch_bytes(n, cin[h], width))
      return fail(AMX_ERR_WORKSPACE, "head %d: scratch needs %zu bytes (got %zu)", h, amx_mlp_head_scratch_bytes(n, cin[h], width), scratch_bytes);
    if (!d_dy[h] || !d_x[h] || !d_z[h] || !d_y[h] || !d_mean[h] || !d_rstd[h] || !d_scratch[h]) return fail(AMX_ERR_INVALID, "head %d: null buffer", h);
    for (int l = 0; l < n_layers; ++l) {
      const int i = h * n_layers + l;
      if (!w[i] || !dw[i] || (gamma[i] && (!dgamma[i] || !dbeta[i]))) return fail(AMX_ERR_INVALID, "head %d layer %d: bad parameter pointers", h, l);
    }
  }
  for (int l = n_layers - 1; l >= 0; --l) {
    const float *dy[amx::MLP_MAXB], *y[amx::MLP_MAXB], *z[amx::MLP_MAXB], *mu[amx::MLP_MAXB], *rs[amx::MLP_MAXB], *gl[amx::MLP_MAXB];
    const float *x[amx::MLP_MAXB], *wl[amx::MLP_MAXB];
    float *dz[amx::MLP_MAXB], *dg[amx::MLP_MAXB], *db[amx::MLP_MAXB], *dwl[amx::MLP_MAXB], *dx[amx::MLP_MAXB], *wpart[amx::MLP_MAXB];
    int k[amx::MLP_MAXB];
    for (int h = 0; h < n_heads; ++h) {
      const int i = h * n_layers + l;
      float* dzb = (float*)d_scratch[h];
      float* dprev = dzb + plane;
      dy[h] = l + 1 < n_layers ? dprev : d_dy[h];
      y[h] = d_y[h] + l * plane; z[h] = d_z[h] + l * plane; mu[h] = d_mean[h] + (size_t)l * width; rs[h] = d_rstd[h] + (size_t)l * width;
      gl[h] = gamma[i]; x[h] = l ? d_y[h] + (l - 1) * plane : d_x[h]; wl[h] = w[i]; k[h] = l ? width : cin[h];
      dz[h] = dzb; dg[h] = gamma[i] ? dgamma[i] : nullptr; db[h] = gamma[i] ? dbeta[i] : nullptr; dwl[h] = dw[i];
      dx[h] = l ? dprev : d_dx[h]; wpart[h] = dprev + plane;
    }
    AMX_HIP(amx::launch_mlp_heads_layer_backward(n_heads, dy, y, z, mu, rs, gl, l + 1 < n_layers ? act : AMX_ACT_NONE, slope, x, wl, n, k,
                                                 width, dz, dg, db, dwl, dx, wpart, (hipStream_t)stream));
  }
  return AMX_OK;
}

int amx_supcon_loss_batch(int n_losses, const float* const* d_feat, const int* const* d_labels, int n, int c, float temperature,
                          int weigh_rarity, int balance_denominator, int sqrt_mode, float* const* d_loss, float* const* d_grad,
                          void* d_scratch, size_t scratch_bytes, void* stream) {
  if (!d_feat || !d_labels || !d_loss || !d_scratch) return fail(AMX_ERR_INVALID, "null argument");
  if (n_losses < 1 || n_losses > amx::MLP_MAXB) return fail(AMX_ERR_INVALID, "1 <= losses <= %d (got %d)", amx::MLP_MAXB, n_losses);
  if (n < 2 || n > 16384 || c < 1 || !(temperature > 0.f)) return fail(AMX_ERR_INVALID, "bad sizes (n=%d c=%d T=%g)", n, c, temperature);
  if (scratch_bytes < n_losses * amx::supcon_scratch_bytes(n, c))
    return fail(AMX_ERR_WORKSPACE, "scratch needs %zu bytes (got %zu)", n_losses * amx::supcon_scratch_bytes(n, c), scratch_bytes);
  for (int b = 0; b < n_losses; ++b) {
    if (!d_feat[b] || !d_labels[b] || !d_loss[b]) return fail(AMX_ERR_INVALID, "loss %d: null buffer", b);
    if (d_grad && (!d_grad[b] != !d_grad[0])) return fail(AMX_ERR_INVALID, "gradients for all losses or for none");
  }
  AMX_HIP(amx::launch_supcon_batch(n_losses, d_feat, d_labels, n, c, temperature, weigh_rarity, balance_denominator, sqrt_mode, d_loss,
                                   d_grad, d_scratch, (hipStream_t)stream));
  return AMX_OK;
}

int amx_gather_labels_batch(const float* d_seg, int sd, int sh, int sw, int n_maps, const long long* const* d_coords, int p, const int* dims,
                            int views, int* const* d_labels, void* stream) {
  if (!d_seg || !d_coords || !dims || !d_labels || p < 1 || views < 1) return fail(AMX_ERR_INVALID, "gather_labels: bad arguments");
  if (n_maps < 1 || n_maps > amx::MLP_MAXB) return fail(AMX_ERR_INVALID, "1 <= maps <= %d (got %d)", amx::MLP_MAXB, n_maps);
  if (sd < 1 || sh < 1 || sw < 1) return fail(AMX_ERR_SHAPE, "gather_labels: non-positive shape");
  for (int b = 0; b < n_maps; ++b)
    if (!d_coords[b] || !d_labels[b] || dims[3 * b] < 1 || dims[3 * b + 1] < 1 || dims[3 * b + 2] < 1)
      return fail(AMX_ERR_SHAPE, "gather_labels: map %d: bad arguments", b);
  AMX_HIP(amx::launch_gather_labels_batch(d_seg, sd, sh, sw, n_maps, d_coords, p, dims, views, d_labels, (hipStream_t)stream));
  return AMX_OK;
}

int amx_sample_coords(const long long* d_draws, int n_draws, int num, int d0, int d1, int d2, long long* d_coords, void* stream) {
  if (!d_draws || !d_coords) return fail(AMX_ERR_INVALID, "null argument");
  if (num < 1 || n_draws < num || n_draws > 4096) return fail(AMX_ERR_INVALID, "1 <= num <= n_draws <= 4096 (got %d, %d)", num, n_draws);
  if (d0 < 1 || d1 < 1 || d2 < 1) return fail(AMX_ERR_SHAPE, "non-positive shape");
  AMX_HIP(amx::launch_sample_coords(d_draws, n_draws, num, d0, d1, d2, d_coords, (hipStream_t)stream));
  return AMX_OK;
}

int amx_import_input(const float* d_src, void* d_dst, int n, int cin, int d, int hh, int w, int precision, void* stream) {
  if (!d_src || !d_dst || n < 1 || d < 1 || hh < 1 || w < 1) return fail(AMX_ERR_INVALID, "import_input: bad arguments");
  if (cin < 1 || cin > 16) return fail(AMX_ERR_SHAPE, "import_input: 1 <= input channels <= 16 (got %d)", cin);
  if (precision != AMX_PREC_F16 && precision != AMX_PREC_BF16) return fail(AMX_ERR_INVALID, "import_input: f16 / bf16 storage");
  AMX_HIP(amx::launch_import_input(d_src, d_dst, n, cin, (long long)d * hh * w, precision, (hipStream_t)stream));
  return AMX_OK;
}

int amx_gather_labels(const float* d_seg, int sd, int sh, int sw, const long long* d_coords, int p, int d, int hh, int w, int views,
                      int* d_labels, void* stream) {
  if (!d_seg || !d_coords || !d_labels || p < 1 || views < 1) return fail(AMX_ERR_INVALID, "gather_labels: bad arguments");
  if (sd < 1 || sh < 1 || sw < 1 || d < 1 || hh < 1 || w < 1) return fail(AMX_ERR_SHAPE, "gather_labels: non-positive shape");
  AMX_HIP(amx::launch_gather_labels(d_seg, sd, sh, sw, d_coords, p, d, hh, w, views, d_labels, (hipStream_t)stream));
  return AMX_OK;
}

int amx_sample_perm(const long long* d_keys, int d0, int d1, int d2, int num, long long* d_coords, void* stream) {
  if (!d_keys || !d_coords) return fail(AMX_ERR_INVALID, "null argument");
  if (d0 < 1 || d1 < 1 || d2 < 1) return fail(AMX_ERR_SHAPE, "non-positive shape");
  const long long nvox = (long long)d0 * d1 * d2;
  if (nvox > 4096 || num < 1 || num > nvox) return fail(AMX_ERR_INVALID, "sample_perm: 1 <= num <= d0 d1 d2 <= 4096 (got %d of %lld)", num, nvox);
  AMX_HIP(amx::launch_sample_perm(d_keys, (int)nvox, num, d1, d2, d_coords, (hipStream_t)stream));
  return AMX_OK;
}

int amx_gather_rows(const void* d_src, int dtype, long long src_sn, long long src_sz, long long src_sy, long long src_sx, long long src_sc,
                    const long long* d_coords, int n, int p, int c, float* d_rows, void* stream) {
  if (!d_src || !d_coords || !d_rows || n < 1 || p < 1 || c < 1 || dtype < 0 || dtype > 2) return fail(AMX_ERR_INVALID, "gather_rows: bad arguments");
  AMX_HIP(amx::launch_gather_rows(d_src, dtype, src_sn, src_sz, src_sy, src_sx, src_sc, d_coords, n, p, c, d_rows, (hipStream_t)stream));
  return AMX_OK;
}

int amx_scatter_rows(const float* d_rows, const long long* d_coords, void* d_dst, int precision, long long dst_sn, long long dst_sz,
                     long long dst_sy, long long dst_sx, int n, int p, int c, int accumulate, void* stream) {
  if (!d_rows || !d_coords || !d_dst || n < 1 || p < 1 || c < 1 || (precision != AMX_PREC_F16 && precision != AMX_PREC_BF16))
    return fail(AMX_ERR_INVALID, "scatter_rows: bad arguments");
  AMX_HIP(amx::launch_scatter_rows(d_rows, d_coords, d_dst, precision, dst_sn, dst_sz, dst_sy, dst_sx, n, p, c, accumulate, (hipStream_t)stream));
  return AMX_OK;
}

size_t amx_conv3d_backward_sampled_scratch_bytes(int p) { return amx::sampled_conv_backward_scratch_bytes(p); }

int amx_conv3d_backward_sampled(const float* d_grows, const long long* d_coords, const void* d_x, int x_channels, const float* d_w, int n,
                                int p, int d, int hh, int w, int cout, int cin, float* d_dw, void* d_din, int din_channels, void* d_scratch,
                                size_t scratch_bytes, int precision, void* stream) {
  if (!d_scratch || scratch_bytes < amx::sampled_conv_backward_scratch_bytes(p))
    return fail(AMX_ERR_WORKSPACE, "conv3d_backward_sampled: scratch needs %zu bytes", amx::sampled_conv_backward_scratch_bytes(p));
  if (!d_grows || !d_coords || !d_x || !d_w || !d_dw || n < 1 || p < 1) return fail(AMX_ERR_INVALID, "conv3d_backward_sampled: bad arguments");
  if (precision != AMX_PREC_F16 && precision != AMX_PREC_BF16) return fail(AMX_ERR_INVALID, "conv3d_backward_sampled: f16 / bf16 storage");
  if (cout < 1 || cout > 16 || cin < 1 || cin > 16 || x_channels < cin || (d_din && din_channels < cin))
    return fail(AMX_ERR_SHAPE, "conv3d_backward_sampled: 1 <= cout, cin <= 16 (got %d, %d)", cout, cin);
  if (d < 2 || hh < 2 || w < 2 || p > 1024) return fail(AMX_ERR_SHAPE, "conv3d_backward_sampled: sizes >= 2, at most 1024 sampled voxels");
  AMX_HIP(amx::launch_sampled_conv_backward(d_grows, d_coords, d_x, x_channels, d_w, n, p, d, hh, w, cout, cin, d_dw, d_din, din_channels,
                                            d_scratch, precision, (hipStream_t)stream));
  return AMX_OK;
}

int amx_upsample2_trilinear_backward(const void* d_gout, void* d_gin, int n, int din, int hin, int win, int c, int precision,
                                     void* stream) {
  if (!d_gout || !d_gin || c % 8 || n < 1 || din < 1 || hin < 1 || win < 1) return fail(AMX_ERR_INVALID, "bad argument");
  AMX_HIP(amx::launch_upsample2_trilinear_backward(d_gout, d_gin, n, din, hin, win, c, precision, (hipStream_t)stream));
  return AMX_OK;
}

int amx_export_ncdhw(const void* d_src, int c, int n, int d, int hh, int w, float* d_out, int precision, void* stream) {
  if (!d_src || !d_out || c % 8 || c < 8 || n < 1 || d < 1 || hh < 1 || w < 1) return fail(AMX_ERR_INVALID, "bad argument");
  AMX_HIP(amx::launch_export_ncdhw(d_src, c, nullptr, 0, 0, n, d, hh, w, d_out, precision, (hipStream_t)stream));
  return AMX_OK;
}

int amx_import_ncdhw(const float* d_src, void* d_dst, int n, int c, int d, int hh, int w, long long dst_sn, long long dst_sz,
                     long long dst_sy, long long dst_sx, int accumulate, int precision, void* stream) {
  if (!d_src || !d_dst || c % 8 || c < 8 || n < 1 || d < 1 || hh < 1 || w < 1) return fail(AMX_ERR_INVALID, "bad argument");
  if (dst_sx < (long long)c * 2 || (dst_sx & 15) || (dst_sy & 15) || (dst_sz & 15) || (dst_sn & 15) || ((size_t)d_dst & 15))
    return fail(AMX_ERR_INVALID, "destination strides must be multiples of 16 bytes with a voxel pitch >= 2 * c");
  AMX_HIP(amx::launch_import_ncdhw(d_src, d_dst, n, c, d, hh, w, dst_sn, dst_sz, dst_sy, dst_sx, accumulate, precision,
                                   (hipStream_t)stream));
  return AMX_OK;
}

int amx_upcat_split_backward(const void* d_dcat, void* d_dskip, void* d_dlow, int n, int dlow, int hlow, int wlow, int c0, int c1,
                             int accumulate_skip, int precision, void* stream) {
  if (!d_dcat || (!d_dskip && c0 > 0) || !d_dlow || n < 1 || dlow < 1 || hlow < 1 || wlow < 1 || c0 < 0 || c1 < 8 || c0 % 8 || c1 % 8)
    return fail(AMX_ERR_INVALID, "bad argument");                       // c0 == 0: no skip part (the whole tensor is summed over children)
  AMX_HIP(amx::launch_upcat_split(d_dcat, d_dskip, d_dlow, n, dlow, hlow, wlow, c0, c1, accumulate_skip, 0, precision,
                                  (hipStream_t)stream));
  return AMX_OK;
}

int amx_upcat_split_backward_framed(const void* d_g_framed, void* d_dskip, void* d_dlow, int n, int dlow, int hlow, int wlow, int c0,
                                    int c1, int accumulate_skip, int precision, void* stream) {
  if (!d_g_framed || (!d_dskip && c0 > 0) || !d_dlow || n < 1 || dlow < 1 || hlow < 1 || wlow < 1 || c0 < 0 || c1 < 8 || c0 % 8 || c1 % 8)
    return fail(AMX_ERR_INVALID, "bad argument");                       // c0 == 0: no skip part (the whole tensor is summed over children)
  AMX_HIP(amx::launch_upcat_split(d_g_framed, d_dskip, d_dlow, n, dlow, hlow, wlow, c0, c1, accumulate_skip, 1, precision,
                                  (hipStream_t)stream));
  return AMX_OK;
}

}  // extern "C"
