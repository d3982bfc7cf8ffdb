/* anatomix_amd -- public C ABI of the MI355X (gfx950) anatomix UNet feature-extraction path.
 *
 * The reference (neel-dey/anatomix) is pure Python and has no FFI: its boundary for this path is
 * the nn.Module surface of anatomix.model.network.Unet.  Each entry point below names the
 * reference interface it stands in for (paths relative to the reference checkout).  The Python
 * mirror anatomix_amd.model.network.Unet binds these symbols with ctypes.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer owned by the caller (PyTorch's allocator);
 *     the library owns only its packed-weight buffers (allocated at amx_unet_create/load);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it and
 *     the library never synchronises, so calls are capturable in a hipGraph;
 *   - every function returns 0 on success or a negative amx_status; amx_last_error() gives the
 *     message of the last failure on the calling thread.  Nothing aborts or throws;
 *   - a handle is not thread-safe; use one handle per module per device.
 */
#ifndef ANATOMIX_AMD_H
#define ANATOMIX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMX_VERSION 100 /* 0.1.0 */

typedef enum amx_status {
  AMX_OK = 0,
  AMX_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
  AMX_ERR_SHAPE = -2,       /* spatial size not divisible by 2^num_downs, bottleneck < 2, ... */
  AMX_ERR_NOT_LOADED = -3,  /* forward before every conv received its parameters */
  AMX_ERR_WORKSPACE = -4,   /* workspace too small / misaligned */
  AMX_ERR_HIP = -5,         /* a HIP runtime call failed */
  AMX_ERR_OVERFLOW = -6     /* f16 / f16x2 storage: a value outside the f16 range (or a NaN) was produced */
} amx_status;

enum { AMX_NORM_NONE = 0, AMX_NORM_BATCH_EVAL = 1, AMX_NORM_INSTANCE = 2, AMX_NORM_INSTANCE_AFFINE = 3 };
enum { AMX_ACT_NONE = 0, AMX_ACT_RELU = 1, AMX_ACT_LRELU = 2 };
enum { AMX_POOL_MAX = 0, AMX_POOL_AVG = 1 };
enum { AMX_INTERP_NEAREST = 0, AMX_INTERP_TRILINEAR = 1 };
/* Storage precision of activations and packed weights (every MFMA accumulates in fp32).
 *   F16 / BF16     one 16-bit value per element;
 *   F16X2 / BF16X2 "strict": every element is a hi + lo pair of 16-bit values (22 / 16 mantissa bits), a voxel stores
 *                  [hi(C) | lo(C)], each product runs as three MFMAs (Wh*xh + Wh*xl + Wl*xh).  fp32-grade results --
 *                  the reference's inference callers run the network in fp32
 *                  (anatomix/registration/convex_adam_utils.py:194-219: .float().cuda(), no autocast).  BF16X2 keeps
 *                  fp32's exponent range; F16X2 is ~8x more accurate but overflows beyond 65504 like F16.
 *   F16X2_MX       strict storage (f16 hi + lo pairs) whose two CORRECTION products run on the block-scaled fp8 matrix
 *                  instruction of CDNA4 at twice the f16 rate: Wh*xh on the f16 MFMA, Wh*xl + Wl*xh from e4m3 copies the
 *                  producing kernels store beside the pair (6C bytes per voxel, row-planar: see below).  2.0 instead of 3.0
 *                  MFMA-equivalents per product; operands carry 15-16 significant bits instead of 22: the InstanceNorm
 *                  variant `anatomix-dev` stays inside the 1e-3 tolerance of the fp32 reference. */
enum { AMX_PREC_F16 = 0, AMX_PREC_BF16 = 1, AMX_PREC_F16X2 = 2, AMX_PREC_BF16X2 = 3, AMX_PREC_F16X2_MX = 4 };
/* Tensor layouts of the single-operator entries below.  F16 .. BF16X2: channels-last voxels, [n][d][h][w][c] (x2 in the strict
 * precisions: [hi(c) | lo(c)]).  F16X2_MX: ROW-PLANAR, 6 c bytes per voxel -- a row (n, z, y) of w voxels is 3 c/16 planes of w x 32
 * bytes: plane k = the f16 hi halves of channels 16k .. 16k+15 of the row's voxels, plane c/16 + k their lo halves, plane 2 c/16 + k
 * the e4m3 copies [e4m3(2^11 lo) x 16 | e4m3(hi) x 16]; so the 32 bytes per voxel that a convolution stage gathers are contiguous
 * along x.  Convolutions read hi and the copies; norm apply / pool / upsample write all three; a convolution writes hi and lo only
 * (its output is normalised before the next convolution reads it).  amx_instance_norm, which has no row length, takes a sample as
 * one row of `voxels` voxels. */

/* Constructor arguments of anatomix/model/network.py:262-279 (Unet.__init__) that shape the
 * arithmetic.  dimension is fixed at 3, pad_type at 'reflect', residual_connection at False. */
typedef struct amx_unet_cfg {
  int32_t input_nc;       /* network.py:265; 1 .. 16 (the fused sliding-window entries need 1) */
  int32_t output_nc;      /* network.py:266; any positive count (the sliding-window accumulation needs <= 32) */
  int32_t num_downs;      /* network.py:267 */
  int32_t ngf;            /* network.py:268; 8, 16, 24 (the reference default) or 32 */
  int32_t norm;           /* AMX_NORM_*   <- norm=      network.py:269,127-168 */
  float norm_eps;         /*              <- norm_eps=  network.py:278 */
  int32_t activation;     /* AMX_ACT_*    <- activation= network.py:271,171-204 */
  float act_slope;        /* 0.3 for 'lrelu' (network.py:191) */
  int32_t final_act;      /* AMX_ACT_*    <- final_act= network.py:270 */
  int32_t pooling;        /* AMX_POOL_*   <- pooling=   network.py:275,297 */
  int32_t interp;         /* AMX_INTERP_* <- interp=    network.py:276,407 */
  int32_t doubleconv;     /* network.py:273 */
  int32_t use_skip;       /* network.py:277 */
  int32_t precision;      /* AMX_PREC_*: storage type of activations / weights (fp32 accumulate); the *X2 values are the
                           * strict mode that meets the reference's fp32 results to ~1e-5 */
} amx_unet_cfg;

typedef struct amx_unet amx_unet_t;

int amx_version(void);

/* Test aid (no reference counterpart): overwrites the LDS of every compute unit with `pattern` (one 160 KB workgroup per CU, twice over),
 * so that a test can show that a kernel's results do not depend on what the previous kernel left there (tests/test_stem_fused_gpu.py). */
int amx_debug_fill_lds(unsigned pattern, void* stream);
const char* amx_last_error(void);

/* Unet.__init__ (network.py:262-465): builds the layer plan (same module indices as the
 * reference's nn.Sequential) and allocates packed-weight storage on the current device. */
int amx_unet_create(amx_unet_t** out, const amx_unet_cfg* cfg);
void amx_unet_destroy(amx_unet_t* h);

/* Layer plan queries: number of children of Unet.model, and per conv its module index
 * (the `i` of state_dict key `model.{i}.weight`), channel counts and the index of the norm
 * module that follows it (-1 if none). */
int amx_unet_num_modules(const amx_unet_t* h);
int amx_unet_num_convs(const amx_unet_t* h);
int amx_unet_conv_info(const amx_unet_t* h, int conv, int* module_idx, int* cin, int* cout, int* norm_module_idx);

/* nn.Module.load_state_dict for one Conv3d (+ the BatchNorm3d that follows it), see
 * anatomix/model/load_from_hf.py:39-49.  d_weight is fp32 [Cout][Cin][3][3][3]; d_bias (conv
 * bias), d_gamma/d_beta (norm affine), d_mean/d_var (BatchNorm running stats) are fp32 [Cout]
 * or NULL.  Eval-mode BatchNorm is folded into the packed weights here. */
int amx_unet_load_conv(amx_unet_t* h, int module_idx, const float* d_weight, const float* d_bias,
                       const float* d_gamma, const float* d_beta, const float* d_mean,
                       const float* d_var, void* stream);

/* One record per kernel launch of a profiled forward (amx_unet_forward_profiled). */
typedef struct amx_launch_record {
  char kernel[64];      /* kernel template instance, e.g. "conv3d_k3<f16,1x8x32,q1,nch1,o0>" */
  int32_t module_idx;   /* index of the conv / pool module in Unet.model */
  int32_t cin, cout;    /* logical input / output channels */
  int32_t n, d, h, w;   /* output extent of the launch */
  float ms;             /* hipEvent time from this launch to the next (includes the boundary) */
  double flops;         /* algorithmic: 2*27*cin*cout*voxels (0 for pools) */
  double bytes;         /* algorithmic: input read once + output written once + weights */
} amx_launch_record;

/* Bytes of scratch the forward needs for a batch of n volumes of d x h x w. */
size_t amx_unet_workspace_bytes(const amx_unet_t* h, int n, int d, int w_h, int w_w);

/* Unet.forward(input) standard branch (network.py:530-548), eval mode.
 * d_x: fp32 [n][input_nc][d][h][w]; d_y: fp32 [n][output_nc][d][h][w] (both NCDHW contiguous). */
int amx_unet_forward(amx_unet_t* h, const float* d_x, float* d_y, int n, int d, int hh, int w,
                     void* d_workspace, size_t workspace_bytes, void* stream);

/* Range safety of the f16 / f16x2 storage modes.  f16 overflows at 65504; folded BatchNorm gains of a real checkpoint are
 * unbounded, and an Inf that meets a ReLU or a max-pool can come out finite and wrong.  Every epilogue that stores f16
 * values therefore tests them, and when one is out of range (or NaN):
 *   - the forward that produced it has its output tensor overwritten with NaN on the device (last launch of the
 *     forward; sliding-window accumulation volumes are the caller's and are covered by the status call);
 *   - the NEXT call on this handle, or this function, returns AMX_ERR_OVERFLOW once (amx_last_error says which mode to
 *     switch to) and clears the condition.
 * synchronize != 0 waits for `stream` first, so the answer covers every forward enqueued so far.  bf16 / bf16x2 keep
 * fp32's exponent range (they overflow where the fp32 reference does) and never raise it. */
int amx_unet_numerics_status(amx_unet_t* h, int synchronize, void* stream);

/* Channels and resolution level (spatial size = input size >> level) of `feat` after Unet.model[module_idx]
 * as Unet.forward's `layers` branch sees it: for an nn.Upsample id that is AFTER torch.cat((skip, up), 1)
 * (network.py:500-502). */
int amx_unet_module_info(const amx_unet_t* h, int module_idx, int* channels, int* level);

/* Unet.forward(input, layers, encode_only) (network.py:475-529), eval mode.  tap_modules: HOST array of n_taps
 * module ids, strictly ascending (the reference collects in traversal order whatever the order of `layers`);
 * d_tap_out: HOST array of n_taps device buffers, fp32 NCDHW [n][channels][d>>level][h>>level][w>>level] per
 * amx_unet_module_info.  stop_module >= 0 reproduces encode_only: the forward ends after that module (d_y is
 * then left untouched unless the network got that far), modules after it do not run -- in particular an in-place
 * activation that would otherwise overwrite the tapped tensor.  Tap semantics follow the reference exactly:
 *   conv id followed by a norm ........ the pre-norm convolution output;
 *   norm id followed by an activation . the ACTIVATED tensor (nn.ReLU(inplace=True) aliases it, network.py:188-196);
 *   activation / pool id .............. that module's output;   upsample id: the concatenated tensor. */
int amx_unet_forward_taps(amx_unet_t* h, const float* d_x, float* d_y, int n, int d, int hh, int w,
                          void* d_workspace, size_t workspace_bytes, const int* tap_modules, int n_taps,
                          float* const* d_tap_out, int stop_module, void* stream);

/* Same forward, with a hipEvent recorded on `stream` around every launch; synchronises the
 * stream before returning and fills up to max_records records (profiling aid for bench.py's
 * roofline figures -- not used in the timed region). */
int amx_unet_forward_profiled(amx_unet_t* h, const float* d_x, float* d_y, int n, int d, int hh, int w,
                              void* d_workspace, size_t workspace_bytes, void* stream,
                              amx_launch_record* records, int max_records, int* n_records);

/* One sliding-window step of monai.inferers.sliding_window_inference as called from
 * anatomix/registration/convex_adam_utils.py:202-219: runs the forward on the roi-sized window of
 * volume d_vol (fp32 [vd][vh][vw], single channel) whose corner is (oz,oy,ox) and accumulates
 *   d_acc[c][oz+z][oy+y][ox+x] += d_wmap[z][y][x] * feature[c][z][y][x]
 * (d_acc fp32 [output_nc][vd][vh][vw]).  Windows that overlap must be issued on one stream. */
int amx_unet_forward_window(amx_unet_t* h, const float* d_vol, int vd, int vh, int vw, int oz, int oy,
                            int ox, int rd, int rh, int rw, const float* d_wmap, float* d_acc,
                            void* d_workspace, size_t workspace_bytes, void* stream);

/* n_windows sliding-window steps in one call (the reference hands sw_batch_size windows to the predictor at a
 * time, convex_adam_utils.py:205; any batching gives the same result because every norm mode is per sample or
 * uses running statistics).  offsets_zyx: HOST array [n_windows][3] of window corners.  The stem and the output
 * convolution run once per window in the given order (so overlapping accumulations are ordered on the stream);
 * all layers between them run on the whole batch.  Workspace: amx_unet_workspace_bytes(h, n_windows, rd, rh, rw). */
int amx_unet_forward_windows(amx_unet_t* h, const float* d_vol, int vd, int vh, int vw, int n_windows,
                             const int* offsets_zyx, int rd, int rh, int rw, const float* d_wmap, float* d_acc,
                             void* d_workspace, size_t workspace_bytes, void* stream);

/* Two window batches in flight.  Same as amx_unet_forward_windows, for a caller that alternates TWO streams (and two
 * workspaces) over consecutive window batches: slot = 0 / 1 alternately.  Everything up to the output conv of batch k+1 overlaps
 * batch k; the accumulating launches of a batch first wait for the other slot's accumulations, so overlapping windows are still
 * added in window order and the result is bit-identical to the single-stream sequence.  The caller joins both streams before
 * reading d_acc.  (MONAI's sliding_window_inference runs its sw_batch groups strictly one after the other,
 * convex_adam_utils.py:205-219 is the call.) */
int amx_unet_forward_windows_pipelined(amx_unet_t* h, const float* d_vol, int vd, int vh, int vw, int n_windows,
                                       const int* offsets_zyx, int rd, int rh, int rw, const float* d_wmap, float* d_acc,
                                       void* d_workspace, size_t workspace_bytes, int slot, void* stream);

/* Final step of sliding_window_inference: d_acc[c][v] /= d_cnt[v] in place. */
int amx_sw_normalize(float* d_acc, const float* d_cnt, int channels, long long voxels, void* stream);

/* d_cnt[oz+z][oy+y][ox+x] += d_wmap[z][y][x] for one window (the count map of
 * sliding_window_inference). */
int amx_sw_count(float* d_cnt, int vd, int vh, int vw, int oz, int oy, int ox, int rd, int rh, int rw,
                 const float* d_wmap, void* stream);

/* Single-layer entry (per-kernel parity / roofline tests): y = act(conv3x3x3_reflect(cat(x0,
 * up2_nearest(x1))) * scale + shift).  x0: 16-bit NDHWC [n][d][h][w][c0]; x1: 16-bit NDHWC
 * [n][d/2][h/2][w/2][c1] or NULL (c1 = 0); d_weight fp32 [cout][c0+c1][27]; d_scale/d_shift fp32
 * [cout] or NULL; output 16-bit NDHWC (d_out16) or fp32 NCDHW (d_out32), exactly one non-NULL.
 * d_wpk is scratch for the packed weights: amx_conv3d_packed_bytes(c0+c1, cout) bytes. */
size_t amx_conv3d_packed_bytes(int cin, int cout);
int amx_conv3d_k3_reflect(const void* d_x0, int c0, const void* d_x1, int c1, const float* d_weight,
                          const float* d_scale, const float* d_shift, int cout, int n, int d, int hh,
                          int w, int act, float slope, int precision, void* d_wpk, void* d_out16,
                          float* d_out32, void* stream);

/* The same call with caller-owned scratch for the layers that split K across workgroups (deep levels with few voxels:
 * conv3d_k3_ks writes fp32 partial tensors per K slice, a second kernel sums them in slice order -- deterministic).
 * amx_conv3d_scratch_bytes returns 0 when the shape does not split; d_scratch may then be NULL. */
size_t amx_conv3d_scratch_bytes(int c0, int c1, int cout, int n, int d, int hh, int w, int precision);
int amx_conv3d_k3_reflect_ws(const void* d_x0, int c0, const void* d_x1, int c1, const float* d_weight,
                             const float* d_scale, const float* d_shift, int cout, int n, int d, int hh,
                             int w, int act, float slope, int precision, void* d_wpk, void* d_out16,
                             float* d_out32, void* d_scratch, size_t scratch_bytes, void* stream);

/* The same convolution with a described weight tensor, so that callers need no host-side reshuffling:
 *   weight_mode 0: d_weight fp32 [cout_real][cin_real][27]; input channels cin_real .. c0+c1-1 and output channels
 *                  cout_real .. cout-1 are zero padding (the stem: one real input channel in a 16-channel tensor);
 *   weight_mode 1: DATA-GRADIENT weights taken from the forward tensor d_weight fp32 [cin_real][cout_real][27]
 *                  (= the forward conv's [Cout][Cin][27]): taps flipped, channels transposed -- run on the zero-framed
 *                  output gradient and followed by amx_pad_fold it is the adjoint of the forward conv. */
int amx_conv3d_k3_reflect_ex(const void* d_x0, int c0, const void* d_x1, int c1, const float* d_weight, int weight_mode,
                             int cin_real, int cout_real, const float* d_scale, const float* d_shift, int cout, int n,
                             int d, int hh, int w, int act, float slope, int precision, void* d_wpk, void* d_out16,
                             float* d_out32, void* stream);

/* Packing of MANY layers in one launch (the training step: every conv's weights change with every optimizer step, and packing each
 * inside its conv call put 40 small launches on the step's critical path).  Request i packs d_weight (described as for
 * amx_conv3d_k3_reflect_ex: weight_mode 0 / 1, cin_real of cin_pad stored input channels, cout_real of cout) into d_wpk
 * (amx_conv3d_packed_bytes(cin_pad, cout) bytes) for a layer of spatial WIDTH w (the tile shape -- hence the packing -- depends on
 * it).  The conv is then called with weight_mode | AMX_WEIGHTS_PREPACKED and that d_wpk (d_weight may be NULL).  Plain 16-bit
 * precisions; not for the 16 + up32 -> 16 merged-tap layer. */
#define AMX_WEIGHTS_PREPACKED 16
typedef struct amx_pack_req {
  const float* d_weight;
  void* d_wpk;
  int weight_mode, cin_real, cin_pad, cout_real, cout, w;
} amx_pack_req;
int amx_conv3d_pack_batch(const amx_pack_req* reqs, int count, int precision, void* stream);

/* Data gradient of the reflect-padded conv WITHOUT the padded-domain detour (amx_train.hip): d_dy_framed is the zero-framed output
 * gradient [n][d+4][hh+4][w+4][c_dy] (the buffer amx_bn_act_backward writes); amx_conv3d_dgrad_interior runs the forward kernel on its
 * interior with the halo read from the frame (the zero-padded correlation with the flipped, transposed FORWARD weights d_weight
 * fp32 [cin_real = forward Cout][cout_real = forward Cin][27]; weight_flags 0 or AMX_WEIGHTS_PREPACKED with a weight_mode-1 packing)
 * into the dense 16-bit d_out16 [n][d][hh][w][cout]; amx_conv3d_dgrad_fold_shell then adds the folded shell terms of the reflect
 * adjoint to the voxels with a coordinate 1 or size - 2.  Together = amx_conv3d_k3_reflect_ex(weight_mode 1) on the framed domain
 * followed by amx_pad_fold.  Shapes: what the z-march kernels take (16 / 32 channels in, 16 / 32 out, not 32 -> 16, w >= 32, d, hh >= 8):
 * amx_conv3d_dgrad_interior_supported returns 1. */
int amx_conv3d_dgrad_interior_supported(int c_dy, int cout, int d, int hh, int w, int precision);
int amx_conv3d_dgrad_interior(const void* d_dy_framed, int c_dy, const float* d_weight, int weight_flags, int cin_real, int cout_real, int cout,
                              int n, int d, int hh, int w, int precision, void* d_wpk, void* d_out16, void* stream);
size_t amx_conv3d_dgrad_shell_scratch_bytes(void);     /* d_scratch of amx_conv3d_dgrad_fold_shell: the shell sources' MFMA fragment table */
int amx_conv3d_dgrad_fold_shell(const void* d_dy_framed, int c_dy, const float* d_weight, int co_real, int ci_real, void* d_dx, int c_dx, int n,
                                int d, int hh, int w, int precision, void* d_scratch, void* stream);

/* nn.MaxPool3d(2) / nn.AvgPool3d(2) on a 16-bit NDHWC tensor (network.py:297,368). */
int amx_pool2(const void* d_in, void* d_out, int n, int d_out_, int h_out, int w_out, int c, int avg,
              int precision, void* stream);

/* nn.InstanceNorm3d(c, eps, affine = d_gamma != NULL) followed by the activation, IN PLACE on a 16-bit NDHWC
 * tensor [n][voxels][c] (network.py:157-163: 'instance' / 'instance_affine'; biased variance over the voxels of
 * each (n, c) plane).  d_scratch: amx_instance_norm_scratch_bytes(n, c) bytes of fp32 partial sums. */
size_t amx_instance_norm_scratch_bytes(int n, int c);
int amx_instance_norm(void* d_x, const float* d_gamma, const float* d_beta, float eps, int n, long long voxels,
                      int c, int act, float slope, void* d_scratch, int precision, void* stream);

/* nn.Upsample(scale_factor=2, mode='trilinear') (align_corners=False; network.py:407): 16-bit NDHWC
 * [n][din][hin][win][c] -> [n][2 din][2 hin][2 win][c]. */
int amx_upsample2_trilinear(const void* d_in, void* d_out, int n, int din, int hin, int win, int c,
                            int precision, void* stream);

/* Adjoint of torch.cat((skip, nn.Upsample(2, 'nearest')(low)), 1) (network.py:500-502, 545) applied to the data gradient of the
 * concat conv: d_dcat 16-bit [n][2 dlow][2 hlow][2 wlow][c0 + c1] -> d_dskip [n][2 dlow][2 hlow][2 wlow][c0] (first c0
 * channels; added to the existing content when accumulate_skip) and d_dlow [n][dlow][hlow][wlow][c1] (sum of the 8 children of
 * every low-resolution voxel).  One pass over d_dcat.  c0 == 0 (d_dskip may be NULL): no skip part, c0 and c1 multiples of 8. */
int amx_upcat_split_backward(const void* d_dcat, void* d_dskip, void* d_dlow, int n, int dlow, int hlow, int wlow, int c0, int c1,
                             int accumulate_skip, int precision, void* stream);
/* The same with the reflect-padding adjoint (amx_pad_fold) applied while reading: d_g_framed is the data-gradient conv's raw result
 * on the padded domain, [n][2 dlow + 4][2 hlow + 4][2 wlow + 4][c0 + c1]; the folded full-resolution gradient is never written and
 * the children are summed in fp32 before the one rounding.  c0 == 0 (d_dskip may be NULL): the tensor has no skip part. */
int amx_upcat_split_backward_framed(const void* d_g_framed, void* d_dskip, void* d_dlow, int n, int dlow, int hlow, int wlow, int c0,
                                    int c1, int accumulate_skip, int precision, void* stream);

/* Layout conversions between the library's 16-bit channels-last activations and torch's fp32 NCDHW tensors, for the
 * feature taps of the differentiable forward (network.py:475-529 returns the taps as fp32 NCDHW tensors) and their
 * gradients coming back: export  d_src 16-bit [n][d][h][w][c] -> d_out fp32 [n][c][d][h][w];
 * import  d_src fp32 [n][c][d][h][w] -> d_dst 16-bit through byte strides (voxel pitch dst_sx >= 2 c; lets the interior of
 * a zero-framed gradient buffer be the destination), adding to the destination when accumulate != 0.  c % 8 == 0. */
/* The network input for the training path: d_src fp32 [n][cin][d][hh][w] (1 <= cin <= 16) -> d_dst 16-bit channels-last
 * [n][d][hh][w][16], the real channels first and the rest zero -- one pass (the differentiable forward pads the input to one
 * 16-channel chunk so that its first conv is an ordinary 16 -> ngf layer, network.py:310-333 with input_nc = 1 or 2). */
int amx_import_input(const float* d_src, void* d_dst, int n, int cin, int d, int hh, int w, int precision, void* stream);
int amx_export_ncdhw(const void* d_src, int c, int n, int d, int hh, int w, float* d_out, int precision, void* stream);
int amx_import_ncdhw(const float* d_src, void* d_dst, int n, int c, int d, int hh, int w, long long dst_sn, long long dst_sz,
                     long long dst_sy, long long dst_sx, int accumulate, int precision, void* stream);

/* Adjoint of amx_upsample2_trilinear (autograd of nn.Upsample(2, 'trilinear'), network.py:407, in the training path):
 * d_gout 16-bit [n][2 din][2 hin][2 win][c] -> d_gin [n][din][hin][win][c]. */
int amx_upsample2_trilinear_backward(const void* d_gout, void* d_gin, int n, int din, int hin, int win, int c, int precision,
                                     void* stream);

/* The same layer as amx_conv3d_k3_reflect with c1 > 0, computed the way amx_unet_forward runs the wider nearest-upsample
 * concat layers (network.py:403-435: nn.Upsample(2,'nearest') -> torch.cat((skip, up), 1) -> nn.Conv3d(3, reflect) -> norm ->
 * act): (1) the ordinary 27-tap convolution over the skip channels writes raw partial sums into d_partial; (2) the merged-tap
 * convolution over the upsampled channels at LOW resolution -- the 27 taps over a nearest-upsampled tensor collapse to 2x2x2
 * parity-dependent taps, 3.4x fewer multiply-accumulates -- adds them, the bias and the activation.  Needs c0 == cout,
 * c1 a multiple of 32, cout >= 32 (>= 16 in the strict precisions), w >= 16, even dims; AMX_ERR_INVALID otherwise.  d_wpk:
 * amx_conv3d_upcat_merged_packed_bytes(c0, c1, cout) bytes of scratch for both packings; d_partial: n*d*h*w*cout*2 bytes. */
size_t amx_conv3d_upcat_merged_packed_bytes(int c0, int c1, int cout);
int amx_conv3d_upcat_merged(const void* d_x0, int c0, const void* d_x1, int c1, const float* d_weight, const float* d_scale,
                            const float* d_shift, int cout, int n, int d, int hh, int w, int act, float slope, int precision,
                            void* d_wpk, void* d_partial, void* d_out16, void* stream);

/* ---- Training-path operators (the UNet inside the contrastive step, pretraining/models/supcl_model.py:603-661, runs in
 * train mode and is differentiated).  All activations / gradients: dense 16-bit channels-last [n][d][h][w][c];
 * parameter gradients and statistics fp32.  d_scratch: amx_train_scratch_bytes(c) bytes. */
size_t amx_train_scratch_bytes(int c);

/* nn.BatchNorm3d in TRAIN mode + activation (network.py:148-152,171-196): y = act(gamma (x - mean) rstd + beta) with the
 * biased batch variance over n * voxels rows; saves mean / rstd (each [c], may be NULL), updates the running statistics
 * (running_var with the unbiased variance; pointers may be NULL).  d_y may equal d_x. */
int amx_bn_train_forward(const void* d_x, void* d_y, const float* d_gamma, const float* d_beta, float eps, int n,
                         long long voxels, int c, int act, float slope, void* d_scratch, float* d_save_mean,
                         float* d_save_rstd, float* d_running_mean, float* d_running_var, float momentum, int precision,
                         void* stream);

/* Adjoint of the above (and of a bare activation when d_mean == NULL): dz = dy * act'(y); dbeta = sum dz;
 * dgamma = sum dz * xhat; dx = gamma rstd (dz - dbeta / M - xhat dgamma / M), written into the INTERIOR of the zero-framed
 * buffer d_dx_framed [n][d+4][h+4][w+4][c] (the caller zeroes the frame once), the input of the data-gradient conv. */
int amx_bn_act_backward(const void* d_dy, const void* d_y, const void* d_x, const float* d_mean, const float* d_rstd,
                        const float* d_gamma, float* d_dgamma, float* d_dbeta, void* d_dx_framed, int n, int d, int hh, int w,
                        int c, int act, float slope, void* d_scratch, int precision, void* stream);
/* The same adjoint of norm + activation WITHOUT reading the activated output: act' only needs the sign of its argument, and that
 * argument is recomputed from the saved pre-norm tensor, z = a x + b with a = rstd gamma, b = beta - mean rstd gamma (the
 * coefficients amx_bn_train_forward applied; d_gamma / d_beta NULL = 1 / 0).  Each of the two passes reads two tensors instead of
 * three.  For relu / lrelu the result equals amx_bn_act_backward's wherever the stored y kept the sign of z (always in bf16; in f16
 * except for |z| below its smallest subnormal). */
int amx_bn_act_backward_recompute(const void* d_dy, const void* d_x, const float* d_mean, const float* d_rstd, const float* d_gamma,
                                  const float* d_beta, float* d_dgamma, float* d_dbeta, void* d_dx_framed, int n, int d, int hh, int w,
                                  int c, int act, float slope, void* d_scratch, int precision, void* stream);

/* Adjoint of the reflect padding of nn.Conv3d(padding='same', padding_mode='reflect') (network.py:310-318).  The data
 * gradient of the conv is amx_conv3d_k3_reflect run on the framed (d+4)(h+4)(w+4) output gradient with the weights
 * flipped along the three taps and transposed (cin <-> cout); this folds the result [n][d+4][h+4][w+4][c] back onto
 * [n][d][h][w][c]: voxel i collects padded positions i, -1 (if i == 1) and L (if i == L-2) per axis. */
int amx_pad_fold(const void* d_g_framed, void* d_din, int n, int d, int hh, int w, int c, int accumulate, int precision,
                 void* stream);

/* Adjoint of nn.MaxPool3d(2) (network.py:297,368): d_dp [n][dout][hout][wout][c] is routed to the FIRST maximum of each
 * 2x2x2 window of d_in [n][2 dout][2 hout][2 wout][c] (z, y, x order: torch's tie rule). */
int amx_pool2_max_backward(const void* d_dp, const void* d_in, void* d_din, int n, int dout, int hout, int wout, int c,
                           int accumulate, int precision, void* stream);

/* Weight gradient of nn.Conv3d(k3, reflect): d_dw fp32 [cout][cin_real][3][3][3] (+)= sum over voxels of dy (x) input.
 * d_dy: 16-bit [n][d][h][w][cout] through BYTE strides (so the interior of a framed buffer can be passed);
 * input = cat(d_x0 [c0 ch, full resolution], nearest-upsampled d_x1 [c1 ch, half resolution]) as in the forward;
 * cin_real <= c0 + c1 trims zero-padded input channels (the stem: 1 real channel padded to 16). */
size_t amx_conv3d_wgrad_scratch_bytes(int n, int d, int hh, int w, int cout, int cin_pad);
int amx_conv3d_wgrad(const void* d_dy, long long dy_sn, long long dy_sz, long long dy_sy, long long dy_sx, const void* d_x0,
                     int c0, const void* d_x1, int c1, int cin_real, int cout, int n, int d, int hh, int w, float* d_dw,
                     int accumulate, void* d_scratch, size_t scratch_bytes, int precision, void* stream);

/* Fused attention core of the 3D ViT variant `anatomix-dev-vit` (PrimusV2-S: anatomix/model/vit3d/architectures.py:231-260,
 * registry entry load_from_hf.py:25-35; the blocks themselves live in the third-party dynamic-network-architectures / timm
 * packages, see oracle/vit_ref.py).  Replaces, inside every EVA block, the sequence
 *     q, k = q_norm(q), k_norm(k)                  per-head nn.LayerNorm(head_dim), the wrapper's qk_norm (architectures.py:108-115)
 *     q, k = rope(q), rope(k)                      on the tokens after the n_prefix register tokens (x*cos + rot(x)*sin)
 *     out  = softmax(q k^T / sqrt(head_dim)) v     torch F.scaled_dot_product_attention
 * d_q / d_k / d_v / d_out: fp32 [b][n][heads*head_dim] (the layout the q / k / v projections produce and the output
 * projection consumes).  d_*n_w / d_*n_b: fp32 [head_dim] or NULL (no QK norm).  d_rope: fp32 [n - n_prefix][2*head_dim] =
 * per token [sin | cos], or NULL.  head_dim even, <= 80.  f16 MFMA operands, fp32 LayerNorm / rotation / softmax / accumulate. */
size_t amx_attention_scratch_bytes(int b, int heads, int n, int head_dim);
int amx_attention_qknorm_rope(const float* d_q, const float* d_k, const float* d_v, const float* d_qn_w, const float* d_qn_b,
                              const float* d_kn_w, const float* d_kn_b, float norm_eps, const float* d_rope, int n_prefix, int b,
                              int n, int heads, int head_dim, float* d_out, void* d_scratch, size_t scratch_bytes, void* stream);

/* softmax(q k^T / sqrt(head_dim)) v alone, on the f16 operands a previous amx_attention_qknorm_rope call with the SAME
 * (b, n, heads, head_dim) left in d_scratch: the flash kernel without its preparation pass (what one EVA block of amx_vit_forward
 * launches after its q | k and v projections; bench.py times it as the ViT's dominant kernel). */
int amx_attention_prepared(const void* d_scratch, size_t scratch_bytes, int b, int n, int heads, int head_dim, float* d_out, void* stream);

/* ---- The whole 3D ViT variant `anatomix-dev-vit` (PrimusV2-S) as one forward ------------------------------------------------
 * Replaces PrimusV2.forward (anatomix/model/vit3d/architectures.py:231-260 with the wrapper's extensions :89-165, tokenizer
 * deep_tokenizer.py:12-68, registry entry load_from_hf.py:25-35): conv tokenizer (stem + three stride-2 residual stages + 1x1x1
 * projection), position embedding + register tokens, `depth` EVA blocks (LayerNorm, q/k/v, per-head QK LayerNorm + rotary
 * embedding + softmax attention, inner LayerNorm, output projection, LayerScale; SwiGLU MLP with sub-LayerNorm), final LayerNorm,
 * patch decoder (three ConvTranspose3d(k = 2, s = 2) with channel LayerNorm + GELU between them), ChannelDemean.  Every kernel is
 * this library's (amx_tokenizer.hip, amx_gemm.hip, amx_attention.hip); none of it calls a vendor GEMM / conv library.
 * The blocks themselves live in third-party packages absent from this image: arithmetic follows oracle/vit_ref.py's restatement
 * (PARITY WITH THE UPSTREAM PACKAGE UNPINNED). */
typedef struct amx_vit_cfg {
  int32_t input_channels;       /* 1 */
  int32_t num_classes;          /* output channels, a multiple of 4 */
  int32_t embed_dim;            /* architectures.py:20-25 (PRIMUS_CONFIGS: 396 / 792 / 864 / 1056), a multiple of 4, <= 1280 */
  int32_t depth;                /* eva_depth */
  int32_t heads;                /* eva_numheads; head_dim = embed_dim / heads even, <= 80 */
  int32_t num_register_tokens;  /* architectures.py:117-120 */
  int32_t grid_d, grid_h, grid_w; /* token grid = input_shape / 8 */
  int32_t hidden;               /* SwiGLU hidden width (int(embed_dim * 4 * 2 / 3)), 16 .. 3072 (a multiple of 4 above 1088) */
  int32_t dec1, dec2;           /* channel widths after the first / second transposed conv (oracle/vit_ref.py::vit_plan) */
  int32_t qk_norm;              /* architectures.py:108-115 */
  int32_t scale_attn_inner;     /* LayerNorm between attention and its output projection */
  int32_t layer_scale;          /* gamma_1 / gamma_2 present (init_values is not None) */
  float in_eps;                 /* tokenizer InstanceNorm epsilon (deep_tokenizer.py:66-68) */
  int32_t out_norm;             /* 0 none, 1 ChannelDemean (architectures.py:28-33); other modes are applied by the caller */
  int32_t decoder_split;        /* 1: hi + lo f16 operands in the decoder (fp32-grade), 0: plain f16 */
  int32_t stem_split;           /* 1: the stem's output (the largest tensor) keeps its lo plane; 0: single f16 -- half the bytes of the
                                 * two kernels that touch it, +3.7e-4 rel-L2 on the output (measured) */
} amx_vit_cfg;
typedef struct amx_vit amx_vit_t;

int amx_vit_create(amx_vit_t** out, const amx_vit_cfg* cfg);
void amx_vit_destroy(amx_vit_t* h);
/* Parameter slots: amx_vit_param_name(h, i) is the state_dict key (anatomix_amd.model.vit3d.PrimusV2) of slot i. */
int amx_vit_num_params(const amx_vit_t* h);
const char* amx_vit_param_name(const amx_vit_t* h, int idx);
/* d_params[i]: contiguous fp32 device tensor of slot i; d_rope: fp32 [grid tokens][2 * head_dim] = per token [sin | cos].
 * Packs / copies everything into library-owned memory (call again after the parameters change). */
int amx_vit_load(amx_vit_t* h, const float* const* d_params, int count, const float* d_rope, void* stream);
size_t amx_vit_workspace_bytes(const amx_vit_t* h, int n);
/* d_x: fp32 [n][1][8 grid_d][8 grid_h][8 grid_w]; d_y: fp32 [n][num_classes][same volume].  d_ws: amx_vit_workspace_bytes(h, n)
 * bytes, 256-byte aligned.  n_blocks < 0: all blocks (a smaller count is a debugging aid). */
int amx_vit_forward(amx_vit_t* h, const float* d_x, float* d_y, int n, void* d_ws, size_t ws_bytes, int n_blocks, void* stream);
int amx_vit_debug_read(amx_vit_t* h, const char* name, void* d_dst, size_t max_bytes, size_t* bytes, void* stream);
/* y[m][n] = sum_k x[m][k] w[n][k] + b[n] through the ViT's MFMA product kernel (f16 operands, or hi + lo pairs when split != 0):
 * a test entry for nn.Linear-shaped products; synchronous (allocates and frees its operand buffers). */
int amx_linear(const float* d_x, const float* d_w, const float* d_b, int m, int k, int n, int split, float* d_y, void* stream);

/* SupPatchNCELoss.forward + its backward (pretraining/models/supcl_model.py:73-226) for one nce layer.
 * d_feat: fp32 [n][c], n = views * patches anchors in (view, patch) order (features.view(ntps * num_patches, nc),
 * supcl_model.py:134); d_labels: int32 [n], the segmentation class of every anchor (the label gather of :100-112,
 * tiled over the views); flags = opt.weigh_rarity / opt.balance_denominator / (opt.weighting_mode == 'sqrt');
 * d_loss: 1 float; d_grad: fp32 [n][c] = d loss / d feat, or NULL for the forward only.
 * d_scratch: amx_supcon_scratch_bytes(n, c) bytes.  Deterministic (fixed-order reductions). */
size_t amx_supcon_scratch_bytes(int n, int c);
int amx_supcon_loss(const float* d_feat, const int* d_labels, int n, int c, float temperature, int weigh_rarity,
                    int balance_denominator, int sqrt_mode, float* d_loss, float* d_grad, void* d_scratch,
                    size_t scratch_bytes, void* stream);

/* Projection head of PatchSampleF (pretraining/models/pretraining_networks.py:338-350, applied at :505-511) in train
 * mode: n_layers x [Linear(bias=False) -> BatchNorm1d(batch statistics) -> activation], no activation after the last
 * layer, affine parameters optional per layer.  All device tensors fp32, row-major:
 *   d_x [n][cin] the sampled features; w[l] [width][cin or width]; gamma[l] / beta[l] [width] or NULL (affine=False);
 *   running_mean[l] / running_var[l] [width] updated in place like nn.BatchNorm1d (momentum, unbiased variance) or NULL;
 *   d_z, d_y [n_layers][n][width]: pre-norm and post-activation outputs of every layer (kept for the backward;
 *   the head's output is d_y + (n_layers - 1) * n * width); d_mean, d_rstd [n_layers][width] the batch statistics.
 * w, gamma, beta, running_mean, running_var are HOST arrays of n_layers device pointers.
 * Limits: 1 <= n <= 2048, cin % 4 == 0, width % 8 == 0, 1 <= n_layers <= 8.  Two kernels per layer. */
int amx_mlp_head_forward(const float* d_x, int n, int cin, int width, int n_layers, const float* const* w,
                         const float* const* gamma, const float* const* beta, float* const* running_mean,
                         float* const* running_var, float eps, float momentum, int act, float slope, float* d_z, float* d_y,
                         float* d_mean, float* d_rstd, void* stream);

/* Backward of amx_mlp_head_forward: d_dy [n][width] = d loss / d head output.  Writes dw[l] (shape of w[l]), dgamma[l] /
 * dbeta[l] (where gamma[l] is not NULL) and d_dx [n][cin] (nullable).  d_scratch: amx_mlp_head_scratch_bytes(n, cin, width).
 * Four small kernels per layer; deterministic (fixed-order reductions). */
size_t amx_mlp_head_scratch_bytes(int n, int cin, int width);
int amx_mlp_head_backward(const float* d_dy, const float* d_x, int n, int cin, int width, int n_layers,
                          const float* const* w, const float* const* gamma, int act, float slope, const float* d_z,
                          const float* d_y, const float* d_mean, const float* d_rstd, float* const* dw, float* const* dgamma,
                          float* const* dbeta, float* d_dx, void* d_scratch, size_t scratch_bytes, void* stream);

/* Backward of a k3 reflect convolution whose output gradient is nonzero only at p sampled voxels per sample -- the contrastive step
 * taps the output conv (module 65) at 512 voxels and nothing else reads the network's output (supcl_model.py:801-843): d_grows fp32
 * [n][p][cout] at d_coords int64 [p][3]; d_x the conv's 16-bit channels-last input [n][d][h][w][x_channels]; d_w fp32
 * [cout][cin][27].  d_dw fp32 [cout][cin][27] is written; d_din (optional) is a ZERO-FILLED 16-bit [n][d][h][w][din_channels] tensor
 * that receives the data gradient at the <= 27 p voxels per sample it is nonzero at.  Same rounding points as scatter_rows + the dense
 * weight / data gradient (rows rounded to the storage type, 16-bit weights in the data gradient, fp32 sums, one rounding), fixed
 * summation order.  cout, cin <= 16.  d_scratch: amx_conv3d_backward_sampled_scratch_bytes(p) bytes (neighbour masks of the samples,
 * partial weight-gradient sums). */
size_t amx_conv3d_backward_sampled_scratch_bytes(int p);
int amx_conv3d_backward_sampled(const float* d_grows, const long long* d_coords, const void* d_x, int x_channels, const float* d_w, int n,
                                int p, int d, int hh, int w, int cout, int cin, float* d_dw, void* d_din, int din_channels, void* d_scratch,
                                size_t scratch_bytes, int precision, void* stream);

/* The heads and losses of ONE contrastive step as batches (supcl_model.py:801-843 walks the nce layers one by one; the chains are
 * independent and identical in structure).  A replayed HIP graph pays per DEPENDENT node, not per byte: n_heads chains of ~30 small
 * launches each become one chain whose every launch serves all heads (<= 8; same rows n, same width, same depth; own input widths
 * cin[h]).  Arrays over (head, layer) are flattened [h * n_layers + l]; d_z / d_y / d_mean / d_rstd / d_scratch are per head with
 * the single-head layouts and sizes above.  Per head the arithmetic and its order are those of amx_mlp_head_forward / _backward:
 * bit-identical results.  amx_supcon_loss_batch: n_losses problems of one shape [n][c] (scratch: n_losses slices of
 * amx_supcon_scratch_bytes(n, c); d_grad NULL: losses only); amx_gather_labels_batch: the class ids of n_maps feature maps of sizes
 * dims[3 m ..] = (d, h, w) from one segmentation. */
int amx_mlp_heads_forward(int n_heads, const float* const* d_x, int n, const int* cin, int width, int n_layers, const float* const* w,
                          const float* const* gamma, const float* const* beta, float* const* running_mean, float* const* running_var,
                          float eps, float momentum, int act, float slope, float* const* d_z, float* const* d_y, float* const* d_mean,
                          float* const* d_rstd, void* stream);
int amx_mlp_heads_backward(int n_heads, const float* const* d_dy, const float* const* d_x, int n, const int* cin, int width, int n_layers,
                           const float* const* w, const float* const* gamma, int act, float slope, const float* const* d_z,
                           const float* const* d_y, const float* const* d_mean, const float* const* d_rstd, float* const* dw,
                           float* const* dgamma, float* const* dbeta, float* const* d_dx, void* const* d_scratch, size_t scratch_bytes,
                           void* stream);
int amx_supcon_loss_batch(int n_losses, const float* const* d_feat, const int* const* d_labels, int n, int c, float temperature,
                          int weigh_rarity, int balance_denominator, int sqrt_mode, float* const* d_loss, float* const* d_grad,
                          void* d_scratch, size_t scratch_bytes, void* stream);
int amx_gather_labels_batch(const float* d_seg, int sd, int sh, int sw, int n_maps, const long long* const* d_coords, int p, const int* dims,
                            int views, int* const* d_labels, void* stream);

/* Patch coordinates of PatchSampleF's no-mask branch (pretraining_networks.py:443-470: `randperm(n_voxels)[:num]`, then the
 * flat ids unravelled): d_draws int64 [n_draws], values in [0, d0*d1*d2) drawn WITH replacement by the caller's generator;
 * d_coords int64 [num][3] receives the C-order coordinates of the first `num` distinct draws in draw order -- the same
 * distribution as a random permutation's head, without sorting every voxel.  num <= n_draws <= 4096.  One launch. */
/* Sampled feature taps of the contrastive step (pretraining_networks.py:472-480 gathers feat[:, :, x, y, z] at num_patches
 * coordinates): rows[n][p][c] (fp32) = src[n][coords[p]][c] read in place -- src a 16-bit channels-last tensor (dtype = AMX_PREC_F16 /
 * AMX_PREC_BF16) or an fp32 tensor (dtype 2, e.g. the NCDHW network output), ELEMENT strides -- and its adjoint
 * dst[n][coords[p]][c] (= or +=) rows[n][p][c] into a 16-bit channels-last tensor (BYTE strides; fp32 add, one rounding; coords must
 * be distinct).  d_coords: int64 [p][3] = (z, y, x). */
int amx_gather_rows(const void* d_src, int dtype, long long src_sn, long long src_sz, long long src_sy, long long src_sx, long long src_sc,
                    const long long* d_coords, int n, int p, int c, float* d_rows, void* stream);
int amx_scatter_rows(const float* d_rows, const long long* d_coords, void* d_dst, int precision, long long dst_sn, long long dst_sz,
                     long long dst_sy, long long dst_sx, int n, int p, int c, int accumulate, void* stream);

int amx_sample_coords(const long long* d_draws, int n_draws, int num, int d0, int d1, int d2, long long* d_coords, void* stream);
/* The same branch on a SMALL grid (d0 d1 d2 <= 4096 voxels, e.g. 512 patches from an 8^3 map = all of it): d_keys holds one random
 * non-negative 64-bit key per voxel (torch.randint: torch's generator seeds it); the voxels are ordered by key (ties by index) in one
 * launch and the first `num` are unravelled to d_coords [num][3] -- a uniformly random permutation prefix, replacing
 * torch.randperm(n)[:num] (pretraining_networks.py:443-470) and its dozen small launches. */
int amx_sample_perm(const long long* d_keys, int d0, int d1, int d2, int num, long long* d_coords, void* stream);
/* Class ids of the sampled patches for SupPatchNCELoss (supcl_model.py:100-123: F.interpolate(seg, size = the feature map, mode =
 * 'nearest') gathered at the patch coordinates): d_seg fp32 [sd][sh][sw] (one label map shared by the views), d_coords int64
 * [p][3] in the (d, hh, w) grid of the feature map -> d_labels int32 [views][p] (round to nearest, tiled over the views). */
int amx_gather_labels(const float* d_seg, int sd, int sh, int sw, const long long* d_coords, int p, int d, int hh, int w, int views,
                      int* d_labels, void* stream);

/* The optimizer step of the contrastive step: torch.optim.AdamW as the reference builds it for netG and netF
 * (pretraining/models/supcl_model.py:510-516, 584-590; stepped at :628-661), every parameter tensor of one optimizer in ONE
 * launch per 48 tensors.  `tensors` is a HOST array; every pointer in it is a device pointer to contiguous fp32 (step: one fp32
 * scalar holding the step count t AFTER this step's increment -- torch's capturable state layout, so the call can be captured
 * in a HIP graph and a state_dict moves between this and torch.optim.AdamW).  In place:
 *   p *= 1 - lr wd;  m += (g - m)(1 - beta1);  v = v beta2 + (1 - beta2) g g;
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)            (maximize: g -> -g).  amsgrad is not offered.
 * The hyper-parameters are doubles because torch holds them as Python floats: 1 - beta2, 1 - lr wd and the bias corrections are
 * formed in double and rounded once (1 - 0.999f differs from (float)(1 - 0.999) by 1.3e-5). */
typedef struct {
  void* param;
  const void* grad;
  void* exp_avg;
  void* exp_avg_sq;
  const void* step;
  long long numel;
} amx_adamw_tensor;
int amx_adamw_step(const amx_adamw_tensor* tensors, int count, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int maximize, void* stream);
/* The same step with the hyper-parameters READ FROM THE DEVICE when the kernel runs: d_hyper = five doubles {lr, beta1, beta2, eps,
 * weight_decay}.  By-value arguments are frozen into a captured HIP graph; the reference changes lr every epoch through its
 * schedulers (pretraining/models/base_model.py: update_learning_rate / get_scheduler), so a graph-replayed step takes them from a
 * buffer that a captured host-to-device copy refreshes (anatomix_amd/pretraining/optim.py).  Values are not range-checked here. */
int amx_adamw_step_dev(const amx_adamw_tensor* tensors, int count, const double* d_hyper, int maximize, void* stream);

/* ---- registration feature post-processing (what the reference does to the extracted features before the convex
 * optimisation; all fp32, planar [C][H][W][D] device tensors, batch 1 as everywhere in that pipeline) ---- */

/* MINDSSC(img, radius, dilation) (anatomix/registration/convex_adam_utils.py:311-406): d_img fp32 [H][W][D] ->
 * d_out fp32 [12][H][W][D], channels in the reference's final (permuted) order.  radius in {1, 2}, dilation in [1, 4].
 * d_scratch: amx_mindssc_scratch_bytes(H, W, D).  The global mean the descriptor variance is clamped against
 * (`mind_var.mean().item()`, :389-393) stays on the device: no host synchronisation. */
size_t amx_mindssc_scratch_bytes(int H, int W, int D);
int amx_mindssc(const float* d_img, int H, int W, int D, int radius, int dilation, float* d_out, void* d_scratch,
                size_t scratch_bytes, void* stream);

/* F.avg_pool3d(cat(scale_a * a, scale_b * b), g, stride=g) in one pass: the `pred * downscale_feat_scalar`,
 * merge_features' concat (instance_optimization.py:111-117) and the grid_sp pooling of
 * run_convex_adam_with_network_feats.py:164-205.  d_a fp32 [ca][H][W][D] (may be NULL with ca == 0), d_b [cb][H][W][D];
 * d_out [ca + cb][H/g][W/g][D/g] (floor division like avg_pool3d). */
int amx_avg_pool3d_cat(const float* d_a, int ca, float scale_a, const float* d_b, int cb, float scale_b, int H, int W, int D,
                       int g, float* d_out, void* stream);

/* One pass of apply_avg_pool3d (convex_adam_utils.py:105-131): F.avg_pool3d(x, k, padding=k/2, stride=1), zero padding
 * counted in the divisor.  d_in, d_out fp32 [c][H][W][D], distinct buffers; k odd, 3 <= k <= 9. */
int amx_box_filter3d(const float* d_in, float* d_out, int c, int H, int W, int D, int k, void* stream);

/* correlate(mind_fix, mind_mov, disp_hw, ...) (convex_adam_utils.py:409-491): d_fix, d_mov fp32 [c][h][w][d] (the pooled
 * features) -> d_ssd fp32 [(2 disp_hw + 1)^3][h][w][d] (twice box-filtered sum of squared differences, displacement
 * index (dx * k + dy) * k + dz as the reference's view/transpose/reshape leaves it) and d_argmin int64 [h][w][d]
 * (nullable).  disp_hw in {1, 2, 3}.  d_scratch: amx_correlate_scratch_bytes(h, w, d, disp_hw). */
size_t amx_correlate_scratch_bytes(int h, int w, int d, int disp_hw);
int amx_correlate_ssd(const float* d_fix, const float* d_mov, int c, int h, int w, int d, int disp_hw, float* d_ssd,
                      long long* d_argmin, void* d_scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ANATOMIX_AMD_H */
