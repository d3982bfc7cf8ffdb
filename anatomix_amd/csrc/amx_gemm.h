// anatomix_amd -- internal declarations of the 3D ViT kernels (amx_gemm.hip, amx_tokenizer.hip); not part of the public C ABI.
#pragma once
#include <hip/hip_runtime.h>

namespace amx {

enum GemmEpi {
  EPI_F32 = 0,      // out32[m][n] = acc + bias
  EPI_F16 = 1,      // out16[m][n] = acc + bias
  EPI_RESID = 2,    // out32[m][n] += gamma[n] * (acc + bias[n])
  EPI_SWIGLU = 3,   // tile pairs (gate, value): out16[m][h] = silu(gate + bg) * (value + bx)
  EPI_SCATTER = 4,  // ConvTranspose3d(k = 2, s = 2): n = parity * Cp + c -> out32[voxel(2 z + dz, 2 y + dy, 2 x + dx)][c]
  EPI_TOKENS = 5,   // out32[b][nreg + v][n] = acc + bias + pos[v][n]   (m = b * V + v)
  EPI_PLANAR = 6,   // last ConvTranspose3d: features ordered ((dz, dy), c, dx); planar fp32 out[b][c][2 z + dz][2 y + dy][2 x + dx] - sub[b][c]
  EPI_SCATTER_LN = 7, // ConvTranspose3d + channel LayerNorm + GELU: hi / lo operand rows of the next stage (Cp == 128)
  EPI_QK = 8,         // q | k projection with head-padded features (5 tiles per head): bias, per-head LayerNorm, rotary embedding,
                      // f16 operands of the attention kernel (Qp rows / K fragment tiles) -- replaces the fp32 q, k tensors + attn_prep
  EPI_VT = 9          // v projection, token rows as the A operand: V^T fragment tiles (+ the ones row) of the attention kernel
};

struct GemmParams {
  const char* a_hi;             // token rows, f16 [M][lda]
  const char* a_lo;             // remainder plane (same layout) or null
  int lda;                      // halves per row: a multiple of 32, columns >= K zero
  int M, KS;                    // rows; K steps of 32
  const char* w_hi;             // packed fragments [ntiles][KS][64][8]
  const char* w_lo;
  int ntiles;                   // tiles of 16 output features
  int Nreal;                    // valid features (a multiple of 4)
  const float* bias;            // ntiles * 16 floats in packed feature order, or null
  const float* gamma;           // EPI_RESID: LayerScale, or null
  void* out;
  long long ldo;                // elements per output row
  int gd, gh, gw, Cp, Creal;    // EPI_SCATTER: input grid, padded / real channels per parity
  int V, nreg;                  // EPI_TOKENS: patch tokens per sample, register tokens in front
  const float* pos;             // EPI_TOKENS: [V][ldo]
  const float* sub;             // EPI_PLANAR: per (sample, channel) constant subtracted from the output (ChannelDemean), or null
  const float* lnw;             // EPI_SCATTER_LN: LayerNorm weight / bias [Creal], eps; output planes out (hi) / out_lo, ldo halves per row
  const float* lnb;
  float eps;
  void* out_lo;
  // EPI_QK / EPI_VT (attention operands; layouts in amx_attention.hip)
  const float* rope;            // [T - n_prefix][hd / 2][4] = per token and rotation pair {sin 2i, sin 2i+1, cos 2i, cos 2i+1}, or null
  const float *qnw, *qnb, *knw, *knb;   // per-head LayerNorm of q / k, padded to Cp floats, or null
  float att_eps, qscale;        // LayerNorm epsilon; log2(e) / sqrt(hd)
  int T, n_prefix, heads, hd, npad, nblk_pad;   // tokens per sample, register tokens, heads, head_dim, padded tokens, key blocks
  void *Qp, *Kp, *Vt;
};

hipError_t launch_gemm(const GemmParams& p, int epi, hipStream_t st);
hipError_t launch_pack_gemm(const float* s0, const float* s1, const float* s2, int r0, int r1, int r2, int K, int mode, int Cp, int Creal,
                            int ntiles, int KS, void* hi, void* lo, hipStream_t st);
hipError_t launch_vec_place(float* dst, const float* src, int n, float fill, hipStream_t st);
hipError_t launch_rope_pairs(float* dst, const float* src, int tokens, int hd, hipStream_t st);
hipError_t launch_headpad_vec(float* dst, const float* src, int heads, int hd, int hd_pad, hipStream_t st);   // [heads][hd] -> [heads][hd_pad], zero fill
hipError_t launch_swiglu_bias(float* dst, const float* bg, const float* bx, int hidden, hipStream_t st);
hipError_t launch_ln_rows(const void* in, int in_f16, long long ldi, int C, const float* w, const float* b, float eps, int M, int rows_out,
                          int rows_in, int skip, int gelu, void* hi, void* lo, int ldo, hipStream_t st);
hipError_t launch_place_registers(const float* reg, int nreg, int E, int rows_per_b, int nb, float* tok, hipStream_t st);
hipError_t launch_colsum(const void* hi, const void* lo, int ld, int C, int nb, int rows_per_b, int nchunk, float* out, hipStream_t st);
hipError_t launch_demean(const float* colsum, int nchunk, int ld, int K, long long rows_per_b, const float* W, int Creal, const float* bias, int nb,
                         float* mean, hipStream_t st);
hipError_t launch_export_planar(const float* in, int C, long long vox, int nb, const float* sub, float* out, hipStream_t st);

// ---- tokenizer (amx_tokenizer.hip) ----------------------------------------------------------------------------------
struct TokConvParams {
  const char* x_hi;             // input activations, f16 channels-last [N][D][H][W][Cin] (hi plane)
  const char* x_lo;             // remainder plane, or null (single-f16 input: two MFMAs per product)
  int N, D, H, W, Cin;          // INPUT extent
  int Do, Ho, Wo, Cout;         // OUTPUT extent (D / stride ...)
  const char* w_hi;             // packed [Cout / 16][taps * Cin / 32][64][8]
  const char* w_lo;
  const float* bias;            // [Cout] or null
  float* raw;                   // fp32 channels-last [N][Do][Ho][Wo][Cout]: conv + bias
  float* stats;                 // per wave {sum, sumsq}[Cout]: [(n * waves_per_sample + w)][Cout][2]
};
hipError_t launch_tokconv(const TokConvParams& p, int taps, int stride, hipStream_t st);
int tokconv_slots(const TokConvParams& p);      // waves per sample (= partial statistic slots per sample)
hipError_t launch_pack_tokconv(const float* w, int Cout, int Cin, int taps, void* hi, void* lo, hipStream_t st);

struct TokStemParams {
  const float* x;               // fp32 [N][1][D][H][W]
  int N, D, H, W;
  const char* w_hi;             // packed [2][1][64][8] (32 outputs x 27 taps padded to 32)
  const char* w_lo;
  const float* bias;            // [32]
  float* stats;                 // pass 0: per wave {sum, sumsq}[32]
  const float* scale;           // pass 1: [N][32] gamma * rstd
  const float* shift;           //         [N][32] beta - mean * scale
  float slope;
  char* h_hi; char* h_lo;       // pass 1: lrelu(IN(conv)) channels-last [N][D][H][W][32] (h_lo null: no remainder plane)
  char* p_hi; char* p_lo;       //         its 2x2x2 average, [N][D/2][H/2][W/2][32]
};
hipError_t launch_tokstem(const TokStemParams& p, int pass, hipStream_t st);
int tokstem_slots(int D, int H, int W);
hipError_t launch_pack_tokstem(const float* w, void* hi, void* lo, hipStream_t st);

// per (n, c): slots -> scale = gamma * rstd, shift = beta - mean * scale (biased variance, double combine)
hipError_t launch_tok_finalize(const float* stats, int N, int slots, int C, long long count, const float* gamma, const float* beta, float eps,
                               float* scale, float* shift, hipStream_t st);
// y = lrelu(raw * scale + shift) -> hi / lo planes
hipError_t launch_tok_apply(const float* raw, int N, long long vox, int C, const float* scale, const float* shift, float slope, void* hi, void* lo,
                            hipStream_t st);
// h = lrelu(raw_a * sa + ta + raw_b * sb + tb) -> hi / lo planes, and its 2x2x2 average -> p_hi / p_lo (or null)
hipError_t launch_tok_combine(const float* raw_a, const float* raw_b, int N, int D, int H, int W, int C, const float* sa, const float* ta,
                              const float* sb, const float* tb, float slope, void* h_hi, void* h_lo, void* p_hi, void* p_lo, hipStream_t st);

}  // namespace amx
