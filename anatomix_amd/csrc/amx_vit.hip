// anatomix_amd -- the 3D ViT variant `anatomix-dev-vit` (PrimusV2-S) as ONE C-ABI forward: conv tokenizer -> EVA blocks ->
// patch decoder -> ChannelDemean, every kernel from this library (amx_tokenizer.hip, amx_gemm.hip, amx_attention.hip).
// Reference surface: anatomix/model/vit3d/architectures.py:231-260 (PrimusV2), :89-165 (_PrimusExtensions), deep_tokenizer.py:12-68,
// registry entry load_from_hf.py:25-35; the upstream blocks (dynamic-network-architectures / timm, absent from this image) are
// restated in oracle/vit_ref.py -- PARITY WITH THE UPSTREAM PACKAGE IS UNPINNED, see there.
//
// Precision plan (rel-L2 against the fp32 restatement, measured per choice in DESIGN.md section 8):
//   tokenizer      fp32 raw conv outputs, hi + lo f16 activations and weights, 3 MFMAs per product (ten InstanceNorms in a row);
//   EVA blocks     fp32 residual stream, f16 operands (LayerNorm outputs, weights, q / k / v / p), fp32 accumulate;
//   decoder        hi + lo operands (cfg.decoder_split), fp32 raw outputs, LayerNorm + GELU in fp32.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/anatomix_amd.h"
#include "amx_gemm.h"

namespace amx {
int set_error(int code, const char* msg);                                     // amx_api.hip (thread-local message)
size_t attention_scratch_bytes(int b, int heads, int n);
void attention_operands(void* scratch, int b, int heads, int n, void** Qp, void** Kp, void** Vt, int* npad_out, int* nblk_pad_out);
hipError_t launch_attention_fwd(const void* Qp, const void* Kp, const void* Vt, int b, int n, int heads, int hd, float* out, hipStream_t st);
hipError_t launch_attention_ld(const float* q, const float* k, const float* v, int ld, const float* qn_w, const float* qn_b, const float* kn_w,
                               const float* kn_b, float eps, const float* rope, int n_prefix, int b, int n, int heads, int hd, float* out,
                               void* scratch, hipStream_t st);
}  // namespace amx

namespace {

int vfail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return amx::set_error(code, buf);
}
#define VIT_HIP(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return vfail(AMX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

inline int up(int v, int m) { return (v + m - 1) / m * m; }

struct Packed {            // one packed matrix (hi [+ lo]) in the parameter arena
  char* hi = nullptr;
  char* lo = nullptr;
  int ntiles = 0, KS = 0;
};

struct Arena {
  char* base = nullptr;
  size_t off = 0;
  template <typename T = char>
  T* take(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += bytes;
    return p;
  }
};

struct Block {
  float *n1w, *n1b, *qkb, *vb, *qnw, *qnb, *knw, *knb, *anw, *anb, *projb, *g1, *n2w, *n2b, *fc1b, *mnw, *mnb, *fc2b, *g2;
  Packed qk, v, proj, fc1, fc2;       // q | k and v projections with every head padded to 80 features (5 tiles)
};
struct Stage {
  Packed c1, c2, sk;
  float *c1b, *n1w, *n1b, *c2b, *n2w, *n2b, *snw, *snb;
  int cin, cout;
};
struct DecStage {
  Packed w;
  float *bias, *lnw, *lnb;
  const float* w_raw;       // fp32 copy of the LAST stage's weight (ChannelDemean through linearity)
  int cin, cout, cp;
};

}  // namespace

struct amx_vit {
  amx_vit_cfg cfg;
  int E, Ep, hd, V, T, hidden;                 // embed dim, padded to 32, head dim, patch tokens, tokens per sample
  std::vector<std::string> names;              // parameter slots in load order
  bool loaded = false;
  char* params = nullptr;                      // one device allocation for everything packed
  size_t params_bytes = 0;
  // tokenizer
  Packed stem_w;
  float *stem_b, *stem_nw, *stem_nb;
  Stage st[3];
  Packed tokproj;
  float *tokproj_b, *pos, *regs, *rope;
  std::vector<Block> blk;
  float *fnw, *fnb;
  DecStage dec[3];
  // workspace bookkeeping of the last forward (debug reads)
  struct Named { std::string name; char* ptr; size_t bytes; };
  std::vector<Named> dbg;
};

namespace {

const int kStageC[4] = {32, 32, 64, 128};

void build_names(amx_vit* h) {
  auto& n = h->names;
  const amx_vit_cfg& c = h->cfg;
  n = {"down_projection.stem.conv.weight", "down_projection.stem.conv.bias", "down_projection.stem.norm.weight", "down_projection.stem.norm.bias"};
  for (int k = 0; k < 3; ++k) {
    const std::string p = "down_projection.stages." + std::to_string(k);
    for (const char* s : {".conv1.weight", ".conv1.bias", ".norm1.weight", ".norm1.bias", ".conv2.weight", ".conv2.bias", ".norm2.weight", ".norm2.bias",
                          ".skip.weight", ".skip_norm.weight", ".skip_norm.bias"})
      n.push_back(p + s);
  }
  n.push_back("down_projection.proj.weight");
  n.push_back("down_projection.proj.bias");
  if (c.num_register_tokens > 0) n.push_back("register_tokens");
  n.push_back("eva.pos_embed");
  for (int b = 0; b < c.depth; ++b) {
    const std::string p = "eva.blocks." + std::to_string(b);
    for (const char* s : {".norm1.weight", ".norm1.bias", ".attn.q_proj.weight", ".attn.q_proj.bias", ".attn.k_proj.weight", ".attn.v_proj.weight",
                          ".attn.v_proj.bias", ".attn.proj.weight", ".attn.proj.bias"})
      n.push_back(p + s);
    if (c.qk_norm)
      for (const char* s : {".attn.q_norm.weight", ".attn.q_norm.bias", ".attn.k_norm.weight", ".attn.k_norm.bias"}) n.push_back(p + s);
    if (c.scale_attn_inner)
      for (const char* s : {".attn.norm.weight", ".attn.norm.bias"}) n.push_back(p + s);
    if (c.layer_scale) n.push_back(p + ".gamma_1");
    for (const char* s : {".norm2.weight", ".norm2.bias", ".mlp.fc1_g.weight", ".mlp.fc1_g.bias", ".mlp.fc1_x.weight", ".mlp.fc1_x.bias", ".mlp.norm.weight",
                          ".mlp.norm.bias", ".mlp.fc2.weight", ".mlp.fc2.bias"})
      n.push_back(p + s);
    if (c.layer_scale) n.push_back(p + ".gamma_2");
  }
  n.push_back("eva.norm.weight");
  n.push_back("eva.norm.bias");
  for (int k = 0; k < 3; ++k) {
    const std::string p = "up_projection.decode." + std::to_string(k);
    if (k < 2) {
      for (const char* s : {".0.weight", ".0.bias", ".1.weight", ".1.bias"}) n.push_back(p + s);
    } else {
      n.push_back(p + ".weight");
      n.push_back(p + ".bias");
    }
  }
}

Packed take_packed(Arena& a, int ntiles, int KS, bool split) {
  Packed p;
  p.ntiles = ntiles;
  p.KS = KS;
  p.hi = a.take((size_t)ntiles * KS * 1024);
  if (split) p.lo = a.take((size_t)ntiles * KS * 1024);
  return p;
}

// Lays out the parameter arena (dry run with base == null gives the size).
void layout_params(amx_vit* h, Arena& a) {
  const amx_vit_cfg& c = h->cfg;
  const int E = h->E, hid = h->hidden;
  auto f = [&](int n) { return a.take<float>((size_t)n * sizeof(float)); };
  h->stem_w = take_packed(a, 2, 1, true);
  h->stem_b = f(32); h->stem_nw = f(32); h->stem_nb = f(32);
  for (int k = 0; k < 3; ++k) {
    Stage& s = h->st[k];
    s.cin = kStageC[k]; s.cout = kStageC[k + 1];
    s.c1 = take_packed(a, s.cout / 16, 27 * s.cin / 32, true);
    s.c2 = take_packed(a, s.cout / 16, 27 * s.cout / 32, true);
    s.sk = take_packed(a, s.cout / 16, s.cin / 32, true);
    s.c1b = f(s.cout); s.n1w = f(s.cout); s.n1b = f(s.cout); s.c2b = f(s.cout); s.n2w = f(s.cout); s.n2b = f(s.cout); s.snw = f(s.cout); s.snb = f(s.cout);
  }
  h->tokproj = take_packed(a, up(E, 16) / 16, kStageC[3] / 32, true);
  h->tokproj_b = f(up(E, 16));
  h->pos = f(h->V * E);
  h->regs = f(std::max(1, c.num_register_tokens) * E);
  h->rope = f(h->V * 2 * h->hd);
  h->blk.resize(c.depth);
  const int KSe = h->Ep / 32, KSh = up(hid, 32) / 32;
  for (auto& b : h->blk) {
    b.qk = take_packed(a, 2 * c.heads * 5, KSe, false);
    b.v = take_packed(a, c.heads * 5, KSe, false);
    b.proj = take_packed(a, up(E, 16) / 16, KSe, false);
    b.fc1 = take_packed(a, 2 * up(hid, 16) / 16, KSe, false);      // hidden rounded up to whole (gate, value) tile pairs, zero weights behind it
    b.fc2 = take_packed(a, up(E, 16) / 16, KSh, false);
    b.n1w = f(E); b.n1b = f(E); b.qkb = f(2 * c.heads * 80); b.vb = f(c.heads * 80); b.qnw = f(80); b.qnb = f(80); b.knw = f(80); b.knb = f(80);
    b.anw = f(E); b.anb = f(E); b.projb = f(up(E, 16)); b.g1 = f(up(E, 16)); b.n2w = f(E); b.n2b = f(E); b.fc1b = f(2 * up(hid, 16));
    b.mnw = f(hid); b.mnb = f(hid); b.fc2b = f(up(E, 16)); b.g2 = f(up(E, 16));
  }
  h->fnw = f(E); h->fnb = f(E);
  const int dc[4] = {E, c.dec1, c.dec2, c.num_classes};
  for (int k = 0; k < 3; ++k) {
    DecStage& d = h->dec[k];
    d.cin = dc[k]; d.cout = dc[k + 1]; d.cp = up(d.cout, 16);
    d.w = take_packed(a, 8 * d.cp / 16, up(d.cin, 32) / 32, c.decoder_split != 0);
    d.bias = f(d.cp);
    d.lnw = f(d.cout); d.lnb = f(d.cout);
    d.w_raw = k == 2 ? f(d.cin * d.cout * 8) : nullptr;
  }
}

}  // namespace

extern "C" {

int amx_vit_create(amx_vit_t** out, const amx_vit_cfg* cfg) {
  if (!out || !cfg) return vfail(AMX_ERR_INVALID, "null argument");
  *out = nullptr;
  const amx_vit_cfg& c = *cfg;
  if (c.input_channels != 1) return vfail(AMX_ERR_INVALID, "vit: input_channels must be 1 (got %d)", c.input_channels);
  if (c.embed_dim < 32 || c.embed_dim % 4 || c.heads < 1 || c.embed_dim % c.heads) return vfail(AMX_ERR_INVALID, "vit: embed_dim %d / heads %d", c.embed_dim, c.heads);
  const int hd = c.embed_dim / c.heads;
  if ((hd & 1) || hd > 80) return vfail(AMX_ERR_INVALID, "vit: head_dim must be even and <= 80 (got %d)", hd);
  if (c.depth < 1 || c.num_register_tokens < 0) return vfail(AMX_ERR_INVALID, "vit: depth / register tokens");
  if (c.grid_d < 2 || c.grid_h < 2 || c.grid_w < 2 || (c.grid_w % 2) || ((long long)c.grid_d * c.grid_h * c.grid_w) % 64 || (c.grid_w * 8) % 16)
    return vfail(AMX_ERR_SHAPE, "vit: token grid %dx%dx%d (even width, a multiple of 64 tokens)", c.grid_d, c.grid_h, c.grid_w);
  if (c.hidden < 16 || c.hidden > 3072 || ((c.hidden % 4) && c.hidden > 1088))
    return vfail(AMX_ERR_INVALID, "vit: SwiGLU hidden width %d must be 16 .. 3072 (a multiple of 4 above 1088)", c.hidden);
  if (c.embed_dim > 1280) return vfail(AMX_ERR_INVALID, "vit: embed_dim <= 1280");
  if (c.dec1 % 4 || c.dec2 % 4 || c.num_classes % 4 || c.dec1 < 4 || c.dec2 < 4 || c.num_classes < 4 || c.dec1 > 1024 || c.dec2 > 256 || c.num_classes > 128)
    return vfail(AMX_ERR_INVALID, "vit: decoder widths %d / %d / %d must be multiples of 4 (<= 1024 / 256, classes <= 128)", c.dec1, c.dec2, c.num_classes);
  if (c.out_norm != 0 && c.out_norm != 1) return vfail(AMX_ERR_INVALID, "vit: out_norm 0 (none) or 1 (demean)");
  amx_vit* h = new amx_vit();
  h->cfg = c;
  h->E = c.embed_dim; h->Ep = up(c.embed_dim, 32); h->hd = hd; h->hidden = c.hidden;
  h->V = c.grid_d * c.grid_h * c.grid_w;
  h->T = h->V + c.num_register_tokens;
  build_names(h);
  Arena dry;
  layout_params(h, dry);
  h->params_bytes = dry.off + 256;
  hipError_t e = hipMalloc((void**)&h->params, h->params_bytes);
  if (e != hipSuccess) { delete h; return vfail(AMX_ERR_HIP, "hipMalloc(%zu): %s", dry.off, hipGetErrorString(e)); }
  Arena real;
  real.base = h->params;
  layout_params(h, real);
  *out = h;
  return AMX_OK;
}

void amx_vit_destroy(amx_vit_t* h) {
  if (!h) return;
  if (h->params) (void)hipFree(h->params);
  delete h;
}

int amx_vit_num_params(const amx_vit_t* h) { return h ? (int)h->names.size() : 0; }
const char* amx_vit_param_name(const amx_vit_t* h, int idx) { return h && idx >= 0 && idx < (int)h->names.size() ? h->names[idx].c_str() : nullptr; }

int amx_vit_load(amx_vit_t* h, const float* const* d_params, int count, const float* d_rope, void* stream) {
  if (!h || !d_params || !d_rope) return vfail(AMX_ERR_INVALID, "null argument");
  if (count != (int)h->names.size()) return vfail(AMX_ERR_INVALID, "vit: expected %d parameter tensors, got %d", (int)h->names.size(), count);
  for (int i = 0; i < count; ++i)
    if (!d_params[i]) return vfail(AMX_ERR_INVALID, "vit: parameter %s is null", h->names[i].c_str());
  hipStream_t st = (hipStream_t)stream;
  const amx_vit_cfg& c = h->cfg;
  const int E = h->E, hid = h->hidden;
  int i = 0;
  auto next = [&]() { return d_params[i++]; };
  auto vec = [&](float* dst, const float* src, int n, int npad = 0) -> hipError_t {
    hipError_t e = amx::launch_vec_place(dst, src, n, 0.f, st);
    if (e == hipSuccess && npad > n) e = amx::launch_vec_place(dst + n, nullptr, npad - n, 0.f, st);
    return e;
  };
  VIT_HIP(hipMemsetAsync(h->params, 0, h->params_bytes, st));
  VIT_HIP(amx::launch_pack_tokstem(next(), h->stem_w.hi, h->stem_w.lo, st));
  VIT_HIP(vec(h->stem_b, next(), 32));
  VIT_HIP(vec(h->stem_nw, next(), 32));
  VIT_HIP(vec(h->stem_nb, next(), 32));
  for (int k = 0; k < 3; ++k) {
    Stage& s = h->st[k];
    VIT_HIP(amx::launch_pack_tokconv(next(), s.cout, s.cin, 27, s.c1.hi, s.c1.lo, st));
    VIT_HIP(vec(s.c1b, next(), s.cout));
    VIT_HIP(vec(s.n1w, next(), s.cout));
    VIT_HIP(vec(s.n1b, next(), s.cout));
    VIT_HIP(amx::launch_pack_tokconv(next(), s.cout, s.cout, 27, s.c2.hi, s.c2.lo, st));
    VIT_HIP(vec(s.c2b, next(), s.cout));
    VIT_HIP(vec(s.n2w, next(), s.cout));
    VIT_HIP(vec(s.n2b, next(), s.cout));
    VIT_HIP(amx::launch_pack_tokconv(next(), s.cout, s.cin, 1, s.sk.hi, s.sk.lo, st));
    VIT_HIP(vec(s.snw, next(), s.cout));
    VIT_HIP(vec(s.snb, next(), s.cout));
  }
  VIT_HIP(amx::launch_pack_gemm(next(), nullptr, nullptr, E, 0, 0, kStageC[3], 0, 0, 0, h->tokproj.ntiles, h->tokproj.KS, h->tokproj.hi, h->tokproj.lo, st));
  VIT_HIP(vec(h->tokproj_b, next(), E, up(E, 16)));
  if (c.num_register_tokens > 0) VIT_HIP(vec(h->regs, next(), c.num_register_tokens * E));
  VIT_HIP(vec(h->pos, next(), h->V * E));
  VIT_HIP(amx::launch_rope_pairs(h->rope, d_rope, h->V, h->hd, st));
  for (auto& b : h->blk) {
    VIT_HIP(vec(b.n1w, next(), E));
    VIT_HIP(vec(b.n1b, next(), E));
    const float *qw = next(), *qb = next(), *kw = next(), *vw = next(), *vb = next();
    VIT_HIP(amx::launch_pack_gemm(qw, kw, nullptr, E, 0, 0, E, 4, 80, h->hd, b.qk.ntiles, b.qk.KS, b.qk.hi, nullptr, st));
    VIT_HIP(amx::launch_pack_gemm(vw, nullptr, nullptr, E, 0, 0, E, 4, 80, h->hd, b.v.ntiles, b.v.KS, b.v.hi, nullptr, st));
    VIT_HIP(amx::launch_headpad_vec(b.qkb, qb, c.heads, h->hd, 80, st));
    VIT_HIP(amx::launch_headpad_vec(b.qkb + c.heads * 80, nullptr, c.heads, h->hd, 80, st));     // EVA: the key projection has no bias
    VIT_HIP(amx::launch_headpad_vec(b.vb, vb, c.heads, h->hd, 80, st));
    VIT_HIP(amx::launch_pack_gemm(next(), nullptr, nullptr, E, 0, 0, E, 0, 0, 0, b.proj.ntiles, b.proj.KS, b.proj.hi, nullptr, st));
    VIT_HIP(vec(b.projb, next(), E, up(E, 16)));
    if (c.qk_norm) {
      VIT_HIP(vec(b.qnw, next(), h->hd, 80));
      VIT_HIP(vec(b.qnb, next(), h->hd, 80));
      VIT_HIP(vec(b.knw, next(), h->hd, 80));
      VIT_HIP(vec(b.knb, next(), h->hd, 80));
    }
    if (c.scale_attn_inner) {
      VIT_HIP(vec(b.anw, next(), E));
      VIT_HIP(vec(b.anb, next(), E));
    }
    if (c.layer_scale) VIT_HIP(vec(b.g1, next(), E, up(E, 16)));
    else VIT_HIP(amx::launch_vec_place(b.g1, nullptr, up(E, 16), 1.f, st));
    VIT_HIP(vec(b.n2w, next(), E));
    VIT_HIP(vec(b.n2b, next(), E));
    const float *gw = next(), *gb = next(), *xw = next(), *xb = next();
    VIT_HIP(amx::launch_pack_gemm(gw, xw, nullptr, hid, hid, 0, E, 1, 0, 0, b.fc1.ntiles, b.fc1.KS, b.fc1.hi, nullptr, st));
    VIT_HIP(amx::launch_swiglu_bias(b.fc1b, gb, xb, hid, st));
    VIT_HIP(vec(b.mnw, next(), hid));
    VIT_HIP(vec(b.mnb, next(), hid));
    VIT_HIP(amx::launch_pack_gemm(next(), nullptr, nullptr, E, 0, 0, hid, 0, 0, 0, b.fc2.ntiles, b.fc2.KS, b.fc2.hi, nullptr, st));
    VIT_HIP(vec(b.fc2b, next(), E, up(E, 16)));
    if (c.layer_scale) VIT_HIP(vec(b.g2, next(), E, up(E, 16)));
    else VIT_HIP(amx::launch_vec_place(b.g2, nullptr, up(E, 16), 1.f, st));
  }
  VIT_HIP(vec(h->fnw, next(), E));
  VIT_HIP(vec(h->fnb, next(), E));
  for (int k = 0; k < 3; ++k) {
    DecStage& d = h->dec[k];
    const float* w = next();
    VIT_HIP(amx::launch_pack_gemm(w, nullptr, nullptr, 0, 0, 0, d.cin, k == 2 ? 3 : 2, d.cp, d.cout, d.w.ntiles, d.w.KS, d.w.hi, d.w.lo, st));
    if (d.w_raw) VIT_HIP(vec((float*)d.w_raw, w, d.cin * d.cout * 8));
    VIT_HIP(vec(d.bias, next(), d.cout, d.cp));
    if (k < 2) {
      VIT_HIP(vec(d.lnw, next(), d.cout));
      VIT_HIP(vec(d.lnb, next(), d.cout));
    }
  }
  if (i != count) return vfail(AMX_ERR_INVALID, "vit: internal parameter count mismatch (%d of %d consumed)", i, count);
  h->loaded = true;
  return AMX_OK;
}

}  // extern "C"

namespace {

// Workspace plan of one forward at batch n.  Regions: the token stream (lives across phases) + max(tokenizer, blocks, decoder).
struct Plan {
  // common
  float* tok;
  // tokenizer
  float *stats, *sc[4], *sh[4];            // statistic slots; (scale, shift) sets: 0 stem / conv1, 1 conv2, 2 skip
  char *h_hi[4], *h_lo[4], *p_hi[3], *p_lo[3];
  float *raw1, *raw2, *raws;
  char *a1_hi, *a1_lo;
  // blocks
  char *A1, *A2, *Hd, *att;
  float *AO;
  // decoder
  char *Ad_hi[3], *Ad_lo[3];
  float *rawd[3], *colsum, *mean;
  size_t total;
};

constexpr int kColChunks = 128;

Plan make_plan(const amx_vit* h, int n, char* base) {
  Plan P{};
  const amx_vit_cfg& c = h->cfg;
  const int E = h->E, Ep = h->Ep, T = h->T, V = h->V, hid = h->hidden;
  const long long M = (long long)n * T;
  Arena a;
  a.base = base;
  P.tok = a.take<float>((size_t)M * E * 4);
  const size_t common = a.off;
  // ---- tokenizer
  const long long v0 = (long long)V * 512;                 // voxels per sample at full resolution (8^3 per token)
  P.stats = a.take<float>((size_t)n * (size_t)(v0 / 256 + 64) * 128 * 2 * 4);   // upper bound: <= v0 / 256 slots per sample (32 voxels per wave at 1/8 of the input), <= 128 channels
  for (int k = 0; k < 3; ++k) { P.sc[k] = a.take<float>((size_t)n * 128 * 4); P.sh[k] = a.take<float>((size_t)n * 128 * 4); }
  long long vk = v0;
  for (int k = 0; k < 4; ++k) {
    P.h_hi[k] = a.take((size_t)n * vk * kStageC[k] * 2);
    P.h_lo[k] = a.take((size_t)n * vk * kStageC[k] * 2);
    if (k < 3) {
      P.p_hi[k] = a.take((size_t)n * (vk / 8) * kStageC[k] * 2);
      P.p_lo[k] = a.take((size_t)n * (vk / 8) * kStageC[k] * 2);
    }
    vk /= 8;
  }
  const size_t rawb = (size_t)n * (v0 / 8) * 32 * 4;      // the largest raw tensor: stage 0 output (64^3 x 32); later stages are half that
  P.raw1 = a.take<float>(rawb);
  P.raw2 = a.take<float>(rawb);
  P.raws = a.take<float>(rawb);
  P.a1_hi = a.take(rawb / 2);
  P.a1_lo = a.take(rawb / 2);
  const size_t tok_end = a.off;
  // ---- blocks (alias the tokenizer region)
  a.off = common;
  const long long Mp = (M + 255) / 256 * 256;
  P.A1 = a.take((size_t)Mp * Ep * 2);
  P.A2 = a.take((size_t)Mp * up(hid, 32) * 2);
  P.Hd = a.take((size_t)Mp * up(hid, 16) * 2);
  P.AO = a.take<float>((size_t)M * E * 4);
  P.att = a.take(amx::attention_scratch_bytes(n, c.heads, T));
  const size_t blk_end = a.off;
  // ---- decoder (aliases both)
  a.off = common;
  long long rows = (long long)n * V;
  const int dk[3] = {Ep, up(c.dec1, 32), up(c.dec2, 32)};
  for (int k = 0; k < 3; ++k) {
    P.Ad_hi[k] = a.take((size_t)rows * dk[k] * 2);
    P.Ad_lo[k] = c.decoder_split ? a.take((size_t)rows * dk[k] * 2) : nullptr;
    rows *= 8;
    P.rawd[k] = k < 2 ? a.take<float>((size_t)rows * h->dec[k].cout * 4) : nullptr;     // the last stage writes the planar output itself
  }
  P.colsum = a.take<float>((size_t)n * kColChunks * 256 * 4);
  P.mean = a.take<float>((size_t)n * 128 * 4);
  const size_t dec_end = a.off;
  P.total = std::max(tok_end, std::max(blk_end, dec_end)) + 256;
  return P;
}

}  // namespace

extern "C" {

size_t amx_vit_workspace_bytes(const amx_vit_t* h, int n) {
  if (!h || n < 1) return 0;
  return make_plan(h, n, nullptr).total;
}

int amx_vit_forward(amx_vit_t* h, const float* d_x, float* d_y, int n, void* d_ws, size_t ws_bytes, int n_blocks, void* stream) {
  if (!h || !d_x || !d_y || !d_ws) return vfail(AMX_ERR_INVALID, "null argument");
  if (!h->loaded) return vfail(AMX_ERR_NOT_LOADED, "vit: forward before amx_vit_load");
  if (n < 1) return vfail(AMX_ERR_INVALID, "vit: batch %d", n);
  if ((uintptr_t)d_ws & 255) return vfail(AMX_ERR_WORKSPACE, "vit: workspace must be 256-byte aligned");
  const Plan P = make_plan(h, n, (char*)d_ws);
  if (ws_bytes < P.total) return vfail(AMX_ERR_WORKSPACE, "vit: workspace needs %zu bytes (got %zu)", P.total, ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  const amx_vit_cfg& c = h->cfg;
  const int E = h->E, Ep = h->Ep, T = h->T, V = h->V, hid = h->hidden, nreg = c.num_register_tokens;
  const int M = n * T;
  const int D0 = c.grid_d * 8, H0 = c.grid_h * 8, W0 = c.grid_w * 8;
#ifdef AMX_VIT_DEBUG_STOP    // debugging aid of tools/vit_bisect.py (build with -DAMX_VIT_DEBUG_STOP): return after a stage.  Compiled OUT of the product.
  static const int dbg_stop = exp_env("AMX_VIT_STOP") ? atoi(exp_env("AMX_VIT_STOP")) : 0;
#else
  constexpr int dbg_stop = 0;
#endif
  h->dbg.clear();
  auto note = [&](const char* name, const void* p, size_t bytes) { h->dbg.push_back({name, (char*)p, bytes}); };

  // ------------------------------------------------------------------ tokenizer
  {
    amx::TokStemParams sp{};
    sp.x = d_x; sp.N = n; sp.D = D0; sp.H = H0; sp.W = W0;
    sp.w_hi = h->stem_w.hi; sp.w_lo = h->stem_w.lo; sp.bias = h->stem_b; sp.stats = P.stats;
    sp.scale = P.sc[0]; sp.shift = P.sh[0]; sp.slope = 0.01f;
    sp.h_hi = P.h_hi[0]; sp.h_lo = c.stem_split ? P.h_lo[0] : nullptr; sp.p_hi = P.p_hi[0]; sp.p_lo = P.p_lo[0];
    VIT_HIP(amx::launch_tokstem(sp, 0, st));
    VIT_HIP(amx::launch_tok_finalize(P.stats, n, amx::tokstem_slots(D0, H0, W0), 32, (long long)D0 * H0 * W0, h->stem_nw, h->stem_nb, c.in_eps, P.sc[0], P.sh[0], st));
    VIT_HIP(amx::launch_tokstem(sp, 1, st));
    note("h0_hi", P.h_hi[0], (size_t)n * D0 * H0 * W0 * 32 * 2);
    note("p0_hi", P.p_hi[0], (size_t)n * D0 * H0 * W0 * 4 * 2);
  }
  if (dbg_stop == 10) return AMX_OK;
  int Dk = D0, Hk = H0, Wk = W0;
  for (int k = 0; k < 3; ++k) {
    const Stage& s = h->st[k];
    const int Do = Dk / 2, Ho = Hk / 2, Wo = Wk / 2;
    const long long vo = (long long)Do * Ho * Wo;
    amx::TokConvParams cp{};
    cp.N = n; cp.Do = Do; cp.Ho = Ho; cp.Wo = Wo; cp.Cout = s.cout; cp.stats = P.stats;
    // conv1: stride 2 from the stage input
    cp.x_hi = P.h_hi[k]; cp.x_lo = (k == 0 && !c.stem_split) ? nullptr : P.h_lo[k]; cp.D = Dk; cp.H = Hk; cp.W = Wk; cp.Cin = s.cin;
    cp.w_hi = s.c1.hi; cp.w_lo = s.c1.lo; cp.bias = s.c1b; cp.raw = P.raw1;
    VIT_HIP(amx::launch_tokconv(cp, 27, 2, st));
    VIT_HIP(amx::launch_tok_finalize(P.stats, n, amx::tokconv_slots(cp), s.cout, vo, s.n1w, s.n1b, c.in_eps, P.sc[0], P.sh[0], st));
    VIT_HIP(amx::launch_tok_apply(P.raw1, n, vo, s.cout, P.sc[0], P.sh[0], 0.01f, P.a1_hi, P.a1_lo, st));
    // conv2
    cp.x_hi = P.a1_hi; cp.x_lo = P.a1_lo; cp.D = Do; cp.H = Ho; cp.W = Wo; cp.Cin = s.cout;
    cp.w_hi = s.c2.hi; cp.w_lo = s.c2.lo; cp.bias = s.c2b; cp.raw = P.raw2;
    VIT_HIP(amx::launch_tokconv(cp, 27, 1, st));
    VIT_HIP(amx::launch_tok_finalize(P.stats, n, amx::tokconv_slots(cp), s.cout, vo, s.n2w, s.n2b, c.in_eps, P.sc[1], P.sh[1], st));
    // skip: 1x1x1 conv of the average-pooled stage input (no bias)
    cp.x_hi = P.p_hi[k]; cp.x_lo = P.p_lo[k]; cp.Cin = s.cin;
    cp.w_hi = s.sk.hi; cp.w_lo = s.sk.lo; cp.bias = nullptr; cp.raw = P.raws;
    VIT_HIP(amx::launch_tokconv(cp, 1, 1, st));
    VIT_HIP(amx::launch_tok_finalize(P.stats, n, amx::tokconv_slots(cp), s.cout, vo, s.snw, s.snb, c.in_eps, P.sc[2], P.sh[2], st));
    VIT_HIP(amx::launch_tok_combine(P.raw2, P.raws, n, Do, Ho, Wo, s.cout, P.sc[1], P.sh[1], P.sc[2], P.sh[2], 0.01f, P.h_hi[k + 1], P.h_lo[k + 1],
                                    k < 2 ? P.p_hi[k + 1] : nullptr, k < 2 ? P.p_lo[k + 1] : nullptr, st));
    if (k == 0) { note("raw1_s0", P.raw1, (size_t)n * vo * s.cout * 4); }
    Dk = Do; Hk = Ho; Wk = Wo;
    if (dbg_stop == 11 + k) return AMX_OK;
  }
  {
    amx::GemmParams g{};
    g.a_hi = P.h_hi[3]; g.a_lo = P.h_lo[3]; g.lda = kStageC[3]; g.M = n * V; g.KS = h->tokproj.KS;
    g.w_hi = h->tokproj.hi; g.w_lo = h->tokproj.lo; g.ntiles = h->tokproj.ntiles; g.Nreal = E; g.bias = h->tokproj_b;
    g.out = P.tok; g.ldo = E; g.V = V; g.nreg = nreg; g.pos = h->pos;
    VIT_HIP(amx::launch_gemm(g, amx::EPI_TOKENS, st));
    if (dbg_stop == 14) return AMX_OK;
    VIT_HIP(amx::launch_place_registers(h->regs, nreg, E, T, n, P.tok, st));
  }
  note("tokens", P.tok, (size_t)M * E * 4);
  if (dbg_stop == 1) return AMX_OK;

  // ------------------------------------------------------------------ EVA blocks
  const int nb = n_blocks < 0 || n_blocks > c.depth ? c.depth : n_blocks;
  void *Qp, *Kp, *Vt;
  int att_npad, att_nblk;
  amx::attention_operands(P.att, n, c.heads, T, &Qp, &Kp, &Vt, &att_npad, &att_nblk);
  // padding of the operand buffers (token rows / keys beyond T, feature columns beyond head_dim) is zero and never written again
  if (nb > 0) VIT_HIP(hipMemsetAsync(P.att, 0, amx::attention_scratch_bytes(n, c.heads, T), st));
  for (int bi = 0; bi < nb; ++bi) {
    const Block& b = h->blk[bi];
    amx::GemmParams g{};
    g.a_hi = P.A1; g.lda = Ep; g.M = M;
    VIT_HIP(amx::launch_ln_rows(P.tok, 0, E, E, b.n1w, b.n1b, 1e-6f, M, M, M, 0, 0, P.A1, nullptr, Ep, st));
    // q | k and v projections write the attention kernel's f16 operands themselves (bias, per-head LayerNorm, rotary embedding,
    // fragment layouts in the epilogue): no fp32 q / k / v tensors, no separate preparation pass
    g.att_eps = 1e-5f; g.qscale = 1.4426950408889634f / sqrtf((float)h->hd); g.rope = h->rope; g.T = T; g.n_prefix = nreg; g.heads = c.heads;
    g.hd = h->hd; g.npad = att_npad; g.nblk_pad = att_nblk; g.Qp = Qp; g.Kp = Kp; g.Vt = Vt; g.Cp = 80;
    g.qnw = c.qk_norm ? b.qnw : nullptr; g.qnb = b.qnb; g.knw = c.qk_norm ? b.knw : nullptr; g.knb = b.knb;
    g.KS = b.qk.KS; g.w_hi = b.qk.hi; g.ntiles = b.qk.ntiles; g.Nreal = b.qk.ntiles * 16; g.bias = b.qkb; g.out = Qp; g.ldo = 0;
    VIT_HIP(amx::launch_gemm(g, amx::EPI_QK, st));
    g.KS = b.v.KS; g.w_hi = b.v.hi; g.ntiles = b.v.ntiles; g.Nreal = b.v.ntiles * 16; g.bias = b.vb; g.out = Vt;
    VIT_HIP(amx::launch_gemm(g, amx::EPI_VT, st));
    VIT_HIP(amx::launch_attention_fwd(Qp, Kp, Vt, n, T, c.heads, h->hd, P.AO, st));
    VIT_HIP(amx::launch_ln_rows(P.AO, 0, E, E, c.scale_attn_inner ? b.anw : nullptr, b.anb, 1e-5f, M, M, M, 0, 0, P.A1, nullptr, Ep, st));
    g.KS = b.proj.KS; g.w_hi = b.proj.hi; g.ntiles = b.proj.ntiles; g.Nreal = E; g.bias = b.projb; g.gamma = b.g1; g.out = P.tok; g.ldo = E;
    VIT_HIP(amx::launch_gemm(g, amx::EPI_RESID, st));
    VIT_HIP(amx::launch_ln_rows(P.tok, 0, E, E, b.n2w, b.n2b, 1e-6f, M, M, M, 0, 0, P.A1, nullptr, Ep, st));
    g.KS = b.fc1.KS; g.w_hi = b.fc1.hi; g.ntiles = b.fc1.ntiles; g.Nreal = 2 * up(hid, 16); g.bias = b.fc1b; g.gamma = nullptr; g.out = P.Hd; g.ldo = up(hid, 16);
    VIT_HIP(amx::launch_gemm(g, amx::EPI_SWIGLU, st));
    VIT_HIP(amx::launch_ln_rows(P.Hd, 1, up(hid, 16), hid, b.mnw, b.mnb, 1e-6f, M, M, M, 0, 0, P.A2, nullptr, up(hid, 32), st));
    g.a_hi = P.A2; g.lda = up(hid, 32);
    g.KS = b.fc2.KS; g.w_hi = b.fc2.hi; g.ntiles = b.fc2.ntiles; g.Nreal = E; g.bias = b.fc2b; g.gamma = b.g2; g.out = P.tok; g.ldo = E;
    VIT_HIP(amx::launch_gemm(g, amx::EPI_RESID, st));
  }
  note("tokens_out", P.tok, (size_t)M * E * 4);

  // ------------------------------------------------------------------ final norm (drops the register tokens) + decoder
  const int Mv = n * V;
  VIT_HIP(amx::launch_ln_rows(P.tok, 0, E, E, h->fnw, h->fnb, 1e-6f, Mv, V, T, nreg, 0, P.Ad_hi[0], P.Ad_lo[0], Ep, st));
  if (dbg_stop == 2) return AMX_OK;
  int gd = c.grid_d, gh = c.grid_h, gw = c.grid_w;
  long long rows = Mv;
  for (int k = 0; k < 3; ++k) {
    const DecStage& d = h->dec[k];
    amx::GemmParams g{};
    g.a_hi = P.Ad_hi[k]; g.a_lo = P.Ad_lo[k]; g.lda = up(d.cin, 32); g.M = (int)rows; g.KS = d.w.KS;
    g.w_hi = d.w.hi; g.w_lo = d.w.lo; g.ntiles = d.w.ntiles; g.Nreal = 8 * d.cp; g.bias = d.bias;
    g.out = P.rawd[k]; g.ldo = d.cout; g.gd = gd; g.gh = gh; g.gw = gw; g.Cp = d.cp; g.Creal = d.cout;
    if (k == 2 && c.out_norm == 1) {      // ChannelDemean: the mean of every output channel follows from the column means of this stage's input
      VIT_HIP(amx::launch_colsum(P.Ad_hi[2], P.Ad_lo[2], up(d.cin, 32), d.cin, n, (int)(rows / n), kColChunks, P.colsum, st));
      VIT_HIP(amx::launch_demean(P.colsum, kColChunks, up(d.cin, 32), d.cin, rows / n, d.w_raw, d.cout, d.bias, n, P.mean, st));
    }
    if (k == 2) {                         // planar fp32 output written by the product kernel itself
      g.out = d_y; g.sub = c.out_norm == 1 ? P.mean : nullptr;
      VIT_HIP(amx::launch_gemm(g, amx::EPI_PLANAR, st));
    } else if (d.cp == 128 && up(d.cout, 32) <= 128) {   // channel LayerNorm + GELU fused: the next stage's operand rows come straight out
      g.out = P.Ad_hi[k + 1]; g.out_lo = P.Ad_lo[k + 1]; g.ldo = up(d.cout, 32); g.lnw = d.lnw; g.lnb = d.lnb; g.eps = 1e-6f;
      VIT_HIP(amx::launch_gemm(g, amx::EPI_SCATTER_LN, st));
    } else {
      VIT_HIP(amx::launch_gemm(g, amx::EPI_SCATTER, st));
      VIT_HIP(amx::launch_ln_rows(P.rawd[k], 0, d.cout, d.cout, d.lnw, d.lnb, 1e-6f, (int)(rows * 8), (int)(rows * 8), (int)(rows * 8), 0, 1, P.Ad_hi[k + 1],
                                  P.Ad_lo[k + 1], up(d.cout, 32), st));
    }
    rows *= 8; gd *= 2; gh *= 2; gw *= 2;
    if (dbg_stop == 3 + k) return AMX_OK;
  }
  return AMX_OK;
}

/* Copies a workspace buffer of the LAST forward on this handle (test / debugging aid): names "tokens" (after the tokenizer:
 * fp32 [n][T][E]), "tokens_out" (after the blocks), "h0_hi", "p0_hi", "raw1_s0". */
int amx_vit_debug_read(amx_vit_t* h, const char* name, void* d_dst, size_t max_bytes, size_t* bytes, void* stream) {
  if (!h || !name) return vfail(AMX_ERR_INVALID, "null argument");
  for (const auto& e : h->dbg)
    if (e.name == name) {
      if (bytes) *bytes = e.bytes;
      if (d_dst) {
        if (max_bytes < e.bytes) return vfail(AMX_ERR_WORKSPACE, "debug_read %s: %zu bytes needed", name, e.bytes);
        VIT_HIP(hipMemcpyAsync(d_dst, e.ptr, e.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
      }
      return AMX_OK;
    }
  return vfail(AMX_ERR_INVALID, "debug_read: no buffer named %s", name);
}

int amx_linear(const float* d_x, const float* d_w, const float* d_b, int m, int k, int n, int split, float* d_y, void* stream) {
  if (!d_x || !d_w || !d_y) return vfail(AMX_ERR_INVALID, "null argument");
  if (m < 1 || k < 1 || n < 4 || (n % 4) || k > 64 * 17) return vfail(AMX_ERR_INVALID, "linear: m >= 1, 1 <= k <= 1088, n a multiple of 4 (got %d %d %d)", m, k, n);
  hipStream_t st = (hipStream_t)stream;
  const int kp = up(k, 32), ntiles = up(n, 16) / 16, KS = kp / 32;
  const size_t abytes = (size_t)m * kp * 2, wbytes = (size_t)ntiles * KS * 1024, bbytes = (size_t)ntiles * 16 * 4;
  char* buf = nullptr;
  VIT_HIP(hipMalloc((void**)&buf, 2 * abytes + 2 * wbytes + bbytes + 1024));
  char *a_hi = buf, *a_lo = buf + abytes, *w_hi = buf + 2 * abytes, *w_lo = w_hi + wbytes;
  float* bias = (float*)(w_lo + wbytes);
  int rc = AMX_OK;
  hipError_t e = amx::launch_ln_rows(d_x, 0, k, k, nullptr, nullptr, 0.f, m, m, m, 0, 0, a_hi, split ? a_lo : nullptr, kp, st);
  if (e == hipSuccess) e = amx::launch_pack_gemm(d_w, nullptr, nullptr, n, 0, 0, k, 0, 0, 0, ntiles, KS, w_hi, split ? w_lo : nullptr, st);
  if (e == hipSuccess) e = amx::launch_vec_place(bias, d_b, n, 0.f, st);
  if (e == hipSuccess && ntiles * 16 > n) e = amx::launch_vec_place(bias + n, nullptr, ntiles * 16 - n, 0.f, st);
  if (e == hipSuccess) {
    amx::GemmParams g{};
    g.a_hi = a_hi; g.a_lo = split ? a_lo : nullptr; g.lda = kp; g.M = m; g.KS = KS; g.w_hi = w_hi; g.w_lo = split ? w_lo : nullptr;
    g.ntiles = ntiles; g.Nreal = n; g.bias = bias; g.out = d_y; g.ldo = n;
    e = amx::launch_gemm(g, amx::EPI_F32, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) rc = vfail(AMX_ERR_HIP, "linear: %s", hipGetErrorString(e));
  (void)hipFree(buf);
  return rc;
}

}  // extern "C"
