"""Where the ViT forward spends its time: tokenizer / per-block linears / attention core / decoder (events on the stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anatomix_amd.model.load_from_hf import build_variant
from oracle import vit_ref as V
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = build_variant("anatomix-dev-vit"); m.load_state_dict(V.synthetic_state_dict(V.VIT_VARIANTS["anatomix-dev-vit"], 0)); m = m.to(dev).eval()
x = V.synthetic_input(1, B).to(dev)
def timed(fn, reps=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out
with torch.no_grad():
    t_tok, feat = timed(lambda: m.down_projection(x))
    tok = torch.cat((m.register_tokens.expand(B, -1, -1), feat.flatten(2).transpose(1, 2) + m.eva.pos_embed), 1)
    blk = m.eva.blocks[0]
    t_blk, _ = timed(lambda: blk(tok, m.rope_table, 8))
    h = blk.norm1(tok)
    q, k, v = blk.attn.q_proj(h), blk.attn.k_proj(h), blk.attn.v_proj(h)
    t_core, _ = timed(lambda: blk.attn.core_hip(q, k, v, m.rope_table, 8))
    t_core_torch, _ = timed(lambda: blk.attn.core_torch(q, k, v, m.rope_table, 8))
    t_qkv, _ = timed(lambda: (blk.attn.q_proj(h), blk.attn.k_proj(h), blk.attn.v_proj(h)))
    t_mlp, _ = timed(lambda: blk.mlp(blk.norm2(tok)))
    t_ln, _ = timed(lambda: blk.norm1(tok))
    vol = tok[:, 8:].transpose(1, 2).reshape(B, 396, 16, 16, 16).contiguous()
    t_dec, _ = timed(lambda: m.up_projection(vol))
    t_all, _ = timed(lambda: m(x))
print(f"B={B}: total {t_all:.2f} ms | tokenizer {t_tok:.2f} | one block {t_blk:.3f} (x12 = {12*t_blk:.2f}): attention core {t_core:.3f} (same core as torch ops, fp32: {t_core_torch:.3f}), qkv linears {t_qkv:.3f}, mlp+norm2 {t_mlp:.3f}, one LayerNorm {t_ln:.3f} | decoder {t_dec:.2f}")
# ---- tokenizer breakdown (stem conv / norm+act / the three residual stages / projection)
with torch.no_grad():
    tk = m.down_projection
    t_c, y0 = timed(lambda: tk.stem.conv(x))
    t_n, y1 = timed(lambda: torch.nn.functional.leaky_relu(tk.stem.norm(y0), 0.01))
    parts = []
    h = y1
    for st in tk.stages:
        t_s, h2 = timed(lambda st=st, h=h: st(h))
        t_c1, _ = timed(lambda st=st, h=h: st.conv1(h))
        parts.append((t_s, t_c1))
        h = h2
    t_p, _ = timed(lambda: tk.proj(h))
    print(f"tokenizer: stem conv {t_c:.2f}, stem norm+act {t_n:.2f}, stages (total, conv1) {[(round(a, 2), round(b, 2)) for a, b in parts]}, proj {t_p:.2f}")
    t_dec_old, _ = timed(lambda: m.up_projection.decode(vol))
    print(f"decoder: GEMM path {t_dec:.2f} ms, module chain {t_dec_old:.2f} ms")
