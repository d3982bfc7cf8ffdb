"""GPU: the ViT engine (amx_vit_forward: own kernels for the conv tokenizer, the token-matrix products, the attention and the
decoder) against the oracle restatement, stage by stage: the MFMA product kernel alone (amx_linear), the tokenizer's tokens, the
decoder with no blocks in between, a few blocks, and the whole network at reduced and odd sizes.  (The 128^3 operating point is
tests/test_vit_gpu.py.)  Parity with the upstream package is UNPINNED (oracle/vit_ref.py)."""
import ctypes

import numpy as np
import pytest
import torch

from _util import max_rel, rel_l2
from anatomix_amd import _lib
from anatomix_amd.model.vit3d import PrimusV2
from oracle import vit_ref as V

pytestmark = pytest.mark.gpu


def _linear(x, w, b, split):
    lib = _lib.load()
    y = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.amx_linear(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), x.shape[0], x.shape[1], w.shape[0], int(split), _lib.ptr(y),
                                  ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return y


@pytest.mark.parametrize("M,K,N", [(16416, 396, 1188), (4104, 396, 396), (1000, 1056, 396), (300, 396, 2112), (130, 128, 396),
                                   (77, 40, 20), (4096, 216, 960)])
@pytest.mark.parametrize("split", [0, 1])
def test_linear_product_matches_float64(device, M, K, N, split):
    rs = np.random.RandomState(M + K + N)
    x = torch.from_numpy(rs.randn(M, K).astype(np.float32)).to(device)
    w = torch.from_numpy((rs.randn(N, K) / np.sqrt(K)).astype(np.float32)).to(device)
    b = torch.from_numpy(rs.randn(N).astype(np.float32)).to(device)
    got = _linear(x, w, b, split).cpu()
    ref = (x.double() @ w.double().t() + b.double()).float().cpu()
    assert torch.isfinite(got).all()
    e = rel_l2(got, ref)
    assert e < (2e-6 if split else 6e-4), e                    # hi + lo operands: fp32-grade; plain f16 operands: 2^-11 per factor


def _model(kw, seed, device):
    sd = V.synthetic_state_dict(kw, seed)
    m = PrimusV2(**kw)
    m.load_state_dict(sd, strict=True)
    return m.to(device).eval(), sd


def _oracle_tokens(x, sd, kw):
    feat = V.tokenizer(x, sd, kw)
    tok = feat.flatten(2).transpose(1, 2) + sd["eva.pos_embed"]
    return torch.cat((sd["register_tokens"].expand(x.shape[0], -1, -1), tok), 1)


@pytest.mark.parametrize("size,batch", [((32, 32, 32), 2), ((64, 64, 64), 1), ((32, 64, 32), 3)])
def test_tokenizer_tokens_match_the_oracle(device, size, batch):
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=size, eva_depth=1)
    m, sd = _model(kw, 3, device)
    x = V.synthetic_input(11, batch, size)
    with torch.no_grad():
        m.forward_hip(x.to(device), n_blocks=0)
        got = m.debug_read("tokens").view(batch, -1, kw["embed_dim"]).cpu()
    ref = _oracle_tokens(x.double(), {k: v.double() for k, v in sd.items()}, kw).float()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    e = rel_l2(got, ref)
    assert e < 2e-5, e                                          # strict (hi + lo) tokenizer: fp32-grade


@pytest.mark.parametrize("depth", [0, 1, 3])
def test_engine_forward_matches_the_oracle_at_64(device, depth):
    """depth 0: tokenizer -> final norm -> decoder -> demean only (isolates the decoder); then with blocks."""
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(64, 64, 64), eva_depth=max(depth, 1))
    m, sd = _model(kw, 4, device)
    x = V.synthetic_input(12, 2, (64, 64, 64))
    with torch.no_grad():
        y = m.forward_hip(x.to(device), n_blocks=depth).cpu()
    ref = V.forward(x, sd, dict(kw, eva_depth=depth), dtype=torch.float64).float()
    assert y.shape == ref.shape and torch.isfinite(y).all()
    e = rel_l2(y, ref)
    print(f"depth {depth}: rel-L2 {e:.2e} max-rel {max_rel(y, ref):.2e}")
    assert e < (1e-4 if depth == 0 else 6e-4), e


def test_engine_is_deterministic_and_follows_parameter_updates(device):
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(32, 32, 32), eva_depth=2)
    m, sd = _model(kw, 5, device)
    x = V.synthetic_input(13, 2, (32, 32, 32)).to(device)
    with torch.no_grad():
        a, b = m(x), m(x)
        assert torch.equal(a, b)
        m.eva.blocks[1].gamma_2.mul_(0.5)                        # in-place update: the engine must repack
        c = m(x)
    assert not torch.equal(a, c)
    sd2 = {k: v.clone() for k, v in m.state_dict().items()}
    ref = V.forward(x.cpu(), {k: v.cpu() for k, v in sd2.items()}, kw, dtype=torch.float64).float()
    assert rel_l2(c.cpu(), ref) < 6e-4


def test_engine_and_torch_composition_agree_and_other_configurations_run(device):
    """No QK norm, no inner norm, no LayerScale, no register tokens, 16 output classes, out_norm none."""
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(32, 32, 32), eva_depth=2, qk_norm=False, scale_attn_inner=False,
              init_values=None, num_register_tokens=0, num_classes=16, out_norm="none")
    m = PrimusV2(**kw).to(device).eval()
    x = V.synthetic_input(14, 1, (32, 32, 32)).to(device)
    with torch.no_grad():
        y = m(x)
        m.use_engine = False
        y_t = m(x)
    assert y.shape == (1, 16, 32, 32, 32)
    assert rel_l2(y.cpu(), y_t.cpu()) < 1e-3


@pytest.mark.parametrize("name,embed,heads", [("B", 792, 12), ("M", 864, 12), ("L", 1056, 16)])
def test_engine_runs_the_wider_primus_configurations(device, name, embed, heads):
    """PRIMUS_CONFIGS B / M / L (architectures.py:20-25): head_dim 66 / 72 / 66, SwiGLU hidden 2112 / 2304 / 2816, decoder widths
    derived from the embedding width; two blocks at 32^3 against the restatement.  (L: a head's weight slice no longer fits LDS --
    the q | k and v projections take the streaming kernel.)"""
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(32, 32, 32), eva_depth=2, embed_dim=embed, eva_numheads=heads)
    m, sd = _model(kw, 7, device)
    x = V.synthetic_input(15, 2, (32, 32, 32))
    with torch.no_grad():
        y = m(x.to(device)).cpu()
    ref = V.forward(x, sd, kw, dtype=torch.float64).float()
    assert y.shape == ref.shape and torch.isfinite(y).all()
    e = rel_l2(y, ref)
    print(f"PrimusV2-{name}: rel-L2 {e:.2e}")
    assert e < 6e-4, e


@pytest.mark.parametrize("seed", range(8))
def test_engine_against_the_torch_composition_on_random_configurations(device, seed):
    """Seeded sweep over the constructor envelope (token grids incl. non-cubic, batch, heads x head_dim, register tokens, classes,
    flags): engine vs the same module composed of torch operators (fp32 everywhere except its HIP attention core)."""
    rs = np.random.RandomState(1000 + seed)
    grid = [(4, 4, 4), (4, 8, 4), (8, 4, 2), (2, 8, 4), (8, 8, 2), (4, 4, 8)][rs.randint(6)]
    heads, hd = [(6, 66), (4, 60), (8, 36), (4, 78), (12, 66), (2, 18)][rs.randint(6)]      # head_dim: a multiple of 6 (rotary bands); embed_dim: a multiple of 4 (engine)
    kw = dict(input_channels=1, num_classes=int(rs.choice([4, 16, 32, 36])), embed_dim=heads * hd, patch_embed_size=(8, 8, 8),
              input_shape=tuple(8 * g for g in grid), eva_depth=int(rs.randint(1, 3)), eva_numheads=heads,
              num_register_tokens=int(rs.choice([0, 3, 8])), init_values=[None, 0.1][rs.randint(2)], scale_attn_inner=bool(rs.randint(2)),
              qk_norm=bool(rs.randint(2)), out_norm=["none", "demean", "instance"][rs.randint(3)], out_norm_eps=1e-2, in_eps=1e-2)
    torch.manual_seed(seed)
    m = PrimusV2(**kw).to(device).eval()
    with torch.no_grad():
        for prm in m.parameters():                                  # away from the all-ones / zeros defaults of norms and LayerScale
            if prm.dim() == 1:
                prm.add_(0.1 * torch.randn_like(prm))
    batch = int(rs.randint(1, 4))
    x = torch.rand(batch, 1, *kw["input_shape"], device=device)
    with torch.no_grad():
        y = m(x)
        m.use_engine = False
        y_t = m(x)
    assert y.shape == y_t.shape and torch.isfinite(y).all()
    e = rel_l2(y.cpu(), y_t.cpu())
    print(f"sweep seed {seed}: rel-L2 {e:.2e}")
    assert e < 1e-3, (kw, batch, e)


def test_single_f16_stem_output_is_an_opt_in_within_tolerance(device, monkeypatch):
    """amx_vit_cfg.stem_split = 0 (env AMX_VIT_STEM_SPLIT=0): the stem's output without its lo plane -- faster, measured 3.7e-4 on
    its own; the default keeps the plane (fp32-grade tokens, test_tokenizer_tokens_match_the_oracle)."""
    monkeypatch.setenv("AMX_EXPERIMENT", "1")                   # the Python A/B switches are only read under this gate (_lib.exp_env)
    monkeypatch.setenv("AMX_VIT_STEM_SPLIT", "0")
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(64, 64, 64), eva_depth=2)
    m, sd = _model(kw, 8, device)
    assert m._vit_cfg["stem_split"] == 0
    x = V.synthetic_input(16, 1, (64, 64, 64))
    with torch.no_grad():
        y = m(x.to(device)).cpu()
    ref = V.forward(x, sd, kw, dtype=torch.float64).float()
    e = rel_l2(y, ref)
    assert 5e-5 < e < 1e-3, e


def test_engine_refuses_what_it_cannot_run(device):
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(32, 32, 32), eva_depth=1)
    m, _ = _model(kw, 6, device)
    with torch.no_grad(), pytest.raises(ValueError, match="built for inputs"):
        m(torch.zeros(1, 1, 64, 64, 64, device=device))
    kw2 = dict(kw, input_shape=(24, 24, 24))                        # 27 tokens: not a multiple of 64
    m2 = PrimusV2(**kw2).to(device).eval()
    # outside the engine's envelope: a RuntimeError that carries the engine's reason and names the switch to the torch composition
    with torch.no_grad(), pytest.raises(RuntimeError, match="token grid") as ei:
        m2(torch.zeros(1, 1, 24, 24, 24, device=device))
    assert "use_engine" in str(ei.value)
