"""GPU: the fused attention core of the 3D ViT (amx_attention_qknorm_rope: per-head QK LayerNorm + rotary embedding + MFMA flash
attention) against a float64 restatement, and the whole `anatomix-dev-vit` forward on the HIP path against the oracle at the
128^3 operating size (BASELINE configs[4]).  The oracle restates published algorithms -- parity with the upstream
dynamic-network-architectures package is UNPINNED (oracle/vit_ref.py)."""
import numpy as np
import pytest
import torch

from _util import max_rel, rel_l2
import anatomix_amd
from anatomix_amd.model.vit3d import PrimusV2
from anatomix_amd.model.vit3d.architectures import EvaAttention, build_rope_table
from oracle import vit_ref as V

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,heads,hd,nreg,qk_norm,rope", [
    (1, 4104, 6, 66, 8, True, True),        # the model's shape: 16^3 patch tokens + 8 register tokens
    (2, 520, 6, 66, 8, True, True),         # 8^3 grid, batch 2
    (1, 100, 2, 64, 0, False, False),       # plain softmax attention, one partial key block, no registers
    (2, 200, 3, 18, 4, True, False),        # small head_dim (padding of both operand widths), no rotary
    (1, 129, 1, 80, 1, False, False),       # widest supported head, one query tile beyond a workgroup
])
def test_attention_core_matches_float64(device, B, N, heads, hd, nreg, qk_norm, rope):
    rs = np.random.RandomState(B * 1000 + N)
    E = heads * hd
    att = EvaAttention(E, heads, qk_norm, False).to(device)
    if qk_norm:
        for ln in (att.q_norm, att.k_norm):
            ln.weight.data = torch.from_numpy(rs.uniform(0.5, 1.5, hd).astype(np.float32)).to(device)
            ln.bias.data = torch.from_numpy((rs.randn(hd) * 0.1).astype(np.float32)).to(device)
    q, k, v = [torch.from_numpy(rs.randn(B, N, E).astype(np.float32)).to(device) for _ in range(3)]
    table = None
    if rope:
        side = round((N - nreg) ** (1 / 3))
        assert side ** 3 == N - nreg
        table = build_rope_table((side,) * 3, hd).to(device)
    with torch.no_grad():
        got = att.core_hip(q, k, v, table, nreg).cpu()
        att64 = att.double()
        ref = att64.core_torch(q.double(), k.double(), v.double(), None if table is None else table.double(), nreg).float().cpu()
    assert torch.isfinite(got).all()
    # f16 MFMA operands (q, k, probabilities, v), fp32 everything else
    assert rel_l2(got, ref) < 1e-3 and max_rel(got, ref) < 4e-3, (rel_l2(got, ref), max_rel(got, ref))


def test_attention_is_deterministic_and_refuses_bad_shapes(device):
    att = EvaAttention(6 * 66, 6, True, False).to(device)
    q, k, v = [torch.randn(1, 300, 396, device=device) for _ in range(3)]
    with torch.no_grad():
        a, b = att.core_hip(q, k, v, None, 0), att.core_hip(q, k, v, None, 0)
    assert torch.equal(a, b)
    wide = EvaAttention(2 * 128, 2, False, False).to(device)           # head_dim 128 > 80
    with torch.no_grad(), pytest.raises(RuntimeError, match="attention"):
        wide.core_hip(torch.randn(1, 64, 256, device=device), torch.randn(1, 64, 256, device=device),
                      torch.randn(1, 64, 256, device=device), None, 0)


def test_attention_on_prepared_operands_repeats_the_fused_call(device):
    """amx_attention_prepared (the flash kernel alone on the operands the fused call left in the scratch) == the fused call."""
    import ctypes
    from anatomix_amd import _lib
    att = EvaAttention(6 * 66, 6, True, False).to(device)
    q, k, v = [torch.randn(2, 520, 396, device=device) for _ in range(3)]
    table = build_rope_table((8, 8, 8), 66).to(device)
    lib = _lib.load()
    with torch.no_grad(), torch.cuda.device(device):
        want = att.core_hip(q, k, v, table, 8)
        got = torch.empty_like(want)
        nb = lib.amx_attention_scratch_bytes(2, 6, 520, 66)
        _lib.check(lib.amx_attention_prepared(_lib.ptr(att._scratch), nb, 2, 520, 6, 66, _lib.ptr(got),
                                              ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    assert torch.equal(got, want)


def test_vit_forward_on_the_hip_path_matches_the_oracle_at_128(device):
    """BASELINE configs[4]: 1 x 1 x 128^3 -> 1 x 32 x 128^3.  (One CPU oracle forward: ~0.5 TFLOP, about a minute.)"""
    kw = V.VIT_VARIANTS["anatomix-dev-vit"]
    sd = V.synthetic_state_dict(kw, 0)
    m = PrimusV2(**kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(device).eval()
    x = V.synthetic_input(100, 1)
    with torch.no_grad():
        y = m(x.to(device)).cpu()
        ref = V.forward(x, sd, kw)
    assert y.shape == (1, 32, 128, 128, 128) and torch.isfinite(y).all()
    e, mx = rel_l2(y, ref), max_rel(y, ref)
    print(f"anatomix-dev-vit 128^3: rel-L2 {e:.2e} max-rel {mx:.2e}")
    assert e <= 1e-3, (e, mx)


def test_vit_hip_and_torch_attention_paths_agree_small(device):
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(64, 64, 64), eva_depth=3)
    sd = V.synthetic_state_dict(kw, 2)
    m = PrimusV2(**kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(device).eval()
    x = V.synthetic_input(7, 2, (64, 64, 64)).to(device)
    with torch.no_grad():
        y = m(x)
    with torch.enable_grad():                                       # autograd on: the same math as torch ops
        y_t = m(x).detach()
    ref = V.forward(x.cpu(), sd, kw, dtype=torch.float64).float()
    assert rel_l2(y.cpu(), ref) <= 1e-3 and rel_l2(y_t.cpu(), ref) <= 1e-4


def test_vit_engine_refuses_host_parameters_and_names_the_switch_outside_its_envelope(device):
    """Advisor findings (round 3): the engine takes raw device pointers, so a module still on the host with a GPU input must raise
    (it used to fault the GPU); a configuration the engine does not cover must say how to run the torch composition instead."""
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(64, 64, 64), eva_depth=1)
    m = PrimusV2(**kw).eval()                                        # parameters on the CPU
    x = V.synthetic_input(7, 1, (64, 64, 64)).to(device)
    with torch.no_grad(), pytest.raises(RuntimeError, match="move the module"):
        m.forward_hip(x)
    kw2 = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(80, 80, 80), eva_depth=1)   # 10^3 token grid: not a multiple of 64
    m2 = PrimusV2(**kw2).to(device).eval()
    x2 = V.synthetic_input(7, 1, (80, 80, 80)).to(device)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="use_engine = False"):
            m2(x2)
        m2.use_engine = False
        y = m2(x2)
    assert y.shape == (1, 32, 80, 80, 80) and torch.isfinite(y).all()
