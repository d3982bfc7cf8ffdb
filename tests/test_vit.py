"""CPU: the 3D ViT variant (`anatomix-dev-vit`, BASELINE configs[4]) -- registry entry, constructor surface, the UNet-compatible
forward contract of the reference wrapper (anatomix/model/vit3d/architectures.py:122-165) and the module against the oracle
restatement (oracle/vit_ref.py; parity with the upstream package is unpinned, see there)."""
import pytest
import torch

import anatomix_amd
from anatomix_amd.model.load_from_hf import ANATOMIX_VARIANTS, build_variant
from anatomix_amd.model.vit3d import PRIMUS_CONFIGS, PrimusV2, build_out_norm
from oracle import vit_ref as V

SMALL = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(32, 32, 32), eva_depth=2)


def test_registry_entry_and_parameter_count():
    assert ANATOMIX_VARIANTS["anatomix-dev-vit"]["vit_kwargs"] == V.VIT_VARIANTS["anatomix-dev-vit"]     # load_from_hf.py:25-35
    assert PRIMUS_CONFIGS["S"] == {"eva_depth": 12, "eva_numheads": 6, "embed_dim": 396}                 # architectures.py:20-25
    m = build_variant("anatomix-dev-vit")
    sd = V.synthetic_state_dict(V.VIT_VARIANTS["anatomix-dev-vit"], 0)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
    n = sum(p.numel() for p in m.parameters())
    assert 25.5e6 < n < 27e6, n                                          # "anatomix-dev-vit 26M" (BASELINE.json configs[4])
    assert m.register_tokens.shape == (1, 8, 396) and m.eva.blocks[0].attn.q_norm.normalized_shape == (66,)


def test_module_matches_the_oracle_restatement_and_the_forward_contract():
    m = PrimusV2(**SMALL)
    sd = V.synthetic_state_dict(SMALL, 1)
    m.load_state_dict(sd, strict=True)
    m.eval()
    x = V.synthetic_input(5, 2, (32, 32, 32))
    with torch.no_grad():
        y = m(x)
        ref = V.forward(x, sd, SMALL, dtype=torch.float64).float()
        out, feats = m(x, [3, 5], False)
        enc = m(x, [3], True)
        out_m, mask = m(x, True)                                          # positional ret_mask of the upstream forward
    assert y.shape == (2, 32, 32, 32, 32)
    assert ((y - ref).norm() / ref.norm()).item() < 1e-5
    assert torch.equal(out, y) and len(feats) == 1 and feats[0] is out and len(enc) == 1 and torch.equal(enc[0], y)
    assert torch.equal(out_m, y) and mask.shape == (2, 1, 4, 4, 4) and bool(mask.all())
    assert abs(y.mean(dim=(2, 3, 4))).max() < 1e-4                        # ChannelDemean output norm (architectures.py:28-33)


def test_constructor_refusals_and_output_norm_modes():
    with pytest.raises(ValueError, match="8x stride"):
        PrimusV2(**dict(SMALL, patch_embed_size=(4, 4, 4)))
    with pytest.raises(NotImplementedError):
        PrimusV2(**dict(SMALL, drop_path_rate=0.1))
    x = torch.randn(1, 4, 3, 3, 3)
    assert torch.allclose(build_out_norm("demean", 4, 1e-2)(x).mean((2, 3, 4)), torch.zeros(1, 4), atol=1e-6)
    assert isinstance(build_out_norm(False, 4, 1e-2), torch.nn.Identity) and isinstance(build_out_norm(True, 4, 1e-2), torch.nn.InstanceNorm3d)
    ln = build_out_norm("layernorm", 4, 1e-5)(x)
    assert torch.allclose(ln.mean(1), torch.zeros(1, 3, 3, 3), atol=1e-5)
    with pytest.raises(ValueError):
        build_out_norm("nope", 4, 1e-2)


def test_copies_and_replicas_do_not_share_the_engine_handle():
    """The native handle belongs to one module object on one device: copy / deepcopy / pickle / DataParallel replicas start
    without it (a shared raw pointer would be a double free in __del__)."""
    import copy
    import pickle
    kw = dict(V.VIT_VARIANTS["anatomix-dev-vit"], input_shape=(32, 32, 32), eva_depth=1)
    m = PrimusV2(**kw)
    m.__dict__["_handle"] = object()                 # stands for a live native handle
    m.__dict__["_engine_sig"] = ("x",)
    for other in (copy.copy(m), copy.deepcopy(m), m._replicate_for_data_parallel()):
        assert other._handle is None and other._engine_sig is None
    m.__dict__["_handle"] = None
    assert pickle.loads(pickle.dumps(m))._handle is None


def test_head_width_must_carry_whole_rotary_bands():
    with pytest.raises(ValueError, match="multiple of 6"):
        PrimusV2(**dict(SMALL, embed_dim=256, eva_numheads=4))          # head_dim 64: 10 bands x 3 axes x 2 = 60 channels only
