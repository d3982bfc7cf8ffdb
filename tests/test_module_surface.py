"""CPU: the drop-in boundary.  anatomix_amd.Unet keeps the reference's constructor signature,
module indices, state_dict keys/dtypes and forward contract, and the C-ABI
library exports every symbol include/anatomix_amd.h declares."""
import inspect
import os
import re

import pytest
import torch

import anatomix_amd
from anatomix_amd import _lib
from oracle import unet_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constructor_signature_matches_reference():
    sig = inspect.signature(anatomix_amd.Unet.__init__)
    names = list(sig.parameters)[1:]
    assert names == ["dimension", "input_nc", "output_nc", "num_downs", "ngf", "norm", "final_act", "activation",
                     "pad_type", "doubleconv", "residual_connection", "pooling", "interp", "use_skip_connection",
                     "norm_eps"]                                     # network.py:262-279
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect._empty}
    assert d == dict(ngf=24, norm="batch", final_act="none", activation="relu", pad_type="reflect", doubleconv=True,
                     residual_connection=False, pooling="Max", interp="nearest", use_skip_connection=True,
                     norm_eps=1e-5)
    fsig = inspect.signature(anatomix_amd.Unet.forward)
    assert list(fsig.parameters)[1:] == ["input", "layers", "encode_only", "verbose"]


@pytest.mark.parametrize("variant", ["anatomix", "anatomix-dev"])
def test_state_dict_keys_and_strict_load(variant, capsys):
    kw = R.VARIANTS[variant]
    m = anatomix_amd.Unet(**kw)
    printed = capsys.readouterr().out
    assert "Encoder skip connect id" in printed and "Decoder skip connect id" in printed   # network.py:447-448
    sd = R.synthetic_state_dict(kw, 0)
    assert list(m.state_dict().keys()) == list(sd.keys())
    for k, v in m.state_dict().items():
        assert v.shape == sd[k].shape and v.dtype == sd[k].dtype, k
    m.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError):
        bad = dict(sd); bad.pop(next(iter(bad)))
        m.load_state_dict(bad, strict=True)
    plan = R.build_plan(**{k: v for k, v in kw.items() if k != "dimension"})
    assert m.encoder_idx == plan.encoder_idx and m.decoder_idx == plan.decoder_idx
    assert len(m.model) == len(plan.kinds)
    assert sum(p.numel() for p in m.parameters()) == {"anatomix": 5899344, "anatomix-dev": 94369568}[variant]
    assert m.model[2] is m.model[5]          # one shared activation instance (network.py:188-189)


def test_compiled_checkpoint_prefix_is_stripped():
    from anatomix_amd.model.load_from_hf import _load_handling_compile, build_variant
    kw = R.VARIANTS["anatomix"]
    sd = {"_orig_mod." + k: v for k, v in R.synthetic_state_dict(kw, 0).items()}
    m = _load_handling_compile(build_variant("anatomix"), sd)
    assert torch.equal(m.model[0].weight, sd["_orig_mod.model.0.weight"])
    with pytest.raises(ValueError):
        build_variant("nope")


@pytest.mark.parametrize("wrap", ["module.", "_orig_mod.", "_orig_mod.module.", "module._orig_mod.", ""])
def test_wrapped_checkpoints_of_the_trainer_load(wrap, tmp_path):
    """pretraining/models/base_model.py:340-349,458-466: the trainer saves its networks through nn.DataParallel (``module.``) and / or
    torch.compile (``_orig_mod.``); its loader strips both, decided on the first key.  A real DataParallel / compile wrapper produces
    the keys here, the file goes through torch.save / load_from_hf(weights_path=) like a published checkpoint."""
    from collections import OrderedDict
    from anatomix_amd.model.load_from_hf import build_variant, convert_dict, load_from_hf
    kw = R.VARIANTS["anatomix"]
    ref = R.synthetic_state_dict(kw, 3)
    src = build_variant("anatomix")
    src.load_state_dict(ref)
    if wrap == "module.":
        saved = torch.nn.DataParallel(src).state_dict()                       # the real wrapper's key layout
    else:
        saved = OrderedDict((wrap + k, v) for k, v in src.state_dict().items())
    assert all(k.startswith(wrap) for k in saved)
    path = tmp_path / "latest_net_G.pth"
    torch.save(saved, path)
    m = load_from_hf("anatomix", weights_path=str(path))
    got = m.state_dict()
    assert list(got) == list(ref)
    for k in ref:
        assert torch.equal(got[k], ref[k]), k
    assert list(convert_dict(ref)) == list(ref)                               # plain dicts pass through untouched


def test_stock_module_path_equals_oracle_when_opted_in():
    kw = R.VARIANTS["anatomix"]
    sd = R.synthetic_state_dict(kw, 0)
    m = anatomix_amd.Unet(**kw)
    m.load_state_dict(sd)
    m.eval()
    x = R.synthetic_input(3, 1, (32, 32, 32))
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="HIP kernels"):
            m(x)                              # CPU tensor: loud, no silent fallback
        m.allow_torch_path = True
        with pytest.warns(UserWarning):
            y = m(x)
        out, feats = m(x, [27, 31], False)
        enc = m(x, [27, 31], True)
        ref, rfeats = R.forward(x, sd, kw, layers=[27, 31])
    assert torch.allclose(y, ref, atol=1e-6) and torch.equal(out, y)
    assert all(torch.allclose(a, b, atol=1e-6) for a, b in zip(feats, rfeats)) and len(enc) == 2


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "anatomix_amd.h")).read()
    body = hdr.split('extern "C" {', 1)[1]
    declared = set(re.findall(r"\b(amx_[a-z0-9_]+)\s*\(", body))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()                          # raises if the .so is missing or lacks a symbol
    assert lib.amx_version() == 100
    for name in declared:
        assert hasattr(lib, name)


def test_load_model_contract(tmp_path):
    """convex_adam_utils.py:16-78: exactly one of ckpt_path / hf_variant, missing file is an error, architecture arguments
    keyword-only, and the model that comes back always has the checkpoint's weights."""
    from anatomix_amd.registration import load_model
    with pytest.raises(ValueError, match="exactly one"):
        load_model()
    with pytest.raises(ValueError, match="exactly one"):
        load_model("a.pth", "anatomix")
    with pytest.raises(FileNotFoundError):
        load_model(str(tmp_path / "missing.pth"))
    with pytest.raises(ValueError, match="scratch"):
        load_model("scratch")
    with pytest.raises(TypeError):
        load_model(None, None, 16)                       # architecture arguments are keyword-only, as in the reference
    kw = R.VARIANTS["anatomix"]
    sd = R.synthetic_state_dict(kw, 3)
    path = tmp_path / "anatomix.pth"
    torch.save({"_orig_mod." + k: v for k, v in sd.items()}, path)
    for m in (load_model(str(path), device="cpu"), load_model(hf_variant="anatomix", weights_path=str(path), device="cpu")):
        assert not m.training and torch.equal(m.model[0].weight, sd["model.0.weight"])
        assert torch.equal(m.model[1].running_var, sd["model.1.running_var"])


def test_copies_do_not_share_the_native_handle():
    """The C handle and the workspace belong to one module object: copies (copy / deepcopy / DataParallel replicas) start
    without one instead of sharing a raw pointer."""
    import copy
    m = anatomix_amd.Unet(**R.VARIANTS["anatomix"])
    m._handle, m._handle_key, m._workspace, m._weights_dirty = object(), (0, "f16"), torch.zeros(4), False
    try:
        for c in (copy.copy(m), copy.deepcopy(m), m._replicate_for_data_parallel()):
            assert c._handle is None and c._handle_key is None and c._workspace is None and c._weights_dirty
        assert m._handle is not None and not m._weights_dirty
    finally:
        m._handle = None                                  # not a real handle: keep __del__ away from it


def test_hip_gate_mirrors_the_library_limits():
    """Configurations the C side would refuse are refused by the Python gate with a reason (so allow_torch_path applies)."""
    dev_like = torch.zeros(1, 1, 32, 32, 32)
    for kwargs in (dict(ngf=48), dict(ngf=64), dict(ngf=12), dict(ngf=16, input_nc=17)):     # (ngf 8 / 24, any output_nc, input_nc <= 16 run)
        kw = dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16)
        kw.update(kwargs)
        m = anatomix_amd.Unet(**kw).eval()
        class _X:                                         # is_cuda is checked first: a stand-in for a GPU tensor
            is_cuda = True; requires_grad = False; shape = (1, kw["input_nc"], 32, 32, 32)
            def dim(self): return 5
        with torch.no_grad():
            assert m.hip_unsupported_reason(_X()) is not None, kwargs
