"""GPU parity of the whole forward (anatomix_amd.Unet on the HIP kernels) against the CPU oracle.

Two distances are checked:
  * kernel correctness: HIP output vs ``oracle.unet_ref.forward_lowp`` (same rounding points,
    fp32 accumulation) -- differences are accumulation order + rare 1-ulp storage flips;
  * precision claim: HIP f16 output vs the fp32 oracle must meet the north-star tolerance
    (<= 1e-3 relative) on the SURVEY-spec synthetic weights.
"""
import numpy as np
import pytest
import torch

from _util import max_rel, rel_l2
import anatomix_amd
from oracle import unet_ref as R

pytestmark = pytest.mark.gpu

KW = R.VARIANTS["anatomix"]


def _model(device, seed, gain, precision="f16"):
    m = anatomix_amd.Unet(**KW)
    sd = R.synthetic_state_dict(KW, seed, gain=gain)
    m.load_state_dict(sd, strict=True)
    m.precision = precision
    return m.to(device).eval(), sd


@pytest.mark.parametrize("size,n", [((32, 32, 32), 1), ((64, 64, 64), 1), ((32, 48, 64), 2)])
@pytest.mark.parametrize("gain", [1.0, 2 ** 0.5])
def test_forward_matches_emulated_oracle(device, size, n, gain):
    m, sd = _model(device, 0, gain)
    x = R.synthetic_input(100, n, size)
    with torch.no_grad():
        y = m(x.to(device)).cpu()
        ref = R.forward_lowp(x, sd, KW, torch.float16)
    assert y.shape == ref.shape and y.dtype == torch.float32
    assert torch.isfinite(y).all()
    # kernel-vs-emulation distance (same rounding points; the rest is accumulation order + 1-ulp storage flips that the deeper
    # layers amplify), NOT the precision claim: 4.6e-4 at gain 1, 1.6e-3 .. 1.8e-3 on He-scaled weights, where f16 storage itself
    # is 1.9e-3 from fp32 -- outside the tolerance, which is why He-scaled checkpoints belong to `precision = "strict"`
    assert rel_l2(y, ref) < (1e-3 if gain == 1.0 else 2.5e-3), (rel_l2(y, ref), max_rel(y, ref))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_forward_f16_meets_1e3_vs_fp32_oracle(device, seed):
    """north_star: feature maps within 1e-3 (relative) of the fp32 reference -- held in the rel-L2 sense with f16 storage on all
    four weight seeds (measured 5.1e-4 .. 6.7e-4, tools/tol_probe.py).  The worst single voxel relative to the largest feature
    (max-norm) is 8e-4 .. 1.4e-3 over the same seeds: f16 storage does NOT hold 1e-3 in the max-norm on every seed -- the bound
    below is what it does hold; `precision = "strict"` holds both norms with >= 60x margin (tests/test_strict_precision_gpu.py)."""
    m, sd = _model(device, seed, 1.0)
    x = R.synthetic_input(100 + seed, 1, (64, 64, 64))
    with torch.no_grad():
        y = m(x.to(device)).cpu()
        ref = R.forward(x, sd, KW)
    assert rel_l2(y, ref) <= 1e-3, rel_l2(y, ref)
    assert max_rel(y, ref) <= 1.5e-3, max_rel(y, ref)


def test_forward_bf16_mode_runs_and_is_bf16_accurate(device):
    m, sd = _model(device, 0, 1.0, precision="bf16")
    x = R.synthetic_input(100, 1, (32, 32, 32))
    with torch.no_grad():
        y = m(x.to(device)).cpu()
        ref_emul = R.forward_lowp(x, sd, KW, torch.bfloat16)
        ref = R.forward(x, sd, KW)
    assert rel_l2(y, ref_emul) < 8e-3
    assert rel_l2(y, ref) < 2e-2


def test_forward_128_probes(device):
    """BASELINE config 1 size: 1x1x128^3.  Full fp32 oracle on the host (~1.5 s)."""
    m, sd = _model(device, 0, 1.0)
    x = R.synthetic_input(100, 1, (128, 128, 128))
    with torch.no_grad():
        y = m(x.to(device)).cpu()
        ref = R.forward(x, sd, KW)
    assert rel_l2(y, ref) <= 1e-3, rel_l2(y, ref)


def test_batch_independence_and_determinism(device):
    m, _ = _model(device, 0, 1.0)
    x = R.synthetic_input(5, 2, (32, 32, 32)).to(device)
    with torch.no_grad():
        y2 = m(x)
        y0 = m(x[:1])
        y1 = m(x[1:])
        y2b = m(x)
    assert torch.equal(y2[:1], y0) and torch.equal(y2[1:], y1)
    assert torch.equal(y2, y2b)


def test_shape_errors_and_unsupported_modes_are_loud(device):
    m, _ = _model(device, 0, 1.0)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="divisible"):
            m(torch.zeros(1, 1, 40, 32, 32, device=device))
        with pytest.raises(RuntimeError, match="smaller than 2"):
            m(torch.zeros(1, 1, 16, 32, 32, device=device))
    # autograd on an eval-mode network runs the differentiable path (frozen BatchNorm statistics) ...
    y = m(torch.zeros(1, 1, 32, 32, 32, device=device))
    assert y.requires_grad and y.shape == (1, 16, 32, 32, 32)
    # ... and what that path does not cover is refused with its reason (rows wider than the weight-gradient tile)
    with pytest.raises(RuntimeError, match="32 <= W <= 128"):
        m(torch.zeros(1, 1, 16, 16, 160, device=device))
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="not on a GPU"):
            m(torch.zeros(1, 1, 32, 32, 32))


def test_weights_repacked_after_load_state_dict(device):
    m, sd = _model(device, 0, 1.0)
    x = R.synthetic_input(9, 1, (32, 32, 32))
    with torch.no_grad():
        y0 = m(x.to(device)).cpu()
        sd1 = R.synthetic_state_dict(KW, 1)
        m.load_state_dict(sd1, strict=True)
        y1 = m(x.to(device)).cpu()
        ref1 = R.forward_lowp(x, sd1, KW, torch.float16)
    assert not torch.allclose(y0, y1)
    assert rel_l2(y1, ref1) < 1e-3


def test_torch_compile_graph_breaks_cleanly(device):
    """The reference wraps the module in torch.compile (README.md:64, pretraining/trainers/train.py:223)."""
    m, _ = _model(device, 0, 1.0)
    x = R.synthetic_input(100, 1, (32, 32, 32)).to(device)
    with torch.no_grad():
        y0 = m(x)
        y1 = torch.compile(m)(x)
        seq = torch.nn.Sequential(m, torch.nn.Conv3d(16, 3, 1).to(device))
        a, b = seq(x), torch.compile(seq)(x)
    assert torch.equal(y0, y1)
    assert torch.allclose(a, b, atol=1e-5)


def test_in_place_parameter_updates_are_picked_up(device):
    """optimizer.step() / nn.init.* / net.apply(init) change parameters in place without any module hook firing."""
    m, sd = _model(device, 0, 1.0)
    x = R.synthetic_input(9, 1, (32, 32, 32))
    with torch.no_grad():
        y0 = m(x.to(device)).cpu()
        m.model[65].weight.mul_(2.0)                       # in place, no hook
        m.model[1].running_var.add_(0.5)                   # a buffer, too
        y1 = m(x.to(device)).cpu()
        sd2 = {k: v.clone() for k, v in sd.items()}
        sd2["model.65.weight"] = sd2["model.65.weight"] * 2.0
        sd2["model.1.running_var"] = sd2["model.1.running_var"] + 0.5
        ref = R.forward(x, sd2, KW)
    assert not torch.allclose(y0, y1)
    assert rel_l2(y1, ref) < 1e-3


@pytest.mark.parametrize("kw,size", [
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16), (8, 12, 16)),     # W < 32: output through the export pass
    (dict(dimension=3, input_nc=1, output_nc=64, num_downs=1, ngf=16), (32, 32, 32)),    # more than 32 output channels
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=3, ngf=16, doubleconv=False), (16, 16, 24)),
])
def test_small_volumes_and_wide_outputs(device, kw, size):
    m = anatomix_amd.Unet(**kw)
    sd = R.synthetic_state_dict(kw, 4)
    m.load_state_dict(sd, strict=True)
    m = m.to(device).eval()
    x = R.synthetic_input(21, 2, size)
    with torch.no_grad():
        y = m(x.to(device)).cpu()
        ref = R.forward(x, sd, kw)
    assert y.shape == ref.shape
    assert rel_l2(y, ref) <= (1.5e-3 if kw["output_nc"] > 32 else 1e-3), rel_l2(y, ref)


def test_forward_is_hip_graph_capturable(device):
    """No entry of the forward synchronises or touches host memory after load_state_dict, so a forward can be captured in a
    HIP graph and replayed on new input (include/anatomix_amd.h: 'asynchronous on the given stream')."""
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = anatomix_amd.Unet(**R.VARIANTS["anatomix"])
    model.load_state_dict(R.synthetic_state_dict(R.VARIANTS["anatomix"], 0))
    model = model.to(device).eval()
    x = torch.rand(2, 1, 64, 64, 64, device=device)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(x)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = model(x)
        x.copy_(torch.rand_like(x))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, model(x))


def test_results_do_not_depend_on_memory_nobody_wrote():
    """tools/poison_check.py: unusual configurations (wide output conv, tiny volumes, the dev variant) after filling the
    allocator's free memory with NaNs / large values.  Caught the level-0 arena slot being sized by ngf while an output conv with
    more channels than ngf was staged in it."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "poison_check.py")], capture_output=True, text=True, timeout=300)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:]
    assert out.returncode == 0 and tail == "poison check: 0 bad results", out.stdout[-2000:] + out.stderr[-2000:]


def test_concurrent_chunks_do_not_change_results(device):
    """Batches of >= 8 volumes run as chunks of 4 on two HIP streams (network.py::_forward_chunks): same values as one launch
    sequence, for full and ragged chunk counts, and the caller's stream sees the finished output."""
    m, _ = _model(device, 0, 1.0)
    x = R.synthetic_input(7, 9, (32, 32, 32)).to(device)
    with torch.no_grad():
        m.concurrent_chunks = 0
        want = m(x)
        m.concurrent_chunks = 4
        got9 = m(x)                     # chunks 4 + 4 + 1
        got8 = m(x[:8])                 # chunks 4 + 4
        m.concurrent_chunks = 2
        got_small = m(x[:5])            # chunks 2 + 2 + 1
        s = torch.cuda.Stream(device)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):      # a non-default caller stream
            m.concurrent_chunks = 4
            got_side = m(x)
        torch.cuda.current_stream(device).wait_stream(s)
    assert torch.equal(got9, want) and torch.equal(got8, want[:8]) and torch.equal(got_small, want[:5]) and torch.equal(got_side, want)


@pytest.mark.parametrize("kw,size,layers", [
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=4), (32, 32, 64), []),                       # the reference's DEFAULT width ngf = 24
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=2), (16, 32, 64), [1, 8, 9, 16, 23, 27, 30]),     # taps incl. a padded 24-channel tensor,
    (dict(dimension=3, input_nc=1, output_nc=32, num_downs=2, ngf=8), (16, 16, 32), [2, 9, 23]),            # the pool and the post-concat Upsample id
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=24, norm="instance", pooling="Avg", interp="trilinear",
          norm_eps=1e-2), (16, 16, 32), [5, 23]),
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=24, use_skip_connection=False), (16, 16, 32), []),
])
def test_widths_that_are_not_multiples_of_16(device, kw, size, layers):
    """network.py:268: `ngf=24` is the constructor's default.  The ngf-wide tensors are stored padded to 32 (8 -> 16) channels with
    exact zeros; weights, norms, concat offsets and the feature taps see the reference's channel counts."""
    m = anatomix_amd.Unet(**kw)
    sd = R.synthetic_state_dict({**dict(ngf=24), **kw}, 6)
    m.load_state_dict(sd, strict=True)
    m = m.to(device).eval()
    x = R.synthetic_input(33, 2, size)
    full = {**dict(ngf=24), **kw}
    with torch.no_grad():
        if layers:
            y, feats = m(x.to(device), layers)
            ref, rfeats = R.forward(x, sd, full, layers=layers)
        else:
            y, feats, rfeats = m(x.to(device)), [], []
            ref = R.forward(x, sd, full)
    assert y.shape == ref.shape and torch.isfinite(y).all()
    strict = kw.get("norm") == "instance"                 # InstanceNorm networks default to strict precision
    assert rel_l2(y.cpu(), ref) <= (2e-4 if strict else 1.2e-3), rel_l2(y.cpu(), ref)
    assert len(feats) == len(rfeats)
    for a, b in zip(feats, rfeats):
        assert a.shape == b.shape and rel_l2(a.cpu(), b) <= (2e-4 if strict else 2e-3), (a.shape, rel_l2(a.cpu(), b))


@pytest.mark.parametrize("kw,size,layers", [
    # the argparse defaults of the reference's pretraining options (base_options.py:69-84): two input channels, 33 output channels
    (dict(dimension=3, input_nc=2, output_nc=33, num_downs=2, ngf=16), (16, 32, 64), []),
    (dict(dimension=3, input_nc=3, output_nc=5, num_downs=2, ngf=24), (16, 16, 32), [0, 2, 9, 23, 37]),
    (dict(dimension=3, input_nc=4, output_nc=16, num_downs=1, ngf=32, norm="instance", norm_eps=1e-2), (8, 16, 32), [1]),
])
def test_input_and_output_channel_counts_of_the_constructor_envelope(device, kw, size, layers):
    """input_nc > 1: the fp32 NCDHW input is imported into a 16-channel tensor and the first conv is an ordinary layer; output_nc
    not a multiple of 16: the output conv stores padded channels and one export pass writes the fp32 NCDHW result."""
    m = anatomix_amd.Unet(**kw)
    sd = R.synthetic_state_dict(kw, 8)
    m.load_state_dict(sd, strict=True)
    m = m.to(device).eval()
    x = torch.cat([R.synthetic_input(50 + c, 2, size) for c in range(kw["input_nc"])], dim=1)
    with torch.no_grad():
        if layers:
            y, feats = m(x.to(device), layers)
            ref, rfeats = R.forward(x, sd, kw, layers=layers)
        else:
            y, feats, rfeats = m(x.to(device)), [], []
            ref = R.forward(x, sd, kw)
    assert y.shape == ref.shape == (2, kw["output_nc"]) + size and torch.isfinite(y).all()
    strict = kw.get("norm") == "instance"
    assert rel_l2(y.cpu(), ref) <= (2e-4 if strict else 1.5e-3), rel_l2(y.cpu(), ref)
    for a, b in zip(feats, rfeats):
        assert a.shape == b.shape and rel_l2(a.cpu(), b) <= (2e-4 if strict else 2e-3), (a.shape, rel_l2(a.cpu(), b))
    # the fused sliding-window entry is single-channel only: the generic path serves these networks
    from anatomix_amd.registration.sliding_window import _fused_ok
    assert not _fused_ok(m, x.to(device), size) or kw["input_nc"] == 1
