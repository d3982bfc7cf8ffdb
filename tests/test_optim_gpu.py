"""GPU: FusedAdamW (amx_adamw_step, one launch per optimizer) against torch.optim.AdamW -- the optimizer the reference builds for
netG and netF (pretraining/models/supcl_model.py:510-516, 584-590)."""
import copy

import pytest
import torch

from anatomix_amd.pretraining import FusedAdamW

pytestmark = pytest.mark.gpu

SHAPES = [(16, 1, 3, 3, 3), (16,), (64, 32, 3, 3, 3), (1,), (7, 5), (4099,), (128, 128, 3, 3, 3), (256, 128), (3,)] + [(33,)] * 60


def _params(device, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(device)) for s in SHAPES]


def _grads(params, seed, skip=()):
    g = torch.Generator().manual_seed(seed)
    out = []
    for k, p in enumerate(params):
        t = (torch.randn(*p.shape, generator=g) * 10.0 ** float(torch.randint(-4, 1, (1,), generator=g))).to(p.device)
        out.append(None if k in skip else t)
    return out


@pytest.mark.parametrize("kw", [dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5),      # the reference's step
                                dict(lr=1e-2, betas=(0.5, 0.9), eps=1e-6, weight_decay=0.1),
                                dict(lr=1e-3, weight_decay=0.0, maximize=True)])
def test_fused_adamw_follows_torch_adamw(device, kw):
    """Eight steps, 69 tensors (two launches of <= 48 descriptors; odd sizes take the scalar path), some gradients absent on some
    steps (torch skips those parameters and does not advance their step count)."""
    a = _params(device, 0)
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    opt_a, opt_b = FusedAdamW(a, **kw), torch.optim.AdamW(b, **kw)
    for it in range(8):
        skip = {2, 5} if it in (1, 2) else ()
        for p, q, g in zip(a, b, _grads(a, 100 + it, skip)):
            p.grad = None if g is None else g.clone()
            q.grad = None if g is None else g.clone()
        opt_a.step()
        opt_b.step()
    for k, (p, q) in enumerate(zip(a, b)):
        assert torch.isfinite(p).all()
        # one update moves a parameter by ~lr; the two implementations round a handful of fp32 operations differently
        assert torch.allclose(p, q, rtol=2e-6, atol=2e-6 * kw["lr"] * 8 + 1e-9), (k, (p - q).abs().max().item())
        sa, sb = opt_a.state[p], opt_b.state[q]
        assert float(sa["step"]) == float(sb["step"]) == (6.0 if k in (2, 5) else 8.0)
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-6, atol=1e-6 * float(sb["exp_avg"].abs().max()))
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-30)


def test_state_dict_moves_between_the_two_optimizers(device):
    kw = dict(lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    a = _params(device, 1)[:9]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    opt_a, opt_b = FusedAdamW(a, **kw), torch.optim.AdamW(b, **kw)            # b: step counts on the HOST (not capturable)
    for it in range(3):
        for p, q, g in zip(a, b, _grads(a, 7 + it)):
            p.grad, q.grad = g.clone(), g.clone()
        opt_a.step()
        opt_b.step()
    # swap the states: torch's -> fused, fused -> torch (capturable, as its step tensors live on the device)
    sd_a, sd_b = copy.deepcopy(opt_a.state_dict()), copy.deepcopy(opt_b.state_dict())
    opt_a2 = FusedAdamW(a, **kw)
    opt_a2.load_state_dict(sd_b)
    opt_b2 = torch.optim.AdamW(b, capturable=True, **kw)
    opt_b2.load_state_dict(sd_a)
    for it in range(3):
        for p, q, g in zip(a, b, _grads(a, 50 + it)):
            p.grad, q.grad = g.clone(), g.clone()
        opt_a2.step()
        opt_b2.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=2e-6, atol=1e-8)
        assert float(opt_a2.state[p]["step"]) == float(opt_b2.state[q]["step"]) == 6.0


def test_fused_adamw_replays_from_a_graph(device):
    kw = dict(lr=1e-2, weight_decay=1e-2)
    a = _params(device, 2)[:12]
    opt_a = FusedAdamW(a, **kw)
    grads = _grads(a, 9)
    for p, g in zip(a, grads):
        p.grad = g.clone()
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        opt_a.step()                                                 # creates the state outside the capture
    torch.cuda.current_stream(device).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt_a.step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize(device)
    steps = float(opt_a.state[a[0]]["step"])
    assert steps == 4.0                                              # the eager step and three replays; a capture executes nothing
    b2 = [torch.nn.Parameter(p.detach().clone()) for p in _params(device, 2)[:12]]
    opt_c = FusedAdamW(b2, **kw)
    for p, g in zip(b2, grads):
        p.grad = g.clone()
    for _ in range(4):
        opt_c.step()
    for p, q in zip(a, b2):
        assert torch.equal(p, q)                                     # same kernel, same inputs: bit-identical


def test_a_replayed_step_follows_the_learning_rate_schedule(device):
    """Advisor finding (round 3): by-value kernel arguments are frozen into a captured graph, and the reference changes lr every
    epoch (pretraining/models/base_model.py: update_learning_rate).  The step reads {lr, betas, eps, weight_decay} from a device
    buffer refreshed by a captured copy of a pinned host mirror: replays after `group['lr'] = ...` + refresh_hyperparameters()
    (what GraphedContrastiveStep does before every replay) must equal eager steps with the same schedule, bit for bit."""
    kw = dict(lr=1e-2, weight_decay=1e-2)
    sched = [1e-2, 1e-2, 5e-3, 2.5e-3, 0.0, 1e-3]
    a = _params(device, 4)[:12]
    opt_a = FusedAdamW(a, **kw)
    grads = _grads(a, 11)
    for p, g in zip(a, grads):
        p.grad = g.clone()
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        opt_a.step()                                                 # schedule entry 0, eagerly (creates the state)
    torch.cuda.current_stream(device).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt_a.step()
    for lr in sched[1:]:
        opt_a.param_groups[0]["lr"] = lr
        if lr == 2.5e-3:
            opt_a.param_groups[0]["weight_decay"] = 0.1              # the other hyper-parameters travel the same way
        opt_a.refresh_hyperparameters()
        graph.replay()
    torch.cuda.synchronize(device)
    b2 = [torch.nn.Parameter(p.detach().clone()) for p in _params(device, 4)[:12]]
    opt_c = FusedAdamW(b2, **kw)
    for p, g in zip(b2, grads):
        p.grad = g.clone()
    for lr in sched:
        opt_c.param_groups[0]["lr"] = lr
        if lr == 2.5e-3:
            opt_c.param_groups[0]["weight_decay"] = 0.1
        opt_c.step()
    frozen = [torch.nn.Parameter(p.detach().clone()) for p in _params(device, 4)[:12]]
    opt_f = FusedAdamW(frozen, **kw)
    for p, g in zip(frozen, grads):
        p.grad = g.clone()
    for _ in sched:
        opt_f.step()                                                 # what a frozen lr would have produced
    for p, q, r in zip(a, b2, frozen):
        assert torch.equal(p, q)
        assert not torch.equal(p, r)
    with pytest.raises(ValueError):
        opt_a.param_groups[0]["lr"] = -1.0
        opt_a.refresh_hyperparameters()


def test_refusals(device):
    with pytest.raises(NotImplementedError):
        FusedAdamW([torch.nn.Parameter(torch.zeros(3, device=device))], amsgrad=True)
    p = torch.nn.Parameter(torch.zeros(3, device=device, dtype=torch.float16))
    p.grad = torch.zeros_like(p)
    with pytest.raises(RuntimeError, match="fp32"):
        FusedAdamW([p]).step()


def test_contrastive_step_with_the_fused_optimizer_follows_the_stock_one(device):
    """Four eager contrastive steps at 64^3 on fixed coordinates, once with torch.optim.AdamW and once with FusedAdamW on copies of the
    same networks: the gradients come from the same kernels, so the trajectories differ only by the optimizers' fp32 rounding; then
    the graph-replayed step with the fused optimizers inside the graph trains (the loss goes down)."""
    import contextlib, copy, io
    from argparse import Namespace
    import numpy as np
    import anatomix_amd
    from anatomix_amd.pretraining import GraphedContrastiveStep, PatchSampleF, SupPatchNCELoss, contrastive_step
    from oracle import pretrain_inputs as PI, unet_ref as R
    kw = R.VARIANTS["anatomix"]
    with contextlib.redirect_stdout(io.StringIO()):
        netG = anatomix_amd.Unet(**kw)
        netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5))
        netF = PatchSampleF(use_mlp=True, init_type="kaiming", nc=256, n_mlps=3)
        netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=device) for c in (128, 256, 128, 64, 32, 16)])
    netG.precision = "bf16"
    netG, netF = netG.to(device).train(), netF.to(device).train()
    netG2, netF2 = copy.deepcopy(netG), copy.deepcopy(netF)
    nopt = Namespace(nce_T=0.33, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
    crits = [SupPatchNCELoss(nopt) for _ in PI.NCE_LAYERS]
    A, B, seg = [t.to(device) for t in PI.step_inputs(64)]
    okw = dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    stock = (torch.optim.AdamW(netG.parameters(), **okw), torch.optim.AdamW(netF.parameters(), **okw))
    fused = (FusedAdamW(netG2.parameters(), **okw), FusedAdamW(netF2.parameters(), **okw))
    ids = None
    torch.manual_seed(3)                     # the coordinates of step 0 are drawn here; how far the later losses drift depends on them
    for it in range(4):
        r1 = contrastive_step(netG, netF, crits, A, B, seg, PI.NCE_LAYERS, num_patches=64, optimizers=stock, sample_ids=ids)
        ids = r1["sample_ids"]
        r2 = contrastive_step(netG2, netF2, crits, A, B, seg, PI.NCE_LAYERS, num_patches=64, optimizers=fused, sample_ids=ids)
        # (unseeded, the step-3 losses were 2.5e-3 apart for some draws -- the bound was 2e-3 -- and 1e-3 for others: the drift is
        #  chaotic amplification of 1-ulp differences, see below; the sharp statement of this test is the first step)
        assert abs(r1["loss"] - r2["loss"]) < 5e-3 * abs(r1["loss"]), (it, r1["loss"], r2["loss"])
        if it == 0:
            # identical gradients went into the first update: the parameters differ by the optimizers' fp32 rounding only.  (From the
            # second step on, bf16 activations amplify those 1-ulp differences, and Adam's normalised update turns a changed sign of
            # a tiny gradient into +-lr: later steps are held through the losses.)
            assert r1["loss"] == r2["loss"]
            for p, q in zip(list(netG.parameters()) + list(netF.parameters()), list(netG2.parameters()) + list(netF2.parameters())):
                assert torch.allclose(p, q, rtol=1e-6, atol=1e-9), (p - q).abs().max().item()
    assert float(fused[0].state[next(netG2.parameters())]["step"]) == 4.0
    trained = GraphedContrastiveStep(netG2, netF2, crits, PI.NCE_LAYERS, fused, num_patches=64, warmup=2)
    assert trained.opt_in_graph
    losses = [trained(A, B, seg)["loss"] for _ in range(12)]
    assert all(np.isfinite(losses)) and np.mean(losses[-3:]) < np.mean(losses[:3]), losses
