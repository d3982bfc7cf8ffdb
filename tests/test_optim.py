"""CPU: the host side of FusedAdamW (anatomix_amd/pretraining/optim.py) -- constructor contract of torch.optim.AdamW as the reference
calls it (pretraining/models/supcl_model.py:510-516), refusals, and that nothing but the HIP kernel ever updates a parameter."""
import pytest
import torch

from anatomix_amd.pretraining import FusedAdamW


def test_constructor_mirrors_adamw_and_refuses_what_is_not_offered():
    p = [torch.nn.Parameter(torch.zeros(3))]
    opt = FusedAdamW(p, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    ref = torch.optim.AdamW(p, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    g, r = opt.param_groups[0], ref.param_groups[0]
    for k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize"):
        assert g[k] == r[k]
    assert set(r) <= set(g) | {"params"}                      # every hyper-parameter key torch writes into a checkpoint is there
    assert g["capturable"] is True                            # what GraphedContrastiveStep checks before putting step() in the graph
    with pytest.raises(NotImplementedError):
        FusedAdamW(p, amsgrad=True)
    for bad in (dict(lr=-1.0), dict(eps=-1.0), dict(betas=(1.0, 0.9)), dict(betas=(0.9, 1.0)), dict(weight_decay=-0.1)):
        with pytest.raises(ValueError):
            FusedAdamW(p, **bad)
    with pytest.raises(ValueError):
        FusedAdamW(p, lr=torch.tensor(1e-3))


def test_there_is_no_host_path():
    p = torch.nn.Parameter(torch.ones(5))
    p.grad = torch.ones(5)
    opt = FusedAdamW([p], lr=0.1)
    with pytest.raises(RuntimeError, match="CUDA"):
        opt.step()
    assert torch.equal(p.detach(), torch.ones(5)) and len(opt.state[p]) == 0
    q = torch.nn.Parameter(torch.ones(5))                      # no gradient: skipped like torch does, no state created
    FusedAdamW([q]).step()
    assert len(FusedAdamW([q]).state) == 0
