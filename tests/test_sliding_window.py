"""CPU: host logic of the sliding-window caller (anatomix_amd.registration) against the numpy oracle
and against construction-level properties (the MONAI algorithm is restated, parity unpinned)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from anatomix_amd.registration.sliding_window import importance_map, sliding_window_inference, window_starts
from anatomix_amd.registration.convex_adam_utils import minmax
from oracle import sliding_window_ref as O


def test_window_schedule_of_the_reference_configuration():
    # 256^3 volume, roi 128, overlap 0.8 -> interval 25 -> 7 starts per axis, 343 windows
    s = window_starts((256, 256, 256), (128, 128, 128), 0.8)
    assert len(s) == 343 and s[0] == (0, 0, 0) and s[-1] == (128, 128, 128)
    assert sorted({a for a, _, _ in s}) == [0, 25, 50, 75, 100, 125, 128]
    assert s[1] == (0, 0, 25)                                      # last axis fastest
    assert sorted({c for _, _, c in window_starts((128, 160, 192), (128,) * 3, 0.8)}) == [0, 25, 50, 64]
    assert sorted({b for _, b, _ in window_starts((128, 160, 192), (128,) * 3, 0.8)}) == [0, 25, 32]
    assert window_starts((128, 128, 128), (128,) * 3, 0.8) == [(0, 0, 0)]
    assert len(window_starts((192, 192, 192), (128,) * 3, 0.7)) == 3 ** 3    # interval 38: 0, 38, 64
    for size in [(130, 200, 128), (256, 129, 177)]:
        assert window_starts(size, (128,) * 3, 0.8) == [(z, y, x) for z in O.starts_1d(size[0], 128, 0.8)
                                                        for y in O.starts_1d(size[1], 128, 0.8)
                                                        for x in O.starts_1d(size[2], 128, 0.8)]


def test_importance_map():
    m = importance_map((128, 128, 128), "gaussian", 0.25)
    assert m.shape == (128, 128, 128) and m.dtype == torch.float32
    assert torch.equal(m, m.flip(0)) and torch.equal(m, m.flip(2))
    assert abs(m.max().item() - np.exp(-0.25 / 2048.0) ** 3) < 1e-6      # sigma = 32, centre offset 0.5 per axis
    assert m.min().item() >= 1e-3 and abs(m.min().item() - np.exp(-63.5 ** 2 / 2048.0) ** 3) < 1e-7
    np.testing.assert_allclose(m.numpy(), O.gaussian_map((128, 128, 128), 0.25), rtol=1e-6)
    assert torch.equal(importance_map((4, 5, 6), "constant"), torch.ones(4, 5, 6))
    tiny = importance_map((16, 16, 16), "gaussian", 0.05)               # forces the 1e-3 floor
    assert tiny.min().item() == pytest.approx(1e-3)


def _conv_predictor():
    torch.manual_seed(0)
    w = torch.randn(3, 1, 3, 3, 3) * 0.2
    return lambda x: F.conv3d(F.pad(x, (1,) * 6, mode="reflect"), w)


@pytest.mark.parametrize("mode", ["constant", "gaussian"])
def test_identity_and_constant_predictors(mode):
    x = torch.rand(1, 1, 40, 52, 37)
    y = sliding_window_inference(x, (16, 24, 16), 3, lambda t: t, overlap=0.6, mode=mode, sigma_scale=0.25)
    assert torch.allclose(y, x, atol=1e-6)
    y = sliding_window_inference(x, (16, 24, 16), 2, lambda t: torch.full_like(t, 2.5).repeat(1, 4, 1, 1, 1), overlap=0.5,
                                 mode=mode)
    assert y.shape == (1, 4, 40, 52, 37) and torch.allclose(y, torch.full_like(y, 2.5), atol=1e-5)


def test_matches_numpy_oracle_and_is_batching_independent():
    pred = _conv_predictor()
    x = torch.rand(2, 1, 40, 33, 48)
    ref = np.stack([O.sliding_window(x[b].numpy(), (16, 16, 32), lambda a: pred(torch.from_numpy(a)).numpy(), 0.8, "gaussian",
                                     0.25) for b in range(2)])
    for bs in (1, 2, 5):
        y = sliding_window_inference(x, (16, 16, 32), bs, pred, overlap=0.8, mode="gaussian", sigma_scale=0.25)
        np.testing.assert_allclose(y.numpy(), ref, rtol=1e-5, atol=1e-6)


def test_single_window_equals_direct_call_and_small_volumes_are_padded():
    pred = _conv_predictor()
    x = torch.rand(1, 1, 16, 16, 16)
    assert torch.allclose(sliding_window_inference(x, 16, 2, pred, overlap=0.8, mode="gaussian"), pred(x), atol=1e-6)
    small = torch.rand(1, 1, 10, 16, 13)
    y = sliding_window_inference(small, 16, 2, lambda t: t, overlap=0.8)
    assert y.shape == small.shape and torch.allclose(y, small, atol=1e-6)


def test_minmax_matches_reference_semantics():
    a = np.array([[-2.0, 0.0], [4.0, 10.0]], dtype=np.float32)
    np.testing.assert_allclose(minmax(a), (a + 2) / 12)
    np.testing.assert_allclose(minmax(a, minclip=0.0), np.clip(a, 0, None) / 10)          # ONE bound is enough
    np.testing.assert_allclose(minmax(a, maxclip=4.0), (np.clip(a, None, 4) + 2) / 6)
    np.testing.assert_allclose(minmax(a, 0.0, 4.0), np.clip(a, 0, 4) / 4)
    with np.errstate(invalid="ignore", divide="ignore"):
        assert np.isnan(minmax(np.ones((2, 2), np.float32))).all()                        # no zero-range guard


def _dist_worker(rank, world, port, out):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pred = _conv_predictor()
    torch.manual_seed(1)
    x = torch.rand(2, 1, 40, 33, 48)
    kw = dict(overlap=0.8, mode="gaussian", sigma_scale=0.25, group=dist.group.WORLD)
    y = sliding_window_inference(x, (16, 16, 32), 2, pred, **kw)                        # gathered: full volume on every rank
    slab, z0, z1 = sliding_window_inference(x, (16, 16, 32), 2, pred, return_slab=True, **kw)
    torch.save(dict(y=y, slab=slab, z0=z0, z1=z1), f"{out}.{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_window_sharding_over_ranks_matches_single_process(tmp_path, world):
    """N > 1 path on CPU (gloo): windows dealt to the ranks in z-ordered runs, ONE exchange step -- the z-slab reduce-scatter by
    direct transfers of the touched planes -- then each rank normalises its own slab; gathered and sharded results."""
    import socket
    import torch.multiprocessing as mp
    from anatomix_amd.registration.sliding_window import slab_bounds
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "y.pt")
    mp.spawn(_dist_worker, args=(world, port, out), nprocs=world, join=True)
    pred = _conv_predictor()
    torch.manual_seed(1)
    x = torch.rand(2, 1, 40, 33, 48)
    ref = sliding_window_inference(x, (16, 16, 32), 2, pred, overlap=0.8, mode="gaussian", sigma_scale=0.25)
    bounds = slab_bounds(40, world)
    for r in range(world):
        d = torch.load(f"{out}.{r}")
        assert torch.allclose(d["y"], ref, rtol=1e-5, atol=1e-6)
        assert (d["z0"], d["z1"]) == (bounds[r], bounds[r + 1])
        assert torch.allclose(d["slab"], ref[:, :, d["z0"]:d["z1"]], rtol=1e-5, atol=1e-6)
