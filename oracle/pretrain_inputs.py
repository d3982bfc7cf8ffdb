"""Deterministic synthetic inputs / parameters of the contrastive-pretraining parity tests.

TEST INFRASTRUCTURE ONLY.  Shared by ``oracle/make_golden_pretrain.py`` (which feeds them to the real reference) and
by the tests (which feed them to this repository's implementation), so that only OUTPUTS are stored as fixtures.
Constants are the launcher defaults of the reference (pretraining/scripts/pretrain_anatomix.py).
"""
from collections import OrderedDict

import numpy as np
import torch

NCE_T = 0.33
NUM_PATCHES = 512
NETF_NC = 256
NCE_LAYERS = [27, 31, 38, 45, 52, 65]

# name -> (views, patches, channels, feature-map size, segmentation size, label classes)
LOSS_CASES = {
    "p512c256": (2, 512, 256, (16, 16, 16), (32, 32, 32), 8),
    "p64c128": (2, 64, 128, (4, 4, 4), (32, 32, 32), 4),          # layer 31 at a 64^3 input: only 64 voxels
    "p300c32": (2, 300, 32, (8, 12, 16), (32, 48, 64), 5),        # ragged sizes, P not a multiple of 16
}


def blocky_seg(rs: np.random.RandomState, size, classes: int, block: int = 8) -> torch.Tensor:
    """Piecewise-constant labels (so positives exist at every feature scale): [1,1,*size] float."""
    coarse = rs.randint(0, classes, [max(s // block, 1) for s in size])
    seg = np.kron(coarse, np.ones([block] * 3))[: size[0], : size[1], : size[2]]
    return torch.from_numpy(seg.astype(np.float32))[None, None]


def loss_inputs(case: str):
    v, p, c, fsize, ssize, classes = LOSS_CASES[case]
    rs = np.random.RandomState(sum(map(ord, case)))
    feats = torch.from_numpy(rs.randn(v, p, c).astype(np.float32))
    # correlate the two views a little, as aligned views are
    feats[1] = 0.6 * feats[0] + 0.8 * feats[1]
    seg = blocky_seg(rs, ssize, classes)
    coords = torch.from_numpy(np.stack([rs.randint(0, fsize[a], p) for a in range(3)], 1).astype(np.int64))
    return feats, seg, coords, fsize


def sampler_inputs():
    rs = np.random.RandomState(21)
    feats = [torch.from_numpy(rs.randn(2, 32, 16, 16, 16).astype(np.float32)),
             torch.from_numpy(rs.randn(2, 64, 8, 8, 8).astype(np.float32))]
    ids = [torch.from_numpy(np.stack([rs.randint(0, f.shape[2 + a], NUM_PATCHES) for a in range(3)], 1).astype(np.int64))
           for f in feats]
    return feats, ids


def mlp_state_dict(in_channels, seed: int, nc: int = NETF_NC) -> "OrderedDict[str, torch.Tensor]":
    """Parameters + buffers of PatchSampleF's mlp_k (n_mlps = 3), keys as the reference module names them."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for k, cin in enumerate(in_channels):
        pre = f"mlp_{k}."
        sd[pre + "0.weight"] = torch.from_numpy((rs.randn(nc, cin) * np.sqrt(2.0 / cin)).astype(np.float32))
        for bn in (1, 4):
            sd[pre + f"{bn}.weight"] = torch.from_numpy(rs.uniform(0.8, 1.2, nc).astype(np.float32))
            sd[pre + f"{bn}.bias"] = torch.from_numpy((rs.randn(nc) * 0.05).astype(np.float32))
            sd[pre + f"{bn}.running_mean"] = torch.zeros(nc)
            sd[pre + f"{bn}.running_var"] = torch.ones(nc)
            sd[pre + f"{bn}.num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
        sd[pre + "3.weight"] = torch.from_numpy((rs.randn(nc, nc) * np.sqrt(2.0 / nc)).astype(np.float32))
        sd[pre + "6.weight"] = torch.from_numpy((rs.randn(nc, nc) * np.sqrt(2.0 / nc)).astype(np.float32))
        sd[pre + "7.running_mean"] = torch.zeros(nc)
        sd[pre + "7.running_var"] = torch.ones(nc)
        sd[pre + "7.num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
    return sd


def step_inputs(size: int):
    """Two aligned views of one synthetic volume + its label map: A, B [1,1,S,S,S], seg [1,1,S,S,S]."""
    rs = np.random.RandomState(1234)
    seg = blocky_seg(rs, (size,) * 3, 8, block=8)
    base = torch.from_numpy(rs.rand(1, 1, size, size, size).astype(np.float32))
    A = (0.5 * base + 0.06 * seg).clamp(0, 1)
    B = (0.5 * torch.from_numpy(rs.rand(1, 1, size, size, size).astype(np.float32)) + 0.25 * base + 0.04 * seg).clamp(0, 1)
    return A, B, seg


def step_sample_ids(sizes):
    """Captured sample ids (instead of the reference's device randperm): min(512, #voxels) coordinates per layer."""
    rs = np.random.RandomState(77)
    ids = []
    for s in sizes:
        nvox = int(s[0]) * int(s[1]) * int(s[2])
        flat = rs.permutation(nvox)[: min(NUM_PATCHES, nvox)]
        ids.append(torch.from_numpy(np.stack(np.unravel_index(flat, tuple(int(v) for v in s)), 1).astype(np.int64)))
    return ids
