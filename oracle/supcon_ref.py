"""Oracle: CPU restatement of the reference's supervised patch contrastive loss and patch sampler.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against the imported reference by
``oracle/make_golden_pretrain.py`` (build container) -> ``tests/golden/pretrain_golden.npz``.

Reference lines restated (paths relative to /root/reference/pretraining):
  * SupPatchNCELoss.forward ............ models/supcl_model.py:73-226
      labels: nearest-resize of the segmentation to the feature map size, gathered at the sampled coordinates
      (100-112); positives mask eq(labels, labels^T) tiled over the two views (131, 152); cosine Gram matrix of the
      2P anchors with F.normalize(eps=1e-8), divided by the temperature (60-71, 144-147); row max subtracted
      (150-151); the diagonal leaves the partition function and the positives (160-170); either the plain
      log-partition (203-204) or the class-balanced one (172-196); minus the mean log-probability of the positives
      (209-212), plain mean over anchors or rarity-weighted mean (213-224).
  * PatchSampleF.forward ............... models/pretraining_networks.py:363-519
      gather feat[:, :, x, y, z] at the sampled coordinates -> [views * P, C], then the per-layer MLP
      Linear(C,256,no bias)-BN1d-ReLU-Linear-BN1d-ReLU-Linear-BN1d(affine=False) (n_mlps = 3, 338-350).
All arithmetic is torch on CPU; ``dtype`` float64 gives the yardstick the HIP kernels are measured against.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def gather_labels(seg: torch.Tensor, coords: torch.Tensor, size) -> torch.Tensor:
    """supcl_model.py:100-112 -- seg [1,1,H,W,D] -> labels [1,P] at the feature-map resolution ``size``."""
    s = F.interpolate(seg.float(), size=tuple(size), mode="nearest")
    return s.squeeze(1)[:, coords[:, 0], coords[:, 1], coords[:, 2]]


def supcon_loss(features: torch.Tensor, labels: torch.Tensor, temperature: float, weigh_rarity: bool = False,
                balance_denominator: bool = False, weighting_mode: str = "raw", dtype=torch.float64) -> torch.Tensor:
    """features [views, P, C]; labels [1, P] (one segmentation, shared by the views).  Returns the scalar loss."""
    v, p, c = features.shape
    x = features.reshape(v * p, c).to(dtype)
    xn = x / x.norm(dim=1, keepdim=True).clamp_min(1e-8)              # F.normalize(eps=1e-8)
    logits = xn @ xn.t() / temperature
    logits = logits - logits.max(dim=1, keepdim=True).values.detach()
    same = torch.eq(labels, labels.t()).to(dtype).repeat(v, v)        # incl. the diagonal
    class_counts = same.sum(1)
    off_diag = 1.0 - torch.eye(v * p, dtype=dtype)
    pos = same * off_diag
    if balance_denominator:
        n_per_class = class_counts.unsqueeze(0) - same
        if weighting_mode == "sqrt":
            n_per_class = n_per_class.sqrt()
        log_w = torch.log(off_diag / n_per_class)
        log_prob = logits - torch.logsumexp(logits + log_w, dim=1, keepdim=True)
    else:
        log_prob = logits - torch.log((torch.exp(logits) * off_diag).sum(1, keepdim=True))
    loss = -(pos * log_prob).sum(1) / pos.sum(1)
    if weigh_rarity:
        counts = class_counts.sqrt() if weighting_mode == "sqrt" else class_counts
        w = 1.0 / counts
        return (w * loss).sum() / w.sum()
    return loss.mean()


def sample_features(feat: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """pretraining_networks.py:438-447,497-500 -- feat [views,C,X,Y,Z], coords [P,3] -> [views*P, C]."""
    xs = feat[:, :, coords[:, 0], coords[:, 1], coords[:, 2]]        # [views, C, P]
    return xs.permute(0, 2, 1).flatten(0, 1)


def mlp_forward(x: torch.Tensor, sd: dict, prefix: str, eps: float = 1e-5) -> torch.Tensor:
    """The n_mlps = 3 projection head in TRAIN mode (batch statistics), parameters from a state_dict."""
    def bn(t, w=None, b=None):
        return F.batch_norm(t, None, None, w, b, True, 0.1, eps)
    h = F.linear(x, sd[prefix + "0.weight"])
    h = F.relu(bn(h, sd[prefix + "1.weight"], sd[prefix + "1.bias"]))
    h = F.linear(h, sd[prefix + "3.weight"])
    h = F.relu(bn(h, sd[prefix + "4.weight"], sd[prefix + "4.bias"]))
    h = F.linear(h, sd[prefix + "6.weight"])
    return bn(h)
