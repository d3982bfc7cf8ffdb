"""PatchSampleF with the reference's constructor and call contract (pretraining_networks.py:264-519)."""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..model.network import get_actvn_layer, get_norm_layer
from . import mlp_head


class PatchSampleF(nn.Module):
    """Samples ``num_patches`` voxel coordinates per feature map (the SAME coordinates for every view in the batch),
    gathers ``feat[:, :, x, y, z]`` -> ``[views * P, C]`` and projects it with a lazily created per-layer MLP
    ``mlp_k`` (Linear(C, nc, no bias) - norm - act [- Linear - norm - act] - Linear - norm(affine=False)).

    Differences from the reference, both deliberate: tensors stay on the device of ``feats`` (the reference
    hard-codes ``.cuda()``, pretraining_networks.py:400,405), and the given-``patch_ids`` branch returns the given
    coordinates (the reference leaves ``coords`` unbound there, :432-447,499)."""

    def __init__(self, use_mlp=False, init_type="normal", init_gain=0.02, nc=256, gpu_ids=[], n_mlps=2, activation="relu",
                 norm="batch", norm_eps=1e-5):
        super().__init__()
        self.use_mlp = use_mlp
        print("Use MLP: {}".format(use_mlp))
        self.nc = nc
        self.mlp_init = False
        self.init_type, self.init_gain, self.gpu_ids = init_type, init_gain, gpu_ids
        self.n_mlps, self.activation, self.normtype, self.norm_eps = n_mlps, activation, norm, norm_eps

    def create_mlp(self, feats):
        for mlp_id, feat in enumerate(feats):
            cin = feat.shape[1]
            norm = get_norm_layer(1, self.normtype, eps=self.norm_eps)
            act = get_actvn_layer(self.activation)
            if self.n_mlps == 2:
                layers = [nn.Linear(cin, self.nc, bias=False), norm(self.nc), act,
                          nn.Linear(self.nc, self.nc, bias=False), norm(self.nc, affine=False)]
            elif self.n_mlps == 3:
                layers = [nn.Linear(cin, self.nc, bias=False), norm(self.nc), act,
                          nn.Linear(self.nc, self.nc, bias=False), norm(self.nc), act,
                          nn.Linear(self.nc, self.nc, bias=False), norm(self.nc, affine=False)]
            else:
                raise NotImplementedError
            mlp = nn.Sequential(*layers).to(feat.device)
            setattr(self, "mlp_%d" % mlp_id, mlp)
            print("mlp_%d created, input nc %d" % (mlp_id, cin))
        self._init_weights()
        self.mlp_init = True

    def _init_weights(self):
        """init_net -> init_weights (pretraining_networks.py:666-715): Linear / Conv weights by ``init_type``; the
        BatchNorm1d layers are NOT matched by its class-name test (only 'BatchNorm3d' is) and keep gamma=1, beta=0."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                if self.init_type == "normal":
                    nn.init.normal_(m.weight, 0.0, self.init_gain)
                elif self.init_type == "xavier":
                    nn.init.xavier_normal_(m.weight, gain=self.init_gain)
                elif self.init_type == "kaiming":
                    nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in")
                elif self.init_type == "orthogonal":
                    nn.init.orthogonal_(m.weight, gain=self.init_gain)
                else:
                    raise NotImplementedError("initialization method [%s] is not implemented" % self.init_type)

    @staticmethod
    def _sample_perm(device, num, dims):
        """``randperm(nvox)[:num]`` on a small grid (<= 4096 voxels): one random key per voxel from torch's generator, ordered and
        unravelled by one HIP kernel (amx_sample_perm) instead of torch.randperm's dozen launches."""
        import ctypes
        from .. import _lib
        lib = _lib.load()
        d = [1] * (3 - len(dims)) + [int(v) for v in dims]
        keys = torch.randint(1 << 62, (d[0] * d[1] * d[2],), device=device, dtype=torch.int64)
        coords = torch.empty((num, 3), dtype=torch.int64, device=device)
        with torch.cuda.device(device):
            st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            _lib.check(lib.amx_sample_perm(_lib.ptr(keys), d[0], d[1], d[2], num, _lib.ptr(coords), st))
        return coords[:, 3 - len(dims):]

    @staticmethod
    def _sample_distinct(device, nvox, num, dims):
        """``randperm(nvox)[:num]`` without sorting every voxel: 2 * num draws with replacement from torch's generator, then
        the first ``num`` distinct ones in draw order (same distribution) unravelled by one HIP kernel (amx_sample_coords)."""
        import ctypes
        from .. import _lib
        lib = _lib.load()
        draws = torch.randint(nvox, (2 * num,), device=device, dtype=torch.int64)
        d = [1] * (3 - len(dims)) + [int(v) for v in dims]
        coords = torch.empty((num, 3), dtype=torch.int64, device=device)
        with torch.cuda.device(device):
            st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            _lib.check(lib.amx_sample_coords(_lib.ptr(draws), 2 * num, num, d[0], d[1], d[2], _lib.ptr(coords), st))
        return coords[:, 3 - len(dims):]

    def forward(self, feats, num_patches=64, patch_ids=None, mask=None, verbose=False, streams=None):
        """``streams`` (extension, CUDA only): one stream per feature map; layer k's sampling and head then run on streams[k]
        (which first waits for the caller's stream), so the independent per-layer chains can overlap.  The outputs stay on
        those streams: the caller continues each layer's work there and joins the streams itself."""
        return_ids, return_feats = [], []
        ndims = feats[0].dim() - 2
        if ndims not in (2, 3):
            raise NotImplementedError
        if self.use_mlp and not self.mlp_init:
            self.create_mlp(feats)
        ambient = torch.cuda.current_stream(feats[0].device) if streams is not None else None
        for k, feat in enumerate(feats):
            if streams is not None:
                streams[k].wait_stream(ambient)
            with (torch.cuda.stream(streams[k]) if streams is not None else contextlib.nullcontext()):
                x_sample, coords = self._one_layer(k, feat, num_patches, patch_ids, mask)
            return_ids.append(coords)
            return_feats.append(x_sample)
        return return_feats, return_ids

    def draw_coords(self, k, dims, num_patches, patch_ids, device):
        """The no-mask sampling of feature map k (pretraining_networks.py:443-470): the given ids, or ``num_patches`` distinct
        voxels of a grid of ``dims`` -- consumes torch's generator exactly as ``forward`` does for that layer."""
        if patch_ids is not None:
            return patch_ids[k].to(device)
        # every voxel of the grid is a candidate: the k-th entry of torch.where(all-ones) is the C-order
        # unravelling of k (no host round trip anywhere on this path)
        dims = [int(v) for v in dims]
        ndims = len(dims)
        nvox = 1
        for v in dims:
            nvox *= v
        num = int(min(num_patches, nvox))
        if device.type == "cuda" and nvox >= 8 * num and 2 * num <= 4096:
            return self._sample_distinct(device, nvox, num, dims)
        if device.type == "cuda" and nvox <= 4096 and ndims <= 3:
            return self._sample_perm(device, num, dims)
        flat = torch.randperm(nvox, device=device)[:num]
        cs = []
        for a in range(ndims - 1, -1, -1):
            cs.append(flat % dims[a])
            flat = torch.div(flat, dims[a], rounding_mode="floor")
        return torch.stack(cs[::-1], dim=1)

    def forward_rows(self, rows, coords, streams=None, batched=False):
        """The heads on rows that were gathered elsewhere (``model.train.forward_train_sampled``): ``rows[k]`` [views, P, C] at
        ``coords[k]`` -- what ``forward`` computes from the dense feature maps, without the dense feature maps.  ``batched``: all
        heads as ONE chain of launches when they share a structure (``mlp_head.run_heads``; same values), else layer by layer."""
        if self.use_mlp and not self.mlp_init:
            self.create_mlp([torch.zeros((1, r.shape[2], 1, 1, 1), device=r.device) for r in rows])
        if batched and self.use_mlp:
            mlps = [getattr(self, "mlp_%d" % k) for k in range(len(rows))]
            xs = [r.flatten(0, 1) for r in rows]
            if all(x.is_cuda for x in xs) and mlp_head.batchable(mlps, xs) is None:
                ys = mlp_head.run_heads(mlps, xs)
                return [y.view(r.shape[0], r.shape[1], -1) for y, r in zip(ys, rows)], list(coords)
        ambient = torch.cuda.current_stream(rows[0].device) if streams is not None else None
        out = []
        forked = set()
        for k, r in enumerate(rows):
            if streams is not None and id(streams[k]) not in forked:      # (layers may share a stream: one fork per distinct stream)
                streams[k].wait_stream(ambient)
                forked.add(id(streams[k]))
            with (torch.cuda.stream(streams[k]) if streams is not None else contextlib.nullcontext()):
                out.append(self._head(k, r.flatten(0, 1), r.shape[0], r.shape[1]))
        return out, list(coords)

    def _one_layer(self, k, feat, num_patches, patch_ids, mask):
        """Sampling + gather + head of feature map k (pretraining_networks.py:432-511)."""
        ndims = feat.dim() - 2
        if num_patches > 0:
            if patch_ids is None and mask is not None:
                m = F.interpolate(mask, size=feat.shape[2:], mode="nearest").to(feat.device)
                fg = torch.where(m > 0)[2:]
                perm = torch.randperm(fg[0].shape[0], device=feat.device)[: int(min(num_patches, fg[0].shape[0]))]
                coords = torch.stack([f[perm] for f in fg], dim=1)
            else:
                coords = self.draw_coords(k, feat.shape[2:], num_patches, patch_ids, feat.device)
            idx = (slice(None), slice(None)) + tuple(coords[:, a] for a in range(ndims))
            x_sample = feat[idx]                                   # [views, C, P]
        else:
            x_sample, coords = feat.flatten(2), []
        nviews, nc, nsample = x_sample.size()
        x_sample = x_sample.permute(0, 2, 1).flatten(0, 1)         # [views * P, C]
        return self._head(k, x_sample, nviews, nsample), coords

    def _head(self, k, x_sample, nviews, nsample):
        if self.use_mlp:
            mlp = getattr(self, "mlp_%d" % k)
            # train-mode heads of the structure the reference builds run on the HIP kernels (one call per head and
            # direction); anything else -- eval mode, CPU tensors, exotic shapes -- goes through the stock modules
            if x_sample.is_cuda and mlp_head.unsupported_reason(mlp, x_sample) is None:
                x_sample = mlp_head.run_head(mlp, x_sample).view(nviews, nsample, -1)
            else:
                x_sample = mlp(x_sample).view(nviews, nsample, -1)
        return x_sample
