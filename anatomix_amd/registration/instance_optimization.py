"""``merge_features`` with the surface of ``anatomix.registration.instance_optimization`` (reference :16-119): MIND-SSC
descriptors of the two images concatenated in front of the (already down-scaled) network features.  The descriptor runs
on the HIP kernel; the masked branch's distance-transform fill is host logic exactly as in the reference (scipy on the
CPU) and only its MIND-SSC call is accelerated.  The optimisation half of that reference module (run_stage1_registration,
create_warp, run_instance_opt) is the solver, outside the feature path.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .convex_adam_utils import MINDSSC


def _fill_outside_mask(img, mask_vol):
    """instance_optimization.py:59-81: voxels outside the (eroded) mask take the value of the nearest inside voxel, found
    on the 2x subsampled grid with a Euclidean distance transform and brought back with trilinear interpolation."""
    from scipy.ndimage import distance_transform_edt as edt

    h, w, d = img.shape[-3:]
    avg = F.avg_pool3d(F.pad(mask_vol.view(1, 1, h, w, d), (1,) * 6, mode="replicate"), 3, stride=1)
    mask = (avg > 0.9).float()
    _, idx = edt((mask[0, 0, ::2, ::2, ::2] == 0).squeeze().cpu().numpy(), return_indices=True)
    idx = [torch.from_numpy(i).to(img.device).long() for i in idx]
    sub = img[..., ::2, ::2, ::2].reshape(-1)
    # flat index into the subsampled volume with the reference's own integer arithmetic (left to right: ((i0*D)//2*W)//2, :71);
    # equal to i0*(D/2)*(W/2) + i1*(D/2) + i2 on even extents.  An odd extent does not survive the subsample / x2 round trip: the
    # masked assignment below then raises IndexError, as the reference does (fixture flag `odd_dim_raises`).
    flat = idx[0] * d // 2 * w // 2 + idx[1] * d // 2 + idx[2]
    filled = F.interpolate(sub[flat].unsqueeze(0).unsqueeze(0), scale_factor=2, mode="trilinear")
    keep = mask.view(-1) != 0
    filled.view(-1)[keep] = img.reshape(-1)[keep]
    return filled


def merge_features(use_mask, pred_fixed, pred_moving, mask_fixed, mask_moving, fixed_img, moving_img):
    """Returns (mind_fixed, mind_moving, merged_fixed, merged_moving); merged = cat([mind, pred], 1)
    (instance_optimization.py:16-119; MINDSSC(img, 1, 2) as there).  When only the grid_sp-pooled merged features are
    needed, ``convex_adam_utils.smooth_merged_features`` produces them without this full-resolution concat."""
    if use_mask:
        fixed_img = _fill_outside_mask(fixed_img, mask_fixed)
        moving_img = _fill_outside_mask(moving_img, mask_moving)
        pred_fixed = pred_fixed * mask_fixed[None, None, ...]
        pred_moving = pred_moving * mask_moving[None, None, ...]
    mind_fixed = MINDSSC(fixed_img, 1, 2)
    mind_moving = MINDSSC(moving_img, 1, 2)
    return (mind_fixed, mind_moving, torch.cat([mind_fixed, pred_fixed], dim=1), torch.cat([mind_moving, pred_moving], dim=1))
