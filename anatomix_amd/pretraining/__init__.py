"""MI355X-side mirror of the reference's contrastive-pretraining pieces (reference: pretraining/models/).

  SupPatchNCELoss ... pretraining/models/supcl_model.py:16-226     (HIP forward+backward kernel, amx_supcon_loss)
  PatchSampleF ...... pretraining/models/pretraining_networks.py:264-519  (projection heads: HIP forward + backward,
                      amx_mlp_head_forward / _backward)
  contrastive_step .. the per-batch body of SupCLModel.optimize_parameters / forward / calculate_NCE_loss
                      (supcl_model.py:603-661, 723-843) without the option parsing / logging around it.
  FusedAdamW ........ torch.optim.AdamW as built at supcl_model.py:510-516, 584-590: one HIP launch per optimizer (amx_adamw_step)
The UNet inside the step runs forward and backward on the HIP kernels (anatomix_amd.model.train).
"""
from .supcon import SupPatchNCELoss
from .patch_sample import PatchSampleF
from .step import contrastive_step, GraphedContrastiveStep, StepRecord
from .data_parallel import GradientBuckets
from .optim import FusedAdamW
from .data import H5SupCLDataset, random_crop

__all__ = ["SupPatchNCELoss", "PatchSampleF", "contrastive_step", "GraphedContrastiveStep", "StepRecord", "GradientBuckets", "FusedAdamW", "H5SupCLDataset", "random_crop"]
