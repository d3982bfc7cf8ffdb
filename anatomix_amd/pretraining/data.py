"""Two-view HDF5 data loading of the contrastive pretraining: the container read, view
selection, normalisation and cropping of ``H5SupCLDataset`` (pretraining/data/h5supcl_dataset.py:66-104, 185-258, 313-360) and
``random_crop`` (pretraining/data/data_utils.py:81-138), on top of ``anatomix_amd.io.hdf5`` instead of h5py.

Same option names as the reference's ``opt`` namespace (``dataroot, isTrain, data_ndims, load_mask, load_mode, view_order,
crop_size, resize, augment, batch_size``), same random draws in the same order (``torch.randint`` for the views, ``np.random.randint``
for the crop centre), so a seeded loader yields the samples the reference's loader would.  The TorchIO augmentation / resize
branches (h5supcl_dataset.py:106-181, 260-325) are data synthesis and out of scope: they raise.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.utils.data

from ..io.hdf5 import H5File
from ..io.normalize import normalize_img


def random_crop(return_dict, img_keys, crop_size, dimension):
    """data_utils.py:81-160: one random centre per sample, the same window cut out of every tensor named in ``img_keys``."""
    crange = crop_size // 2
    if dimension == 3:
        sx, sy, sz = return_dict["A"].shape[1:]
        cx = np.random.randint(crange, sx - crange) if sx > 2 * crange else crange
        cy = np.random.randint(crange, sy - crange) if sy > 2 * crange else crange
        cz = np.random.randint(crange, sz - crange) if sz > 2 * crange else crange
        for k in img_keys:
            tmp = return_dict[k]
            if len(tmp.shape) == 4:
                return_dict[k] = tmp[:, cx - crange:cx + crange, cy - crange:cy + crange, cz - crange:cz + crange]
            elif len(tmp.shape) == 3:
                return_dict[k] = tmp[cx - crange:cx + crange, cy - crange:cy + crange, cz - crange:cz + crange]
            else:
                raise NotImplementedError("Unexpected data shape for cropping.")
        return return_dict
    if dimension == 2:
        sx, sy = return_dict["A"].shape[1:]
        cx = np.random.randint(crange, sx - crange) if sx > 2 * crange else crange
        cy = np.random.randint(crange, sy - crange) if sy > 2 * crange else crange
        for k in img_keys:
            tmp = return_dict[k]
            if len(tmp.shape) == 3:
                return_dict[k] = tmp[:, cx - crange:cx + crange, cy - crange:cy + crange]
            elif len(tmp.shape) == 2:
                return_dict[k] = tmp[cx - crange:cx + crange, cy - crange:cy + crange]
            else:
                raise NotImplementedError("Unexpected data shape for cropping.")
        return return_dict
    raise NotImplementedError("Only 2D or 3D data supported for cropping.")


class H5SupCLDataset(torch.utils.data.Dataset):
    """``{dataroot}/{train|val}_data.hdf5``: one group per subject with ``img`` = [views, X, Y, Z] and ``seg`` = [X, Y, Z]
    (written by synthetic-data-generation/step3_generate_h5_w_segs.py:28-51).  ``A`` is view i, ``B`` view j."""

    def __init__(self, opt):
        self.opt = opt
        self.folder = opt.dataroot
        self.isTrain = opt.isTrain
        key = "train" if self.isTrain else "val"
        self.dimension = opt.data_ndims
        self.h5_data = self.folder + f"/{key}_data.hdf5"
        self.load_mask = getattr(opt, "load_mask", False)
        self.percentile = 99.99                                # h5supcl_dataset.py:83-84
        self.zero_centered = False
        self.mode = opt.load_mode
        if self.mode != "twoview":
            raise NotImplementedError("Only 'twoview' mode is implemented.")
        if not os.path.exists(self.h5_data):
            raise FileNotFoundError(self.h5_data)
        with H5File(self.h5_data, "r") as f:                   # the length is needed up front; the file is reopened per item
            self.subj_id = list(f.keys())
            self.len = len(self.subj_id)
        self.load_seg = True
        self.crop_size = opt.crop_size
        if getattr(opt, "resize", False):
            raise NotImplementedError("opt.resize (TorchIO Resize, h5supcl_dataset.py:109-117) is not part of the accelerated path")
        if self.isTrain and getattr(opt, "augment", False):
            raise NotImplementedError("opt.augment (TorchIO augmentation, h5supcl_dataset.py:121-177) is not part of the accelerated path")

    def __getitem__(self, item):
        return_dict = dict()
        with H5File(self.h5_data, "r") as hf:
            while item >= len(self.subj_id):                   # __len__ may exceed the subject count (batch_size)
                item = torch.randint(0, self.len, ()).numpy()
            assert self.dimension == 3, f"Only support 3D data loading in mode {self.mode}"
            subj = self.subj_id[item]
            img = hf[subj]["img"]
            n_tps_per_subj = img.shape[0]
            if self.opt.view_order:
                i = torch.randint(0, n_tps_per_subj - 1, ()).numpy()
                j = i + 1
            else:
                i = torch.randint(0, n_tps_per_subj, ()).numpy()
                j = torch.randint(0, n_tps_per_subj, ()).numpy()
                while j == i:
                    j = torch.randint(0, n_tps_per_subj, ()).numpy()
            img_keys = ["A", "B", "A_seg", "B_seg"]
            A_orig = normalize_img(img[int(i)], percentile=self.percentile, zero_centered=self.zero_centered)[None, ...]
            return_dict["A_id"] = np.asarray([item])
            B_orig = normalize_img(img[int(j)], percentile=self.percentile, zero_centered=self.zero_centered)[None, ...]
            return_dict["meta"] = "%s" % (subj,)
            return_dict["B_id"] = np.asarray([item])
            AB_seg_orig = np.array(hf[subj]["seg"])
            return_dict["A"] = torch.from_numpy(A_orig).float()
            return_dict["B"] = torch.from_numpy(B_orig).float()
            return_dict["A_seg"] = torch.from_numpy(AB_seg_orig[np.newaxis, ...]).float()
            return_dict["B_seg"] = torch.from_numpy(AB_seg_orig[np.newaxis, ...]).float()
            if self.load_mask:
                raise NotImplementedError("Mask loading is not implemented.")
        return_dict["keys"] = img_keys
        if self.crop_size > 0 and self.opt.isTrain and (not getattr(self.opt, "resize", False)):
            return_dict = random_crop(return_dict, img_keys, self.crop_size, self.dimension)
        return return_dict

    def __len__(self):
        return max(self.len, self.opt.batch_size)
