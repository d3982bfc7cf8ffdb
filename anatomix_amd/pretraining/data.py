"""Two-view HDF5 sample assembly for the contrastive pretraining.

Behavioural counterpart of ``H5SupCLDataset`` (pretraining/data/h5supcl_dataset.py:66-104, 185-258, 313-372) and
``random_crop`` (pretraining/data/data_utils.py:81-175), reading the container through ``anatomix_amd.io.hdf5`` (no h5py).

What is kept from the reference is the CONTRACT, not the text: the option names of its ``opt`` namespace (``dataroot, isTrain,
data_ndims, load_mask, load_mode, view_order, crop_size, resize, augment, batch_size``), the keys and dtypes of the sample
dictionary, and the order in which random numbers are consumed (``torch.randint`` for the subject redraw and the views, then
one ``np.random.randint`` per spatial axis for the crop centre) -- so a seeded loader yields the samples the reference's loader
yields.  The TorchIO resize / augmentation branches (h5supcl_dataset.py:106-181, 260-325) are data synthesis, out of scope: they
raise.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.utils.data

from ..io.hdf5 import H5File
from ..io.normalize import normalize_img

_SAMPLE_KEYS = ("A", "B", "A_seg", "B_seg")


def _crop_window(extent, half):
    """One axis of the crop: a window of 2*half voxels around a random centre.  The centre is only DRAWN when the axis is longer
    than the window (data_utils.py:113-127) -- an axis that already fits consumes no random number."""
    centre = np.random.randint(half, extent - half) if extent > 2 * half else half
    return slice(centre - half, centre + half)


def random_crop(return_dict, img_keys, crop_size, dimension):
    """The same random window cut out of every tensor named in ``img_keys`` (data_utils.py:81-175).

    The window is drawn from the spatial extent of ``return_dict["A"]`` (channel axis first), one draw per axis in axis order.
    Entries may carry a leading channel axis (ndim == dimension + 1) or not (ndim == dimension)."""
    if dimension not in (2, 3):
        raise NotImplementedError("Only 2D or 3D data supported for cropping.")
    half = crop_size // 2
    spatial = tuple(return_dict["A"].shape[1:1 + dimension])
    if len(spatial) != dimension:
        raise NotImplementedError("Unexpected data shape for cropping.")
    window = tuple(_crop_window(n, half) for n in spatial)
    for name in img_keys:
        lead = return_dict[name].ndim - dimension
        if lead not in (0, 1):
            raise NotImplementedError("Unexpected data shape for cropping.")
        return_dict[name] = return_dict[name][(slice(None),) * lead + window]
    return return_dict


def _scalar_randint(high, low=0):
    """A 0-d ``torch.randint`` as a Python int: the reference draws its indices from torch's global generator this way."""
    return int(torch.randint(low, high, ()))


def _draw_views(n_views, consecutive):
    """Indices (i, j) of the two views of one subject (h5supcl_dataset.py:222-231).  ``consecutive``: j follows i; otherwise two
    independent draws, the second repeated until it differs from the first."""
    if consecutive:
        first = _scalar_randint(n_views - 1)
        return first, first + 1
    first, second = _scalar_randint(n_views), _scalar_randint(n_views)
    while second == first:
        second = _scalar_randint(n_views)
    return first, second


class H5SupCLDataset(torch.utils.data.Dataset):
    """``{dataroot}/{train|val}_data.hdf5``: one group per subject with ``img`` = [views, X, Y, Z] and ``seg`` = [X, Y, Z]
    (the container synthetic-data-generation/step3_generate_h5_w_segs.py:28-51 writes).  ``A`` / ``B`` are two views of the same
    subject and share its label map."""

    percentile = 99.99                 # h5supcl_dataset.py:83-84
    zero_centered = False

    def __init__(self, opt):
        if opt.load_mode != "twoview":
            raise NotImplementedError("Only 'twoview' mode is implemented.")
        if getattr(opt, "resize", False):
            raise NotImplementedError("opt.resize (TorchIO Resize, h5supcl_dataset.py:109-117) is not part of the accelerated path")
        if opt.isTrain and getattr(opt, "augment", False):
            raise NotImplementedError("opt.augment (TorchIO augmentation, h5supcl_dataset.py:121-177) is not part of the accelerated path")
        self.opt = opt
        self.mode = opt.load_mode
        self.isTrain = bool(opt.isTrain)
        self.dimension = opt.data_ndims
        self.folder = opt.dataroot
        self.h5_data = "%s/%s_data.hdf5" % (opt.dataroot, "train" if self.isTrain else "val")
        self.load_mask = getattr(opt, "load_mask", False)
        self.load_seg = True
        self.crop_size = opt.crop_size
        if not os.path.exists(self.h5_data):
            raise FileNotFoundError(self.h5_data)
        with H5File(self.h5_data, "r") as container:     # subjects in the library's name order; the file is reopened per sample
            self.subj_id = list(container.keys())
        self.len = len(self.subj_id)

    def __len__(self):
        return max(self.len, self.opt.batch_size)        # h5supcl_dataset.py:363-372: never shorter than one batch

    def _read_subject(self, index):
        """(name, view A, view B, label map) of subject ``index`` as numpy arrays; the views normalised to [0, 1]."""
        name = self.subj_id[index]
        with H5File(self.h5_data, "r") as container:
            views, labels = container[name]["img"], container[name]["seg"]
            a, b = _draw_views(views.shape[0], bool(self.opt.view_order))
            pair = [normalize_img(views[k], percentile=self.percentile, zero_centered=self.zero_centered) for k in (a, b)]
            return name, pair[0], pair[1], np.array(labels)

    def __getitem__(self, item):
        if self.dimension != 3:
            raise AssertionError(f"Only support 3D data loading in mode {self.mode}")
        if self.load_mask:
            raise NotImplementedError("Mask loading is not implemented.")
        while item >= self.len:                          # __len__ may exceed the subject count: redraw (h5supcl_dataset.py:212-213)
            item = _scalar_randint(self.len)
        name, view_a, view_b, labels = self._read_subject(item)
        seg = torch.from_numpy(labels[None]).float()
        sample = {
            "A": torch.from_numpy(view_a[None]).float(),
            "B": torch.from_numpy(view_b[None]).float(),
            "A_seg": seg,
            "B_seg": seg.clone(),
            "A_id": np.asarray([item]),
            "B_id": np.asarray([item]),
            "meta": str(name),
            "keys": list(_SAMPLE_KEYS),
        }
        if self.crop_size > 0 and self.isTrain:
            sample = random_crop(sample, sample["keys"], self.crop_size, self.dimension)
        return sample
