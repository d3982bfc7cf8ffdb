"""``normalize_img`` of the pretraining data loader (pretraining/data/data_utils.py:4-46; used with percentile 99.99 on the
uint8 views of the HDF5 two-view dataset, pretraining/data/h5supcl_dataset.py:238-252)."""
import numpy as np


def normalize_img(array, percentile=None, zero_centered=True, verbose=False):
    """(array - min) / (upper - min) with upper = max or the given percentile (values above it exceed 1; a flat image is left
    as it is), then mapped to [-1, 1] when ``zero_centered``."""
    lo = np.min(array)
    hi = np.percentile(array, percentile) if percentile is not None else np.max(array)
    if verbose:
        print("original range: {},{}".format(lo, hi))
    if hi - lo > 0:
        array = (array - lo) / (hi - lo)
    if zero_centered:
        array = array * 2 - 1
    if verbose:
        print("normalized to range {}, {}".format(np.min(array), np.max(array)))
    return array
