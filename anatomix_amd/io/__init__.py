"""On-disk adapters around the feature path: NIfTI volumes in / out (what the reference reads and writes
through nibabel in anatomix/registration/run_convex_adam_with_network_feats.py:125-147, 270-325), the intensity
normalisation of the pretraining data loader (pretraining/data/data_utils.py:4-46) and checkpoint loading
(``anatomix_amd.model.load_from_hf``), and a read-only HDF5 reader for the two-view pretraining containers
(pretraining/data/h5supcl_dataset.py:100-102,208-258; no h5py needed).  Host-side numpy only."""
from .hdf5 import H5File  # noqa: F401
from .nifti import load_nifti, save_nifti  # noqa: F401
from .normalize import normalize_img  # noqa: F401
