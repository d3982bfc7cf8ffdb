"""HDF5 container reading without h5py (anatomix_amd/io/hdf5.py) and the two-view dataset on top of it
(anatomix_amd/pretraining/data.py; reference: pretraining/data/h5supcl_dataset.py:66-104,185-360).

The fixtures were written AND read back by the real library (oracle/make_golden_hdf5.py: h5py 3.3.0 / HDF5 1.10.6), so every
comparison below is against what h5py returns for the same file."""
import os
import shutil
from argparse import Namespace

import numpy as np
import pytest
import torch

from anatomix_amd.io import H5File, normalize_img
from anatomix_amd.io.hdf5 import H5Dataset, H5FormatError, H5Group
from anatomix_amd.pretraining import H5SupCLDataset, random_crop

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = {"two_view": "two_view_train_data.hdf5", "variants": "hdf5_variants.hdf5", "latest": "hdf5_latest.hdf5"}


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(GOLD, "hdf5_expected.npz"))


@pytest.mark.parametrize("tag", sorted(FILES))
def test_every_dataset_equals_h5py_readback(tag, expected):
    names = [k.split(":", 1)[1] for k in expected.files if k.startswith(tag + ":") and not k.endswith("__keys__")]
    assert names
    with H5File(os.path.join(GOLD, FILES[tag])) as f:
        assert list(f.keys()) == [str(k) for k in expected[f"{tag}:__keys__"]]        # h5py order: by name
        for name in names:
            want = expected[f"{tag}:{name}"]
            ds = f[name]
            assert isinstance(ds, H5Dataset)
            got = ds[...]
            assert ds.shape == want.shape and got.shape == want.shape, name
            assert got.dtype.newbyteorder("=") == want.dtype.newbyteorder("="), name
            assert np.array_equal(got, want, equal_nan=True), name                      # bit-exact
            assert np.array_equal(np.array(ds), want, equal_nan=True)


def test_reference_container_layout(expected):
    """step3_generate_h5_w_segs.py:28-51: groups "%06d" with img = uint8 [2, X, Y, Z] and seg = uint8 [X, Y, Z]."""
    with H5File(os.path.join(GOLD, FILES["two_view"]), "r", libver="latest", swmr=False) as hf:     # the loader's call, :208
        assert list(hf.keys()) == ["%06d" % i for i in range(6)] and len(hf) == 6
        for subj in hf:
            g = hf[subj]
            assert isinstance(g, H5Group) and sorted(g.keys()) == ["img", "seg"]
            img, seg = g["img"], g["seg"]
            assert img.dtype == np.uint8 and seg.dtype == np.uint8 and img.shape == (2,) + seg.shape
            want = expected[f"two_view:{subj}/img"]
            for i in range(2):
                assert np.array_equal(img[i], want[i])                                    # hf[subj]["img"][i], :238-247
            assert np.array_equal(img[-1], want[1]) and np.array_equal(img[0:1], want[0:1])
            assert np.array_equal(img[1, 2:5, :, 3], want[1, 2:5, :, 3])
            assert np.array_equal(np.array(seg), expected[f"two_view:{subj}/seg"])       # np.array(hf[subj]["seg"]), :258


def test_groups_paths_and_errors(tmp_path, expected):
    with H5File(os.path.join(GOLD, FILES["variants"])) as f:
        assert len(f["many"]) == 300 and list(f["many"].keys()) == ["s%04d" % i for i in range(300)]
        assert np.array_equal(f["many/s0123"][...], np.arange(123, 126))
        assert np.array_equal(f["outer"]["inner"]["deep"][...], f["/outer/inner/deep"][...])
        assert "outer" in f and "nope" not in f
        assert f["scalar"].shape == () and f["scalar"][()] == np.float32(2.5)
        assert f["unwritten"][...].tolist() == [[0.0] * 4] * 3                            # never written: fill value
        gz = f["gz"]
        assert np.array_equal(gz[1], expected["variants:gz"][1]) and np.array_equal(gz[0:2, 3], expected["variants:gz"][0:2, 3])
        with pytest.raises(KeyError):
            f["nope"]
        with pytest.raises(IndexError):
            f["f32"][5]
    bad = tmp_path / "not.hdf5"
    bad.write_bytes(b"\x00" * 4096)
    with pytest.raises(H5FormatError):
        H5File(str(bad))
    cut = tmp_path / "cut.hdf5"
    cut.write_bytes(open(os.path.join(GOLD, FILES["two_view"]), "rb").read()[:3000])
    with pytest.raises((H5FormatError, KeyError)):
        with H5File(str(cut)) as f:
            f["000005"]["img"][...]
    with pytest.raises(ValueError):
        H5File(os.path.join(GOLD, FILES["two_view"]), "w")


def _opt(root, **kw):
    o = dict(dataroot=str(root), isTrain=True, data_ndims=3, load_mask=False, load_mode="twoview", view_order=False, crop_size=0,
             resize=False, augment=False, batch_size=1)
    o.update(kw)
    return Namespace(**o)


@pytest.fixture()
def dataroot(tmp_path):
    shutil.copy(os.path.join(GOLD, FILES["two_view"]), tmp_path / "train_data.hdf5")
    shutil.copy(os.path.join(GOLD, FILES["two_view"]), tmp_path / "val_data.hdf5")
    return tmp_path


def test_two_view_dataset_samples(dataroot, expected):
    ds = H5SupCLDataset(_opt(dataroot, batch_size=4))
    assert ds.subj_id == ["%06d" % i for i in range(6)] and len(ds) == 6
    assert len(H5SupCLDataset(_opt(dataroot, batch_size=32))) == 32                       # max(len, batch_size), :362-372
    for item in (0, 3, 5):
        torch.manual_seed(100 + item)
        s = ds[item]
        # replay the reference's draws (:222-232): i, j ~ randint(0, 2), j redrawn until it differs
        torch.manual_seed(100 + item)
        i = int(torch.randint(0, 2, ()))
        j = int(torch.randint(0, 2, ()))
        while j == i:
            j = int(torch.randint(0, 2, ()))
        img = expected["two_view:%06d/img" % item]
        seg = expected["two_view:%06d/seg" % item]
        A = normalize_img(img[i], percentile=99.99, zero_centered=False)[None]
        B = normalize_img(img[j], percentile=99.99, zero_centered=False)[None]
        assert s["keys"] == ["A", "B", "A_seg", "B_seg"] and s["meta"] == "%06d" % item
        assert torch.equal(s["A"], torch.from_numpy(A).float()) and torch.equal(s["B"], torch.from_numpy(B).float())
        assert s["A"].dtype == torch.float32 and s["A"].shape == (1,) + seg.shape
        assert torch.equal(s["A_seg"], torch.from_numpy(seg[None]).float()) and torch.equal(s["B_seg"], s["A_seg"])
        assert s["A_id"].tolist() == [item] and s["B_id"].tolist() == [item]
        assert 0.0 <= float(s["A"].min()) and float(s["A"].max()) >= 1.0                  # percentile 99.99: the top values exceed 1


def test_two_view_dataset_order_wrap_crop_and_val(dataroot, expected):
    ds = H5SupCLDataset(_opt(dataroot, view_order=True))
    s = ds[2]
    img = expected["two_view:000002/img"]
    assert torch.equal(s["A"][0], torch.from_numpy(normalize_img(img[0], percentile=99.99, zero_centered=False)).float())
    assert torch.equal(s["B"][0], torch.from_numpy(normalize_img(img[1], percentile=99.99, zero_centered=False)).float())
    # an index beyond the subject count is redrawn (:211-213)
    big = H5SupCLDataset(_opt(dataroot, batch_size=16, view_order=True))
    torch.manual_seed(7)
    s = big[11]
    torch.manual_seed(7)
    item = int(torch.randint(0, 6, ()))
    assert s["meta"] == "%06d" % item and s["A_id"].tolist() == [item]
    # random crop: one centre per sample, the same window in all four tensors (data_utils.py:98-137)
    crop = H5SupCLDataset(_opt(dataroot, crop_size=6, view_order=True))
    np.random.seed(3)
    s = crop[0]
    np.random.seed(3)
    sx, sy, sz = 12, 10, 8
    cx, cy, cz = np.random.randint(3, sx - 3), np.random.randint(3, sy - 3), np.random.randint(3, sz - 3)
    full = H5SupCLDataset(_opt(dataroot, view_order=True))[0]
    for k in ("A", "B", "A_seg", "B_seg"):
        assert s[k].shape == (1, 6, 6, 6)
        assert torch.equal(s[k], full[k][:, cx - 3:cx + 3, cy - 3:cy + 3, cz - 3:cz + 3])
    # validation split: other file, never cropped (:357)
    val = H5SupCLDataset(_opt(dataroot, isTrain=False, crop_size=6, view_order=True))
    assert val.h5_data.endswith("val_data.hdf5") and val[0]["A"].shape == (1, 12, 10, 8)
    # volumes smaller than the crop keep the reference's degenerate window (centre = crange)
    d = {"A": torch.zeros(1, 4, 4, 4), "B": torch.zeros(1, 4, 4, 4)}
    assert random_crop(d, ["A", "B"], 8, 3)["A"].shape == (1, 4, 4, 4)


def test_two_view_dataset_refuses_what_is_out_of_scope(dataroot, tmp_path):
    with pytest.raises(NotImplementedError):
        H5SupCLDataset(_opt(dataroot, resize=True))
    with pytest.raises(NotImplementedError):
        H5SupCLDataset(_opt(dataroot, augment=True))
    with pytest.raises(NotImplementedError):
        H5SupCLDataset(_opt(dataroot, load_mode="single"))
    with pytest.raises(FileNotFoundError):
        H5SupCLDataset(_opt(tmp_path / "missing"))
    loader = torch.utils.data.DataLoader(H5SupCLDataset(_opt(dataroot, crop_size=6)), batch_size=2, num_workers=0)
    batch = next(iter(loader))
    assert batch["A"].shape == (2, 1, 6, 6, 6) and batch["A_seg"].shape == (2, 1, 6, 6, 6)
