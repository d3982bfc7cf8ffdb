"""Writes the HDF5 fixtures of tests/test_hdf5.py with the REAL library and records what h5py reads back.

Run with an interpreter that has h5py (the build container ships one outside the default environment):
    /opt/conda/bin/python3.9 oracle/make_golden_hdf5.py
(h5py 3.3.0 / HDF5 1.10.6 there).  Test infrastructure only.

  tests/golden/two_view_train_data.hdf5   the reference's container layout, written the way its generator writes it
                                          (synthetic-data-generation/step3_generate_h5_w_segs.py:28-51: one group "%06d" per
                                          subject, grp["img"] = uint8 [2, X, Y, Z], grp["seg"] = uint8 [X, Y, Z]); 6 subjects
  tests/golden/hdf5_variants.hdf5         what else a user-made container may hold: a group with 300 links (several symbol-table nodes and
                                          B-tree levels), float / int16 / big-endian / scalar datasets, chunked + gzip + shuffle,
                                          chunked without filters with ragged edge chunks, a compact dataset, a nested group
  tests/golden/hdf5_latest.hdf5           a file written with libver="latest" (version-2 object headers, compact links)
  tests/golden/hdf5_expected.npz          every dataset of the three files as h5py read it back, keyed "<file>:<path>"
"""
import os

import h5py
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    rng = np.random.default_rng(20240917)
    expected = {}

    # ---- the reference's layout
    path = os.path.join(ROOT, "two_view_train_data.hdf5")
    with h5py.File(path, "w") as f:
        for i in range(6):
            grp = f.create_group("{:06}".format(i))
            shape = (12, 10, 8) if i != 3 else (9, 11, 7)
            view1 = rng.integers(0, 256, shape).astype(np.uint8)
            view2 = rng.integers(0, 256, shape).astype(np.uint8)
            grp["img"] = np.stack((view1, view2), axis=0)
            grp["seg"] = rng.integers(0, 5, shape).astype(np.uint8)

    # ---- variations
    path2 = os.path.join(ROOT, "hdf5_variants.hdf5")
    with h5py.File(path2, "w") as f:
        many = f.create_group("many")
        for i in range(300):
            many["s{:04}".format(i)] = np.arange(i, i + 3, dtype=np.int32)
        f["f32"] = rng.standard_normal((5, 6, 7)).astype(np.float32)
        f["f64"] = rng.standard_normal((4, 3))
        f["f16"] = rng.standard_normal((8,)).astype(np.float16)
        f["i16"] = rng.integers(-3000, 3000, (3, 9)).astype(np.int16)
        f["be_u16"] = rng.integers(0, 60000, (4, 5)).astype(">u2")
        f["be_f32"] = rng.standard_normal((6,)).astype(">f4")
        f["scalar"] = np.float32(2.5)
        f.create_dataset("gz", data=rng.integers(0, 4, (2, 20, 18, 16)).astype(np.uint8), chunks=(1, 8, 8, 8), compression="gzip",
                         compression_opts=4, shuffle=True)
        f.create_dataset("gz_f32", data=rng.standard_normal((10, 33)).astype(np.float32), chunks=(4, 16), compression="gzip", shuffle=True,
                         fletcher32=True)
        f.create_dataset("chunked_raw", data=rng.integers(0, 1000, (7, 13)).astype(np.int64), chunks=(3, 5))
        f.create_dataset("unwritten", shape=(3, 4), dtype=np.float32)
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        arr = rng.integers(0, 255, (4, 6)).astype(np.uint8)
        space = h5py.h5s.create_simple(arr.shape)
        dsid = h5py.h5d.create(f.id, b"compact", h5py.h5t.NATIVE_UINT8, space, dcpl)
        dsid.write(h5py.h5s.ALL, h5py.h5s.ALL, arr)
        sub = f.create_group("outer").create_group("inner")
        sub["deep"] = np.arange(24, dtype=np.uint8).reshape(2, 3, 4)

    path3 = os.path.join(ROOT, "hdf5_latest.hdf5")
    with h5py.File(path3, "w", libver="latest") as f:
        for i in range(4):
            g = f.create_group("{:06}".format(i))
            g["img"] = rng.integers(0, 256, (2, 6, 5, 4)).astype(np.uint8)
            g["seg"] = rng.integers(0, 3, (6, 5, 4)).astype(np.uint8)

    def record(tag, name, obj):
        if isinstance(obj, h5py.Dataset):
            expected[f"{tag}:{name}"] = np.asarray(obj[()])

    for tag, p in (("two_view", path), ("variants", path2), ("latest", path3)):
        with h5py.File(p, "r") as f:
            f.visititems(lambda n, o, tag=tag: record(tag, n, o))
            expected[f"{tag}:__keys__"] = np.array(list(f.keys()))
    np.savez_compressed(os.path.join(ROOT, "hdf5_expected.npz"), **expected)
    for p in (path, path2, path3, os.path.join(ROOT, "hdf5_expected.npz")):
        print(os.path.basename(p), os.path.getsize(p), "bytes")
    print("h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version)


if __name__ == "__main__":
    main()
