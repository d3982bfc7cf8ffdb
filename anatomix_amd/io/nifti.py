"""Minimal NIfTI-1 single-file reader / writer (.nii, .nii.gz) in numpy -- the volumes the reference's registration and
segmentation scripts exchange through nibabel (``nib.load(p).get_fdata()`` / ``nib.Nifti1Image(arr, affine)``,
run_convex_adam_with_network_feats.py:125-147, 270-325).  nibabel is not a dependency of this package; the layout follows
the public NIfTI-1 header definition (348-byte header, data at vox_offset, first index fastest)."""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _quaternion_affine(hdr, pixdim):
    b, c, d = hdr["quatern_b"], hdr["quatern_c"], hdr["quatern_d"]
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    rot = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                    [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                    [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    qfac = -1.0 if pixdim[0] < 0 else 1.0
    aff = np.eye(4)
    aff[:3, :3] = rot * np.array([pixdim[1], pixdim[2], pixdim[3] * qfac])
    aff[:3, 3] = [hdr["qoffset_x"], hdr["qoffset_y"], hdr["qoffset_z"]]
    return aff


def load_nifti(path):
    """Returns (data float64 [i, j, k, ...] with scl_slope / scl_inter applied -- what ``get_fdata()`` gives --, affine 4x4
    (sform if set, else qform, else the voxel sizes), header dict)."""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError(f"{path}: shorter than a NIfTI-1 header")
    end = "<" if struct.unpack("<i", raw[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", raw[:4])[0] != 348:
        raise ValueError(f"{path}: sizeof_hdr is not 348 (not a NIfTI-1 file)")
    if raw[344:348] not in (b"n+1\0", b"ni1\0"):
        raise ValueError(f"{path}: bad magic {raw[344:348]!r}")
    if raw[344:348] == b"ni1\0":
        raise ValueError(f"{path}: header/image pairs (.hdr/.img) are not supported, only single-file NIfTI-1")
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(end + "2h", raw[70:74])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "2h", raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + "6f", raw[256:280])
    srow = np.array(struct.unpack(end + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype code {datatype}")
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError(f"{path}: bad dim[0] = {ndim}")
    shape = tuple(int(v) for v in dim[1:1 + ndim])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    count = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    if len(raw) < off + count * dt.itemsize:
        raise ValueError(f"{path}: file ends before the {count} voxels the header announces")
    data = np.frombuffer(raw, dtype=dt, count=count, offset=off).reshape(shape, order="F").astype(np.float64)
    if slope != 0 and not np.isnan(slope) and not (slope == 1 and inter == 0):
        data = data * float(slope) + float(inter)
    hdr = dict(dim=dim, datatype=datatype, bitpix=bitpix, pixdim=pixdim, vox_offset=vox_offset, scl_slope=slope, scl_inter=inter,
               qform_code=qform_code, sform_code=sform_code, quatern_b=qb, quatern_c=qc, quatern_d=qd, qoffset_x=qx,
               qoffset_y=qy, qoffset_z=qz, endianness=end)
    if sform_code > 0:
        affine = np.vstack([srow, [0, 0, 0, 1]])
    elif qform_code > 0:
        affine = _quaternion_affine(hdr, pixdim)
    else:
        affine = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0]).astype(np.float64)
    return data, affine, hdr


def save_nifti(path, array, affine=None, dtype=None):
    """Writes ``array`` ([i, j, k, ...], up to 7 dims) as single-file NIfTI-1 with ``affine`` as the sform (and a matching
    qform offset); dtype defaults to the array's (float64 is kept, bool becomes uint8)."""
    arr = np.asarray(array)
    if dtype is not None:
        arr = arr.astype(dtype)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype.name not in _CODES:
        raise ValueError(f"cannot store dtype {arr.dtype} in NIfTI-1")
    if not 1 <= arr.ndim <= 7:
        raise ValueError("NIfTI-1 holds 1 to 7 dimensions")
    affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    vox = np.sqrt((affine[:3, :3] ** 2).sum(0))
    dim = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    pixdim = [1.0] + [float(v) for v in vox] + [1.0] * 4
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<2h", hdr, 70, _CODES[arr.dtype.name], arr.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, *pixdim)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2 | (8 << 3)                                   # xyzt_units: millimetres, seconds
    struct.pack_into("<2h", hdr, 252, 0, 2)                  # qform unknown, sform = aligned
    struct.pack_into("<3f", hdr, 268, *[float(v) for v in affine[:3, 3]])
    struct.pack_into("<12f", hdr, 280, *[float(v) for v in affine[:3, :].reshape(-1)])
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")
        f.write(np.asfortranarray(arr).astype(arr.dtype.newbyteorder("<")).tobytes(order="F"))
