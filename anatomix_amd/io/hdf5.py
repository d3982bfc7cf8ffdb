"""Read-only HDF5 access for the pretraining containers, numpy only (h5py is not a dependency of this package).

What the reference stores (synthetic-data-generation/step3_generate_h5_w_segs.py:28-51, read back by
pretraining/data/h5supcl_dataset.py:100-102,208-258): one file per split, one group per subject named ``"%06d"``, in it
``img`` = uint8 ``[2, X, Y, Z]`` (the two views) and ``seg`` = uint8 ``[X, Y, Z]``, written by h5py with its defaults --
i.e. the "earliest" file format of the HDF5 library: version-0 superblock, version-1 object headers, groups as symbol
tables (B-tree v1 + local heap), contiguous little-endian datasets.  This module implements exactly the structures such files
can contain, plus the common variations a user-made container brings along (chunked / gzip / shuffle datasets, compact
datasets, version-2 object headers with compact link storage, big-endian and floating-point element types), following the
public "HDF5 File Format Specification Version 3.0".  Anything else raises ``NotImplementedError`` naming the feature.

Pinned: ``oracle/make_golden_hdf5.py`` writes the fixtures under ``tests/golden/`` with the real library (h5py 3.3.0 / HDF5
1.10.6 found in the build container) and ``tests/test_hdf5.py`` compares every dataset read through this module with what
h5py read back.

    with H5File(path) as f:
        ids = list(f.keys())                 # like h5py: link names in name order
        views = f[ids[0]]["img"]             # H5Dataset: .shape, .dtype, views[i], views[...], np.array(views)
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = {4: 0xFFFFFFFF, 8: 0xFFFFFFFFFFFFFFFF}

# header message types (spec IV.A.2)
_MSG_DATASPACE, _MSG_LINKINFO, _MSG_DATATYPE, _MSG_LINK, _MSG_LAYOUT, _MSG_FILTERS = 0x1, 0x2, 0x3, 0x6, 0x8, 0xB
_MSG_CONTINUATION, _MSG_SYMTAB = 0x10, 0x11


class H5FormatError(ValueError):
    pass


class _Reader:
    def __init__(self, path):
        self.fh = open(path, "rb")
        self.path = path
        self.base = 0
        self.so = self.sl = 8

    def at(self, addr: int, n: int) -> bytes:
        self.fh.seek(self.base + addr)
        b = self.fh.read(n)
        if len(b) != n:
            raise H5FormatError(f"{self.path}: truncated file (wanted {n} bytes at {addr})")
        return b

    def uint(self, b: bytes, off: int, n: int) -> int:
        return int.from_bytes(b[off:off + n], "little")

    def close(self):
        self.fh.close()


def _parse_superblock(r: _Reader) -> int:
    """Returns the address of the root group's object header (spec II.A)."""
    off = 0
    while True:                                    # the superblock may sit at 0, 512, 1024, 2048, ...
        r.fh.seek(off)
        if r.fh.read(8) == _SIG:
            break
        off = 512 if off == 0 else off * 2
        r.fh.seek(0, 2)
        if off >= r.fh.tell():
            raise H5FormatError(f"{r.path}: not an HDF5 file (no superblock signature)")
    r.base = 0
    head = r.at(off, 64 + 48)
    version = head[8]
    if version in (0, 1):
        r.so, r.sl = head[13], head[14]
        p = 24 + (4 if version == 1 else 0)        # v1 adds indexed-storage K + reserved
        base = r.uint(head, p, r.so)
        p += 4 * r.so                               # base, free-space, end-of-file, driver-info addresses
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        root = r.uint(head, p + r.so, r.so)
        r.base = base
        return root
    if version in (2, 3):
        r.so, r.sl = head[9], head[10]
        base = r.uint(head, 12, r.so)
        root = r.uint(head, 12 + 3 * r.so, r.so)
        r.base = base
        return root
    raise NotImplementedError(f"{r.path}: superblock version {version}")


def _messages(r: _Reader, addr: int) -> List[Tuple[int, bytes]]:
    """All header messages of the object at `addr` as (type, body), following continuation blocks (spec IV.A.1)."""
    first = r.at(addr, 16)
    out: List[Tuple[int, bytes]] = []
    if first[:4] == b"OHDR":                       # version 2 object header
        flags = first[5]
        p = 6
        if flags & 0x20:
            p += 16                                # access, modification, change, birth times
        if flags & 0x10:
            p += 4                                 # max compact / min dense attribute counts
        szlen = 1 << (flags & 3)
        hdr = r.at(addr, p + szlen)
        chunk0 = r.uint(hdr, p, szlen)
        blocks = [(addr + p + szlen, chunk0)]
        order = bool(flags & 0x04)
        while blocks:
            a, n = blocks.pop(0)
            b = r.at(a, n)
            q = 0
            while q + 4 <= n:
                mtype, msize = b[q], r.uint(b, q + 1, 2)
                q += 4 + (2 if order else 0)
                body = b[q:q + msize]
                q += msize
                if mtype == _MSG_CONTINUATION:
                    ca, cl = r.uint(body, 0, r.so), r.uint(body, r.so, r.sl)
                    if r.at(ca, 4) != b"OCHK":
                        raise H5FormatError(f"{r.path}: bad continuation block at {ca}")
                    blocks.append((ca + 4, cl - 8))            # between the signature and the checksum
                elif mtype != 0:
                    out.append((mtype, body))
        return out
    if first[0] != 1:
        raise H5FormatError(f"{r.path}: object header version {first[0]} at {addr}")
    nmsg = r.uint(first, 2, 2)
    size = r.uint(first, 8, 4)
    blocks = [(addr + 16, size)]
    while blocks and len(out) < nmsg + 64:
        a, n = blocks.pop(0)
        b = r.at(a, n)
        q = 0
        while q + 8 <= n:
            mtype, msize = r.uint(b, q, 2), r.uint(b, q + 2, 2)
            body = b[q + 8:q + 8 + msize]
            q += 8 + msize
            if mtype == _MSG_CONTINUATION:
                blocks.append((r.uint(body, 0, r.so), r.uint(body, r.so, r.sl)))
            elif mtype != 0:
                out.append((mtype, body))
    return out


def _heap_name(r: _Reader, heap_data: int, off: int) -> str:
    r.fh.seek(r.base + heap_data + off)
    name = b""
    while True:
        piece = r.fh.read(64)
        end = piece.find(b"\x00")
        if end >= 0 or not piece:
            return (name + (piece[:end] if end >= 0 else piece)).decode("utf-8")
        name += piece


def _symtab_links(r: _Reader, btree: int, heap: int) -> Dict[str, int]:
    """Old-style group: B-tree v1 of symbol-table nodes, names in the local heap (spec III.A.1, III.B, III.D)."""
    h = r.at(heap, 8 + 2 * r.sl + r.so)
    if h[:4] != b"HEAP":
        raise H5FormatError(f"{r.path}: bad local heap at {heap}")
    heap_data = r.uint(h, 8 + 2 * r.sl, r.so)
    links: Dict[str, int] = {}

    def walk(node: int):
        head = r.at(node, 8 + 2 * r.so)
        if head[:4] == b"SNOD":
            nsym = r.uint(head, 6, 2)
            esz = 2 * r.so + 24
            b = r.at(node + 8, nsym * esz)
            for i in range(nsym):
                name_off = r.uint(b, i * esz, r.so)
                links[_heap_name(r, heap_data, name_off)] = r.uint(b, i * esz + r.so, r.so)
            return
        if head[:4] != b"TREE" or head[4] != 0:
            raise H5FormatError(f"{r.path}: bad group B-tree node at {node}")
        used = r.uint(head, 6, 2)
        body = r.at(node + 8 + 2 * r.so, (used + 1) * r.sl + used * r.so)
        p = r.sl                                    # key 0
        for _ in range(used):
            walk(r.uint(body, p, r.so))
            p += r.so + r.sl

    walk(btree)
    return links


def _link_message(r: _Reader, body: bytes) -> Optional[Tuple[str, int]]:
    """Compact new-style link (spec IV.A.2.g); hard links only."""
    flags = body[1]
    p = 2
    ltype = 0
    if flags & 0x08:
        ltype = body[p]
        p += 1
    if flags & 0x04:
        p += 8                                     # creation order
    if flags & 0x10:
        p += 1                                     # character set
    nlen = 1 << (flags & 3)
    n = r.uint(body, p, nlen)
    p += nlen
    name = body[p:p + n].decode("utf-8")
    p += n
    if ltype != 0:
        return None                                # soft / external links are not followed
    return name, r.uint(body, p, r.so)


class H5Group:
    def __init__(self, r: _Reader, addr: int, name: str):
        self._r, self._addr, self.name = r, addr, name
        self._links: Optional[Dict[str, int]] = None

    def _load(self) -> Dict[str, int]:
        if self._links is None:
            links: Dict[str, int] = {}
            for mtype, body in _messages(self._r, self._addr):
                if mtype == _MSG_SYMTAB:
                    links.update(_symtab_links(self._r, self._r.uint(body, 0, self._r.so), self._r.uint(body, self._r.so, self._r.so)))
                elif mtype == _MSG_LINK:
                    kv = _link_message(self._r, body)
                    if kv:
                        links[kv[0]] = kv[1]
                elif mtype == _MSG_LINKINFO:
                    so = self._r.so
                    p = 2 + (8 if body[1] & 1 else 0)
                    if self._r.uint(body, p, so) != _UNDEF[so]:
                        raise NotImplementedError(f"{self._r.path}: group {self.name!r} stores its links densely (fractal heap); "
                                                  "rewrite the file with the default (earliest) format")
            self._links = dict(sorted(links.items()))          # h5py iterates in name order
        return self._links

    def keys(self):
        return self._load().keys()

    def __iter__(self) -> Iterator[str]:
        return iter(self._load())

    def __len__(self):
        return len(self._load())

    def __contains__(self, name):
        return name in self._load()

    def __getitem__(self, name: str):
        node = self
        for part in [p for p in name.split("/") if p]:
            if not isinstance(node, H5Group):
                raise KeyError(name)
            links = node._load()
            if part not in links:
                raise KeyError(f"Unable to open object (object {part!r} doesn't exist)")
            node = _open_object(node._r, links[part], (node.name.rstrip("/") + "/" + part))
        return node

    def __repr__(self):
        return f"<H5Group {self.name!r} ({len(self)} members)>"


def _dtype_of(body: bytes, path: str) -> np.dtype:
    cls, bits0 = body[0] & 0x0F, body[1]
    size = int.from_bytes(body[4:8], "little")
    order = ">" if bits0 & 1 else "<"
    if cls == 0:                                    # fixed-point
        return np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}")
    if cls == 1:                                    # floating-point (IEEE layouts only)
        if size not in (2, 4, 8):
            raise NotImplementedError(f"{path}: {size}-byte floating-point elements")
        return np.dtype(f"{order}f{size}")
    raise NotImplementedError(f"{path}: datatype class {cls} (only integer and floating-point datasets are supported)")


class H5Dataset:
    def __init__(self, r: _Reader, addr: int, name: str, msgs):
        self._r, self.name = r, name
        self.shape: Tuple[int, ...] = ()
        self.dtype = None
        self._layout = None
        self._filters: List[Tuple[int, Tuple[int, ...]]] = []
        for mtype, body in msgs:
            if mtype == _MSG_DATASPACE:
                ver, rank = body[0], body[1]
                p = 8 if ver == 1 else 4
                if ver == 2 and body[3] == 2:
                    raise NotImplementedError(f"{name}: null dataspace")
                self.shape = tuple(r.uint(body, p + i * r.sl, r.sl) for i in range(rank))
            elif mtype == _MSG_DATATYPE:
                self.dtype = _dtype_of(body, name)
            elif mtype == _MSG_LAYOUT:
                self._layout = body
            elif mtype == _MSG_FILTERS:
                self._filters = self._parse_filters(body)
        if self.dtype is None or self._layout is None:
            raise H5FormatError(f"{r.path}: {name} is not a dataset")
        # version 4 (libver='latest') differs from version 3 only in how chunks are indexed
        if self._layout[0] not in (3, 4) or (self._layout[0] == 4 and self._layout[1] == 2):
            raise NotImplementedError(f"{name}: data layout message version {self._layout[0]}, class {self._layout[1]} (the chunk "
                                      "indexes of libver='latest'); rewrite the file with the default format")

    @staticmethod
    def _parse_filters(body: bytes):
        ver, nf = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(nf):
            fid = int.from_bytes(body[p:p + 2], "little")
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(body[p:p + 2], "little")
                p += 2
            p += 2                                  # flags
            ncd = int.from_bytes(body[p:p + 2], "little")
            p += 2
            if ver == 1:
                nlen = (nlen + 7) // 8 * 8
            p += nlen
            cd = tuple(int.from_bytes(body[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncd))
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    # ---- whole-array read --------------------------------------------------------------------------------------------
    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of a scalar dataset")
        return self.shape[0]

    def _contiguous_address(self) -> Optional[int]:
        if self._layout[1] == 1:
            a = self._r.uint(self._layout, 2, self._r.so)
            return None if a == _UNDEF[self._r.so] else a
        return None

    def _read_rows(self, start: int, stop: int) -> np.ndarray:
        """Elements [start, stop) along the first axis (the whole array for a scalar)."""
        r, cls = self._r, self._layout[1]
        rest = self.shape[1:]
        row = int(np.prod(rest, dtype=np.int64)) * self.dtype.itemsize if self.shape else self.dtype.itemsize
        shape = ((stop - start,) + rest) if self.shape else ()
        nbytes = row * (stop - start if self.shape else 1)
        if cls == 0:                                # compact: the data sit in the message
            n = r.uint(self._layout, 2, 2)
            raw = self._layout[4:4 + n][start * row:start * row + nbytes]
            return np.frombuffer(raw, self.dtype).reshape(shape).copy()
        if cls == 1:
            a = self._contiguous_address()
            if a is None:                           # never written: fill value (zero)
                return np.zeros(shape, self.dtype)
            return np.frombuffer(r.at(a + start * row, nbytes), self.dtype).reshape(shape).copy()
        if cls == 2:
            return self._read_chunked(start, stop)
        raise NotImplementedError(f"{self.name}: layout class {cls}")

    def _read_chunked(self, start: int, stop: int) -> np.ndarray:
        r = self._r
        ndim1 = self._layout[2]                     # rank + 1
        btree = r.uint(self._layout, 3, r.so)
        cdims = tuple(r.uint(self._layout, 3 + r.so + 4 * i, 4) for i in range(ndim1 - 1))
        out = np.zeros((stop - start,) + self.shape[1:], self.dtype)
        if btree == _UNDEF[r.so]:
            return out
        csize = int(np.prod(cdims, dtype=np.int64)) * self.dtype.itemsize

        def decode(raw: bytes, mask: int) -> np.ndarray:
            for k in range(len(self._filters) - 1, -1, -1):       # the pipeline is undone last filter first
                fid, _cd = self._filters[k]
                if mask & (1 << k):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                      # shuffle: byte planes -> elements
                    es = self.dtype.itemsize
                    raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
                elif fid == 3:                      # fletcher32 checksum appended
                    raw = raw[:-4]
                else:
                    raise NotImplementedError(f"{self.name}: HDF5 filter id {fid} (only gzip, shuffle, fletcher32)")
            return np.frombuffer(raw[:csize], self.dtype).reshape(cdims)

        def walk(node: int):
            head = r.at(node, 8 + 2 * r.so)
            if head[:4] != b"TREE" or head[4] != 1:
                raise H5FormatError(f"{r.path}: bad chunk B-tree node at {node}")
            level, used = head[5], r.uint(head, 6, 2)
            ksz = 8 + 8 * ndim1
            body = r.at(node + 8 + 2 * r.so, used * (ksz + r.so) + ksz)
            for e in range(used):
                k = e * (ksz + r.so)
                nbytes, mask = r.uint(body, k, 4), r.uint(body, k + 4, 4)
                offs = tuple(r.uint(body, k + 8 + 8 * i, 8) for i in range(ndim1 - 1))
                child = r.uint(body, k + ksz, r.so)
                if level > 0:
                    walk(child)
                    continue
                if offs[0] >= stop or offs[0] + cdims[0] <= start:
                    continue
                chunk = decode(r.at(child, nbytes), mask)
                src, dst = [], []
                for ax, (o, c, s) in enumerate(zip(offs, cdims, self.shape)):
                    lo, hi = o, min(o + c, s)
                    if ax == 0:
                        lo, hi = max(lo, start), min(hi, stop)
                        dst.append(slice(lo - start, hi - start))
                    else:
                        dst.append(slice(lo, hi))
                    src.append(slice(lo - o, hi - o))
                out[tuple(dst)] = chunk[tuple(src)]

        walk(btree)
        return out

    # ---- numpy-style access (what the data loader uses: ds[i], ds[()] / np.array(ds), ds.shape) ----------------------
    def __getitem__(self, key):
        if key is Ellipsis or key == ():
            return self._read_rows(0, self.shape[0]) if self.shape else self._read_rows(0, 1)
        if not self.shape:
            raise IndexError("scalar dataset: use ds[()]")
        first, rest = (key[0], key[1:]) if isinstance(key, tuple) else (key, ())
        n = self.shape[0]
        if isinstance(first, (int, np.integer)):
            i = int(first)
            if i < 0:
                i += n
            if not 0 <= i < n:
                raise IndexError(f"index {int(first)} out of range for axis 0 with size {n}")
            block = self._read_rows(i, i + 1)[0]
        elif isinstance(first, slice):
            lo, hi, step = first.indices(n)
            if step != 1:
                block = self._read_rows(0, n)[first]
            else:
                block = self._read_rows(lo, max(hi, lo))
        else:
            block = self._read_rows(0, n)[first]
        if not rest:
            return block
        return block[rest] if isinstance(first, (int, np.integer)) else block[(slice(None),) + rest]

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return f"<H5Dataset {self.name!r}: shape {self.shape}, type {self.dtype.str!r}>"


def _open_object(r: _Reader, addr: int, name: str):
    msgs = _messages(r, addr)
    kinds = {t for t, _ in msgs}
    if _MSG_LAYOUT in kinds and _MSG_DATATYPE in kinds:
        return H5Dataset(r, addr, name, msgs)
    return H5Group(r, addr, name)


class H5File(H5Group):
    """Read-only file object with the part of ``h5py.File``'s surface the reference's loader touches
    (h5supcl_dataset.py:100-102, 208-258): context manager, ``keys()``, ``f[subject]["img"]``."""

    def __init__(self, path, mode: str = "r", **_ignored):
        if mode != "r":
            raise ValueError("anatomix_amd.io.hdf5 is read-only")
        r = _Reader(path)
        try:
            root = _parse_superblock(r)
        except Exception:
            r.close()
            raise
        super().__init__(r, root, "/")
        self.filename = str(path)

    def close(self):
        self._r.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
