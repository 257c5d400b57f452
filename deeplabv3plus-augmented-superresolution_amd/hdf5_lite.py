"""Minimal HDF5 reader / writer in pure Python + numpy for the files the reference exchanges through h5py
(h5py is not installed next to the product's interpreter):

* the augmented-copies interchange file of ``compute_augmented_feature_maps`` (superresolution_scripts/
  augmentation_utils.py:117-136: datasets ``class_masks`` / ``max_masks`` / ``angles`` / ``shifts``, attributes
  ``filename`` / ``mode`` (str) and ``angle_max`` / ``shift_max`` (numbers)) read back by ``load_SR_data``
  (superres_utils.py:154-210);
* Keras ``.h5`` weight files (model.py:129-145: nested groups ``<layer>/<layer>/<variable>:0``).

Scope = what h5py writes by default (libver "earliest"): superblock version 0, version-1 object headers (with
continuation blocks), old-style groups (symbol-table message -> v1 B-tree -> symbol-table nodes -> local heap),
contiguous or compact dataset layout, little/big-endian fixed-point and floating-point types, fixed-length and
variable-length (global heap) strings, attribute messages versions 1-3.  Chunked / compressed datasets, new-style
(fractal-heap) groups and superblock >= 2 raise ``Hdf5Error`` -- nothing is guessed.

The writer produces the same classic layout, so files written here open in h5py / libhdf5 (verified in this image with
the h5py 3.3.0 of /opt/conda, tests/test_hdf5_lite.py) and string attributes come back as ``str`` exactly like h5py's.

Host-side format code only: no device work, nothing here is on the timed path.
"""
from __future__ import annotations

import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(Exception):
    pass


# ================================================================================================ reader
class _Reader:
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != SIGNATURE:
            raise Hdf5Error("not an HDF5 file (bad signature)")
        ver = buf[8]
        if ver not in (0, 1):
            raise Hdf5Error(f"superblock version {ver} is not supported (only the classic version 0/1 layout)")
        if buf[13] != 8 or buf[14] != 8:
            raise Hdf5Error("only 8-byte offsets and lengths are supported")
        off = 24 + (4 if ver == 1 else 0)
        self.base = self.u64(off)
        root_entry = off + 32                       # base, free-space, end-of-file, driver-info addresses
        self.root_header = self.u64(root_entry + 8)
        self._gcol = {}
        self.shapes_only = False

    def u16(self, o): return struct.unpack_from("<H", self.b, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.b, o)[0]
    def u64(self, o): return struct.unpack_from("<Q", self.b, o)[0]

    # ---- object header (version 1) -> list of (type, flags, offset, size) ----------------------------------
    def messages(self, addr):
        addr += self.base
        if self.b[addr] != 1:
            raise Hdf5Error(f"object header version {self.b[addr]} at {addr:#x} is not supported (version 1 only)")
        nmsg = self.u16(addr + 2)
        size = self.u32(addr + 8)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            o, n = blocks.pop(0)
            end = o + n
            while o + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self.u16(o), self.u16(o + 2), self.b[o + 4]
                out.append((mtype, flags, o + 8, msize))
                if mtype == 0x10:                   # continuation
                    blocks.append((self.base + self.u64(o + 8), self.u64(o + 16)))
                o += 8 + msize
        return out

    # ---- datatype -> (numpy dtype | ("vlen_str",) | ("str", n), size_in_file, bytes consumed) --------------
    def datatype(self, o):
        cls, ver = self.b[o] & 15, self.b[o] >> 4
        bits = self.b[o + 1] | (self.b[o + 2] << 8) | (self.b[o + 3] << 16)
        size = self.u32(o + 4)
        if cls == 0:                                # fixed point
            dt = np.dtype(("<" if not bits & 1 else ">") + ("i" if bits & 8 else "u") + str(size))
            return dt, size, 8 + 4
        if cls == 1:                                # floating point
            if size not in (2, 4, 8):
                raise Hdf5Error(f"float size {size} not supported")
            return np.dtype(("<" if not bits & 1 else ">") + "f" + str(size)), size, 8 + 12
        if cls == 3:                                # fixed-length string
            return ("str", size, (bits >> 4) & 15), size, 8
        if cls == 9:                                # variable length
            if bits & 15 != 1:
                raise Hdf5Error("variable-length sequences are not supported (strings only)")
            _, _, used = self.datatype(o + 8)
            return ("vlen_str",), size, 8 + used
        raise Hdf5Error(f"datatype class {cls} (version {ver}) is not supported")

    def dataspace(self, o):
        ver = self.b[o]
        rank, flags = self.b[o + 1], self.b[o + 2]
        if ver == 1:
            p = o + 8
        elif ver == 2:
            if self.b[o + 3] == 2:
                raise Hdf5Error("null dataspace")
            p = o + 4
        else:
            raise Hdf5Error(f"dataspace version {ver} not supported")
        dims = tuple(self.u64(p + 8 * i) for i in range(rank))
        used = (p - o) + 8 * rank * (2 if flags & 1 else 1)
        return dims, used

    def global_heap_object(self, addr, index):
        addr += self.base
        if addr not in self._gcol:
            if self.b[addr:addr + 4] != b"GCOL":
                raise Hdf5Error(f"bad global heap signature at {addr:#x}")
            size = self.u64(addr + 8)
            objs, o = {}, addr + 16
            while o + 16 <= addr + size:
                idx, n = self.u16(o), self.u64(o + 8)
                if idx == 0:
                    break
                objs[idx] = bytes(self.b[o + 16:o + 16 + n])
                o += 16 + ((n + 7) & ~7)
            self._gcol[addr] = objs
        return self._gcol[addr][index]

    def decode(self, dt, dims, raw_off, raw=None):
        """Value of a dataset / attribute with datatype dt and shape dims whose bytes start at raw_off."""
        count = int(np.prod(dims, dtype=np.int64)) if dims else 1
        if isinstance(dt, np.dtype):
            a = np.frombuffer(self.b, dtype=dt, count=count, offset=raw_off).reshape(dims)
            a = a.astype(dt.newbyteorder("="), copy=True)
            return a if dims else a[()]
        if dt[0] == "vlen_str":
            vals = []
            for i in range(count):
                o = raw_off + 16 * i
                n, addr, idx = self.u32(o), self.u64(o + 4), self.u32(o + 12)
                vals.append(self.global_heap_object(addr, idx)[:n].decode("utf-8") if n else "")
        else:
            n = dt[1]
            vals = [bytes(self.b[raw_off + n * i:raw_off + n * (i + 1)]).split(b"\0")[0] for i in range(count)]
            if dt[2] == 1:                          # UTF-8 fixed-length strings come back as str, ASCII as bytes (h5py)
                vals = [v.decode("utf-8") for v in vals]
        if not dims:
            return vals[0]
        return np.array(vals, dtype=object).reshape(dims)

    def attribute(self, o):
        ver = self.b[o]
        nsz, tsz, ssz = self.u16(o + 2), self.u16(o + 4), self.u16(o + 6)
        p = o + 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        if ver not in (1, 2, 3):
            raise Hdf5Error(f"attribute message version {ver} not supported")
        name = bytes(self.b[p:p + nsz]).split(b"\0")[0].decode("utf-8")
        p += pad(nsz)
        dt, _, _ = self.datatype(p)
        p += pad(tsz)
        dims, _ = self.dataspace(p)
        p += pad(ssz)
        return name, self.decode(dt, dims, p)

    # ---- groups ------------------------------------------------------------------------------------------
    def group_entries(self, btree, heap):
        heap += self.base
        if self.b[heap:heap + 4] != b"HEAP":
            raise Hdf5Error("bad local heap signature")
        heap_data = self.base + self.u64(heap + 24)

        def name_at(off):
            e = self.b.find(b"\0", heap_data + off)
            return bytes(self.b[heap_data + off:e]).decode("utf-8")

        out = []

        def walk(node):
            node += self.base
            sig = bytes(self.b[node:node + 4])
            if sig == b"TREE":
                if self.b[node + 4] != 0:
                    raise Hdf5Error("not a group B-tree")
                n = self.u16(node + 6)
                for i in range(n):
                    walk(self.u64(node + 24 + 8 + 16 * i))
            elif sig == b"SNOD":
                n = self.u16(node + 6)
                for i in range(n):
                    e = node + 8 + 40 * i
                    out.append((name_at(self.u64(e)), self.u64(e + 8)))
            else:
                raise Hdf5Error(f"unexpected node signature {sig!r} in a group B-tree")

        walk(btree)
        return out

    def visit(self, header, path, datasets, attrs):
        msgs = self.messages(header)
        types = {m[0] for m in msgs}
        my_attrs = {}
        for mtype, _, o, _ in msgs:
            if mtype == 0x0C:
                k, v = self.attribute(o)
                my_attrs[k] = v
        attrs[path or "/"] = my_attrs
        if 0x11 in types:                           # group
            o = next(m[2] for m in msgs if m[0] == 0x11)
            for name, child in self.group_entries(self.u64(o), self.u64(o + 8)):
                self.visit(child, f"{path}/{name}" if path else name, datasets, attrs)
            return
        if 0x02 in types or 0x06 in types and 0x08 not in types:
            raise Hdf5Error(f"{path or '/'}: new-style (link-message / fractal-heap) groups are not supported")
        if 0x08 not in types:
            return                                  # a committed datatype or an empty new-style group: nothing to read
        dt = dims = None
        for mtype, _, o, _ in msgs:
            if mtype == 0x03:
                dt, _, _ = self.datatype(o)
            elif mtype == 0x01:
                dims, _ = self.dataspace(o)
            elif mtype == 0x0B:
                raise Hdf5Error(f"{path}: filtered (compressed) datasets are not supported")
        o = next(m[2] for m in msgs if m[0] == 0x08)
        if self.b[o] != 3:
            raise Hdf5Error(f"{path}: data layout message version {self.b[o]} not supported")
        cls = self.b[o + 1]
        if cls == 0:
            raw = o + 4
        elif cls == 1:
            addr = self.u64(o + 2)
            if addr == UNDEF:                       # never written: the fill value (0)
                datasets[path] = tuple(dims) if self.shapes_only else np.zeros(dims, dt if isinstance(dt, np.dtype) else object)
                return
            raw = self.base + addr
        else:
            raise Hdf5Error(f"{path}: chunked datasets are not supported")
        datasets[path] = tuple(dims) if self.shapes_only else self.decode(dt, dims, raw)


def shapes(path):
    """-> {"group/name": shape tuple} from the object headers alone (the file is memory-mapped; no dataset is read)."""
    import mmap
    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as buf:
        r = _Reader(buf)
        r.shapes_only = True
        datasets, attrs = {}, {}
        r.visit(r.root_header, "", datasets, attrs)
    return datasets


def header(path):
    """-> (shapes {"group/name": shape tuple}, attrs) from the object headers alone: like read() without touching a dataset's
    bytes (the file is memory-mapped), for callers that only need to know what a file holds."""
    import mmap
    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as buf:
        r = _Reader(buf)
        r.shapes_only = True
        datasets, attrs = {}, {}
        r.visit(r.root_header, "", datasets, attrs)
    return datasets, attrs


def read(path):
    """-> (datasets {"group/sub/name": ndarray}, attrs {"/" | "group" | "group/name": {attr: value}})."""
    with open(path, "rb") as f:
        buf = f.read()
    r = _Reader(buf)
    datasets, attrs = {}, {}
    r.visit(r.root_header, "", datasets, attrs)
    return datasets, attrs


# ================================================================================================ writer
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        props = {2: (0, 16, 10, 5, 0, 10, 15), 4: (0, 32, 23, 8, 0, 23, 127), 8: (0, 64, 52, 11, 0, 52, 1023)}[dt.itemsize]
        bit_off, prec, epos, esize, mpos, msize, bias = props
        sign = prec - 1
        return struct.pack("<BBBBI", 0x11, 0x20, sign, 0, dt.itemsize) + struct.pack("<HHBBBBI", bit_off, prec, epos, esize, mpos, msize, bias)
    if dt.kind in "iu":
        return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    raise Hdf5Error(f"dtype {dt} cannot be written")


_VLEN_STR = struct.pack("<BBBBI", 0x19, 0x01, 0x01, 0, 16) + struct.pack("<BBBBI", 0x10, 0, 0, 0, 1) + struct.pack("<HH", 0, 8)


def _dataspace_msg(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


def _object_header(messages):
    body = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body


def write(path, datasets, attrs=None):
    """Flat file: datasets {name: array-like} in the root group, attrs {name: str | int | float} on the root group.
    Arrays are stored contiguous, little-endian, as float32/float64/int32/int64/...; str attributes as variable-length
    UTF-8 strings, numbers as float64 / int64 scalars -- what h5py produces for the reference's writer."""
    attrs = dict(attrs or {})
    names = sorted(datasets, key=lambda s: s.encode("utf-8"))
    if len(names) > 8:
        raise Hdf5Error("at most 8 datasets (one symbol-table node)")
    for n in list(names) + list(attrs):
        if not n or "/" in n:
            raise Hdf5Error(f"bad name {n!r}")
    arrays = {}
    for n in names:
        a = np.asarray(datasets[n])
        if a.dtype.kind not in "fiu":
            raise Hdf5Error(f"dataset {n}: dtype {a.dtype} cannot be written")
        arrays[n] = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False))

    # ---- fixed layout: superblock | root header | B-tree | local heap | symbol node | global heap | objects ----
    SUPER = 96
    heap_names = b"\0" * 8
    name_off = {}
    for n in names:
        name_off[n] = len(heap_names)
        heap_names += _pad8(n.encode("utf-8") + b"\0")
    heap_free = len(heap_names)
    heap_names += struct.pack("<QQ", 1, 32) + b"\0" * 16           # one free block (next = H5HL_FREE_NULL, size 32)

    gheap_objs, gheap_body = [], b""
    for k, v in attrs.items():
        if isinstance(v, str):
            raw = v.encode("utf-8")
            gheap_objs.append((k, len(raw)))
            gheap_body += struct.pack("<HH4xQ", len(gheap_objs), 0, len(raw)) + _pad8(raw)
    GHEAP_SIZE = max(4096, (16 + len(gheap_body) + 16 + 4095) // 4096 * 4096)

    def attr_msgs(gheap_addr):
        out, sidx = [], 0
        for k, v in attrs.items():
            nm = k.encode("utf-8") + b"\0"
            if isinstance(v, str):
                sidx += 1
                dt, data = _VLEN_STR, struct.pack("<IQI", len(v.encode("utf-8")), gheap_addr, sidx)
            elif isinstance(v, (bool, int, np.integer)):
                dt, data = _dtype_msg(np.int64), struct.pack("<q", int(v))
            elif isinstance(v, (float, np.floating)):
                dt, data = _dtype_msg(np.float64), struct.pack("<d", float(v))
            else:
                raise Hdf5Error(f"attribute {k}: type {type(v).__name__} cannot be written")
            sp = struct.pack("<BBB5x", 1, 0, 0)
            out.append(_msg(0x0C, struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + data))
        return out

    root_len = len(_object_header([_msg(0x11, b"\0" * 16)] + attr_msgs(0)))
    root_addr = SUPER
    btree_addr = root_addr + root_len
    BTREE_SIZE = 24 + (2 * 16 + 1) * 8 + 2 * 16 * 8
    heap_addr = btree_addr + BTREE_SIZE
    heap_data_addr = heap_addr + 32
    snod_addr = heap_data_addr + len(heap_names)
    SNOD_SIZE = 8 + 8 * 40
    gheap_addr = (snod_addr + SNOD_SIZE + 7) & ~7
    cursor = gheap_addr + (GHEAP_SIZE if gheap_objs else 0)

    headers, blobs = {}, []
    for n in names:
        a = arrays[n]
        hdr_len = len(_object_header([_msg(0x01, _dataspace_msg(a.shape)), _msg(0x03, _dtype_msg(a.dtype), 1),
                                      _msg(0x05, struct.pack("<BBBBI", 2, 2, 2, 1, 0), 1), _msg(0x08, b"\0" * 18)]))
        hdr_addr = cursor
        data_addr = (hdr_addr + hdr_len + 7) & ~7
        has_data = a.nbytes > 0
        layout = struct.pack("<BBQQ", 3, 1, data_addr if has_data else UNDEF, a.nbytes)
        hdr = _object_header([_msg(0x01, _dataspace_msg(a.shape)), _msg(0x03, _dtype_msg(a.dtype), 1),
                              _msg(0x05, struct.pack("<BBBBI", 2, 2, 2, 1, 0), 1), _msg(0x08, layout)])
        assert len(hdr) == hdr_len
        headers[n] = hdr_addr
        blobs.append((hdr_addr, hdr))
        if has_data:
            blobs.append((data_addr, a.tobytes()))
        cursor = (data_addr + a.nbytes + 7) & ~7
    eof = cursor

    out = bytearray(eof)
    out[0:8] = SIGNATURE
    struct.pack_into("<BBBBBBBBHHI", out, 8, 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    struct.pack_into("<QQQQ", out, 24, 0, UNDEF, eof, UNDEF)
    struct.pack_into("<QQII", out, 56, 0, root_addr, 1, 0)                 # root symbol-table entry, cached B-tree / heap
    struct.pack_into("<QQ", out, 80, btree_addr, heap_addr)
    root = _object_header([_msg(0x11, struct.pack("<QQ", btree_addr, heap_addr))] + attr_msgs(gheap_addr))
    assert len(root) == root_len
    out[root_addr:root_addr + root_len] = root
    # B-tree: one leaf-level node pointing at one symbol-table node (or none for an empty group)
    struct.pack_into("<4sBBHQQ", out, btree_addr, b"TREE", 0, 0, 1 if names else 0, UNDEF, UNDEF)
    if names:
        struct.pack_into("<QQQ", out, btree_addr + 24, 0, snod_addr, name_off[names[-1]])
    struct.pack_into("<4sB3xQQQ", out, heap_addr, b"HEAP", 0, len(heap_names), heap_free, heap_data_addr)
    out[heap_data_addr:heap_data_addr + len(heap_names)] = heap_names
    struct.pack_into("<4sBxH", out, snod_addr, b"SNOD", 1, len(names))
    for i, n in enumerate(names):
        struct.pack_into("<QQII16x", out, snod_addr + 8 + 40 * i, name_off[n], headers[n], 0, 0)
    if gheap_objs:
        struct.pack_into("<4sB3xQ", out, gheap_addr, b"GCOL", 1, GHEAP_SIZE)
        out[gheap_addr + 16:gheap_addr + 16 + len(gheap_body)] = gheap_body
        free_at = gheap_addr + 16 + len(gheap_body)
        struct.pack_into("<HH4xQ", out, free_at, 0, 0, gheap_addr + GHEAP_SIZE - free_at)
    for addr, blob in blobs:
        out[addr:addr + len(blob)] = blob
    with open(path, "wb") as f:
        f.write(out)
    return path
