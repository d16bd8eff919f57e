"""Minimal pure-Python reader for the subset of HDF5 that h5py / Keras 2.x write with default settings -- enough to read
a Keras full-model checkpoint (``ModelCheckpoint(save_weights_only=False)``, scann_model.py:166-177) without h5py or libhdf5:

* superblock version 0 / 1, version-1 object headers (+ continuation blocks), old-style groups (symbol table message ->
  version-1 B-tree -> symbol-table nodes -> local heap);
* datasets with contiguous or compact layout (what ``create_dataset(name, shape, dtype)`` gives without chunking or
  compression; a chunked or filtered dataset raises), little- or big-endian fixed-point and floating-point types, fixed-length
  strings;
* attributes (message versions 1-3) of those types, of variable-length strings (global heap), scalar or array.

File-format reference: "HDF5 File Format Specification Version 2.0" (sections III.A superblock, III.B/III.D B-trees and
heaps, IV.A object headers and messages).  Checked against files written by h5py 3.3 / libhdf5 1.10.6
(tools/make_keras_h5_fixture.py, tests/test_keras_import.py).
"""
from __future__ import annotations

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


class Dataset:
    def __init__(self, f, name, dtype, shape, reader, attrs):
        self.file, self.name, self.dtype, self.shape, self._reader, self.attrs = f, name, dtype, shape, reader, attrs

    def read(self):
        return self._reader()

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a.astype(dtype) if dtype is not None else a


class Group:
    def __init__(self, f, name, links, attrs):
        self.file, self.name, self._links, self.attrs = f, name, links, attrs
        self._cache = {}

    def keys(self):
        return list(self._links)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __iter__(self):
        return iter(self._links)

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            if part not in node._cache:
                node._cache[part] = node.file._object(node._links[part], (node.name.rstrip("/") + "/" + part))
            node = node._cache[part]
        return node

    def visit_datasets(self, prefix=""):
        """Yield (path relative to this group, Dataset) for every dataset below it, in link-name order per group."""
        for k in self._links:
            child = self[k]
            if isinstance(child, Group):
                yield from child.visit_datasets(prefix + k + "/")
            else:
                yield prefix + k, child


class File(Group):
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        if not b.startswith(_SIG):
            raise Hdf5Error("%s: not an HDF5 file (or a user block precedes the superblock: not supported)" % path)
        ver = b[8]
        if ver not in (0, 1):
            raise Hdf5Error("%s: superblock version %d (written with libver='latest'); this reader handles versions 0 and 1" % (path, ver))
        self.O, self.L = b[13], b[14]  # size of offsets / lengths
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise Hdf5Error("unsupported offset / length size")
        p = 24 + (4 if ver == 1 else 0)
        self.base = self._off(p)
        p += 4 * self.O  # base, free-space info, end of file, driver info
        root_hdr = self._off(p + self.O)  # root symbol-table entry: link-name offset, object header address, ...
        root = self._object(root_hdr, "/")
        if not isinstance(root, Group):
            raise Hdf5Error("root object is not a group")
        Group.__init__(self, self, "/", root._links, root.attrs)

    # -- primitive readers ---------------------------------------------------------------------------------------------
    def _off(self, p):
        return int.from_bytes(self.buf[p:p + self.O], "little")

    def _len(self, p):
        return int.from_bytes(self.buf[p:p + self.L], "little")

    # -- object headers ------------------------------------------------------------------------------------------------
    def _messages(self, addr):
        b = self.buf
        a = self.base + addr
        if b[a] != 1:
            raise Hdf5Error("object header version %d at %d: only version-1 headers (libver='earliest', the h5py default) are handled" % (b[a], addr))
        n_msg = struct.unpack_from("<H", b, a + 2)[0]
        size = struct.unpack_from("<I", b, a + 8)[0]
        blocks = [(a + 16, size)]
        out = []
        while blocks and len(out) < n_msg:
            p, remaining = blocks.pop(0)
            end = p + remaining
            while p + 8 <= end and len(out) < n_msg:
                mtype, msize, flags = struct.unpack_from("<HHB", b, p)
                body = p + 8
                if mtype == 0x10:  # continuation
                    blocks.append((self.base + self._off(body), self._len(body + self.O)))
                out.append((mtype, flags, body, msize))
                p = body + msize
        return out

    def _object(self, addr, name):
        links, attrs = None, {}
        space = dtype = layout = None
        for mtype, flags, body, size in self._messages(addr):
            if mtype == 0x11:  # symbol table -> old-style group
                links = self._group_links(self._off(body), self._off(body + self.O))
            elif mtype == 0x02 or mtype == 0x06:
                raise Hdf5Error("%s: new-style (link-message) group; re-save with h5py defaults" % name)
            elif mtype == 0x01:
                space = self._dataspace(body)
            elif mtype == 0x03:
                dtype = self._datatype(body)
            elif mtype == 0x08:
                layout = (body, size)
            elif mtype == 0x0B:
                raise Hdf5Error("%s: filtered (compressed) dataset: not supported" % name)
            elif mtype == 0x0C:
                k, v = self._attribute(body)
                attrs[k] = v
        if links is not None:
            return Group(self, name, links, attrs)
        if space is None or dtype is None or layout is None:
            raise Hdf5Error("%s: neither a group nor a complete dataset" % name)
        return Dataset(self, name, dtype[0], space, lambda: self._read_layout(layout[0], dtype, space), attrs)

    # -- groups --------------------------------------------------------------------------------------------------------
    def _heap_name(self, heap_data, off):
        end = self.buf.index(b"\0", heap_data + off)
        return self.buf[heap_data + off:end].decode("utf-8")

    def _group_links(self, btree, heap):
        b = self.buf
        h = self.base + heap
        if b[h:h + 4] != b"HEAP":
            raise Hdf5Error("bad local heap signature")
        heap_data = self.base + self._off(h + 8 + 2 * self.L)
        links = {}

        def walk(node):
            a = self.base + node
            if b[a:a + 4] != b"TREE":
                raise Hdf5Error("bad B-tree signature")
            level, used = b[a + 5], struct.unpack_from("<H", b, a + 6)[0]
            p = a + 8 + 2 * self.O
            for i in range(used):
                child = self._off(p + self.L)  # key_i (L bytes), child_i (O bytes), ...
                p += self.L + self.O
                if level > 0:
                    walk(child)
                else:
                    s = self.base + child
                    if b[s:s + 4] != b"SNOD":
                        raise Hdf5Error("bad symbol-table node signature")
                    n = struct.unpack_from("<H", b, s + 6)[0]
                    e = s + 8
                    for _ in range(n):
                        links[self._heap_name(heap_data, self._off(e))] = self._off(e + self.O)
                        e += 2 * self.O + 24
        walk(btree)
        return links

    # -- dataspace / datatype ------------------------------------------------------------------------------------------
    def _dataspace(self, p):
        b = self.buf
        ver, rank, flags = b[p], b[p + 1], b[p + 2]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if b[p + 3] == 2:
                return None  # null dataspace
            q = p + 4
        else:
            raise Hdf5Error("dataspace version %d" % ver)
        return tuple(self._len(q + i * self.L) for i in range(rank))

    def _datatype(self, p):
        """-> (numpy dtype | 'vlen_str', byte size, message length)"""
        b = self.buf
        cls, ver = b[p] & 0x0F, b[p] >> 4
        bits0 = b[p + 1]
        size = struct.unpack_from("<I", b, p + 4)[0]
        if cls == 0:  # fixed point
            order = ">" if bits0 & 1 else "<"
            signed = bool(bits0 & 0x08)
            return np.dtype("%s%s%d" % (order, "i" if signed else "u", size)), size, 8 + 4
        if cls == 1:  # floating point
            order = ">" if bits0 & 1 else "<"
            return np.dtype("%sf%d" % (order, size)), size, 8 + 12
        if cls == 3:  # fixed-length string
            return np.dtype("S%d" % size), size, 8
        if cls == 9:  # variable length
            if (bits0 & 0x0F) == 1:
                return "vlen_str", 4 + self.O + 4, None
            raise Hdf5Error("variable-length sequence datatype: not supported")
        raise Hdf5Error("datatype class %d (version %d): not supported" % (cls, ver))

    def _decode(self, raw, dt, shape):
        dtype, size, _ = dt
        n = int(np.prod(shape)) if shape else 1
        if dtype == "vlen_str":
            out = []
            for i in range(n):
                p = i * size
                ln = struct.unpack_from("<I", raw, p)[0]
                coll = int.from_bytes(raw[p + 4:p + 4 + self.O], "little")
                idx = struct.unpack_from("<I", raw, p + 4 + self.O)[0]
                out.append(self._global_heap(coll, idx)[:ln].decode("utf-8"))
            return out[0] if not shape else np.array(out, dtype=object).reshape(shape)
        a = np.frombuffer(raw, dtype=dtype, count=n)
        if dtype.kind in "fiu" and dtype.byteorder == ">":
            a = a.astype(dtype.newbyteorder("<"))
        return a.reshape(shape).copy() if shape else a[0]

    def _global_heap(self, coll, idx):
        b = self.buf
        a = self.base + coll
        if b[a:a + 4] != b"GCOL":
            raise Hdf5Error("bad global heap signature")
        size = self._len(a + 8)
        p, end = a + 8 + self.L, a + size
        while p + 8 + self.L <= end:
            oid = struct.unpack_from("<H", b, p)[0]
            osz = self._len(p + 8)
            if oid == idx:
                return b[p + 8 + self.L:p + 8 + self.L + osz]
            if oid == 0:
                break
            p += 8 + self.L + (osz + 7) // 8 * 8
        raise Hdf5Error("global heap object %d not found" % idx)

    # -- attributes ----------------------------------------------------------------------------------------------------
    def _attribute(self, p):
        b = self.buf
        ver = b[p]
        nsz, tsz, ssz = struct.unpack_from("<HHH", b, p + 2)
        if ver == 1:
            pad = lambda x: (x + 7) // 8 * 8  # noqa: E731
            q = p + 8
        elif ver in (2, 3):
            if b[p + 1] & 3:
                raise Hdf5Error("shared attribute datatype / dataspace: not supported")
            pad = lambda x: x  # noqa: E731
            q = p + 8 + (1 if ver == 3 else 0)
        else:
            raise Hdf5Error("attribute message version %d" % ver)
        name = b[q:q + nsz].split(b"\0")[0].decode("utf-8")
        q += pad(nsz)
        dt = self._datatype(q)
        q += pad(tsz)
        shape = self._dataspace(q)
        q += pad(ssz)
        if shape is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        return name, self._decode(b[q:q + n * dt[1]], dt, shape)

    # -- dataset storage -----------------------------------------------------------------------------------------------
    def _read_layout(self, p, dt, shape):
        b = self.buf
        ver = b[p]
        n = (int(np.prod(shape)) if shape else 1) * dt[1]
        if ver == 3:
            cls = b[p + 1]
            if cls == 0:  # compact
                sz = struct.unpack_from("<H", b, p + 2)[0]
                return self._decode(b[p + 4:p + 4 + sz], dt, shape)
            if cls == 1:  # contiguous
                addr = self._off(p + 2)
                if addr == _UNDEF >> (64 - 8 * self.O):
                    return np.zeros(shape, dtype=dt[0])  # never written: fill value
                return self._decode(b[self.base + addr:self.base + addr + n], dt, shape)
            raise Hdf5Error("chunked dataset: not supported (Keras writes its weights contiguous)")
        if ver in (1, 2):
            rank, cls = b[p + 1], b[p + 2]
            if cls == 1:
                addr = self._off(p + 8)
                return self._decode(b[self.base + addr:self.base + addr + n], dt, shape)
            if cls == 0:
                q = p + 8 + 4 * rank
                sz = struct.unpack_from("<I", b, q)[0]
                return self._decode(b[q + 4:q + 4 + sz], dt, shape)
        raise Hdf5Error("data layout version %d / class not supported" % ver)
