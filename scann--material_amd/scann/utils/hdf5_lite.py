"""Minimal pure-Python reader for the subset of HDF5 that h5py / Keras 2.x write with default settings -- enough to read
a Keras full-model checkpoint (``ModelCheckpoint(save_weights_only=False)``, scann_model.py:166-177) without h5py or libhdf5:

* superblock version 0 / 1, version-1 object headers (+ continuation blocks), old-style groups (symbol table message ->
  version-1 B-tree -> symbol-table nodes -> local heap);
* datasets with contiguous, compact or CHUNKED layout (version-1 chunk B-tree) without filters -- what
  ``create_dataset(name, shape, dtype[, chunks=...])`` gives without compression; a filtered (compressed / shuffled) dataset
  raises -- little- or big-endian fixed-point and floating-point types, fixed-length strings;
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


#: what to do with a file this reader refuses: libhdf5's own tool rewrites it in the subset handled here
REPACK = ("re-write it with libhdf5's `h5repack --low=0 --high=0 -l CONTI -f NONE in.h5 out.h5` (oldest format, contiguous datasets, no "
          "filters), or on a machine with h5py: `python tools/keras_h5_to_container.py` reads the result")


class Hdf5Error(ValueError):
    """An HDF5 feature outside the subset above; the message names the feature and ``REPACK``, the way out."""

    def __init__(self, msg):
        super().__init__(msg + " -- " + REPACK if "h5repack" not in msg else msg)


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
            raise Hdf5Error("%s: superblock version %d (a file written with libver='latest': version-2 object headers, link-message "
                            "groups, fractal heaps); this reader handles superblock versions 0 and 1" % (path, ver))
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
            what = "version-2 object header ('OHDR')" if b[a:a + 4] == b"OHDR" else "object header version %d" % b[a]
            raise Hdf5Error("%s at %d: only version-1 headers (libver='earliest', the h5py / Keras default) are handled" % (what, addr))
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
                raise Hdf5Error("%s: new-style group (link / link-info messages, libver='latest' or track_order=True)" % name)
            elif mtype == 0x01:
                space = self._dataspace(body)
            elif mtype == 0x03:
                dtype = self._datatype(body)
            elif mtype == 0x08:
                layout = (body, size)
            elif mtype == 0x0B:
                raise Hdf5Error("%s: dataset with a filter pipeline (%s): not supported" % (name, self._filter_names(body)))
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
            if cls == 2:  # chunked: version-1 B-tree (node type 1) over the chunks
                ndim = b[p + 2]  # rank + 1: the last "dimension" is the element size
                btree = self._off(p + 3)
                q = p + 3 + self.O
                cdims = struct.unpack_from("<%dI" % ndim, b, q)
                if ndim - 1 != len(shape) or cdims[-1] != dt[1]:
                    raise Hdf5Error("chunked dataset: layout message disagrees with the dataspace / datatype")
                if dt[0] == "vlen_str":
                    raise Hdf5Error("chunked dataset of variable-length strings: not supported")
                out = np.zeros(shape, dtype=dt[0])
                if btree != _UNDEF >> (64 - 8 * self.O):
                    self._read_chunks(btree, cdims[:-1], dt, out)
                if out.dtype.byteorder == ">":
                    out = out.astype(out.dtype.newbyteorder("<"))
                return out
            raise Hdf5Error("data layout class %d (virtual dataset?): not supported" % cls)
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

    def _read_chunks(self, node, cdims, dt, out):
        """Walk a version-1 chunk B-tree (III.A.1, node type 1): key = chunk size, filter mask, offsets (rank + 1 x 8 bytes)."""
        b = self.buf
        a = self.base + node
        if b[a:a + 4] != b"TREE" or b[a + 4] != 1:
            raise Hdf5Error("bad chunk B-tree node")
        level, used = b[a + 5], struct.unpack_from("<H", b, a + 6)[0]
        rank = len(cdims)
        key = 8 + 8 * (rank + 1)
        p = a + 8 + 2 * self.O
        n_chunk = int(np.prod(cdims))
        for _ in range(used):
            size, mask = struct.unpack_from("<II", b, p)
            offs = struct.unpack_from("<%dQ" % rank, b, p + 8)
            child = self._off(p + key)
            p += key + self.O
            if level > 0:
                self._read_chunks(child, cdims, dt, out)
                continue
            if mask or size != n_chunk * dt[1]:
                raise Hdf5Error("filtered chunk (mask %#x, %d bytes for %d elements)" % (mask, size, n_chunk))
            chunk = np.frombuffer(b, dtype=dt[0], count=n_chunk, offset=self.base + child).reshape(cdims)
            sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, out.shape))
            out[sel] = chunk[tuple(slice(0, x.stop - x.start) for x in sel)]

    def _filter_names(self, p):
        """Names of the filters of a pipeline message (IV.A.2.l), for the error text."""
        known = {1: "deflate / gzip", 2: "shuffle", 3: "fletcher32", 4: "szip", 5: "nbit", 6: "scaleoffset", 32000: "lzf"}
        b = self.buf
        try:
            ver, n = b[p], b[p + 1]
            q = p + (8 if ver == 1 else 2)
            ids = []
            for _ in range(n):
                fid = struct.unpack_from("<H", b, q)[0]
                ids.append(known.get(fid, "filter %d" % fid))
                if ver == 1 or fid >= 256:
                    nlen = struct.unpack_from("<H", b, q + 2)[0]
                    nvals = struct.unpack_from("<H", b, q + 6)[0]
                    q += 8 + (nlen + 7) // 8 * 8 if ver == 1 else 8 + nlen
                else:
                    nvals = struct.unpack_from("<H", b, q + 4)[0]
                    q += 6
                q += 4 * nvals + (4 if ver == 1 and nvals % 2 else 0)
            return ", ".join(ids) or "unknown"
        except Exception:
            return "unknown"
