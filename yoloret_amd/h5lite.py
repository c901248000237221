"""Minimal HDF5 reader for Keras weights-only checkpoints (``model.save_weights('x.h5')`` /
``ModelCheckpoint(save_weights_only=True)``; reference code/train.py:74-91,182-186, read back by
``load_weights`` at code/yolo.py:87).

h5py / libhdf5 are not part of the runtime image, and the product must read the reference's
checkpoint format itself, so this module parses the subset of the HDF5 file format (HDF5 File Format
Specification 2.0/3.0, restated here from the public spec) that h5py's default settings produce:

  * superblock version 0/1 (and 2/3), 8-byte offsets and lengths;
  * old-style groups: symbol-table message -> v1 B-tree (node type 0) -> symbol-table nodes -> local heap;
  * new-style groups with compact link storage (Link messages in the object header);
  * version-1 and version-2 object headers with continuation blocks;
  * datasets: little-endian IEEE float32/float64 and integer types; compact, contiguous and chunked (v1 B-tree,
    node type 1) layouts; deflate (gzip) and shuffle filters;
  * attributes (message 0x000C, versions 1-3) holding fixed-length strings, variable-length strings (global
    heap) and numeric arrays - Keras keeps ``layer_names`` / ``weight_names`` there.

Anything else raises ``H5Error`` with the feature named.  Pure Python + NumPy + zlib.
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class _Type:
    """A datatype message: numpy dtype for numeric classes, ('S', n) / 'vlen-str' for strings."""

    def __init__(self, kind, size, dtype=None):
        self.kind, self.size, self.dtype = kind, size, dtype


class Dataset:
    def __init__(self, f, shape, typ, layout, filters, attrs):
        self._f, self.shape, self._type, self._layout, self._filters, self.attrs = f, shape, typ, layout, filters, attrs

    @property
    def dtype(self):
        return self._type.dtype

    def read(self):
        """The whole dataset as a NumPy array (C order)."""
        f, t = self._f, self._type
        if t.kind != 'num':
            raise H5Error('dataset of class %s: only numeric datasets are supported' % t.kind)
        n = int(np.prod(self.shape)) if self.shape else 1
        kind = self._layout[0]
        if kind == 'compact':
            raw = self._layout[1]
        elif kind == 'contiguous':
            addr, size = self._layout[1], self._layout[2]
            raw = b'\0' * (n * t.size) if addr == UNDEF else f._read(addr, n * t.size)
        else:
            return self._read_chunked(n)
        return np.frombuffer(raw[:n * t.size], t.dtype).reshape(self.shape).copy()

    def _read_chunked(self, n):
        f, t = self._f, self._type
        _, btree, cdims = self._layout          # cdims: chunk dims (elements) incl. the trailing element-size dim
        rank = len(self.shape)
        cshape = tuple(cdims[:rank])
        out = np.zeros(self.shape, t.dtype)
        if btree == UNDEF:
            return out
        for offs, size, mask, addr in f._chunk_btree(btree, rank):
            raw = f._read(addr, size)
            for i, (fid, _flags, cvals) in reversed(list(enumerate(self._filters))):   # undo the pipeline back to front
                if mask & (1 << i):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:   # shuffle: bytes of every element were transposed
                    es = cvals[0] if cvals else t.size
                    a = np.frombuffer(raw, np.uint8)
                    ne = a.size // es
                    raw = a[:ne * es].reshape(es, ne).T.tobytes() + a[ne * es:].tobytes()
                elif fid == 3:   # fletcher32: 4 trailing checksum bytes
                    raw = raw[:-4]
                else:
                    raise H5Error('filter id %d is not supported (deflate, shuffle, fletcher32 are)' % fid)
            chunk = np.frombuffer(raw[:int(np.prod(cshape)) * t.size], t.dtype).reshape(cshape)
            sl_out, sl_in = [], []
            for o, c, s in zip(offs, cshape, self.shape):
                e = min(o + c, s)
                sl_out.append(slice(o, e))
                sl_in.append(slice(0, e - o))
            out[tuple(sl_out)] = chunk[tuple(sl_in)]
        return out


class Group:
    def __init__(self, f, links, attrs):
        self._f, self._links, self.attrs = f, links, attrs

    def keys(self):
        return list(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            node = node._f._object(node._links[part])
        return node


class File(Group):
    def __init__(self, path_or_bytes):
        if isinstance(path_or_bytes, (bytes, bytearray)):
            self._b = bytes(path_or_bytes)
        else:
            with open(path_or_bytes, 'rb') as fh:
                self._b = fh.read()
        self._cache = {}
        root = self._superblock()
        g = self._object(root)
        if not isinstance(g, Group):
            raise H5Error('the root object is not a group')
        super().__init__(self, g._links, g.attrs)

    # ------------------------------------------------------------------ low level
    def _read(self, addr, n):
        addr += self._base
        if addr < 0 or addr + n > len(self._b):
            raise H5Error('read of %d bytes at %d runs past the end of the file (truncated?)' % (n, addr))
        return self._b[addr:addr + n]

    def _u(self, addr, n):
        return int.from_bytes(self._read(addr, n), 'little')

    def _superblock(self):
        sig = b'\x89HDF\r\n\x1a\n'
        pos = 0
        while pos < len(self._b) and self._b[pos:pos + 8] != sig:
            pos = 512 if pos == 0 else pos * 2
        if self._b[pos:pos + 8] != sig:
            raise H5Error('not an HDF5 file (signature not found)')
        self._base = 0
        ver = self._b[pos + 8]
        if ver in (0, 1):
            so, sl = self._b[pos + 13], self._b[pos + 14]
            if (so, sl) != (8, 8):
                raise H5Error('offsets/lengths of %d/%d bytes (only 8/8 supported)' % (so, sl))
            p = pos + 24 + (4 if ver == 1 else 0)
            self._base = int.from_bytes(self._b[p:p + 8], 'little')
            p += 32                                 # base, free-space, end-of-file, driver-info addresses
            return int.from_bytes(self._b[p + 8:p + 16], 'little')   # root symbol-table entry: object header address
        if ver in (2, 3):
            if (self._b[pos + 9], self._b[pos + 10]) != (8, 8):
                raise H5Error('only 8-byte offsets/lengths are supported')
            self._base = int.from_bytes(self._b[pos + 12:pos + 20], 'little')
            return int.from_bytes(self._b[pos + 36:pos + 44], 'little')
        raise H5Error('superblock version %d is not supported' % ver)

    # ------------------------------------------------------------------ object headers
    def _messages(self, addr):
        """-> list of (type, flags, body bytes) of the object header at addr (v1 or v2)."""
        msgs = []
        if self._read(addr, 4) == b'OHDR':
            flags = self._u(addr + 5, 1)
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szlen = 1 << (flags & 3)
            chunk0 = self._u(p, szlen)
            p += szlen
            blocks = [(p, chunk0)]
            tracked = bool(flags & 0x04)
            seen = {p}
            while blocks:
                p, n = blocks.pop(0)
                end = p + n
                while p + 4 <= end:
                    mtype, msize, mflags = self._u(p, 1), self._u(p + 1, 2), self._u(p + 3, 1)
                    p += 4 + (2 if tracked else 0)
                    body = self._read(p, msize)
                    p += msize
                    if mtype == 0x10:
                        caddr, clen = struct.unpack_from('<QQ', body)
                        if self._read(caddr, 4) != b'OCHK':
                            raise H5Error('bad object-header continuation block')
                        if caddr + 4 in seen or len(seen) > 4096:
                            raise H5Error('object-header continuation blocks form a cycle')
                        seen.add(caddr + 4)
                        blocks.append((caddr + 4, clen - 8))       # minus signature and checksum
                    elif mtype != 0:
                        msgs.append((mtype, mflags, body))
            return msgs
        ver = self._u(addr, 1)
        if ver != 1:
            raise H5Error('object header version %d at %d is not supported' % (ver, addr))
        nmsg = self._u(addr + 2, 2)
        hsize = self._u(addr + 8, 4)
        blocks = [(addr + 16, hsize)]
        seen = {addr + 16}
        while blocks and len(msgs) < nmsg + 64:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end:
                mtype, msize, mflags = self._u(p, 2), self._u(p + 2, 2), self._u(p + 4, 1)
                body = self._read(p + 8, msize)
                p += 8 + msize
                if mtype == 0x10:
                    caddr, clen = struct.unpack_from('<QQ', body)
                    if caddr in seen or len(seen) > 4096:     # a block pointing back at itself would never end
                        raise H5Error('object-header continuation blocks form a cycle')
                    seen.add(caddr)
                    blocks.append((caddr, clen))
                elif mtype != 0:
                    msgs.append((mtype, mflags, body))
        return msgs

    def _object(self, addr):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._messages(addr)
        attrs, links = {}, None
        shape = typ = layout = None
        filters = []
        for mtype, mflags, body in msgs:
            if mflags & 0x02:
                raise H5Error('shared object-header messages are not supported')
            if mtype == 0x01:
                shape = self._dataspace(body)
            elif mtype == 0x03:
                typ = self._datatype(body)[0]
            elif mtype == 0x08:
                layout = self._layout(body)
            elif mtype == 0x0B:
                filters = self._pipeline(body)
            elif mtype == 0x0C:
                k, v = self._attribute(body)
                attrs[k] = v
            elif mtype == 0x11:
                btree, heap = struct.unpack_from('<QQ', body)
                links = self._old_group(btree, heap)
            elif mtype == 0x06:
                links = links or {}
                name, target = self._link(body)
                links[name] = target
            elif mtype == 0x02:
                links = links or {}
                fheap = struct.unpack_from('<Q', body, 2 + (8 if body[1] & 1 else 0))[0]
                if fheap != UNDEF:
                    raise H5Error('groups with dense link storage (fractal heap) are not supported')
            elif mtype == 0x15:
                if struct.unpack_from('<Q', body, 2 + (2 if body[1] & 1 else 0))[0] != UNDEF:
                    raise H5Error('dense attribute storage (fractal heap) is not supported')
        if layout is not None and typ is not None:
            obj = Dataset(self, shape if shape is not None else (), typ, layout, filters, attrs)
        else:
            obj = Group(self, links or {}, attrs)
        self._cache[addr] = obj
        return obj

    # ------------------------------------------------------------------ messages
    @staticmethod
    def _dataspace(body):
        ver, rank, flags = body[0], body[1], body[2]
        p = 8 if ver == 1 else 4
        if ver == 2 and body[3] == 2:
            return None                              # null dataspace
        return tuple(struct.unpack_from('<%dQ' % rank, body, p)) if rank else ()

    def _datatype(self, body):
        """-> (_Type, bytes consumed)."""
        cls, ver = body[0] & 0x0F, body[0] >> 4
        b0, b1 = body[1], body[2]
        size = struct.unpack_from('<I', body, 4)[0]
        if cls == 0:      # fixed point
            if b0 & 1:
                raise H5Error('big-endian integers are not supported')
            return _Type('num', size, np.dtype('<%s%d' % ('i' if b0 & 8 else 'u', size))), 8 + 4
        if cls == 1:      # floating point
            if b0 & 1:
                raise H5Error('big-endian floats are not supported')
            if size not in (2, 4, 8):
                raise H5Error('%d-byte floats are not supported' % size)
            return _Type('num', size, np.dtype('<f%d' % size)), 8 + 12
        if cls == 3:      # fixed-length string
            return _Type('str', size), 8
        if cls == 9:      # variable length
            base, used = self._datatype(body[8:])
            if (b0 & 0x0F) == 1:
                return _Type('vstr', size), 8 + used
            raise H5Error('variable-length sequences are not supported')
        raise H5Error('datatype class %d is not supported' % cls)

    @staticmethod
    def _layout(body):
        ver = body[0]
        if ver == 4 and body[1] == 2:
            raise H5Error('chunked datasets of layout version 4 (libver="latest" chunk indexes) are not supported')
        if ver in (3, 4):
            cls = body[1]
            if cls == 0:
                n = struct.unpack_from('<H', body, 2)[0]
                return ('compact', body[4:4 + n])
            if cls == 1:
                addr, size = struct.unpack_from('<QQ', body, 2)
                return ('contiguous', addr, size)
            if cls == 2:
                rank = body[2]
                addr = struct.unpack_from('<Q', body, 3)[0]
                dims = struct.unpack_from('<%dI' % rank, body, 11)
                return ('chunked', addr, dims)
        if ver in (1, 2):
            rank, cls = body[1], body[2]
            p = 8
            addr = UNDEF
            if cls != 0:
                addr = struct.unpack_from('<Q', body, p)[0]
                p += 8
            dims = struct.unpack_from('<%dI' % rank, body, p)
            p += 4 * rank
            if cls == 1:
                return ('contiguous', addr, None)
            if cls == 2:
                return ('chunked', addr, dims)
            n = struct.unpack_from('<I', body, p)[0]
            return ('compact', body[p + 4:p + 4 + n])
        raise H5Error('data layout message version %d (class %d) is not supported' % (ver, body[1]))

    @staticmethod
    def _pipeline(body):
        ver, nf = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(nf):
            fid, = struct.unpack_from('<H', body, p)
            if ver == 1 or fid >= 256:
                nlen, flags, ncv = struct.unpack_from('<HHH', body, p + 2)
                p += 8
                p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            else:
                flags, ncv = struct.unpack_from('<HH', body, p + 2)
                p += 6
            cv = struct.unpack_from('<%dI' % ncv, body, p)
            p += 4 * ncv
            if ver == 1 and ncv % 2:
                p += 4
            out.append((fid, flags, cv))
        return out

    def _attribute(self, body):
        ver = body[0]
        nlen, tlen, slen = struct.unpack_from('<HHH', body, 2)
        p = 8
        if ver == 3:
            p = 9
        pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
        name = body[p:p + nlen].split(b'\0')[0].decode('utf8')
        p += pad(nlen)
        typ, _ = self._datatype(body[p:p + tlen])
        p += pad(tlen)
        shape = self._dataspace(body[p:p + slen])
        p += pad(slen)
        data = body[p:]
        n = int(np.prod(shape)) if shape else 1
        if shape is None:
            return name, None
        if typ.kind == 'num':
            a = np.frombuffer(data[:n * typ.size], typ.dtype).reshape(shape).copy()
            return name, (a if shape else a[()])
        if typ.kind == 'str':
            vals = [data[i * typ.size:(i + 1) * typ.size].split(b'\0')[0] for i in range(n)]
        else:   # variable-length strings: (length u32, global heap collection address u64, object index u32)
            vals = []
            for i in range(n):
                ln, gaddr, idx = struct.unpack_from('<IQI', data, 16 * i)
                vals.append(self._global_heap(gaddr, idx)[:ln] if gaddr not in (0, UNDEF) else b'')
        if not shape:
            return name, vals[0]
        return name, np.array(vals, dtype=object).reshape(shape)

    def _global_heap(self, addr, index):
        if self._read(addr, 4) != b'GCOL':
            raise H5Error('bad global heap collection')
        size = self._u(addr + 8, 8)
        p, end = addr + 16, addr + size
        while p + 16 <= end:
            idx, _ref, _res, osize = struct.unpack('<HHIQ', self._read(p, 16))
            if idx == index:
                return self._read(p + 16, osize)
            if idx == 0:
                break
            p += 16 + (osize + 7) // 8 * 8
        raise H5Error('global heap object %d not found' % index)

    @staticmethod
    def _link(body):
        ver, flags = body[0], body[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        szlen = 1 << (flags & 3)
        nlen = int.from_bytes(body[p:p + szlen], 'little')
        p += szlen
        name = body[p:p + nlen].decode('utf8')
        p += nlen
        if ltype != 0:
            raise H5Error('soft / external links are not supported (link %r)' % name)
        return name, struct.unpack_from('<Q', body, p)[0]

    # ------------------------------------------------------------------ old-style groups, chunk index
    def _old_group(self, btree, heap):
        if self._read(heap, 4) != b'HEAP':
            raise H5Error('bad local heap')
        data_addr = self._u(heap + 24, 8)
        links = {}

        def name_at(off):
            end = self._b.index(b'\0', self._base + data_addr + off)
            return self._b[self._base + data_addr + off:end].decode('utf8')

        def walk(addr):
            sig = self._read(addr, 4)
            if sig == b'TREE':
                ntype, level, used = self._u(addr + 4, 1), self._u(addr + 5, 1), self._u(addr + 6, 2)
                if ntype != 0:
                    raise H5Error('group B-tree of node type %d' % ntype)
                p = addr + 24
                for i in range(used):
                    walk(self._u(p + 8 + 16 * i, 8))            # key_i (8), child_i (8), ...
            elif sig == b'SNOD':
                n = self._u(addr + 6, 2)
                p = addr + 8
                for i in range(n):
                    off, ohdr = struct.unpack('<QQ', self._read(p + 40 * i, 16))
                    links[name_at(off)] = ohdr
            else:
                raise H5Error('unexpected block %r in a group B-tree' % sig)
        walk(btree)
        return links

    def _chunk_btree(self, addr, rank):
        """Yields (element offsets[rank], stored size, filter mask, address) for every chunk."""
        if self._read(addr, 4) != b'TREE':
            raise H5Error('bad chunk B-tree node')
        ntype, level, used = self._u(addr + 4, 1), self._u(addr + 5, 1), self._u(addr + 6, 2)
        if ntype != 1:
            raise H5Error('chunk B-tree of node type %d' % ntype)
        ksize = 8 + 8 * (rank + 1)
        p = addr + 24
        for i in range(used):
            key = self._read(p + i * (ksize + 8), ksize)
            size, mask = struct.unpack_from('<II', key)
            offs = struct.unpack_from('<%dQ' % rank, key, 8)
            child = self._u(p + i * (ksize + 8) + ksize, 8)
            if level == 0:
                yield offs, size, mask, child
            else:
                yield from self._chunk_btree(child, rank)


# ---------------------------------------------------------------------------------------- Keras layout
def read_keras_weights(path_or_bytes):
    """-> ordered {layer name: {weight base name ('kernel', 'gamma', ...): float32 array}} from a Keras weights-only
    HDF5 file (or the ``model_weights`` group of a full-model file): root/group attribute ``layer_names``; per layer
    a group with attribute ``weight_names`` (e.g. b'conv2d_3/kernel:0') naming its datasets."""
    try:
        return _read_keras_weights(path_or_bytes)
    except H5Error:
        raise
    except (ValueError, IndexError, KeyError, struct.error, zlib.error, OverflowError, RecursionError) as e:   # (RecursionError: a cyclic B-tree)
        raise H5Error('corrupt or truncated HDF5 file (%s: %s)' % (type(e).__name__, e))


def _read_keras_weights(path_or_bytes):
    f = File(path_or_bytes)
    root = f['model_weights'] if 'layer_names' not in f.attrs and 'layer_names0' not in f.attrs and 'model_weights' in f else f

    def strs(v):
        return [x.decode('utf8') if isinstance(x, bytes) else str(x) for x in np.asarray(v, dtype=object).ravel()]

    def names(attrs, key):
        """Keras splits a name list that would exceed HDF5's 64 KB attribute limit into `key`0, `key`1, ...
        (keras/saving/hdf5_format.py: save_attributes_to_hdf5_group)."""
        if key in attrs:
            return strs(attrs[key])
        out, i = [], 0
        while '%s%d' % (key, i) in attrs:
            out += strs(attrs['%s%d' % (key, i)])
            i += 1
        return out if i else None
    lnames = names(root.attrs, 'layer_names')
    if lnames is None:
        raise H5Error('no "layer_names" attribute: not a Keras weights file')
    out = {}
    for lname in lnames:
        g = root[lname]
        wd = {}
        for wname in names(g.attrs, 'weight_names') or []:
            arr = g[wname].read()
            base = wname.split('/')[-1].split(':')[0]
            if base in wd:
                raise H5Error('layer %s has two weights called %s' % (lname, base))
            wd[base] = np.asarray(arr, np.float32)
        out[lname] = wd
    return out
