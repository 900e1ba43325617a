"""Dependency-free reader and writer for the HDF5 subset Keras weight files use (numpy only: no h5py, no torch).

The reference saves and loads its checkpoints through Keras (`predictive_net.save_weights` / `load_weights`,
speechless/net.py:209-212, 558-572), i.e. HDF5 files written by h5py with the library's default ("earliest") format:

    superblock version 0 - groups as symbol tables (B-tree v1 + local heap + SNOD nodes) - version-1 object headers with
    continuation blocks - contiguous (or compact) datasets of little-endian IEEE floats / integers - attributes (versions 1-3)
    holding fixed-length or variable-length strings, scalars or 1-D arrays

`read(path)` returns the whole tree; `write(path, tree)` produces a file of the same subset that the real library reads
(tests/test_h5lite.py checks both directions against files written / read by h5py under an interpreter that has it).
Everything outside the subset (chunked or filtered datasets, version-2 object headers, dense link storage) raises
`H5Unsupported` with the name of the feature instead of guessing.  HDF5 File Format Specification version 2.0/3.0,
sections II (superblock), III.A (B-trees, version 1), III.B (symbol table nodes), III.D (local heaps), III.E (global
heap), IV.A.1.a (version-1 object headers), IV.A.2 (header messages).
"""
import struct
from collections import OrderedDict

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Unsupported(ValueError):
    pass


class Group(OrderedDict):
    """name -> Group | Dataset, in the file's (alphabetical) link order; `.attrs` holds the attributes."""

    def __init__(self):
        super().__init__()
        self.attrs = OrderedDict()


class Dataset:
    def __init__(self, value, attrs=None):
        self.value = value
        self.attrs = attrs if attrs is not None else OrderedDict()


# ---------------------------------------------------------------------------------------------------------------- reading
class _Reader:
    def __init__(self, data):
        self.d = data
        if data[:8] != SIGNATURE:
            raise ValueError("not an HDF5 file (signature missing at offset 0)")
        version = data[8]
        if version not in (0, 1):
            raise H5Unsupported("superblock version {} (only the version-0/1 layout h5py writes by default)".format(version))
        self.so, self.sl = data[13], data[14]  # size of offsets / lengths
        if (self.so, self.sl) != (8, 8):
            raise H5Unsupported("{}-byte offsets / {}-byte lengths".format(self.so, self.sl))
        pos = 24 + (4 if version == 1 else 0)
        self.base = self.u64(pos)
        root_entry = pos + 4 * 8
        self.root_header = self.u64(root_entry + 8)

    def u16(self, p):
        return struct.unpack_from("<H", self.d, p)[0]

    def u32(self, p):
        return struct.unpack_from("<I", self.d, p)[0]

    def u64(self, p):
        return struct.unpack_from("<Q", self.d, p)[0]

    # ---- object headers
    def messages(self, addr):
        """[(type, flags, bytes)] of the version-1 object header at addr, continuation blocks followed."""
        d = self.d
        addr += self.base
        if d[addr:addr + 4] == b"OHDR":
            raise H5Unsupported("version-2 object headers (file written with libver='latest')")
        if d[addr] != 1:
            raise H5Unsupported("object header version {}".format(d[addr]))
        count = self.u16(addr + 2)
        size = self.u32(addr + 8)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < count:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < count:
                mtype, msize, flags = self.u16(pos), self.u16(pos + 2), d[pos + 4]
                body = d[pos + 8: pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:  # continuation: offset, length
                    blocks.append((struct.unpack_from("<Q", body, 0)[0] + self.base, struct.unpack_from("<Q", body, 8)[0]))
                out.append((mtype, flags, body))
        return out

    # ---- datatypes / dataspaces
    def datatype(self, b):
        """-> (kind, numpy dtype or None, element size, extra)"""
        cls, version = b[0] & 15, b[0] >> 4
        bits0 = b[1]
        size = struct.unpack_from("<I", b, 4)[0]
        if cls == 0:  # fixed point
            if bits0 & 1:
                raise H5Unsupported("big-endian integers")
            return "num", np.dtype("<{}{}".format("i" if bits0 & 8 else "u", size)), size, None
        if cls == 1:  # floating point
            if bits0 & 1:
                raise H5Unsupported("big-endian floats")
            if size not in (2, 4, 8):
                raise H5Unsupported("{}-byte floats".format(size))
            return "num", np.dtype("<f{}".format(size)), size, None
        if cls == 3:  # fixed-length string
            return "str", None, size, None
        if cls == 9:  # variable length
            if (bits0 & 15) != 1:
                raise H5Unsupported("variable-length sequences")
            return "vstr", None, size, None
        raise H5Unsupported("datatype class {} (version {})".format(cls, version))

    def dataspace(self, b):
        version, rank, flags = b[0], b[1], b[2]
        if version == 1:
            pos = 8
        elif version == 2:
            if b[3] == 2:  # null dataspace
                return None
            pos = 4
        else:
            raise H5Unsupported("dataspace version {}".format(version))
        return tuple(struct.unpack_from("<Q", b, pos + 8 * i)[0] for i in range(rank))

    def vlen_string(self, b):
        length, addr, index = struct.unpack_from("<IQI", b, 0)
        if length == 0 and addr == 0:
            return ""
        d = self.d
        addr += self.base
        if d[addr:addr + 4] != b"GCOL":
            raise ValueError("global heap collection signature missing")
        end = addr + self.u64(addr + 8)
        pos = addr + 16
        while pos + 16 <= end:
            obj_index, obj_size = self.u16(pos), self.u64(pos + 8)
            if obj_index == index:
                return bytes(d[pos + 16: pos + 16 + length]).decode("utf8", "replace")
            if obj_index == 0:
                break
            pos += 16 + (obj_size + 7) // 8 * 8
        raise ValueError("global heap object {} not found".format(index))

    def decode(self, kind, dtype, esize, shape, raw):
        n = 1 if shape is None or shape == () else int(np.prod(shape))
        if shape is None:
            n = 0
        if kind == "num":
            arr = np.frombuffer(raw, dtype=dtype, count=n).copy()
            return arr.reshape(shape) if shape else (arr[0] if n else arr)
        if kind == "str":
            items = [bytes(raw[i * esize:(i + 1) * esize]).split(b"\x00", 1)[0].decode("utf8", "replace") for i in range(n)]
        else:
            items = [self.vlen_string(raw[i * esize:(i + 1) * esize]) for i in range(n)]
        if shape == ():
            return items[0]
        return items

    def attribute(self, b):
        version = b[0]
        name_size, dt_size, ds_size = struct.unpack_from("<HHH", b, 2)
        pos = 8 if version < 3 else 9

        def take(size, padded):
            nonlocal pos
            chunk = b[pos: pos + size]
            pos += (size + 7) // 8 * 8 if padded else size
            return chunk
        if version not in (1, 2, 3):
            raise H5Unsupported("attribute message version {}".format(version))
        if version >= 2 and (b[1] & 3):
            raise H5Unsupported("shared datatype / dataspace in an attribute")
        padded = version == 1
        name = bytes(take(name_size, padded)).split(b"\x00", 1)[0].decode("utf8")
        kind, dtype, esize, _ = self.datatype(take(dt_size, padded))
        shape = self.dataspace(take(ds_size, padded))
        return name, self.decode(kind, dtype, esize, shape, b[pos:])

    # ---- groups
    def heap_string(self, heap_addr, offset):
        d = self.d
        heap_addr += self.base
        if d[heap_addr:heap_addr + 4] != b"HEAP":
            raise ValueError("local heap signature missing")
        seg = self.u64(heap_addr + 24) + self.base
        end = d.index(b"\x00", seg + offset)
        return bytes(d[seg + offset:end]).decode("utf8")

    def group_entries(self, btree, heap):
        """(name, object header address) of every link below the version-1 group B-tree at `btree`"""
        d = self.d
        node = btree + self.base
        if d[node:node + 4] == b"SNOD":
            n = self.u16(node + 6)
            out = []
            for i in range(n):
                e = node + 8 + i * 40
                out.append((self.heap_string(heap, self.u64(e)), self.u64(e + 8)))
            return out
        if d[node:node + 4] != b"TREE":
            raise ValueError("B-tree node signature missing")
        if d[node + 4] != 0:
            raise H5Unsupported("B-tree node type {} where a group node was expected".format(d[node + 4]))
        used = self.u16(node + 6)
        out = []
        pos = node + 8 + 16
        for i in range(used):
            child = self.u64(pos + 8)  # key i (8 bytes), child i (8 bytes)
            out += self.group_entries(child, heap)
            pos += 16
        return out

    def obj(self, header_addr):
        msgs = self.messages(header_addr)
        attrs = OrderedDict()
        symtab = layout = dtype = space = None
        for mtype, flags, body in msgs:
            if mtype == 0x000C:
                name, value = self.attribute(body)
                attrs[name] = value
            elif mtype == 0x0011:
                symtab = struct.unpack_from("<QQ", body, 0)
            elif mtype == 0x0008:
                layout = body
            elif mtype == 0x0003:
                dtype = body
            elif mtype == 0x0001:
                space = body
            elif mtype in (0x0002, 0x0006):
                raise H5Unsupported("link messages (dense / compact link storage of libver='latest')")
            elif mtype == 0x000B:
                raise H5Unsupported("filtered (compressed) datasets")
            elif mtype == 0x0015:
                raise H5Unsupported("dense attribute storage")
        if symtab is not None:
            g = Group()
            g.attrs = attrs
            for name, addr in self.group_entries(symtab[0], symtab[1]):
                g[name] = self.obj(addr)
            return g
        if layout is None or dtype is None or space is None:
            raise H5Unsupported("object that is neither a symbol-table group nor a dataset")
        kind, np_dtype, esize, _ = self.datatype(dtype)
        shape = self.dataspace(space)
        n = 0 if shape is None else (int(np.prod(shape)) if shape else 1)
        version = layout[0]
        if version == 3:
            cls = layout[1]
            if cls == 1:
                addr, size = struct.unpack_from("<QQ", layout, 2)
                raw = b"" if addr == UNDEF else self.d[addr + self.base: addr + self.base + size]
            elif cls == 0:
                size = struct.unpack_from("<H", layout, 2)[0]
                raw = layout[4:4 + size]
            else:
                raise H5Unsupported("chunked datasets")
        elif version in (1, 2):
            rank, cls = layout[1], layout[2]
            if cls != 1:
                raise H5Unsupported("layout class {} in a version-{} layout message".format(cls, version))
            addr = struct.unpack_from("<Q", layout, 8)[0]
            raw = self.d[addr + self.base: addr + self.base + n * esize]
        else:
            raise H5Unsupported("data layout message version {}".format(version))
        if len(raw) < n * esize:  # never written: fill value 0
            raw = bytes(raw) + b"\x00" * (n * esize - len(raw))
        return Dataset(self.decode(kind, np_dtype, esize, shape, raw), attrs)


def read(path):
    """The whole file as a tree: Group (an OrderedDict name -> Group | Dataset, with .attrs) at the root."""
    with open(str(path), "rb") as f:
        data = f.read()
    r = _Reader(data)
    root = r.obj(r.root_header)
    if not isinstance(root, Group):
        raise ValueError("root object is not a group")
    return root


# ---------------------------------------------------------------------------------------------------------------- writing
class _Writer:
    """Appends structures to a bytearray; every structure starts 8-byte aligned."""
    LEAF_K = 64  # symbol table nodes hold up to 2 * LEAF_K links: one node per group (Keras models have < 128 layers)

    def __init__(self):
        self.buf = bytearray(96)  # superblock (version 0, 8-byte offsets: 24 + 4 * 8 + 40 bytes), filled in at the end

    def alloc(self, data):
        while len(self.buf) % 8:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    @staticmethod
    def pad8(b):
        return b + b"\x00" * (-len(b) % 8)

    def datatype(self, value):
        """-> (message body, element bytes of the value in file order)"""
        if isinstance(value, (list, tuple)) and all(isinstance(v, (str, bytes)) for v in value) or \
                isinstance(value, (str, bytes)):
            items = [value] if isinstance(value, (str, bytes)) else list(value)
            raw = [v.encode("utf8") if isinstance(v, str) else bytes(v) for v in items]
            size = max([len(r) for r in raw] + [1])
            body = struct.pack("<BBBBI", 0x13, 0x01, 0, 0, size)  # class 3 (string) version 1, null-padded, ASCII
            return body, b"".join(r + b"\x00" * (size - len(r)) for r in raw)
        arr = np.ascontiguousarray(value)
        if arr.dtype.kind == "f":
            size = arr.dtype.itemsize
            props = {2: (0, 16, 10, 5, 0, 10, 15), 4: (0, 32, 23, 8, 0, 23, 127), 8: (0, 64, 52, 11, 0, 52, 1023)}[size]
            sign_loc = size * 8 - 1
            body = struct.pack("<BBBBI", 0x11, 0x20, sign_loc, 0, size) + struct.pack("<HHBBBBI", *props)
            return body, arr.astype("<f{}".format(size)).tobytes()
        if arr.dtype.kind in "iu":
            size = arr.dtype.itemsize
            body = struct.pack("<BBBBI", 0x10, 0x08 if arr.dtype.kind == "i" else 0, 0, 0, size) + \
                struct.pack("<HH", 0, size * 8)
            return body, arr.astype("<{}{}".format(arr.dtype.kind, size)).tobytes()
        raise H5Unsupported("cannot write values of dtype {}".format(arr.dtype))

    @staticmethod
    def dataspace(shape):
        if shape is None:
            raise H5Unsupported("null dataspaces")
        body = struct.pack("<BBBB4x", 1, len(shape), 0, 0)
        return body + b"".join(struct.pack("<Q", int(n)) for n in shape)

    @staticmethod
    def shape_of(value):
        if isinstance(value, (str, bytes)):
            return ()
        if isinstance(value, (list, tuple)):
            return (len(value),)
        return tuple(np.asarray(value).shape)

    def message(self, mtype, body, flags=0):
        body = self.pad8(body)
        return struct.pack("<HHB3x", mtype, len(body), flags) + body

    def attribute_message(self, name, value):
        dt, raw = self.datatype(value)
        ds = self.dataspace(self.shape_of(value))
        nm = name.encode("utf8") + b"\x00"
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + self.pad8(nm) + self.pad8(dt) + self.pad8(ds) + raw
        return self.message(0x000C, body)

    def object_header(self, messages):
        body = b"".join(messages)
        header = struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body))
        return self.alloc(header + body)

    def dataset(self, ds):
        dt, raw = self.datatype(ds.value)
        shape = self.shape_of(ds.value)
        addr = self.alloc(raw) if raw else UNDEF
        msgs = [self.message(0x0001, self.dataspace(shape)), self.message(0x0003, dt, flags=1),
                self.message(0x0008, struct.pack("<BBQQ", 3, 1, addr, len(raw)))]
        msgs += [self.attribute_message(k, v) for k, v in ds.attrs.items()]
        return self.object_header(msgs)

    def group(self, g):
        """-> (object header address, B-tree address, heap address)"""
        children = [(name, (self.group(obj)[0] if isinstance(obj, Group) else self.dataset(obj)))
                    for name, obj in g.items()]
        children.sort(key=lambda e: e[0].encode("utf8"))  # links are kept in name order
        if len(children) > 2 * self.LEAF_K:
            raise H5Unsupported("more than {} links in one group".format(2 * self.LEAF_K))
        # local heap: offset 0 holds the empty string (the B-tree's first key), then the link names
        heap_data = bytearray(8)
        offsets = []
        for name, _ in children:
            offsets.append(len(heap_data))
            heap_data += self.pad8(name.encode("utf8") + b"\x00")
        free = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)  # one free block: next = 1 (none), size 16
        seg = self.alloc(bytes(heap_data))
        heap = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free, seg))
        snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(children))
        for (name, addr), off in zip(children, offsets):
            snod += struct.pack("<QQII16x", off, addr, 0, 0)
        snod += b"\x00" * (40 * (2 * self.LEAF_K - len(children)))
        snod_addr = self.alloc(snod)
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if children else 0, UNDEF, UNDEF)
        tree += struct.pack("<QQQ", 0, snod_addr, offsets[-1] if offsets else 0)
        tree += b"\x00" * (16 * (2 * 16 - 1))  # room for the 2K = 32 entries of an internal node (K = 16)
        tree_addr = self.alloc(tree)
        msgs = [self.message(0x0011, struct.pack("<QQ", tree_addr, heap))]
        msgs += [self.attribute_message(k, v) for k, v in g.attrs.items()]
        return self.object_header(msgs), tree_addr, heap

    def finish(self, root):
        header, tree, heap = self.group(root)
        while len(self.buf) % 8:
            self.buf.append(0)
        eof = len(self.buf)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, header, 1, 0) + struct.pack("<QQ", tree, heap)  # root entry, cached symbol table
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write(path, root):
    """Writes the tree (Group / Dataset, values: numpy float / integer arrays, str, bytes, lists of str) as HDF5."""
    data = _Writer().finish(root)
    with open(str(path), "wb") as f:
        f.write(data)


# ---------------------------------------------------------------------------------------------------------------- Keras layer files
def _strings(value):
    if value is None:
        return []
    if isinstance(value, (str, bytes)):
        value = [value]
    return [v.decode("utf8") if isinstance(v, bytes) else str(v) for v in value]


def read_keras_weights(path):
    """[(layer name, {weight name: array})] of a Keras weight file (`save_weights`) or full-model file (`model.save`:
    the tree sits under `model_weights`), in `layer_names` order, weight-less layers (Dropout, Lambda) left out.
    Weight names are Keras' (`<layer>/kernel:0`, `<layer>/bias:0`; Keras-1 files: `<layer>_W`, `<layer>_b`)."""
    root = read(path)
    if "model_weights" in root:
        root = root["model_weights"]
    names = _strings(root.attrs.get("layer_names")) or list(root)
    out = []
    for name in names:
        group = root[name]
        weights = OrderedDict()
        for weight_name in _strings(group.attrs.get("weight_names")):
            node = group
            for part in weight_name.split("/"):
                node = node[part]
            weights[weight_name] = np.asarray(node.value)
        if weights:
            out.append((name, weights))
    return out


def write_keras_weights(path, layers, backend="tensorflow", keras_version="2.0.2"):
    """layers: [(layer name, [(weight name, array)])] -> the file Keras 2.0's `save_weights` writes (net.py:572)."""
    root = Group()
    root.attrs["layer_names"] = [name for name, _ in layers]
    root.attrs["backend"] = backend
    root.attrs["keras_version"] = keras_version
    for name, weights in layers:
        group = Group()
        group.attrs["weight_names"] = [weight_name for weight_name, _ in weights]
        for weight_name, value in weights:
            node = group
            parts = weight_name.split("/")
            for part in parts[:-1]:
                node = node.setdefault(part, Group())
            node[parts[-1]] = Dataset(np.asarray(value))
        root[name] = group
    write(path, root)
