"""Dependency-free reader/writer for the HDF5 subset Keras 2.2.4 weight files use (h5py is not on the
target image).  Reference call sites: ``MaskRCNN.load_weights`` (mrcnn/model.py:2197-2239, h5py +
keras.engine.saving.load_weights_from_hdf5_group_by_name) and the per-epoch
``ModelCheckpoint(save_weights_only=True)`` (model.py:2461-2462).

File layout handled (what h5py 2.x / libhdf5 1.8-1.10 write by default, "earliest" format):
  superblock v0/v1, version-1 object headers (with continuation blocks), old-style groups (symbol-table
  message -> v1 B-tree -> SNOD nodes + local heap), contiguous datasets of IEEE floats / integers,
  version-1..3 attribute messages holding numeric or fixed-length string arrays.
Not handled (clear error): new-style groups (link messages / fractal heaps), chunked or filtered
datasets, variable-length data.  [3P: written from the public HDF5 File Format Specification v1.1/2.0;
no h5py here to cross-check -- parity unpinned; write/read round trips are tested.]

Keras layout: root (or group ``model_weights``) has attr ``layer_names``; each layer
group has attr ``weight_names`` (e.g. b"conv1/kernel:0") and the datasets live at
``<layer>/<weight_name>``; the RPN layers sit inside the nested model group ``rpn_model``.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"


# =====================================================================================================
#  reader
# =====================================================================================================
class H5Object(object):
    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.attrs = {}
        self.children = None        # name -> address (groups)
        self.dataset = None         # (dtype, shape, data address, size) for datasets


class H5File(object):
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        if b[:8] != SIG:
            raise ValueError("%s is not an HDF5 file" % path)
        ver = b[8]
        if ver not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d (only the 'earliest' format 0/1 that h5py 2.x "
                                      "writes is supported)" % ver)
        if b[13] != 8 or b[14] != 8:
            raise NotImplementedError("only 8-byte offsets/lengths are supported")
        p = 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", b, p)[0]
        p += 32                                          # base, free-space, eof, driver
        _, root_hdr, cache, _, btree, heap = struct.unpack_from("<QQIIQQ", b, p)
        self.root = self.obj(root_hdr)

    # ---- low level ------------------------------------------------------------------------------
    def _messages(self, addr):
        b = self.buf
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise NotImplementedError("object header version %d (new-style file); re-save with libver='earliest'" % ver)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", b, p)
                body = p + 8
                if mtype == 0x0010:                                   # continuation
                    off, ln = struct.unpack_from("<QQ", b, body)
                    blocks.append((off + self.base, ln))
                out.append((mtype, body, msize))
                p = body + msize
        return out

    def _heap_name(self, heap_addr, off):
        b = self.buf
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise ValueError("bad local heap")
        data_addr = struct.unpack_from("<Q", b, heap_addr + 24)[0] + self.base
        s = data_addr + off
        e = b.index(b"\0", s)
        return b[s:e].decode("utf8")

    def _walk_btree(self, addr, heap, out):
        b = self.buf
        if b[addr:addr + 4] == b"TREE":
            ntype, level, used = struct.unpack_from("<BBH", b, addr + 4)
            p = addr + 24
            for i in range(used):
                child = struct.unpack_from("<Q", b, p + 8)[0]
                self._walk_btree(child + self.base, heap, out)
                p += 16
        elif b[addr:addr + 4] == b"SNOD":
            n = struct.unpack_from("<H", b, addr + 6)[0]
            p = addr + 8
            for i in range(n):
                name_off, hdr = struct.unpack_from("<QQ", b, p)
                out[self._heap_name(heap, name_off)] = hdr + self.base
                p += 40
        else:
            raise ValueError("bad group node at %d" % addr)

    @staticmethod
    def _dtype(b, p):
        cv = b[p]
        cls, ver = cv & 0x0F, cv >> 4
        bits = b[p + 1] | (b[p + 2] << 8) | (b[p + 3] << 16)
        size = struct.unpack_from("<I", b, p + 4)[0]
        order = ">" if (bits & 1) else "<"
        if cls == 0:
            return np.dtype("%s%s%d" % (order, "i" if (bits & 8) else "u", size))
        if cls == 1:
            return np.dtype("%sf%d" % (order, size))
        if cls == 3:
            return np.dtype("S%d" % size)
        if cls == 9:
            return None                                   # variable length: not needed for weights
        raise NotImplementedError("HDF5 datatype class %d" % cls)

    @staticmethod
    def _space(b, p):
        ver, rank, flags = b[p], b[p + 1], b[p + 2]
        q = p + (8 if ver == 1 else 4)
        return tuple(struct.unpack_from("<%dQ" % rank, b, q)) if rank else ()

    def _attr(self, body):
        b = self.buf
        ver = b[body]
        nsz, tsz, ssz = struct.unpack_from("<HHH", b, body + 2)
        p = body + 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = b[p:p + nsz].split(b"\0")[0].decode("utf8")
        p += pad(nsz)
        dt = self._dtype(b, p)
        tp = p
        p += pad(tsz)
        shape = self._space(b, p)
        p += pad(ssz)
        if dt is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        val = np.frombuffer(b, dtype=dt, count=n, offset=p).reshape(shape)
        return name, val.copy()

    def obj(self, addr):
        o = H5Object(self, addr)
        b = self.buf
        dt = shape = None
        for mtype, body, msize in self._messages(addr):
            if mtype == 0x0011:
                btree, heap = struct.unpack_from("<QQ", b, body)
                o.children = {}
                self._walk_btree(btree + self.base, heap + self.base, o.children)
            elif mtype in (0x0002, 0x0006):
                raise NotImplementedError("new-style HDF5 group (link messages); re-save with libver='earliest'")
            elif mtype == 0x0001:
                shape = self._space(b, body)
            elif mtype == 0x0003:
                dt = self._dtype(b, body)
            elif mtype == 0x0008:
                ver = b[body]
                if ver == 3:
                    cls = b[body + 1]
                    if cls == 1:
                        daddr, dsize = struct.unpack_from("<QQ", b, body + 2)
                        o.dataset = [None, None, daddr + self.base if daddr != UNDEF else None, dsize]
                    elif cls == 0:
                        dsize = struct.unpack_from("<H", b, body + 2)[0]
                        o.dataset = [None, None, body + 4, dsize]
                    else:
                        raise NotImplementedError("chunked HDF5 datasets are not supported (Keras weights are contiguous)")
                else:
                    rank, cls = b[body + 1], b[body + 2]
                    if cls != 1:
                        raise NotImplementedError("only contiguous datasets are supported")
                    daddr = struct.unpack_from("<Q", b, body + 8)[0]
                    o.dataset = [None, None, daddr + self.base, None]
            elif mtype == 0x000C:
                k, v = self._attr(body)
                o.attrs[k] = v
        if o.dataset is not None:
            o.dataset[0], o.dataset[1] = dt, shape
        return o

    def get(self, obj, name):
        for part in name.split("/"):
            if obj.children is None or part not in obj.children:
                raise KeyError(name)
            obj = self.obj(obj.children[part])
        return obj

    def read(self, obj):
        dt, shape, addr, size = obj.dataset
        n = int(np.prod(shape)) if shape else 1
        if addr is None:
            return np.zeros(shape, dtype=dt.newbyteorder("="))
        return np.frombuffer(self.buf, dtype=dt, count=n, offset=addr).reshape(shape).astype(dt.newbyteorder("="))


def load_keras_weights(path):
    """{"<layer>/<kernel|bias|gamma|beta|moving_mean|moving_variance>": ndarray} from a Keras weight file."""
    f = H5File(path)
    root = f.root
    if "layer_names" not in root.attrs and root.children and "model_weights" in root.children:
        root = f.get(root, "model_weights")                       # full-model file (model.py:2218-2219)
    out = {}

    def visit(group):
        names = group.attrs.get("weight_names")
        if names is None:
            return
        for wn in names.reshape(-1):
            wn = wn.decode("utf8") if isinstance(wn, bytes) else str(wn)
            arr = f.read(f.get(group, wn))
            parts = wn.split("/")
            key = parts[-2] + "/" + parts[-1].split(":")[0]
            out[key] = arr

    layer_names = root.attrs.get("layer_names")
    layers = [n.decode("utf8") for n in layer_names.reshape(-1)] if layer_names is not None else list(root.children)
    for ln in layers:
        if root.children is None or ln not in root.children:
            continue
        visit(f.obj(root.children[ln]))
    return out


# =====================================================================================================
#  writer
# =====================================================================================================
def _pad8(b):
    return b + b"\0" * ((8 - len(b) % 8) % 8)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _dt_f32():
    return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _dt_str(n):
    return struct.pack("<BBBBI", 0x13, 0x01, 0x00, 0x00, n)          # null-padded ASCII, as h5py stores numpy 'S'


def _space(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)


def _attr_msg(name, dt, shape, data):
    nm = name.encode("utf8") + b"\0"
    sp = _space(shape)
    body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + data
    return _msg(0x000C, body)


def _str_array_attr(name, strings):
    enc = [s.encode("utf8") for s in strings]
    n = max([len(e) for e in enc] + [1])
    data = b"".join(e.ljust(n, b"\0") for e in enc)
    return _attr_msg(name, _dt_str(n), (len(enc),), data)


class _Writer(object):
    LEAF_K, INTERNAL_K = 32, 16           # 64 symbols per SNOD, 32 SNODs per B-tree node: 2048 links per group

    def __init__(self):
        self.chunks = []
        self.pos = 96                      # after the superblock

    def alloc(self, data):
        data = _pad8(data)
        addr = self.pos
        self.chunks.append((addr, data))
        self.pos += len(data)
        return addr

    def object_header(self, messages):
        body = b"".join(messages)
        return self.alloc(struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body)

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr, dtype="<f4")
        daddr = self.alloc(arr.tobytes() if arr.size else b"\0" * 8)
        msgs = [_msg(0x0001, _space(arr.shape)), _msg(0x0003, _dt_f32(), flags=1),
                _msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),
                _msg(0x0008, struct.pack("<BBQQ", 3, 1, daddr, arr.nbytes))]
        return self.object_header(msgs)

    def group(self, links, attr_msgs=()):
        """links: {name: object header address}.  Returns (header addr, btree addr, heap addr)."""
        names = sorted(links, key=lambda s: s.encode("utf8"))
        heap_data = b"\0" * 8
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _pad8(n.encode("utf8") + b"\0")
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 32) + b"\0" * 16            # one free block (next = H5HL_FREE_NULL)
        data_addr = self.alloc(heap_data)
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, data_addr))
        per = 2 * self.LEAF_K
        snods = []
        for i in range(0, max(len(names), 1), per):
            part = names[i:i + per]
            ents = b"".join(struct.pack("<QQII16x", offs[n], links[n], 0, 0) for n in part)
            node = b"SNOD" + struct.pack("<BxH", 1, len(part)) + ents
            node = node.ljust(8 + per * 40, b"\0")
            snods.append((self.alloc(node), offs[part[-1]] if part else 0))
        if len(snods) > 2 * self.INTERNAL_K:
            raise ValueError("too many links in one HDF5 group for the single-level writer")
        keys = struct.pack("<Q", 0)
        body = b""
        for addr, last in snods:
            body += struct.pack("<Q", addr)
            body += struct.pack("<Q", last)
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF) + keys + body
        tree = tree.ljust(24 + (2 * self.INTERNAL_K + 1) * 8 + 2 * self.INTERNAL_K * 8, b"\0")
        btree_addr = self.alloc(tree)
        hdr = self.object_header([_msg(0x0011, struct.pack("<QQ", btree_addr, heap_addr))] + list(attr_msgs))
        return hdr, btree_addr, heap_addr

    def finish(self, path, root):
        hdr, btree, heap = root
        sb = SIG + struct.pack("<BBBxBBBxHHI", 0, 0, 0, 0, 8, 8, self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, self.pos, UNDEF)
        sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96, len(sb)
        with open(path, "wb") as fh:
            fh.write(sb)
            for addr, data in self.chunks:
                assert fh.tell() == addr
                fh.write(data)


def save_keras_weights(path, tensors, layout=None):
    """Writes {"<layer>/<weight>": array} as a Keras ``save_weights`` file: one group per layer (RPN layers
    nested under ``rpn_model`` when `layout` marks them), attrs ``layer_names`` / ``weight_names``."""
    nested = {}
    if layout is not None:
        nested = {l.name: l.group for l in layout.layers if l.group}
    order = ["kernel", "bias", "gamma", "beta", "moving_mean", "moving_variance"]
    by_layer = {}
    for key in tensors:
        layer, w = key.split("/")
        by_layer.setdefault(layer, []).append(w)
    w = _Writer()
    top = {}            # top-level layer name -> {inner layer: [weights]}
    for layer, ws in by_layer.items():
        top.setdefault(nested.get(layer, layer), {})[layer] = sorted(ws, key=lambda x: order.index(x) if x in order else 99)
    top_links, layer_names = {}, []
    for tname in top:
        inner_links, weight_names = {}, []
        for layer, ws in top[tname].items():
            dsets = {}
            for wn in ws:
                dsets[wn + ":0"] = w.dataset(tensors[layer + "/" + wn])
                weight_names.append(layer + "/" + wn + ":0")
            inner_links[layer] = w.group(dsets)[0]
        top_links[tname] = w.group(inner_links, [_str_array_attr("weight_names", weight_names)])[0]
        layer_names.append(tname)
    attrs = [_str_array_attr("layer_names", layer_names), _str_array_attr("backend", ["tensorflow"]),
             _str_array_attr("keras_version", ["2.2.4"])]
    w.finish(path, w.group(top_links, attrs))
