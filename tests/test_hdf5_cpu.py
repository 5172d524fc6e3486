"""hdf5_min's READER against byte images it did not write (scope rows a16 / f2; mrcnn/model.py:2197-2239).

h5py / libhdf5 are absent and the reference ships only an LFS pointer for its weight file, so no Keras-written file
exists to read.  What can be done is to stop testing the reader against its own writer only: `_Asm` below assembles
files independently, straight from the HDF5 File Format Specification (version 0/1 superblock, version-1 object
headers, v1 B-trees, local heaps), making the choices libhdf5 1.8/1.10 + h5py 2.x make and hdf5_min's writer does
NOT: a full-model file with the ``model_weights`` wrapper group, attributes appended in object-header CONTINUATION
blocks (h5py sets attrs after creating the group), NIL and modification-time messages in between, version-3
attribute messages and version-2 dataspaces, a variable-length string attribute (``keras_version``), group B-trees
with two levels and many small SNODs, a superblock of version 1, a float64 and a compact dataset, the nested
``rpn_model`` group of the Keras layout.  Still no substitute for a file written by Keras: a16 / f2 stay "parity
unpinned" in DESIGN.md."""
import os
import struct

import numpy as np
import pytest

import caesar_mrcnn_amd  # noqa: F401
from caesar_mrcnn_amd import hdf5_min

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


class _Asm(object):
    """Spec-level HDF5 assembler (old-style groups only).  Options choose among equivalent encodings."""

    def __init__(self, sb_version=1, leaf_k=2, internal_k=2, attr_version=3, space_version=2, continuation=True):
        self.sbv, self.leaf_k, self.internal_k = sb_version, leaf_k, internal_k
        self.attr_version, self.space_version, self.continuation = attr_version, space_version, continuation
        self.sb_size = 96 + (4 if sb_version == 1 else 0)
        self.buf = bytearray(self.sb_size)
        self.buf += b"\xAA" * 40                     # junk the reader must never look at (a real file has free space too)

    def alloc(self, data):
        self.buf += b"\0" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    # ---- messages ---------------------------------------------------------------------------------
    @staticmethod
    def msg(mtype, body, flags=0):
        body = _pad8(body)
        return struct.pack("<HHB3x", mtype, len(body), flags) + body

    def space(self, shape):
        if self.space_version == 2:
            return struct.pack("<BBBB", 2, len(shape), 0, 1 if shape else 0) + b"".join(struct.pack("<Q", d) for d in shape)
        return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)

    @staticmethod
    def dt_float(size, big=False):
        if size == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        else:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        return struct.pack("<BBBBI", 0x11, 0x20 | (1 if big else 0), 0x3F if size == 8 else 0x1F, 0x00, size) + props

    @staticmethod
    def dt_str(n):
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, n)             # fixed length, null padded, ASCII

    @staticmethod
    def dt_vlen_str():
        base = struct.pack("<BBBBI", 0x13, 0x00, 0, 0, 1)
        return struct.pack("<BBBBI", 0x19, 0x01, 0, 0, 16) + base      # class 9, type = string

    def attr(self, name, dt, shape, data):
        nm = name.encode() + b"\0"
        sp = self.space(shape)
        v = self.attr_version
        if v == 1:
            body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + data
        elif v == 2:
            body = struct.pack("<BBHHH", 2, 0, len(nm), len(dt), len(sp)) + nm + dt + sp + data
        else:
            body = struct.pack("<BBHHHB", 3, 0, len(nm), len(dt), len(sp), 0) + nm + dt + sp + data
        return self.msg(0x000C, body)

    def str_attr(self, name, strings):
        enc = [s.encode() for s in strings]
        n = max(len(e) for e in enc)
        return self.attr(name, self.dt_str(n), (len(enc),), b"".join(e.ljust(n, b"\0") for e in enc))

    def vlen_attr(self, name):
        return self.attr(name, self.dt_vlen_str(), (), struct.pack("<IQI", 5, 0x1234, 1))     # a global-heap reference

    # ---- objects ----------------------------------------------------------------------------------
    def header(self, first, later=()):
        """Object header whose `later` messages (attributes set after creation) sit in a continuation block."""
        nil = self.msg(0x0000, b"\0" * 8)
        mtime = self.msg(0x0012, struct.pack("<B3xI", 1, 1700000000))
        first = list(first) + [mtime]
        later = list(later)
        if later and self.continuation:
            cont = self.alloc(b"".join([nil] + later))
            clen = len(b"".join([nil] + later))
            first = first + [self.msg(0x0010, struct.pack("<QQ", cont, clen))]
            n = len(first) + 1 + len(later)
        else:
            first = first + later
            n = len(first)
        body = b"".join(first)
        return self.alloc(struct.pack("<BxHII4x", 1, n, 1, len(body)) + body)

    def dataset(self, arr, compact=False, big=False):
        dt = self.dt_float(arr.dtype.itemsize, big)
        raw = arr.astype((">" if big else "<") + "f%d" % arr.dtype.itemsize).tobytes()
        if compact:
            layout = struct.pack("<BBH", 3, 0, len(raw)) + raw
        else:
            layout = struct.pack("<BBQQ", 3, 1, self.alloc(raw + b"\xEE" * 8), len(raw))
        msgs = [self.msg(0x0001, self.space(arr.shape)), self.msg(0x0003, dt, flags=1),
                self.msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)), self.msg(0x0008, layout)]
        return self.header(msgs)

    def group(self, links, attrs=()):
        names = sorted(links, key=lambda s: s.encode())
        heap = bytearray(b"\0" * 8)
        off = {}
        for n in names:
            off[n] = len(heap)
            heap += _pad8(n.encode() + b"\0")
        free = len(heap)
        heap += struct.pack("<QQ", 1, 24) + b"\0" * 8
        data_addr = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free, data_addr))
        per = 2 * self.leaf_k
        level = []                                  # (address, heap offset of the largest name below)
        for i in range(0, max(len(names), 1), per):
            part = names[i:i + per]
            ents = b"".join(struct.pack("<QQII16x", off[n], links[n], 0, 0) for n in part)
            node = (b"SNOD" + struct.pack("<BxH", 1, len(part)) + ents).ljust(8 + per * 40, b"\0")
            level.append((self.alloc(node), off[part[-1]] if part else 0))
        lvl = 0
        while True:
            nxt = []
            fan = 2 * self.internal_k
            for i in range(0, len(level), fan):
                part = level[i:i + fan]
                body = struct.pack("<Q", 0)
                for addr, last in part:
                    body += struct.pack("<QQ", addr, last)
                node = b"TREE" + struct.pack("<BBHQQ", 0, lvl, len(part), UNDEF, UNDEF) + body
                node = node.ljust(24 + (2 * fan + 1) * 8, b"\0")
                nxt.append((self.alloc(node), part[-1][1]))
            level, lvl = nxt, lvl + 1
            if len(level) == 1:
                break
        btree = level[0][0]
        return self.header([self.msg(0x0011, struct.pack("<QQ", btree, heap_addr))], attrs), btree, heap_addr

    def finish(self, path, root):
        hdr, btree, heap = root
        sb = hdf5_min.SIG + struct.pack("<BBBxBBBxHHI", self.sbv, 0, 0, 0, 8, 8, self.leaf_k, self.internal_k, 0)
        if self.sbv == 1:
            sb += struct.pack("<HH", 32, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == self.sb_size
        self.buf[:len(sb)] = sb
        with open(path, "wb") as fh:
            fh.write(bytes(self.buf))


def _tensors(rng, n_layers=11):
    """Layer set with the Keras naming of the model (conv + BatchNorm, the nested rpn_model, a Dense head)."""
    t, nested = {}, {}
    for i in range(n_layers):
        t["res2%s_branch2a/kernel" % chr(97 + i)] = rng.standard_normal((1, 1, 8, 4)).astype(np.float32)
        t["res2%s_branch2a/bias" % chr(97 + i)] = rng.standard_normal(4).astype(np.float32)
        for k in ("gamma", "beta", "moving_mean", "moving_variance"):
            t["bn2%s_branch2a/%s" % (chr(97 + i), k)] = rng.standard_normal(4).astype(np.float32)
    for name, shape in (("rpn_conv_shared", (3, 3, 4, 8)), ("rpn_class_raw", (1, 1, 8, 6)), ("rpn_bbox_pred", (1, 1, 8, 12))):
        t[name + "/kernel"] = rng.standard_normal(shape).astype(np.float32)
        t[name + "/bias"] = rng.standard_normal(shape[-1]).astype(np.float32)
        nested[name] = "rpn_model"
    t["mrcnn_class_logits/kernel"] = rng.standard_normal((16, 4)).astype(np.float32)
    t["mrcnn_class_logits/bias"] = rng.standard_normal(4).astype(np.float32)
    return t, nested


def _assemble(path, tensors, nested, wrapper, weight_names_first, **opts):
    a = _Asm(**opts)
    top = {}
    for key in tensors:
        layer = key.split("/")[0]
        top.setdefault(nested.get(layer, layer), {}).setdefault(layer, []).append(key.split("/")[1])
    top_links, layer_names = {}, []
    for k, (tname, layers) in enumerate(sorted(top.items(), key=lambda kv: kv[0][::-1])):     # creation order != sorted order
        inner, wnames = {}, []
        for layer, ws in layers.items():
            ds = {}
            for j, wn in enumerate(ws):
                arr = tensors[layer + "/" + wn]
                if wn == "moving_mean":
                    arr = arr.astype(np.float64)                     # a float64 dataset: must come back as numbers, not bytes
                ds[wn + ":0"] = a.dataset(arr, compact=(wn == "bias" and k % 2 == 0), big=(wn == "beta"))
                wnames.append(layer + "/" + wn + ":0")
            inner[layer] = a.group(ds)[0]
        attrs = [a.str_attr("weight_names", wnames)]
        extra = [a.vlen_attr("some_vlen_note")]
        top_links[tname] = a.group(inner, attrs + extra if weight_names_first else extra + attrs)[0]
        layer_names.append(tname)
    wattrs = [a.vlen_attr("keras_version"), a.vlen_attr("backend"), a.str_attr("layer_names", layer_names)]
    weights_group = a.group(top_links, wattrs)
    if wrapper:                                      # model.save(): root holds model_config etc., weights one level down
        opt = a.group({"iterations:0": a.dataset(np.zeros((), np.float32))})
        root = a.group({"model_weights": weights_group[0], "optimizer_weights": opt[0]},
                       [a.vlen_attr("model_config"), a.vlen_attr("training_config")])
    else:
        root = weights_group
    a.finish(path, root)


VARIANTS = [
    dict(wrapper=False, weight_names_first=True, sb_version=0, attr_version=1, space_version=1, continuation=False, leaf_k=4,
         internal_k=16),
    dict(wrapper=False, weight_names_first=False, sb_version=0, attr_version=1, space_version=1, continuation=True, leaf_k=2,
         internal_k=2),
    dict(wrapper=True, weight_names_first=False, sb_version=1, attr_version=3, space_version=2, continuation=True, leaf_k=1,
         internal_k=1),
    dict(wrapper=True, weight_names_first=True, sb_version=1, attr_version=2, space_version=2, continuation=True, leaf_k=2,
         internal_k=3),
]


@pytest.mark.parametrize("variant", range(len(VARIANTS)))
def test_reader_on_independently_assembled_files(tmp_path, variant):
    rng = np.random.default_rng(100 + variant)
    tensors, nested = _tensors(rng)
    path = os.path.join(str(tmp_path), "w%d.h5" % variant)
    _assemble(path, tensors, nested, **VARIANTS[variant])
    got = hdf5_min.load_keras_weights(path)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].shape == v.shape and got[k].dtype.kind == "f" and got[k].dtype.isnative, k
        np.testing.assert_array_equal(got[k].astype(np.float32), v, err_msg=k)
    f = hdf5_min.H5File(path)
    root = f.get(f.root, "model_weights") if VARIANTS[variant]["wrapper"] else f.root
    assert "rpn_model" in root.children and "rpn_conv_shared" in f.get(root, "rpn_model").children
    assert root.attrs["keras_version"] is None            # variable-length string: recognised, skipped, nothing misparsed
    assert len(root.attrs["layer_names"]) == len(root.children)


def test_writer_output_is_read_back_by_the_independent_walk(tmp_path):
    """The other direction: what hdf5_min WRITES, walked here by hand from the specification's offsets (no use of the
    reader's group / B-tree code): superblock -> root header -> symbol-table message -> TREE -> SNOD -> heap names."""
    rng = np.random.default_rng(7)
    tensors, _ = _tensors(rng, n_layers=3)
    path = os.path.join(str(tmp_path), "out.h5")
    hdf5_min.save_keras_weights(path, tensors)
    b = open(path, "rb").read()
    assert b[:8] == hdf5_min.SIG and b[8] == 0 and b[13] == 8 and b[14] == 8
    eof = struct.unpack_from("<Q", b, 24 + 16)[0]
    assert eof == len(b)
    root_hdr = struct.unpack_from("<Q", b, 24 + 32 + 8)[0]
    ver, nmsg, _, hsize = struct.unpack_from("<BxHII", b, root_hdr)
    assert ver == 1
    p, names = root_hdr + 16, None
    for _ in range(nmsg):
        mtype, msize = struct.unpack_from("<HH", b, p)
        if mtype == 0x0011:
            tree, heap = struct.unpack_from("<QQ", b, p + 8)
            assert b[tree:tree + 4] == b"TREE" and b[heap:heap + 4] == b"HEAP"
            data = struct.unpack_from("<Q", b, heap + 24)[0]
            snod = struct.unpack_from("<Q", b, tree + 24 + 8)[0]
            assert b[snod:snod + 4] == b"SNOD"
            n = struct.unpack_from("<H", b, snod + 6)[0]
            names = []
            for i in range(n):
                off = struct.unpack_from("<Q", b, snod + 8 + 40 * i)[0]
                names.append(b[data + off:b.index(b"\0", data + off)].decode())
        p += 8 + msize
    layers = sorted({k.split("/")[0] for k in tensors})
    assert names == sorted(layers, key=lambda s: s.encode())           # links sorted by name, as the B-tree requires


def test_reader_refuses_what_it_does_not_implement(tmp_path):
    a = _Asm()
    chunked = a.header([a.msg(0x0001, a.space((4,))), a.msg(0x0003, a.dt_float(4), flags=1),
                        a.msg(0x0008, struct.pack("<BBBQII", 3, 2, 2, 0, 4, 4))])
    a.finish(os.path.join(str(tmp_path), "c.h5"), a.group({"x": chunked}))
    f = hdf5_min.H5File(os.path.join(str(tmp_path), "c.h5"))
    with pytest.raises(NotImplementedError):
        f.get(f.root, "x")
    a = _Asm()
    newstyle = a.header([a.msg(0x0002, struct.pack("<BB", 0, 0))])     # link-info message: new-style group
    a.finish(os.path.join(str(tmp_path), "n.h5"), a.group({"g": newstyle}))
    f = hdf5_min.H5File(os.path.join(str(tmp_path), "n.h5"))
    with pytest.raises(NotImplementedError):
        f.get(f.root, "g")
    open(os.path.join(str(tmp_path), "bad.h5"), "wb").write(b"not hdf5 at all")
    with pytest.raises(ValueError):
        hdf5_min.H5File(os.path.join(str(tmp_path), "bad.h5"))
