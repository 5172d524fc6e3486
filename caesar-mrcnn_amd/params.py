"""Weight inventory of the Mask R-CNN graph and its flat device layout.

Layer names and kernel shapes are the reference's (``name=`` arguments in mrcnn/model.py:99-210,
933-952, 1013-1090, 2005-2019): they are the by-name contract of Keras ``.h5`` files.
All trainable tensors live in ONE flat fp32 buffer (forward order; BatchNorm gammas and betas in two
contiguous blocks at the end) so that the regulariser, the global-norm clip, the SGD update and the
data-parallel all-reduce are single passes over contiguous memory.  Gradients and momentum mirror
that layout.  BatchNorm moving statistics (never trained: TRAIN_BN=False) sit in two more blocks that
are channel-aligned with the gamma/beta blocks, so one ``bn_fold`` launch serves every layer.
"""
import re
from collections import OrderedDict

import numpy as np

ALIGN = 64  # floats; keeps every tensor 256-byte aligned for 16-byte vector loads


class LayerSpec(object):
    __slots__ = ("name", "kind", "shape", "bn", "stride", "padding", "group")

    def __init__(self, name, kind, shape, bn=None, stride=1, padding="same", group=None):
        self.name, self.kind, self.shape, self.bn = name, kind, tuple(shape), bn
        self.stride, self.padding, self.group = stride, padding, group


def backbone_layers(arch):
    """Conv layers of resnet_graph / custom_backbone (model.py:175-244) in forward order."""
    if arch == "custom":
        stem, stages = 16, [(2, [16, 16, 64], "abc"), (3, [32, 32, 128], "abcd"), (4, [64, 64, 256], "ab"),
                            (5, [128, 128, 512], "abc")]
    else:
        n4 = {"resnet50": 5, "resnet101": 22}[arch]
        stem, stages = 64, [(2, [64, 64, 256], "abc"), (3, [128, 128, 512], "abcd"),
                            (4, [256, 256, 1024], "a" + "".join(chr(98 + i) for i in range(n4))),
                            (5, [512, 512, 2048], "abc")]
    L = [LayerSpec("conv1", "conv", (7, 7, 3, stem), bn="bn_conv1", stride=2, padding=(3, 3))]
    cin = stem
    for stage, (f1, f2, f3), blocks in stages:
        for bi, b in enumerate(blocks):
            s = 2 if (bi == 0 and stage > 2) else 1
            base, bnb = "res%d%s_branch" % (stage, b), "bn%d%s_branch" % (stage, b)
            L.append(LayerSpec(base + "2a", "conv", (1, 1, cin, f1), bn=bnb + "2a", stride=s, padding="valid"))
            L.append(LayerSpec(base + "2b", "conv", (3, 3, f1, f2), bn=bnb + "2b"))
            L.append(LayerSpec(base + "2c", "conv", (1, 1, f2, f3), bn=bnb + "2c", padding="valid"))
            if bi == 0:
                L.append(LayerSpec(base + "1", "conv", (1, 1, cin, f3), bn=bnb + "1", stride=s, padding="valid"))
            cin = f3
    return L, [st[1][2] for st in stages]


def model_layers(config):
    arch = config.BACKBONE
    if callable(arch):
        raise NotImplementedError("callable BACKBONE is a TF-graph hook and has no meaning here")
    L, (c2, c3, c4, c5) = backbone_layers(arch)
    P = config.TOP_DOWN_PYRAMID_SIZE
    C = config.NUM_CLASSES
    na = len(config.RPN_ANCHOR_RATIOS)
    fc = config.FPN_CLASSIF_FC_LAYERS_SIZE
    ps = config.POOL_SIZE
    for name, cin in (("fpn_c5p5", c5), ("fpn_c4p4", c4), ("fpn_c3p3", c3), ("fpn_c2p2", c2)):
        L.append(LayerSpec(name, "conv", (1, 1, cin, P), padding="valid"))
    for name in ("fpn_p2", "fpn_p3", "fpn_p4", "fpn_p5"):
        L.append(LayerSpec(name, "conv", (3, 3, P, P)))
    L.append(LayerSpec("rpn_conv_shared", "conv", (3, 3, P, 512), group="rpn_model"))
    L.append(LayerSpec("rpn_class_raw", "conv", (1, 1, 512, 2 * na), padding="valid", group="rpn_model"))
    L.append(LayerSpec("rpn_bbox_pred", "conv", (1, 1, 512, 4 * na), padding="valid", group="rpn_model"))
    L.append(LayerSpec("mrcnn_class_conv1", "conv", (ps, ps, P, fc), bn="mrcnn_class_bn1", padding="valid"))
    L.append(LayerSpec("mrcnn_class_conv2", "conv", (1, 1, fc, fc), bn="mrcnn_class_bn2", padding="valid"))
    L.append(LayerSpec("mrcnn_class_logits", "dense", (1, 1, fc, C), padding="valid"))
    L.append(LayerSpec("mrcnn_bbox_fc", "dense", (1, 1, fc, 4 * C), padding="valid"))
    for i in range(1, 5):
        L.append(LayerSpec("mrcnn_mask_conv%d" % i, "conv", (3, 3, P, P), bn="mrcnn_mask_bn%d" % i))
    # internal GEMM layout [Cin, (a, b, co)]; Keras stores (2, 2, Cout, Cin)
    L.append(LayerSpec("mrcnn_mask_deconv", "deconv", (P, 2, 2, P)))
    L.append(LayerSpec("mrcnn_mask", "conv", (1, 1, P, C), padding="valid"))
    return L


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class ParamLayout(object):
    """Offsets of every tensor in the flat buffers (host-side description; no device memory here)."""

    def __init__(self, config):
        self.layers = model_layers(config)
        self.by_name = OrderedDict((l.name, l) for l in self.layers)
        self.segments = []          # (tensor name, offset, numel, shape, is_bn_affine)
        self.offsets = {}
        off = 0
        for l in self.layers:
            for suffix, shape in (("kernel", l.shape), ("bias", (self._cout(l),))):
                n = int(np.prod(shape))
                self.offsets[l.name + "/" + suffix] = (off, n, tuple(shape))
                self.segments.append((l.name + "/" + suffix, off, n, tuple(shape), False))
                off += _align(n)
        # BatchNorm blocks: channel order = layer order; gamma block then beta block
        self.bn_layers = [l for l in self.layers if l.bn]
        self.bn_channel_offset = {}
        ch = 0
        for l in self.bn_layers:
            self.bn_channel_offset[l.bn] = ch
            ch += _align(self._cout(l))
        self.bn_channels = ch
        self.gamma_offset = off
        self.beta_offset = off + ch
        for l in self.bn_layers:
            c0, n = self.bn_channel_offset[l.bn], self._cout(l)
            self.offsets[l.bn + "/gamma"] = (self.gamma_offset + c0, n, (n,))
            self.offsets[l.bn + "/beta"] = (self.beta_offset + c0, n, (n,))
        for l in self.bn_layers:
            c0, n = self.bn_channel_offset[l.bn], self._cout(l)
            self.segments.append((l.bn + "/gamma", self.gamma_offset + c0, n, (n,), True))
        for l in self.bn_layers:
            c0, n = self.bn_channel_offset[l.bn], self._cout(l)
            self.segments.append((l.bn + "/beta", self.beta_offset + c0, n, (n,), True))
        self.total = off + 2 * ch

    @staticmethod
    def _cout(l):
        return l.shape[3]

    def layer_of(self, tensor_name):
        base = tensor_name.split("/")[0]
        for l in self.layers:
            if l.name == base or l.bn == base:
                return l
        raise KeyError(tensor_name)

    def keras_layer_name(self, tensor_name):
        return tensor_name.split("/")[0]

    def trainable_mask(self, layer_regex):
        """Per-segment 0/1 mask for a layer-name regular expression (model.py:2320-2355, 2432-2443)."""
        presets = {
            "heads": r"(mrcnn\_.*)|(rpn\_.*)|(fpn\_.*)",
            "3+": r"(res3.*)|(bn3.*)|(res4.*)|(bn4.*)|(res5.*)|(bn5.*)|(mrcnn\_.*)|(rpn\_.*)|(fpn\_.*)",
            "4+": r"(res4.*)|(bn4.*)|(res5.*)|(bn5.*)|(mrcnn\_.*)|(rpn\_.*)|(fpn\_.*)",
            "5+": r"(res5.*)|(bn5.*)|(mrcnn\_.*)|(rpn\_.*)|(fpn\_.*)",
            "all": ".*",
        }
        rx = presets.get(layer_regex, layer_regex)
        return np.array([1 if re.fullmatch(rx, s[0].split("/")[0]) else 0 for s in self.segments], dtype=np.uint8)

    def l2_coefficients(self, weight_decay, mask=None):
        """d/dw of keras.regularizers.l2(wd)(w)/size(w) = 2*wd*w/size(w); gamma/beta excluded
        (model.py:2287-2291), frozen tensors excluded (only trainable_weights are regularised)."""
        out = np.zeros(len(self.segments), dtype=np.float32)
        for i, (name, _, n, _, is_bn) in enumerate(self.segments):
            if not is_bn and (mask is None or mask[i]):
                out[i] = np.float32(2.0 * weight_decay / n)
        return out


def granule_coefficients(layout, weight_decay, mask):
    """Per-64-float-granule table for the optimiser kernels: L2 gradient coefficient of the tensor the
    granule belongs to if it is trainable, -1 for frozen tensors and alignment padding."""
    g = np.full(layout.total // ALIGN, -1.0, dtype=np.float32)
    l2 = layout.l2_coefficients(weight_decay, mask)
    for i, (name, off, n, _, _) in enumerate(layout.segments):
        if mask[i]:
            g[off // ALIGN:(off + n + ALIGN - 1) // ALIGN] = l2[i]
    return g


def glorot_uniform(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def init_weights(layout, seed=0, perturb_bn=False):
    """Keras default initialisation (glorot_uniform kernels, zero biases, BN gamma=1 beta=0 mean=0
    var=1) as {tensor name: ndarray}.  ``perturb_bn`` randomises the BatchNorm tensors and biases so
    tests exercise every term of the epilogue."""
    rng = np.random.RandomState(seed)
    w = OrderedDict()
    for l in layout.layers:
        if l.kind == "deconv":
            cin, a, b, co = l.shape
            w[l.name + "/kernel"] = glorot_uniform(rng, l.shape, cin * a * b, co * a * b)
            nb = co
        else:
            kh, kw, cin, co = l.shape
            w[l.name + "/kernel"] = glorot_uniform(rng, l.shape, kh * kw * cin, kh * kw * co)
            nb = co
        w[l.name + "/bias"] = (rng.uniform(-0.1, 0.1, nb).astype(np.float32) if perturb_bn
                               else np.zeros(nb, np.float32))
        if l.bn:
            if perturb_bn:
                w[l.bn + "/gamma"] = rng.uniform(0.5, 1.5, nb).astype(np.float32)
                w[l.bn + "/beta"] = rng.uniform(-0.2, 0.2, nb).astype(np.float32)
                w[l.bn + "/moving_mean"] = rng.uniform(-0.2, 0.2, nb).astype(np.float32)
                w[l.bn + "/moving_variance"] = rng.uniform(0.5, 1.5, nb).astype(np.float32)
            else:
                w[l.bn + "/gamma"] = np.ones(nb, np.float32)
                w[l.bn + "/beta"] = np.zeros(nb, np.float32)
                w[l.bn + "/moving_mean"] = np.zeros(nb, np.float32)
                w[l.bn + "/moving_variance"] = np.ones(nb, np.float32)
    return w


def deconv_keras_to_gemm(k):
    """Keras Conv2DTranspose kernel (2, 2, Cout, Cin) -> GEMM layout (Cin, 2, 2, Cout)."""
    return np.ascontiguousarray(np.transpose(k, (3, 0, 1, 2)))


def deconv_gemm_to_keras(g):
    return np.ascontiguousarray(np.transpose(g, (1, 2, 3, 0)))
