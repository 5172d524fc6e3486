"""Device-side execution of the Mask R-CNN graph wired by MaskRCNN.build (mrcnn/model.py:1935-2166):
forward (inference and training), the hand-derived backward pass, and the optimiser step
(MaskRCNN.compile, mrcnn/model.py:2255-2291).  Every arithmetic step is a launch into
libmrcnn_hip.so; torch only owns the buffers.

Gradient plumbing conventions
  * all parameter gradients live in one flat buffer (params.ParamLayout) that is zeroed per step;
    bias / BatchNorm gamma,beta gradients are accumulated by the epilogue-backward kernel;
  * a tensor consumed by several ops gets its gradient summed by chaining: every data-gradient
    convolution can add an existing buffer in its epilogue (res_mode SAME, in place);
  * the ROIAlign adjoints scatter-add into zero-initialised pyramid gradients first.
"""
import contextlib
import gc
import os
import numpy as np
import torch

from . import ops
from . import _hip as _hip_mod
from ._hip import ACT_NONE, ACT_RELU, ACT_SIGMOID, RES_NONE, RES_SAME, RES_UP2
from .params import ParamLayout, deconv_gemm_to_keras, deconv_keras_to_gemm, granule_coefficients, init_weights

# configs[4], stage 3: besides the mask head, these layers run on the 16-bit matrix cores when head_dtype is set (and
# their shapes fit the 16-bit kernels): the FPN smoothing convolutions, the shared RPN convolution over all five levels,
# and the two FC layers of the class head (K = 12 544: weight-bandwidth halves) -- 82 of the 161 GFLOP per image of the
# 512 x 512 trunk.  float32 at their boundaries (casts), float32 master weights, accumulation and gradients.
H16_WIDE_LAYERS = ("fpn_p2", "fpn_p3", "fpn_p4", "fpn_p5", "rpn_conv_shared", "mrcnn_class_conv1", "mrcnn_class_conv2")

LOSS_NAMES = ("rpn_class_loss", "rpn_bbox_loss", "mrcnn_class_loss", "mrcnn_bbox_loss", "mrcnn_mask_loss")



@contextlib.contextmanager
def _capture(graph):
    """torch.cuda.graph(graph) with the garbage collector out of the way: this torch no longer collects before a capture
    (torch.compiler.config.force_cudagraph_gc), and a dead cycle that owns a CUDAGraph, a stream or device memory (an earlier
    engine) freed by a collection that happens to run INSIDE the capture makes hipGraphExecDestroy / hipFree fail under the global
    capture mode -- thrown from a destructor, which aborts the process.  Collect before, keep the collector off while capturing."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was:
            gc.enable()

class ConvOp(object):
    """One named Conv2D/Dense(+BatchNorm) layer: views into the flat buffers + fwd/bwd launches."""

    def __init__(self, model, spec):
        L = model.layout
        self.model = model
        self.spec = spec
        self.name = spec.name
        off, n, shape = L.offsets[spec.name + "/kernel"]
        if spec.kind == "deconv":                       # [Cin, (a,b,co)] GEMM == 1x1 conv Cin -> 4*Cd
            self.wshape = (1, 1, shape[0], shape[1] * shape[2] * shape[3])
        else:
            self.wshape = tuple(shape)
        kh, kw, cin, cout = self.wshape
        self.w = model.params[off:off + n].view(self.wshape)
        self.dw = model.grads[off:off + n].view(self.wshape)
        self.wt = model.wt[off:off + n].view(kh, kw, cout, cin)
        boff, bn_, _ = L.offsets[spec.name + "/bias"]
        self.b = model.params[boff:boff + bn_]
        self.db = model.grads[boff:boff + bn_]
        self.stride, self.padding = spec.stride, spec.padding
        self.bn = spec.bn
        self.scale = self.shift = self.mean = self.rstd = self.dgamma = self.dbeta = None
        if spec.bn:
            c0, c = L.bn_channel_offset[spec.bn], spec.shape[3]
            self.scale, self.shift = model.bn_scale[c0:c0 + c], model.bn_shift[c0:c0 + c]
            self.mean, self.rstd = model.bn_mean[c0:c0 + c], model.bn_rstd[c0:c0 + c]
            self.dgamma = model.grads[L.gamma_offset + c0:L.gamma_offset + c0 + c]
            self.dbeta = model.grads[L.beta_offset + c0:L.beta_offset + c0 + c]

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, x, act=ACT_NONE, res=None, res_mode=RES_NONE, train=False):
        z = None
        d = ops.conv_desc(tuple(x.shape), self.wshape, self.stride, self.padding, act, res_mode)
        out = ops.empty((d.N, d.OH, d.OW, d.Cout), torch.float32, x.device)
        if train and self.bn:
            z = ops.empty_like(out)
        ops.conv2d(x, self.w, self.b, self.scale, self.shift, res, out=out, z_out=z, desc=d)
        ctx = (x, z, out, act) if train else None
        return out, ctx

    # ---- backward pieces ---------------------------------------------------------------------
    def epilogue_bwd(self, dout, ctx, want_dy=False):
        """Returns (dz, dy): gradient w.r.t. the raw conv output, and (optionally) the gradient after
        the activation mask, which is what a residual input receives."""
        x, z, out, act = ctx
        if self.bn is None and act == ACT_NONE:
            ops.epilogue_bwd(dout, dbias=self.db)
            return dout, dout
        dz = ops.empty_like(dout)
        dy = ops.empty_like(dout) if want_dy else None
        ops.epilogue_bwd(dout, out if act != ACT_NONE else None, z, self.scale, self.mean, self.rstd, dy, dz,
                         self.dgamma, self.dbeta, self.db, act)
        return dz, dy

    def wgrad(self, dz, ctx, accumulate=False):
        self.model.wgrad_async(ctx[0], dz, self.wshape, self.stride, self.padding, self.dw, accumulate)

    def wgrad_item(self, dz, ctx, accumulate=False):
        """The arguments of this layer's weight gradient, for MaskRCNNEngine.wgrad_group."""
        return (ctx[0], dz, self.wshape, self.stride, self.padding, self.dw, accumulate)

    def dgrad(self, dz, ctx, out=None, accumulate=False):
        """dx = adjoint of the convolution applied to dz.  `out` (shape of x) receives the result;
        accumulate=True adds to what `out` already holds (in place)."""
        x = ctx[0]
        kh, kw, cin, cout = self.wshape
        N, H, W, _ = x.shape
        fresh = self.model.wt_valid                     # refreshed for all layers at once (refresh_wt)
        if not fresh and not (self.padding == "valid" and kh > 1):
            ops.weight_flip_transpose(self.w, self.wt)
        res = out if accumulate else None
        rm = RES_SAME if accumulate else RES_NONE
        if self.stride == 1 and self.padding == "same":
            if out is None:
                out = ops.empty_like(x)
            ops.conv2d(dz, self.wt, res=res, stride=1, padding=((kh - 1) // 2, (kw - 1) // 2), res_mode=rm, out=out)
        elif kh == 1 and kw == 1 and self.stride == 1:
            if out is None:
                out = ops.empty_like(x)
            ops.conv2d(dz, self.wt, res=res, stride=1, padding="valid", res_mode=rm, out=out)
        elif kh == 1 and kw == 1 and self.stride == 2:
            if out is None:
                out = ops.empty_like(x)
                ops.fill_zero(out)
            d = ops.conv_desc(tuple(dz.shape), (1, 1, cout, cin), 1, "valid", ACT_NONE, rm)
            d.out_w_stride, d.out_h_stride, d.out_n_stride = 2 * cin, 2 * W * cin, H * W * cin
            ops.conv2d(dz, self.wt, res=res, out=out, desc=d)
        elif self.padding == "valid" and x.shape[1] == kh and x.shape[2] == kw:
            # "FC as VALID conv" (mrcnn_class_conv1): dx[M, (kh,kw,ci)] = dz[M, Cout] . W^T
            if out is None:
                out = ops.empty_like(x)
            wt2 = self.wt.view(1, 1, cout, kh * kw * cin)
            if not fresh:
                ops.weight_flip_transpose(self.w.view(1, 1, kh * kw * cin, cout), wt2)
            ops.conv2d(dz.view(N, 1, 1, cout), wt2, res=None if res is None else res.view(N, 1, 1, -1), stride=1,
                       padding="valid", res_mode=rm, out=out.view(N, 1, 1, kh * kw * cin))
        else:
            raise NotImplementedError("dgrad for layer %s" % self.name)
        return out


class Block(object):
    def __init__(self, model, stage, letter, has_shortcut, stride):
        base = "res%d%s_branch" % (stage, letter)
        self.c2a, self.c2b, self.c2c = model.op(base + "2a"), model.op(base + "2b"), model.op(base + "2c")
        self.c1 = model.op(base + "1") if has_shortcut else None
        self.stage, self.stride = stage, stride


_SIDE_STREAMS = {}


def _side_streams(device, wprio=0, aprio=0):
    """The weight-gradient and auxiliary streams of a device: ONE pair per process, shared by every engine.  HIP maps streams
    onto a handful of hardware queues (4 by default) in the order they are first used, and two streams that land on the same
    queue run one after the other.  The first pair created in a process gets queues of its own beside the default stream's;
    a later engine that took fresh streams of its own could land its weight-gradient stream on the main stream's queue --
    measured: the 512 x 512 16-bit step 24.0 -> 29.1 ms for the third model built in one process (bench.py's configs[4] leg
    behind the detect / train-loop legs), back to 24.6 with GPU_MAX_HW_QUEUES=8.  Engines never run concurrently with each
    other on one device, so sharing costs nothing."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return None, None
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), wprio, aprio)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device=dev, priority=wprio), torch.cuda.Stream(device=dev, priority=aprio))
    return _SIDE_STREAMS[key]


class MaskRCNNEngine(object):
    def __init__(self, config, device, weights=None, seed=0):
        self.cfg = config
        self.dev = device
        self.layout = L = ParamLayout(config)
        n = L.total
        f = dict(dtype=torch.float32, device=device)
        self.params = torch.zeros(n, **f)
        self.grads = torch.zeros(n, **f)
        self.momentum = torch.zeros(n, **f)
        self.wt = torch.empty(n, **f)
        c = max(L.bn_channels, 64)
        self.bn_mean, self.bn_var = torch.zeros(c, **f), torch.ones(c, **f)
        self.bn_scale, self.bn_shift, self.bn_rstd = torch.empty(c, **f), torch.empty(c, **f), torch.empty(c, **f)
        self.sumsq = torch.zeros(1, **f)
        self._ops = {}
        for l in L.layers:
            self._ops[l.name] = ConvOp(self, l)
        arch = config.BACKBONE
        n4 = {"resnet50": 5, "resnet101": 22, "custom": 1}[arch]
        letters = {2: "abc", 3: "abcd", 4: "a" + "".join(chr(98 + i) for i in range(n4)), 5: "abc"}
        self.stages = [[Block(self, s, b, i == 0, 2 if (i == 0 and s > 2) else 1) for i, b in enumerate(letters[s])]
                       for s in (2, 3, 4, 5)]
        # data-gradient operands (flipped / transposed kernels) of all layers: one table-driven launch per step
        entries = []
        for l in L.layers:
            o = self._ops[l.name]
            kh, kw, cin, cout = o.wshape
            off = L.offsets[l.name + "/kernel"][0]
            if o.padding == "valid" and kh > 1:                    # FC as VALID conv: plain [K, Cout] transpose
                entries.append((off, 1, 1, kh * kw * cin, cout))
            else:
                entries.append((off, kh, kw, cin, cout))
        self._flip_table = ops.flip_table(entries, device) if torch.device(device).type == "cuda" else None
        self.arena = ops.StepArena()               # tensors of a training step, reused step after step
        if torch.device(device).type == "cuda":
            from . import _hip
            _hip.set_device_index(torch.device(device).index if torch.device(device).index is not None
                                  else torch.cuda.current_device())
        self.wt_valid = False
        self.set_trainable("all")
        self.set_weights(weights if weights is not None else init_weights(L, seed))
        self._anchor_cache = {}
        self._infer_graphs = {}
        self._train_graphs = {}
        self._train_tapes = {}
        # contiguous gradient ranges in the order the backward pass finalises them (for overlapped
        # data-parallel reduction): heads+RPN+FPN kernels, then res5..res2, then conv1 + BatchNorm blocks
        first = lambda prefix: min(o for n_, (o, _, _) in L.offsets.items() if n_.startswith(prefix) and "/kernel" in n_)
        s2, s3, s4, s5, fp = first("res2"), first("res3"), first("res4"), first("res5"), first("fpn_")
        hd = first("mrcnn_")
        self.grad_ranges = {"heads": (hd, L.gamma_offset), "tail": (fp, hd), 5: (s5, fp), 4: (s4, s5), 3: (s3, s4),
                            2: (s2, s3), "head": (0, s2), "bn": (L.gamma_offset, L.total)}
        self.grad_ready = None          # callable(start, end) or None
        self.forced_rpn_rois = None     # test hook: [B, POST_NMS_ROIS_TRAINING, 4] used instead of the ProposalLayer's output
        self.sparse_mask_bwd = True     # skip the exactly-zero rows of the mask-head backward
        # BASELINE configs[4], stage 1: torch.float16 / torch.bfloat16 runs the four 3x3 convolutions of the mask head
        # (forward, data and weight gradient) on the 16-bit matrix cores; master weights, accumulation and every
        # gradient buffer stay float32.  None (default) = float32 everywhere.  loss_scale guards float16 gradients.
        self.head_dtype = None
        self.loss_scale = 4096.0
        self.skipped_steps = torch.zeros(1, dtype=torch.int32, device=device)   # steps the guarded optimiser refused (float16)
        self._skipped_seen = 0
        self.h16_wide = os.environ.get("MRCNN_H16_WIDE", "1") != "0"   # False: only the mask head in 16 bits (round-1 stages 1-2)
        self.h16_blocks = os.environ.get("MRCNN_H16_BLOCKS", "1") != "0"   # bottleneck blocks in 16 bits (stage 4)
        self.h16_all_blocks = os.environ.get("MRCNN_H16_ALL_BLOCKS", "1") != "0"   # 0: only the identity blocks of res4 / res5
        self.h16_fused_bwd = os.environ.get("MRCNN_H16_FUSED_BWD", "1") != "0"     # 16-bit data gradients carry the lower layer's epilogue backward
        self.h16_roialign = os.environ.get("MRCNN_H16_ROIALIGN", "1") != "0"       # ROI heads gather from the 16-bit pyramid (mrcnn_roialign_fwd_h16 / _bwd_h16)
        self._h16 = {}
        # Winograd F(2x2, 3x3) for the float32 3x3 convolutions of the mask head (forward and data gradients): 2.25 x fewer
        # matrix-core flops through three launches per layer; MRCNN_WINOGRAD=0 keeps the direct kernels
        self.winograd = os.environ.get("MRCNN_WINOGRAD", "1") != "0"
        self.winograd_wgrad = os.environ.get("MRCNN_WINOGRAD_WGRAD", "1") != "0"    # weight gradients through the same domain
        # the trunk's large 3x3 layers (FPN smoothing, RPN shared convolution on levels of >= 16 384 pixels) through F(2x2, 3x3)
        # in the forward pass (A/B switch, see DESIGN 4.1e for the measurement)
        self.sk16_infer = os.environ.get("MRCNN_SK16_INFER", "1") != "0"
        # the mask head of the inference graph: 4 = uniform F(4x4, 3x3) tiles (16 overhanging tiles per 14 x 14 map: 36 x 1600 GEMM rows
        # for 100 detections where F(2x2) has 16 x 4900; 1.3e-5 of the result's range instead of 1.6e-6, on probabilities that are
        # thresholded at 0.5) -- detect graph 2.73 -> 2.60 ms; 2 = F(2x2) as in rounds 1-2
        self.infer_mask_tile = int(os.environ.get("MRCNN_WINOGRAD_INFER_MASK_TILE", "4"))
        self.trunk_winograd = os.environ.get("MRCNN_WINOGRAD_TRUNK", "1") != "0"
        self.trunk_winograd_min_rows = int(os.environ.get("MRCNN_WINOGRAD_TRUNK_MIN_ROWS", "4096"))
        # forward as two half-batch chains on two streams: +0.7 ms with the F(2x2) layers, level with the uniform F(4x4) tiling, a loss with
        # the mixed tiling (its small tile groups halve again: ResNet-101 / 4 images 37.6 -> 37.4 ms, ResNet-50 / 2 images 20.8 -> 20.0) -> off
        self.winograd_split = os.environ.get("MRCNN_WINOGRAD_SPLIT", "0") != "0"
        self._wino_V = {}               # layer -> input transform V of this step's forward pass
        self._wino = {}                 # layer -> [U forward, U data gradient] (allocated once, refreshed after weight updates)
        self._wino_valid = {}           # layer -> [forward valid, data-gradient valid]
        self._h16_store = {}            # dtype -> {layer: (W^T image, data-gradient image)}, stable addresses
        self._h16_tables = {}           # (dtype, layer names) -> table of the one-launch refresh (ops.h16_image_table)
        self._h16_valid = False
        self.fused_mask_out_bwd = True  # single-pass backward of the mask-head output stage
        # the persistent 256 x 256 16-bit kernel for the mask head's DATA gradients too (they run beside the weight gradients
        # of the other stream; MRCNN_H16_PHASE_BWD=0 keeps them on the small-LDS kernel that shares CUs)
        self.h16_phase_bwd = os.environ.get("MRCNN_H16_PHASE_BWD", "1") != "0"
        self.fused_dgrad_epilogue = True  # data-gradient convs of the mask head apply the lower layer's epilogue backward
        # stream priorities (HIP: -1 high, 0 normal, 1 low where the runtime has three levels): the weight gradients are off the
        # critical path, so their stream may be given a lower priority than the chain of data gradients it runs beside
        wprio = int(os.environ.get("MRCNN_WGRAD_PRIO", "0"))
        aprio = int(os.environ.get("MRCNN_AUX_PRIO", "0"))
        self.wgrad_stream, self.aux_stream = _side_streams(device, wprio, aprio)
        # A/B switch, default OFF (measured, DESIGN.md "rejected"): the mask head's weight gradients feed nothing in the
        # backward chain, so they can be held back and issued (auxiliary stream) when the backbone's backward pass starts,
        # to fill the chip beside its ~330 small launches; "wgrad_lds_pad" then keeps one workgroup slot per CU free for
        # those.  Result on ResNet-101, 4 images, dense: 60.56 -> 60.26 ms (pad 0), 60.81 (pad 8192): the small kernels are
        # paced by the matrix pipe of their SIMD (serial K chains of 64-cycle fp32 MFMAs), which the large kernel keeps
        # 88 % busy -- beside it they run 4x slower (18 -> 73 us) and the large kernels 1.45x slower: zero sum.
        self.defer_mask_wgrad = os.environ.get("MRCNN_DEFER_MASK_WGRAD", "0") != "0"
        self.defer_lds_pad = int(os.environ.get("MRCNN_DEFER_LDS_PAD", "8192"))
        self._deferred = []
        self.gather_roialign_bwd = os.environ.get("MRCNN_GATHER_ROIALIGN_BWD", "1") != "0"   # class-head ROIAlign adjoint in gather form
        self.multi_launch = os.environ.get("MRCNN_MULTI_LAUNCH", "1") != "0"     # independent small convolutions share launches

    def op(self, name):
        return self._ops[name]

    # ---- weight gradients on a second stream -------------------------------------------------------
    def wgrad_async(self, x, dz, wshape, stride, padding, dw, accumulate):
        """A layer's weight gradient depends only on (x, dz); nothing in the backward chain reads it.  It is
        launched on a side stream so that it overlaps the data-gradient convolution of the same layer: the
        partial last round of one grid (and the small grids of the backbone) fill the CUs the other leaves
        idle.  All weight gradients share that one stream (and its slab workspace), in program order."""
        ws = self.wgrad_stream
        if ws is None:
            ops.conv2d_wgrad(x, dz, wshape, stride, padding, dw=dw, accumulate=accumulate)
            return
        ev = _hip_mod.ev_record(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(ws):
            _hip_mod.ev_wait(ws, ev)
            ops.conv2d_wgrad(x, dz, wshape, stride, padding, dw=dw, accumulate=accumulate)
        dz.record_stream(ws)
        x.record_stream(ws)

    def wgrad_group(self, items):
        """Weight gradients of several layers (ConvOp.wgrad_item) behind one event: one shared launch when they all
        fit the LDS-DMA kernel (the three convolutions of a bottleneck block on the small feature maps), else one by one."""
        ws = self.wgrad_stream

        def run():
            if not (self.multi_launch and 1 < len(items) <= 4 and ops.conv2d_wgrad_multi(items)):
                for x, dz, wshape, stride, padding, dw, acc in items:
                    ops.conv2d_wgrad(x, dz, wshape, stride, padding, dw=dw, accumulate=acc)
        if ws is None:
            run()
            return
        ev = _hip_mod.ev_record(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(ws):
            _hip_mod.ev_wait(ws, ev)
            run()
        for x, dz, *_ in items:
            dz.record_stream(ws)
            x.record_stream(ws)

    def wgrad_h16_async(self, x, dz, wshape, dw, multiplier, padding="same", accumulate=False):
        ws = self.wgrad_stream
        if ws is None:
            ops.conv2d_wgrad_h16(x, dz, wshape, 1, padding, dw=dw, accumulate=accumulate, multiplier=multiplier)
            return
        ev = _hip_mod.ev_record(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(ws):
            _hip_mod.ev_wait(ws, ev)
            ops.conv2d_wgrad_h16(x, dz, wshape, 1, padding, dw=dw, accumulate=accumulate, multiplier=multiplier)
        dz.record_stream(ws)
        x.record_stream(ws)

    # every site that invalidates the 16-bit weight images (optimiser step, set_weights, graph / tape replay, warm-up roll
    # back) changes the weights: the Winograd-domain kernels go stale with them
    @property
    def _h16_valid(self):
        return self._h16_valid_flag

    @_h16_valid.setter
    def _h16_valid(self, v):
        self._h16_valid_flag = bool(v)
        if not v:
            self._wino_valid = {}

    def _wino_U(self, op, which, xshape, train=True, infer_tile=2):
        """Winograd-domain kernel of layer `op` for an input of shape xshape (in training the shape picks the tile size; inference
        takes `infer_tile`, whatever the batch: F(2x2, 3x3) for the trunk -- the RPN scores feed index-exact proposal selection --,
        the uniform F(4x4, 3x3) tiling for the mask head, whose output is thresholded at 0.5): which = 0 forward (from op.w), 1 data
        gradient (from the flipped / transposed op.wt, which must be current).  Refreshed lazily after every weight update, in place."""
        tile = ops.winograd_tile(tuple(xshape)) if train else infer_tile
        key = (op.name, tile)
        ent = self._wino.setdefault(key, [None, None])
        ok = self._wino_valid.setdefault(key, [False, False])
        if not ok[which]:
            src = op.w if which == 0 else op.wt
            ent[which] = ops.winograd_weights(src, out=ent[which], tile=tile)
            ok[which] = True
        return ent[which]

    def _wino_ok(self, op, xshape, min_rows=None):
        return (self.winograd and self.head_dtype is None and
                ops.winograd_ok(tuple(xshape), op.wshape, op.stride, op.padding, min_rows))

    def _wino_trunk_ok(self, op, xshape):
        """Trunk layers (FPN smoothing, RPN shared convolution) take the F(2x2, 3x3) path from 4 096 pixels on (P3 at 256 x 256
        x 4 images, P2 at batch-1 detect): measured product default 20.22 -> 20.03 ms, detect 3.14 -> 3.10 ms, ResNet-50 step
        19.67 -> 19.50 ms, headline level; the mask head keeps its own threshold (ops._WINO_MIN_ROWS)."""
        return self.trunk_winograd and self._wino_ok(op, xshape, self.trunk_winograd_min_rows)

    def _ensure_h16(self):
        """16-bit operand images (W^T and the rotated data-gradient image) of the mask-head convolutions."""
        if self._h16_valid and self._h16.get("dtype") == self.head_dtype:
            return
        # one set of images per dtype, allocated once and refreshed IN PLACE: captured HIP graphs (infer_graphed) hold
        # their addresses, so a weight update must never move them
        imgs = self._h16_store.setdefault(self.head_dtype, {})
        names = ["mrcnn_mask_conv%d" % i for i in range(1, 5)] + ["mrcnn_mask_deconv"]   # deconv: GEMM matrix [Cin, 4*Cd]
        if self.h16_wide:
            names += [n for n in H16_WIDE_LAYERS if self._h16_shape_ok(self.op(n))]
            for stage in self.stages:
                for blk in stage:
                    if self._h16_block(blk):
                        names += [blk.c2a.name, blk.c2b.name, blk.c2c.name] + ([blk.c1.name] if blk.c1 is not None else [])
        # steady state: every image exists (addresses stable) -> ONE table-driven launch refreshes them all; only the
        # class-head FC layers' second image (a plain cast of the HWIO kernel) keeps its own launch
        tkey = (self.head_dtype, tuple(names))
        tab = self._h16_tables.get(tkey)
        if tab is not None:
            ops.weights_to_h16_batched(self.params, tab[0], self.head_dtype)
            for op, wn in tab[1]:
                kh, kw, cin, cout = op.wshape
                ops.cast_to_h16(op.w.view(kh * kw * cin, cout), self.head_dtype, out=wn)
            self._h16 = dict(imgs, dtype=self.head_dtype)
            self._h16_valid = True
            return
        for name in names:
            op = self.op(name)
            if op.padding == "valid" and op.wshape[0] > 1:
                # "FC as VALID conv" (mrcnn_class_conv1): forward image W^T as usual; its data gradient is the GEMM
                # dx[M, K] = dz[M, Cout] . W^T, whose operand image [K][Cout] is the HWIO kernel itself, in 16 bits
                kh, kw, cin, cout = op.wshape
                old = imgs.get(name)
                wf, _ = ops.weights_to_h16(op.w, self.head_dtype, want_dgrad=False, out=None if old is None else (old[0], None))
                wn = old[1] if old is not None else torch.empty((kh * kw * cin, cout), dtype=self.head_dtype, device=self.dev)
                ops.cast_to_h16(op.w.view(kh * kw * cin, cout), self.head_dtype, out=wn)   # never from the step arena
                imgs[name] = (wf, wn)
            else:
                imgs[name] = ops.weights_to_h16(op.w, self.head_dtype, out=imgs.get(name))
        entries, casts = [], []
        for name in names:
            op = self.op(name)
            if op.padding == "valid" and op.wshape[0] > 1:
                entries.append((op.w, imgs[name][0], None))
                casts.append((op, imgs[name][1]))
            else:
                entries.append((op.w, imgs[name][0], imgs[name][1]))
        try:
            self._h16_tables[tkey] = (ops.h16_image_table(self.params, entries, self.dev), casts)
        except AssertionError:
            pass                                            # a kernel that is not a view into the flat buffer: stay layer by layer
        self._h16 = dict(imgs, dtype=self.head_dtype)
        self._h16_valid = True

    @staticmethod
    def _h16_shape_ok(op):
        """Shapes ALL 16-bit kernels take: forward Cin % 32 == 0, Cout % 128 == 0; weight gradient Cin % 256 == 0."""
        kh, kw, cin, cout = op.wshape
        return op.stride == 1 and cin % 256 == 0 and cout % 128 == 0 and kh * kw <= 64

    @staticmethod
    def _h16_fwd_ok(op):
        """Shapes the small-tile forward / data-gradient kernel takes (both directions: Cin and Cout multiples of 64)."""
        kh, kw, cin, cout = op.wshape
        return cin % 64 == 0 and cout % 64 == 0 and kh * kw <= 64 and (op.stride == 1 or (kh == 1 and kw == 1))

    def _h16_layer(self, name):
        """True when layer `name` runs on the 16-bit matrix cores in the current mode."""
        return self.head_dtype is not None and self.h16_wide and name in H16_WIDE_LAYERS and self._h16_shape_ok(self.op(name))

    def _h16_fwd(self, op, x16, act, train, res=None):
        """conv + bias + frozen BN (+ 16-bit residual) + activation of layer `op` on a 16-bit input: 16-bit result; ctx for
        _h16_bwd."""
        d = ops.conv_desc(tuple(x16.shape), op.wshape, op.stride, op.padding, act)
        z = ops.empty((d.N, d.OH, d.OW, d.Cout), x16.dtype, x16.device) if (train and op.bn) else None
        y = ops.conv2d_h16(x16, self._h16[op.name][0], op.wshape, op.b, op.scale, op.shift, op.stride, op.padding, act, z_out=z,
                           res=res)
        return y, ((x16, z, y, act) if train else None)

    def _h16_bwd(self, op, d16, ctx, S, accumulate_w=False, need_dx=True, want_dy=False, dx_res=None):
        """Backward of _h16_fwd.  d16: gradient w.r.t. the activated output, 16 bit, times the loss scale S.  Bias / BN
        sums and the weight gradient land unscaled in the float32 gradient buffer; returns dx (16 bit, times S; plus
        dx_res when given), or (dx, dy) with want_dy: dy = d16 * act' is what a shortcut into this layer's Add receives."""
        x16, z, y, act = ctx
        got = ops.epilogue_bwd_h16(d16, y if act != ACT_NONE else None, z, op.scale, op.mean, op.rstd, op.dgamma, op.dbeta, op.db,
                                   act, 1.0 / S, want_dy=want_dy)
        dz, dy = got if want_dy else (got, None)
        if self._h16_shape_ok(op):
            self.wgrad_h16_async(x16, dz, op.wshape, op.dw, 1.0 / S, op.padding, accumulate_w)
        else:
            self._wgrad_from_h16(op, x16, dz, 1.0 / S, accumulate_w)
        if not need_dx:
            return (dz, dy) if want_dy else dz                    # the caller issues the data gradient itself
        dx = self._h16_dgrad(op, dz, dx_res)
        return (dx, dy) if want_dy else dx

    def _wgrad_from_h16(self, op, x16, dz16, mult, accumulate):
        """Weight gradient of a layer whose channel counts are below the 16-bit weight-gradient kernel's 256-channel tile
        (res2 / res3, the first block of a stage): both operands are widened to float32 ON THE WEIGHT-GRADIENT STREAM and the
        float32 kernel runs there -- off the critical path, which stays in 16 bits."""
        ws = self.wgrad_stream
        ev = _hip_mod.ev_record(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(ws):
            _hip_mod.ev_wait(ws, ev)
            x32 = ops.cast_from_h16(x16)
            dz32 = ops.cast_from_h16(dz16, mult)
            ops.conv2d_wgrad(x32, dz32, op.wshape, op.stride, op.padding, dw=op.dw, accumulate=accumulate)
        for t in (x16, dz16):
            t.record_stream(ws)

    def _h16_dgrad(self, op, dz, dx_res=None, out=None, out_strides=None):
        kh, kw, cin, cout = op.wshape
        if op.padding == "valid" and kh > 1:                     # FC as VALID conv: one GEMM over the flattened window
            M = dz.shape[0]
            dx = ops.conv2d_h16(dz.view(M, 1, 1, cout), self._h16[op.name][1], (1, 1, cout, kh * kw * cin), None, None, None, 1,
                                "valid", ACT_NONE)
            return dx.view(M, kh, kw, cin)
        return ops.conv2d_h16(dz, self._h16[op.name][1], (kh, kw, cout, cin), None, None, None, 1,
                              ((kh - 1) // 2, (kw - 1) // 2) if op.padding == "same" else "valid", ACT_NONE, res=dx_res, out=out,
                              out_strides=out_strides)

    # ---- identity bottleneck blocks in 16 bits (configs[4], stage 4) ------------------------------------------
    def _h16_block(self, blk):
        """A bottleneck block (mrcnn/model.py:99-172) that runs in 16 bits: every block whose convolutions fit the small-tile
        kernel (channel counts multiples of 64: all of ResNet-50 / 101's 16 / 33 blocks) -- identity blocks and the first
        block of each stage (projection shortcut, stride 2 on the 1x1 convolutions)."""
        ops_ = [blk.c2a, blk.c2b, blk.c2c] + ([blk.c1] if blk.c1 is not None else [])
        return (self.head_dtype is not None and self.h16_wide and self.h16_blocks and self.wgrad_stream is not None and
                all(self._h16_fwd_ok(o) for o in ops_) and (self.h16_all_blocks or
                                                             (blk.c1 is None and all(self._h16_shape_ok(o) for o in ops_))))

    def _block_fwd_h16(self, blk, x16, train):
        a, ca = self._h16_fwd(blk.c2a, x16, ACT_RELU, train)
        b, cb = self._h16_fwd(blk.c2b, a, ACT_RELU, train)
        sc, c1c = (x16, None) if blk.c1 is None else self._h16_fwd(blk.c1, x16, ACT_NONE, train)
        y, cc = self._h16_fwd(blk.c2c, b, ACT_RELU, train, res=sc)
        return y, ("h16", ca, cb, cc, c1c)

    def _h16_wgrad(self, op, x16, dz, S, accumulate=False):
        if self._h16_shape_ok(op):
            self.wgrad_h16_async(x16, dz, op.wshape, op.dw, 1.0 / S, op.padding, accumulate)
        else:
            self._wgrad_from_h16(op, x16, dz, 1.0 / S, accumulate)

    def _h16_dgrad_ep(self, op, dz, below, below_ctx, S):
        """dz of layer `below` from the dz of layer `op` above it: the 16-bit data gradient with `below`'s epilogue
        backward in its own epilogue (mrcnn_conv2d_dgrad_ep_h16); the two launches when the shape has no fused kernel."""
        kh, kw, cin, cout = op.wshape
        _, z, y, act = below_ctx
        pad = ((kh - 1) // 2, (kw - 1) // 2) if op.padding == "same" else "valid"
        if self.h16_fused_bwd and op.stride == 1:
            got = ops.conv2d_dgrad_ep_h16(dz, self._h16[op.name][1], (kh, kw, cout, cin), pad, y if act != ACT_NONE else None, z,
                                          below.scale, below.mean, below.rstd, below.dgamma, below.dbeta, below.db, act, 1.0 / S)
            if got is not None:
                return got
        d = self._h16_dgrad(op, dz)
        return ops.epilogue_bwd_h16(d, y if act != ACT_NONE else None, z, below.scale, below.mean, below.rstd, below.dgamma,
                                    below.dbeta, below.db, act, 1.0 / S)

    def _block_bwd_h16(self, blk, d16, ctxs, S):
        """d16: gradient w.r.t. the block output (16 bit, times S) -> gradient w.r.t. the block input.  Four launches on the
        main stream for an identity block (output-stage backward, two data gradients that carry the backward epilogue of the
        layer below, the last data gradient with the shortcut's gradient on its residual port); weight gradients on the side
        stream."""
        _, ca, cb, cc, c1c = ctxs
        c2a, c2b, c2c = blk.c2a, blk.c2b, blk.c2c
        dzc, dy = ops.epilogue_bwd_h16(d16, cc[2], cc[1], c2c.scale, c2c.mean, c2c.rstd, c2c.dgamma, c2c.dbeta, c2c.db, ACT_RELU,
                                       1.0 / S, want_dy=True)
        self._h16_wgrad(c2c, cc[0], dzc, S)
        dzb = self._h16_dgrad_ep(c2c, dzc, c2b, cb, S)
        self._h16_wgrad(c2b, cb[0], dzb, S)
        dza = self._h16_dgrad_ep(c2b, dzb, c2a, ca, S)
        self._h16_wgrad(c2a, ca[0], dza, S)
        if blk.c1 is None:
            return self._h16_dgrad(c2a, dza, dx_res=dy)           # identity shortcut: dx = dgrad_2a + dy
        # projection shortcut: dx = dgrad_1(dy) + dgrad_2a(dza); with stride 2 both 1x1 data gradients scatter into the
        # even pixels of a zeroed tensor (the second adds through the residual port, same strides)
        x16 = ca[0]
        N, H, W, C = x16.shape
        st = c2a.stride
        dz1 = self._h16_bwd(blk.c1, dy, c1c, S, need_dx=False)
        dx = ops.empty((N, H, W, C), x16.dtype, x16.device)
        strides = None
        if st != 1:
            ops.fill_zero(dx)
            strides = (H * W * C, st * W * C, st * C)
        self._h16_dgrad(blk.c1, dz1, out=dx, out_strides=strides)
        self._h16_dgrad(c2a, dza, dx_res=dx, out=dx, out_strides=strides)
        return dx

    def _S(self):
        return float(self.loss_scale) if self.head_dtype == torch.float16 else 1.0

    # ---- independent small convolutions in one launch (mrcnn_conv2d_fwd_multi) ------------------------
    def _forward_multi(self, layers, xs, act=ACT_NONE, train=False):
        """[ConvOp.forward(x, act) for layer, x in zip(layers, xs)] as one launch; falls back to the loop.  With trunk_winograd the
        3 x 3 layers large enough for the Winograd path (FPN smoothing / RPN shared convolution on P2, P3 of big inputs) leave the
        group and run F(2x2, 3x3) on their own; their training contexts are the direct kernels' (the backward pass is unchanged)."""
        if self.trunk_winograd and any(self._wino_trunk_ok(op, x.shape) for op, x in zip(layers, xs)):
            res = [None] * len(layers)
            rest = []
            for i, (op, x) in enumerate(zip(layers, xs)):
                if self._wino_trunk_ok(op, x.shape):
                    out = ops.empty(tuple(x.shape[:3]) + (op.wshape[3],), torch.float32, x.device)
                    z = ops.empty_like(out) if (train and op.bn) else None
                    V = None
                    if train and self.winograd_wgrad:       # the input transform is kept: the weight gradient contracts it with A dz A^T
                        V = ops.empty((ops.winograd_v_floats(tuple(x.shape)),), torch.float32, x.device)
                        self._wino_V[(op.name, tuple(x.shape))] = V
                    ops.conv2d_winograd(x, self._wino_U(op, 0, x.shape, train), op.b, op.scale, op.shift, act, out=out, z_out=z, keep_v=V)
                    res[i] = (out, (x, z, out, act) if train else None)
                else:
                    rest.append(i)
            if rest:
                sub = self._forward_multi([layers[i] for i in rest], [xs[i] for i in rest], act, train) if len(rest) > 1 else \
                    [layers[rest[0]].forward(xs[rest[0]], act, train=train)]
                for i, r in zip(rest, sub):
                    res[i] = r
            return res
        if self.multi_launch and 1 < len(layers) <= 5:
            probs, zs = [], []
            for op, x in zip(layers, xs):
                d = ops.conv_desc(tuple(x.shape), op.wshape, op.stride, op.padding, act)
                out = ops.empty((d.N, d.OH, d.OW, d.Cout), torch.float32, x.device)
                z = ops.empty_like(out) if (train and op.bn) else None
                zs.append(z)
                probs.append(dict(x=x, w=op.w, bias=op.b, scale=op.scale, shift=op.shift, stride=op.stride,
                                  padding=op.padding, act=act, out=out, z_out=z))
            outs = ops.conv2d_multi(probs)
            if outs is not None:
                return [(o, (x, z, o, act) if train else None) for o, x, z in zip(outs, xs, zs)]
        return [op.forward(x, act, train=train) for op, x in zip(layers, xs)]

    def _dgrad_multi(self, layers, dzs, ctxs, outs=None, accumulate=False):
        """[ConvOp.dgrad(dz, ctx, out, accumulate)] for stride-1 SAME / 1x1 layers as one launch (else the loop)."""
        outs = list(outs) if outs is not None else [None] * len(layers)
        accs = list(accumulate) if isinstance(accumulate, (list, tuple)) else [accumulate] * len(layers)
        ok = self.multi_launch and self.wt_valid and 1 < len(layers) <= 5 and all(
            op.stride == 1 and (op.padding == "same" or (op.wshape[0] == 1 and op.wshape[1] == 1)) for op in layers)
        if ok:
            probs = []
            for i, (op, dz, ctx) in enumerate(zip(layers, dzs, ctxs)):
                kh, kw = op.wshape[0], op.wshape[1]
                if outs[i] is None:
                    assert not accs[i]
                    outs[i] = ops.empty_like(ctx[0])
                probs.append(dict(x=dz, w=op.wt, padding=((kh - 1) // 2, (kw - 1) // 2) if op.padding == "same" else "valid",
                                  res=outs[i] if accs[i] else None, res_mode=RES_SAME if accs[i] else RES_NONE,
                                  out=outs[i]))
            if ops.conv2d_multi(probs) is not None:
                return outs
        return [op.dgrad(dz, ctx, out=o, accumulate=a) for op, dz, ctx, o, a in zip(layers, dzs, ctxs, outs, accs)]

    def _bwd_group(self, layers, dzs, ctxs, outs=None, accumulate=False, wgrad_acc=None):
        """Weight gradients (weight-gradient stream) and data gradients of a group of stride-1 convolutions from their dz.  Layers
        whose forward went through the Winograd path (trunk_winograd: a kept input transform under (name, input shape)) take it
        here too -- weight gradient from V, data gradient F(2x2, 3x3) and, where the destination accumulates, one add pass --,
        the others share the multi-problem launches as before.  wgrad_acc[i]: the layer's weight gradient adds to dw (shared
        weights: the RPN convolution over the pyramid levels)."""
        n = len(layers)
        outs = list(outs) if outs is not None else [None] * n
        accs = list(accumulate) if isinstance(accumulate, (list, tuple)) else [accumulate] * n
        wacc = list(wgrad_acc) if wgrad_acc is not None else [False] * n
        Vs = [self._wino_V.pop((op.name, tuple(c[0].shape)), None) for op, c in zip(layers, ctxs)]
        rest = [i for i in range(n) if Vs[i] is None]
        if rest and not any(wacc):                          # independent weights: the direct ones of the group share a launch as before
            self.wgrad_group([layers[i].wgrad_item(dzs[i], ctxs[i]) for i in rest])
        for i in range(n):                                  # (shared weights: in layer order, each adding to the one before)
            if Vs[i] is not None:
                self._wino_wgrad_async(Vs[i], tuple(ctxs[i][0].shape), dzs[i], layers[i].dw, wacc[i])   # never deferred: trunk
            elif any(wacc):
                self.wgrad_async(*layers[i].wgrad_item(dzs[i], ctxs[i], accumulate=wacc[i]))
        if rest:
            sub = self._dgrad_multi([layers[i] for i in rest], [dzs[i] for i in rest], [ctxs[i] for i in rest],
                                    outs=[outs[i] for i in rest], accumulate=[accs[i] for i in rest])
            for i, o in zip(rest, sub):
                outs[i] = o
        for i in range(n):
            if Vs[i] is None:
                continue
            op = layers[i]
            d = ops.conv2d_winograd(dzs[i], self._wino_U(op, 1, dzs[i].shape))
            if accs[i]:
                ops.add_inplace(outs[i], d)
            else:
                outs[i] = d
        return outs

    def join_wgrad(self):
        if self.wgrad_stream is not None:
            _hip_mod.stream_wait(torch.cuda.current_stream(self.dev), self.wgrad_stream)

    # ---- deferred mask-head weight gradients ----------------------------------------------------------
    def _mask_wgrad(self, kind, *args):
        """A weight gradient of the mask head: now (on the weight-gradient stream) or, with ``defer_mask_wgrad``, beside
        the backbone's backward pass (``_flush_deferred``).  kind "f32": ConvOp.wgrad_item tuple; "h16": the arguments of
        wgrad_h16_async."""
        if self.defer_mask_wgrad and self.aux_stream is not None:
            self._deferred.append((kind, args))
        elif kind == "f32":
            self.wgrad_async(*args)
        elif kind == "wino":
            self._wino_wgrad_async(*args)
        else:
            self.wgrad_h16_async(*args)

    def _wino_wgrad_async(self, V, xshape, dz, dw, acc):
        """Weight gradient from a kept Winograd input transform, on the weight-gradient stream behind the current one."""
        ws = self.wgrad_stream
        if ws is None:
            ops.conv2d_wgrad_winograd(V, xshape, dz, dw, accumulate=acc)
            return
        ev = _hip_mod.ev_record(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(ws):
            _hip_mod.ev_wait(ws, ev)
            ops.conv2d_wgrad_winograd(V, xshape, dz, dw, accumulate=acc)
        dz.record_stream(ws)
        V.record_stream(ws)

    def _flush_deferred(self):
        """Issue the deferred weight gradients on the auxiliary stream, behind everything the main stream has done so far."""
        if not self._deferred:
            return
        aux, main = self.aux_stream, torch.cuda.current_stream(self.dev)
        ev = _hip_mod.ev_record(main)
        if self.defer_lds_pad:
            ops.tuning_set("wgrad_lds_pad", self.defer_lds_pad)
        try:
            with torch.cuda.stream(aux):
                _hip_mod.ev_wait(aux, ev)
                for kind, a in self._deferred:
                    if kind == "f32":
                        x, dz, wshape, stride, padding, dw, acc = a
                        ops.conv2d_wgrad(x, dz, wshape, stride, padding, dw=dw, accumulate=acc)
                    elif kind == "wino":
                        x, xshape, dz, dw, acc = a
                        ops.conv2d_wgrad_winograd(x, xshape, dz, dw, accumulate=acc)
                    else:
                        x, dz, wshape, dw, mult = a
                        ops.conv2d_wgrad_h16(x, dz, wshape, 1, "same", dw=dw, multiplier=mult)
                    dz.record_stream(aux)
                    x.record_stream(aux)
        finally:
            if self.defer_lds_pad:
                ops.tuning_set("wgrad_lds_pad", 0)
        self._deferred = []
        self._aux_pending = True

    def join_aux(self):
        if getattr(self, "_aux_pending", False):
            _hip_mod.stream_wait(torch.cuda.current_stream(self.dev), self.aux_stream)
            self._aux_pending = False

    # ---- weights in / out (Keras layouts at this boundary) --------------------------------------
    def refresh_wt(self):
        """Flipped / transposed kernels of every layer from the current parameters (one launch)."""
        ops.weight_flip_transpose_batched(self.params, self.wt, self._flip_table)
        self.wt_valid = True

    def set_weights(self, weights, strict=True):
        L = self.layout
        self.wt_valid = False
        self._h16_valid = False
        for name, (off, n, shape) in L.offsets.items():
            if name not in weights:
                if strict:
                    raise KeyError("missing weight tensor " + name)
                continue
            a = np.asarray(weights[name], dtype=np.float32)
            if name == "mrcnn_mask_deconv/kernel" and a.shape == (2, 2, shape[3], shape[0]):
                a = deconv_keras_to_gemm(a)
            if a.size != n:
                raise ValueError("shape mismatch for %s: %s vs %s" % (name, a.shape, shape))
            self.params[off:off + n].copy_(torch.from_numpy(np.ascontiguousarray(a).reshape(-1)))
        for l in L.bn_layers:
            c0, c = L.bn_channel_offset[l.bn], l.shape[3]
            for key, buf in (("moving_mean", self.bn_mean), ("moving_variance", self.bn_var)):
                if l.bn + "/" + key in weights:
                    buf[c0:c0 + c].copy_(torch.from_numpy(np.asarray(weights[l.bn + "/" + key], dtype=np.float32)))
                elif strict:
                    raise KeyError("missing weight tensor %s/%s" % (l.bn, key))
        self.fold_bn()

    def get_weights(self, grads=False):
        L = self.layout
        src = (self.grads if grads else self.params).detach().cpu().numpy()
        out = {}
        for name, (off, n, shape) in L.offsets.items():
            a = src[off:off + n].reshape(shape).copy()
            if name == "mrcnn_mask_deconv/kernel":
                a = deconv_gemm_to_keras(a)
            out[name] = a
        if not grads:
            mean, var = self.bn_mean.cpu().numpy(), self.bn_var.cpu().numpy()
            for l in L.bn_layers:
                c0, c = L.bn_channel_offset[l.bn], l.shape[3]
                out[l.bn + "/moving_mean"] = mean[c0:c0 + c].copy()
                out[l.bn + "/moving_variance"] = var[c0:c0 + c].copy()
        return out

    def fold_bn(self):
        L = self.layout
        if L.bn_channels:
            g = self.params[L.gamma_offset:L.gamma_offset + L.bn_channels]
            b = self.params[L.beta_offset:L.beta_offset + L.bn_channels]
            ops.bn_fold(g, b, self.bn_mean[:L.bn_channels], self.bn_var[:L.bn_channels],
                        self.bn_scale[:L.bn_channels], self.bn_shift[:L.bn_channels], self.bn_rstd[:L.bn_channels])

    def set_trainable(self, layer_regex):
        mask = self.layout.trainable_mask(layer_regex)
        self.trainable_host = mask
        coef = torch.tensor(granule_coefficients(self.layout, self.cfg.WEIGHT_DECAY, mask), dtype=torch.float32)
        # ONE buffer for the life of the engine, refreshed in place: recorded launch tapes and captured step graphs carry its
        # address in their grad_prepare / sgd_momentum calls (a second train(..., layers=...) on the same model replays them)
        if getattr(self, "gran_coef", None) is None or self.gran_coef.numel() != coef.numel():
            self.gran_coef = torch.empty(coef.numel(), dtype=torch.float32, device=self.dev)
            self._train_tapes, self._train_graphs = {}, {}
        self.gran_coef.copy_(coef)           # (the backward pass computes every gradient whatever the mask: the mask acts here only)

    # ---- anchors -------------------------------------------------------------------------------
    def anchors(self, image_shape):
        from . import utils
        key = tuple(int(v) for v in image_shape)
        if key not in self._anchor_cache:
            a = utils.get_anchors(self.cfg, key)
            self._anchor_cache[key] = torch.tensor(a, dtype=torch.float32, device=self.dev)
        return self._anchor_cache[key]

    # =========================================================================================
    #  forward
    # =========================================================================================
    def _trunk_fwd(self, images, train):
        """resnet_graph + FPN (model.py:175-210, 2005-2026).  Returns pyramid [P2..P6] and the tape."""
        tape = {}
        c1 = self.op("conv1")
        x, tape["conv1"] = c1.forward(images, ACT_RELU, train=train)
        if train:
            x, am = ops.maxpool3x3s2(x, want_argmax=True)
            tape["pool"] = (am, tuple(tape["conv1"][2].shape))
        else:
            x = ops.maxpool3x3s2(x)
        feats = []
        x16 = None                                  # the running activation in 16 bits (valid beside or instead of x)
        for stage in self.stages:
            for blk in stage:
                if self._h16_block(blk):
                    if x16 is None:
                        self._ensure_h16()
                        x16 = ops.cast_to_h16(x, self.head_dtype)
                    x16, ctx = self._block_fwd_h16(blk, x16, train)
                    x = None
                    if train:
                        tape[id(blk)] = ctx
                    continue
                if x is None:
                    x = ops.cast_from_h16(x16)
                x16 = None
                a, ca = blk.c2a.forward(x, ACT_RELU, train=train)
                b, cb = blk.c2b.forward(a, ACT_RELU, train=train)
                if blk.c1 is not None:
                    sc, c1c = blk.c1.forward(x, ACT_NONE, train=train)
                else:
                    sc, c1c = x, None
                x, cc = blk.c2c.forward(b, ACT_RELU, res=sc, res_mode=RES_SAME, train=train)
                if train:
                    tape[id(blk)] = (ca, cb, cc, c1c)
            if x is None:                           # stage output in float32 for the FPN lateral (x16 stays valid for the next block)
                x = ops.cast_from_h16(x16)
            feats.append(x)
        C2, C3, C4, C5 = feats
        P5s, tape["fpn_c5p5"] = self.op("fpn_c5p5").forward(C5, train=train)
        P4s, tape["fpn_c4p4"] = self.op("fpn_c4p4").forward(C4, res=P5s, res_mode=RES_UP2, train=train)
        P3s, tape["fpn_c3p3"] = self.op("fpn_c3p3").forward(C3, res=P4s, res_mode=RES_UP2, train=train)
        P2s, tape["fpn_c2p2"] = self.op("fpn_c2p2").forward(C2, res=P3s, res_mode=RES_UP2, train=train)
        names = ("fpn_p2", "fpn_p3", "fpn_p4", "fpn_p5")
        self._p16 = None
        self._p16_heads = None                     # 16-bit P2..P5 of this pass for the ROI heads' 16-bit ROIAlign (configs[4])
        if all(self._h16_layer(n) for n in names):
            self._ensure_h16()
            outs, p16 = [], []
            for n, x in zip(names, (P2s, P3s, P4s, P5s)):
                y16, tape[n] = self._h16_fwd(self.op(n), ops.cast_to_h16(x, self.head_dtype), ACT_NONE, train)
                p16.append(y16)
                outs.append(ops.cast_from_h16(y16))             # float32 copies for ROIAlign and P6
            P2, P3, P4, P5 = outs
            P6 = ops.subsample2(P5)
            self._p16 = p16 + [ops.cast_to_h16(P6, self.head_dtype)]
            if self.h16_roialign and all(t.shape[3] % 4 == 0 for t in p16):
                self._p16_heads = list(p16)
            return [P2, P3, P4, P5, P6], tape
        res = self._forward_multi([self.op(n) for n in names], [P2s, P3s, P4s, P5s], train=train)
        (P2, P3, P4, P5) = [r[0] for r in res]
        for n, r in zip(names, res):
            tape[n] = r[1]
        P6 = ops.subsample2(P5)
        return [P2, P3, P4, P5, P6], tape

    def _rpn_fwd(self, pyramid, train):
        """rpn_graph over the 5 levels with shared weights, outputs concatenated level-major
        (model.py:916-957, 2040-2055)."""
        B = pyramid[0].shape[0]
        na = len(self.cfg.RPN_ANCHOR_RATIOS)
        A = sum(p.shape[1] * p.shape[2] * na for p in pyramid)
        logits = ops.empty((B, A, 2), torch.float32, self.dev)
        bbox = ops.empty((B, A, 4), torch.float32, self.dev)
        shared, cls, box = self.op("rpn_conv_shared"), self.op("rpn_class_raw"), self.op("rpn_bbox_pred")
        tape, off = [], 0
        h16 = self._h16_layer("rpn_conv_shared") and self.multi_launch and len(pyramid) <= 5
        if h16:
            self._ensure_h16()
            p16 = self._p16 if self._p16 is not None else [ops.cast_to_h16(p, self.head_dtype) for p in pyramid]
            s16 = [self._h16_fwd(shared, x16, ACT_RELU, train) for x16 in p16]
            ss = [ops.cast_from_h16(y) for y, _ in s16]          # the two small heads (6 / 12 columns) stay float32
        if self.multi_launch and len(pyramid) <= 5:
            # the levels are independent and P3..P6 are a handful of workgroups each: one launch per layer for all levels
            if not h16:
                if self.trunk_winograd and any(self._wino_trunk_ok(shared, p.shape) for p in pyramid):
                    ss = [r[0] for r in self._forward_multi([shared] * len(pyramid), pyramid, ACT_RELU, train)]
                else:
                    ss = ops.conv2d_multi([dict(x=p, w=shared.w, bias=shared.b, act=ACT_RELU) for p in pyramid])
            assert ss is not None                       # one weight tensor: the levels always share a launch shape
            offs = []
            for p in pyramid:
                offs.append(off)
                off += p.shape[1] * p.shape[2] * na
            ops.conv2d_multi([dict(x=s, w=cls.w, bias=cls.b, padding="valid", out_ptr=logits.data_ptr() + o * 2 * 4,
                                   out_strides=(A * 2, s.shape[2] * 2 * na, 2 * na)) for s, o in zip(ss, offs)])
            ops.conv2d_multi([dict(x=s, w=box.w, bias=box.b, padding="valid", out_ptr=bbox.data_ptr() + o * 4 * 4,
                                   out_strides=(A * 4, s.shape[2] * 4 * na, 4 * na)) for s, o in zip(ss, offs)])
            for lvl, (p, s, o) in enumerate(zip(pyramid, ss, offs)):
                cs = (s16[lvl][1] if h16 else (p, None, s, ACT_RELU)) if train else None
                tape.append((cs, (s, None, None, ACT_NONE), o, p.shape[1], p.shape[2]))
            self._p16 = None
            return logits, ops.softmax_rows(logits), bbox, tape
        for p in pyramid:
            s, cs = shared.forward(p, ACT_RELU, train=train)
            H, W = p.shape[1], p.shape[2]
            ops.conv2d_into(s, cls.w, cls.b, logits.data_ptr() + off * 2 * 4, A * 2, W * 2 * na, 2 * na, 1, "valid")
            ops.conv2d_into(s, box.w, box.b, bbox.data_ptr() + off * 4 * 4, A * 4, W * 4 * na, 4 * na, 1, "valid")
            tape.append((cs, (s, None, None, ACT_NONE), off, H, W))
            off += H * W * na
        probs = ops.softmax_rows(logits)
        return logits, probs, bbox, tape

    def _class_head_fwd(self, rois, fms, image_area, train):
        """fpn_classifier_graph (model.py:986-1039)."""
        cfg = self.cfg
        B, R = rois.shape[0], rois.shape[1]
        h16 = self._h16_layer("mrcnn_class_conv1") and self._h16_layer("mrcnn_class_conv2")
        p16h = getattr(self, "_p16_heads", None) if h16 else None
        if p16h is not None:                       # 16-bit pyramid -> 16-bit pooled features: no float32 gather, no cast pass
            x16 = ops.roialign_h16(rois, p16h, cfg.POOL_SIZE, image_area).view(B * R, cfg.POOL_SIZE, cfg.POOL_SIZE, -1)
        else:
            pooled = ops.roialign(rois, fms, cfg.POOL_SIZE, image_area)
            x = pooled.view(B * R, cfg.POOL_SIZE, cfg.POOL_SIZE, -1)
        if h16:
            self._ensure_h16()
            if p16h is None:
                x16 = ops.cast_to_h16(x, self.head_dtype)
            h1, c1 = self._h16_fwd(self.op("mrcnn_class_conv1"), x16, ACT_RELU, train)
            h2_16, c2 = self._h16_fwd(self.op("mrcnn_class_conv2"), h1, ACT_RELU, train)
            h2 = ops.cast_from_h16(h2_16)                        # the 4- / 16-column output layers stay float32
        else:
            h1, c1 = self.op("mrcnn_class_conv1").forward(x, ACT_RELU, train=train)
            h2, c2 = self.op("mrcnn_class_conv2").forward(h1, ACT_RELU, train=train)
        lg, c3 = self.op("mrcnn_class_logits").forward(h2, train=train)
        bb, c4 = self.op("mrcnn_bbox_fc").forward(h2, train=train)
        logits = lg.view(B, R, cfg.NUM_CLASSES)
        probs = ops.softmax_rows(logits)
        return logits, probs, bb.view(B, R, cfg.NUM_CLASSES, 4), (c1, c2, c3, c4)

    def _mask_head_fwd(self, rois, fms, image_area, train):
        """build_fpn_mask_graph (model.py:1042-1091)."""
        cfg = self.cfg
        B, R = rois.shape[0], rois.shape[1]
        p16h = getattr(self, "_p16_heads", None) if self.head_dtype is not None else None
        if p16h is not None:
            x = None
            x16 = ops.roialign_h16(rois, p16h, cfg.MASK_POOL_SIZE, image_area).view(B * R, cfg.MASK_POOL_SIZE, cfg.MASK_POOL_SIZE, -1)
        else:
            pooled = ops.roialign(rois, fms, cfg.MASK_POOL_SIZE, image_area)
            x = pooled.view(B * R, cfg.MASK_POOL_SIZE, cfg.MASK_POOL_SIZE, -1)
        ctxs = []
        if self.head_dtype is None and self.winograd_split and self.wgrad_stream is not None and x.shape[0] >= 512 and \
                all(self._wino_ok(self.op("mrcnn_mask_conv%d" % i), (x.shape[0] // 2,) + tuple(x.shape[1:3]) + (self.op("mrcnn_mask_conv%d" % i).wshape[2],))
                    for i in range(1, 5)):
            x, ctxs = self._mask_convs_fwd_split(x, train)
        elif self.head_dtype is None:
            for i in range(1, 5):
                op = self.op("mrcnn_mask_conv%d" % i)
                if self._wino_ok(op, x.shape):
                    out = ops.empty(tuple(x.shape[:3]) + (op.wshape[3],), torch.float32, self.dev)
                    z = ops.empty_like(out) if (train and op.bn) else None
                    # training keeps the input transform: the layer's weight gradient contracts it with the transformed dz
                    V = ops.empty((ops.winograd_v_floats(tuple(x.shape)),), torch.float32, self.dev) if (train and self.winograd_wgrad) else None
                    if V is not None:
                        self._wino_V[op.name] = [(V, 0, x.shape[0])]
                    ops.conv2d_winograd(x, self._wino_U(op, 0, x.shape, train, self.infer_mask_tile), op.b, op.scale, op.shift, ACT_RELU, out=out, z_out=z, keep_v=V)
                    x, c = out, ((x, z, out, ACT_RELU) if train else None)
                else:
                    x, c = op.forward(x, ACT_RELU, train=train)
                ctxs.append(c)
        else:                                   # 16-bit matrix cores; float32 again from the deconvolution on
            self._ensure_h16()
            h = x16 if p16h is not None else ops.cast_to_h16(x, self.head_dtype)
            for i in range(1, 5):
                op = self.op("mrcnn_mask_conv%d" % i)
                z = ops.empty(tuple(h.shape[:3]) + (op.wshape[3],), self.head_dtype, self.dev) if (train and op.bn) else None
                y = ops.conv2d_h16(h, self._h16[op.name][0], op.wshape, op.b, op.scale, op.shift, 1, "same", ACT_RELU, z_out=z)
                ctxs.append((h, z, y, ACT_RELU) if train else None)
                h = y
            dc, mop = self.op("mrcnn_mask_deconv"), self.op("mrcnn_mask")
            Cd = dc.wshape[3] // 4
            C_ = mop.wshape[3]
            if Cd % 256 == 0 and C_ <= 16:             # deconvolution + mask 1x1 conv + sigmoid on the 16-bit tensors too
                up = ops.deconv2x2_h16(h, self._h16[dc.name][0], dc.b, Cd, ACT_RELU)
                m = ops.mask_out_fwd_h16(up, mop.w.view(mop.wshape[2], C_), mop.b)
                ctxs.append((h, None, up, ACT_RELU) if train else None)
                ctxs.append((up, None, m, ACT_SIGMOID) if train else None)
                return m.view(B, R, m.shape[1], m.shape[2], m.shape[3]), ctxs
            x = ops.cast_from_h16(h)
        dc = self.op("mrcnn_mask_deconv")
        up = ops.deconv2x2(x, dc.w.view(dc.wshape[2], dc.wshape[3]), dc.b, ACT_RELU)
        m, cm = self.op("mrcnn_mask").forward(up, ACT_SIGMOID, train=train)
        ctxs.append((x, None, up, ACT_RELU) if train else None)
        ctxs.append(cm)
        return m.view(B, R, m.shape[1], m.shape[2], m.shape[3]), ctxs

    def _mask_convs_fwd_split(self, x, train):
        """The four Winograd layers of the mask head on the two halves of the ROI rows, one half per stream (main and the
        weight-gradient stream, idle during the forward pass) and the second half one transform behind the first: a layer is
        a memory pass, a matrix-core pass and a memory pass in a row, so two chains that are out of phase keep both busy.
        Activations / z are whole tensors (the backward pass and the transposed convolution see one batch); the input transforms
        kept for the weight gradients are per half."""
        main, side = torch.cuda.current_stream(self.dev), self.wgrad_stream
        N = x.shape[0]
        cut = (N // 2 + 3) // 4 * 4
        halves = ((0, cut, main), (cut, N, side))
        ops_ = [self.op("mrcnn_mask_conv%d" % i) for i in range(1, 5)]
        hshape = lambda op, a, b: (b - a,) + tuple(x.shape[1:3]) + (op.wshape[2],)
        Us = [[self._wino_U(op, 0, hshape(op, a, b), train, self.infer_mask_tile) for a, b, _ in halves] for op in ops_]   # weight transforms on the main stream, before the fork
        outs, zs, ctxs = [], [], []
        cur = x
        for op in ops_:
            out = ops.empty(tuple(x.shape[:3]) + (op.wshape[3],), torch.float32, self.dev)
            z = ops.empty_like(out) if (train and op.bn) else None
            outs.append(out); zs.append(z)
            ctxs.append((cur, z, out, ACT_RELU) if train else None)
            cur = out
            if train and self.winograd_wgrad:
                self._wino_V[op.name] = [(ops.empty((ops.winograd_v_floats(hshape(op, a, b)),), torch.float32, self.dev), a, b)
                                         for a, b, _ in halves]
        ev0 = _hip_mod.ev_record(main)                                   # x and the U's are ready
        lag = []
        for hi, (a, b, st) in enumerate(halves):
            with torch.cuda.stream(st):
                if st is not main:
                    _hip_mod.ev_wait(st, ev0)
                    _hip_mod.ev_wait(st, lag[0])                         # one input transform behind the first half
                inp = x[a:b]
                for li, op in enumerate(ops_):
                    V = self._wino_V[op.name][hi][0] if (train and self.winograd_wgrad) else None
                    mark = (lambda: lag.append(_hip_mod.ev_record(main))) if (st is main and li == 0) else None
                    ops.conv2d_winograd(inp, Us[li][hi], op.b, op.scale, op.shift, ACT_RELU, out=outs[li][a:b],
                                        z_out=None if zs[li] is None else zs[li][a:b], keep_v=V, after_input=mark)
                    inp = outs[li][a:b]
        _hip_mod.stream_wait(main, side)
        return outs[-1], ctxs

    def infer(self, images, windows_norm):
        """Inference graph (model.py:2133-2159).  images [B,H,W,3] float32 device tensor (molded);
        windows_norm [B,4] device.  Returns the 7 outputs of the Keras inference model."""
        cfg = self.cfg
        B, H, W = images.shape[0], images.shape[1], images.shape[2]
        area = float(H * W)
        # layers of a few hundred pixels (res4 / res5 / P5 / P6 at batch 1) take the 16 x 16-tile single-launch kernel while the
        # inference trunk is issued: detect graph 3.01 -> 2.73 ms (ResNet-101), 2.35 -> 2.23 (ResNet-50).  Training keeps its kernels.
        ops.tuning_set("sk16", 1 if self.sk16_infer else 0)
        try:
            pyr, _ = self._trunk_fwd(images, False)
            _, rpn_probs, rpn_bbox, _ = self._rpn_fwd(pyr, False)
        finally:
            ops.tuning_set("sk16", 0)
        anchors = self.anchors((H, W, images.shape[3]))
        rois = ops.proposals(rpn_probs, rpn_bbox, anchors, cfg.PRE_NMS_LIMIT, cfg.POST_NMS_ROIS_INFERENCE,
                             cfg.RPN_NMS_THRESHOLD, np.asarray(cfg.RPN_BBOX_STD_DEV, np.float32))
        _, probs, bbox, _ = self._class_head_fwd(rois, pyr[:4], area, False)
        det = ops.detections(rois, probs, bbox, windows_norm, cfg.DETECTION_MAX_INSTANCES,
                             cfg.DETECTION_MIN_CONFIDENCE, cfg.DETECTION_NMS_THRESHOLD,
                             np.asarray(cfg.BBOX_STD_DEV, np.float32))
        det_boxes = ops.empty((B, cfg.DETECTION_MAX_INSTANCES, 4), torch.float32, self.dev)
        ops.copy2d(det_boxes.data_ptr(), 16, det.data_ptr(), 24, 16, B * cfg.DETECTION_MAX_INSTANCES)
        masks, _ = self._mask_head_fwd(det_boxes, pyr[:4], area, False)
        return {"detections": det, "mrcnn_class": probs, "mrcnn_bbox": bbox, "mrcnn_mask": masks, "rpn_rois": rois,
                "rpn_class": rpn_probs, "rpn_bbox": rpn_bbox, "pyramid": pyr}

    def infer_graphed(self, images, windows_norm):
        """infer() replayed from a HIP graph (one capture per input shape): the ~450 launches of a
        batch-1 detect pass are host-bound when issued one by one.  `images` / `windows_norm` may be host
        or device tensors; they are copied into the graph's static input buffers."""
        key = (tuple(images.shape), tuple(windows_norm.shape), id(self.cfg), self.head_dtype)
        entry = self._infer_graphs.get(key)
        if self.head_dtype is not None:
            self._ensure_h16()      # outside the graph: the 16-bit weight images follow set_weights / apply_gradients in place
        if entry is None:
            sx = ops.empty(tuple(images.shape), torch.float32, self.dev)
            sw = ops.empty(tuple(windows_norm.shape), torch.float32, self.dev)
            sx.copy_(images)
            sw.copy_(windows_norm)
            side = torch.cuda.Stream(device=self.dev)
            _hip_mod.stream_wait(side, torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(2):                      # sizes every workspace / caches the anchors
                    self.infer(sx, sw)
            _hip_mod.stream_wait(torch.cuda.current_stream(self.dev), side)
            graph = torch.cuda.CUDAGraph()
            with _capture(graph):
                out = self.infer(sx, sw)
            entry = (graph, sx, sw, out)
            self._infer_graphs[key] = entry
        graph, sx, sw, out = entry
        sx.copy_(images, non_blocking=True)
        sw.copy_(windows_norm, non_blocking=True)
        graph.replay()
        return out

    # =========================================================================================
    #  training step: forward + backward (gradients left in self.grads)
    # =========================================================================================
    def loss_weights(self):
        cfg = self.cfg
        return [float(cfg.LOSS_WEIGHTS.get(n, 1.)) if cfg.USE_LOSSES.get(n, True) else 0.0 for n in LOSS_NAMES]

    def forward_backward(self, *args, **kwargs):
        """forward_backward_impl with the step arena active: every tensor the step allocates is the one the
        previous step used at the same point (ops.StepArena).  Tensors of `self.last` (keep_outputs) are
        therefore valid until the next step."""
        ops.set_arena(self.arena)
        self.arena.begin()
        try:
            return self._forward_backward(*args, **kwargs)
        finally:
            self.arena.end()

    def _forward_backward(self, images, rpn_match, rpn_bbox_t, gt_class_ids, gt_boxes_norm, gt_masks, active_class_ids,
                          rand_keys, keep_outputs=False):
        """One training forward/backward on this rank's batch.  All arguments are device tensors:
        images [B,H,W,3] f32, rpn_match [B,A,1] i32, rpn_bbox_t [B,Np,4] f32, gt_class_ids [B,G] i32,
        gt_boxes_norm [B,G,4] f32 (norm_boxes_graph applied), gt_masks [B,H,W,G] u8,
        active_class_ids [C] i32 (row 0 of the batch, model.py:1182), rand_keys [B,R] f32.
        Returns losses[5] (device)."""
        cfg = self.cfg
        B, H, W = images.shape[0], images.shape[1], images.shape[2]
        area = float(H * W)
        # gradient zeroing and the flipped / transposed weight images are only needed by the backward pass: they run on
        # the weight-gradient stream beside the forward pass (0.2 ms off the critical path)
        prep_ev = None
        if self.wgrad_stream is not None:
            ev0 = _hip_mod.ev_record(torch.cuda.current_stream(self.dev))
            h16_ev = None
            with torch.cuda.stream(self.wgrad_stream):
                _hip_mod.ev_wait(self.wgrad_stream, ev0)
                if self.head_dtype is not None:
                    self._ensure_h16()                 # 16-bit weight images: first needed by the FPN (wide mode) / after the trunk
                    h16_ev = _hip_mod.ev_record(self.wgrad_stream)
                ops.fill_zero(self.grads)
                if not self.wt_valid:
                    self.refresh_wt()
            prep_ev = _hip_mod.ev_record(self.wgrad_stream)
            if h16_ev is not None and self.h16_wide:
                _hip_mod.ev_wait(torch.cuda.current_stream(self.dev), h16_ev)
        else:
            ops.fill_zero(self.grads)
            if not self.wt_valid:
                self.refresh_wt()
        pyr, tape = self._trunk_fwd(images, True)
        rpn_logits, rpn_probs, rpn_bbox, rpn_tape = self._rpn_fwd(pyr, True)
        anchors = self.anchors((H, W, images.shape[3]))
        rpn_rois = ops.proposals(rpn_probs, rpn_bbox, anchors, cfg.PRE_NMS_LIMIT, cfg.POST_NMS_ROIS_TRAINING,
                                 cfg.RPN_NMS_THRESHOLD, np.asarray(cfg.RPN_BBOX_STD_DEV, np.float32))
        if self.forced_rpn_rois is not None:     # tests: compare two precisions of the differentiable part on one ROI set
            rpn_rois = self.forced_rpn_rois
        rois, tcls, tbbox, tmask, assign, counts = ops.detection_targets(
            rpn_rois, gt_class_ids, gt_boxes_norm, gt_masks, rand_keys, cfg.TRAIN_ROIS_PER_IMAGE,
            cfg.ROI_POSITIVE_RATIO, cfg.BBOX_STD_DEV, cfg.MASK_SHAPE, cfg.USE_MINI_MASK)
        # the two heads are independent: the (small) class/box head runs on the auxiliary stream beside the
        # mask head, forward and backward
        main, aux = torch.cuda.current_stream(self.dev), self.aux_stream
        # The mask loss only sees positive ROIs (mrcnn_mask_loss_graph, model.py:1250-1257) and
        # DetectionTargetLayer puts them first (model.py:696), at most int(T * ROI_POSITIVE_RATIO) per image:
        # the mask-head outputs of every other row are never read and their gradient rows are exactly zero.
        # With ``sparse_mask_bwd`` (default) the mask head -- forward and backward -- therefore runs on rows
        # [0, quota) of each image only: same losses, same gradients, ~1/3 of the mask-head FLOPs, no host
        # synchronisation.  ``sparse_mask_bwd = False`` computes (and multiplies the zeros of) all rows like TF.
        T = rois.shape[1]
        quota = min(T, int(T * cfg.ROI_POSITIVE_RATIO))
        sparse = self.sparse_mask_bwd and 0 < quota < T
        if sparse:
            rois_m = ops.empty((B, quota, 4), torch.float32, self.dev)
            ops.copy2d(rois_m.data_ptr(), quota * 16, rois.data_ptr(), T * 16, quota * 16, B)
        else:
            rois_m = rois
        if prep_ev is not None and self.head_dtype is not None:
            _hip_mod.ev_wait(main, prep_ev)                    # the mask head reads the 16-bit weight images prepared on the side stream
        if aux is not None:
            ev = _hip_mod.ev_record(main)
            with torch.cuda.stream(aux):
                _hip_mod.ev_wait(aux, ev)
                logits, probs, mbbox, ctx_cls = self._class_head_fwd(rois, pyr[:4], area, True)
            mmask, ctx_mask = self._mask_head_fwd(rois_m, pyr[:4], area, True)
            _hip_mod.stream_wait(main, aux)
            for t in (logits, probs, mbbox):
                t.record_stream(main)
            rois.record_stream(aux)
        else:
            logits, probs, mbbox, ctx_cls = self._class_head_fwd(rois, pyr[:4], area, True)
            mmask, ctx_mask = self._mask_head_fwd(rois_m, pyr[:4], area, True)
        if sparse:                                   # rows [quota, T) of the loss input: zeros, never read as positives
            row = int(np.prod(mmask.shape[2:])) * 4
            full = ops.empty((B, T) + tuple(mmask.shape[2:]), torch.float32, self.dev)
            ops.fill_zero(full)
            ops.copy2d(full.data_ptr(), T * row, mmask.data_ptr(), quota * row, quota * row, B)
            mmask = full
        out = ops.losses_fwd_bwd(rpn_match, rpn_bbox_t, rpn_logits, rpn_bbox, tcls, tbbox, tmask, active_class_ids,
                                 logits, mbbox, mmask, self.loss_weights(),
                                 cfg.MASK_LOSS_FUNCTION == "dice_coef_loss")
        losses, d_rpn_logits, d_rpn_bbox, d_logits, d_mbbox, d_mmask = out
        if prep_ev is not None:
            _hip_mod.ev_wait(main, prep_ev)

        # ---- pyramid gradient accumulators ------------------------------------------------------
        dP = [ops.empty_like(p) for p in pyr[:4]]
        for t in dP:
            ops.fill_zero(t)
        if aux is not None:
            ev = _hip_mod.ev_record(main)
            with torch.cuda.stream(aux):
                _hip_mod.ev_wait(aux, ev)
                self._class_head_bwd(d_logits, d_mbbox, ctx_cls, rois, dP, area)
                # the RPN backward needs nothing from the heads: it follows the class head on this stream, beside the
                # (much longer) mask head.  Its data gradients add to dP with plain read-modify-writes, so the mask head's
                # ROIAlign adjoint (atomics on the same maps, last thing on the main stream) waits for this event.
                dP6 = self._rpn_bwd(d_rpn_logits, d_rpn_bbox, rpn_tape, dP)
                ev_aux = _hip_mod.ev_record(aux)
            self._mask_head_bwd(d_mmask, ctx_mask, rois_m, dP, area, before_adjoint=ev_aux)
            _hip_mod.stream_wait(main, aux)
            for t in (d_logits, d_mbbox, d_rpn_logits, d_rpn_bbox):
                t.record_stream(aux)
        else:
            self._mask_head_bwd(d_mmask, ctx_mask, rois_m, dP, area)
            self._class_head_bwd(d_logits, d_mbbox, ctx_cls, rois, dP, area)
            dP6 = self._rpn_bwd(d_rpn_logits, d_rpn_bbox, rpn_tape, dP)
        deferred = bool(self._deferred)
        self._flush_deferred()
        if self.grad_ready and not deferred:     # ~2/3 of the gradient bytes (FC1 alone is 51 MB) are final here,
            self.join_wgrad()
            self.grad_ready(*self.grad_ranges["heads"])     # with the whole FPN/backbone backward left to hide them
        self._trunk_bwd(dP, dP6, tape)
        self.join_aux()
        if self.grad_ready and deferred:         # the mask head's kernels became final beside the backbone's backward pass
            self.grad_ready(*self.grad_ranges["heads"])
        if keep_outputs:
            self.last = {"rpn_class_logits": rpn_logits, "rpn_class": rpn_probs, "rpn_bbox": rpn_bbox,
                         "rpn_rois": rpn_rois, "rois": rois, "target_class_ids": tcls, "target_bbox": tbbox,
                         "target_mask": tmask, "mrcnn_class_logits": logits, "mrcnn_class": probs,
                         "mrcnn_bbox": mbbox, "mrcnn_mask": mmask, "counts": counts, "pyramid": pyr}
        return losses

    # ---- head backward --------------------------------------------------------------------------
    def _mask_head_bwd(self, d_mmask, ctxs, rois, dP, area, before_adjoint=None):
        """Backward of build_fpn_mask_graph on the rows the forward ran on: ``rois`` is [B, R, 4] with R = T
        (all train ROIs) or the positive quota (see forward_backward); d_mmask is always [B, T, ...]."""
        B, T, R = d_mmask.shape[0], d_mmask.shape[1], rois.shape[1]
        if R == T:
            g = d_mmask
        else:
            row = int(np.prod(d_mmask.shape[2:])) * 4
            g = ops.empty((B, R) + tuple(d_mmask.shape[2:]), torch.float32, self.dev)
            ops.copy2d(g.data_ptr(), R * row, d_mmask.data_ptr(), T * row, R * row, B)
        self._mask_head_bwd_rows(g.view((B * R,) + tuple(g.shape[2:])), ctxs, rois, dP, area, before_adjoint)

    def _mask_head_bwd_rows(self, g, ctxs, rois, dP, area, before_adjoint=None):
        if not self.h16_phase_bwd:
            ops.tuning_set("h16_phase", 0)
        try:
            self._mask_head_bwd_rows_impl(g, ctxs, rois, dP, area, before_adjoint)
        finally:
            if not self.h16_phase_bwd:
                ops.tuning_set("h16_phase", 1)

    def _mask_head_bwd_rows_impl(self, g, ctxs, rois, dP, area, before_adjoint=None):
        cfg = self.cfg
        c1, c2, c3, c4, cdec, cm = ctxs
        mop, dc = self.op("mrcnn_mask"), self.op("mrcnn_mask_deconv")
        x_in, _, up, _ = cdec
        C_ = cm[2].shape[-1]
        S = float(self.loss_scale) if self.head_dtype == torch.float16 else 1.0
        d16 = None
        if up.dtype != torch.float32:
            # output stage and deconvolution on the 16-bit tensors: dzg comes out scaled, in 16 bits
            dzg = ops.mask_out_bwd_h16(g, cm[2], up, mop.w.view(mop.wshape[2], mop.wshape[3]),
                                       mop.dw.view(mop.wshape[2], mop.wshape[3]), mop.db, dc.db, S)
            self._mask_wgrad("h16", x_in, dzg, dc.wshape, dc.dw, 1.0 / S)
            d16 = ops.conv2d_h16(dzg, self._h16[dc.name][1], (1, 1, dc.wshape[3], dc.wshape[2]), None, None, None, 1, "valid",
                                 ACT_NONE)
        elif self.fused_mask_out_bwd and C_ <= 16 and up.shape[-1] % 64 == 0 and up.shape[-1] <= 1024:
            # one pass over `up`: sigmoid', 1x1 conv wgrad/dgrad/bias, ReLU mask, deconv bias, GEMM regrouping
            # (dw accumulates: the flat gradient buffer was zeroed at the start of the step)
            dzg = ops.mask_out_bwd(g, cm[2], up, mop.w.view(mop.wshape[2], mop.wshape[3]),
                                   mop.dw.view(mop.wshape[2], mop.wshape[3]), mop.db, dc.db)
        else:
            dz, _ = mop.epilogue_bwd(g, cm)
            mop.wgrad(dz, cm)
            d_up = mop.dgrad(dz, cm)                                    # [M,28,28,256]
            # deconv: relu mask + bias, regroup to GEMM columns, then the two GEMM adjoints
            dzu = ops.empty_like(d_up)
            ops.epilogue_bwd(d_up, up, None, None, None, None, None, dzu, None, None, dc.db, ACT_RELU)
            dzg = ops.pixel_unshuffle2(dzu)                             # [M,14,14,1024]
        if d16 is None:
            self._mask_wgrad("f32", x_in, dzg, dc.wshape, 1, "valid", dc.dw, False)
            if not self.wt_valid:
                ops.weight_flip_transpose(dc.w, dc.wt)
        if self.head_dtype is None:
            # every data-gradient convolution applies the epilogue backward of the layer below in its own epilogue
            # (mrcnn_conv2d_dgrad_ep): the gradient w.r.t. the activated output is never written or re-read
            chain = [(self.op("mrcnn_mask_conv%d" % i), c) for i, c in ((4, c4), (3, c3), (2, c2), (1, c1))]
            dz = self._dgrad_ep(dzg, dc.wt, "valid", chain[0][0], chain[0][1])
            for k, (op, c) in enumerate(chain):
                Vs = self._wino_V.pop(op.name, None)
                if Vs is not None and self.winograd_wgrad:
                    for vi, (V, a, b) in enumerate(Vs):                  # one call per forward chain (whole batch, or two halves)
                        self._mask_wgrad("wino", V, (b - a,) + tuple(c[0].shape[1:]), dz[a:b], op.dw, vi > 0)
                else:
                    self._mask_wgrad("f32", *op.wgrad_item(dz, c))
                kh, kw, cin, cout = op.wshape
                wino = self._wino_ok(op, c[0].shape)
                if wino and not self.wt_valid:
                    ops.weight_flip_transpose(op.w, op.wt)
                if k + 1 < len(chain):
                    below, bctx = chain[k + 1]
                    if not self.wt_valid and not wino:
                        ops.weight_flip_transpose(op.w, op.wt)
                    if wino and self.fused_dgrad_epilogue and below.bn is not None:
                        _, bz, bout, bact = bctx
                        dz = ops.conv2d_dgrad_ep_winograd(dz, self._wino_U(op, 1, dz.shape), bout if bact != ACT_NONE else None, bz, below.scale,
                                                          below.mean, below.rstd, below.dgamma, below.dbeta, below.db, bact,
                                                          fwd_shift=below.shift)
                    elif wino:
                        dz = below.epilogue_bwd(ops.conv2d_winograd(dz, self._wino_U(op, 1, dz.shape)), bctx)[0]
                    else:
                        dz = self._dgrad_ep(dz, op.wt, ((kh - 1) // 2, (kw - 1) // 2), below, bctx)
                elif wino:
                    d = ops.conv2d_winograd(dz, self._wino_U(op, 1, dz.shape))
                else:
                    d = op.dgrad(dz, c)
        else:
            if d16 is None:
                d = ops.conv2d(dzg, dc.wt, stride=1, padding="valid")
                d16 = ops.cast_to_h16(d, self.head_dtype, multiplier=S)
            for i, c in ((4, c4), (3, c3), (2, c2), (1, c1)):
                op = self.op("mrcnn_mask_conv%d" % i)
                kh, kw, cin, cout = op.wshape
                xin, z, y, _ = c
                dz = ops.epilogue_bwd_h16(d16, y, z, op.scale, op.mean, op.rstd, op.dgamma, op.dbeta, op.db, ACT_RELU, 1.0 / S)
                self._mask_wgrad("h16", xin, dz, op.wshape, op.dw, 1.0 / S)
                d16 = ops.conv2d_h16(dz, self._h16[op.name][1], (kh, kw, cout, cin), None, None, None, 1,
                                     ((kh - 1) // 2, (kw - 1) // 2), ACT_NONE)
            d = None if self.h16_roialign else ops.cast_from_h16(d16, 1.0 / S)
        B, R = rois.shape[0], rois.shape[1]
        if before_adjoint is not None:
            _hip_mod.ev_wait(torch.cuda.current_stream(self.dev), before_adjoint)
        if self.head_dtype is not None and d is None:     # the 16-bit gradient goes straight into the float32 pyramid gradients
            ops.roialign_bwd_h16(rois, d16.view(B, R, cfg.MASK_POOL_SIZE, cfg.MASK_POOL_SIZE, -1), dP, cfg.MASK_POOL_SIZE, area, 1.0 / S)
            return
        # scatter form here: the gather form measured 2 ms slower on the positive rows (14x14 samples of ~150 overlapping
        # positives: >1000 rows on the hottest pixels) and the dense head relies on the skipping of exactly-zero rows
        ops.roialign_bwd(rois, d.view(B, R, cfg.MASK_POOL_SIZE, cfg.MASK_POOL_SIZE, -1), dP, cfg.MASK_POOL_SIZE, area)

    def _dgrad_ep(self, dz, wt, padding, below, below_ctx):
        """dz of layer `below` from the dz of the layer above: data-gradient convolution with `wt` fused with
        `below`'s epilogue backward when the layer is large enough for that kernel, the two separate calls otherwise."""
        _, z, out, act = below_ctx
        if self.fused_dgrad_epilogue and below.bn is not None:
            got = ops.conv2d_dgrad_ep(dz, wt, padding, out if act != ACT_NONE else None, z, below.scale, below.mean, below.rstd,
                                      below.dgamma, below.dbeta, below.db, act)
            if got is not None:
                return got
        d = ops.conv2d(dz, wt, stride=1, padding=padding)
        return below.epilogue_bwd(d, below_ctx)[0]

    def _class_head_bwd(self, d_logits, d_mbbox, ctxs, rois, dP, area):
        cfg = self.cfg
        c1, c2, c3, c4 = ctxs
        lo, bo = self.op("mrcnn_class_logits"), self.op("mrcnn_bbox_fc")
        M = d_logits.shape[0] * d_logits.shape[1]
        g1 = d_logits.view(M, 1, 1, -1)
        g2 = d_mbbox.view(M, 1, 1, -1)
        dz, _ = lo.epilogue_bwd(g1, c3)
        lo.wgrad(dz, c3)
        d_h2 = lo.dgrad(dz, c3)
        dz, _ = bo.epilogue_bwd(g2, c4)
        bo.wgrad(dz, c4)
        bo.dgrad(dz, c4, out=d_h2, accumulate=True)
        op2, op1 = self.op("mrcnn_class_conv2"), self.op("mrcnn_class_conv1")
        if c2[0].dtype != torch.float32:                        # the two FC layers ran in 16 bits
            S = self._S()
            d_h1 = self._h16_bwd(op2, ops.cast_to_h16(d_h2, self.head_dtype, multiplier=S), c2, S)
            d_pool = ops.cast_from_h16(self._h16_bwd(op1, d_h1, c1, S), 1.0 / S)
        else:
            dz, _ = op2.epilogue_bwd(d_h2, c2)
            op2.wgrad(dz, c2)
            d_h1 = op2.dgrad(dz, c2)
            dz, _ = op1.epilogue_bwd(d_h1, c1)
            op1.wgrad(dz, c1)
            d_pool = op1.dgrad(dz, c1)
        B, R = rois.shape[0], rois.shape[1]
        ops.roialign_bwd(rois, d_pool.view(B, R, cfg.POOL_SIZE, cfg.POOL_SIZE, -1), dP, cfg.POOL_SIZE, area,
                         dense=self.gather_roialign_bwd)

    def _rpn_bwd(self, d_logits, d_bbox, rpn_tape, dP):
        na = len(self.cfg.RPN_ANCHOR_RATIOS)
        B, A = d_logits.shape[0], d_logits.shape[1]
        shared, cls, box = self.op("rpn_conv_shared"), self.op("rpn_class_raw"), self.op("rpn_bbox_pred")
        dP6 = None
        if self.multi_launch and self.wt_valid and len(rpn_tape) == 5:
            # level by level only what is elementwise; the data-gradient convolutions of the five levels share launches
            n = len(rpn_tape)
            dzl, dzb, heads = [], [], []
            for lvl, (cs, chead, off, H, W) in enumerate(rpn_tape):
                gl = ops.empty((B, H, W, 2 * na), torch.float32, self.dev)
                gb = ops.empty((B, H, W, 4 * na), torch.float32, self.dev)
                ops.copy2d(gl.data_ptr(), H * W * 2 * na * 4, d_logits.data_ptr() + off * 2 * 4, A * 2 * 4, H * W * 2 * na * 4, B)
                ops.copy2d(gb.data_ptr(), H * W * 4 * na * 4, d_bbox.data_ptr() + off * 4 * 4, A * 4 * 4, H * W * 4 * na * 4, B)
                dz, _ = cls.epilogue_bwd(gl, chead)
                cls.wgrad(dz, chead, accumulate=lvl > 0)             # weights shared by the 5 levels
                dzl.append(dz)
                dz, _ = box.epilogue_bwd(gb, chead)
                box.wgrad(dz, chead, accumulate=lvl > 0)
                dzb.append(dz)
                heads.append(chead)
            d_s = self._dgrad_multi([cls] * n, dzl, heads)
            self._dgrad_multi([box] * n, dzb, heads, outs=d_s, accumulate=True)
            if self._h16_layer("rpn_conv_shared") and rpn_tape[0][0][0].dtype != torch.float32:
                S = self._S()
                for lvl, (cs, chead, off, H, W) in enumerate(rpn_tape):
                    dx16 = self._h16_bwd(shared, ops.cast_to_h16(d_s[lvl], self.head_dtype, multiplier=S), cs, S, accumulate_w=lvl > 0)
                    if lvl < 4:
                        ops.axpy_from_h16(dx16, dP[lvl], 1.0 / S)       # P2..P5 also feed the ROI heads: add to their gradients
                    else:
                        dP6 = ops.cast_from_h16(dx16, 1.0 / S)
                return dP6
            dzs, ctxs = [], []
            wino = any((shared.name, tuple(t[0][0].shape)) in self._wino_V for t in rpn_tape)
            for lvl, (cs, chead, off, H, W) in enumerate(rpn_tape):
                dz, _ = shared.epilogue_bwd(d_s[lvl], cs)
                if not wino:
                    shared.wgrad(dz, cs, accumulate=lvl > 0)
                dzs.append(dz); ctxs.append(cs)
            if wino:                                        # some level's forward took the Winograd path (trunk_winograd)
                return self._bwd_group([shared] * n, dzs, ctxs, outs=list(dP[:4]) + [None], accumulate=[True] * 4 + [False],
                                       wgrad_acc=[lvl > 0 for lvl in range(n)])[4]
            return self._dgrad_multi([shared] * n, dzs, ctxs, outs=list(dP[:4]) + [None], accumulate=[True] * 4 + [False])[4]
        for lvl, (cs, chead, off, H, W) in enumerate(rpn_tape):
            gl = ops.empty((B, H, W, 2 * na), torch.float32, self.dev)
            gb = ops.empty((B, H, W, 4 * na), torch.float32, self.dev)
            ops.copy2d(gl.data_ptr(), H * W * 2 * na * 4, d_logits.data_ptr() + off * 2 * 4, A * 2 * 4, H * W * 2 * na * 4, B)
            ops.copy2d(gb.data_ptr(), H * W * 4 * na * 4, d_bbox.data_ptr() + off * 4 * 4, A * 4 * 4, H * W * 4 * na * 4, B)
            acc = lvl > 0                                            # weights shared by the 5 levels
            dz, _ = cls.epilogue_bwd(gl, chead)
            cls.wgrad(dz, chead, accumulate=acc)
            d_s = cls.dgrad(dz, chead)
            dz, _ = box.epilogue_bwd(gb, chead)
            box.wgrad(dz, chead, accumulate=acc)
            box.dgrad(dz, chead, out=d_s, accumulate=True)
            dz, _ = shared.epilogue_bwd(d_s, cs)
            shared.wgrad(dz, cs, accumulate=acc)
            if lvl < 4:
                shared.dgrad(dz, cs, out=dP[lvl], accumulate=True)
            else:
                dP6 = shared.dgrad(dz, cs)
        return dP6

    # ---- FPN + backbone backward ------------------------------------------------------------------
    def _trunk_bwd(self, dP, dP6, tape):
        dP2, dP3, dP4, dP5 = dP
        ops.subsample2_bwd_acc(dP6, dP5)
        layers, dzs, cs = [], [], []
        if tape["fpn_p5"][0].dtype != torch.float32:                 # smoothing convolutions ran in 16 bits
            S = self._S()
            d16 = [self._h16_bwd(self.op(name), ops.cast_to_h16(g, self.head_dtype, multiplier=S), tape[name], S)
                   for name, g in (("fpn_p5", dP5), ("fpn_p4", dP4), ("fpn_p3", dP3), ("fpn_p2", dP2))]
            d5s, d4s, d3s, d2s = [ops.cast_from_h16(t, 1.0 / S) for t in d16]
        else:
            for name, g in (("fpn_p5", dP5), ("fpn_p4", dP4), ("fpn_p3", dP3), ("fpn_p2", dP2)):
                op, c = self.op(name), tape[name]
                dz, _ = op.epilogue_bwd(g, c)
                layers.append(op); dzs.append(dz); cs.append(c)
            if any((op.name, tuple(c[0].shape)) in self._wino_V for op, c in zip(layers, cs)):
                d5s, d4s, d3s, d2s = self._bwd_group(layers, dzs, cs)
            else:
                self.wgrad_group([op.wgrad_item(dz, c) for op, dz, c in zip(layers, dzs, cs)])
                d5s, d4s, d3s, d2s = self._dgrad_multi(layers, dzs, cs)    # four independent 3x3 data gradients: one launch
        ops.upsample2_bwd(d2s, d3s, True)
        ops.upsample2_bwd(d3s, d4s, True)
        ops.upsample2_bwd(d4s, d5s, True)
        layers, dzs, cs = [], [], []
        for name, g in (("fpn_c2p2", d2s), ("fpn_c3p3", d3s), ("fpn_c4p4", d4s), ("fpn_c5p5", d5s)):
            op, c = self.op(name), tape[name]
            dz, _ = op.epilogue_bwd(g, c)
            layers.append(op); dzs.append(dz); cs.append(c)
        self.wgrad_group([op.wgrad_item(dz, c) for op, dz, c in zip(layers, dzs, cs)])
        dC = self._dgrad_multi(layers, dzs, cs)                         # and the four lateral 1x1 ones
        if self.grad_ready:
            self.join_wgrad()
            self.grad_ready(*self.grad_ranges["tail"])
        d_out = dC[3]                                   # gradient w.r.t. C5
        d16 = None                                      # the running gradient in 16 bits (times S), between 16-bit blocks
        S = self._S()
        for si in (3, 2, 1, 0):
            stage = self.stages[si]
            for bi in range(len(stage) - 1, -1, -1):
                blk = stage[bi]
                if tape[id(blk)][0] == "h16":
                    if d16 is None:
                        d16 = ops.cast_to_h16(d_out, self.head_dtype, multiplier=S)
                    d16 = self._block_bwd_h16(blk, d16, tape[id(blk)], S)
                    d_out = None
                    if bi == 0 and si > 0:          # first block of a stage: its input C_(s-1) also feeds the FPN lateral
                        d_out, d16 = ops.axpy_from_h16(d16, dC[si - 1], 1.0 / S), None
                    continue
                if d_out is None:
                    d_out, d16 = ops.cast_from_h16(d16, 1.0 / S), None
                # the first block of stage s+1 consumes C_s, whose FPN gradient is already in dC[si-1]
                acc_buf = dC[si - 1] if (bi == 0 and si > 0) else None
                # the block before this one in the forward order: its output epilogue backward (ReLU mask, BN of its 2c
                # convolution) rides on this block's last data gradient when that is a split-K layer
                below = None
                if bi > 0:
                    below = (stage[bi - 1], tape[id(stage[bi - 1])][2])
                elif si > 0:
                    prev = self.stages[si - 1][-1]
                    below = (prev, tape[id(prev)][2])
                if below is not None and tape[id(below[0])][0] == "h16":
                    below = None                        # the block below runs in 16 bits: it differentiates its own output stage
                d_out = self._block_bwd(blk, d_out, tape[id(blk)], acc_buf, below)
            if self.grad_ready:
                self.join_wgrad()
                self.grad_ready(*self.grad_ranges[si + 2])
        assert not isinstance(d_out, tuple)             # the first block of the network has no block below it
        if d_out is None:
            d_out = ops.cast_from_h16(d16, 1.0 / S)
        am, pre_shape = tape["pool"]
        d_relu = ops.maxpool3x3s2_bwd(d_out, am, pre_shape)
        c1 = self.op("conv1")
        dz, _ = c1.epilogue_bwd(d_relu, tape["conv1"])
        c1.wgrad(dz, tape["conv1"])
        self.join_wgrad()
        if self.grad_ready:
            self.grad_ready(*self.grad_ranges["head"])
            self.grad_ready(*self.grad_ranges["bn"])

    def _block_bwd(self, blk, d_out, ctxs, acc_buf, below=None):
        """Backward of one bottleneck block.  ``d_out``: gradient w.r.t. the block output, or the pair (dz of the 2c
        convolution, gradient after the output ReLU) when the block after it already applied this block's output
        epilogue backward (see ``below``).  Returns the gradient w.r.t. the block input, or such a pair for ``below``."""
        ca, cb, cc, c1c = ctxs
        if isinstance(d_out, tuple):
            dzc, dy = d_out
        else:
            dzc, dy = blk.c2c.epilogue_bwd(d_out, cc, want_dy=True)
        # 2c -> 2b -> 2a: each data-gradient convolution (or its split-K reduction) applies the epilogue backward of
        # the layer below (mrcnn_conv2d_dgrad_ep)
        if not self.wt_valid:
            ops.weight_flip_transpose(blk.c2c.w, blk.c2c.wt)
            ops.weight_flip_transpose(blk.c2b.w, blk.c2b.wt)
        dzb = self._dgrad_ep(dzc, blk.c2c.wt, "valid", blk.c2b, cb)
        dza = self._dgrad_ep(dzb, blk.c2b.wt, (1, 1), blk.c2a, ca)
        # the block's weight gradients travel together (they trail the data gradients on their stream anyway)
        items = [blk.c2c.wgrad_item(dzc, cc), blk.c2b.wgrad_item(dzb, cb), blk.c2a.wgrad_item(dza, ca)]
        if blk.c1 is None:
            self.wgrad_group(items)
            # identity shortcut: dx = dy + dgrad_2a
            fused = self._final_dgrad_fused(blk, dza, dy, below)
            return fused if fused is not None else blk.c2a.dgrad(dza, ca, out=dy, accumulate=True)   # in place on dy
        dz1, _ = blk.c1.epilogue_bwd(dy, c1c)
        self.wgrad_group(items + [blk.c1.wgrad_item(dz1, c1c)])
        if acc_buf is not None:
            dx = blk.c1.dgrad(dz1, c1c, out=acc_buf, accumulate=True)
        else:
            dx = blk.c1.dgrad(dz1, c1c)
        fused = self._final_dgrad_fused(blk, dza, dx, below)
        return fused if fused is not None else blk.c2a.dgrad(dza, ca, out=dx, accumulate=True)

    def _final_dgrad_fused(self, blk, dza, acc, below):
        """dx = dgrad_2a(dza) + acc with the output epilogue backward of the block below fused into the split-K reduction:
        returns (dz of below's 2c convolution, masked gradient for below's shortcut), or None when that data gradient is not
        a dense stride-1 split-K layer (the caller then takes the two-launch route)."""
        if below is None or not self.fused_dgrad_epilogue or not self.wt_valid or blk.c2a.stride != 1:
            return None
        pblk, pcc = below
        op = pblk.c2c
        _, z, out, act = pcc
        if op.bn is None or act != ACT_RELU:
            return None
        dy_below = ops.empty_like(acc)
        got = ops.conv2d_dgrad_ep(dza, blk.c2a.wt, "valid", out, z, op.scale, op.mean, op.rstd, op.dgamma, op.dbeta, op.db, act,
                                  res=acc, dy_out=dy_below)
        return None if got is None else (got, dy_below)

    # =========================================================================================
    #  the whole training step as one HIP graph
    # =========================================================================================
    _MODE_ATTRS = ("sparse_mask_bwd", "h16_wide", "h16_blocks", "h16_all_blocks", "h16_fused_bwd", "h16_phase_bwd", "winograd",
                   "winograd_wgrad", "winograd_split", "fused_mask_out_bwd", "fused_dgrad_epilogue", "defer_mask_wgrad",
                   "gather_roialign_bwd", "multi_launch", "h16_roialign", "trunk_winograd", "trunk_winograd_min_rows")

    def _mode_key(self):
        """Every engine switch a captured graph / recorded launch tape bakes in besides the tensors: a replay is only valid for the
        mode it was made in (tests and tools flip these attributes between steps of one engine)."""
        # ... and what a launch bakes in by VALUE: the loss weights and the clip norm are scalar arguments of recorded calls
        # (cfg is mutable: run.py patches it in place).  The trainable mask / weight decay live in gran_coef, a buffer with a
        # stable address that set_trainable refreshes in place, so recordings survive a change of `layers`.
        return tuple(getattr(self, a, None) for a in self._MODE_ATTRS) + (
            float(self.loss_scale), self.forced_rpn_rois is not None, tuple(self.loss_weights()),
            float(self.cfg.GRADIENT_CLIP_NORM))

    def step_graphed(self, dev_inputs, learning_rate, momentum):
        """forward_backward + apply_gradients (single rank) replayed from a HIP graph: one capture per input signature,
        learning rate / momentum and engine mode.  A step is ~670 launches in float32 and ~930 with the 16-bit blocks; issued
        one by one they cost 15-21 ms of host time, more than the device needs in the 16-bit modes.  The three streams of the
        step (main, weight gradients, auxiliary) fork from and re-join the capturing stream, so the graph keeps their
        concurrency.  Inputs are copied into the graph's static buffers; the returned losses tensor is the graph's static
        output (valid until the next replay).  Nothing on this path may use memset / memcpy nodes."""
        key = (tuple((tuple(t.shape), t.dtype) for t in dev_inputs), float(learning_rate), float(momentum), self.head_dtype,
               id(self.cfg), self._mode_key())
        entry = self._train_graphs.get(key)
        main = torch.cuda.current_stream(self.dev)
        if entry is None:
            assert self.grad_ready is None, "graph replay is single-rank: gradient hooks cannot be captured"
            static = [torch.empty_like(t) for t in dev_inputs]
            for s_, t in zip(static, dev_inputs):
                s_.copy_(t)
            # two eager steps size every workspace, arena slot and weight image; they must not count as training steps
            keep = (self.params.clone(), self.momentum.clone(), self.skipped_steps.clone())
            side = torch.cuda.Stream(device=self.dev)
            _hip_mod.stream_wait(side, main)
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.forward_backward(*static)
                    self.apply_gradients(learning_rate, momentum, 1)
                self.params.copy_(keep[0])
                self.momentum.copy_(keep[1])
                self.skipped_steps.copy_(keep[2])
                self.wt_valid = self._h16_valid = False
                self.fold_bn()
            _hip_mod.stream_wait(main, side)
            del keep
            graph = torch.cuda.CUDAGraph()
            with _capture(graph):
                losses = self.forward_backward(*static)
                self.apply_gradients(learning_rate, momentum, 1)
            entry = (graph, static, losses)
            self._train_graphs[key] = entry
        graph, static, losses = entry
        for s_, t in zip(static, dev_inputs):
            if s_.data_ptr() != t.data_ptr():
                s_.copy_(t, non_blocking=True)
        graph.replay()
        # the captured step ends with the optimiser: the flipped / 16-bit weight images are stale for eager callers
        self.wt_valid = self._h16_valid = False
        return losses

    def step_taped(self, dev_inputs, learning_rate, momentum, world_size=1, reducer=None):
        """forward_backward + apply_gradients (single rank) from a recorded launch tape (_hip.tape_*): the first call per
        input signature / rates / mode runs the step through the engine while every launch -- C-ABI calls with their
        descriptors and pointers, event records / waits between the three streams -- is written down; later calls copy the
        inputs into the recording's buffers and issue the same calls again.  Same launches on the same streams in the same
        order as the eager step (so, unlike the HIP-graph replay, the same overlap), without the engine's Python per launch.
        Valid because the step arena, the workspaces and the weight images have stable addresses.
        Data parallelism (`reducer`: parallel.GradReducer, self.grad_ready = reducer.ready): the gradient hooks are part of the
        recording -- event hand-offs to the exchange stream, the mrcnn_allreduce_grad calls and the final join -- so a
        data-parallel step is re-issued like a single-rank one.  The settle pass runs WITHOUT the hooks (local, rolled
        back): every call of this function then enters exactly one set of collectives whether it records or replays, so ranks
        that record in different steps cannot desynchronise."""
        key = (tuple((tuple(t.shape), t.dtype) for t in dev_inputs), float(learning_rate), float(momentum), self.head_dtype,
               id(self.cfg), torch.cuda.current_stream(self.dev).cuda_stream, self._mode_key(), int(world_size),
               None if reducer is None else id(reducer))
        entry = self._train_tapes.get(key)
        if entry is None:
            assert (self.grad_ready is None) == (reducer is None), "gradient hooks are recorded through their GradReducer"
            static = [torch.empty_like(t) for t in dev_inputs]
            for s_, t in zip(static, dev_inputs):
                s_.copy_(t)
            # one eager step settles arena slots, workspaces and weight images (rolled back: it is not a training step)
            keep = (self.params.clone(), self.momentum.clone(), self.skipped_steps.clone())
            hook, self.grad_ready = self.grad_ready, None
            try:
                self.forward_backward(*static)
            finally:
                self.grad_ready = hook
            self.apply_gradients(learning_rate, momentum, 1)
            self.params.copy_(keep[0])
            self.momentum.copy_(keep[1])
            self.skipped_steps.copy_(keep[2])               # a settle pass that overflowed in float16 is not a skipped training step
            self.wt_valid = self._h16_valid = False
            self.fold_bn()
            del keep
            _hip_mod.tape_begin()
            try:
                losses = self.forward_backward(*static)
                if reducer is not None:
                    reducer.finish()
                self.apply_gradients(learning_rate, momentum, world_size)
            finally:
                tape = _hip_mod.tape_end()
            # the recorded pointers are arena slots and workspaces: hold them, so a later step of another shape or mode
            # (which replaces slots) cannot hand their memory to someone else while this tape is alive
            self._train_tapes[key] = (tape, static, losses, list(self.arena.slots), list(ops._ws_cache.values()), reducer)
            return losses                                   # the recording pass was this step
        tape, static, losses = entry[:3]
        for s_, t in zip(static, dev_inputs):
            if s_.data_ptr() != t.data_ptr():
                s_.copy_(t, non_blocking=True)
        _hip_mod.tape_replay(tape)
        self.wt_valid = self._h16_valid = False
        return losses

    # =========================================================================================
    #  optimiser (MaskRCNN.compile, model.py:2255-2291)
    # =========================================================================================
    def skipped_step_count(self):
        """Steps whose float16 gradients were not finite (skipped on the device).  Synchronises: call it where the host
        reads the losses anyway."""
        return int(self.skipped_steps.item())

    def adapt_loss_scale(self, floor=1.0):
        """Dynamic part of the loss scaling, done wherever the host looks (MaskRCNN.train: once per epoch): halve the scale
        once for every step skipped since the last look.  Returns the number of newly skipped steps."""
        n = self.skipped_step_count()
        new = n - self._skipped_seen
        self._skipped_seen = n
        if new > 0:
            self.loss_scale = max(float(floor), self.loss_scale / (2.0 ** min(new, 8)))
        return new

    def apply_gradients(self, learning_rate, momentum, world_size=1):
        """grads (already summed over ranks) -> /world, + L2 term, global-norm clip, SGD-momentum."""
        cfg = self.cfg
        ops.grad_prepare(self.grads, self.params, 1.0 / world_size, self.gran_coef, self.sumsq)
        # float16 mode: a gradient that overflowed under the static loss scale must not reach the weights -- the guarded form
        # skips the update on the device and counts it (skipped_step_count / adapt_loss_scale); float32 and bfloat16 keep the
        # reference's plain Keras semantics
        guard = self.skipped_steps if self.head_dtype == torch.float16 else None
        ops.sgd_momentum(self.params, self.momentum, self.grads, self.sumsq, cfg.GRADIENT_CLIP_NORM, learning_rate,
                         momentum, self.gran_coef, skipped=guard)
        self.wt_valid = False
        self._h16_valid = False
        self.fold_bn()
