"""Thin host-side operator layer over the C-ABI (one function per kernel group).

Every function takes/returns torch CUDA tensors that only serve as device-memory handles; the
arithmetic happens in libmrcnn_hip.so.  Shapes follow the reference graph (NHWC float32).
"""
import ctypes as C
import math
import os

import torch

from . import _hip
from ._hip import (ACT_NONE, ACT_RELU, ACT_SIGMOID, OUT_DECONV2, OUT_NHWC, RES_NONE, RES_SAME, RES_UP2,
                   check, current_stream, ptr)

BN_EPS = 1e-3  # Keras BatchNormalization default epsilon [3P]


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _hip.HipPathError("hot-path operators need device tensors (got a CPU tensor)")
        if t is not None and not t.is_contiguous():
            raise _hip.HipPathError("hot-path operators need contiguous tensors")


def same_padding(size, k, stride):
    """TF 'SAME': out = ceil(size/stride); pad_before = total//2 (asymmetric when total is odd)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2


def conv_desc(x_shape, w_shape, stride=1, padding="same", act=ACT_NONE, res_mode=RES_NONE):
    N, H, W, Cin = x_shape
    KH, KW, wcin, Cout = w_shape
    assert wcin == Cin, (x_shape, w_shape)
    d = _hip.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = N, H, W, Cin, Cout, KH, KW, stride
    if padding == "same":
        d.OH, d.pad_t = same_padding(H, KH, stride)
        d.OW, d.pad_l = same_padding(W, KW, stride)
    elif padding == "valid":
        d.OH, d.OW = (H - KH) // stride + 1, (W - KW) // stride + 1
        d.pad_t = d.pad_l = 0
    else:  # explicit symmetric zero padding (ZeroPadding2D + valid conv)
        ph, pw = padding
        d.OH, d.OW = (H + 2 * ph - KH) // stride + 1, (W + 2 * pw - KW) // stride + 1
        d.pad_t, d.pad_l = ph, pw
    d.act, d.res_mode, d.out_mode, d.cmod = act, res_mode, OUT_NHWC, Cout
    d.out_w_stride = Cout
    d.out_h_stride = d.OW * Cout
    d.out_n_stride = d.OH * d.OW * Cout
    return d


def conv2d(x, w, bias=None, scale=None, shift=None, res=None, stride=1, padding="same", act=ACT_NONE,
           res_mode=RES_NONE, out=None, z_out=None, desc=None):
    """out = act(bn(conv(x, w) + bias) + res).  w is HWIO.  Returns out [N, OH, OW, Cout]."""
    _need_cuda(x, w, bias, scale, shift, res, out, z_out)
    d = desc or conv_desc(tuple(x.shape), tuple(w.shape), stride, padding, act, res_mode)
    if out is None:
        out = empty((d.N, d.OH, d.OW, d.Cout), torch.float32, x.device)
    nbytes = _hip.lib().mrcnn_conv2d_fwd_workspace(C.byref(d))
    ws = workspace(nbytes, x.device, "conv_splitk") if nbytes else None
    check(_hip.lib().mrcnn_conv2d_fwd_ws(C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(scale), ptr(shift), ptr(res),
                                         ptr(out), ptr(z_out), ptr(ws), ws.numel() if ws is not None else 0,
                                         current_stream()), "mrcnn_conv2d_fwd")
    return out


ERR_UNSUPPORTED = -4


def conv2d_dgrad_ep(dz, w_t, padding, below_out, below_z, scale, mean, rstd, dgamma, dbeta, dbias, act, res=None, out=None,
                    dy_out=None):
    """Data-gradient convolution (stride 1) fused with the epilogue backward of the layer below; returns dz_below, or
    None when the layer is too small for the fused kernel (the caller then uses conv2d + epilogue_bwd)."""
    _need_cuda(dz, w_t, below_out, below_z, scale, mean, rstd, dgamma, dbeta, dbias, res, out, dy_out)
    d = conv_desc(tuple(dz.shape), tuple(w_t.shape), 1, padding, ACT_NONE, RES_SAME if res is not None else RES_NONE)
    ep = _hip.BwdEpilogue()
    ep.out, ep.z, ep.scale, ep.mean, ep.rstd = ptr(below_out), ptr(below_z), ptr(scale), ptr(mean), ptr(rstd)
    ep.dgamma, ep.dbeta, ep.dbias, ep.act = ptr(dgamma), ptr(dbeta), ptr(dbias), act
    ep.dy = ptr(dy_out)                                # optional: y * act' (gradient a residual branch below receives)
    if out is None:
        out = empty((d.N, d.OH, d.OW, d.Cout), torch.float32, dz.device)
    nbytes = _hip.lib().mrcnn_conv2d_fwd_workspace(C.byref(d))
    ws = workspace(nbytes, dz.device, "conv_splitk") if nbytes else None
    rc = _hip.lib().mrcnn_conv2d_dgrad_ep(C.byref(d), ptr(dz), ptr(w_t), ptr(res), ptr(out), C.byref(ep), ptr(ws),
                                          ws.numel() if ws is not None else 0, current_stream())
    if rc == ERR_UNSUPPORTED:
        return None
    check(rc, "mrcnn_conv2d_dgrad_ep")
    return out


def conv2d_into(x, w, bias, out_view_ptr, n_stride, h_stride, w_stride, stride=1, padding="same", act=ACT_NONE):
    """Conv whose output rows land inside a larger buffer (e.g. one pyramid level of the concatenated
    RPN outputs).  out_view_ptr is the device address of element (0,0,0,0)."""
    _need_cuda(x, w, bias)
    d = conv_desc(tuple(x.shape), tuple(w.shape), stride, padding, act)
    d.out_n_stride, d.out_h_stride, d.out_w_stride = n_stride, h_stride, w_stride
    check(_hip.lib().mrcnn_conv2d_fwd(C.byref(d), ptr(x), ptr(w), ptr(bias), None, None, None, out_view_ptr, None,
                                      current_stream()), "mrcnn_conv2d_fwd(into)")


def _conv2d_problem(pr):
    """One problem of conv2d_multi as an ordinary launch."""
    if "out_ptr" in pr:
        x, w = pr["x"], pr["w"]
        _need_cuda(x, w, pr.get("bias"))
        d = conv_desc(tuple(x.shape), tuple(w.shape), pr.get("stride", 1), pr.get("padding", "same"), pr.get("act", ACT_NONE))
        d.out_n_stride, d.out_h_stride, d.out_w_stride = pr["out_strides"]
        nbytes = _hip.lib().mrcnn_conv2d_fwd_workspace(C.byref(d))
        ws = workspace(nbytes, x.device, "conv_splitk") if nbytes else None
        check(_hip.lib().mrcnn_conv2d_fwd_ws(C.byref(d), ptr(x), ptr(w), ptr(pr.get("bias")), ptr(pr.get("scale")),
                                             ptr(pr.get("shift")), None, pr["out_ptr"], None, ptr(ws),
                                             ws.numel() if ws is not None else 0, current_stream()), "mrcnn_conv2d_fwd(into)")
        return None
    return conv2d(pr["x"], pr["w"], pr.get("bias"), pr.get("scale"), pr.get("shift"), pr.get("res"), pr.get("stride", 1),
                  pr.get("padding", "same"), pr.get("act", ACT_NONE), pr.get("res_mode", RES_NONE), out=pr.get("out"),
                  z_out=pr.get("z_out"))


def conv2d_multi(problems):
    """Up to 5 independent convolutions in one launch (mrcnn_conv2d_fwd_multi).  Each problem is a dict:
    x, w, and optionally bias, scale, shift, res, res_mode, stride, padding, act, z_out, and either `out` (a dense
    tensor, allocated when missing) or `out_ptr` + `out_strides` (n, h, w strides in floats, as conv2d_into).
    Returns the list of `out` tensors (None where out_ptr was given), or None when the problems do not share a
    launch shape (the caller then issues them one by one)."""
    # problems large enough for the 128x128 LDS-DMA kernel (>= 640 tiles) fill the chip on their own and run faster
    # there than on the 64-row tiles of the shared launch: they go out individually, the small rest travels together
    def big(pr):
        x, w = pr["x"], pr["w"]
        d = conv_desc(tuple(x.shape), tuple(w.shape), pr.get("stride", 1), pr.get("padding", "same"))
        return d.Cout % 128 == 0 and d.Cin % 32 == 0 and ((d.N * d.OH * d.OW + 127) // 128) * (d.Cout // 128) >= 640
    flags = [big(pr) for pr in problems]
    if any(flags):
        small = [pr for pr, f in zip(problems, flags) if not f]
        so = conv2d_multi(small) if len(small) > 1 else None
        if so is None:
            so = [_conv2d_problem(pr) for pr in small]
        small_outs = iter(so)
        return [_conv2d_problem(pr) if f else next(small_outs) for pr, f in zip(problems, flags)]
    n = len(problems)
    arr = (_hip.ConvProblem * n)()
    outs = []
    for q, pr in zip(arr, problems):
        x, w = pr["x"], pr["w"]
        _need_cuda(x, w, pr.get("bias"), pr.get("scale"), pr.get("shift"), pr.get("res"), pr.get("out"), pr.get("z_out"))
        d = conv_desc(tuple(x.shape), tuple(w.shape), pr.get("stride", 1), pr.get("padding", "same"), pr.get("act", ACT_NONE),
                      pr.get("res_mode", RES_NONE))
        out = pr.get("out")
        if "out_ptr" in pr:
            d.out_n_stride, d.out_h_stride, d.out_w_stride = pr["out_strides"]
            q.out = pr["out_ptr"]
            out = None
        else:
            if out is None:
                out = empty((d.N, d.OH, d.OW, d.Cout), torch.float32, x.device)
            q.out = ptr(out)
        outs.append(out)
        q.d = d
        q.x, q.w, q.bias, q.scale, q.shift = ptr(x), ptr(w), ptr(pr.get("bias")), ptr(pr.get("scale")), ptr(pr.get("shift"))
        q.res, q.z_out = ptr(pr.get("res")), ptr(pr.get("z_out"))
    nbytes = _hip.lib().mrcnn_conv2d_fwd_multi_workspace(arr, n)
    ws = workspace(nbytes, problems[0]["x"].device, "conv_splitk") if nbytes else None
    rc = _hip.lib().mrcnn_conv2d_fwd_multi(arr, n, ptr(ws), ws.numel() if ws is not None else 0, current_stream())
    if rc == ERR_UNSUPPORTED:
        return None
    check(rc, "mrcnn_conv2d_fwd_multi")
    return outs


_DECONV_GEMM = os.environ.get("MRCNN_DECONV_GEMM", "1") != "0"          # transposed convolution on the persistent GEMM (A/B switch)
_DECONV_GEMM_MIN_ROWS = int(os.environ.get("MRCNN_DECONV_GEMM_MIN_ROWS", "16384"))


def deconv2x2(x, w_gemm, bias, act=ACT_RELU, out=None):
    """Conv2DTranspose(2x2, stride 2): x [N,H,W,Cin], w_gemm [Cin, 4*Cd] with column (a*2+b)*Cd+co."""
    _need_cuda(x, w_gemm, bias, out)
    N, H, W, Cin = x.shape
    Cd = w_gemm.shape[1] // 4
    if out is None:
        out = empty((N, 2 * H, 2 * W, Cd), torch.float32, x.device)
    if (_DECONV_GEMM and Cin % 16 == 0 and Cd % 32 == 0 and (4 * Cd) % 128 == 0 and act in (ACT_NONE, ACT_RELU) and
            N * H * W * Cin * 4 < 0x7FFFFFF0 and N * H * W >= _DECONV_GEMM_MIN_ROWS):
        # one filter tap of K: the persistent GEMM with the pixel-shuffle store in its epilogue
        check(_hip.lib().mrcnn_deconv2x2_gemm(ptr(x), ptr(w_gemm), ptr(bias), ptr(out), N, H, W, Cin, Cd, act, current_stream()),
              "mrcnn_deconv2x2_gemm")
        return out
    d = _hip.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad_t, d.pad_l = N, H, W, Cin, 4 * Cd, 1, 1, 1, 0, 0
    d.OH, d.OW, d.act, d.res_mode, d.out_mode, d.cmod = H, W, act, RES_NONE, OUT_DECONV2, Cd
    d.out_w_stride = Cd
    d.out_h_stride = 2 * W * Cd
    d.out_n_stride = 4 * H * W * Cd
    check(_hip.lib().mrcnn_conv2d_fwd(C.byref(d), ptr(x), ptr(w_gemm), ptr(bias), None, None, None, ptr(out), None,
                                      current_stream()), "mrcnn_conv2d_fwd(deconv)")
    return out


_ws_cache = {}
_ws_retired = []


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, tag) -- nothing is allocated inside the C-ABI calls."""
    key = (str(device), tag, current_stream())        # one scratch buffer per stream: concurrent branches never share
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)     # captured HIP graphs may still point at it: never hand it back
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


class StepArena(object):
    """Reuses the tensors one training step allocates.  A step requests its activations, gradients and
    scratch in a fixed order with fixed shapes, so the k-th request of a step gets the k-th tensor of the
    previous step (a shape change simply replaces that slot).  Nothing is handed out twice within a step,
    and every side stream is joined before a step ends, so the reuse is race free.  Saves the ~20 ms/step the
    caching allocator costs for ~2 500 requests."""

    def __init__(self):
        self.slots, self.pos, self.active = [], 0, False

    def begin(self):
        self.pos, self.active = 0, True

    def end(self):
        self.active = False

    def take(self, shape, dtype, device):
        k = self.pos
        self.pos = k + 1
        if k < len(self.slots):
            t = self.slots[k]
            if t.shape == shape and t.dtype == dtype:
                return t
            t = torch.empty(shape, dtype=dtype, device=device)
            self.slots[k] = t
            return t
        t = torch.empty(shape, dtype=dtype, device=device)
        self.slots.append(t)
        return t


_arena = None


def set_arena(arena):
    global _arena
    _arena = arena


def empty(shape, dtype, device):
    if _arena is not None and _arena.active:
        return _arena.take(tuple(shape), dtype, device)
    return torch.empty(tuple(shape), dtype=dtype, device=device)


def empty_like(t):
    return empty(t.shape, t.dtype, t.device)


def conv2d_wgrad(x, dy, w_shape, stride=1, padding="same", dw=None, accumulate=False, desc=None):
    _need_cuda(x, dy, dw)
    d = desc or conv_desc(tuple(x.shape), tuple(w_shape), stride, padding)
    if dw is None:
        dw = empty(tuple(w_shape), torch.float32, x.device)
    nbytes = _hip.lib().mrcnn_conv2d_wgrad_workspace(C.byref(d))
    ws = workspace(nbytes, x.device, "wgrad")
    check(_hip.lib().mrcnn_conv2d_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(ws), ws.numel(),
                                        1 if accumulate else 0, current_stream()), "mrcnn_conv2d_wgrad")
    return dw


def conv2d_wgrad_multi(items):
    """The weight gradients of up to 4 layers in one launch (mrcnn_conv2d_wgrad_multi).  items: [(x, dy, w_shape, stride,
    padding, dw, accumulate)].  Returns False (nothing launched) when a layer does not fit the shared kernel."""
    n = len(items)
    arr = (_hip.WgradProblem * n)()
    for q, (x, dy, w_shape, stride, padding, dw, accumulate) in zip(arr, items):
        _need_cuda(x, dy, dw)
        q.d = conv_desc(tuple(x.shape), tuple(w_shape), stride, padding)
        q.x, q.dy, q.dw, q.accumulate = ptr(x), ptr(dy), ptr(dw), 1 if accumulate else 0
    nbytes = _hip.lib().mrcnn_conv2d_wgrad_multi_workspace(arr, n)
    if nbytes == 0:
        return False
    ws = workspace(nbytes, items[0][0].device, "wgrad")
    rc = _hip.lib().mrcnn_conv2d_wgrad_multi(arr, n, ptr(ws), ws.numel(), current_stream())
    if rc == ERR_UNSUPPORTED:
        return False
    check(rc, "mrcnn_conv2d_wgrad_multi")
    return True


def flip_table(entries, device):
    """entries: [(offset_floats, KH, KW, Cin, Cout)] -> (device table, n_layers, total_tiles) for
    weight_flip_transpose_batched."""
    import numpy as np
    rec = np.zeros(len(entries), dtype=[("off", "<i8"), ("KH", "<i4"), ("KW", "<i4"), ("Cin", "<i4"), ("Cout", "<i4"),
                                        ("first", "<i4"), ("pad", "<i4")])
    tiles = 0
    for i, (off, kh, kw, cin, cout) in enumerate(entries):
        rec[i] = (off, kh, kw, cin, cout, tiles, 0)
        tiles += kh * kw * ((cin + 31) // 32) * ((cout + 31) // 32)
    t = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
    return t, len(entries), tiles


def weight_flip_transpose_batched(params, params_t, table):
    _need_cuda(params, params_t, table[0])
    check(_hip.lib().mrcnn_weight_flip_transpose_batched(ptr(params), ptr(params_t), ptr(table[0]), table[1], table[2],
                                                         current_stream()), "mrcnn_weight_flip_transpose_batched")


def weight_flip_transpose(w, out=None):
    _need_cuda(w, out)
    KH, KW, Cin, Cout = w.shape
    if out is None:
        out = empty((KH, KW, Cout, Cin), torch.float32, w.device)
    check(_hip.lib().mrcnn_weight_flip_transpose(ptr(w), ptr(out), KH, KW, Cin, Cout, current_stream()),
          "mrcnn_weight_flip_transpose")
    return out


def bn_fold(gamma, beta, mean, var, scale, shift, rstd=None, eps=BN_EPS):
    _need_cuda(gamma, beta, mean, var, scale, shift, rstd)
    check(_hip.lib().mrcnn_bn_fold(ptr(gamma), ptr(beta), ptr(mean), ptr(var), eps, ptr(scale), ptr(shift),
                                   ptr(rstd), gamma.numel(), current_stream()), "mrcnn_bn_fold")


def epilogue_bwd(dout, out=None, z=None, scale=None, mean=None, rstd=None, dy_out=None, dz_out=None,
                 dgamma=None, dbeta=None, dbias=None, act=ACT_NONE):
    _need_cuda(dout, out, z, scale, mean, rstd, dy_out, dz_out, dgamma, dbeta, dbias)
    C_ = dout.shape[-1]
    M = dout.numel() // C_
    check(_hip.lib().mrcnn_epilogue_bwd(ptr(dout), ptr(out), ptr(z), ptr(scale), ptr(mean), ptr(rstd), ptr(dy_out),
                                        ptr(dz_out), ptr(dgamma), ptr(dbeta), ptr(dbias), M, C_, act,
                                        current_stream()), "mrcnn_epilogue_bwd")


def maxpool3x3s2(x, want_argmax=False):
    _need_cuda(x)
    N, H, W, C_ = x.shape
    OH, pt = same_padding(H, 3, 2)
    OW, pl = same_padding(W, 3, 2)
    out = empty((N, OH, OW, C_), torch.float32, x.device)
    am = empty((N, OH, OW, C_), torch.int32, x.device) if want_argmax else None
    check(_hip.lib().mrcnn_maxpool3x3s2_fwd(ptr(x), ptr(out), ptr(am), N, H, W, C_, OH, OW, pt, pl,
                                            current_stream()), "mrcnn_maxpool3x3s2_fwd")
    return (out, am) if want_argmax else out


def maxpool3x3s2_bwd(dout, argmax, in_shape):
    _need_cuda(dout, argmax)
    N, H, W, C_ = in_shape
    dx = empty(in_shape, torch.float32, dout.device)
    check(_hip.lib().mrcnn_maxpool3x3s2_bwd(ptr(dout), ptr(argmax), ptr(dx), N, H, W, C_, dout.shape[1],
                                            dout.shape[2], current_stream()), "mrcnn_maxpool3x3s2_bwd")
    return dx


def subsample2(x):
    _need_cuda(x)
    N, H, W, C_ = x.shape
    out = empty((N, (H + 1) // 2, (W + 1) // 2, C_), torch.float32, x.device)
    check(_hip.lib().mrcnn_subsample2_fwd(ptr(x), ptr(out), N, H, W, C_, current_stream()), "mrcnn_subsample2_fwd")
    return out


def subsample2_bwd_acc(dout, dx):
    _need_cuda(dout, dx)
    N, H, W, C_ = dx.shape
    check(_hip.lib().mrcnn_subsample2_bwd_acc(ptr(dout), ptr(dx), N, H, W, C_, current_stream()),
          "mrcnn_subsample2_bwd_acc")


def upsample2_bwd(dout, dsrc, accumulate):
    _need_cuda(dout, dsrc)
    N, H, W, C_ = dsrc.shape
    check(_hip.lib().mrcnn_upsample2_bwd(ptr(dout), ptr(dsrc), N, H, W, C_, 1 if accumulate else 0,
                                         current_stream()), "mrcnn_upsample2_bwd")


def add_inplace(dst, src):
    _need_cuda(dst, src)
    assert dst.numel() == src.numel()
    check(_hip.lib().mrcnn_add_inplace(ptr(dst), ptr(src), dst.numel(), current_stream()), "mrcnn_add_inplace")


def softmax_rows(logits, out=None):
    _need_cuda(logits, out)
    C_ = logits.shape[-1]
    if out is None:
        out = empty_like(logits)
    check(_hip.lib().mrcnn_softmax_rows(ptr(logits), ptr(out), logits.numel() // C_, C_, current_stream()),
          "mrcnn_softmax_rows")
    return out


def _roi_desc(boxes, fms, pool, image_area):
    d = _hip.RoiAlignDesc()
    d.B, d.R, d.P, d.C = boxes.shape[0], boxes.shape[1], pool, fms[0].shape[3]
    for i, f in enumerate(fms):
        d.H[i], d.W[i] = f.shape[1], f.shape[2]
    d.image_area = float(image_area)
    return d


def roialign(boxes, fms, pool, image_area, want_levels=False):
    """PyramidROIAlign: boxes [B,R,4], fms = [P2,P3,P4,P5] -> [B,R,pool,pool,C]."""
    _need_cuda(boxes, *fms)
    d = _roi_desc(boxes, fms, pool, image_area)
    out = empty((d.B, d.R, pool, pool, d.C), torch.float32, boxes.device)
    lv = empty((d.B, d.R), torch.int32, boxes.device) if want_levels else None
    check(_hip.lib().mrcnn_roialign_fwd(C.byref(d), ptr(boxes), ptr(fms[0]), ptr(fms[1]), ptr(fms[2]), ptr(fms[3]),
                                        ptr(out), ptr(lv), current_stream()), "mrcnn_roialign_fwd")
    return (out, lv) if want_levels else out


def roialign_bwd(boxes, dout, dfms, pool, image_area, dense=False):
    """Scatter-add dout [B,R,pool,pool,C] into the (already initialised) gradient maps dfms.  dense=True: every ROI
    carries gradient -- use the gather form (one wave per pyramid pixel) when the shape allows."""
    _need_cuda(boxes, dout, *dfms)
    d = _roi_desc(boxes, dfms, pool, image_area)
    if dense:
        rc = _hip.lib().mrcnn_roialign_bwd_gather(C.byref(d), ptr(boxes), ptr(dout), ptr(dfms[0]), ptr(dfms[1]), ptr(dfms[2]),
                                                  ptr(dfms[3]), current_stream())
        if rc != ERR_UNSUPPORTED:
            check(rc, "mrcnn_roialign_bwd_gather")
            return
    check(_hip.lib().mrcnn_roialign_bwd(C.byref(d), ptr(boxes), ptr(dout), ptr(dfms[0]), ptr(dfms[1]), ptr(dfms[2]),
                                        ptr(dfms[3]), current_stream()), "mrcnn_roialign_bwd")


def roialign_h16(boxes, fms16, pool, image_area):
    """PyramidROIAlign on 16-bit pyramid levels -> 16-bit [B,R,pool,pool,C] (interpolation in float32, one rounding)."""
    _need_cuda(boxes, *fms16)
    dt = fms16[0].dtype
    assert dt in _H16 and all(f.dtype == dt and f.is_contiguous() for f in fms16)
    d = _roi_desc(boxes, fms16, pool, image_area)
    out = empty((d.B, d.R, pool, pool, d.C), dt, boxes.device)
    check(_hip.lib().mrcnn_roialign_fwd_h16(C.byref(d), _H16[dt], ptr(boxes), ptr(fms16[0]), ptr(fms16[1]), ptr(fms16[2]), ptr(fms16[3]),
                                            ptr(out), current_stream()), "mrcnn_roialign_fwd_h16")
    return out


def roialign_bwd_h16(boxes, dout16, dfms, pool, image_area, multiplier=1.0):
    """Scatter-add the 16-bit gradient dout16 [B,R,pool,pool,C] * multiplier into the float32 gradient maps dfms."""
    _need_cuda(boxes, dout16, *dfms)
    assert dout16.dtype in _H16 and dout16.is_contiguous()
    d = _roi_desc(boxes, dfms, pool, image_area)
    check(_hip.lib().mrcnn_roialign_bwd_h16(C.byref(d), _H16[dout16.dtype], ptr(boxes), ptr(dout16), float(multiplier), ptr(dfms[0]),
                                            ptr(dfms[1]), ptr(dfms[2]), ptr(dfms[3]), current_stream()), "mrcnn_roialign_bwd_h16")


def proposals(rpn_probs, rpn_bbox, anchors, pre_nms_limit, proposal_count, nms_threshold, std_dev, debug=False):
    """ProposalLayer. rpn_probs [B,A,2], rpn_bbox [B,A,4], anchors [A,4] (normalised) -> rois [B,count,4]."""
    _need_cuda(rpn_probs, rpn_bbox, anchors)
    B, A = rpn_probs.shape[0], rpn_probs.shape[1]
    d = _hip.ProposalDesc()
    d.B, d.A, d.pre_nms_limit, d.proposal_count, d.nms_threshold = B, A, pre_nms_limit, proposal_count, nms_threshold
    for i in range(4):
        d.std_dev[i] = float(std_dev[i])
    K = min(pre_nms_limit, A)
    rois = empty((B, proposal_count, 4), torch.float32, rpn_probs.device)
    top_idx = keep_idx = num_keep = None
    if debug:
        top_idx = empty((B, K), torch.int32, rois.device)
        keep_idx = empty((B, proposal_count), torch.int32, rois.device)
        num_keep = empty((B,), torch.int32, rois.device)
    nbytes = _hip.lib().mrcnn_proposal_workspace(C.byref(d))
    ws = workspace(nbytes, rois.device, "proposal")
    check(_hip.lib().mrcnn_proposal_fwd(C.byref(d), ptr(rpn_probs), ptr(rpn_bbox), ptr(anchors), ptr(rois),
                                        ptr(top_idx), ptr(keep_idx), ptr(num_keep), ptr(ws), ws.numel(),
                                        current_stream()), "mrcnn_proposal_fwd")
    if debug:
        # sorted, decoded, clipped boxes live at the head of the workspace: [B, K, 4]
        base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
        boxes = ws[base: base + B * K * 16].view(torch.float32).view(B, K, 4).clone()
        return rois, top_idx, keep_idx, num_keep, boxes
    return rois


def proposal_status(rpn_probs, pre_nms_limit, proposal_count):
    """Health words of the last proposals() call of this shape on the current stream (synchronises):
    int32 [B, 3] = keys collected, keys announced, stores refused by the bounds guard of the multi-workgroup
    selection (mrcnn_proposal_status_offset)."""
    B, A = rpn_probs.shape[0], rpn_probs.shape[1]
    d = _hip.ProposalDesc()
    d.B, d.A, d.pre_nms_limit, d.proposal_count = B, A, pre_nms_limit, proposal_count
    ws = workspace(_hip.lib().mrcnn_proposal_workspace(C.byref(d)), rpn_probs.device, "proposal")
    stride = C.c_size_t(0)
    off = _hip.lib().mrcnn_proposal_status_offset(C.byref(d), ptr(ws), C.byref(stride))
    torch.cuda.synchronize()
    words = ws.cpu().numpy()
    import numpy as np
    return np.stack([np.frombuffer(words[off + b * stride.value: off + b * stride.value + 12].tobytes(), np.uint32)
                     for b in range(B)]).astype(np.int64)


def detection_targets(proposals_, gt_class_ids, gt_boxes, gt_masks, rand_keys, train_rois, positive_ratio,
                      bbox_std_dev, mask_shape, use_mini_mask=False):
    _need_cuda(proposals_, gt_class_ids, gt_boxes, gt_masks, rand_keys)
    B, R = proposals_.shape[0], proposals_.shape[1]
    G = gt_boxes.shape[1]
    d = _hip.DetTargetDesc()
    d.B, d.R, d.G, d.T = B, R, G, train_rois
    d.MH, d.MW = gt_masks.shape[1], gt_masks.shape[2]
    d.mask_h, d.mask_w = mask_shape
    d.positive_count = int(train_rois * positive_ratio)
    import numpy as np
    d.negative_ratio_r = float(np.float32(1.0 / positive_ratio))
    for i in range(4):
        d.bbox_std_dev[i] = float(np.float32(bbox_std_dev[i]))
    d.use_mini_mask = 1 if use_mini_mask else 0
    dev = proposals_.device
    rois = empty((B, train_rois, 4), torch.float32, dev)
    tcls = empty((B, train_rois), torch.int32, dev)
    tbbox = empty((B, train_rois, 4), torch.float32, dev)
    tmask = empty((B, train_rois, mask_shape[0], mask_shape[1]), torch.float32, dev)
    assign = empty((B, train_rois), torch.int32, dev)
    counts = empty((B, 2), torch.int32, dev)
    check(_hip.lib().mrcnn_detection_targets(C.byref(d), ptr(proposals_), ptr(gt_class_ids), ptr(gt_boxes),
                                             ptr(gt_masks), ptr(rand_keys), ptr(rois), ptr(tcls), ptr(tbbox),
                                             ptr(tmask), ptr(assign), ptr(counts), current_stream()),
          "mrcnn_detection_targets")
    return rois, tcls, tbbox, tmask, assign, counts


def rpn_targets(anchors_px, gt_class_ids, gt_boxes_px, rand_keys, n_train, bbox_std_dev):
    """build_rpn_targets (mrcnn/model.py:1536-1644) for a batch: anchors_px [A,4] float64 pixels,
    gt_class_ids [B,G] int32, gt_boxes_px [B,G,4] int32, rand_keys [B,A] float32 ->
    rpn_match [B,A,1] int32, rpn_bbox [B,n_train,4] float32."""
    _need_cuda(anchors_px, gt_class_ids, gt_boxes_px, rand_keys)
    assert anchors_px.dtype == torch.float64 and gt_boxes_px.dtype == torch.int32 and gt_class_ids.dtype == torch.int32
    assert rand_keys.dtype == torch.float32
    B, G = gt_class_ids.shape
    A = anchors_px.shape[0]
    assert rand_keys.shape == (B, A) and gt_boxes_px.shape == (B, G, 4)
    d = _hip.RpnTargetDesc()
    d.B, d.A, d.G, d.n_train = B, A, G, int(n_train)
    for i in range(4):
        d.bbox_std_dev[i] = float(bbox_std_dev[i])
    dev = anchors_px.device
    match = empty((B, A, 1), torch.int32, dev)
    bbox = empty((B, int(n_train), 4), torch.float32, dev)
    nbytes = _hip.lib().mrcnn_rpn_targets_workspace(C.byref(d))
    ws = workspace(nbytes, dev, "rpn_targets")
    check(_hip.lib().mrcnn_rpn_targets(C.byref(d), ptr(anchors_px), ptr(gt_class_ids), ptr(gt_boxes_px), ptr(rand_keys),
                                       ptr(match), ptr(bbox), ptr(ws), nbytes, current_stream()), "mrcnn_rpn_targets")
    return match, bbox


def detections(rois, probs, deltas, windows, max_instances, min_confidence, nms_threshold, bbox_std_dev):
    _need_cuda(rois, probs, deltas, windows)
    B, R, C_ = probs.shape
    d = _hip.DetectionDesc()
    d.B, d.R, d.C, d.max_instances = B, R, C_, max_instances
    d.min_confidence, d.nms_threshold = float(min_confidence), float(nms_threshold)
    for i in range(4):
        d.bbox_std_dev[i] = float(bbox_std_dev[i])
    out = empty((B, max_instances, 6), torch.float32, rois.device)
    nbytes = _hip.lib().mrcnn_detection_workspace(C.byref(d))
    ws = workspace(nbytes, rois.device, "detection")
    check(_hip.lib().mrcnn_detection_fwd(C.byref(d), ptr(rois), ptr(probs), ptr(deltas), ptr(windows), ptr(out),
                                         ptr(ws), ws.numel(), current_stream()), "mrcnn_detection_fwd")
    return out


def losses_fwd_bwd(rpn_match, rpn_bbox_t, rpn_logits, rpn_bbox, tcls, tbbox, tmask, active_class_ids, cls_logits,
                   mbbox, mmask, weights, dice=False):
    """Returns (losses[5], d_rpn_logits, d_rpn_bbox, d_cls_logits, d_mbbox, d_mmask)."""
    _need_cuda(rpn_match, rpn_bbox_t, rpn_logits, rpn_bbox, tcls, tbbox, tmask, active_class_ids, cls_logits, mbbox,
               mmask)
    d = _hip.LossDesc()
    d.B, d.A = rpn_logits.shape[0], rpn_logits.shape[1]
    d.T, d.C = cls_logits.shape[1], cls_logits.shape[2]
    d.mask_h, d.mask_w = tmask.shape[2], tmask.shape[3]
    d.max_rpn_pos = rpn_bbox_t.shape[1]
    d.mask_loss_dice = 1 if dice else 0
    for i in range(5):
        d.w[i] = float(weights[i])
    dev = rpn_logits.device
    losses = torch.empty((5,), dtype=torch.float32, device=dev)
    g = [empty_like(t) for t in (rpn_logits, rpn_bbox, cls_logits, mbbox, mmask)]
    nbytes = _hip.lib().mrcnn_losses_workspace(C.byref(d))
    ws = workspace(nbytes, dev, "losses")
    check(_hip.lib().mrcnn_losses_fwd_bwd(C.byref(d), ptr(rpn_match), ptr(rpn_bbox_t), ptr(rpn_logits), ptr(rpn_bbox),
                                          ptr(tcls), ptr(tbbox), ptr(tmask), ptr(active_class_ids), ptr(cls_logits),
                                          ptr(mbbox), ptr(mmask), ptr(losses), ptr(g[0]), ptr(g[1]), ptr(g[2]),
                                          ptr(g[3]), ptr(g[4]), ptr(ws), ws.numel(), current_stream()),
          "mrcnn_losses_fwd_bwd")
    return (losses,) + tuple(g)


def grad_prepare(grads, params, grad_scale, gran_coef, sumsq_out):
    _need_cuda(grads, params, gran_coef, sumsq_out)
    check(_hip.lib().mrcnn_grad_prepare(ptr(grads), ptr(params), float(grad_scale), ptr(gran_coef), grads.numel(),
                                        ptr(sumsq_out), current_stream()), "mrcnn_grad_prepare")


def sumsq(g, out):
    _need_cuda(g, out)
    check(_hip.lib().mrcnn_sumsq(ptr(g), g.numel(), ptr(out), current_stream()), "mrcnn_sumsq")


def sgd_momentum(params, mom, grads, sumsq_t, clipnorm, lr, momentum, gran_coef, skipped=None):
    """skipped (int32 device tensor, 1 element): the guarded mixed-precision form -- a non-finite gradient norm skips the
    update on the device and counts the step there."""
    _need_cuda(params, mom, grads, sumsq_t, gran_coef, skipped)
    if skipped is not None:
        check(_hip.lib().mrcnn_sgd_momentum_guarded(ptr(params), ptr(mom), ptr(grads), ptr(sumsq_t), float(clipnorm), float(lr),
                                                    float(momentum), ptr(gran_coef), params.numel(), ptr(skipped),
                                                    current_stream()), "mrcnn_sgd_momentum_guarded")
        return
    check(_hip.lib().mrcnn_sgd_momentum(ptr(params), ptr(mom), ptr(grads), ptr(sumsq_t), float(clipnorm), float(lr),
                                        float(momentum), ptr(gran_coef), params.numel(), current_stream()),
          "mrcnn_sgd_momentum")


def pixel_unshuffle2(src, out=None):
    """[N,2H,2W,C] -> [N,H,W,4C] with column (a*2+b)*C + c."""
    _need_cuda(src, out)
    N, H2, W2, C_ = src.shape
    if out is None:
        out = empty((N, H2 // 2, W2 // 2, 4 * C_), torch.float32, src.device)
    check(_hip.lib().mrcnn_pixel_unshuffle2(ptr(src), ptr(out), N, H2 // 2, W2 // 2, C_, current_stream()),
          "mrcnn_pixel_unshuffle2")
    return out


def tuning_set(key, value):
    check(_hip.lib().mrcnn_tuning_set(key.encode(), int(value)), "mrcnn_tuning_set(%s)" % key)


def copy2d(dst_ptr, dst_pitch, src_ptr, src_pitch, row_bytes, rows):
    check(_hip.lib().mrcnn_copy2d(dst_ptr, dst_pitch, src_ptr, src_pitch, row_bytes, rows, current_stream()),
          "mrcnn_copy2d")


def unpack_mask_bits(packed, n_used, G):
    """packed [..., ceil(n_used / 8)] uint8 (numpy.packbits along the instance axis, bitorder="little") -> [..., G] uint8 planes,
    zeros from n_used on."""
    _need_cuda(packed)
    lead = tuple(packed.shape[:-1])
    out = torch.empty(lead + (G,), dtype=torch.uint8, device=packed.device)
    npix = 1
    for v in lead:
        npix *= v
    check(_hip.lib().mrcnn_unpack_mask_bits(ptr(packed), ptr(out), npix, packed.shape[-1], n_used, G, current_stream()),
          "mrcnn_unpack_mask_bits")
    return out


def unmold_masks(mrcnn_mask, dets, image_hw, packed=False, out=None):
    """utils.unmold_mask for all detections of one image on the device (mrcnn/model.py:2607-2619, mrcnn/utils.py:629-645).
    mrcnn_mask [R, MH, MW, C] float32 device; dets [n, 6] int32 device = (y1, x1, y2, x2, class_id, row of mrcnn_mask), boxes in
    pixels of the original image.  Returns uint8 [H, W, n] (0 / 1; .view(bool) is the reference's array) or, packed, [H, W,
    ceil(n / 8)] with detection d in bit d & 7 of byte d >> 3."""
    _need_cuda(mrcnn_mask, dets, out)
    assert mrcnn_mask.dtype == torch.float32 and mrcnn_mask.dim() == 4 and mrcnn_mask.is_contiguous()
    assert dets.dtype == torch.int32 and dets.dim() == 2 and dets.shape[1] == 6 and dets.is_contiguous()
    R, MH, MW, C_ = mrcnn_mask.shape
    n = dets.shape[0]
    H, W = int(image_hw[0]), int(image_hw[1])
    shape = (H, W, (n + 7) // 8 if packed else n)
    if out is None:
        out = torch.empty(shape, dtype=torch.uint8, device=mrcnn_mask.device)
    assert tuple(out.shape) == shape and out.dtype == torch.uint8 and out.is_contiguous()
    if n == 0 or H == 0 or W == 0:
        return out
    L = _hip.lib()
    nbytes = L.mrcnn_unmold_masks_workspace(n, MH, MW)
    ws = workspace(nbytes, mrcnn_mask.device, "unmold")
    check(L.mrcnn_unmold_masks(ptr(mrcnn_mask), R, MH, MW, C_, ptr(dets), n, H, W, 1 if packed else 0, ptr(out), ptr(ws),
                               ws.numel(), current_stream()), "mrcnn_unmold_masks")
    return out


def fits_to_rgb(raw, H, W, zscale_contrasts=(0.25, 0.25, 0.25), big_endian=False, out=None):
    """utils.read_fits's numeric part on the device (mrcnn/utils.py:1088-1157): raw = the tile's H * W float32 values as a
    device tensor (any dtype of 4 * H * W bytes: float32, or the file's big-endian bytes as uint8 with big_endian=True) ->
    uint8 [H, W, 3] (NaN -> min, per-channel zscale, / max, round(255 x))."""
    _need_cuda(raw, out)
    assert raw.is_contiguous() and raw.numel() * raw.element_size() == 4 * H * W, "raw must hold H * W float32 values"
    if out is None:
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=raw.device)
    L = _hip.lib()
    nbytes = L.mrcnn_fits_workspace(H, W)
    ws = workspace(nbytes, raw.device, "fits")
    zc = (C.c_double * 3)(*[float(v) for v in zscale_contrasts])
    check(L.mrcnn_fits_to_rgb(ptr(raw), 1 if big_endian else 0, H, W, zc, ptr(out), ptr(ws), ws.numel(), current_stream()),
          "mrcnn_fits_to_rgb")
    return out


def mold_image_u8(src, out_hw, pad_tl, canvas_hw, mean_pixel, out=None):
    """utils.resize_image's pixel work + mold_image on the device (mrcnn/model.py:2519-2556): src [h, w, C] uint8 device ->
    float32 [OH, OW, C]: the image scaled to out_hw (bilinear, float64, clipped to the image's range, truncated to uint8; a copy
    when out_hw == (h, w)) at pad_tl inside a zero canvas, minus mean_pixel."""
    import numpy as np
    _need_cuda(src, out)
    assert src.dtype == torch.uint8 and src.dim() == 3 and src.is_contiguous()
    h, w, C_ = src.shape
    (oh, ow), (top, left), (OH, OW) = out_hw, pad_tl, canvas_hw
    if out is None:
        out = torch.empty((OH, OW, C_), dtype=torch.float32, device=src.device)
    assert tuple(out.shape) == (OH, OW, C_) and out.dtype == torch.float32 and out.is_contiguous()
    ws = workspace(64, src.device, "mold")
    mean = (C.c_double * C_)(*[float(v) for v in np.broadcast_to(np.asarray(mean_pixel, np.float64), (C_,))])
    check(_hip.lib().mrcnn_mold_image_u8(ptr(src), h, w, C_, int(oh), int(ow), int(top), int(left), int(OH), int(OW), mean, ptr(out),
                                         ptr(ws), ws.numel(), current_stream()), "mrcnn_mold_image_u8")
    return out


def fill_zero(t):
    _need_cuda(t)
    check(_hip.lib().mrcnn_fill_zero(ptr(t), t.numel() * t.element_size(), current_stream()), "mrcnn_fill_zero")


def mask_out_bwd(d_mask, mask, up, w_mask, dw_mask, db_mask, db_deconv):
    """Fused backward of sigmoid(conv1x1(relu(deconv))) -> dzg [M, H/2, W/2, 4*Cd]; the three small
    gradients accumulate in place."""
    _need_cuda(d_mask, mask, up, w_mask, dw_mask, db_mask, db_deconv)
    M, H, W, Cd = up.shape
    C_ = mask.shape[-1]
    dzg = empty((M, H // 2, W // 2, 4 * Cd), torch.float32, up.device)
    check(_hip.lib().mrcnn_mask_out_bwd(ptr(d_mask), ptr(mask), ptr(up), ptr(w_mask), ptr(dzg), ptr(dw_mask),
                                        ptr(db_mask), ptr(db_deconv), M, H, W, Cd, C_, current_stream()),
          "mrcnn_mask_out_bwd")
    return dzg


# ---- 16-bit matrix-core path (configs[4], stage 1: ROI-head convolutions) ---------------------------------------
_H16 = {torch.float16: 0, torch.bfloat16: 1}


def weights_to_h16(w, dtype=torch.float16, want_dgrad=True, out=None):
    """float32 HWIO kernel -> (W^T [Cout, KH*KW*Cin], data-gradient image [Cin, KH*KW*Cout]) in `dtype`.
    `out` = a pair returned by an earlier call: refreshed in place (addresses captured in HIP graphs stay valid)."""
    _need_cuda(w)
    KH, KW, Cin, Cout = w.shape
    if out is not None:
        wf, wd = out
        assert wf.dtype == dtype and tuple(wf.shape) == (Cout, KH * KW * Cin)
    else:
        wf = torch.empty((Cout, KH * KW * Cin), dtype=dtype, device=w.device)
        wd = torch.empty((Cin, KH * KW * Cout), dtype=dtype, device=w.device) if want_dgrad else None
    check(_hip.lib().mrcnn_weights_to_h16(ptr(w), ptr(wf), ptr(wd), KH, KW, Cin, Cout, _H16[dtype], current_stream()),
          "mrcnn_weights_to_h16")
    return wf, wd


# ---- Winograd F(2x2, 3x3) / F(4x4, 3x3), float32 (mask-head 3x3 convolutions) ---------------------------------------------
_WINO_PERSISTENT_GEMM = os.environ.get("MRCNN_WINOGRAD_GEMM", "persistent") != "blds"     # A/B: the one-tile-per-workgroup LDS-DMA kernel
_WINO_MIN_ROWS = int(os.environ.get("MRCNN_WINOGRAD_MIN_ROWS", "16384"))   # 84 ROIs of 14 x 14: detect's 100 detections take the path (3.48 -> 3.18 ms/image)
# 4: F(4x4, 3x3) for layers of at least _WINO_TILE4_MIN_ROWS pixels (512 ROIs of 14 x 14: the training step's layers and their
# halves; detect's 100 ROIs per image stay on F(2x2, 3x3)); headline step 44.5 -> 40.9 ms.  2: F(2x2, 3x3) everywhere
_WINO_TILE = int(os.environ.get("MRCNN_WINOGRAD_TILE", "4"))
_WINO_TILE4_MIN_ROWS = int(os.environ.get("MRCNN_WINOGRAD_TILE4_MIN_ROWS", "100352"))


_WINO_MIXED = os.environ.get("MRCNN_WINOGRAD_MIXED", "1") != "0"          # tile 4 on maps with extent % 4 == 2: four groups, no overhang
TILE_MIXED = 6                                                              # "4 + 2": 4 x 4 tiles and a last row / column of 2


def winograd_tile(xshape):
    """Tiling of a TRAINING layer of this input shape: 2 (F(2x2,3x3) everywhere), 4 (F(4x4,3x3): 36 GEMMs over a quarter of the
    tiles, about one decimal digit less accurate: 1.3e-5 of the result's range against 1.6e-6) for layers of at least
    MRCNN_WINOGRAD_TILE4_MIN_ROWS pixels, or TILE_MIXED when H and W are 2 mod 4 (the mask head's 14 x 14): 4 x 4 tiles plus a
    last row / column of 4 x 2, 2 x 4 and 2 x 2 tiles -- 484 multiplications per map and channel pair where the uniform 4 x 4
    tiling with its overhang needs 576.  MRCNN_WINOGRAD_TILE=2 keeps 2 everywhere, MRCNN_WINOGRAD_MIXED=0 the uniform tiling.
    A pure function of the shape, so the forward pass, the data gradient and the weight gradient of a layer agree on it."""
    N, H, W, _ = xshape
    if not (_WINO_TILE == 4 and N * H * W >= _WINO_TILE4_MIN_ROWS):
        return 2
    return TILE_MIXED if (_WINO_MIXED and H % 4 == 2 and W % 4 == 2 and H > 4 and W > 4) else 4


def winograd_groups(H, W, tile):
    """The tile groups (mrcnn_wino_group) of an H x W map under a tiling."""
    G = _hip.WinoGroup
    if tile == 2:
        assert H % 2 == 0 and W % 2 == 0
        return [G(2, 2, H // 2, W // 2, 0, 0)]
    if tile == 4:
        return [G(4, 4, (H + 3) // 4, (W + 3) // 4, 0, 0)]
    assert tile == TILE_MIXED and H % 4 == 2 and W % 4 == 2
    return [G(4, 4, H // 4, W // 4, 0, 0), G(4, 2, H // 4, 1, 0, W - 2), G(2, 4, 1, W // 4, H - 2, 0), G(2, 2, 1, 1, H - 2, W - 2)]


def _wino_nb(g):
    return (g.oth + 2) * (g.otw + 2)


def _wino_tile_of(U):
    if isinstance(U, (list, tuple)):
        assert [u.shape[0] for u in U] == [36, 24, 24, 16], "U is not a mixed Winograd weight transform"
        return TILE_MIXED
    nb = U.shape[0]
    assert nb in (16, 36), "U is not a Winograd weight transform"
    return 2 if nb == 16 else 4


def winograd_ok(xshape, wshape, stride=1, padding="same", min_rows=None):
    """Shapes the Winograd path takes: 3 x 3, stride 1, 'same' (or explicit 1, 1), even H and W, channels that fit the batched
    GEMM (Cin % 16, Cout % 128) and the transforms (multiples of 4), enough rows to fill the chip."""
    N, H, W, Cin = xshape
    kh, kw, cin, cout = wshape
    if min_rows is None:
        min_rows = _WINO_MIN_ROWS
    return (kh == 3 and kw == 3 and stride == 1 and padding in ("same", (1, 1)) and H % 2 == 0 and W % 2 == 0 and cin == Cin and
            Cin % 16 == 0 and cout % 128 == 0 and N * H * W >= min_rows and N * (H // 2) * (W // 2) * 16 * max(Cin, cout) < 2 ** 31)


def winograd_weights(w, out=None, tile=2):
    """HWIO 3 x 3 kernel -> U = G g G^T: [(tile + 2)^2, Cin, Cout] for the uniform tilings, a list of four (36, 24, 24, 16
    matrices) for TILE_MIXED."""
    kh, kw, cin, cout = w.shape
    assert kh == 3 and kw == 3 and w.is_contiguous()
    if tile != TILE_MIXED:
        _need_cuda(w, out)
        nb = (tile + 2) ** 2
        if out is None:
            out = torch.empty((nb, cin, cout), dtype=torch.float32, device=w.device)
        assert out.shape[0] == nb
        check(_hip.lib().mrcnn_winograd_weights(ptr(w), ptr(out), cin, cout, tile, current_stream()), "mrcnn_winograd_weights")
        return out
    shapes = ((4, 4), (4, 2), (2, 4), (2, 2))
    if out is None:
        out = [torch.empty(((a + 2) * (b + 2), cin, cout), dtype=torch.float32, device=w.device) for a, b in shapes]
    _need_cuda(w, *out)
    for (a, b), u in zip(shapes, out):
        assert u.shape[0] == (a + 2) * (b + 2)
        check(_hip.lib().mrcnn_winograd_weights_g(ptr(w), ptr(u), cin, cout, a, b, current_stream()), "mrcnn_winograd_weights_g")
    return out


def _winograd_run(x, U, finish, keep_v=None, after_input=None, fuse=None):
    """Per tile group: input transform, the group's transform-domain GEMMs, then finish(group, Mt) -- the output transform the caller
    wants.  V of all groups lives in one flat buffer (keep_v: kept for the weight gradient), Mt is a per-stream scratch buffer
    reused group after group (stream order)."""
    N, H, W, Cin = x.shape
    tile = _wino_tile_of(U)
    Us = U if isinstance(U, (list, tuple)) else [U]
    cout = Us[0].shape[2]
    groups = winograd_groups(H, W, tile)
    lib = _hip.lib()
    nvs = [lib.mrcnn_winograd_group_floats(C.byref(g), N, Cin) for g in groups]
    nms = [lib.mrcnn_winograd_group_floats(C.byref(g), N, cout) for g in groups]
    nv = sum(nvs)
    V = keep_v if keep_v is not None else workspace(nv * 4, x.device, "winograd_v")[:nv * 4].view(torch.float32)
    assert keep_v is None or (keep_v.numel() == nv and keep_v.dtype == torch.float32)
    Mt = workspace(max(nms) * 4, x.device, "winograd_m")
    gemm = lib.mrcnn_winograd_gemm if _WINO_PERSISTENT_GEMM else lib.mrcnn_gemm_batched_f32
    off = 0
    for gi, (g, u) in enumerate(zip(groups, Us)):
        Vg = V[off:off + nvs[gi]]
        off += nvs[gi]
        check(lib.mrcnn_winograd_input_g(ptr(x), ptr(Vg), N, H, W, Cin, C.byref(g), current_stream()), "mrcnn_winograd_input_g")
        if gi == 0 and after_input is not None:
            after_input()                # e.g. an event: a second chain on another stream starts one transform behind this one
        nb = _wino_nb(g)
        rows = nvs[gi] // (nb * Cin)
        if fuse is not None and _WINO_FUSE and _WINO_PERSISTENT_GEMM and g.oth == 4 and g.otw == 4 and cout == 256 and \
                (nb * rows // 128) >= _WINO_FUSE_MIN_TILES:
            # the group's output transform inside the GEMM launch (mrcnn_winograd_gemm_fused): no second launch, its memory pass runs
            # beside the other workgroups' MFMAs
            f = fuse(g)
            f.counters = ptr(_wino_counters(x.device, rows // 128))
            rc = lib.mrcnn_winograd_gemm_fused(ptr(Vg), ptr(u), ptr(Mt), nb, rows, Cin, cout, C.byref(f), current_stream())
            if rc != ERR_UNSUPPORTED:
                check(rc, "mrcnn_winograd_gemm_fused")
                continue
        check(gemm(ptr(Vg), ptr(u), ptr(Mt), nb, rows, Cin, cout, current_stream()), "mrcnn_winograd_gemm")
        finish(g, Mt)


# output transform of the 4 x 4 group fused into its GEMM launch (mrcnn_winograd_gemm_fused).  Built, bit-identical to the separate
# launches (test_winograd_fused_output_equals_unfused) -- and OFF: measured 1.74 -> 5.3 ms per forward layer, 1.88 -> 12 ms per data
# gradient at 2048 ROIs (tools/wino_fuse_probe.py; 5.15 / 11.1 ms even without the per-tile release fence).  One workgroup
# transforming a 128-tile row block is a chain of 64 dependent load rounds from the memory side (~0.5 ms), it holds up that
# workgroup's next tiles and with them the 35 other row blocks they belong to -- DESIGN.md 4.1e
_WINO_FUSE = os.environ.get("MRCNN_WINOGRAD_FUSE", "0") != "0"
_WINO_FUSE_MIN_TILES = int(os.environ.get("MRCNN_WINOGRAD_FUSE_MIN_TILES", "1536"))   # GEMM tiles (two rounds of 768 workgroups) from which it pays
_wino_counter_cache = {}


def _wino_counters(device, n):
    """Row-block counters of mrcnn_winograd_gemm_fused: zero on entry, left zero by the kernel -- one buffer per stream."""
    key = (str(device), current_stream())
    t = _wino_counter_cache.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 1024), dtype=torch.int32, device=device)
        _wino_counter_cache[key] = t
    return t


def winograd_v_floats(xshape, tile=None):
    N, H, W, Cin = xshape
    if tile is None:
        tile = winograd_tile(xshape)
    lib = _hip.lib()
    return sum(lib.mrcnn_winograd_group_floats(C.byref(g), N, Cin) for g in winograd_groups(H, W, tile))


def conv2d_winograd(x, U, bias=None, scale=None, shift=None, act=ACT_NONE, out=None, z_out=None, keep_v=None, after_input=None):
    """3 x 3 'same' stride-1 convolution with its epilogue through the Winograd domain (U = winograd_weights(w, tile=...))."""
    _need_cuda(x, bias, scale, shift, out, z_out, keep_v)
    N, H, W, _ = x.shape
    cout = (U[0] if isinstance(U, (list, tuple)) else U).shape[2]
    if out is None:
        out = empty((N, H, W, cout), torch.float32, x.device)
    lib = _hip.lib()

    def finish(g, Mt):
        check(lib.mrcnn_winograd_output_g(ptr(Mt), ptr(out), ptr(z_out), ptr(bias), ptr(scale), ptr(shift), N, H, W, cout, act, C.byref(g),
                                          current_stream()), "mrcnn_winograd_output_g")

    def fuse(g):
        f = _hip.WinoFuse()
        f.mode, f.N, f.H, f.W, f.act, f.g = 1, N, H, W, act, g
        f.out, f.z, f.bias, f.scale, f.shift = ptr(out), ptr(z_out), ptr(bias), ptr(scale), ptr(shift)
        return f
    _winograd_run(x, U, finish, keep_v, after_input, fuse if act in (ACT_NONE, ACT_RELU) else None)
    return out


_WINO_ZMASK = os.environ.get("MRCNN_WINOGRAD_ZMASK", "1") != "0"       # ReLU mask of the layer below from its stored z (its `out` is not read)


def conv2d_dgrad_ep_winograd(dz, Ut, below_out, below_z, scale, mean, rstd, dgamma, dbeta, dbias, act, fwd_shift=None):
    """Data gradient of a 3 x 3 'same' convolution (Ut = winograd_weights of the flipped / transposed kernel) fused with the
    epilogue backward of the layer below: returns dz_below, channel sums are added to dgamma / dbeta / dbias.
    fwd_shift: the shift of the layer below's forward epilogue out = max(scale * z + shift, 0) -- with it (and z, scale) the ReLU
    mask is recomputed from z, bit for bit the forward's decision, and below_out is not read (18 % of the pass's bytes)."""
    _need_cuda(dz, below_out, below_z, scale, mean, rstd, dgamma, dbeta, dbias, fwd_shift)
    N, H, W, _ = dz.shape
    cout = (Ut[0] if isinstance(Ut, (list, tuple)) else Ut).shape[2]
    out = empty((N, H, W, cout), torch.float32, dz.device)
    lib = _hip.lib()
    zmask = _WINO_ZMASK and act == ACT_RELU and fwd_shift is not None and below_z is not None and scale is not None

    def finish(g, Mt):
        if zmask:
            check(lib.mrcnn_winograd_output_bwd_zmask_g(ptr(Mt), ptr(out), ptr(below_z), ptr(scale), ptr(fwd_shift), ptr(mean), ptr(rstd),
                                                        ptr(dgamma), ptr(dbeta), ptr(dbias), N, H, W, cout, C.byref(g), current_stream()),
                  "mrcnn_winograd_output_bwd_zmask_g")
            return
        check(lib.mrcnn_winograd_output_bwd_g(ptr(Mt), ptr(out), ptr(below_out), ptr(below_z), ptr(scale), ptr(mean), ptr(rstd),
                                              ptr(dgamma), ptr(dbeta), ptr(dbias), N, H, W, cout, act, C.byref(g), current_stream()),
              "mrcnn_winograd_output_bwd_g")

    def fuse(g):
        f = _hip.WinoFuse()
        f.mode, f.N, f.H, f.W, f.act, f.g = 2, N, H, W, act, g
        f.out, f.scale, f.shift = ptr(out), ptr(scale), (ptr(fwd_shift) if zmask else None)
        f.below_out, f.below_z, f.mean, f.rstd = (None if zmask else ptr(below_out)), ptr(below_z), ptr(mean), ptr(rstd)
        f.dgamma, f.dbeta, f.dbias = ptr(dgamma), ptr(dbeta), ptr(dbias)
        return f
    _winograd_run(dz, Ut, finish, fuse=fuse)
    return out


def conv2d_wgrad_winograd(V, xshape, dz, dw, accumulate=False, tile=None):
    """Weight gradient of a 3 x 3 'same' convolution through the Winograd domain: V = the forward's input transform of x
    (conv2d_winograd(..., keep_v=V)), dz [N, H, W, Cout] -> dw [3, 3, Cin, Cout] float32.  tile: the forward's (default:
    winograd_tile(xshape), what the forward chose for this shape)."""
    _need_cuda(V, dz, dw)
    N, H, W, Cin = xshape
    cout = dz.shape[3]
    if tile is None:
        tile = winograd_tile(xshape)
    groups = winograd_groups(H, W, tile)
    lib = _hip.lib()
    nvs = [lib.mrcnn_winograd_group_floats(C.byref(g), N, Cin) for g in groups]
    nms = [lib.mrcnn_winograd_group_floats(C.byref(g), N, cout) for g in groups]
    assert V.numel() == sum(nvs), "V was made with another tiling"
    dM_all = workspace(max(nms) * 4, dz.device, "winograd_dm")
    dU_all = workspace(36 * Cin * cout * 4, dz.device, "winograd_du")
    off = 0
    for gi, g in enumerate(groups):
        nb = _wino_nb(g)
        rows = nms[gi] // (nb * cout)
        T = N * g.th_n * g.tw_n
        dM = dM_all[:nms[gi] * 4].view(torch.float32).view(nb, rows, cout)
        dU = dU_all[:nb * Cin * cout * 4].view(torch.float32).view(nb, Cin, cout)
        check(lib.mrcnn_winograd_dy_g(ptr(dz), ptr(dM), N, H, W, cout, C.byref(g), current_stream()), "mrcnn_winograd_dy_g")
        Vv = V[off:off + nvs[gi]].view(nb, rows, Cin)
        off += nvs[gi]
        items = [(Vv[k, :T].view(T, 1, 1, Cin), dM[k, :T].view(T, 1, 1, cout), (1, 1, Cin, cout), 1, "valid", dU[k].view(1, 1, Cin, cout), False)
                 for k in range(nb)]
        per = 16 if nb == 16 else 12                           # GEMMs per launch (+ one launch for their slab reductions): the argument block holds 16
        for i in range(0, nb, per):
            if not conv2d_wgrad_multi(items[i:i + per]):
                for x_, dy_, wshape, stride, padding, dw_, acc in items[i:i + per]:
                    conv2d_wgrad(x_, dy_, wshape, stride, padding, dw=dw_, accumulate=acc)
        check(lib.mrcnn_winograd_dw_g(ptr(dU), ptr(dw), Cin, cout, 1 if (accumulate or gi > 0) else 0, g.oth, g.otw, current_stream()),
              "mrcnn_winograd_dw_g")
    return dw


def h16_image_table(params, entries, device):
    """entries: [(w (a view into the flat float32 `params`), wf, wd or None)] -> (device table, n_layers, total_tiles) for
    weights_to_h16_batched.  The images' addresses are baked in: build it once the images exist and never move."""
    import numpy as np
    rec = np.zeros(len(entries), dtype=[("off", "<i8"), ("wf", "<u8"), ("wd", "<u8"), ("KH", "<i4"), ("KW", "<i4"), ("Cin", "<i4"),
                                        ("Cout", "<i4"), ("first", "<i4"), ("pad", "<i4")])
    tiles = 0
    for i, (w, wf, wd) in enumerate(entries):
        kh, kw, cin, cout = w.shape
        off = (w.data_ptr() - params.data_ptr()) // 4
        assert 0 <= off and off + w.numel() <= params.numel() and w.is_contiguous()
        rec[i] = (off, wf.data_ptr() if wf is not None else 0, wd.data_ptr() if wd is not None else 0, kh, kw, cin, cout, tiles, 0)
        tiles += kh * kw * ((cin + 31) // 32) * ((cout + 31) // 32)
    t = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
    return t, len(entries), tiles


def weights_to_h16_batched(params, table, dtype):
    _need_cuda(params, table[0])
    check(_hip.lib().mrcnn_weights_to_h16_batched(ptr(params), ptr(table[0]), table[1], table[2], _H16[dtype], current_stream()),
          "mrcnn_weights_to_h16_batched")


def conv2d_h16(x, w_t, kshape, bias=None, scale=None, shift=None, stride=1, padding="same", act=ACT_NONE, z_out=None,
               out=None, res=None, out_strides=None):
    """16-bit convolution: x [N,H,W,Cin] half/bfloat16, w_t = W^T [Cout, KH*KW*Cin]; kshape = (KH, KW, Cin, Cout).
    res: 16-bit tensor of out's shape added before the activation; out_strides (n, h, w element strides): strided store
    into `out` (small-tile kernel)."""
    _need_cuda(x, w_t, bias, scale, shift, z_out, out, res)
    assert x.dtype in _H16 and w_t.dtype == x.dtype and (res is None or res.dtype == x.dtype)
    d = conv_desc(tuple(x.shape), tuple(kshape), stride, padding, act, RES_SAME if res is not None else RES_NONE)
    if out_strides is not None:
        assert out is not None
        d.out_n_stride, d.out_h_stride, d.out_w_stride = out_strides
    if out is None:
        out = empty((d.N, d.OH, d.OW, d.Cout), x.dtype, x.device)
    check(_hip.lib().mrcnn_conv2d_fwd_h16_res(C.byref(d), _H16[x.dtype], ptr(x), ptr(w_t), ptr(bias), ptr(scale), ptr(shift),
                                              ptr(res), ptr(out), ptr(z_out), current_stream()), "mrcnn_conv2d_fwd_h16")
    return out


def conv2d_dgrad_ep_h16(dz, w_t, kshape, padding, below_out, below_z, scale, mean, rstd, dgamma, dbeta, dbias, act,
                        grad_multiplier=1.0, res=None, want_dy=False):
    """16-bit data gradient fused with the epilogue backward of the layer below (mrcnn_conv2d_dgrad_ep_h16).  Returns
    dz_below (or (dz_below, dy)), or None when the shape has no fused kernel."""
    _need_cuda(dz, w_t, below_out, below_z, scale, mean, rstd, dgamma, dbeta, dbias, res)
    d = conv_desc(tuple(dz.shape), tuple(kshape), 1, padding, ACT_NONE, RES_SAME if res is not None else RES_NONE)
    ep = _hip.BwdEpilogueH16()
    ep.out, ep.z, ep.scale, ep.mean, ep.rstd = ptr(below_out), ptr(below_z), ptr(scale), ptr(mean), ptr(rstd)
    ep.dgamma, ep.dbeta, ep.dbias, ep.act = ptr(dgamma), ptr(dbeta), ptr(dbias), act
    out = empty((d.N, d.OH, d.OW, d.Cout), dz.dtype, dz.device)
    dy = empty((d.N, d.OH, d.OW, d.Cout), dz.dtype, dz.device) if want_dy else None
    ep.dy, ep.grad_multiplier = ptr(dy), float(grad_multiplier)
    rc = _hip.lib().mrcnn_conv2d_dgrad_ep_h16(C.byref(d), _H16[dz.dtype], ptr(dz), ptr(w_t), ptr(res), ptr(out), C.byref(ep),
                                              current_stream())
    if rc == ERR_UNSUPPORTED:
        return None
    check(rc, "mrcnn_conv2d_dgrad_ep_h16")
    return (out, dy) if want_dy else out


def conv2d_h16_supported(x_shape, kshape, stride=1, padding="same", res=False):
    d = conv_desc(tuple(x_shape), tuple(kshape), stride, padding)
    return bool(_hip.lib().mrcnn_conv2d_fwd_h16_supported(C.byref(d), 1 if res else 0))


def deconv2x2_h16(x, w_t, bias, Cd, act=ACT_RELU):
    """Conv2DTranspose(2x2, stride 2) on 16-bit x [N,H,W,Cin]: w_t = W^T [4*Cd, Cin] of the GEMM matrix."""
    _need_cuda(x, w_t, bias)
    N, H, W, Cin = x.shape
    d = _hip.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad_t, d.pad_l = N, H, W, Cin, 4 * Cd, 1, 1, 1, 0, 0
    d.OH, d.OW, d.act, d.res_mode, d.out_mode, d.cmod = H, W, act, RES_NONE, OUT_DECONV2, Cd
    d.out_w_stride, d.out_h_stride, d.out_n_stride = Cd, 2 * W * Cd, 4 * H * W * Cd
    out = empty((N, 2 * H, 2 * W, Cd), x.dtype, x.device)
    check(_hip.lib().mrcnn_conv2d_fwd_h16(C.byref(d), _H16[x.dtype], ptr(x), ptr(w_t), ptr(bias), None, None, ptr(out), None,
                                          current_stream()), "mrcnn_conv2d_fwd_h16(deconv)")
    return out


def mask_out_fwd_h16(up, w_mask, b_mask):
    """sigmoid(1x1 conv) of the 16-bit deconvolution output -> float32 [M,H,W,C]."""
    _need_cuda(up, w_mask, b_mask)
    M, H, W, Cd = up.shape
    C_ = w_mask.shape[-1]
    out = empty((M, H, W, C_), torch.float32, up.device)
    check(_hip.lib().mrcnn_mask_out_fwd_h16(_H16[up.dtype], ptr(up), ptr(w_mask), ptr(b_mask), ptr(out), M * H * W, Cd, C_,
                                            current_stream()), "mrcnn_mask_out_fwd_h16")
    return out


def mask_out_bwd_h16(d_mask_out, mask_out, up, w_mask, dw_mask, db_mask, db_deconv, loss_scale):
    """One-pass backward of the mask-head output stage on a 16-bit `up`; returns dzg [M,H/2,W/2,4*Cd] (16 bit, scaled)."""
    _need_cuda(d_mask_out, mask_out, up, w_mask, dw_mask, db_mask, db_deconv)
    M, H, W, Cd = up.shape
    C_ = mask_out.shape[-1]
    dzg = empty((M, H // 2, W // 2, 4 * Cd), up.dtype, up.device)
    check(_hip.lib().mrcnn_mask_out_bwd_h16(_H16[up.dtype], ptr(d_mask_out), ptr(mask_out), ptr(up), ptr(w_mask), ptr(dzg),
                                            ptr(dw_mask), ptr(db_mask), ptr(db_deconv), M, H, W, Cd, C_, float(loss_scale),
                                            current_stream()), "mrcnn_mask_out_bwd_h16")
    return dzg


def conv2d_wgrad_h16(x, dy, w_shape, stride=1, padding="same", dw=None, accumulate=False, multiplier=1.0):
    """dw (float32 HWIO) from 16-bit x / dy; multiplier undoes a loss scale."""
    _need_cuda(x, dy, dw)
    assert x.dtype in _H16 and dy.dtype == x.dtype
    d = conv_desc(tuple(x.shape), tuple(w_shape), stride, padding)
    if dw is None:
        dw = empty(tuple(w_shape), torch.float32, x.device)
    nbytes = _hip.lib().mrcnn_conv2d_wgrad_h16_workspace(C.byref(d))
    ws = workspace(nbytes, x.device, "wgrad_h16")
    check(_hip.lib().mrcnn_conv2d_wgrad_h16(C.byref(d), _H16[x.dtype], ptr(x), ptr(dy), ptr(dw), ptr(ws), ws.numel(),
                                            1 if accumulate else 0, float(multiplier), current_stream()),
          "mrcnn_conv2d_wgrad_h16")
    return dw


def epilogue_bwd_h16(dout, out=None, z=None, scale=None, mean=None, rstd=None, dgamma=None, dbeta=None, dbias=None,
                     act=ACT_NONE, grad_multiplier=1.0, want_dy=False):
    """16-bit twin of epilogue_bwd: returns dz (same dtype as dout), or (dz, dy) with want_dy (dy = dout * act': the
    gradient a shortcut receives); channel sums go to the float32 gradients."""
    _need_cuda(dout, out, z, scale, mean, rstd, dgamma, dbeta, dbias)
    C_ = dout.shape[-1]
    M = dout.numel() // C_
    dz = empty(dout.shape, dout.dtype, dout.device)
    dy = empty(dout.shape, dout.dtype, dout.device) if want_dy else None
    check(_hip.lib().mrcnn_epilogue_bwd_h16_dy(_H16[dout.dtype], ptr(dout), ptr(out), ptr(z), ptr(scale), ptr(mean), ptr(rstd),
                                               ptr(dz), ptr(dy), ptr(dgamma), ptr(dbeta), ptr(dbias), M, C_, act,
                                               float(grad_multiplier), current_stream()), "mrcnn_epilogue_bwd_h16")
    return (dz, dy) if want_dy else dz


def cast_to_h16(src, dtype=torch.float16, out=None, multiplier=1.0):
    _need_cuda(src, out)
    if out is None:
        out = empty(src.shape, dtype, src.device)
    check(_hip.lib().mrcnn_cast_to_h16(ptr(src), ptr(out), src.numel(), _H16[dtype], float(multiplier), current_stream()),
          "mrcnn_cast_to_h16")
    return out


def cast_from_h16(src, multiplier=1.0, out=None):
    _need_cuda(src, out)
    if out is None:
        out = empty(src.shape, torch.float32, src.device)
    check(_hip.lib().mrcnn_cast_from_h16(ptr(src), ptr(out), src.numel(), _H16[src.dtype], float(multiplier),
                                         current_stream()), "mrcnn_cast_from_h16")
    return out


def axpy_from_h16(src, dst, multiplier=1.0):
    """dst (float32) += multiplier * src (16 bit), elementwise."""
    _need_cuda(src, dst)
    assert dst.dtype == torch.float32 and dst.numel() == src.numel()
    check(_hip.lib().mrcnn_axpy_from_h16(ptr(src), ptr(dst), src.numel(), _H16[src.dtype], float(multiplier), current_stream()),
          "mrcnn_axpy_from_h16")
    return dst
