#!/usr/bin/env python3
"""The Winograd path on the mask-head layer, tile 2 (F(2x2,3x3)), tile 4 (F(4x4,3x3) with overhang) and the mixed tiling: forward, data gradient with the fused
epilogue backward and weight gradient against the direct float32 kernels (error relative to the result's maximum), and the time of
every stage (tools only).  usage: wino_proto.py [ROIs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops, _hip
lib = _hip.lib()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = W = 14; Cc = 256
st, ptr = ops.current_stream, ops.ptr
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
torch.manual_seed(0)
x = torch.relu(torch.randn(N, H, W, Cc, device=dev))
w = torch.randn(3, 3, Cc, Cc, device=dev) * 0.03
b = torch.randn(Cc, device=dev) * 0.1
sc = torch.rand(Cc, device=dev) + 0.5; sh = torch.randn(Cc, device=dev) * 0.1
mean = torch.randn(Cc, device=dev) * 0.1; rstd = torch.rand(Cc, device=dev) + 0.5
dy = torch.randn(N, H, W, Cc, device=dev)
wt = w.flip(0, 1).permute(0, 1, 3, 2).contiguous()
yr = torch.empty_like(x); zr = torch.empty_like(x)
ops.conv2d(x, w, b, sc, sh, act=1, out=yr, z_out=zr)
dwr = torch.empty_like(w); ops.conv2d_wgrad(x, dy, (3, 3, Cc, Cc), 1, "same", dw=dwr)
sums_r = [torch.zeros(Cc, device=dev) for _ in range(3)]
dxr = ops.conv2d_dgrad_ep(dy, wt, "same", yr, zr, sc, mean, rstd, sums_r[0], sums_r[1], sums_r[2], 1)
t_d = timed(lambda: ops.conv2d(x, w, b, sc, sh, act=1, out=yr, z_out=zr))
t_dw = timed(lambda: ops.conv2d_wgrad(x, dy, (3, 3, Cc, Cc), 1, "same", dw=dwr))
print("direct: forward %.3f ms, weight gradient %.3f ms" % (t_d, t_dw))
import ctypes as C
for tile in (2, 4, ops.TILE_MIXED):
    g0 = ops.winograd_groups(H, W, tile)[0]
    nb = (g0.oth + 2) * (g0.otw + 2)
    nv0 = lib.mrcnn_winograd_group_floats(C.byref(g0), N, Cc)
    rows = nv0 // (nb * Cc)
    V = torch.empty(ops.winograd_v_floats((N, H, W, Cc), tile), device=dev); Mt = torch.empty(nv0, device=dev)
    U = ops.winograd_weights(w, tile=tile); Ut = ops.winograd_weights(wt, tile=tile)
    y = torch.empty_like(x); z = torch.empty_like(x)
    ops.conv2d_winograd(x, U, b, sc, sh, 1, out=y, z_out=z, keep_v=V)
    dw = torch.empty_like(w); ops.conv2d_wgrad_winograd(V, tuple(x.shape), dy, dw, tile=tile)
    line = "tile %d: forward %.3g (z %.3g), weight gradient %.3g" % (tile, rel(y, yr), rel(z, zr), rel(dw, dwr))
    if dxr is not None:
        sums = [torch.zeros(Cc, device=dev) for _ in range(3)]
        dx = ops.conv2d_dgrad_ep_winograd(dy, Ut, yr, zr, sc, mean, rstd, sums[0], sums[1], sums[2], 1)
        line += ", data gradient %.3g, channel sums %.3g %.3g %.3g" % ((rel(dx, dxr),) + tuple(rel(a, c) for a, c in zip(sums, sums_r)))
    print(line, flush=True)
    U0 = U[0] if isinstance(U, (list, tuple)) else U
    t_in = timed(lambda: lib.mrcnn_winograd_input_g(ptr(x), ptr(V), N, H, W, Cc, C.byref(g0), st()))
    t_g = timed(lambda: lib.mrcnn_winograd_gemm(ptr(V), ptr(U0), ptr(Mt), nb, rows, Cc, Cc, st()))
    t_out = timed(lambda: lib.mrcnn_winograd_output_g(ptr(Mt), ptr(y), ptr(z), ptr(b), ptr(sc), ptr(sh), N, H, W, Cc, 1, C.byref(g0), st()))
    t_l = timed(lambda: ops.conv2d_winograd(x, U, b, sc, sh, 1, out=y, z_out=z))
    t_w = timed(lambda: ops.conv2d_wgrad_winograd(V, tuple(x.shape), dy, dw, tile=tile))
    t_dyt = timed(lambda: lib.mrcnn_winograd_dy_g(ptr(dy), ptr(Mt), N, H, W, Cc, C.byref(g0), st()))
    print("        main group: input %.3f ms, %d GEMMs %.3f ms (%.1f TFLOP/s), output %.3f ms; layer %.3f ms; weight gradient %.3f ms (dy transform %.3f)"
          % (t_in, nb, t_g, 2.0 * nb * rows * Cc * Cc / t_g / 1e9, t_out, t_l, t_w, t_dyt), flush=True)
