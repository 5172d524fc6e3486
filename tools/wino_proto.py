#!/usr/bin/env python3
"""Winograd F(2x2,3x3) prototype on the mask-head layer: correctness against the direct float32 kernel, stage timings (tools only)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops, _hip
lib = _hip.lib()
P = C.c_void_p
lib.mrcnn_winograd_input.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_int, P]
lib.mrcnn_winograd_weights.argtypes = [P, P, C.c_int, C.c_int, P]
lib.mrcnn_winograd_output.argtypes = [P, P, P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = W = 14; Cc = 256
T = N * 49
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
torch.manual_seed(0)
x = torch.randn(N, H, W, Cc, device=dev)
w = torch.randn(3, 3, Cc, Cc, device=dev) * 0.03
b = torch.randn(Cc, device=dev) * 0.1
sc = torch.rand(Cc, device=dev) + 0.5; sh = torch.randn(Cc, device=dev) * 0.1
V = torch.empty(16, T, Cc, device=dev); Mt = torch.empty(16, T, Cc, device=dev)
U = torch.empty(16, Cc, Cc, device=dev)
y = torch.empty_like(x); z = torch.empty_like(x)
f_in = lambda: lib.mrcnn_winograd_input(ptr(x), ptr(V), N, H, W, Cc, st())
f_w = lambda: lib.mrcnn_winograd_weights(ptr(w), ptr(U), Cc, Cc, st())
def f_gemm():
    for k in range(16):
        ops.conv2d(V[k].view(T, 1, 1, Cc), U[k].view(1, 1, Cc, Cc), None, None, None, out=Mt[k].view(T, 1, 1, Cc), stride=1, padding="valid")
f_out = lambda: lib.mrcnn_winograd_output(ptr(Mt), ptr(y), ptr(z), ptr(b), ptr(sc), ptr(sh), N, H, W, Cc, 1, st())
assert f_w() == 0 and f_in() == 0
f_gemm()
assert f_out() == 0
yr = torch.empty_like(x); zr = torch.empty_like(x)
ops.conv2d(x, w, b, sc, sh, act=1, out=yr, z_out=zr)
torch.cuda.synchronize()
print("max |y - direct| / max |direct| = %.3g   z: %.3g" % (float((y - yr).abs().max() / yr.abs().max()), float((z - zr).abs().max() / zr.abs().max())))
t_in, t_g, t_out, t_w = timed(f_in), timed(f_gemm), timed(f_out), timed(f_w)
Vb = V.view(16 * T, 1, 1, Cc); Mb = Mt.view(16 * T, 1, 1, Cc)
t_g1 = timed(lambda: ops.conv2d(Vb, U[0].view(1, 1, Cc, Cc), None, None, None, out=Mb, stride=1, padding="valid"))
t_d = timed(lambda: ops.conv2d(x, w, b, sc, sh, act=1, out=yr, z_out=zr))
print("input transform %.3f ms, 16 GEMM launches %.3f ms (one launch of the same rows: %.3f), output transform %.3f ms, weights %.3f ms" % (t_in, t_g, t_g1, t_out, t_w))
print("Winograd total %.3f ms (with a batched GEMM launch: %.3f)  direct %.3f ms" % (t_in + t_g + t_out, t_in + t_g1 + t_out, t_d))
# beside a weight gradient on another stream (what the step does)
from caesar_mrcnn_amd.engine import _side_streams
side = _side_streams(dev)[0]; main = torch.cuda.current_stream(dev)
dy = torch.randn_like(x); dw = torch.empty(3, 3, Cc, Cc, device=dev)
def pair(fwd):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ops.conv2d_wgrad(x, dy, (3, 3, Cc, Cc), 1, "same", dw=dw)
    fwd()
    main.wait_stream(side)
def wino():
    f_in(); ops.conv2d(Vb, U[0].view(1, 1, Cc, Cc), None, None, None, out=Mb, stride=1, padding="valid"); f_out()
t_pw = timed(lambda: pair(wino)); t_pd = timed(lambda: pair(lambda: ops.conv2d(x, w, b, sc, sh, act=1, out=yr, z_out=zr)))
print("beside the layer's weight gradient on the side stream: Winograd (batched proxy) %.3f ms, direct %.3f ms" % (t_pw, t_pd))
