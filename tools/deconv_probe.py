#!/usr/bin/env python3
"""The mask head's transposed convolution (2 x 2, stride 2, 256 -> 256 on 14 x 14 ROIs) as the three GEMMs the step runs, isolated:
forward [M x 256] . [256 x 1024] with the pixel-shuffle store, data gradient [M x 1024] . [1024 x 256], weight gradient
[256 x M] . [M x 1024]; 210.5 GFLOP each at 2048 ROIs (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
x = torch.randn(N, 14, 14, 256, device=dev)
wg = torch.randn(256, 1024, device=dev) * 0.05
b = torch.zeros(256, device=dev)
up = torch.empty(N, 28, 28, 256, device=dev)
fl = 2.0 * N * 196 * 256 * 1024
t = timed(lambda: ops.deconv2x2(x, wg, b, 1, out=up))
print("forward (pixel-shuffle store)  %.3f ms  %.1f TFLOP/s" % (t, fl / t / 1e9))
dzg = torch.randn(N, 14, 14, 1024, device=dev)
wt = torch.randn(1, 1, 1024, 256, device=dev) * 0.05
out = torch.empty(N, 14, 14, 256, device=dev)
t = timed(lambda: ops.conv2d(dzg, wt, None, None, None, out=out, stride=1, padding="valid"))
print("data gradient (plain 1x1)      %.3f ms  %.1f TFLOP/s" % (t, fl / t / 1e9))
dw = torch.empty(1, 1, 256, 1024, device=dev)
t = timed(lambda: ops.conv2d_wgrad(x, dzg, (1, 1, 256, 1024), 1, "valid", dw=dw))
print("weight gradient                %.3f ms  %.1f TFLOP/s" % (t, fl / t / 1e9))
# the same forward product through the persistent GEMM (no bias / ReLU / shuffle: what the matrix part could run at)
lib = ops._hip.lib()
Mt = torch.empty(N * 196, 1024, device=dev)
t = timed(lambda: lib.mrcnn_winograd_gemm(ops.ptr(x), ops.ptr(wg), ops.ptr(Mt), 1, N * 196, 256, 1024, ops.current_stream()))
print("forward product, persistent GEMM (plain store) %.3f ms  %.1f TFLOP/s" % (t, fl / t / 1e9))
