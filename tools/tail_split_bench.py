#!/usr/bin/env python3
"""float32 LDS-DMA convolution on M = 200 704 (ResNet-50, 2 images: 2.45 rounds of 128 x 128 tiles), forward and the data
gradient fused with the backward epilogue; run with MRCNN_CONV_TAIL_SPLIT=0 / 1 (read once per process) (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(N, 14, 14, 256, device=dev); w = torch.randn(3, 3, 256, 256, device=dev) * 0.02
b = torch.zeros(256, device=dev); sc = torch.ones(256, device=dev)
out = torch.empty_like(x); z = torch.empty_like(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
fl = 2.0 * N * 196 * 256 * 2304
ms = timed(lambda: ops.conv2d(x, w, b, sc, b, act=1, out=out, z_out=z))
print("tail_split=%s fwd N=%d: %.3f ms  %.1f TFLOP/s (%.1f %% of 157.3)" % (os.environ.get("MRCNN_CONV_TAIL_SPLIT", "1"), N, ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573))
sums = [torch.zeros(256, device=dev) for _ in range(3)]
ms = timed(lambda: ops.conv2d_dgrad_ep(x, w, (1, 1), out, z, sc, b, sc, sums[0], sums[1], sums[2], 1))
print("tail_split=%s dgrad+epilogue-backward: %.3f ms  %.1f TFLOP/s" % (os.environ.get("MRCNN_CONV_TAIL_SPLIT", "1"), ms, fl / ms / 1e9))
