#!/usr/bin/env python3
"""Fixed cost per 128 x 128 tile of the float32 LDS-DMA kernel: 1 x 1 convolutions of the same row count with K = 256 ... 2048
(tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 100352 // 4
for K in (128, 256, 512, 1024, 2048):
    x = torch.randn(rows, 1, 1, K, device=dev); w = torch.randn(1, 1, K, 256, device=dev) * 0.05
    out = torch.empty(rows, 1, 1, 256, device=dev)
    ms = timed(lambda: ops.conv2d(x, w, None, None, None, out=out, stride=1, padding="valid"))
    fl = 2.0 * rows * K * 256
    print("rows %d K %4d: %.3f ms  %.1f TFLOP/s  (%.2f us per K-step of 16 and 1280-tile round)" % (rows, K, ms, fl / ms / 1e9, ms * 1e3 / (K / 16) / (rows / 128 * 2 / 1280)), flush=True)
    del x, out
