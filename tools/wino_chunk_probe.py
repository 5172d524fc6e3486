#!/usr/bin/env python3
"""Does the Infinity Cache keep V / Mt of a Winograd layer when the ROI batch is processed in chunks?  Times one layer whole
and in chunks of R ROIs (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
dev = torch.device("cuda:0")
N, Cc = 2048, 256
x = torch.randn(N, 14, 14, Cc, device=dev); w = torch.randn(3, 3, Cc, Cc, device=dev) * 0.03
b = torch.zeros(Cc, device=dev); sc = torch.ones(Cc, device=dev)
U = ops.winograd_weights(w)
y = torch.empty_like(x); z = torch.empty_like(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for R in (2048, 1024, 512, 256, 128, 64):
    def layer():
        for a in range(0, N, R):
            ops.conv2d_winograd(x[a:a + R], U, b, sc, b, 1, out=y[a:a + R], z_out=z[a:a + R])
    print("chunks of %4d ROIs (%2d x 3 launches, V + Mt %4d MB per chunk): %.3f ms" % (R, N // R, 2 * 16 * R * 49 * Cc * 4 >> 20, timed(layer)), flush=True)
