#!/usr/bin/env python3
"""K / M sweep of the phased 16-bit forward kernel (tools only): fixed cost per workgroup vs cost per K-step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import caesar_mrcnn_amd  # noqa
from caesar_mrcnn_amd import ops
dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tiles = (os.environ.get("TILES") or "phase").split(",")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dtype = torch.bfloat16
for N, Cin, k in ((2048, 256, 3), (2048, 512, 3), (2048, 1024, 3), (2048, 256, 1), (2006, 256, 3), (1672, 256, 3), (334, 256, 3), (335, 256, 3),
                  (669, 256, 3), (1338, 256, 3)):
    x = torch.randn(N, 14, 14, Cin, device=dev).to(dtype)
    w = torch.randn(k, k, Cin, 256, device=dev) * 0.02
    wf, wd = ops.weights_to_h16(w, dtype)
    b = torch.zeros(256, device=dev); sc = torch.ones(256, device=dev)
    out = torch.empty(N, 14, 14, 256, device=dev, dtype=dtype)
    fl = 2.0 * N * 196 * 256 * k * k * Cin
    for tile in tiles:
        os.environ["MRCNN_H16_TILE"] = tile
        if tile == "default":
            del os.environ["MRCNN_H16_TILE"]
        ms = timed(lambda: ops.conv2d_h16(x, wf, (k, k, Cin, 256), b, sc, b, 1, "same" if k == 3 else "valid", 1, out=out))
        wgs = (N * 196 + 255) // 256
        print("%-6s N=%d Cin=%d k=%d: %d workgroups (%.3f rounds), %d K-steps: %.3f ms  %.1f TFLOP/s" %
              (tile, N, Cin, k, wgs, wgs / 256.0, k * k * Cin // 64, ms, fl / ms / 1e9), flush=True)
