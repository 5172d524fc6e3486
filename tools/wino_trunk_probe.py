#!/usr/bin/env python3
"""Would the Winograd path pay on the trunk's large 3x3 layers (RPN shared convolution and FPN smoothing on P2 / P3)?
Direct kernel (as the engine's multi-problem launch runs it, here alone) against conv2d_winograd F(2x2,3x3), forward only,
with the result's deviation from the direct kernel.  tools only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
from caesar_mrcnn_amd._hip import ACT_RELU, ACT_NONE
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for tag, N, S, cout in (("RPN shared P2 256^2 x4", 4, 64, 512), ("FPN p2 256^2 x4", 4, 64, 256), ("RPN shared P3 256^2 x4", 4, 32, 512),
                        ("RPN shared P2 256^2 x1 (detect)", 1, 64, 512), ("RPN shared P2 512^2 x4", 4, 128, 512), ("FPN p2 512^2 x4", 4, 128, 256),
                        ("RPN shared P2 1024^2 x1", 1, 256, 512), ("RPN shared P3 512^2 x4", 4, 64, 512)):
    x = torch.randn((N, S, S, 256), device=dev)
    w = torch.randn((3, 3, 256, cout), device=dev) * 0.02
    b = torch.randn(cout, device=dev) * 0.1
    act = ACT_RELU if cout == 512 else ACT_NONE
    d_ms = timed(lambda: ops.conv2d(x, w, b, act=act))
    ref = ops.conv2d(x, w, b, act=act)
    line = "%-34s direct %.3f ms (%.1f TFLOP/s)" % (tag, d_ms, 2.0 * N * S * S * 2304 * cout / d_ms / 1e9)
    for tile in (2, 4):
        if tile == 4 and S % 4:
            continue
        U = ops.winograd_weights(w, tile=tile)
        try:
            w_ms = timed(lambda: ops.conv2d_winograd(x, U, b, None, None, act))
            got = ops.conv2d_winograd(x, U, b, None, None, act)
            err = float((got - ref).abs().max() / ref.abs().max())
            line += " | F(%dx%d) %.3f ms err %.1e" % (tile, tile, w_ms, err)
        except Exception as e:
            line += " | F(%dx%d) failed: %r" % (tile, tile, e)
    print(line, flush=True)
