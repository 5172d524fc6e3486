#!/usr/bin/env python3
"""Times the 16-bit conv kernels on the mask-head shape (tools only): forward in both tilings, weight gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import caesar_mrcnn_amd  # noqa
from caesar_mrcnn_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
fl = 2.0 * N * 196 * 256 * 2304
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for dtype in (torch.float16, torch.bfloat16):
    x = torch.randn(N, 14, 14, 256, device=dev).to(dtype)
    w = torch.randn(3, 3, 256, 256, device=dev) * 0.02
    wf, wd = ops.weights_to_h16(w, dtype)
    b = torch.zeros(256, device=dev); sc = torch.ones(256, device=dev)
    out = torch.empty(N, 14, 14, 256, device=dev, dtype=dtype)
    for tile in ("small", "big", "phase", "phase-noslab", "phase-nosplit") * 2:       # interleaved rounds in one process
        os.environ["MRCNN_H16_TILE"] = tile.split("-")[0]
        ops.tuning_set("h16_slab", 0 if "noslab" in tile else 1)        # slab-plus-halo staging (conv_fwd_h16q_kernel) vs per-tap staging
        os.environ.pop("MRCNN_H16P_NO_SPLIT", None)
        os.environ.pop("MRCNN_H16P_GRID", None)
        if "nosplit" in tile:
            os.environ["MRCNN_H16P_NO_SPLIT"] = "1"
        if "-g" in tile:
            os.environ["MRCNN_H16P_GRID"] = tile.split("-g")[1]
        ms = timed(lambda: ops.conv2d_h16(x, wf, (3, 3, 256, 256), b, sc, b, 1, "same", 1, out=out))
        print("%s fwd tile=%-22s N=%d: %.3f ms  %.1f TFLOP/s" % (dtype, tile, N, ms, fl / ms / 1e9), flush=True)
    del os.environ["MRCNN_H16_TILE"]
    ops.tuning_set("h16_slab", -1)
    dy = torch.randn(N, 14, 14, 256, device=dev).to(dtype)
    dw = torch.empty(3, 3, 256, 256, device=dev)
    ms = timed(lambda: ops.conv2d_wgrad_h16(x, dy, (3, 3, 256, 256), 1, "same", dw=dw))
    print("%s wgrad N=%d: %.3f ms  %.1f TFLOP/s (incl. table + slab reduce)" % (dtype, N, ms, fl / ms / 1e9), flush=True)
