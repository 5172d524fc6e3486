#!/usr/bin/env python3
"""A Winograd mask-head layer (2048 ROIs, mixed tiling) forward and data gradient: separate launches against the output transform
fused into the main group's GEMM launch (mrcnn_winograd_gemm_fused).  MRCNN_WINOGRAD_FUSE_NORELEASE=1: the fused form without
its per-tile release fence (timing only: what the hand-off costs).  tools only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
x = torch.randn((N, 14, 14, 256), device=dev)
w = torch.randn((3, 3, 256, 256), device=dev) * 0.02
b = torch.zeros(256, device=dev); sc = torch.ones(256, device=dev); sh = torch.zeros(256, device=dev)
mean = torch.zeros(256, device=dev); rstd = torch.ones(256, device=dev)
U = ops.winograd_weights(w, tile=ops.TILE_MIXED)
z = torch.empty_like(x); out = torch.empty_like(x)
sums = [torch.zeros(256, device=dev) for _ in range(3)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for fused in (False, True, False, True):
    ops._WINO_FUSE = fused
    f = timed(lambda: ops.conv2d_winograd(x, U, b, sc, sh, 1, out=out, z_out=z))
    d = timed(lambda: ops.conv2d_dgrad_ep_winograd(x, U, None, z, sc, mean, rstd, sums[0], sums[1], sums[2], 1, fwd_shift=sh))
    print("fused=%-5s forward layer %.3f ms   data gradient + epilogue backward %.3f ms" % (fused, f, d), flush=True)
