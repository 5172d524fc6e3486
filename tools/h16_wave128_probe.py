#!/usr/bin/env python3
"""The 4-wave 128 x 128-per-wave 16-bit forward kernel (MRCNN_H16_TILE=wave128) against the 256 x 128 kernel: identical inputs,
outputs compared, then timed beside the phased kernel (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dtype in (torch.float16, torch.bfloat16):
    for N in (37, 2048):
        torch.manual_seed(N)
        x = torch.randn(N, 14, 14, 256, device=dev).to(dtype)
        w = torch.randn(3, 3, 256, 256, device=dev) * 0.02
        wf, wd = ops.weights_to_h16(w, dtype)
        b = torch.randn(256, device=dev) * 0.1; sc = torch.rand(256, device=dev) + 0.5; sh = torch.randn(256, device=dev) * 0.1
        res = {}
        for tile in ("small", "wave128"):
            os.environ["MRCNN_H16_TILE"] = tile
            out = torch.full((N, 14, 14, 256), 7.0, device=dev, dtype=dtype); z = torch.full_like(out, 7.0)
            ops.conv2d_h16(x, wf, (3, 3, 256, 256), b, sc, sh, 1, "same", 1, out=out, z_out=z)
            torch.cuda.synchronize()
            res[tile] = (out.float(), z.float())
        d_o = float((res["small"][0] - res["wave128"][0]).abs().max()); d_z = float((res["small"][1] - res["wave128"][1]).abs().max())
        print("%s N=%d: max |out diff| %.3g, max |z diff| %.3g (out max %.3g)" % (dtype, N, d_o, d_z, float(res["small"][0].abs().max())), flush=True)
    fl = 2.0 * 2048 * 196 * 256 * 2304
    out = torch.empty(2048, 14, 14, 256, device=dev, dtype=dtype)
    for tile in ("phase", "wave128", "phase", "wave128"):
        os.environ["MRCNN_H16_TILE"] = tile
        ms = timed(lambda: ops.conv2d_h16(x, wf, (3, 3, 256, 256), b, sc, sh, 1, "same", 1, out=out))
        print("%s tile=%-8s %.3f ms  %.1f TFLOP/s" % (dtype, tile, ms, fl / ms / 1e9), flush=True)
    # what the K loop's pieces cost: the same launch without its MFMAs and / or operand reads (results are garbage, timing only)
    os.environ["MRCNN_H16_TILE"] = "wave128"
    for dbg, what in (("1", "no MFMAs"), ("2", "no operand reads"), ("3", "neither: LDS-DMA staging, barriers and the epilogue only")):
        os.environ["MRCNN_H16W_DBG"] = dbg
        ms = timed(lambda: ops.conv2d_h16(x, wf, (3, 3, 256, 256), b, sc, sh, 1, "same", 1, out=out))
        print("%s wave128, %s: %.3f ms" % (dtype, what, ms), flush=True)
    del os.environ["MRCNN_H16W_DBG"]
