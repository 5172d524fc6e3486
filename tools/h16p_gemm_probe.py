#!/usr/bin/env python3
"""The phased kernel as a plain GEMM (1x1 convolution, one 256 x 256 tile per workgroup, long K) against its mask-head shape:
separates the main loop's rate from tile borders / epilogue / implicit-GEMM addressing (tools only)."""
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
os.environ["MRCNN_H16_TILE"] = "phase"
for dtype in (torch.float16, torch.bfloat16):
    for (N, H, W, Cin, Cout, k, what) in ((4, 128, 128, 4096, 256, 1, "GEMM 65536 x 256 x 4096, one tile per workgroup"),
                                          (16, 128, 128, 2048, 256, 1, "GEMM 262144 x 256 x 2048, four tiles per workgroup"),
                                          (4, 128, 128, 256, 256, 3, "3x3 conv 65536 px, one tile per workgroup"),
                                          (2048, 14, 14, 256, 256, 3, "mask-head 3x3 conv, 6.1 tiles per workgroup")):
        x = torch.randn(N, H, W, Cin, device=dev).to(dtype)
        w = torch.randn(k, k, Cin, Cout, device=dev) * 0.02
        wf, _ = ops.weights_to_h16(w, dtype, want_dgrad=False)
        b = torch.zeros(Cout, device=dev)
        out = torch.empty(N, H, W, Cout, device=dev, dtype=dtype)
        pad = "same" if k == 3 else "valid"
        ms = timed(lambda: ops.conv2d_h16(x, wf, (k, k, Cin, Cout), b, None, None, 1, pad, 1, out=out))
        fl = 2.0 * N * H * W * Cout * k * k * Cin
        print("%s %-55s %.3f ms  %.0f TFLOP/s" % (str(dtype)[6:], what, ms, fl / ms / 1e9), flush=True)
        del x, out
