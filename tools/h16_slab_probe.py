#!/usr/bin/env python3
"""Mask-head 3x3 layer (2048 ROIs, float16) on the phased 16-bit kernel: per-tap staging (conv_fwd_h16p_kernel) against the slab-plus-halo
form (conv_fwd_h16q_kernel: a channel chunk's pixels staged once, taps as shifted LDS reads, border taps to a zero row) and against that
form with its border masks knocked out (MRCNN_H16P_TRACE set: timing only, wrong at the map borders).  Tools only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import caesar_mrcnn_amd
from caesar_mrcnn_amd import ops
N = 2048; dev = torch.device("cuda:0")
fl = 2.0 * N * 196 * 256 * 2304
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
dtype = torch.float16
x = torch.randn(N, 14, 14, 256, device=dev).to(dtype)
w = torch.randn(3, 3, 256, 256, device=dev) * 0.02
wf, wd = ops.weights_to_h16(w, dtype)
b = torch.zeros(256, device=dev); sc = torch.ones(256, device=dev)
out = torch.empty(N, 14, 14, 256, device=dev, dtype=dtype)
os.environ["MRCNN_H16_TILE"] = "phase"
trace = torch.zeros(2048, dtype=torch.int64, device=dev)
for rep in range(2):
    for name, slab, dbg in (("phased", 0, False), ("slab", 1, False), ("slab, no masks", 1, True)):
        ops.tuning_set("h16_slab", slab)
        if dbg: os.environ["MRCNN_H16P_TRACE"] = str(trace.data_ptr())
        else: os.environ.pop("MRCNN_H16P_TRACE", None)
        ms = timed(lambda: ops.conv2d_h16(x, wf, (3, 3, 256, 256), b, sc, b, 1, "same", 1, out=out))
        print("%-16s %.3f ms  %.1f TFLOP/s" % (name, ms, fl / ms / 1e9), flush=True)
