#!/usr/bin/env python3
"""Cycle stamps of the phased 16-bit forward kernel (tools only): workgroup 0, waves 0 (group 0) and 4 (group 1), one stamp
at the start of every phase.  Prints the phase-to-phase deltas around the first tile borders."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import caesar_mrcnn_amd  # noqa
from caesar_mrcnn_amd import ops
dev = torch.device("cuda:0")
N, Cin, k = 2048, 256, 3
dtype = torch.bfloat16
x = torch.randn(N, 14, 14, Cin, device=dev).to(dtype)
w = torch.randn(k, k, Cin, 256, device=dev) * 0.02
wf, wd = ops.weights_to_h16(w, dtype)
b = torch.zeros(256, device=dev); sc = torch.ones(256, device=dev)
out = torch.empty(N, 14, 14, 256, device=dev, dtype=dtype)
os.environ["MRCNN_H16_TILE"] = "phase"
for _ in range(3):
    ops.conv2d_h16(x, wf, (k, k, Cin, 256), b, sc, b, 1, "same", 1, out=out)
dbg = torch.zeros(1024, dtype=torch.int64, device=dev)
os.environ["MRCNN_H16P_TRACE"] = str(dbg.data_ptr())
os.environ["MRCNN_H16P_NO_SPLIT"] = "1"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.conv2d_h16(x, wf, (k, k, Cin, 256), b, sc, b, 1, "same", 1, out=out)
e1.record()
torch.cuda.synchronize()
ms_traced = e0.elapsed_time(e1)
del os.environ["MRCNN_H16P_TRACE"]
d = dbg.cpu().numpy().reshape(2, 512)
nk2 = k * k * Cin // 64
rounds = -(-((N * 196 + 255) // 256) // 256)
print("traced launch: %.3f ms for %d rounds = %.1f us per tile" % (ms_traced, rounds, ms_traced * 1e3 / rounds))
for g in range(2):
    st = d[g][d[g] > 0]
    dl = st[1:] - st[:-1]
    print("group %d: %d stamps, %d cycles in all; median phase %d cycles" % (g, len(st), st[-1] - st[0], sorted(dl)[len(dl) // 2]))
    per_tile = nk2 * 4
    for tile in range(min(3, len(dl) // per_tile + 1)):
        seg = dl[tile * per_tile:(tile + 1) * per_tile]
        if len(seg):
            if tile == 1:
                print("  -> shader clock while it ran: %.2f GHz" % (seg.sum() / (ms_traced * 1e6 / rounds)))
            print("  tile %d: %d cycles; first phases: %s ... last 12: %s" % (tile, seg.sum(), " ".join("%d" % v for v in seg[:6]),
                                                                               " ".join("%d" % v for v in seg[-12:])))

