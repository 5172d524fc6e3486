#!/usr/bin/env python3
"""The 16 (tile 2) or 36 (tile 4) transform-domain GEMMs of a Winograd layer: one-tile-per-workgroup LDS-DMA kernel against the
persistent form (tools only).  usage: wino_gemm_bench.py [ROIs] [tile]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
lib = ops._hip.lib()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
C_ = 256
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nb = (tile + 2) ** 2
nv = lib.mrcnn_winograd_buffer_floats(N, 14, 14, C_, tile)
rows = nv // (nb * C_)
V = torch.randn(nv, device=dev); U = torch.randn(nb, C_, C_, device=dev) * 0.05
M1 = torch.empty(nv, device=dev); M2 = torch.empty(nv, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
st, P = ops.current_stream, ops.ptr
f1 = lambda: lib.mrcnn_gemm_batched_f32(P(V), P(U), P(M1), nb, rows, C_, C_, st())
f2 = lambda: lib.mrcnn_winograd_gemm(P(V), P(U), P(M2), nb, rows, C_, C_, st())
assert f1() == 0 and f2() == 0
torch.cuda.synchronize()
print("identical results:", bool(torch.equal(M1, M2)))
fl = 2.0 * nb * N * ((14 + tile - 1) // tile) ** 2 * C_ * C_
for name, f in (("one tile per workgroup", f1), ("persistent", f2), ("one tile per workgroup", f1), ("persistent", f2)):
    ms = timed(f)
    print("%-24s %.3f ms  %.1f TFLOP/s (%.3f of 157.3)" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3))
