#!/usr/bin/env python3
"""First call of the phased kernel in a fresh process on the one-tile-per-workgroup shape: dumps the wrong region (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, H, W, Cin, Cout = 4, 128, 128, 256, 256
dtype = torch.bfloat16 if os.environ.get("DT", "f16") == "bf16" else torch.float16
x = (torch.randn(N, H, W, Cin, device=dev) * 3).to(dtype)
w = torch.randn(3, 3, Cin, Cout, device=dev) / (3 * Cin ** 0.5)
wf, wd = ops.weights_to_h16(w, dtype)
b = torch.randn(Cout, device=dev) * 0.1
os.environ["MRCNN_H16_TILE"] = "phase"
ys = [ops.conv2d_h16(x, wf, (3, 3, Cin, Cout), b, None, None, 1, "same", 1) for _ in range(3)]
torch.cuda.synchronize()
os.environ["MRCNN_H16_TILE"] = "small"
ref = ops.conv2d_h16(x, wf, (3, 3, Cin, Cout), b, None, None, 1, "same", 1).float()
for k, y in enumerate(ys):
    d = ((y.float() - ref).abs() > 0.02) | ~torch.isfinite(y.float())
    idx = d.nonzero()
    if idx.shape[0] == 0:
        print("call %d ok" % k); continue
    print("call %d: %d wrong" % (k, idx.shape[0]))
    yf = y.float().view(-1, Cout); rf = ref.view(-1, Cout); dm = d.view(-1, Cout)
    for m in sorted(set((((i[0] * H + i[1]) * W + i[2]).item()) for i in idx)):
        cols = dm[m].nonzero().flatten().tolist()
        print("  m %6d (tile %3d row %3d): channels %s" % (m, m // 256, m % 256, cols))
        print("     got %s" % " ".join("%.3g" % v for v in yf[m, cols[0]:cols[-1] + 1].tolist()))
        print("     ref %s" % " ".join("%.3g" % v for v in rf[m, cols[0]:cols[-1] + 1].tolist()))
