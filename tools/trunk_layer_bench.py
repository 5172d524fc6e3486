#!/usr/bin/env python3
"""One bottleneck layer of the trunk, float32 path (split-K conv + reduction launch) vs the 16-bit small-tile kernel
(tools only).  Shapes: res4 2a / 2b / 2c and res3 2b at 4 images of 256 x 256 and 512 x 512."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd import ops
dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


os.environ["MRCNN_H16_SMALL"] = "1"
for hw in (16, 32, 64):
    for name, cin, cout, k in (("2a", 1024, 256, 1), ("2b", 256, 256, 3), ("2c", 256, 1024, 1)):
        x = torch.randn(4, hw, hw, cin, device=dev)
        w = torch.randn(k, k, cin, cout, device=dev) * 0.02
        b = torch.zeros(cout, device=dev); sc = torch.ones(cout, device=dev)
        out = torch.empty(4, hw, hw, cout, device=dev)
        t32 = timed(lambda: ops.conv2d(x, w, b, sc, b, None, 1, "same" if k == 3 else "valid", 1, out=out))
        line = "M=%5d %s %4d->%4d k%d  f32 %6.1f us" % (4 * hw * hw, name, cin, cout, k, t32)
        for dt in (torch.float16,):
            xh = x.to(dt); wf, wd = ops.weights_to_h16(w, dt)
            oh = torch.empty(4, hw, hw, cout, device=dev, dtype=dt)
            t16 = timed(lambda: ops.conv2d_h16(xh, wf, (k, k, cin, cout), b, sc, b, 1, "same" if k == 3 else "valid", 1, out=oh))
            gf = 2.0 * 4 * hw * hw * cin * cout * k * k / 1e9
            line += "   f16 small %6.1f us (%5.1f TFLOP/s)" % (t16, gf / t16 * 1e3)
        print(line, flush=True)
