#!/usr/bin/env python3
"""detect latency (batch 1) for a backbone / image size, eager and HIP-graph replay (tools only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
backbone = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = run_py_config(backbone=backbone, imgsize=size, mode="inference")
m = MaskRCNN("inference", cfg, "/tmp/x", device=torch.device("cuda:0"))
if os.environ.get("MRCNN_HEAD_DTYPE"):                       # opt-in 16-bit mask head (float16 | bfloat16)
    m.engine.head_dtype = getattr(torch, os.environ["MRCNN_HEAD_DTYPE"])
x = torch.rand(1, size, size, 3, device="cuda") * 255
w = torch.tensor([[0., 0., 1., 1.]], device="cuda")
for fn, name in ((m.engine.infer, "eager"), (m.engine.infer_graphed, "graph")):
    for i in range(3):
        fn(x, w)
        torch.cuda.synchronize()
        print("warm", name, i, flush=True)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        fn(x, w)
    torch.cuda.synchronize()
    print("%s %dx%d detect %s: %.3f ms/image" % (backbone, size, size, name, (time.time() - t0) / 10 * 1e3))
