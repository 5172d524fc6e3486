#!/usr/bin/env python3
"""Runs N detect passes (batch 1, R50 256^2) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = run_py_config(backbone=sys.argv[2] if len(sys.argv) > 2 else "resnet50", imgsize=256, mode="inference")
m = MaskRCNN("inference", cfg, "/tmp/x", device=torch.device("cuda:0"))
x = torch.rand(1, 256, 256, 3, device="cuda") * 255
w = torch.tensor([[0., 0., 1., 1.]], device="cuda")
for _ in range(n):
    m.engine.infer(x, w)
torch.cuda.synchronize()
