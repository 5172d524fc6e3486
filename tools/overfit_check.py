#!/usr/bin/env python3
"""Sanity beyond gradient parity: the full step (fwd + bwd + clip + SGD-momentum) overfits one synthetic batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
hd = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda", 0)
cfg = run_py_config(num_classes=4, imgsize=256, backbone="resnet50", images_per_gpu=2, gpu_count=1)
model = MaskRCNN("training", cfg, "/tmp/mrcnn_overfit", device=dev, seed=0)
lr = 0.002
model.compile(lr, cfg.LEARNING_MOMENTUM)
if hd:
    model.engine.head_dtype = getattr(torch, hd)
inp = model._to_device(bench.synthetic_batch(cfg, 2, seed=7), rand_keys=np.random.RandomState(0).uniform(0, 1, (2, cfg.POST_NMS_ROIS_TRAINING)))
eng = model.engine
for s in range(steps):
    losses = eng.forward_backward(*inp)
    eng.apply_gradients(lr, cfg.LEARNING_MOMENTUM, 1)
    if s % 10 == 0 or s == steps - 1:
        l = losses.cpu().numpy()
        print("step %3d  total %.4f  rpn_cls %.4f rpn_box %.4f cls %.4f box %.4f mask %.4f" % (s, l.sum(), *l), flush=True)
