#!/usr/bin/env python3
"""Stability soak (tools only): several hundred full steps of the headline workload with fresh synthetic batches, all
three stream-heavy modes; asserts finite losses / parameters throughout.  usage: soak.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
cfg = run_py_config(num_classes=4, imgsize=int(os.environ.get("MRCNN_IMGSIZE", "256")), backbone="resnet101", images_per_gpu=4, gpu_count=1)
model = MaskRCNN("training", cfg, "/tmp/mrcnn_soak", device=dev, seed=0)
model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
eng = model.engine
batches = [model._to_device(bench.synthetic_batch(cfg, 4, seed=100 + i)) for i in range(6)]
modes = ((False, None), (True, None), (True, torch.float16), (False, torch.float16), (False, torch.bfloat16))
if os.environ.get("MRCNN_SOAK_H16_ONLY"):
    modes = modes[2:]
for sparse, hd in modes:
    eng.sparse_mask_bwd, eng.head_dtype = sparse, hd
    t0 = time.time(); worst = 0.0
    for s in range(steps):
        losses = eng.forward_backward(*batches[s % len(batches)])
        eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
        if s % 25 == 24 or s == steps - 1:
            l = losses.cpu().numpy()
            assert np.all(np.isfinite(l)), (sparse, hd, s, l)
            worst = max(worst, float(l.sum()))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(eng.params).all()), (sparse, hd)
    assert eng.skipped_step_count() == 0, "float16 steps skipped: %d" % eng.skipped_step_count()
    print("sparse=%s head=%s: %d steps ok, %.1f ms/step, last total loss %.3f (max seen %.3f)" % (
        sparse, hd, steps, (time.time() - t0) / steps * 1e3, float(l.sum()), worst), flush=True)
