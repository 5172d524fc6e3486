#!/usr/bin/env python3
"""Times mrcnn_detection_fwd for few / many valid candidates (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import caesar_mrcnn_amd  # noqa
from caesar_mrcnn_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
R, C = 1000, 4
for name, conf in (("no valid", 2.0), ("few valid (minconf .7)", 0.7), ("all valid (minconf 0)", 0.0)):
    y1 = rng.uniform(0, 0.8, (1, R)); x1 = rng.uniform(0, 0.8, (1, R))
    rois = np.stack([y1, x1, y1 + rng.uniform(0.02, 0.2, (1, R)), x1 + rng.uniform(0.02, 0.2, (1, R))], -1).astype(np.float32)
    logits = rng.normal(0, 1.5, (1, R, C)).astype(np.float32)
    probs = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
    deltas = rng.normal(0, 0.1, (1, R, C, 4)).astype(np.float32)
    win = np.array([[0, 0, 1, 1]], np.float32)
    t = [torch.tensor(a, device=dev) for a in (rois, probs.astype(np.float32), deltas, win)]
    for _ in range(3):
        out = ops.detections(*t, 100, conf, 0.3, np.array([0.1, 0.1, 0.2, 0.2], np.float32))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ops.detections(*t, 100, conf, 0.3, np.array([0.1, 0.1, 0.2, 0.2], np.float32))
    e1.record(); torch.cuda.synchronize()
    print("%-26s %.1f us, detections %d" % (name, e0.elapsed_time(e1) / 20 * 1e3, int((out[0, :, 4] > 0).sum())))
