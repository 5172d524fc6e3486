#!/usr/bin/env python3
"""Gradients of the 16-bit modes against the all-float32 engine on the same batch and proposals: per-tensor max and L2
error (tools only).  usage: h16_grad_diag.py [backbone] [size] [dtype]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import test_engine_gpu as T
from caesar_mrcnn_amd.model import MaskRCNN
backbone = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dtype = getattr(torch, sys.argv[3]) if len(sys.argv) > 3 else torch.float16
dev = torch.device("cuda:0")
cfg = T._small_cfg(backbone, size)
w = T._weights(cfg, 31, damp=0.5 if backbone != "custom" else None)
inputs, keys = T._train_inputs(cfg, 2, 33)
res, forced = {}, None
for mode in (None, dtype):
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    eng = model.engine
    eng.head_dtype = mode
    eng.forced_rpn_rois = forced
    losses = model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
    torch.cuda.synchronize()
    if mode is None:
        forced = eng.last["rpn_rois"].clone()
    res[mode] = (losses.cpu().numpy(), eng.get_weights(grads=True))
print("losses f32", res[None][0], "\nlosses h16", res[dtype][0])
rows = []
for name, ref in res[None][1].items():
    d = res[dtype][1][name].astype(np.float64) - ref
    rows.append((np.abs(d).max() / max(np.abs(ref).max(), 1e-12), np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-12), name, np.abs(ref).max()))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%-30s max %.4f  l2 %.4f  max|ref| %.3g" % (r[2], r[0], r[1], r[3]))
print("median max-err %.4f, median l2 %.4f" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
