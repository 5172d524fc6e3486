#!/usr/bin/env python3
"""Per-tensor gradient error table of one full-size training step vs the oracle's autograd (tools only).
usage: grad_diag.py [backbone] [nimg] [damp] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import mrcnn_oracle as orc
import test_engine_gpu as T
from caesar_mrcnn_amd.model import MaskRCNN
backbone = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
damp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 53
dev = torch.device("cuda:0")
cfg = T._full_cfg(backbone, 256, nimg=B)
w = T._weights(cfg, seed, damp=damp if damp > 0 else None)
inputs, keys = T._train_inputs(cfg, B, seed + 4)
images, meta, rpn_match, rpn_bbox_t, gt_cls, gt_boxes, gt_masks = inputs
model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
eng = model.engine
eng.sparse_mask_bwd = os.environ.get("SPARSE", "0") == "1"
losses = model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
torch.cuda.synchronize()
last = {k: v.cpu().numpy() for k, v in eng.last.items() if torch.is_tensor(v)}
eng.apply_gradients(0.0, 0.0, world_size=1)
torch.cuda.synchronize()
g = eng.get_weights(grads=True)
print("counts", last["counts"].tolist(), "losses", losses.cpu().numpy().tolist(), flush=True)
o = orc.OracleMaskRCNN(cfg, w, requires_grad=True)
forced = {k: last[k] for k in ("rois", "target_class_ids", "target_bbox", "target_mask")}
ref = o.forward_training(images, rpn_match, rpn_bbox_t.astype(np.float32), gt_cls, gt_boxes, gt_masks,
                         meta[:, 12:].astype(np.int32), orc.get_anchors(cfg, images.shape[1:]), keys, forced=forced)
print("ref losses", [float(l.detach()) for l in ref["losses"]], flush=True)
o.total_loss(ref["losses"]).backward()
rows = []
for name in eng.layout.offsets:
    rg = o.w[name].grad.numpy().astype(np.float64)
    d = g[name].astype(np.float64) - rg
    mx = max(np.abs(rg).max(), 1e-12)
    rows.append((np.abs(d).max() / mx, np.linalg.norm(d) / max(np.linalg.norm(rg), 1e-12), int((np.abs(d) > 1e-3 * mx).sum()), rg.size, mx, name))
rows.sort(reverse=True)
print("%-32s %10s %10s %8s %8s %10s" % ("tensor", "max/max", "l2rel", "n>1e-3", "size", "max|ref|"))
for r in rows[:40]:
    print("%-32s %10.3g %10.3g %8d %8d %10.3g" % (r[5], r[0], r[1], r[2], r[3], r[4]))
print("tensors over 5e-3:", sum(r[0] > 5e-3 for r in rows), "of", len(rows))
