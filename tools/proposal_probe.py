#!/usr/bin/env python3
"""ProposalLayer alone (one image, A = 16 368 anchors of a 256 x 256 input, top 6 000, 1 000 proposals) for foreground scores of
different spread: uniform in [0, 1], clustered around 0.5 (a random-init network), mostly near 0 with a few confident ones (a
trained one).  Per-kernel times come from rocprofv3 --kernel-trace --stats on this script (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from caesar_mrcnn_amd import ops
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd import utils
dev = torch.device("cuda:0")
cfg = run_py_config(imgsize=256, mode="inference")
anchors = utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, cfg.compute_backbone_shapes(cfg.IMAGE_SHAPE)
                                         if hasattr(cfg, "compute_backbone_shapes") else utils.compute_backbone_shapes(cfg, cfg.IMAGE_SHAPE),
                                         cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
anchors_n = torch.tensor(utils.norm_boxes(anchors, (256, 256)).astype(np.float32), device=dev)
A = anchors_n.shape[0]
rng = np.random.default_rng(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fg in (("uniform [0, 1]", rng.uniform(0, 1, A)), ("clustered 0.5 +- 0.01", 0.5 + 0.01 * rng.standard_normal(A)),
                 ("trained-like", np.where(rng.uniform(0, 1, A) < 0.02, rng.uniform(0.5, 1, A), rng.uniform(0, 0.05, A) ** 2))):
    fg = fg.astype(np.float32)
    probs = torch.tensor(np.stack([1 - fg, fg], -1)[None], device=dev)
    deltas = torch.tensor((rng.standard_normal((1, A, 4)) * 0.5).astype(np.float32), device=dev)
    f = lambda: ops.proposals(probs, deltas, anchors_n, 6000, 1000, 0.7, cfg.RPN_BBOX_STD_DEV)
    for _ in range(3): f()
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("%-24s %.1f us per call" % (name, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
