"""Where does the run-to-run gradient noise of the float16 mode sit?  Same weights, same batch, same proposals, two eager
evaluations of the gradient: relative L2 per parameter tensor, for both test batches."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_engine_gpu as T
from caesar_mrcnn_amd.model import MaskRCNN

dev = torch.device("cuda:0")
cfg = T._small_cfg("resnet50", 128)
w = T._weights(cfg, 71, damp=0.5)
batches = [T._train_inputs(cfg, 2, 73), T._train_inputs(cfg, 2, 75)]
model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
eng = model.engine
forced = []
for inputs, keys in batches:
    model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
    forced.append(eng.last["rpn_rois"].clone())
eng.forced_rpn_rois = torch.empty_like(forced[0])
for dtype in (torch.float16, torch.bfloat16, None):
    eng.head_dtype = dtype
    for b, (inputs, keys) in enumerate(batches):
        eng.forced_rpn_rois.copy_(forced[b])
        gs, ls = [], []
        for rep in range(3):
            di = model._to_device(inputs, keys)
            l = eng.forward_backward(*di)
            torch.cuda.synchronize()
            gs.append(eng.grads.cpu().numpy().copy()); ls.append(l.cpu().numpy().copy())
        tot = [float(np.linalg.norm(gs[r] - gs[0]) / np.linalg.norm(gs[0])) for r in (1, 2)]
        print("dtype %s batch %d: losses %s  whole-gradient rel L2 between evaluations: %s" % (dtype, b, ls[0].tolist(), tot))
        worst = []
        for (name, off, n, _, _) in eng.layout.segments:
            a, c = gs[0][off:off + n], gs[1][off:off + n]
            na = float(np.linalg.norm(a))
            if na > 0:
                worst.append((float(np.linalg.norm(a - c)) / na, name, na))
        worst.sort(reverse=True)
        for e, name, na in worst[:6]:
            print("     %-32s rel L2 %.2e  (norm %.3e)" % (name, e, na))
