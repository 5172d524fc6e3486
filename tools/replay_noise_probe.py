"""How far do two runs of the SAME four float16 training steps drift apart (float atomics order differs between runs)?
Sets the bar of tests/test_engine_gpu.py::test_graphed_training_steps_equal_eager[float16-*]: eager vs eager, eager vs tape,
eager vs graph, all on fixed proposals (engine.forced_rpn_rois)."""
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
m0 = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
forced = []
for inputs, keys in batches:
    m0.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
    forced.append(m0.engine.last["rpn_rois"].clone())
del m0


def run(how, dtype):
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    eng = model.engine
    eng.head_dtype = dtype
    eng.forced_rpn_rois = torch.empty_like(forced[0])
    model.compile(0.002, 0.9)
    rec = []
    for s in range(4):
        inputs, keys = batches[s % 2]
        di = model._to_device(inputs, keys)
        eng.forced_rpn_rois.copy_(forced[s % 2])
        if how == "eager":
            ls = eng.forward_backward(*di)
            g = eng.grads.clone()
            eng.apply_gradients(0.002, 0.9, 1)
        else:
            ls = (eng.step_graphed if how == "graph" else eng.step_taped)(di, 0.002, 0.9)
            g = eng.grads.clone()
        torch.cuda.synchronize()
        rec.append((ls.cpu().numpy().copy(), g.cpu().numpy(), eng.momentum.cpu().numpy().copy(), eng.params.cpu().numpy().copy()))
    print(how, dtype, "skipped steps", eng.skipped_step_count())
    return rec


rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
for dtype in (torch.float16, None):
    base = run("eager", dtype)
    for how in ("eager", "tape", "graph"):
        other = run(how, dtype)
        for s in range(4):
            print("%s vs eager, %s, step %d: loss max rel %.2e  grads L2 %.2e  momentum L2 %.2e  params L2 %.2e" % (
                how, dtype, s, float(np.max(np.abs(other[s][0] - base[s][0]) / np.abs(base[s][0]))), rel(other[s][1], base[s][1]),
                rel(other[s][2], base[s][2]), rel(other[s][3], base[s][3])))
