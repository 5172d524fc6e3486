#!/usr/bin/env python3
"""Knock-out timing: run the training step with one kernel family skipped (results are then wrong -- timing only) to see
how much of its duration sits on the step's critical path.  tools only.
usage: knockout.py <mode> <what>   mode: dense-f32 | sparse-f32 | sparse-f16 ...   what: none | roibwd7 | roibwd14 | wgrad_small | wgrad_all | comma list of ops.* function names (memoised after the first step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from caesar_mrcnn_amd import ops
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
mode = sys.argv[1] if len(sys.argv) > 1 else "dense-f32"
what = sys.argv[2] if len(sys.argv) > 2 else "none"
nimg = 4
dev = torch.device("cuda", 0)
cfg = run_py_config(num_classes=4, imgsize=int(os.environ.get("MRCNN_IMGSIZE", "256")), backbone="resnet101", images_per_gpu=nimg, gpu_count=1)
model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
inp = model._to_device(bench.synthetic_batch(cfg, nimg, seed=1234))
eng = model.engine
eng.sparse_mask_bwd = mode.startswith("sparse")
eng.head_dtype = {"f32": None, "f16": torch.float16, "bf16": torch.bfloat16}[mode.split("-")[1]]
orig_roibwd, orig_wgrad = ops.roialign_bwd, ops.conv2d_wgrad
if what.startswith("roibwd"):
    P = int(what[6:])
    ops.roialign_bwd = lambda boxes, dout, dfms, pool, area, dense=False: None if pool == P else orig_roibwd(boxes, dout, dfms, pool, area, dense)
elif what.startswith("wgrad"):
    def wg(x, dz, wshape, stride, padding, dw=None, accumulate=False):
        flops = 2.0 * dz.numel() * wshape[0] * wshape[1] * wshape[2]
        if what == "wgrad_all" or flops < 50e9:
            return dw
        return orig_wgrad(x, dz, wshape, stride, padding, dw=dw, accumulate=accumulate)
    ops.conv2d_wgrad = wg
else:
    # generic: memoise the named ops functions by call order within a step -- after the first (warm-up) step they
    # return the tensors of that step without launching anything
    names = [n for n in what.split(",") if n and n != "none"]
    state = {"k": 0, "cache": {}, "live": False}
    def memo(name, fn):
        def wrapped(*a, **kw):
            key = (name, state["k"]); state["k"] += 1
            ar = ops._arena
            if state["live"] and key in state["cache"]:
                r, delta = state["cache"][key]
                if ar is not None and ar.active:
                    ar.pos += delta                       # keep the step arena's request order aligned
                return r
            p0 = ar.pos if ar is not None and ar.active else 0
            r = fn(*a, **kw)
            state["cache"][key] = (r, (ar.pos - p0) if ar is not None and ar.active else 0)
            return r
        return wrapped
    for n in names:
        setattr(ops, n, memo(n, getattr(ops, n)))
    orig_fb = eng.forward_backward
    def fb(*a, **kw):
        state["k"] = 0
        r = orig_fb(*a, **kw)
        state["live"] = True
        return r
    eng.forward_backward = fb
def steps(n):
    for _ in range(n):
        eng.forward_backward(*inp); eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
steps(3)
torch.cuda.synchronize(); t0 = time.time()
steps(10)
torch.cuda.synchronize()
print("%s  knock-out %-12s %.2f ms/step" % (mode, what, (time.time() - t0) * 100), flush=True)
