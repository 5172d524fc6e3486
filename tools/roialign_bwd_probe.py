#!/usr/bin/env python3
"""Is the mask head's ROIAlign adjoint (scatter form) data dependent?  Times every pool-14 call of a few training steps
alone (synchronised) and prints how many ROI rows carry gradient, non-finite values and the ROI box sizes (tools only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from caesar_mrcnn_amd import ops
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
imgsize = int(os.environ.get("MRCNN_IMGSIZE", "512"))
dev = torch.device("cuda", 0)
cfg = run_py_config(num_classes=4, imgsize=imgsize, backbone="resnet101", images_per_gpu=4, gpu_count=1)
model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
inp = model._to_device(bench.synthetic_batch(cfg, 4, seed=1234))
eng = model.engine
eng.sparse_mask_bwd = False
eng.head_dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": None}[os.environ.get("MRCNN_PROBE_DTYPE", "f16")]
lr = float(os.environ.get("MRCNN_LR", cfg.LEARNING_RATE))
orig = ops.roialign_bwd
log = []
def probe(boxes, dout, dfms, pool, image_area, dense=False):
    if pool != cfg.MASK_POOL_SIZE:
        return orig(boxes, dout, dfms, pool, image_area, dense)
    torch.cuda.synchronize()
    rows = dout.reshape(dout.shape[0] * dout.shape[1], -1)
    nz = int((rows != 0).any(dim=1).sum()); bad = int((~torch.isfinite(rows)).sum())
    bh = (boxes[..., 2] - boxes[..., 0]).flatten(); bw = (boxes[..., 3] - boxes[..., 1]).flatten()
    live = (rows != 0).any(dim=1)
    t0 = time.time(); orig(boxes, dout, dfms, pool, image_area, dense); torch.cuda.synchronize()
    log.append((nz, bad, (time.time() - t0) * 1e3, float((bh * bw)[live].mean()) if nz else 0.0, float(rows.abs().max())))
ops.roialign_bwd = probe
import caesar_mrcnn_amd.engine as E
for step in range(int(os.environ.get("MRCNN_PROBE_STEPS", "60"))):
    losses = eng.forward_backward(*inp); eng.apply_gradients(lr, cfg.LEARNING_MOMENTUM, 1)
    nz, bad, ms, area, mx = log[-1]
    if step < 6 or step % 5 == 0:
        print("step %3d  rows with gradient %4d  non-finite %d  adjoint alone %.3f ms  mean live box area %.4f  max |d| %.3g  losses %s" %
              (step, nz, bad, ms, area, mx, " ".join("%.3f" % float(l) for l in torch.as_tensor(losses).flatten().tolist()) if not isinstance(losses, dict) else ""), flush=True)
