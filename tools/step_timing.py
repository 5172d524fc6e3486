#!/usr/bin/env python3
"""How much of a training step is host launch time?  Times the launch loop with and without the final
device synchronisation (tools only; not part of the product)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN

backbone = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
cfg = run_py_config(num_classes=4, imgsize=256, backbone=backbone, images_per_gpu=nimg, gpu_count=1)
model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
batch = bench.synthetic_batch(cfg, nimg, seed=1234)
inp = model._to_device(batch)
eng = model.engine
for sparse in (False, True):
    eng.sparse_mask_bwd = sparse
    for _ in range(3):
        eng.forward_backward(*inp); eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
    torch.cuda.synchronize()
    K = 10
    t0 = time.time()
    for _ in range(K):
        eng.forward_backward(*inp); eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print("sparse=%s host launch %.2f ms/step, total %.2f ms/step" % (sparse, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
    # per-step with sync after each: exposes host-bound phases
    t0 = time.time()
    for _ in range(K):
        eng.forward_backward(*inp); eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
        torch.cuda.synchronize()
    print("   synced every step: %.2f ms/step" % ((time.time() - t0) / K * 1e3))
