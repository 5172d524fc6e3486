#!/usr/bin/env python3
"""cProfile of the host side of training steps (tools only)."""
import cProfile, os, pstats, sys
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
inp = model._to_device(bench.synthetic_batch(cfg, nimg, seed=1234))
eng = model.engine
eng.sparse_mask_bwd = False
def steps(n):
    for _ in range(n):
        eng.forward_backward(*inp); eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
steps(3)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
steps(5)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
