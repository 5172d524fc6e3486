#!/usr/bin/env python3
"""bench.py's feed-inclusive train-loop leg alone (loader threads -> H2D -> step), with and without the launch tape
(Config.TRAIN_LAUNCH_TAPE): how much of the loop is the main thread's Python (tools only).  usage: train_loop_probe.py [steps]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
for tape in (False, True, False, True):
    cfg = run_py_config(num_classes=4, imgsize=256, backbone="resnet101", images_per_gpu=4, gpu_count=1)
    cfg.TRAIN_LAUNCH_TAPE = tape
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
    model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
    res = {}
    bench._train_loop_leg(argparse.Namespace(steps=steps), res, model, cfg, 4, 1, 0)
    t = res["train_loop"]
    print("tape=%s dense %.1f images/s (%.2f ms/step, waiting %.2f), positive quota %.1f images/s (%.2f ms/step, waiting %.2f)" % (
        tape, t["images_per_s"], t["ms_per_step"], t["ms_per_step_waiting_for_loader"], t["exact_zero_skip"]["images_per_s"],
        t["exact_zero_skip"]["ms_per_step"], t["exact_zero_skip"]["ms_per_step_waiting_for_loader"]), flush=True)
    del model
    torch.cuda.empty_cache()
