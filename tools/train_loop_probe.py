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
if os.environ.get("MRCNN_SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["MRCNN_SWITCH_INTERVAL"]))
dev = torch.device("cuda", 0)
combos = [(True, 2, True), (True, 8, False), (True, 2, False), (True, 4, True), (False, 2, True)]     # (tape, loader threads, device FITS)
if os.environ.get("MRCNN_PROBE_COMBOS"):
    combos = [tuple(int(v) for v in c.split(",")) for c in os.environ["MRCNN_PROBE_COMBOS"].split(";")]
for tape, nw, devfits in combos:
    cfg = run_py_config(num_classes=4, imgsize=256, backbone="resnet101", images_per_gpu=4, gpu_count=1)
    cfg.TRAIN_LAUNCH_TAPE = bool(tape)
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
    model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
    res = {}
    bench._train_loop_leg(argparse.Namespace(steps=steps), res, model, cfg, 4, 1, 0, loader_threads=nw, device_fits=bool(devfits))
    t = res["train_loop"]
    print("tape=%s threads=%d device_fits=%s: dense %.1f images/s (%.2f ms/step, waiting %.2f), positive quota %.1f images/s (%.2f ms/step, waiting %.2f); queued %.1f / %.1f" % (
        bool(tape), nw, bool(devfits), t["images_per_s"], t["ms_per_step"], t["ms_per_step_waiting_for_loader"], t["exact_zero_skip"]["images_per_s"],
        t["exact_zero_skip"]["ms_per_step"], t["exact_zero_skip"]["ms_per_step_waiting_for_loader"], t["batches_queued_when_asked"],
        t["exact_zero_skip"]["batches_queued_when_asked"]), flush=True)
    del model
    torch.cuda.empty_cache()
