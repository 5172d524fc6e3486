#!/usr/bin/env python3
"""Where does the feed-inclusive loop wait?  Times Prefetcher.__next__ (queue.get), _to_device and the step issue separately,
with the queue depth at each request (tools only)."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
from caesar_mrcnn_amd.datagen import Prefetcher, data_generator
dev = torch.device("cuda", 0)
nw = int(os.environ.get("NW", "2"))
devfits = os.environ.get("DEVFITS", "1") != "0"
cfg = run_py_config(num_classes=4, imgsize=256, backbone="resnet101", images_per_gpu=4, gpu_count=1)
model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
ds = bench.synthetic_fits_dataset(cfg, 32, "/tmp/mrcnn_bench_data_probe", seed=1234)
cfg.DEVICE_RPN_TARGETS = True
ds.device = dev if devfits else None
gen = Prefetcher([data_generator(ds, cfg, shuffle=True, batch_size=4, seed=99 + 1000 * k, device_targets=True) for k in range(nw)], depth=2 * nw + 2)
eng = model.engine
eng.sparse_mask_bwd = os.environ.get("DENSE", "0") == "0"
for _ in range(4):
    inputs, _ = next(gen); model.train_on_batch(inputs)
torch.cuda.synchronize()
T = {"get": [], "to_device": [], "issue": [], "depth": []}
t0 = time.perf_counter()
for _ in range(30):
    T["depth"].append(gen._q.qsize())
    a = time.perf_counter()
    inputs, _ = next(gen)
    b = time.perf_counter()
    di = model._to_device(inputs)
    c = time.perf_counter()
    eng.grad_ready = None
    eng.step_taped(di, model._lr, model._momentum)
    d = time.perf_counter()
    T["get"].append(b - a); T["to_device"].append(c - b); T["issue"].append(d - c)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
gen.close()
print("threads %d device_fits %s: %.2f ms/step; get %.2f (max %.2f) to_device %.2f issue %.2f ms; depth %.1f" % (
    nw, devfits, tot / 30 * 1e3, np.mean(T["get"]) * 1e3, np.max(T["get"]) * 1e3, np.mean(T["to_device"]) * 1e3, np.mean(T["issue"]) * 1e3,
    np.mean(T["depth"])))
print("get ms:", " ".join("%.1f" % (v * 1e3) for v in T["get"]))
