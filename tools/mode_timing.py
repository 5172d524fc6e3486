#!/usr/bin/env python3
"""Step time of the engine modes: dense / sparse x float32 / float16 mask head (tools only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
backbone = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
imgsize = int(os.environ.get("MRCNN_IMGSIZE", "256"))
dev = torch.device("cuda", 0)
cfg = run_py_config(num_classes=4, imgsize=imgsize, backbone=backbone, images_per_gpu=nimg, gpu_count=1)
model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
inp = model._to_device(bench.synthetic_batch(cfg, nimg, seed=1234))
eng = model.engine
if os.environ.get("MRCNN_LR"):                                 # e.g. 0: time the same weights every step
    cfg.LEARNING_RATE = float(os.environ["MRCNN_LR"])
if os.environ.get("MRCNN_MAIN_PRIO"):                          # run the step from a stream of this priority (-1 = high)
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["MRCNN_MAIN_PRIO"])))
only = sys.argv[3] if len(sys.argv) > 3 else ""            # e.g. "sparse-f32": profile a single mode
for sparse in (False, True):
    for hd in (None, torch.float16, torch.bfloat16):
        tag = ("sparse" if sparse else "dense") + "-" + {None: "f32", torch.float16: "f16", torch.bfloat16: "bf16"}[hd]
        if only and tag != only:
            continue
        eng.sparse_mask_bwd, eng.head_dtype = sparse, hd
        graphed = os.environ.get("MRCNN_TRAIN_GRAPH", "0") != "0"
        taped = os.environ.get("MRCNN_TRAIN_TAPE", "0") != "0"

        def step():
            if taped:
                eng.step_taped(inp, cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
            elif graphed:
                eng.step_graphed(inp, cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
            else:
                eng.forward_backward(*inp); eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
        for _ in range(3):
            step()
        runs = []
        for _ in range(int(os.environ.get("MRCNN_TIMING_RUNS", "1"))):       # several runs of 10: the streams' races make single runs bimodal
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(10):
                step()
            torch.cuda.synchronize(); runs.append((time.time() - t0) / 10)
        dt = sorted(runs)[len(runs) // 2]
        print("sparse=%-5s head=%-14s %s %.2f ms/step  %.1f img/s%s" % (sparse, hd, "tape" if taped else "graph" if graphed else "eager", dt * 1e3, nimg / dt,
              "   (runs: %s)" % " ".join("%.2f" % (r * 1e3) for r in runs) if len(runs) > 1 else ""))
