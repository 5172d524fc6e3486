#!/usr/bin/env python3
"""bench.py's configs[4] leg alone, N times in one process (tools only): python tools/config4_leg.py [steps] [repeats]
MRCNN_PRELOAD=1 first builds and steps the headline model (what bench.py has done before it reaches this leg)."""
import os, sys, json, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = types.SimpleNamespace(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 20, warmup=3, no_cpu_baseline=True, detect_iters=3,
                             dense_only=False, backbone="resnet101", nimg=4, imgsize=256)
if os.environ.get("MRCNN_PRELOAD"):
    r = bench.measure(args, "resnet101", 4, 0, 0, 1, full=os.environ["MRCNN_PRELOAD"] == "2")
    print("preload: headline %.2f ms/step" % r["ms_per_step"])
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    out = bench.measure_config4(args, 0, 0, 1)
    print(json.dumps({k: v for k, v in out.items() if k.startswith("ms_")}))
