#!/usr/bin/env python3
"""The HBM-bound kernels of the path at the cfg-2 (ResNet-50 256x256, 2 images, training) and cfg-4 (1024x1024 tile,
inference) shapes, each launched REPS times back to back: run plain for HIP-event durations, or under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / `--kernel-trace --stats` (one pass each; tools/pmc_table.py merges
the CSVs into profiles/*_hbm_kernels.md).  Tools only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from caesar_mrcnn_amd import ops

dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "6"))
rng = np.random.default_rng(0)
rows = []


def rois(B, R, zero_from=None):
    """uniform centres, log-uniform sizes covering all four pyramid levels"""
    c = rng.uniform(0.05, 0.95, (B, R, 2)); s = np.exp(rng.uniform(np.log(0.02), np.log(0.6), (B, R, 2)))
    b = np.concatenate([np.clip(c - s / 2, 0, 1), np.clip(c + s / 2, 0, 1)], -1).astype(np.float32)
    if zero_from is not None:
        b[:, zero_from:] = 0
    return torch.tensor(b, device=dev)


def timed(name, shape, alg_bytes, fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REPS
    rows.append({"kernel": name, "shape": shape, "ms": round(ms, 4), "algorithmic_bytes": int(alg_bytes),
                 "TBps": round(alg_bytes / ms / 1e9, 3)})
    print("%-34s %-44s %8.1f us  %8.1f MB  %6.2f TB/s" % (name, shape, ms * 1e3, alg_bytes / 1e6, alg_bytes / ms / 1e9), flush=True)


def pyramid(B, S):
    return [torch.randn((B, S // s, S // s, 256), device=dev) for s in (4, 8, 16, 32)]


for tag, B, S, cases in (("cfg2", 2, 256, (("cls-train", 512, 7), ("mask-train", 512, 14))),
                         ("cfg4", 1, 1024, (("cls-infer", 1000, 7), ("mask-infer", 100, 14)))):
    fms = pyramid(B, S)
    area = float(S * S)
    for what, R, P in cases:
        bx = rois(B, R)
        alg = B * R * P * P * 256 * 5 * 4                       # 4 corner rows + 1 store per bin, fp32
        timed("roialign_kernel<false>", "%s %s B=%d R=%d P=%d" % (tag, what, B, R, P), alg, lambda: ops.roialign(bx, fms, P, area))
        if tag == "cfg2":
            dout = torch.randn((B, R, P, P, 256), device=dev)
            dfm = [torch.zeros_like(f) for f in fms]
            if P == 7:      # class head: every ROI carries gradient -> gather form
                timed("roialign_bwd_gather_kernel", "%s %s B=%d R=%d P=%d" % (tag, what, B, R, P), alg,
                      lambda: ops.roialign_bwd(bx, dout, dfm, P, area, dense=True))
            else:           # mask head: only the positive quota (168 rows) is non-zero, scatter form skips zero rows
                dout[:, 168:] = 0
                alg_s = B * 168 * P * P * 256 * 5 * 4 + B * (R - 168) * P * P * 256 * 4
                timed("roialign_kernel<true> (scatter)", "%s %s B=%d R=%d (168 non-zero) P=%d" % (tag, what, B, R, P), alg_s,
                      lambda: ops.roialign_bwd(bx, dout, dfm, P, area))
    # ProposalLayer at this tile size: top-k + decode + NMS bit matrix + scan
    A = sum((S // s) ** 2 * 3 for s in (4, 8, 16, 32, 64))
    fg = 1 / (1 + np.exp(-rng.normal(0, 2, (B, A))))
    probs = torch.tensor(np.stack([1 - fg, fg], -1).astype(np.float32), device=dev)
    deltas = torch.tensor((rng.standard_normal((B, A, 4)) * 0.3).astype(np.float32), device=dev)
    ctr = rng.uniform(0, 1, (A, 2)); sz = np.exp(rng.uniform(np.log(0.01), np.log(0.4), (A, 2)))
    anchors = torch.tensor(np.concatenate([ctr - sz / 2, ctr + sz / 2], 1).astype(np.float32), device=dev)
    count = 2000 if tag == "cfg2" else 1000
    K = min(6000, A)
    alg = B * (A * 4 + K * (4 + 16 + 16) + K * 16 + 2 * K * K / 8)      # scores + winners + boxes + bit matrix write/read
    timed("proposal layer (all kernels)", "%s B=%d A=%d -> %d -> %d" % (tag, B, A, K, count), alg,
          lambda: ops.proposals(probs, deltas, anchors, 6000, count, 0.7, np.array([0.1, 0.1, 0.2, 0.2], np.float32)))

# optimiser on the ResNet-101 flat buffer (63.6 M parameters): grad_prepare (r2 w1 + norm) and sgd (r3 w2)
n = 63_600_000 // 64 * 64
params, grads, mom = (torch.randn(n, device=dev) * 0.01 for _ in range(3))
coef = torch.full((n // 64,), 1e-6, device=dev)
sumsq = torch.zeros(1, device=dev)
timed("grad_prepare (L2 + sum of squares)", "R101 n=%d" % n, n * 12, lambda: ops.grad_prepare(grads, params, 1.0, coef, sumsq))
timed("sgd_kernel", "R101 n=%d" % n, n * 20, lambda: ops.sgd_momentum(params, mom, grads, sumsq, 5.0, 1e-4, 0.9, coef))
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
