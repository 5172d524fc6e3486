#!/usr/bin/env python3
"""Headline benchmark: Mask R-CNN training throughput (images/s, whole job) and detect latency on
synthetic 256x256 3-class radio tiles, float32, on N MI355X of one node (one process per GPU).
Default workload = the one BASELINE.json's metric is quoted on (ResNet-101 256x256, nimg_per_gpu=4, i.e.
configs[2] per GPU); at N=1 the line also carries configs[1] (ResNet-50, nimg_per_gpu=2) as a second leg.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = forward + backward + gradient all-reduce (N > 1) + SGD update on IMAGES_PER_GPU images per
rank, inputs resident in HBM.  Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synthetic_batch(cfg, B, seed):
    """Noise tiles + elliptical Gaussian sources with masks/boxes/classes and RPN targets
    (SURVEY section 8d synthetic inputs; already 'molded': float32, MEAN_PIXEL = 0)."""
    from caesar_mrcnn_amd import utils
    from caesar_mrcnn_amd.datagen import build_rpn_targets
    rng = np.random.RandomState(seed)
    S, G = int(cfg.IMAGE_SHAPE[0]), cfg.MAX_GT_INSTANCES
    anchors = utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS,
                                             utils.compute_backbone_shapes(cfg, cfg.IMAGE_SHAPE),
                                             cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
    images = np.zeros((B, S, S, 3), np.float32)
    gt_masks = np.zeros((B, S, S, G), bool)
    gt_boxes = np.zeros((B, G, 4), np.int32)
    gt_cls = np.zeros((B, G), np.int32)
    rpn_match = np.zeros((B, anchors.shape[0], 1), np.int32)
    rpn_bbox = np.zeros((B, cfg.RPN_TRAIN_ANCHORS_PER_IMAGE, 4), np.float32)
    yy, xx = np.mgrid[0:S, 0:S]
    for b in range(B):
        img = rng.normal(0, 1, (S, S))
        n = rng.randint(1, 7)
        for g in range(n):
            cy, cx = rng.uniform(12, S - 12, 2)
            sy, sx = rng.uniform(1.5, 12, 2)
            blob = np.exp(-0.5 * (((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2))
            img += blob * rng.uniform(5, 200)
            m = blob > 0.2
            gt_masks[b, :, :, g] = m
            gt_boxes[b, g] = utils.extract_bboxes(m[:, :, None])[0]
            gt_cls[b, g] = rng.randint(1, cfg.NUM_CLASSES)
        lo, hi = np.percentile(img, [1, 99.5])
        img8 = np.clip((img - lo) / max(hi - lo, 1e-6), 0, 1) * 255.0      # stretch -> uint8-range RGB tile
        images[b] = np.round(img8)[..., None]
        m_, bb = build_rpn_targets(images[b].shape, anchors, gt_cls[b, :n], gt_boxes[b, :n], cfg, rng)
        rpn_match[b, :, 0], rpn_bbox[b] = m_, bb
    meta = np.stack([utils.compose_image_meta(b, (S, S, 3), (S, S, 3), (0, 0, S, S), 1.0,
                                              np.ones(cfg.NUM_CLASSES, np.int32)) for b in range(B)])
    return [images, meta, rpn_match, rpn_bbox, gt_cls, gt_boxes, gt_masks]


def cpu_baseline(cfg, weights, batch, cores):
    """The CPU oracle (restatement of the reference graph; NOT TensorFlow) on one image: forward +
    backward + total-loss gradient, timed on the host cores.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import mrcnn_oracle as orc
    torch.set_num_threads(cores)
    one = [np.ascontiguousarray(a[:1]) for a in batch]
    images, meta, rpn_match, rpn_bbox, gt_cls, gt_boxes, gt_masks = one
    o = orc.OracleMaskRCNN(cfg, weights, requires_grad=True)
    anchors = orc.get_anchors(cfg, images.shape[1:])
    keys = np.random.RandomState(0).uniform(0, 1, (1, cfg.POST_NMS_ROIS_TRAINING)).astype(np.float32)
    # threads pinned to `cores` host cores, 3 warm-up + 10 timed iterations, median and p10 / p90
    allowed = sorted(os.sched_getaffinity(0))
    try:
        os.sched_setaffinity(0, set(allowed[:cores]))
    except OSError:
        pass

    def one():
        for w_ in o.w.values():
            w_.grad = None
        t = time.time()
        ref = o.forward_training(images, rpn_match, rpn_bbox.astype(np.float32), gt_cls, gt_boxes, gt_masks,
                                 meta[:, 12:].astype(np.int32), anchors, keys)
        o.total_loss(ref["losses"]).backward()
        return time.time() - t

    t0 = time.time()
    for _ in range(3):
        one()
    times = []
    while len(times) < 10 and (len(times) < 3 or time.time() - t0 < 75.0):      # bounded: a slow host stops early, says so
        times.append(one())
    try:
        os.sched_setaffinity(0, set(allowed))
    except OSError:
        pass
    med, p10, p90 = (float(np.percentile(times, q)) for q in (50, 10, 90))
    return {"value": round(1.0 / med, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "ms_per_image_median": round(med * 1e3, 1), "ms_per_image_p10": round(p10 * 1e3, 1),
            "ms_per_image_p90": round(p90 * 1e3, 1), "iterations": len(times), "warmup": 3,
            "sample": "%d timed training steps (fwd+bwd, after 3 warm-up) of 1 image each of the same workload, CPU oracle "
                      "(torch-CPU fp32 restatement of the reference graph, not TF1), threads pinned to %d cores, %.1f s in all"
                      % (len(times), cores, time.time() - t0)}


def synthetic_fits_dataset(cfg, n_images, root, seed=1234):
    """SURVEY 8(d) synthetic inputs as FILES: noise tiles with 1-6 elliptical Gaussian sources written as BITPIX=-32 FITS
    (+ a NaN border strip on some), one mask FITS per object, one caesar JSON per tile; loaded through the product's
    SourceDataset exactly as `run.py train --datalist_json` would (FITS reader, NaN fill, zscale, uint8 RGB)."""
    from caesar_mrcnn_amd import fits
    from caesar_mrcnn_amd.dataset import SourceDataset
    os.makedirs(root, exist_ok=True)
    rng = np.random.RandomState(seed)
    S = int(cfg.IMAGE_SHAPE[0])
    yy, xx = np.mgrid[0:S, 0:S]
    names = ["sidelobe", "source", "galaxy"]
    jsons = []
    for i in range(n_images):
        img = rng.normal(0, 1, (S, S)).astype(np.float32)
        objs = []
        for g in range(rng.randint(1, 7)):
            cy, cx = rng.uniform(12, S - 12, 2)
            sy, sx = rng.uniform(1.5, 12, 2)
            blob = np.exp(-0.5 * (((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2))
            img += (blob * rng.uniform(5, 200)).astype(np.float32)
            mfile = "mask_%d_%d.fits" % (i, g)
            fits.write_fits(os.path.join(root, mfile), (blob > 0.2).astype(np.float32))
            objs.append({"mask": mfile, "class": names[rng.randint(0, 3)], "nislands": 1, "sidelobe-mixed": 0, "sidelobe-near": 0})
        if i % 4 == 0:
            img[:2, :] = np.nan                          # NaN strip: exercises the NaN -> min fill of read_fits
        fits.write_fits(os.path.join(root, "img_%d.fits" % i), img, {"BUNIT": "JY/BEAM"})
        jf = os.path.join(root, "tile_%d.json" % i)
        with open(jf, "w") as fh:
            json.dump({"img": "img_%d.fits" % i, "objs": objs}, fh)
        jsons.append(jf)
    ds = SourceDataset()
    ds.set_class_dict({"sidelobe": 1, "source": 2, "galaxy": 3})
    for jf in jsons:
        assert ds.load_data_from_json_file(jf, root) == 0
    ds.prepare()
    return ds


def _train_loop_leg(args, res, model, cfg, nimg, world, rank, loader_threads=None, device_fits=None, key="train_loop"):
    """The loop MaskRCNN.train() runs, timed end to end: loader threads (FITS -> zscale -> resize -> GT boxes / masks) ->
    H2D of images and of the USED GT-mask planes -> RPN targets on the device -> step.  Unlike `value`, inputs are NOT
    resident in HBM: this is the feed-inclusive rate (reference: model.py:2487-2499, fit_generator + workers)."""
    import torch
    from caesar_mrcnn_amd.datagen import Prefetcher, data_generator
    ds = synthetic_fits_dataset(cfg, 32, "/tmp/mrcnn_bench_data_r%d" % rank, seed=1234 + rank)
    cfg.DEVICE_RPN_TARGETS = True
    if device_fits is None:
        device_fits = bool(getattr(cfg, "DEVICE_FITS", False))
    ds.device = model.engine.dev if device_fits else None      # FITS -> zscale -> uint8 RGB on the GPU (what MaskRCNN.train sets up)
    # with the tile pre-processing on the device two loader threads keep up (they mostly wait on the GPU, interpreter lock
    # released); the host path needs a thread per image of the batch and more
    nw = loader_threads if loader_threads else (2 if device_fits else min(8, len(os.sched_getaffinity(0))))
    gen = Prefetcher([data_generator(ds, cfg, shuffle=True, batch_size=nimg, seed=99 + 1000 * k, device_targets=True)
                      for k in range(nw)], depth=2 * nw + 2)
    eng = model.engine
    steps = max(args.steps, 10)
    t_fill = time.time()                              # steady state: let the loaders fill their queue once (cold start: first
    while gen._q.qsize() < 2 * nw + 2 and time.time() - t_fill < 10.0:      # file reads, stream / pinned-buffer set-up)
        time.sleep(0.01)

    def run(sparse):
        eng.sparse_mask_bwd = sparse
        for _ in range(3):
            inputs, _ = next(gen)
            model.train_on_batch(inputs)
        torch.cuda.synchronize()
        t0 = time.time()
        wait, h2d, depth = 0.0, 0, 0
        for _ in range(steps):
            depth += gen._q.qsize()
            t1 = time.time()
            inputs, _ = next(gen)
            wait += time.time() - t1
            used = np.flatnonzero(np.any(inputs[4] != 0, axis=0))
            n_used = int(used[-1]) + 1 if used.size else 0
            h2d += inputs[0].nbytes + inputs[6][..., :(n_used + 7) // 8].size + inputs[4].nbytes + inputs[5].nbytes + 4 * inputs[5].size
            model.train_on_batch(inputs)
        torch.cuda.synchronize()
        dt = time.time() - t0
        return {"images_per_s": round(nimg * steps / dt, 3), "ms_per_step": round(dt / steps * 1e3, 3),
                "ms_per_step_waiting_for_loader": round(wait / steps * 1e3, 3), "h2d_bytes_per_step": int(h2d / steps),
                "batches_queued_when_asked": round(depth / steps, 2)}

    try:
        dense = run(False)                            # the headline's work per step (every ROI row through the mask head)
        sparse = run(True)                            # product default (positive-quota rows only): the feed shows here first
    finally:
        gen.close()
        eng.sparse_mask_bwd = True
    res[key] = dict(dense, steps=steps, loader_threads=nw, exact_zero_skip=sparse,
        launch_tape=bool(getattr(cfg, "TRAIN_LAUNCH_TAPE", False)), fits_preprocessing="device (mrcnn_fits_to_rgb)" if device_fits else "host (NumPy)",
        what="MaskRCNN.train()'s own iteration (steps re-issued from the launch recording when launch_tape): Prefetcher threads over data_generator on 32 synthetic FITS tiles "
             "(read_fits + zscale + uint8 RGB + resize + extract_bboxes), per-step H2D of images and the used GT-mask planes, "
             "bit-packed (8 instances per byte), RPN targets built on the device; dense mask head like `value`, and the product-default step under "
             "exact_zero_skip; feed-inclusive, never `value`")


TRAFFIC_PER_LAUNCH = {1024: None, 2048: 1.810e9}   # mask-head conv, PMC passes: profiles/r01_pmc_conv_traffic.md
TRAFFIC_WGRAD_PER_LAUNCH = {1024: None, 2048: 4.55e9}   # its weight gradient (same file)
# (ROIs, tiling) -> HBM-side bytes per launch of the roofline leg's GEMM: 2 x FETCH_SIZE (gfx950 reports half of wide reads, guide) +
# WRITE_SIZE, separate --pmc passes (profiles/r03_pmc_winograd_gemm.txt; round 2: 1.597e9); measured for the default tiling only
TRAFFIC_WINOGRAD_GEMM = {(2048, 6): 1.594e9}


def measure(args, backbone, nimg, rank, local_rank, world, full):
    """Train-step throughput (+ detect latency, roofline, cpu_baseline when ``full``) of one workload."""
    import torch
    import torch.distributed as dist
    from caesar_mrcnn_amd import ops
    from caesar_mrcnn_amd.config import run_py_config
    from caesar_mrcnn_amd.model import MaskRCNN
    from caesar_mrcnn_amd.parallel import GradReducer

    dev = torch.device("cuda", int(os.environ.get("MRCNN_FORCE_DEVICE", local_rank)))   # override: rehearsals on one GPU
    cfg = run_py_config(num_classes=4, imgsize=args.imgsize, backbone=backbone, images_per_gpu=nimg,
                        gpu_count=world)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # stdout carries the ONE JSON line only
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
        model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
    batch = synthetic_batch(cfg, nimg, seed=1234 + rank)
    dev_inputs = model._to_device(batch)
    eng = model.engine
    reducer = GradReducer(eng.grads, world, rank=rank, timing=True) if world > 1 else None
    eng.grad_ready = reducer.ready if reducer else None

    def step():
        losses = eng.forward_backward(*dev_inputs)
        if reducer:
            reducer.finish()
        eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, world)
        return losses

    def timed(nsteps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(nsteps):
            ls = step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.time() - t0
        if world > 1:
            t = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, ls

    # headline: every multiplication of the reference graph is executed (dense mask-head backward)
    eng.sparse_mask_bwd = False
    for _ in range(args.warmup):
        step()
    if reducer:
        reducer.pop_timing()
    dt, losses = timed(args.steps)
    comm = None
    if reducer:
        # what the gradient exchange costs: time on the exchange stream (HIP events around every range), bytes per range,
        # and the EXPOSED part = step time with the exchange - step time of the same step without it (all ranks run both)
        ar_ms, ar_bytes, ranges = reducer.pop_timing()
        eng.grad_ready = None
        reducer_saved, reducer = reducer, None
        step()
        dt_nocomm, _ = timed(args.steps)
        reducer = reducer_saved
        eng.grad_ready = reducer.ready
        comm = {"transport": {"rccl": "RCCL ncclAllReduce through the C-ABI (mrcnn_allreduce_grad)",
                              "direct": "grouped point-to-point reduce-scatter + all-gather through the C-ABI",
                              "torch": "torch.distributed all_reduce"}.get(reducer.mode, reducer.mode),
                "ms_per_step_on_exchange_stream": round(ar_ms / args.steps, 3),
                "bytes_per_step": int(ar_bytes / args.steps), "range_bytes": ranges[:len(ranges) // max(args.steps, 1)],
                "ms_per_step_without_exchange": round(dt_nocomm / args.steps * 1e3, 3),
                "exposed_ms_per_step": round((dt - dt_nocomm) / args.steps * 1e3, 3),
                "algorithmic_bus_GBps": round(2.0 * (world - 1) / world * ar_bytes / max(ar_ms, 1e-9) / 1e6, 2)}
    # product default: the mask head runs on the positive quota of each image only (rows the loss can read)
    eng.sparse_mask_bwd = True
    if args.dense_only:
        dt_sparse = float("nan")
    else:
        step()
        dt_sparse, _ = timed(args.steps)
    # configs[4], stages 1-2 (NOT the headline: reduced precision): the mask head on the fp16 MFMA,
    # every row computed as in the headline leg
    dt_h16 = float("nan")
    if not args.dense_only:
        import torch as _t
        eng.sparse_mask_bwd, eng.head_dtype = False, _t.float16
        step()
        step()
        dt_h16, _ = timed(args.steps)
        eng.sparse_mask_bwd, eng.head_dtype = True, None
    # for transparency: the headline step with the DIRECT 3x3 kernels in the mask head (the Winograd path switched off)
    dt_direct = float("nan")
    if not args.dense_only and getattr(eng, "winograd", False):
        eng.sparse_mask_bwd, eng.winograd = False, False
        step()
        step()
        nd = max(4, args.steps // 2)
        dt_direct, _ = timed(nd)
        dt_direct = dt_direct / nd * args.steps
        eng.sparse_mask_bwd, eng.winograd = True, True
    res = {"backbone": backbone, "nimg": nimg, "ms_per_step": dt / args.steps * 1e3,
           "images_per_s_direct": None if dt_direct != dt_direct else nimg * world * args.steps / dt_direct,
           "images_per_s_f16_mask_head": None if args.dense_only else nimg * world * args.steps / dt_h16,
           "images_per_s": nimg * world * args.steps / dt,
           "images_per_s_sparse": None if args.dense_only else nimg * world * args.steps / dt_sparse,
           "losses": [float(v) for v in losses.cpu().numpy()]}
    if comm is not None:
        res["allreduce"] = comm
    if rank != 0:
        return res

    # ---- detect latency (inference graph, batch 1), same weights -----------------------------------
    # (secondary legs never cost the headline number: a failure is reported in the line instead)
    res.update({"detect_eager_ms": None, "detect_ms": None, "detect_ms_b8": None})
    try:
        _detect_leg(args, res, eng, dev_inputs, dev, backbone, run_py_config, torch)
    except Exception as e:
        res["detect_error"] = repr(e)
    eng.cfg = cfg
    if full:
        try:
            _detect_e2e_leg(args, res, eng, dev, backbone, run_py_config, torch)
        except Exception as e:
            res["detect_e2e"] = {"error": repr(e)}
    if not full:
        return res
    if not args.dense_only:
        try:
            _train_loop_leg(args, res, model, cfg, nimg, world, rank)
        except Exception as e:
            res["train_loop"] = {"error": repr(e)}
    try:
        _roofline_leg(res, ops, torch, dev, nimg, cfg)
    except Exception as e:
        res["roofline"] = {"bound": "mfma", "achieved": None, "peak": 157.3, "unit": "TFLOP/s", "frac": None, "traffic": None,
                           "error": repr(e)}
    if not args.no_cpu_baseline and world == 1:
        try:
            cores = min(16, len(os.sched_getaffinity(0)))   # the GPU box grants 16 host cores per GPU
            res["cpu_baseline"] = cpu_baseline(cfg, eng.get_weights(), batch, cores)
        except Exception as e:          # the baseline is a report, never a reason to lose the GPU number
            res["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": 0, "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    return res


def _detect_leg(args, res, eng, dev_inputs, dev, backbone, run_py_config, torch):
    x1 = dev_inputs[0][:1].contiguous()
    win = torch.tensor([[0.0, 0.0, 1.0, 1.0]], device=dev)
    icfg = run_py_config(num_classes=4, imgsize=args.imgsize, backbone=backbone, mode="inference")
    eng.cfg = icfg
    for _ in range(2):
        eng.infer(x1, win)
    torch.cuda.synchronize()
    t1 = time.time()
    for _ in range(args.detect_iters):
        eng.infer(x1, win)
    torch.cuda.synchronize()
    res["detect_eager_ms"] = (time.time() - t1) / args.detect_iters * 1e3
    for _ in range(2):
        eng.infer_graphed(x1, win)                      # captures on first use
    torch.cuda.synchronize()
    t1 = time.time()
    for _ in range(args.detect_iters):
        eng.infer_graphed(x1, win)
    torch.cuda.synchronize()
    res["detect_ms"] = (time.time() - t1) / args.detect_iters * 1e3
    # reduced precision, reported separately: the same detect with the 16-bit stages (mask head, FPN smoothing, RPN, class FCs,
    # bottleneck blocks) -- never `detect_ms_per_image`
    try:
        eng.head_dtype = torch.float16
        for _ in range(2):
            eng.infer_graphed(x1, win)
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(args.detect_iters):
            eng.infer_graphed(x1, win)
        torch.cuda.synchronize()
        res["detect_ms_f16"] = (time.time() - t1) / args.detect_iters * 1e3
    finally:
        eng.head_dtype = None
    # the same graph on 8 tiles at once (detect() takes a list of images): what batching buys on the latency-bound backbone
    x8 = x1.expand(8, -1, -1, -1).contiguous()
    win8 = win.expand(8, -1).contiguous()
    for _ in range(2):
        eng.infer(x8, win8)
    torch.cuda.synchronize()
    t1 = time.time()
    for _ in range(args.detect_iters):
        eng.infer(x8, win8)
    torch.cuda.synchronize()
    res["detect_ms_b8"] = (time.time() - t1) / args.detect_iters / 8 * 1e3


def _detect_e2e_leg(args, res, eng, dev, backbone, run_py_config, torch):
    """BASELINE's second metric as the CALL a user makes: MaskRCNN.detect([uint8 image]) = mold_inputs (host) -> H2D ->
    inference graph (HIP-graph replay) -> detections D2H -> box arithmetic (host) -> masks resized / pasted on the device
    (mrcnn_unmold_masks) -> D2H of the [H, W, n] planes (mrcnn/model.py:2623-2704).  Median of >= 20 calls; the class head's
    background bias is lowered so that the random-init network returns DETECTION_MAX_INSTANCES detections (the worst case
    for the post-processing).  The split is measured in a second pass with marks inside detect()."""
    from caesar_mrcnn_amd.model import MaskRCNN
    icfg = run_py_config(num_classes=4, imgsize=args.imgsize, backbone=backbone, mode="inference")
    w = eng.get_weights()
    b = np.array(w["mrcnn_class_logits/bias"], dtype=np.float32, copy=True)
    b[0] = -30.0
    w["mrcnn_class_logits/bias"] = b
    model = MaskRCNN("inference", icfg, "/tmp/mrcnn_bench_logs", device=dev, weights=w)
    rng = np.random.RandomState(7)
    S = args.imgsize
    yy, xx = np.mgrid[0:S, 0:S]
    img = rng.normal(0, 1, (S, S))
    for _ in range(6):
        cy, cx = rng.uniform(12, S - 12, 2)
        sy, sx = rng.uniform(1.5, 12, 2)
        img += np.exp(-0.5 * (((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2)) * rng.uniform(5, 200)
    img = np.clip((img - img.min()) / (img.max() - img.min()) * 255.0, 0, 255).astype(np.uint8)
    image = np.stack([img, img, img], axis=-1)
    for _ in range(3):
        r = model.detect([image])[0]
    n_iter = max(20, args.detect_iters)
    times, marks = [], []
    for _ in range(n_iter):
        t0 = time.perf_counter()
        r = model.detect([image])[0]
        times.append(time.perf_counter() - t0)
    for _ in range(n_iter):
        tm = {}
        model.detect([image], timing=tm)
        marks.append(tm)
    model.detect_zero_copy = True                      # opt-in: `masks` as views of pinned memory (no host copy of the planes)
    zc = []
    for _ in range(3 + n_iter):
        t0 = time.perf_counter()
        model.detect([image])
        zc.append(time.perf_counter() - t0)
    model.detect_zero_copy = False
    med = lambda v: float(np.median(v)) * 1e3
    res["detect_e2e"] = {
        "ms_per_image": round(med(times), 3), "p10": round(float(np.percentile(times, 10)) * 1e3, 3),
        "p90": round(float(np.percentile(times, 90)) * 1e3, 3), "calls": n_iter, "n_detections": int(r["rois"].shape[0]),
        "mask_pixels_set": int(r["masks"].sum()),
        "ms_per_image_zero_copy_masks": round(float(np.median(zc[3:])) * 1e3, 3),
        "split_ms": {"mold_host": round(med([m["molded"] - m["start"] for m in marks]), 3),
                     "h2d_graph_d2h_detections": round(med([m["graph_done"] - m["molded"] for m in marks]), 3),
                     "boxes_host_unmold_device_d2h_masks": round(med([m["end"] - m["graph_done"] for m in marks]), 3)},
        "what": "MaskRCNN.detect([uint8 %dx%dx3]) wall clock per call, batch 1, host pre / post-processing and all copies included" % (S, S)}
    del model


def _roofline_leg(res, ops, torch, dev, nimg, cfg):
    """Rooflines of the kernels that dominate the headline step, timed live with HIP events on the launch stream.
    Since the Winograd F(2x2,3x3) path the mask head's eight 3x3 forward / data-gradient convolutions and its
    four 3x3 weight gradients run as transform-domain GEMMs: `roofline` = the persistent batched GEMM kernel (8 launches per
    step, ~14 of the ~45 ms: the largest single item), `roofline_wgrad` = the 16 weight-gradient GEMMs
    of a layer in one multi-problem launch.  The direct 3x3 kernels (detect, small ROI counts, MRCNN_WINOGRAD=0) keep their
    objects as `roofline_direct_conv` / `roofline_direct_wgrad`."""
    M_rois = nimg * cfg.TRAIN_ROIS_PER_IMAGE
    C_ = 256
    xm = torch.randn((M_rois, 14, 14, C_), device=dev)
    wm = torch.randn((3, 3, C_, C_), device=dev) * 0.02
    bm = torch.zeros(C_, device=dev)
    sc = torch.ones(C_, device=dev)
    om = torch.empty((M_rois, 14, 14, C_), device=dev)
    zm = torch.empty((M_rois, 14, 14, C_), device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20

    def timed_ms(fn):
        for _ in range(3):
            fn()
        e0.record()                                     # same (current) stream the launches go to
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    peak = 157.3
    flops_direct = 2.0 * (M_rois * 196) * C_ * 9 * C_          # the layer as a direct convolution
    tile = ops.winograd_tile((M_rois, 14, 14, C_))             # what the step uses for this layer: TILE_MIXED (F(4x4) tiles + a last row / column of 4x2, 2x4, 2x2) unless MRCNN_WINOGRAD_MIXED=0 / MRCNN_WINOGRAD_TILE=2
    import ctypes as _C
    groups = ops.winograd_groups(14, 14, tile)
    lib = ops._hip.lib()
    g0 = groups[0]                                             # the group that carries the layer: 3 x 3 tiles of 4 x 4 (or the uniform tiling)
    nb = (g0.oth + 2) * (g0.otw + 2)
    T = M_rois * g0.th_n * g0.tw_n
    flops_gemm = 2.0 * nb * T * C_ * C_                        # what this launch multiplies
    flops_layer_gemm = sum(2.0 * (g.oth + 2) * (g.otw + 2) * M_rois * g.th_n * g.tw_n * C_ * C_ for g in groups)
    nv = lib.mrcnn_winograd_group_floats(_C.byref(g0), M_rois, C_)
    rows = nv // (nb * C_)
    V = torch.empty(ops.winograd_v_floats((M_rois, 14, 14, C_), tile), device=dev); Mt = torch.empty(nv, device=dev)
    U = ops.winograd_weights(wm, tile=tile)
    U0 = U[0] if isinstance(U, (list, tuple)) else U
    st = ops.current_stream
    P = ops.ptr
    t_in = timed_ms(lambda: lib.mrcnn_winograd_input_g(P(xm), P(V), M_rois, 14, 14, C_, _C.byref(g0), st()))
    t_gemm = timed_ms(lambda: lib.mrcnn_winograd_gemm(P(V), P(U0), P(Mt), nb, rows, C_, C_, st()))
    t_gemm_blds = timed_ms(lambda: lib.mrcnn_gemm_batched_f32(P(V), P(U0), P(Mt), nb, rows, C_, C_, st()))
    t_out = timed_ms(lambda: lib.mrcnn_winograd_output_g(P(Mt), P(om), P(zm), P(bm), P(sc), P(bm), M_rois, 14, 14, C_, 1, _C.byref(g0), st()))
    t_layer = timed_ms(lambda: ops.conv2d_winograd(xm, U, bm, sc, bm, 1, out=om, z_out=zm))
    ach = flops_gemm / (t_gemm * 1e-3) / 1e12
    res["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                       "traffic": TRAFFIC_WINOGRAD_GEMM.get((M_rois, tile)),
                       "kernel": "winograd_gemm_kernel (mrcnn_winograd_gemm): the %d transform-domain GEMMs [%d x 256] . [256 x 256] of the "
                                 "F(%dx%d,3x3) tiles of a Winograd mask-head layer (%.0f %% of the layer's multiplications; tiling %s) in "
                                 "one launch of persistent workgroups (128x256 tiles, K = 256, LDS-DMA operands, next tile's first stage "
                                 "in flight under the current tile's stores: %.1f GFLOP/launch, %.3f ms/launch; the one-tile-per-workgroup "
                                 "LDS-DMA kernel on the same product: %.3f ms)"
                                 % (nb, T, g0.oth, g0.otw, 100.0 * flops_gemm / flops_layer_gemm,
                                    {2: "2x2", 4: "4x4 with overhang", 6: "4x4 + last row / column of 4x2, 2x4, 2x2"}[tile],
                                    flops_gemm / 1e9, t_gemm, t_gemm_blds),
                       "layer_ms": {"input_transform_main_group": round(t_in, 3), "gemm_main_group": round(t_gemm, 3),
                                    "output_transform_with_epilogue_main_group": round(t_out, 3), "whole_layer": round(t_layer, 3)},
                       "layer_gemm_gflop": round(flops_layer_gemm / 1e9, 1),
                       "layer_equivalent_direct_tflops": round(flops_direct / (t_layer * 1e-3) / 1e12, 2),
                       "winograd_tile": tile,
                       "note": "the layer computes what a direct 3x3 convolution of %.1f GFLOP computes (tile 4: 1.3e-5 of its result's "
                               "range, tile 2: 1.6e-6) with %.1f GFLOP of MFMA work plus two memory-bound transform passes; layer_equivalent_direct_tflops = the "
                               "direct convolution's flops over the whole layer's time (a rate the direct kernel would need, not one "
                               "the matrix cores run at)" %
                               (flops_direct / 1e9, flops_layer_gemm / 1e9)}
    # the direct kernel (detect, small ROI counts): the variant with the pre-BN z store and the bare one
    k_ms = timed_ms(lambda: ops.conv2d(xm, wm, bm, sc, bm, act=1, out=om, z_out=zm))
    k_ms_bare = timed_ms(lambda: ops.conv2d(xm, wm, bm, sc, bm, act=1, out=om))
    achieved = flops_direct / (k_ms * 1e-3) / 1e12
    res["roofline_direct_conv"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(achieved / peak, 4), "traffic": TRAFFIC_PER_LAUNCH.get(M_rois),
                                   "achieved_without_z_store": round(flops_direct / (k_ms_bare * 1e-3) / 1e12, 2),
                                   "kernel": "conv_fwd_blds_kernel, 128x128 tile, direct 3x3 (M=%d N=256 K=2304, %.1f GFLOP/launch, %.3f "
                                             "ms/launch with the z store, %.3f without)" % (M_rois * 196, flops_direct / 1e9, k_ms, k_ms_bare)}
    # weight gradient of the same layer: Winograd form (dy transform + 16 GEMMs in one multi-problem launch + slab reduction +
    # dW transform), and the direct kernel
    dym = torch.randn((M_rois, 14, 14, C_), device=dev)
    dwm = torch.empty((3, 3, C_, C_), device=dev)
    ops.conv2d_winograd(xm, U, keep_v=V)
    wl_ms = timed_ms(lambda: ops.conv2d_wgrad_winograd(V, (M_rois, 14, 14, C_), dym, dwm, tile=tile))
    dM = torch.randn((nb, rows, C_), device=dev); dU = torch.empty((nb, C_, C_), device=dev)
    Vv = V[:nv].view(nb, rows, C_)
    items = [(Vv[k, :T].view(T, 1, 1, C_), dM[k, :T].view(T, 1, 1, C_), (1, 1, C_, C_), 1, "valid", dU[k].view(1, 1, C_, C_), False)
             for k in range(nb)]
    per = 16 if nb == 16 else 12                               # GEMMs per launch (what conv2d_wgrad_winograd does)

    def wgrad_gemms():
        for i in range(0, nb, per):
            ops.conv2d_wgrad_multi(items[i:i + per])
    wg_ms = timed_ms(wgrad_gemms)
    w_ach = flops_gemm / (wg_ms * 1e-3) / 1e12
    res["roofline_wgrad"] = {"bound": "mfma", "achieved": round(w_ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(w_ach / peak, 4),
                             "traffic": None,
                             "kernel": "conv_wgrad_blds_multi_kernel<16>: the %d transform-domain weight-gradient GEMMs [256 x %d] . [%d x 256] "
                                       "of a Winograd layer's main tile group in %d launch(es) + their slab reductions (%.1f GFLOP, %.3f ms)" %
                                       (nb, T, T, nb // per, flops_gemm / 1e9, wg_ms),
                             "layer_ms": round(wl_ms, 3),
                             "layer_equivalent_direct_tflops": round(flops_direct / (wl_ms * 1e-3) / 1e12, 2)}
    w_ms = timed_ms(lambda: ops.conv2d_wgrad(xm, dym, (3, 3, C_, C_), 1, "same", dw=dwm))
    res["roofline_direct_wgrad"] = {"bound": "mfma", "achieved": round(flops_direct / (w_ms * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                                    "frac": round(flops_direct / (w_ms * 1e-3) / 1e12 / peak, 4),
                                    "traffic": TRAFFIC_WGRAD_PER_LAUNCH.get(M_rois),
                                    "kernel": "conv_wgrad_blds_kernel<16,true> + pixel table + slab reduction, direct 3x3 (%.1f "
                                              "GFLOP/launch, %.3f ms/launch)" % (flops_direct / 1e9, w_ms)}
    # what the step gets: a layer's data gradient (main stream) and weight gradient (side stream) run side by side
    from caesar_mrcnn_amd.engine import _side_streams
    side = _side_streams(dev)[0]                        # the engines' weight-gradient stream (one per process)
    main = torch.cuda.current_stream(dev)

    def pair():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.conv2d_wgrad_winograd(V, (M_rois, 14, 14, C_), dym, dwm, tile=tile)
        ops.conv2d_winograd(xm, U, bm, sc, bm, 1, out=om)
        main.wait_stream(side)
    p_ms = timed_ms(pair)
    res["roofline"]["pair_beside_wgrad_stream_ms"] = round(p_ms, 3)
    res["roofline"]["pair_equivalent_direct_tflops"] = round(2 * flops_direct / (p_ms * 1e-3) / 1e12, 2)
    del xm, om, dym, V, Mt, dM


def measure_config4(args, rank, local_rank, world):
    """BASELINE.json configs[4]: ResNet-101 with 16-bit weight images / activations on the 16-bit matrix cores, 512 x 512
    tiles, 4 images per GPU.  Mixed precision as built so far: mask head, FPN smoothing, shared RPN
    convolution, class-head FC layers and the identity bottleneck blocks of res4 / res5 in 16 bits (forward, data and
    weight gradient); every bottleneck block's convolutions in 16 bits (res2 / res3 weight gradients widened to float32 on the
    side stream); stem, laterals and the small output layers float32; float32 master weights, accumulation, gradients; loss
    scale 4096 for float16 with the guarded optimiser step (non-finite gradients skip the update).  Own object, never the
    headline."""
    import contextlib
    import torch
    import torch.distributed as dist
    from caesar_mrcnn_amd import ops
    from caesar_mrcnn_amd.config import run_py_config
    from caesar_mrcnn_amd.model import MaskRCNN
    dev = torch.device("cuda", int(os.environ.get("MRCNN_FORCE_DEVICE", local_rank)))
    size, nimg = 512, 4
    cfg = run_py_config(num_classes=4, imgsize=size, backbone="resnet101", images_per_gpu=nimg, gpu_count=world)
    with contextlib.redirect_stdout(sys.stderr):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
        model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
    dev_inputs = model._to_device(synthetic_batch(cfg, nimg, seed=4321 + rank))
    eng = model.engine

    def timed(nsteps):
        for _ in range(3):
            eng.forward_backward(*dev_inputs)
            eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(nsteps):
            losses = eng.forward_backward(*dev_inputs)
            eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
        host_issue.append((time.time() - t0) / nsteps)          # launches issued; the device may still be running
        torch.cuda.synchronize()
        return (time.time() - t0) / nsteps, losses

    host_issue = []

    out = {"workload": "BASELINE.json configs[4]: resnet101+FPN 512x512, nimg_per_gpu=4, train step, 16-bit weights/activations "
                       "on the 16-bit MFMA where built (mask head, FPN smoothing, shared RPN conv, class FCs, every bottleneck "
                       "block; both ROIAlign gathers and the mask head's ROIAlign adjoint read / write 16-bit tensors), float32 elsewhere "
                       "(what each costs in the profiled f16 step, profiles/r03_step_kernels_512_f16.txt: stem 0.19 ms forward + ~0.3 ms weight gradient, FPN laterals 0.28 ms + ~0.5 ms of "
                       "weight gradients on the side stream, the 4- to 16-column output layers < 0.2 ms, the class head's gather-form ROIAlign adjoint 1.0 ms on the auxiliary stream, "
                       "casts at the 16-bit / float32 seams 1.1 ms over 58 launches); float32 master weights and gradients, "
                       "loss scale 4096 + guarded optimiser step (f16)",
           "unit": "images/s"}
    steps = max(5, args.steps // 2)
    for tag, sparse, dt in (("f32", False, None), ("f16", False, torch.float16), ("bf16", False, torch.bfloat16),
                            ("f16_exact_zero_skip", True, torch.float16)):
        eng.sparse_mask_bwd, eng.head_dtype = sparse, dt
        t, losses = timed(steps)
        out["value_" + tag] = round(nimg / t, 3)
        out["ms_per_step_" + tag] = round(t * 1e3, 3)
        out["host_issue_ms_per_step_" + tag] = round(host_issue[-1] * 1e3, 3)
        out["losses_" + tag] = [round(float(v), 5) for v in losses.cpu().numpy()]
    out["skipped_steps_f16"] = eng.skipped_step_count()      # guarded optimiser: steps with non-finite float16 gradients
    eng.sparse_mask_bwd, eng.head_dtype = True, None
    # roofline of the dominant 16-bit kernel (mask-head 3x3 convolution, forward / data gradient), live HIP events
    M_rois = nimg * cfg.TRAIN_ROIS_PER_IMAGE
    flops = 2.0 * (M_rois * 196) * 256 * 2304
    for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        xm = torch.randn((M_rois, 14, 14, 256), device=dev).to(dt)
        wf, _ = ops.weights_to_h16(torch.randn((3, 3, 256, 256), device=dev) * 0.02, dt)
        bm = torch.zeros(256, device=dev); sc = torch.ones(256, device=dev)
        om = torch.empty((M_rois, 14, 14, 256), device=dev, dtype=dt)
        for _ in range(3):
            ops.conv2d_h16(xm, wf, (3, 3, 256, 256), bm, sc, bm, 1, "same", 1, out=om)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv2d_h16(xm, wf, (3, 3, 256, 256), bm, sc, bm, 1, "same", 1, out=om)
        e1.record()
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / 20
        ach = flops / (k_ms * 1e-3) / 1e12
        out["roofline_" + tag] = {"bound": "mfma", "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                  "frac": round(ach / 2500.0, 4), "traffic": None,
                                  "kernel": "conv_fwd_h16p_kernel<%s>, persistent 256x256 tiles in two staggered wave groups "
                                            "(+ conv_fwd_h16s_kernel for the last partial round); mask-head 3x3 conv, M=%d N=256 "
                                            "K=2304, %.1f GFLOP/launch, %.3f ms/launch" % (tag, M_rois * 196, flops / 1e9, k_ms)}
        # the same layer's weight gradient (pixel table + kernel + slab reduction), as the step runs it
        dym = torch.randn((M_rois, 14, 14, 256), device=dev).to(dt)
        dwm = torch.empty((3, 3, 256, 256), device=dev)
        for _ in range(3):
            ops.conv2d_wgrad_h16(xm, dym, (3, 3, 256, 256), 1, "same", dw=dwm)
        e0.record()
        for _ in range(20):
            ops.conv2d_wgrad_h16(xm, dym, (3, 3, 256, 256), 1, "same", dw=dwm)
        e1.record()
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / 20
        ach = flops / (k_ms * 1e-3) / 1e12
        out["roofline_wgrad_" + tag] = {"bound": "mfma", "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                        "frac": round(ach / 2500.0, 4), "traffic": None,
                                        "kernel": "conv_wgrad_h16p_kernel<%s> (256x256 tile per pixel split, two staggered wave "
                                                  "groups) + slab reduction (same layer, %.3f ms/launch)" % (tag, k_ms)}
        del xm, om, dym
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--backbone", default="resnet101", help="BASELINE.json metric: ResNet-101 256x256")
    ap.add_argument("--imgsize", type=int, default=256)
    ap.add_argument("--nimg", type=int, default=4, help="images per GPU (IMAGES_PER_GPU; configs[2]: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] (ResNet-50, 2 img/GPU) leg")
    ap.add_argument("--detect-iters", type=int, default=10)
    ap.add_argument("--dense-only", action="store_true", help="skip the exact-zero-skip timing loop (profiling)")
    ap.add_argument("--no-config4", action="store_true", help="skip the configs[4] leg (ResNet-101 512x512, 16-bit)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from caesar_mrcnn_amd.parallel import init_distributed

    rank, local_rank, world = init_distributed()
    if world != args.gpus and rank == 0:
        sys.stderr.write("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE\n" % (args.gpus, world))
    torch.cuda.set_device(torch.device("cuda", int(os.environ.get("MRCNN_FORCE_DEVICE", local_rank))))

    r_model_refs = []
    r = measure(args, args.backbone, args.nimg, rank, local_rank, world, full=True)
    second = None
    if world == 1 and not args.no_secondary and (args.backbone, args.nimg) != ("resnet50", 2):
        torch.cuda.empty_cache()
        try:
            second = measure(args, "resnet50", 2, rank, local_rank, world, full=False)
        except Exception as e:
            sys.stderr.write("configs[1] leg failed: %r\n" % (e,))

    cfg4 = None
    if world == 1 and not args.no_config4 and not args.dense_only:
        del r_model_refs[:]
        torch.cuda.empty_cache()
        try:
            cfg4 = measure_config4(args, rank, local_rank, world)
        except Exception as e:
            cfg4 = {"error": repr(e)}

    def rnd(v, n=3):
        return None if v is None else round(v, n)

    if rank == 0:
        out = {
            "metric": "train images/sec (whole node) + detect ms/image, ResNet-101 256x256",
            "value": round(r["images_per_s"], 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(r["ms_per_step"], 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s+FPN %dx%d 3-class (+bg), nimg_per_gpu=%d, train step (fwd+bwd+SGD%s), "
                                   "512 train ROIs, 2000 proposals" % (args.backbone, args.imgsize, args.imgsize, args.nimg,
                                                                       "+RCCL all-reduce" if world > 1 else ""),
                       "global_batch": args.nimg * world, "parallelism": "dp%d" % world,
                       "weights": "random init (Keras defaults)"},
            "detect_ms_per_image": rnd(r["detect_ms"]), "detect_ms_per_image_eager": rnd(r["detect_eager_ms"]),
            "detect_ms_per_image_batch8": rnd(r["detect_ms_b8"]),
            "detect_ms_per_image_f16_stages": rnd(r.get("detect_ms_f16")),
            "detect_e2e_ms_per_image": (r.get("detect_e2e") or {}).get("ms_per_image"),
            "detect_e2e": r.get("detect_e2e"),
            "value_direct_conv_kernels": None if r.get("images_per_s_direct") is None else round(r["images_per_s_direct"], 3),
            "note_direct_conv_kernels": "the same step with the Winograd path of the mask head switched off (direct 3x3 implicit-GEMM "
                                        "kernels, MRCNN_WINOGRAD=0): `value` computes the same layers in float32 through F(4x4,3x3) tiles plus "
                                        "a last row / column of 4x2, 2x4, 2x2 tiles -- 3.6x fewer matrix multiplications on 14x14 maps, "
                                        "1.3e-5 of the direct result's range; F(2x2,3x3) under MRCNN_WINOGRAD_TILE=2: 2.25x, 1.6e-6 "
                                        "(DESIGN.md 4.1d)",
            "value_exact_zero_skip": None if args.dense_only else round(r["images_per_s_sparse"], 3),
            "note_exact_zero_skip": "same step with the mask head (forward and backward) run on the <=168 positive-quota ROI "
                                    "rows per image only: the other rows are never read by the loss and carry exactly-zero "
                                    "gradient (identical losses and gradients, "
                                    "tests/test_engine_gpu.py::test_sparse_mask_backward_equals_dense); product default, "
                                    "never the headline value",
            "value_f16_mask_head": None if args.dense_only else round(r["images_per_s_f16_mask_head"], 3),
            "note_f16_mask_head": "BASELINE configs[4], stages 1-2: the headline (dense) step with the mask head (four 3x3 convolutions, "
                                  "transposed convolution, output stage; forward, data and weight gradient) on the fp16 matrix "
                                  "cores, float32 master weights / accumulation / gradients, loss scale 4096; reduced precision, "
                                  "never the headline value",
            "losses_last_step": [round(v, 5) for v in r["losses"]],
            "roofline": r["roofline"],
        }
        for k in ("roofline_wgrad", "roofline_direct_conv", "roofline_direct_wgrad"):
            if k in r:
                out[k] = r[k]
        if "train_loop" in r:
            out["train_loop"] = r["train_loop"]
        if "allreduce" in r:
            out["allreduce"] = r["allreduce"]
        if second is not None:
            out["config1_resnet50_nimg2"] = {
                "workload": "BASELINE.json configs[1]: resnet50+FPN %dx%d, nimg_per_gpu=2, 1 GPU train + detect" % (args.imgsize, args.imgsize),
                "value": round(second["images_per_s"], 3), "unit": "images/s", "ms_per_step": round(second["ms_per_step"], 3),
                "value_exact_zero_skip": None if args.dense_only else round(second["images_per_s_sparse"], 3),
                "detect_ms_per_image": rnd(second["detect_ms"])}
        if cfg4 is not None:
            out["config4_resnet101_512_f16"] = cfg4
        if "cpu_baseline" in r:
            out["cpu_baseline"] = r["cpu_baseline"]
        if "detect_error" in r:
            out["detect_error"] = r["detect_error"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
