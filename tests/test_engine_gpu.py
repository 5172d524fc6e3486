"""Graph-level parity of the HIP engine against the CPU oracle.

The network's discrete stages (top-k, NMS, ROI sampling) amplify float noise into different index
sets, so the comparison is staged ("teacher forced"): every stage of the oracle is fed the GPU's own
inputs for that stage and must reproduce the GPU's outputs -- exactly for indices, within the stated
tolerance for floats (fp32 contractions with a different summation order: rtol 1e-3 on activations
that went through ~50 conv layers, 2e-3 relative to the largest entry on gradients)."""
import numpy as np
import pytest
import torch

import mrcnn_oracle as orc

pytestmark = pytest.mark.gpu


def _small_cfg(backbone="custom", size=128, mode="training", **kw):
    from caesar_mrcnn_amd.config import run_py_config
    cfg = run_py_config(backbone=backbone, imgsize=size, mode=mode)
    cfg.POST_NMS_ROIS_TRAINING = 1500
    cfg.POST_NMS_ROIS_INFERENCE = 200
    cfg.TRAIN_ROIS_PER_IMAGE = 48
    cfg.MAX_GT_INSTANCES = 20
    cfg.RPN_TRAIN_ANCHORS_PER_IMAGE = 64
    cfg.DETECTION_MAX_INSTANCES = 30
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _scene(cfg, B, seed):
    """Synthetic tiles: noise + bright boxes.  Every GT instance is laid over a (pixel-rounded) anchor of
    scale >= 8 px so that, with RPN deltas kept small, some proposals reach IoU >= 0.5 and the
    detection-target / mask / bbox branches are exercised."""
    rng = np.random.default_rng(seed)
    S = int(cfg.IMAGE_SHAPE[0])
    G = cfg.MAX_GT_INSTANCES
    anchors = orc.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS,
                                           orc.compute_backbone_shapes(cfg.BACKBONE_STRIDES, cfg.IMAGE_SHAPE),
                                           cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
    ok = np.where((anchors[:, 0] >= 0) & (anchors[:, 1] >= 0) & (anchors[:, 2] <= S) & (anchors[:, 3] <= S) &
                  ((anchors[:, 2] - anchors[:, 0]) >= 6) & ((anchors[:, 3] - anchors[:, 1]) >= 6))[0]
    images = rng.normal(0, 8, (B, S, S, 3)).astype(np.float32) + 20
    gt_masks = np.zeros((B, S, S, G), bool)
    gt_boxes = np.zeros((B, G, 4), np.int32)
    gt_cls = np.zeros((B, G), np.int32)
    for b in range(B):
        for g in range(G - 4 - b):
            y1, x1, y2, x2 = np.round(anchors[rng.choice(ok)]).astype(int)
            gt_masks[b, y1:y2, x1:x2, g] = True
            gt_masks[b, (y1 + y2) // 2, (x1 + x2) // 2, g] = False
            images[b, y1:y2, x1:x2] += rng.uniform(60, 200)
            gt_boxes[b, g] = [y1, x1, y2, x2]
            gt_cls[b, g] = rng.integers(1, cfg.NUM_CLASSES)
    return images, gt_cls, gt_boxes, gt_masks


def _train_inputs(cfg, B, seed):
    images, gt_cls, gt_boxes, gt_masks = _scene(cfg, B, seed)
    anchors_px = orc.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS,
                                              orc.compute_backbone_shapes(cfg.BACKBONE_STRIDES, cfg.IMAGE_SHAPE),
                                              cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
    rng = np.random.RandomState(seed)
    rpn_match = np.zeros((B, anchors_px.shape[0], 1), np.int32)
    rpn_bbox = np.zeros((B, cfg.RPN_TRAIN_ANCHORS_PER_IMAGE, 4), np.float32)
    for b in range(B):
        n = int((gt_cls[b] > 0).sum())
        m, bb = orc.build_rpn_targets(anchors_px, gt_cls[b, :n], gt_boxes[b, :n], cfg.RPN_TRAIN_ANCHORS_PER_IMAGE,
                                      cfg.RPN_BBOX_STD_DEV, rng)
        rpn_match[b, :, 0], rpn_bbox[b] = m, bb
    meta = np.stack([orc.compose_image_meta(b, images[b].shape, images[b].shape, (0, 0) + images[b].shape[:2], 1.0,
                                            np.ones(cfg.NUM_CLASSES, np.int32)) for b in range(B)])
    keys = np.random.default_rng(seed + 1).uniform(0, 1, (B, cfg.POST_NMS_ROIS_TRAINING)).astype(np.float32)
    return [images, meta, rpn_match, rpn_bbox, gt_cls, gt_boxes, gt_masks], keys


def _weights(cfg, seed, damp=None):
    """damp: factor on the BatchNorm gamma that closes every bottleneck branch.  ResNet-101 sums 33 residual branches;
    with unit-scale random branches the activations grow ~2x in variance per block and the RPN saturates (all proposals
    clip to the full tile, no positive ROI): 0.25 keeps the trunk's output in range so that every branch of the training
    graph carries signal."""
    from caesar_mrcnn_amd.params import ParamLayout, init_weights, deconv_gemm_to_keras
    w = init_weights(ParamLayout(cfg), seed=seed, perturb_bn=True)
    if damp is not None:
        for k in w:
            if k.endswith("_branch2c/gamma"):
                w[k] = w[k] * np.float32(damp)
    w["mrcnn_mask_deconv/kernel"] = deconv_gemm_to_keras(w["mrcnn_mask_deconv/kernel"])   # Keras layout at the boundary
    # keep RPN box deltas small so proposals stay near their anchors and some reach IoU >= 0.5 with the GT
    w["rpn_bbox_pred/kernel"] = w["rpn_bbox_pred/kernel"] * np.float32(0.02)
    w["rpn_bbox_pred/bias"] = w["rpn_bbox_pred/bias"] * np.float32(0.02)
    return w


def _check_gradients(g, o, names, tag=""):
    """Every parameter gradient against the oracle's autograd.  Bar: max |g - ref| <= 5e-3 of max |ref| per tensor.  An
    activation that sits on a ReLU boundary can land on different sides in two float32 evaluations (different summation
    orders), which switches ONE (pixel, channel) of that layer's gradient on or off -- tools/grad_diag.py shows exactly that
    signature (one channel of one layer, one element of its bias / beta gradient).  Such a tensor may reach 2e-2 in the
    max norm as long as its L2 error stays <= 5e-3 (a genuine kernel error is not confined to one channel), and at most
    1 % of the tensors (at least one)."""
    bad, flips = [], []
    for name in names:
        rg = o.w[name].grad.numpy().astype(np.float64)
        d = g[name].astype(np.float64) - rg
        scale = max(float(np.abs(rg).max()), 1e-8)
        err = float(np.abs(d).max()) / scale
        l2 = float(np.linalg.norm(d)) / max(float(np.linalg.norm(rg)), 1e-12)
        if err > 5e-3:
            (flips if (err <= 2e-2 and l2 <= 5e-3) else bad).append((name, err, l2, scale))
    assert not bad, (tag, bad[:8])
    # how many tensors needed the flip allowance is part of the test's output (pytest -rP / -s shows it; a drift from the 0-2
    # seen so far is visible before it reaches the cap)
    print("[gradient check %s] %d tensors, %d within 5e-3, %d tolerated as ReLU-boundary flips (cap %d): %s" % (
        tag, len(names), len(names) - len(flips), len(flips), max(1, len(names) // 100),
        ", ".join("%s max %.2e l2 %.2e" % f[:3] for f in flips)))
    assert len(flips) <= max(1, len(names) // 100), (tag, flips)
    return len(flips)


def _close(got, ref, rtol, name):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = max(float(np.abs(ref).max()), 1e-12)
    err = float(np.abs(got - ref).max()) / scale
    assert err <= rtol, "%s: max error %.3g of max |ref| %.3g (allowed %.3g)" % (name, err, scale, rtol)


@pytest.mark.parametrize("backbone,size", [("custom", 128), ("resnet50", 128)])
def test_inference_graph_staged(dev, backbone, size):
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg(backbone, size, mode="inference")
    w = _weights(cfg, 3)
    model = MaskRCNN("inference", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    images, _, _, _ = _scene(cfg, 1, 7)
    molded, metas, windows = model.mold_inputs([images[0].astype(np.float32)])
    molded = molded.astype(np.float32)
    win = torch.tensor(orc.norm_boxes(windows.astype(np.float32), molded.shape[1:3]), device=dev)
    out = model.engine.infer(torch.tensor(molded, device=dev), win)
    torch.cuda.synchronize()
    o = orc.OracleMaskRCNN(cfg, w)
    x = torch.tensor(molded)
    area = float(size * size)
    with torch.no_grad():
        pyr = o.fpn(*o.backbone(x))
        _, rp, rb = o.rpn(pyr)
    for i, p in enumerate(pyr):
        _close(out["pyramid"][i].cpu(), p, 1e-3, "P%d" % (i + 2))
    _close(out["rpn_class"].cpu(), rp, 1e-3, "rpn_class")
    _close(out["rpn_bbox"].cpu(), rb, 1e-3, "rpn_bbox")
    # proposal layer on the GPU's RPN outputs
    anchors = orc.get_anchors(cfg, molded.shape[1:])
    np.testing.assert_array_equal(model.engine.anchors(molded.shape[1:]).cpu().numpy(), anchors)
    np.testing.assert_array_equal(model.get_anchors(molded.shape[1:]), anchors)
    rois_ref = o.proposal_layer(out["rpn_class"].cpu(), out["rpn_bbox"].cpu(), anchors, cfg.POST_NMS_ROIS_INFERENCE)
    rois = out["rpn_rois"].cpu().numpy()
    # allow a handful of ulp-level NMS flips; rows must agree for >= 98 %
    same = np.mean(np.all(np.abs(rois - rois_ref) < 1e-5, axis=-1))
    assert same >= 0.98, same
    # heads on the GPU's ROIs and the GPU's pyramid
    gp = [p.cpu() for p in out["pyramid"][:4]]
    with torch.no_grad():
        _, probs, bbox = o.classifier_head(rois, gp, area)
    _close(out["mrcnn_class"].cpu(), probs, 1e-3, "mrcnn_class")
    _close(out["mrcnn_bbox"].cpu(), bbox, 1e-3, "mrcnn_bbox")
    det_ref = o.refine_detections(rois[0], out["mrcnn_class"][0].cpu().numpy(), out["mrcnn_bbox"][0].cpu().numpy(),
                                  win[0].cpu().numpy())
    det = out["detections"][0].cpu().numpy()
    np.testing.assert_array_equal(det[:, 4:], det_ref[:, 4:])
    np.testing.assert_allclose(det[:, :4], det_ref[:, :4], atol=2e-6)
    with torch.no_grad():
        masks = o.mask_head(det[None, :, :4], gp, area)
    _close(out["mrcnn_mask"].cpu(), masks, 1e-3, "mrcnn_mask")
    # the public entry point returns the reference's result structure
    res = model.detect([images[0].astype(np.float32)])[0]
    assert set(res) == {"rois", "class_ids", "scores", "masks"}
    assert res["masks"].shape[:2] == images[0].shape[:2] and res["masks"].shape[2] == res["rois"].shape[0]


@pytest.mark.parametrize("backbone,size,dice", [("custom", 128, False), ("resnet50", 128, False), ("custom", 128, True)])
def test_training_step_staged(dev, backbone, size, dice):
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg(backbone, size, MASK_LOSS_FUNCTION="dice_coef_loss" if dice else "binary_crossentropy")
    cfg.LOSS_WEIGHTS = dict(cfg.LOSS_WEIGHTS, rpn_bbox_loss=0.7, mrcnn_mask_loss=1.3)
    B = 2
    w = _weights(cfg, 11)
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    inputs, keys = _train_inputs(cfg, B, 5)
    losses = model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
    torch.cuda.synchronize()
    eng = model.engine
    last = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in eng.last.items() if k != "pyramid"}
    images, meta, rpn_match, rpn_bbox_t, gt_cls, gt_boxes, gt_masks = inputs
    o = orc.OracleMaskRCNN(cfg, w, requires_grad=True)
    anchors = orc.get_anchors(cfg, images.shape[1:])
    # (a) sampling stage on the GPU's proposals
    S = float(size)
    gtn = ((gt_boxes.astype(np.float32) - np.array([0, 0, 1, 1], np.float32)) / np.float32(S - 1)).astype(np.float32)
    npos = 0
    for b in range(B):
        r_rois, r_cls, r_bb, r_m, (P, N) = o.detection_targets(last["rpn_rois"][b], gt_cls[b], gtn[b], gt_masks[b], keys[b])
        assert tuple(last["counts"][b]) == (P, N)
        np.testing.assert_array_equal(last["rois"][b], r_rois)
        np.testing.assert_array_equal(last["target_class_ids"][b], r_cls)
        np.testing.assert_allclose(last["target_bbox"][b], r_bb, rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(last["target_mask"][b], r_m)       # byte work: exact (see test_detection_targets)
        npos += P
    assert npos > 0, "scene produced no positive ROIs; the mask/bbox losses would be untested"
    # (b) differentiable part on identical ROIs / targets
    forced = {k: last[k] for k in ("rois", "target_class_ids", "target_bbox", "target_mask")}
    ref = o.forward_training(images, rpn_match, rpn_bbox_t.astype(np.float32), gt_cls, gt_boxes, gt_masks,
                             meta[:, 12:].astype(np.int32), anchors, keys, forced=forced)
    _close(last["rpn_class_logits"], ref["rpn_class_logits"].detach(), 1e-3, "rpn_class_logits")
    _close(last["mrcnn_class_logits"], ref["mrcnn_class_logits"].detach(), 1e-3, "mrcnn_class_logits")
    # product default: the mask head runs on the positive quota of each image only (rows the loss can read)
    quota = int(cfg.TRAIN_ROIS_PER_IMAGE * cfg.ROI_POSITIVE_RATIO)
    assert eng.sparse_mask_bwd and 0 < quota < cfg.TRAIN_ROIS_PER_IMAGE
    _close(last["mrcnn_mask"][:, :quota], ref["mrcnn_mask"].detach().numpy()[:, :quota], 1e-3, "mrcnn_mask")
    assert not last["mrcnn_mask"][:, quota:].any()
    np.testing.assert_allclose(losses.cpu().numpy(), [float(l.detach()) for l in ref["losses"]], rtol=2e-3, atol=1e-5)
    total = o.total_loss(ref["losses"])
    total.backward()
    g = eng.get_weights(grads=True)
    # the GPU buffer holds the data gradients; the regulariser is added by grad_prepare -> compare after it
    eng.apply_gradients(0.0, 0.0, world_size=1)      # lr = 0: only prepares (adds L2) and clips into eng.grads
    torch.cuda.synchronize()
    g = eng.get_weights(grads=True)
    gnorm = float(eng.sumsq.sqrt())
    ref_sq = 0.0
    bad = []
    for name in eng.layout.offsets:
        rg = o.w[name].grad
        assert rg is not None, name
        rg = rg.numpy()
        ref_sq += float((rg.astype(np.float64) ** 2).sum())
        scale = max(float(np.abs(rg).max()), 1e-8)
        err = float(np.abs(g[name] - rg).max()) / scale
        if err > 5e-3:
            bad.append((name, err, scale))
    assert not bad, "gradient mismatch (name, rel err, max|ref|): %s" % bad[:8]
    np.testing.assert_allclose(gnorm, np.sqrt(ref_sq), rtol=1e-3)


def test_optimizer_update_matches_oracle(dev):
    """grad_prepare + global-norm clip + SGD-momentum on the real flat buffer."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg("custom", 128)
    w = _weights(cfg, 13)
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    eng = model.engine
    rng = np.random.default_rng(3)
    G = (rng.standard_normal(eng.layout.total) * 0.05).astype(np.float32)
    V = (rng.standard_normal(eng.layout.total) * 0.01).astype(np.float32)
    live = np.zeros(eng.layout.total, bool)          # alignment padding between tensors carries no gradient
    for _, off, n, _, _ in eng.layout.segments:
        live[off:off + n] = True
    G[~live] = 0
    eng.grads.copy_(torch.tensor(G)); eng.momentum.copy_(torch.tensor(V))
    world = 4
    params = {k: np.asarray(v, np.float32).copy() for k, v in eng.get_weights().items() if k in eng.layout.offsets}
    grads, vel = {}, {}
    for name, (off, n, shape) in eng.layout.offsets.items():
        p = eng.params[off:off + n].cpu().numpy()
        l2 = 0.0 if ("gamma" in name or "beta" in name) else 2.0 * cfg.WEIGHT_DECAY / n
        grads[name] = G[off:off + n] * np.float32(1.0 / world) + np.float32(l2) * p
        vel[name] = V[off:off + n].copy()
        params[name] = p.copy()
    eng.apply_gradients(0.01, 0.9, world_size=world)
    torch.cuda.synchronize()
    norm = orc.sgd_step(params, grads, vel, 0.01, 0.9, cfg.GRADIENT_CLIP_NORM)
    assert norm > cfg.GRADIENT_CLIP_NORM            # the clip is active in this test
    np.testing.assert_allclose(float(eng.sumsq.sqrt()), norm, rtol=1e-4)
    for name, (off, n, shape) in eng.layout.offsets.items():
        np.testing.assert_allclose(eng.params[off:off + n].cpu().numpy(), params[name], rtol=1e-5, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(eng.momentum[off:off + n].cpu().numpy(), vel[name], rtol=1e-5, atol=1e-7, err_msg=name)


def test_frozen_layers_and_heads_preset(dev):
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg("custom", 128)
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=_weights(cfg, 17))
    model.set_trainable("heads", verbose=0)
    before = model.engine.params.clone()
    inputs, keys = _train_inputs(cfg, 2, 9)
    model.compile(0.01, 0.9)
    model.train_on_batch(inputs, rand_keys=keys)
    torch.cuda.synchronize()
    changed = (model.engine.params != before).cpu().numpy()
    for (name, off, n, _, _), t in zip(model.engine.layout.segments, model.engine.trainable_host):
        assert changed[off:off + n].any() == bool(t) or (t and n < 8), name


def test_taped_steps_follow_set_trainable(dev):
    """A second train(..., layers=...) on the same model at the same rates replays the launch tape recorded under the first
    trainable mask (engine.step_taped).  The recorded grad_prepare / sgd_momentum calls carry the ADDRESS of the per-granule
    trainable / weight-decay coefficients, so set_trainable must refresh that buffer in place (round 2 re-allocated it: the
    replay then read freed memory).  'heads' for two steps, then 'all' for two steps, taped against eager: frozen layers stay
    bit-identical in phase 1 and move in phase 2 in both runs, final parameters agree, ONE tape serves both phases."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg("custom", 128)
    w = _weights(cfg, 17)
    batches = [_train_inputs(cfg, 2, 9), _train_inputs(cfg, 2, 13)]
    snaps = {}
    for taped in (False, True):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        eng = model.engine
        model.compile(0.01, 0.9)
        start = eng.params.clone()
        coef_ptr = None
        for layers in ("heads", "all"):
            model.set_trainable(layers, verbose=0)
            assert coef_ptr in (None, eng.gran_coef.data_ptr())          # same buffer, new contents
            coef_ptr = eng.gran_coef.data_ptr()
            for s in range(2):
                di = model._to_device(*batches[s])
                if taped:
                    eng.step_taped(di, 0.01, 0.9)
                else:
                    eng.forward_backward(*di)
                    eng.apply_gradients(0.01, 0.9, 1)
            torch.cuda.synchronize()
            snaps[(taped, layers)] = eng.params.cpu().numpy().copy()
        if taped:
            assert len(eng._train_tapes) == 1
    start = start.cpu().numpy()
    model.set_trainable("heads", verbose=0)
    heads_mask = list(model.engine.trainable_host)
    for taped in (False, True):
        a, b = snaps[(taped, "heads")], snaps[(taped, "all")]
        for (name, off, n, _, _), t in zip(model.engine.layout.segments, heads_mask):
            if not t:
                assert np.array_equal(a[off:off + n], start[off:off + n]), (taped, name)       # frozen in phase 1
                assert n < 8 or not np.array_equal(b[off:off + n], start[off:off + n]), (taped, name)   # trained in phase 2
    for layers in ("heads", "all"):
        ref, got = snaps[(False, layers)], snaps[(True, layers)]
        assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max(), layers


def test_sparse_mask_backward_equals_dense(dev):
    """Running the mask head (forward and backward) on the positive quota only changes neither the losses
    nor the gradients: the skipped rows are never read by the loss and carry exactly-zero gradient."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg("custom", 128)
    w = _weights(cfg, 19)
    inputs, keys = _train_inputs(cfg, 2, 21)
    grads = []
    for sparse in (True, False):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        model.engine.sparse_mask_bwd = sparse
        losses = model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
        torch.cuda.synchronize()
        grads.append((losses.cpu().numpy(), model.engine.grads.cpu().numpy().copy(),
                      model.engine.last["mrcnn_mask"].cpu().numpy().copy()))
    quota = int(cfg.TRAIN_ROIS_PER_IMAGE * cfg.ROI_POSITIVE_RATIO)
    np.testing.assert_allclose(grads[0][2][:, :quota], grads[1][2][:, :quota], rtol=1e-5, atol=1e-6)
    assert not grads[0][2][:, quota:].any() and grads[1][2][:, quota:].any()       # dense computes every row
    np.testing.assert_allclose(grads[0][0], grads[1][0], rtol=1e-6)
    scale = np.abs(grads[1][1]).max()
    assert np.abs(grads[0][1] - grads[1][1]).max() <= 2e-5 * scale
    assert np.abs(grads[1][1]).sum() > 0


def test_detect_on_reference_fits_cutout_and_graph_replay(dev):
    """cfg-1 plumbing: FITS -> zscale -> uint8 RGB -> resize 132->256 -> detect, ResNet-50 random init;
    the HIP-graph replay returns exactly what the eager launches return."""
    import os
    from caesar_mrcnn_amd import fits
    from caesar_mrcnn_amd.config import run_py_config
    from caesar_mrcnn_amd.model import MaskRCNN
    img, _ = fits.read_fits(os.path.join(os.path.dirname(__file__), "golden", "galaxy0002.fits"))
    cfg = run_py_config(backbone="resnet50", imgsize=256, mode="inference")
    model = MaskRCNN("inference", cfg, "/tmp/mrcnn_logs", device=dev, seed=3)
    molded, metas, windows = model.mold_inputs([img])
    assert molded.shape == (1, 256, 256, 3) and tuple(windows[0]) == (0, 0, 256, 256)
    model.use_hip_graph = False
    eager = model._predict_molded(molded, metas)
    model.use_hip_graph = True
    g1 = model._predict_molded(molded, metas)
    g2 = model._predict_molded(molded, metas)          # replay
    for a, b, c in zip(eager, g1, g2):
        assert np.array_equal(a, b) and np.array_equal(b, c)
    res = model.detect([img])[0]
    assert res["masks"].shape[:2] == (132, 132) and res["rois"].dtype == np.int32
    assert res["rois"].shape[0] == res["class_ids"].shape[0] == res["scores"].shape[0] == res["masks"].shape[2]
    assert res["rois"].shape[0] > 0 and res["rois"].max() <= 132 and (res["class_ids"] > 0).all()
    # detect() un-molds on the device (mrcnn_unmold_masks); the host statement of mrcnn/model.py:2558-2621 on the same graph
    # outputs (utils.unmold_mask per detection) must give the identical boxes and boolean planes
    from caesar_mrcnn_amd import utils
    det, mm = g1[0][0], g1[3][0]
    boxes, class_ids, scores, rows = model._unmold_boxes(det, img.shape, molded[0].shape, windows[0])
    want = np.stack([utils.unmold_mask(mm[r, :, :, c], b, img.shape) for r, c, b in zip(rows, class_ids, boxes)], axis=-1)
    assert res["masks"].dtype == np.bool_ and np.array_equal(res["masks"], want) and want.any()
    assert np.array_equal(res["rois"], boxes) and np.array_equal(res["class_ids"], class_ids) and np.array_equal(res["scores"], scores)
    b2, c2, s2, m2 = model.unmold_detections(det, mm, img.shape, molded[0].shape, windows[0])     # the public method, host arrays in
    assert np.array_equal(m2, want) and np.array_equal(b2, boxes)


def _full_cfg(backbone, size, mode="training", nimg=2):
    from caesar_mrcnn_amd.config import run_py_config
    return run_py_config(backbone=backbone, imgsize=size, mode=mode, images_per_gpu=nimg)


def test_cfg2_full_size_training_step_r50_256(dev):
    """BASELINE configs[1] at its real sizes: ResNet-50+FPN 256x256, 3 classes + bg, nimg_per_gpu=2, 512 train
    ROIs, 2000 proposals, 300 GT slots -- losses and every parameter gradient against the oracle's autograd."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _full_cfg("resnet50", 256)
    B = 2
    w = _weights(cfg, 29)
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    inputs, keys = _train_inputs(cfg, B, 41)
    losses = model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
    torch.cuda.synchronize()
    eng = model.engine
    last = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in eng.last.items() if k != "pyramid"}
    assert int(last["counts"][:, 0].sum()) > 0
    images, meta, rpn_match, rpn_bbox_t, gt_cls, gt_boxes, gt_masks = inputs
    o = orc.OracleMaskRCNN(cfg, w, requires_grad=True)
    forced = {k: last[k] for k in ("rois", "target_class_ids", "target_bbox", "target_mask")}
    ref = o.forward_training(images, rpn_match, rpn_bbox_t.astype(np.float32), gt_cls, gt_boxes, gt_masks,
                             meta[:, 12:].astype(np.int32), orc.get_anchors(cfg, images.shape[1:]), keys, forced=forced)
    np.testing.assert_allclose(losses.cpu().numpy(), [float(l.detach()) for l in ref["losses"]], rtol=2e-3, atol=1e-5)
    o.total_loss(ref["losses"]).backward()
    eng.apply_gradients(0.0, 0.0, world_size=1)
    torch.cuda.synchronize()
    g = eng.get_weights(grads=True)
    _check_gradients(g, o, list(eng.layout.offsets), "cfg2")


def test_cfg3_r101_256_nimg4_training_step(dev):
    """BASELINE configs[2], the per-rank workload of the 8-GPU data-parallel job (and bench.py's headline workload):
    ResNet-101+FPN 256x256, nimg_per_gpu = 4, 512 train ROIs, 2000 proposals.  Losses and every parameter gradient
    against the oracle's autograd (mrcnn/model.py:1935-2166, :2255-2291), with the mask head run densely (all 2048
    ROI rows, the headline leg) and on the positive quota only (product default): one oracle pass serves both,
    because the sampled ROIs / targets are fed to it ("forced") and are identical in the two modes."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _full_cfg("resnet101", 256, nimg=4)
    B = 4
    w = _weights(cfg, 53, damp=0.25)
    inputs, keys = _train_inputs(cfg, B, 57)
    images, meta, rpn_match, rpn_bbox_t, gt_cls, gt_boxes, gt_masks = inputs
    runs = {}
    for mode in ("dense", "sparse"):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        eng = model.engine
        eng.sparse_mask_bwd = mode == "sparse"
        losses = model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
        torch.cuda.synchronize()
        last = {k: v.cpu().numpy() for k, v in eng.last.items() if torch.is_tensor(v)}
        eng.apply_gradients(0.0, 0.0, world_size=1)          # lr 0: adds the L2 term, leaves the clipped gradient in place
        torch.cuda.synchronize()
        runs[mode] = (losses.cpu().numpy(), eng.get_weights(grads=True), last)
        eng_names = list(eng.layout.offsets)
        del model, eng
        torch.cuda.empty_cache()
    last = runs["dense"][2]
    assert (last["counts"][:, 0] > 0).all(), last["counts"]       # every image contributes positive ROIs
    for k in ("rois", "target_class_ids", "target_bbox", "target_mask"):
        np.testing.assert_array_equal(runs["sparse"][2][k], last[k])
    o = orc.OracleMaskRCNN(cfg, w, requires_grad=True)
    forced = {k: last[k] for k in ("rois", "target_class_ids", "target_bbox", "target_mask")}
    ref = o.forward_training(images, rpn_match, rpn_bbox_t.astype(np.float32), gt_cls, gt_boxes, gt_masks,
                             meta[:, 12:].astype(np.int32), orc.get_anchors(cfg, images.shape[1:]), keys, forced=forced)
    o.total_loss(ref["losses"]).backward()
    want = [float(l.detach()) for l in ref["losses"]]
    for mode, (losses, g, _) in runs.items():
        np.testing.assert_allclose(losses, want, rtol=2e-3, atol=1e-5, err_msg=mode)
        _check_gradients(g, o, eng_names, mode)


def test_cfg4_graph_replay_equals_eager_1024(dev):
    """configs[3] shape through the HIP graph: ResNet-101, 1024x1024 (A = 261 888 -> multi-workgroup top-k), captured
    once and replayed; every replay must return exactly the eager result, and the top-k health words must show a clean
    selection (collected == announced == 6000, nothing refused).  This is the replay that ended in a memory fault in
    round 1: run once, not in a loop."""
    from caesar_mrcnn_amd import ops
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _full_cfg("resnet101", 1024, mode="inference", nimg=1)
    model = MaskRCNN("inference", cfg, "/tmp/mrcnn_logs", device=dev, seed=7)
    rng = np.random.default_rng(11)
    x = torch.tensor(rng.uniform(0, 255, (1, 1024, 1024, 3)).astype(np.float32), device=dev)
    win = torch.tensor([[0., 0., 1., 1.]], device=dev)
    keys = ("rpn_rois", "detections", "mrcnn_mask", "mrcnn_class")
    out = model.engine.infer(x, win)
    eager = {k: out[k].cpu().numpy().copy() for k in keys}
    assert np.abs(eager["rpn_rois"]).sum() > 0
    st = ops.proposal_status(out["rpn_class"], cfg.PRE_NMS_LIMIT, cfg.POST_NMS_ROIS_INFERENCE)
    assert st.tolist() == [[6000, 6000, 0]], st
    for rep in range(3):
        out = model.engine.infer_graphed(x, win)
        torch.cuda.synchronize()
        for k in keys:
            assert np.array_equal(out[k].cpu().numpy(), eager[k]), (k, rep)
    # a second input through the same graph (static input buffers are refilled, nothing else changes)
    x2 = torch.tensor(rng.uniform(0, 255, (1, 1024, 1024, 3)).astype(np.float32), device=dev)
    e2 = model.engine.infer(x2, win)["detections"].cpu().numpy().copy()
    g2 = model.engine.infer_graphed(x2, win)["detections"].cpu().numpy()
    assert np.array_equal(e2, g2)


def test_cfg4_r101_1024_tile_inference_staged(dev):
    """BASELINE configs[3]: ResNet-101, 1024x1024 tile, 261 888 anchors -> top-6000 -> NMS -> 1000 proposals,
    ROIAlign over a 256x256 P2: RPN outputs vs the oracle, proposal layer and detection layer index-exact on
    the GPU's own inputs."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _full_cfg("resnet101", 1024, mode="inference")
    w = _weights(cfg, 31)
    model = MaskRCNN("inference", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    rng = np.random.default_rng(5)
    img = (rng.normal(0, 20, (1, 1024, 1024, 3)) + 60).astype(np.float32)
    win = torch.tensor([[0.0, 0.0, 1.0, 1.0]], device=dev)
    out = model.engine.infer(torch.tensor(img, device=dev), win)
    torch.cuda.synchronize()
    assert out["rpn_class"].shape == (1, 261888, 2)
    o = orc.OracleMaskRCNN(cfg, w)
    with torch.no_grad():
        pyr = o.fpn(*o.backbone(torch.tensor(img)))
        _, rp, rb = o.rpn(pyr)
    _close(out["rpn_class"].cpu(), rp, 2e-3, "rpn_class")
    _close(out["rpn_bbox"].cpu(), rb, 2e-3, "rpn_bbox")
    anchors = orc.get_anchors(cfg, (1024, 1024, 3))
    rois_ref, det = o.proposal_layer(out["rpn_class"].cpu(), out["rpn_bbox"].cpu(), anchors, 1000, detail=True)
    rois = out["rpn_rois"].cpu().numpy()
    same = np.mean(np.all(np.abs(rois - rois_ref) < 1e-5, axis=-1))
    assert same >= 0.98, same
    gp = [p.cpu() for p in out["pyramid"][:4]]
    with torch.no_grad():
        _, probs, bbox = o.classifier_head(rois, gp, float(1024 * 1024))
    _close(out["mrcnn_class"].cpu(), probs, 2e-3, "mrcnn_class")
    det_ref = o.refine_detections(rois[0], out["mrcnn_class"][0].cpu().numpy(), out["mrcnn_bbox"][0].cpu().numpy(),
                                  np.array([0, 0, 1, 1], np.float32))
    d = out["detections"][0].cpu().numpy()
    np.testing.assert_array_equal(d[:, 4:], det_ref[:, 4:])
    np.testing.assert_allclose(d[:, :4], det_ref[:, :4], atol=2e-6)


def test_fused_mask_output_backward_equals_unfused(dev):
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg("custom", 128)
    w = _weights(cfg, 37)
    inputs, keys = _train_inputs(cfg, 2, 43)
    grads = []
    for fused in (True, False):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        model.engine.fused_mask_out_bwd = fused
        model.engine.sparse_mask_bwd = False
        model.train_on_batch(inputs, rand_keys=keys, apply=False)
        torch.cuda.synchronize()
        grads.append(model.engine.grads.cpu().numpy().copy())
    scale = np.abs(grads[1]).max()
    assert np.abs(grads[0] - grads[1]).max() <= 2e-5 * scale


@pytest.mark.parametrize("head_dtype", [None, "float16"])
def test_deferred_mask_wgrad_equals_inline(dev, head_dtype):
    """engine.defer_mask_wgrad (an A/B switch, off by default) only moves the mask head's weight-gradient launches (to the auxiliary stream, beside the
    backbone's backward pass): same kernels on the same operands, so losses and the whole gradient buffer agree up to the
    float atomics of the other kernels.  ResNet-50 256x256 so that the large LDS-DMA weight-gradient kernel (and its
    occupancy cap) is the one that runs."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _full_cfg("resnet50", 256)
    w = _weights(cfg, 47)
    inputs, keys = _train_inputs(cfg, 2, 49)
    res = []
    for defer in (True, False):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        eng = model.engine
        eng.defer_mask_wgrad = defer
        eng.sparse_mask_bwd = False
        if head_dtype:
            eng.head_dtype = getattr(torch, head_dtype)
        for _ in range(2):                                   # twice: the second step reuses the arena and every side stream
            losses = model.train_on_batch(inputs, rand_keys=keys, apply=False)
        torch.cuda.synchronize()
        res.append((losses.cpu().numpy(), eng.grads.cpu().numpy().copy()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-6)
    scale = np.abs(res[1][1]).max()
    assert np.abs(res[0][1] - res[1][1]).max() <= 2e-5 * scale and scale > 0


def test_device_rpn_targets_in_training_step(dev):
    """A step fed with rpn_match / rpn_bbox = None (targets built on the GPU, only the used GT-mask planes
    uploaded) equals the step fed with the oracle's host-built targets for the same draw keys."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg("custom", 128)
    w = _weights(cfg, 23)
    inputs, keys = _train_inputs(cfg, 2, 25)
    images, meta, _, _, gt_cls, gt_boxes, gt_masks = inputs
    anchors_px = orc.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS,
                                              orc.compute_backbone_shapes(cfg.BACKBONE_STRIDES, cfg.IMAGE_SHAPE),
                                              cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
    rpn_keys = np.random.default_rng(5).uniform(0, 1, (2, anchors_px.shape[0])).astype(np.float32)
    rpn_match = np.zeros((2, anchors_px.shape[0], 1), np.int32)
    rpn_bbox = np.zeros((2, cfg.RPN_TRAIN_ANCHORS_PER_IMAGE, 4), np.float32)
    for b in range(2):
        n = int((gt_cls[b] > 0).sum())
        m, bb = orc.build_rpn_targets(anchors_px, gt_cls[b, :n], gt_boxes[b, :n], cfg.RPN_TRAIN_ANCHORS_PER_IMAGE,
                                      cfg.RPN_BBOX_STD_DEV, orc.KeyedChoice(rpn_keys[b]))
        rpn_match[b, :, 0], rpn_bbox[b] = m, bb
    res = []
    for device_side in (False, True):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        batch = [images, meta, None if device_side else rpn_match, None if device_side else rpn_bbox,
                 gt_cls, gt_boxes, gt_masks]
        dev_inputs = model._to_device(batch, keys, rpn_keys=rpn_keys)
        if device_side:
            np.testing.assert_array_equal(dev_inputs[1].cpu().numpy(), rpn_match)
            np.testing.assert_allclose(dev_inputs[2].cpu().numpy(), rpn_bbox, rtol=2e-6, atol=1e-6)
            np.testing.assert_array_equal(dev_inputs[5].cpu().numpy(), gt_masks.astype(np.uint8))
        losses = model.engine.forward_backward(*dev_inputs)
        torch.cuda.synchronize()
        res.append((losses.cpu().numpy(), model.engine.grads.cpu().numpy().copy()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-5)
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-5 * np.abs(res[0][1]).max()
    assert (rpn_match == 1).sum() > 0


def test_fused_dgrad_epilogue_equals_unfused(dev):
    """engine.fused_dgrad_epilogue (mask-head data-gradient convs apply the lower layer's epilogue backward) changes no
    gradient beyond float32 atomics order; ResNet-50 256x256 so that the layers are large enough to be fused."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _full_cfg("resnet50", 256)
    w = _weights(cfg, 43)
    inputs, keys = _train_inputs(cfg, 2, 45)
    res = []
    for fused in (True, False):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        model.engine.fused_dgrad_epilogue = fused
        model.engine.sparse_mask_bwd = False                      # all 1024 ROI rows: 3136 tiles per conv
        losses = model.train_on_batch(inputs, rand_keys=keys, apply=False)
        torch.cuda.synchronize()
        res.append((losses.cpu().numpy(), model.engine.grads.cpu().numpy().copy()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=5e-6)       # the loss reductions use float atomics: their order moves the
                                                                      # class loss by up to ~16 ulp between two runs (seen: 1.5e-6)
    scale = np.abs(res[1][1]).max()
    assert np.abs(res[0][1] - res[1][1]).max() <= 2e-5 * scale and scale > 0


def test_winograd_split_forward_equals_whole(dev):
    """engine.winograd_split (the mask head's Winograd layers as two half-batch chains on two streams, off by default since the
    mixed tiling) against the whole-batch chain: same losses, gradients equal up to the order of the float32 atomics and of the
    two accumulating weight-gradient calls per layer.  ResNet-50 256x256, 2 images = 1024 ROI rows = two halves of 512."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _full_cfg("resnet50", 256)
    w = _weights(cfg, 47)
    inputs, keys = _train_inputs(cfg, 2, 49)
    res = []
    for split in (True, False):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        if not model.engine.winograd:
            pytest.skip("MRCNN_WINOGRAD=0: nothing to split")
        model.engine.winograd_split = split
        model.engine.sparse_mask_bwd = False
        losses = model.train_on_batch(inputs, rand_keys=keys, apply=False)
        torch.cuda.synchronize()
        res.append((losses.cpu().numpy(), model.engine.grads.cpu().numpy().copy()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=5e-6)
    scale = np.abs(res[1][1]).max()
    assert np.abs(res[0][1] - res[1][1]).max() <= 2e-5 * scale and scale > 0


def test_graph_replay_with_multiworkgroup_topk_512(dev):
    """512x512 (A = 65 472 anchors >= 32 768): the proposal layer selects its top-k over many workgroups; several
    HIP-graph replays must return exactly what the eager launches return (the selection zeroes its counters itself --
    with a kernel, a captured hipMemsetAsync node was not reliable inside the large graph)."""
    from caesar_mrcnn_amd.config import run_py_config
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = run_py_config(backbone="resnet50", imgsize=512, mode="inference")
    model = MaskRCNN("inference", cfg, "/tmp/mrcnn_logs", device=dev, seed=5)
    rng = np.random.default_rng(9)
    x = torch.tensor(rng.uniform(0, 255, (1, 512, 512, 3)).astype(np.float32), device=dev)
    w = torch.tensor([[0., 0., 1., 1.]], device=dev)
    keys = ("rpn_rois", "detections", "mrcnn_mask")
    eager = {k: model.engine.infer(x, w)[k].cpu().numpy().copy() for k in keys}
    for rep in range(4):
        out = model.engine.infer_graphed(x, w)
        torch.cuda.synchronize()
        for k in keys:
            assert np.array_equal(out[k].cpu().numpy(), eager[k]), (k, rep)
    assert np.abs(eager["rpn_rois"]).sum() > 0


@pytest.mark.parametrize("how", ["graph", "tape"])
@pytest.mark.parametrize("head_dtype", [None, "float16"])
def test_graphed_training_steps_equal_eager(dev, head_dtype, how):
    """engine.step_taped (the step's launches re-issued from the host-side launch tape) and engine.step_graphed (forward + backward + optimiser of a step replayed from one HIP graph, three forked streams
    inside) against the same steps issued eagerly: four steps on two alternating batches, same losses every step and the
    same parameters at the end up to the float atomics (the two warm-up steps taken before the capture are rolled back:
    they must not count as optimiser steps).  ResNet-50 so that split-K, multi-problem and LDS-DMA kernels, the 16-bit
    blocks (float16 case) and every side stream are inside the graph."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg("resnet50", 128)
    w = _weights(cfg, 71, damp=0.5)
    batches = [_train_inputs(cfg, 2, 73), _train_inputs(cfg, 2, 75)]
    forced = None
    if head_dtype:
        # 16-bit rounding moves RPN scores in the 4th digit, which can re-order an NMS decision and hand the two trajectories
        # different ROI sets (the differences that forced rtol 0.15 here in round 2).  Both runs therefore take their proposals
        # from one fixed set per batch (engine.forced_rpn_rois: the float32 engine's proposals at the initial weights, refreshed
        # in place before every step -- the replayed launches read the same buffer), which leaves the differentiable part, the
        # streams and the optimiser under test at float32-like tolerances.
        m0 = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        forced = []
        for inputs, keys in batches:
            m0.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
            forced.append(m0.engine.last["rpn_rois"].clone())
        del m0
    out = {}
    for graphed in (False, True):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        eng = model.engine
        if head_dtype:
            eng.head_dtype = getattr(torch, head_dtype)
            eng.forced_rpn_rois = torch.empty_like(forced[0])
        model.compile(0.002, 0.9)
        losses = []
        for s in range(4):
            inputs, keys = batches[s % 2]
            di = model._to_device(inputs, keys)
            if forced is not None:
                eng.forced_rpn_rois.copy_(forced[s % 2])
            if graphed:
                ls = (eng.step_graphed if how == "graph" else eng.step_taped)(di, 0.002, 0.9)
            else:
                ls = eng.forward_backward(*di)
                eng.apply_gradients(0.002, 0.9, 1)
            losses.append(ls.cpu().numpy().copy())
        torch.cuda.synchronize()
        out[graphed] = (np.stack(losses), eng.params.cpu().numpy().copy(), eng.momentum.cpu().numpy().copy())
        if graphed:
            assert len(eng._train_graphs if how == "graph" else eng._train_tapes) == 1
    # float32: the first two steps see (almost) identical weights: tight; afterwards the atomics-order noise of the updates has
    # been through the network again.  16-bit: same ROI sets in both runs (see above).  What is left is the mode's own
    # run-to-run noise, measured with tools/replay_noise_probe.py (profiles/r03_replay_noise_f16.txt: eager against EAGER, and
    # against tape / graph) and tools/replay_noise_probe2.py (profiles/r03_gradient_noise_by_dtype.txt): two evaluations of one
    # float16 gradient on identical weights differ by 4-8e-5 in relative L2 (float32: 5e-8) -- the order noise of the float
    # atomics moves a few float32 sums across a float16 rounding boundary and every later rounding amplifies that; one update
    # later the gradients of two eager runs are 1-2 % apart, after four steps 2-4 %, the losses 2e-4 / 5e-4 / 8e-4 / 4e-3.
    # The bars are twice that floor (round 2: loss rtol 0.15, momentum 0.5 with free-running proposals); the exactness of the
    # replay itself is the float32 case's business.
    if head_dtype:
        np.testing.assert_allclose(out[True][0][:1], out[False][0][:1], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out[True][0][:3], out[False][0][:3], rtol=2e-3, atol=1e-5)
        np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-2, atol=1e-5)
        for k, t in ((1, 1e-5), (2, 6e-2)):
            assert np.linalg.norm(out[True][k] - out[False][k]) <= t * np.linalg.norm(out[False][k]), k
    else:
        tol = 2e-4
        np.testing.assert_allclose(out[True][0][:2], out[False][0][:2], rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(out[True][0], out[False][0], rtol=tol, atol=1e-5)
        for k, t in ((1, tol), (2, max(tol, 5e-3))):   # parameters; momentum buffer (= -lr * gradient history: atomics-order noise)
            scale = np.abs(out[False][k]).max()
            assert np.abs(out[True][k] - out[False][k]).max() <= t * scale, k
    assert np.abs(out[False][2]).max() > 0


def test_training_step_with_an_image_without_ground_truth(dev):
    """Edge case: one tile of the batch holds no object (all-zero GT rows; the reference's generator skips such images,
    model.py:1786, but DetectionTargetLayer defines the result: with no GT box every proposal is a negative and the
    negative quota is int(r * 0) - 0 = 0, so the image contributes no ROI at all).  The step must run, agree with the
    oracle on all five losses and leave finite gradients."""
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = _small_cfg("custom", 128)
    w = _weights(cfg, 11)
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    inputs, keys = _train_inputs(cfg, 2, 5)
    images, meta, rpn_match, rpn_bbox_t, gt_cls, gt_boxes, gt_masks = inputs
    gt_cls[1] = 0; gt_boxes[1] = 0; gt_masks[1] = False
    rpn_match[1] = 0
    neg = np.random.RandomState(1).choice(rpn_match.shape[1], cfg.RPN_TRAIN_ANCHORS_PER_IMAGE, replace=False)
    rpn_match[1, neg, 0] = -1                                  # negatives only, as build_rpn_targets gives without GT
    rpn_bbox_t[1] = 0
    losses = model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
    torch.cuda.synchronize()
    last = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in model.engine.last.items() if k != "pyramid"}
    counts = np.asarray(last["counts"])
    assert counts[0, 0] > 0 and counts[1].tolist() == [0, 0], counts
    assert bool(torch.isfinite(model.engine.grads).all())
    o = orc.OracleMaskRCNN(cfg, w)
    forced = {k: last[k] for k in ("rois", "target_class_ids", "target_bbox", "target_mask")}
    with torch.no_grad():
        ref = o.forward_training(images, rpn_match, rpn_bbox_t.astype(np.float32), gt_cls, gt_boxes, gt_masks,
                                 meta[:, 12:].astype(np.int32), orc.get_anchors(cfg, images.shape[1:]), keys, forced=forced)
    np.testing.assert_allclose(losses.cpu().numpy(), np.array([float(l) for l in ref["losses"]]), rtol=2e-3, atol=1e-5)


def test_overfit_one_batch_then_detect_it(dev):
    """End to end: 180 full steps (forward, backward, clipnorm, SGD-momentum; 120 at lr 0.002, 60 at 0.0004) on one
    synthetic batch drive the total loss down by > 10x, and the inference graph with the trained weights then finds the
    training objects (a detection with box IoU > 0.5 for at least a third of the ground-truth boxes).  The float atomics
    make the trajectory vary from run to run; at constant lr 0.002 the loss keeps jittering around 0.5 and the objects
    found ranged from 0 to 5 of 6 over repeated runs, with the final low-rate phase it settles near 0.35 and 4-5 of 6 are
    found every time (10 of 10 runs), so the bar sits well below that."""
    import bench
    from caesar_mrcnn_amd.config import run_py_config
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = run_py_config(num_classes=4, imgsize=256, backbone="resnet50", images_per_gpu=2, gpu_count=1)
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, seed=0)
    batch = bench.synthetic_batch(cfg, 2, seed=7)
    keys = np.random.RandomState(0).uniform(0, 1, (2, cfg.POST_NMS_ROIS_TRAINING))
    inp = model._to_device(batch, rand_keys=keys)
    eng = model.engine
    first = last = None
    for s in range(180):
        losses = eng.forward_backward(*inp)
        eng.apply_gradients(0.002 if s < 120 else 0.0004, cfg.LEARNING_MOMENTUM, 1)
        if s == 0:
            first = float(losses.sum())
    last = float(losses.sum())
    assert np.isfinite(last) and last < 0.1 * first, (first, last)
    icfg = run_py_config(num_classes=4, imgsize=256, backbone="resnet50", mode="inference")
    eng.cfg = icfg
    images, _, _, _, gt_cls, gt_boxes, _ = batch
    x = torch.tensor(images, device=dev)
    win = torch.tensor([[0., 0., 1., 1.]] * 2, device=dev)
    det = eng.infer(x, win)["detections"].cpu().numpy()          # [B, 100, (y1, x1, y2, x2, class, score)], normalised
    found = total = 0
    for b in range(2):
        d = det[b][det[b, :, 4] > 0]
        boxes = d[:, :4] * 255.0 + np.array([0, 0, 1, 1])
        for g in range(int((gt_cls[b] > 0).sum())):
            y1, x1, y2, x2 = gt_boxes[b, g]
            total += 1
            for k in range(d.shape[0]):
                iy = max(0.0, min(y2, boxes[k, 2]) - max(y1, boxes[k, 0])); ix = max(0.0, min(x2, boxes[k, 3]) - max(x1, boxes[k, 1]))
                inter = iy * ix
                union = (y2 - y1) * (x2 - x1) + (boxes[k, 2] - boxes[k, 0]) * (boxes[k, 3] - boxes[k, 1]) - inter
                if inter / union > 0.5:
                    found += 1
                    break
    assert total > 0 and 3 * found >= total, (found, total)
