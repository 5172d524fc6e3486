#!/usr/bin/env python3
"""Generates tests/golden/reference_numpy_helpers.npz by IMPORTING the reference's own pure-NumPy
functions (mrcnn/utils.py, mrcnn/model.py, mrcnn/config.py under /root/reference) and recording their
outputs on seeded inputs.  Runs only in the build container (the reference never travels to the GPU
box); the resulting .npz is committed and is what pins the oracle.

TensorFlow / Keras / skimage / astropy are not installed, so inert stand-in modules are registered
before the import: nothing of them is ever called by the functions used here.
Only inputs and outputs are stored -- no reference source text.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_numpy_helpers.npz")


class _Any(object):
    """Inert stand-in: any attribute / call / subclassing works, nothing computes."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Any          # a class, so `class X(KL.BatchNormalization)` / `class Y(KE.Layer)` still parse


def _stub(name, **attrs):
    m = _StubModule(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def install_stubs():
    _stub("tensorflow", __version__="1.13.2")
    _stub("keras", __version__="2.2.4")
    for sub in ("backend", "layers", "engine", "models", "optimizers", "callbacks", "regularizers", "utils"):
        _stub("keras." + sub)
    _stub("skimage", __version__="0.15.0")
    for sub in ("color", "io", "transform", "measure"):
        _stub("skimage." + sub)
    _stub("astropy")
    for sub in ("io", "io.fits", "io.ascii", "units", "modeling", "modeling.parameters", "modeling.core", "wcs",
                "visualization", "stats"):
        _stub("astropy." + sub)
    for name in ("cv2", "imgaug", "imgaug.augmenters", "h5py"):
        _stub(name)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from mrcnn import utils as U
    from mrcnn import model as M
    from mrcnn.config import Config

    class Cfg(Config):          # the run.py effective values that matter for the helpers
        NAME = "fixture"
        NUM_CLASSES = 4
        IMAGE_MIN_DIM = 256
        IMAGE_MAX_DIM = 256
        RPN_ANCHOR_SCALES = (4, 8, 16, 32, 64)
        RPN_TRAIN_ANCHORS_PER_IMAGE = 512
        MAX_GT_INSTANCES = 300

    cfg = Cfg()
    rng = np.random.RandomState(20241004)
    out = {}

    # Config derived fields
    out["cfg_batch_size"] = np.array(cfg.BATCH_SIZE)
    out["cfg_image_shape"] = np.array(cfg.IMAGE_SHAPE)
    out["cfg_image_meta_size"] = np.array(cfg.IMAGE_META_SIZE)

    # anchors
    for size in (256, 512):
        shapes = M.compute_backbone_shapes(cfg, (size, size, 3))
        a = U.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, shapes, cfg.BACKBONE_STRIDES,
                                       cfg.RPN_ANCHOR_STRIDE)
        out["backbone_shapes_%d" % size] = shapes
        if size == 256:
            out["anchors_px_256"] = a
            out["anchors_norm_256"] = U.norm_boxes(a, (size, size))
        else:
            out["anchors_px_512_head"] = a[:64]
            out["anchors_px_512_tail"] = a[-64:]
            out["anchors_px_512_sum"] = np.array([a.shape[0], a.sum(), np.abs(a).sum()])
    # boxes
    b = rng.uniform(0, 200, (50, 4))
    boxes = np.stack([np.minimum(b[:, 0], b[:, 2]), np.minimum(b[:, 1], b[:, 3]),
                      np.maximum(b[:, 0], b[:, 2]) + 1, np.maximum(b[:, 1], b[:, 3]) + 1], 1)
    out["boxes_in"] = boxes
    out["norm_boxes"] = U.norm_boxes(boxes, (256, 256))
    out["denorm_boxes"] = U.denorm_boxes(U.norm_boxes(boxes, (256, 256)), (256, 256))
    deltas = rng.normal(0, 0.3, (50, 4)).astype(np.float32)
    out["deltas_in"] = deltas
    out["apply_box_deltas"] = U.apply_box_deltas(boxes.astype(np.float32), deltas)
    gt = boxes[rng.permutation(50)] + rng.uniform(-3, 3, (50, 4))
    gt[:, 2:] = np.maximum(gt[:, 2:], gt[:, :2] + 1)
    out["gt_in"] = gt
    out["box_refinement"] = U.box_refinement(boxes, gt)
    out["compute_overlaps"] = U.compute_overlaps(boxes, gt[:17])
    # NMS incl. duplicated boxes (ties) and an exact-threshold pair
    nb = np.concatenate([boxes, boxes[:10] + 0.5, boxes[:5]], 0).astype(np.float32)
    ns = rng.uniform(0, 1, nb.shape[0]).astype(np.float32)
    out["nms_boxes"] = nb
    out["nms_scores"] = ns
    for thr in (0.3, 0.5, 0.7):
        out["nms_keep_%d" % int(thr * 10)] = U.non_max_suppression(nb, ns, thr)
    tz = np.array([[0, 0, 0, 0], [1, 2, 3, 4], [0, 0, 0, 0], [5, 6, 7, 8]])
    out["trim_zeros"] = U.trim_zeros(tz)
    # masks -> boxes
    masks = np.zeros((64, 64, 5), bool)
    masks[10:20, 5:9, 0] = True
    masks[0:3, 60:64, 1] = True
    masks[33, 33, 2] = True
    masks[40:64, 0:64, 4] = True
    out["masks_in"] = masks
    out["extract_bboxes"] = U.extract_bboxes(masks)
    # RPN targets (NumPy RNG injected through the module-level np.random seed)
    anchors = out["anchors_px_256"]
    gt_boxes = np.array([[30, 40, 62, 75], [100, 100, 108, 109], [180, 20, 240, 90], [5, 200, 9, 206]], np.int32)
    gt_ids = np.array([1, 2, 3, 1], np.int32)
    np.random.seed(7)
    rm, rb = M.build_rpn_targets((256, 256, 3), anchors, gt_ids, gt_boxes, cfg)
    out["rpn_gt_boxes"] = gt_boxes
    out["rpn_gt_ids"] = gt_ids
    out["rpn_match"] = rm
    out["rpn_bbox"] = rb
    gt_ids_c = np.array([1, -1, 3, 1], np.int32)      # one crowd box
    np.random.seed(11)
    rm2, rb2 = M.build_rpn_targets((256, 256, 3), anchors, gt_ids_c, gt_boxes, cfg)
    out["rpn_gt_ids_crowd"] = gt_ids_c
    out["rpn_match_crowd"] = rm2
    out["rpn_bbox_crowd"] = rb2
    # image meta / molding
    meta = M.compose_image_meta(7, (132, 132, 3), (256, 256, 3), (62, 62, 194, 194), 1.0, np.array([1, 1, 0, 1]))
    out["image_meta"] = meta
    pm = M.parse_image_meta(meta[None])
    out["parse_window"] = pm["window"]
    out["parse_active"] = pm["active_class_ids"]
    img = rng.randint(0, 255, (132, 100, 3)).astype(np.uint8)
    out["resize_in"] = img

    class C2(Cfg):
        MEAN_PIXEL = np.array([1.5, 2.5, 3.5])
    out["mold_image"] = M.mold_image(img, C2())
    # resize_image without scaling (scale == 1: no skimage call): square and pad64 modes
    big = rng.randint(0, 255, (256, 200, 3)).astype(np.uint8)
    out["resize_big_in"] = big
    r = U.resize_image(big, min_dim=256, max_dim=256, min_scale=0, mode="square")
    out["resize_square_img"], out["resize_square_window"] = r[0], np.array(r[1])
    out["resize_square_scale"], out["resize_square_padding"] = np.array(r[2]), np.array(r[3])
    r = U.resize_image(big, min_dim=128, max_dim=None, min_scale=0, mode="pad64")
    out["resize_pad64_img"], out["resize_pad64_window"] = r[0], np.array(r[1])
    # evaluation helpers
    pb = boxes[:12].astype(np.int32)
    out["recall"] = np.array(U.compute_recall(pb, gt[:9].astype(np.int32), 0.5)[0])
    # tile grids (pure Python in the reference)
    out["tiles_a"] = np.array(U.generate_tiles(0, 999, 0, 799, 256, 256, 0.5, 1.0))
    out["tiles_b"] = np.array(U.generate_tiles(10, 521, 5, 300, 128, 100, 1.0, 0.75))
    # ---- round 2 additions ---------------------------------------------------------------------------------
    import hashlib
    # 1024x1024 anchors (configs[3]): too large to store -- SHA-256 of the float64 bytes + head / tail / sums
    shapes = M.compute_backbone_shapes(cfg, (1024, 1024, 3))
    a = U.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, shapes, cfg.BACKBONE_STRIDES,
                                   cfg.RPN_ANCHOR_STRIDE)
    out["backbone_shapes_1024"] = shapes
    out["anchors_px_1024_head"], out["anchors_px_1024_tail"] = a[:64], a[-64:]
    out["anchors_px_1024_sum"] = np.array([a.shape[0], a.sum(), np.abs(a).sum()])
    out["anchors_px_1024_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
    an = U.norm_boxes(a, (1024, 1024))
    out["anchors_norm_1024_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(an).tobytes()).digest(), np.uint8)
    # NMS with pairs whose IoU equals the threshold exactly (0.5 and 0.25 are exact in float32): "iou > threshold"
    # keeps both boxes at equality, suppresses just above it
    eb = np.array([[0, 0, 10, 10], [0, 0, 10, 5],            # IoU 50/100 = 0.5
                   [20, 20, 30, 30], [20, 20, 25, 25],       # IoU 25/100 = 0.25
                   [40, 40, 50, 50], [40, 40, 50, 46],       # IoU 0.6
                   [60, 60, 70, 70], [60, 60, 70, 70]], np.float32)   # duplicate
    es = np.array([0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2], np.float32)
    out["nms_eq_boxes"], out["nms_eq_scores"] = eb, es
    for thr, tag in ((0.5, "50"), (0.25, "25"), (0.6, "60")):
        out["nms_eq_keep_" + tag] = U.non_max_suppression(eb, es, thr)
    # evaluation: compute_matches / compute_ap (mask IoU matching, VOC-style AP)
    H = 48
    gtm = np.zeros((H, H, 6), bool)
    gtb = np.zeros((6, 4), np.int32)
    gtc = np.array([1, 2, 3, 1, 2, 0], np.int32)
    for g in range(5):                                       # 6th GT row stays zero padding
        y, x = rng.randint(0, H - 14, 2); h, w_ = rng.randint(5, 14, 2)
        gtm[y:y + h, x:x + w_, g] = True
        gtb[g] = [y, x, y + h, x + w_]
    n_pred = 9
    pm_ = np.zeros((H, H, n_pred), bool)
    pb_ = np.zeros((n_pred, 4), np.int32)
    pc_ = np.zeros(n_pred, np.int32)
    for k in range(n_pred):
        g = k % 5
        dy, dx = rng.randint(-3, 4, 2)
        y1, x1, y2, x2 = gtb[g]
        y1, x1 = max(0, y1 + dy), max(0, x1 + dx)
        y2, x2 = min(H, y2 + dy), min(H, x2 + dx)
        pm_[y1:y2, x1:x2, k] = True
        pb_[k] = [y1, x1, y2, x2]
        pc_[k] = gtc[g] if k != 3 else (gtc[g] % 3) + 1     # one wrong class
    ps_ = rng.uniform(0.05, 1, n_pred).astype(np.float32)
    out["ev_gt_boxes"], out["ev_gt_ids"], out["ev_gt_masks"] = gtb, gtc, gtm
    out["ev_pred_boxes"], out["ev_pred_ids"], out["ev_pred_scores"], out["ev_pred_masks"] = pb_, pc_, ps_, pm_
    out["ev_overlaps_masks"] = U.compute_overlaps_masks(pm_, gtm[..., :5])
    for thr, tag in ((0.5, "50"), (0.75, "75")):
        gm, pmatch, ov = U.compute_matches(gtb, gtc, gtm, pb_, pc_, ps_, pm_, iou_threshold=thr)
        out["ev_gt_match_" + tag], out["ev_pred_match_" + tag], out["ev_overlaps_" + tag] = gm, pmatch, ov
        ap, prec, rec, _ = U.compute_ap(gtb, gtc, gtm, pb_, pc_, ps_, pm_, iou_threshold=thr)
        out["ev_ap_" + tag], out["ev_precisions_" + tag], out["ev_recalls_" + tag] = np.array(ap), prec, rec
    # mrcnn/graph.py: connected components in the reference's depth-first order
    from mrcnn.graph import Graph
    for tag, nv, ne in (("a", 12, 9), ("b", 30, 40), ("c", 7, 0)):
        edges = np.array([rng.choice(nv, 2, replace=False) for _ in range(ne)], np.int64).reshape(ne, 2)
        gph = Graph(nv)
        for v, w_ in edges:
            gph.addEdge(int(v), int(w_))
        cc = gph.connectedComponents()
        out["graph_%s_n" % tag] = np.array(nv)
        out["graph_%s_edges" % tag] = edges
        out["graph_%s_cc_flat" % tag] = np.array([v for c in cc for v in c], np.int64)
        out["graph_%s_cc_len" % tag] = np.array([len(c) for c in cc], np.int64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
