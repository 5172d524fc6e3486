"""Pins the CPU oracle (and the product's host-side twins in caesar_mrcnn_amd.utils / datagen) against
golden vectors produced by the reference's own NumPy functions (tests/golden/make_reference_fixtures.py,
run in the build container where /root/reference is mounted).  CPU only."""
import os

import numpy as np
import pytest

import mrcnn_oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_numpy_helpers.npz"))


def _cfg():
    from caesar_mrcnn_amd.config import run_py_config
    return run_py_config()


def _impls():
    from caesar_mrcnn_amd import utils as U
    return [("oracle", orc), ("product", U)]


def test_config_derived_fields():
    from caesar_mrcnn_amd.config import Config

    class C(Config):
        NUM_CLASSES = 4
        IMAGE_MIN_DIM = 256
        IMAGE_MAX_DIM = 256
    c = C()
    assert c.BATCH_SIZE == int(G["cfg_batch_size"])
    assert np.array_equal(c.IMAGE_SHAPE, G["cfg_image_shape"])
    assert c.IMAGE_META_SIZE == int(G["cfg_image_meta_size"])
    c.IMAGES_PER_GPU, c.GPU_COUNT = 4, 8           # run.py-style late override stays consistent (App. D-1)
    assert c.BATCH_SIZE == 32


@pytest.mark.parametrize("name,mod", _impls())
def test_anchors(name, mod):
    cfg = _cfg()
    for size in (256, 512):
        if name == "oracle":
            shapes = mod.compute_backbone_shapes(cfg.BACKBONE_STRIDES, (size, size, 3))
        else:
            shapes = mod.compute_backbone_shapes(cfg, (size, size, 3))
        assert np.array_equal(shapes, G["backbone_shapes_%d" % size])
        a = mod.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, shapes, cfg.BACKBONE_STRIDES,
                                         cfg.RPN_ANCHOR_STRIDE)
        if size == 256:
            assert a.dtype == G["anchors_px_256"].dtype and np.array_equal(a, G["anchors_px_256"])
            n = mod.norm_boxes(a, (size, size))
            assert n.dtype == np.float32 and np.array_equal(n, G["anchors_norm_256"])
            assert np.array_equal(mod.get_anchors(cfg, (256, 256, 3)), G["anchors_norm_256"])
        else:
            assert np.array_equal(a[:64], G["anchors_px_512_head"]) and np.array_equal(a[-64:], G["anchors_px_512_tail"])
            assert np.array_equal(np.array([a.shape[0], a.sum(), np.abs(a).sum()]), G["anchors_px_512_sum"])


@pytest.mark.parametrize("name,mod", _impls())
def test_box_helpers(name, mod):
    boxes = G["boxes_in"]
    assert np.array_equal(mod.norm_boxes(boxes, (256, 256)), G["norm_boxes"])
    assert np.array_equal(mod.denorm_boxes(G["norm_boxes"], (256, 256)), G["denorm_boxes"])
    assert np.array_equal(mod.compute_overlaps(boxes, G["gt_in"][:17]), G["compute_overlaps"])
    assert np.array_equal(mod.trim_zeros(np.array([[0, 0, 0, 0], [1, 2, 3, 4], [0, 0, 0, 0], [5, 6, 7, 8]])), G["trim_zeros"])
    assert np.array_equal(mod.extract_bboxes(G["masks_in"]), G["extract_bboxes"])


def test_oracle_box_arithmetic():
    boxes = G["boxes_in"]
    got = orc.apply_box_deltas_np(boxes.astype(np.float32), G["deltas_in"])
    assert got.dtype == np.float32 and np.array_equal(got, G["apply_box_deltas"])
    assert np.array_equal(orc.box_refinement_np(boxes, G["gt_in"]), G["box_refinement"])


def test_oracle_nms_matches_reference_numpy_nms():
    nb, ns = G["nms_boxes"], G["nms_scores"]
    for thr in (0.3, 0.5, 0.7):
        ref = G["nms_keep_%d" % int(thr * 10)]
        assert np.array_equal(orc.non_max_suppression_np(nb, ns, thr), ref)
        # the restated TF kernel agrees with the reference's NumPy NMS wherever both are defined
        # (positive-area boxes, distinct scores): same greedy rule, same IoU > thr test
        tf_keep = orc.tf_non_max_suppression(nb, ns, nb.shape[0], thr)
        assert np.array_equal(tf_keep.astype(np.int32), ref)


def test_rpn_targets():
    from caesar_mrcnn_amd.datagen import build_rpn_targets as product_rpn
    cfg = _cfg()
    anchors = G["anchors_px_256"]
    for ids_key, m_key, b_key, seed in (("rpn_gt_ids", "rpn_match", "rpn_bbox", 7),
                                        ("rpn_gt_ids_crowd", "rpn_match_crowd", "rpn_bbox_crowd", 11)):
        np.random.seed(seed)
        rm, rb = orc.build_rpn_targets(anchors, G[ids_key], G["rpn_gt_boxes"], cfg.RPN_TRAIN_ANCHORS_PER_IMAGE,
                                       cfg.RPN_BBOX_STD_DEV)
        assert np.array_equal(rm, G[m_key]) and np.array_equal(rb, G[b_key])
        np.random.seed(seed)
        rm, rb = product_rpn((256, 256, 3), anchors, G[ids_key], G["rpn_gt_boxes"], cfg)
        assert np.array_equal(rm, G[m_key]) and np.array_equal(rb, G[b_key])
    assert (G["rpn_match"] == 1).sum() > 0 and (G["rpn_match"] == -1).sum() > 0


def test_image_meta_and_molding():
    from caesar_mrcnn_amd import utils as U
    from caesar_mrcnn_amd.config import Config
    args = (7, (132, 132, 3), (256, 256, 3), (62, 62, 194, 194), 1.0, np.array([1, 1, 0, 1]))
    for f in (orc.compose_image_meta, U.compose_image_meta):
        assert np.array_equal(f(*args), G["image_meta"])
    pm = U.parse_image_meta(G["image_meta"][None])
    assert np.array_equal(pm["window"], G["parse_window"]) and np.array_equal(pm["active_class_ids"], G["parse_active"])

    class C2(Config):
        MEAN_PIXEL = np.array([1.5, 2.5, 3.5])
    got = U.mold_image(G["resize_in"], C2())
    assert got.dtype == G["mold_image"].dtype and np.array_equal(got, G["mold_image"])


def test_resize_image_without_scaling():
    from caesar_mrcnn_amd import utils as U
    img, window, scale, padding, crop = U.resize_image(G["resize_big_in"], min_dim=256, max_dim=256, min_scale=0,
                                                       mode="square")
    assert img.dtype == np.uint8 and np.array_equal(img, G["resize_square_img"])
    assert np.array_equal(np.array(window), G["resize_square_window"])
    assert scale == G["resize_square_scale"] and np.array_equal(np.array(padding), G["resize_square_padding"])
    img, window, _, _, _ = U.resize_image(G["resize_big_in"], min_dim=128, max_dim=None, min_scale=0, mode="pad64")
    assert np.array_equal(img, G["resize_pad64_img"]) and np.array_equal(np.array(window), G["resize_pad64_window"])
    with pytest.raises(Exception):
        U.resize_image(G["resize_big_in"], min_dim=128, max_dim=128, mode="bogus")


def test_recall():
    from caesar_mrcnn_amd import utils as U
    pb = G["boxes_in"][:12].astype(np.int32)
    assert U.compute_recall(pb, G["gt_in"][:9].astype(np.int32), 0.5)[0] == float(G["recall"])


def test_bilinear_resize_restatement_properties():
    """skimage.transform.resize is not installed: the restatement (parity unpinned) is checked on
    properties of the half-pixel-centre bilinear warp."""
    from caesar_mrcnn_amd import utils as U
    x = np.arange(12, dtype=np.float64).reshape(3, 4)
    assert np.allclose(U.resize(x, (3, 4)), x)                       # identity
    const = np.full((5, 7), 3.25)
    up = U.resize(const, (13, 20))
    assert np.allclose(up[2:-2, 2:-2], 3.25) and up.max() <= 3.25 + 1e-12     # interior exact, clip holds
    ramp = np.tile(np.arange(8, dtype=np.float64), (8, 1))
    r2 = U.resize(ramp, (16, 16))
    assert np.all(np.diff(r2[8, 1:-1]) >= -1e-12)                   # monotone along the ramp
    m = np.zeros((28, 28)); m[8:20, 8:20] = 1.0
    full = U.unmold_mask(m, (10, 20, 66, 76), (100, 100, 3))
    assert full.dtype == bool and full[:10].sum() == 0 and full[10:66, 20:76].sum() > 0
    assert U.resize(np.ones((4, 4), bool), (8, 8)).max() <= 1.0      # bool input is scaled like img_as_float


# ---- round 2: the remaining reference-held pins ------------------------------------------------------
@pytest.mark.parametrize("name,mod", _impls())
def test_anchors_1024(name, mod):
    """configs[3] (1024x1024, 261 888 anchors): SHA-256 of the reference's float64 array and of its normalised
    float32 twin, plus head / tail rows and sums."""
    import hashlib
    cfg = _cfg()
    shapes = (mod.compute_backbone_shapes(cfg.BACKBONE_STRIDES, (1024, 1024, 3)) if name == "oracle"
              else mod.compute_backbone_shapes(cfg, (1024, 1024, 3)))
    assert np.array_equal(shapes, G["backbone_shapes_1024"])
    a = mod.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, shapes, cfg.BACKBONE_STRIDES,
                                     cfg.RPN_ANCHOR_STRIDE)
    assert a.shape == (261888, 4) and a.dtype == np.float64
    assert np.array_equal(a[:64], G["anchors_px_1024_head"]) and np.array_equal(a[-64:], G["anchors_px_1024_tail"])
    assert np.array_equal(np.array([a.shape[0], a.sum(), np.abs(a).sum()]), G["anchors_px_1024_sum"])
    sha = lambda x: np.frombuffer(hashlib.sha256(np.ascontiguousarray(x).tobytes()).digest(), np.uint8)
    assert np.array_equal(sha(a), G["anchors_px_1024_sha256"])
    assert np.array_equal(sha(mod.norm_boxes(a, (1024, 1024))), G["anchors_norm_1024_sha256"])
    assert np.array_equal(sha(mod.get_anchors(cfg_1024(), (1024, 1024, 3))), G["anchors_norm_1024_sha256"])


def cfg_1024():
    from caesar_mrcnn_amd.config import run_py_config
    return run_py_config(imgsize=1024)


def test_nms_at_exact_threshold_pairs():
    """Pairs whose IoU equals the threshold exactly (0.5, 0.25: exact in float32) are BOTH kept -- the rule is
    `iou > threshold` in the reference's NumPy NMS (utils.py:215) and in the restated TF kernel alike."""
    eb, es = G["nms_eq_boxes"], G["nms_eq_scores"]
    for thr, tag in ((0.5, "50"), (0.25, "25"), (0.6, "60")):
        ref = G["nms_eq_keep_" + tag]
        assert np.array_equal(orc.non_max_suppression_np(eb, es, thr), ref)
        assert np.array_equal(orc.tf_non_max_suppression(eb, es, eb.shape[0], thr).astype(np.int32), ref)
    assert 1 in G["nms_eq_keep_50"] and 3 in G["nms_eq_keep_25"] and 3 not in G["nms_eq_keep_50"][:0]
    assert 1 not in G["nms_eq_keep_25"]           # IoU 0.5 > 0.25: suppressed there


def test_compute_matches_and_ap():
    from caesar_mrcnn_amd import utils as U
    args = (G["ev_gt_boxes"], G["ev_gt_ids"], G["ev_gt_masks"], G["ev_pred_boxes"], G["ev_pred_ids"],
            G["ev_pred_scores"], G["ev_pred_masks"])
    assert np.array_equal(U.compute_overlaps_masks(G["ev_pred_masks"], G["ev_gt_masks"][..., :5]), G["ev_overlaps_masks"])
    for thr, tag in ((0.5, "50"), (0.75, "75")):
        gm, pm, ov = U.compute_matches(*args, iou_threshold=thr)
        assert gm.dtype == G["ev_gt_match_" + tag].dtype
        assert np.array_equal(gm, G["ev_gt_match_" + tag]) and np.array_equal(pm, G["ev_pred_match_" + tag])
        assert np.array_equal(ov, G["ev_overlaps_" + tag])
        ap, prec, rec, _ = U.compute_ap(*args, iou_threshold=thr)
        assert ap == float(G["ev_ap_" + tag])
        assert np.array_equal(prec, G["ev_precisions_" + tag]) and np.array_equal(rec, G["ev_recalls_" + tag])
    assert (G["ev_pred_match_50"] > -1).sum() >= 3 and float(G["ev_ap_50"]) > 0.5


def test_connected_components_order():
    """analyze.connected_components reproduces mrcnn/graph.py's Graph.connectedComponents: component order, and
    the depth-first vertex order inside each component."""
    from caesar_mrcnn_amd.analyze import connected_components
    for tag in "abc":
        n, edges = int(G["graph_%s_n" % tag]), [tuple(int(v) for v in e) for e in G["graph_%s_edges" % tag]]
        cc = connected_components(n, edges)
        assert [len(c) for c in cc] == G["graph_%s_cc_len" % tag].tolist()
        assert [v for c in cc for v in c] == G["graph_%s_cc_flat" % tag].tolist()


def test_resize_restatement_matches_scipy_map_coordinates():
    """utils.resize restates skimage.transform.resize(order=1, mode='constant', cval=0, clip=True) [3P; skimage is not
    installable here].  Second source for its arithmetic: scipy.ndimage.map_coordinates(order=1, mode="grid-constant", cval=0: samples between the edge pixel and the outside blend with the constant, as skimage's warp does) at
    the half-pixel-centre coordinates (o + 0.5) * in / out - 0.5 -- an independent bilinear sampler with the same
    zero-outside rule -- followed by the clip to the input's range.  Up- and down-sampling, masks (28 x 28 -> box) and RGB."""
    from scipy import ndimage
    from caesar_mrcnn_amd import utils
    rng = np.random.default_rng(21)
    for (h, w), (oh, ow) in (((28, 28), (61, 17)), ((28, 28), (5, 90)), ((28, 28), (28, 28)), ((13, 31), (64, 64)), ((132, 132), (256, 256))):
        img = rng.random((h, w))
        ry = (np.arange(oh) + 0.5) * (h / oh) - 0.5
        rx = (np.arange(ow) + 0.5) * (w / ow) - 0.5
        coords = np.stack(np.meshgrid(ry, rx, indexing="ij"))
        want = np.clip(ndimage.map_coordinates(img, coords, order=1, mode="grid-constant", cval=0.0), img.min(), img.max())
        got = utils.resize(img, (oh, ow))
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    rgb = rng.integers(0, 256, (20, 33, 3), dtype=np.uint8)
    got = utils.resize(rgb, (47, 50), preserve_range=True)
    ry = (np.arange(47) + 0.5) * (20 / 47) - 0.5
    rx = (np.arange(50) + 0.5) * (33 / 50) - 0.5
    coords = np.stack(np.meshgrid(ry, rx, indexing="ij"))
    for c in range(3):
        want = np.clip(ndimage.map_coordinates(rgb[:, :, c].astype(np.float64), coords, order=1, mode="grid-constant", cval=0.0), rgb.min(), rgb.max())
        np.testing.assert_allclose(got[:, :, c], want, rtol=0, atol=1e-10)


def test_oracle_crop_and_resize_matches_scipy_sampler():
    """Second source for the ARITHMETIC of the oracle's tf.image.crop_and_resize restatement (TF itself cannot run here: the
    sampling grid in = lo (D - 1) + i (hi - lo) (D - 1) / (crop - 1) and the zero-outside rule are SURVEY App. C-3): scipy's
    independent bilinear sampler scipy.ndimage.map_coordinates(order=1, mode='constant', cval=0) evaluated on that grid -- a
    sample outside [0, D - 1] is the constant, nothing is blended across the edge -- per channel, for boxes inside, partly
    outside and wholly outside the map, and for the crop-size-1 rule.  float32 tolerance."""
    import torch
    from scipy import ndimage
    import mrcnn_oracle as orc
    rng = np.random.default_rng(33)
    H, W, C = 19, 23, 5
    img = rng.standard_normal((2, H, W, C)).astype(np.float32)
    boxes = np.array([[0.1, 0.2, 0.7, 0.9], [0.0, 0.0, 1.0, 1.0], [-0.2, 0.3, 0.5, 1.3], [0.4, 0.4, 0.41, 0.41], [1.2, 0.1, 1.6, 0.5],
                      [0.3, 0.6, 0.9, 0.2]], np.float32)           # the last one: x2 < x1 (a flipped crop, as TF allows)
    idx = np.array([0, 1, 1, 0, 0, 1])
    for ch, cw in ((7, 7), (14, 14), (1, 3), (28, 28)):
        got = orc.crop_and_resize(torch.tensor(img), boxes, idx, ch, cw).numpy()
        for n, (b, bi) in enumerate(zip(boxes.astype(np.float64), idx)):
            ys = b[0] * (H - 1) + np.arange(ch) * (b[2] - b[0]) * (H - 1) / (ch - 1) if ch > 1 else np.array([0.5 * (b[0] + b[2]) * (H - 1)])
            xs = b[1] * (W - 1) + np.arange(cw) * (b[3] - b[1]) * (W - 1) / (cw - 1) if cw > 1 else np.array([0.5 * (b[1] + b[3]) * (W - 1)])
            coords = np.stack(np.meshgrid(ys, xs, indexing="ij"))
            for c in range(C):
                want = ndimage.map_coordinates(img[bi, :, :, c].astype(np.float64), coords, order=1, mode="constant", cval=0.0)
                np.testing.assert_allclose(got[n, :, :, c], want, rtol=0, atol=2e-5, err_msg="box %d crop %dx%d" % (n, ch, cw))


def test_oracle_sgd_clipnorm_matches_torch_optim():
    """keras.optimizers.SGD(lr, momentum, clipnorm) (mrcnn/model.py:2260-2262) is restated in oracle.sgd_step [3P, Keras 2.2.4 not
    installable here].  Second source for its arithmetic: torch.nn.utils.clip_grad_norm_ (one norm over all tensors, every tensor
    scaled by clipnorm / norm) + torch.optim.SGD(momentum) -- Keras keeps v = m v - lr g, torch buf = m buf + g; with a constant
    learning rate v = -lr buf, the same trajectory.  Five steps, two with the clip active, one with the norm below the threshold."""
    import torch
    rng = np.random.default_rng(11)
    shapes = {"a/kernel": (3, 3, 8, 16), "a/bias": (16,), "b/gamma": (16,), "c/kernel": (64, 5)}
    p0 = {k: rng.standard_normal(s).astype(np.float32) for k, s in shapes.items()}
    params = {k: v.copy() for k, v in p0.items()}
    vel = {k: np.zeros_like(v) for k, v in p0.items()}
    tp = {k: torch.nn.Parameter(torch.from_numpy(v.copy())) for k, v in p0.items()}
    opt = torch.optim.SGD(list(tp.values()), lr=0.01, momentum=0.9)
    for step, gscale in enumerate((3.0, 0.02, 1.5, 0.05, 4.0)):        # 1 504 elements: norm ~ 38.8 gscale
        grads = {k: (gscale * rng.standard_normal(s)).astype(np.float32) for k, s in shapes.items()}
        norm = orc.sgd_step(params, grads, vel, 0.01, 0.9, 5.0)
        for k in tp:
            tp[k].grad = torch.from_numpy(grads[k].copy())
        tnorm = float(torch.nn.utils.clip_grad_norm_(list(tp.values()), 5.0))
        opt.step()
        assert abs(norm - tnorm) <= 1e-5 * tnorm
        assert (norm >= 5.0) == (gscale >= 1.0)                              # steps 0, 2, 4 clip; 1, 3 (norm ~ 0.8, 1.9) do not
        for k in tp:
            np.testing.assert_allclose(params[k], tp[k].detach().numpy(), rtol=2e-5, atol=2e-6, err_msg="%s step %d" % (k, step))


def test_oracle_loss_primitives_match_torch_functional():
    """smooth_l1 (mrcnn/model.py:1098-1105) and K.binary_crossentropy (:1259, Keras 2.2.4's clip to [1e-7, 1 - 1e-7] -> logit ->
    sigmoid cross entropy [3P]) against torch.nn.functional's own statements of the same losses."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(4096, generator=g) * 2, torch.randn(4096, generator=g) * 2
    b[:8] = a[:8] + torch.tensor([1.0, -1.0, 0.0, 0.999999, 1.000001, -0.5, 2.0, -2.0])      # both branches and the |d| = 1 seam
    np.testing.assert_allclose(orc.smooth_l1(a, b).numpy(), F.smooth_l1_loss(b, a, reduction="none", beta=1.0).numpy(), rtol=1e-6, atol=1e-7)
    t = (torch.rand(4096, generator=g) > 0.6).float()
    o = torch.rand(4096, generator=g).double()
    o[:4] = torch.tensor([0.0, 1.0, 1e-9, 1 - 1e-9], dtype=torch.float64)    # Keras clips these to eps / 1 - eps before the logarithm
    want = F.binary_cross_entropy(torch.clamp(o, 1e-7, 1 - 1e-7), t.double(), reduction="none")
    np.testing.assert_allclose(orc.keras_binary_crossentropy(t.double(), o).numpy(), want.numpy(), rtol=1e-9, atol=1e-12)
    o32 = o.float()
    got32 = orc.keras_binary_crossentropy(t, o32).numpy()
    np.testing.assert_allclose(got32[4:], want.numpy()[4:], rtol=2e-5, atol=1e-6)


def test_oracle_same_padding_conv_and_pool_match_scipy():
    """Conv2D / MaxPooling2D padding='same' (mrcnn/model.py:99-210) are TF kernels [3P].  The padding rule is pinned by the worked
    example of TensorFlow's own documentation of SAME (width 13, filter 6, stride 5 -> 3 outputs, 1 column before, 2 after) and by
    the backbone's own cases; the arithmetic by scipy: correlate with explicit zero padding, maximum_filter with -inf padding,
    sampled at the window positions the rule gives."""
    import torch
    from scipy import ndimage, signal
    assert orc.same_pad(13, 6, 5) == (3, 1, 2)
    assert orc.same_pad(64, 3, 2) == (32, 0, 1) and orc.same_pad(63, 3, 2) == (32, 1, 1) and orc.same_pad(14, 3, 1) == (14, 1, 1)
    rng = np.random.default_rng(3)
    for H, W, k, s in ((14, 14, 3, 1), (16, 12, 3, 2), (15, 9, 3, 2), (8, 8, 1, 2)):
        x = rng.standard_normal((H, W)).astype(np.float64)
        w = rng.standard_normal((k, k)).astype(np.float64)
        got = orc.conv2d_nhwc(torch.from_numpy(x)[None, :, :, None], torch.from_numpy(w)[:, :, None, None], None, s, "same")[0, :, :, 0].numpy()
        oh, pt, pb = orc.same_pad(H, k, s)
        ow, pl, pr = orc.same_pad(W, k, s)
        full = signal.correlate2d(np.pad(x, ((pt, pb), (pl, pr))), w, mode="valid")
        np.testing.assert_allclose(got, full[::s, ::s][:oh, :ow], rtol=1e-12, atol=1e-12)
    for H, W in ((16, 12), (15, 9)):
        x = rng.standard_normal((H, W)).astype(np.float32)
        got = orc.maxpool3x3s2_same(torch.from_numpy(x)[None, :, :, None])[0, :, :, 0].numpy()
        oh, pt, _ = orc.same_pad(H, 3, 2)
        ow, pl, _ = orc.same_pad(W, 3, 2)
        filt = ndimage.maximum_filter(x, size=3, mode="constant", cval=-np.inf)          # window centred on every pixel
        want = filt[1 - pt::2, 1 - pl::2][:oh, :ow]                                       # window i starts at 2 i - pad_before: centre 2 i + 1 - pad_before
        np.testing.assert_array_equal(got, want)


def test_oracle_transposed_convolution_layout():
    """KL.Conv2DTranspose(256, (2, 2), strides=2) (mrcnn/model.py:1084): Keras stores the kernel as (kh, kw, out, in); with a 2 x 2
    kernel and stride 2 no two taps overlap, so out[n, 2i + a, 2j + b, co] = sum_ci x[n, i, j, ci] k[a, b, co, ci] + bias[co] --
    stated here as one einsum, against the oracle's torch.conv_transpose2d call."""
    import torch
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 5, 4, 6))
    k = rng.standard_normal((2, 2, 3, 6))
    b = rng.standard_normal(3)
    got = orc.conv2d_transpose_2x2(torch.from_numpy(x), torch.from_numpy(k), torch.from_numpy(b)).numpy()
    want = np.einsum("nijc,aboc->niajbo", x, k).reshape(2, 10, 8, 3) + b
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
