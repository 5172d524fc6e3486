"""CPU ORACLE -- test infrastructure only (imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product path).

A float32 CPU restatement of the Mask R-CNN hot path of SKA-INAF/caesar-mrcnn: NumPy for the
index/box arithmetic, PyTorch-CPU float32 ops for the dense contractions (so torch autograd yields the
reference gradients of the training graph).  Every function cites the reference lines it follows
(paths under /root/reference).

Pinning status
  * NumPy helpers (anchors, norm/denorm boxes, IoU, NumPy NMS, box deltas/refinement, RPN targets,
    image meta, resize_image at scale 1): PINNED against golden vectors generated in the build
    container by importing the reference's own functions (tests/golden/make_reference_fixtures.py).
  * Everything that the reference delegates to TensorFlow 1.13 / Keras 2.2.4 (conv/BN/pool padding,
    crop_and_resize, tf.image.non_max_suppression, top_k tie order, losses, SGD/clipnorm):
    PARITY UNPINNED -- TF1 is not installable here and the reference ships no tests or stored
    activations; these follow the published op semantics listed in SURVEY.md Appendix C ([3P]).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3   # keras.layers.BatchNormalization default epsilon [3P]


# =============================================================================================
#  NumPy twins of mrcnn/utils.py (pinned by golden vectors)
# =============================================================================================

def compute_backbone_shapes(strides, image_shape):
    """mrcnn/model.py:75-89."""
    return np.array([[int(math.ceil(image_shape[0] / s)), int(math.ceil(image_shape[1] / s))] for s in strides])


def generate_anchors(scales, ratios, shape, feature_stride, anchor_stride):
    """mrcnn/utils.py:652-688."""
    scales, ratios = np.meshgrid(np.array(scales), np.array(ratios))
    scales, ratios = scales.flatten(), ratios.flatten()
    heights = scales / np.sqrt(ratios)
    widths = scales * np.sqrt(ratios)
    shifts_y = np.arange(0, shape[0], anchor_stride) * feature_stride
    shifts_x = np.arange(0, shape[1], anchor_stride) * feature_stride
    shifts_x, shifts_y = np.meshgrid(shifts_x, shifts_y)
    box_widths, box_centers_x = np.meshgrid(widths, shifts_x)
    box_heights, box_centers_y = np.meshgrid(heights, shifts_y)
    centers = np.stack([box_centers_y, box_centers_x], axis=2).reshape([-1, 2])
    sizes = np.stack([box_heights, box_widths], axis=2).reshape([-1, 2])
    return np.concatenate([centers - 0.5 * sizes, centers + 0.5 * sizes], axis=1)


def generate_pyramid_anchors(scales, ratios, feature_shapes, feature_strides, anchor_stride):
    """mrcnn/utils.py:691-708."""
    return np.concatenate([generate_anchors(scales[i], ratios, feature_shapes[i], feature_strides[i], anchor_stride)
                           for i in range(len(scales))], axis=0)


def norm_boxes(boxes, shape):
    """mrcnn/utils.py:923-937."""
    h, w = shape
    scale = np.array([h - 1, w - 1, h - 1, w - 1])
    shift = np.array([0, 0, 1, 1])
    return np.divide((boxes - shift), scale).astype(np.float32)


def denorm_boxes(boxes, shape):
    """mrcnn/utils.py:940-954."""
    h, w = shape
    scale = np.array([h - 1, w - 1, h - 1, w - 1])
    shift = np.array([0, 0, 1, 1])
    return np.around(np.multiply(boxes, scale) + shift).astype(np.int32)


def get_anchors(config, image_shape):
    """MaskRCNN.get_anchors, mrcnn/model.py:2764-2784 (normalised float32 anchors)."""
    shapes = compute_backbone_shapes(config.BACKBONE_STRIDES, image_shape)
    a = generate_pyramid_anchors(config.RPN_ANCHOR_SCALES, config.RPN_ANCHOR_RATIOS, shapes,
                                 config.BACKBONE_STRIDES, config.RPN_ANCHOR_STRIDE)
    return norm_boxes(a, image_shape[:2])


def compute_iou(box, boxes, box_area, boxes_area):
    """mrcnn/utils.py:75-93."""
    y1 = np.maximum(box[0], boxes[:, 0])
    y2 = np.minimum(box[2], boxes[:, 2])
    x1 = np.maximum(box[1], boxes[:, 1])
    x2 = np.minimum(box[3], boxes[:, 3])
    intersection = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0)
    union = box_area + boxes_area[:] - intersection[:]
    return intersection / union


def compute_overlaps(boxes1, boxes2):
    """mrcnn/utils.py:147-163."""
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    overlaps = np.zeros((boxes1.shape[0], boxes2.shape[0]))
    for i in range(overlaps.shape[1]):
        overlaps[:, i] = compute_iou(boxes2[i], boxes1, area2[i], area1)
    return overlaps


def non_max_suppression_np(boxes, scores, threshold):
    """mrcnn/utils.py:188-222 (the reference's NumPy NMS; second source for the greedy rule)."""
    assert boxes.shape[0] > 0
    if boxes.dtype.kind != "f":
        boxes = boxes.astype(np.float32)
    y1, x1, y2, x2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (y2 - y1) * (x2 - x1)
    ixs = scores.argsort()[::-1]
    pick = []
    while len(ixs) > 0:
        i = ixs[0]
        pick.append(i)
        iou = compute_iou(boxes[i], boxes[ixs[1:]], area[i], area[ixs[1:]])
        remove_ixs = np.where(iou > threshold)[0] + 1
        ixs = np.delete(ixs, remove_ixs)
        ixs = np.delete(ixs, 0)
    return np.array(pick, dtype=np.int32)


def apply_box_deltas_np(boxes, deltas):
    """mrcnn/utils.py:225-246 == apply_box_deltas_graph mrcnn/model.py:287-308 (float32)."""
    boxes = boxes.astype(np.float32)
    height = boxes[:, 2] - boxes[:, 0]
    width = boxes[:, 3] - boxes[:, 1]
    center_y = boxes[:, 0] + np.float32(0.5) * height
    center_x = boxes[:, 1] + np.float32(0.5) * width
    center_y = center_y + deltas[:, 0] * height
    center_x = center_x + deltas[:, 1] * width
    height = height * np.exp(deltas[:, 2])
    width = width * np.exp(deltas[:, 3])
    y1 = center_y - np.float32(0.5) * height
    x1 = center_x - np.float32(0.5) * width
    y2 = y1 + height
    x2 = x1 + width
    return np.stack([y1, x1, y2, x2], axis=1)


def clip_boxes_np(boxes, window):
    """clip_boxes_graph, mrcnn/model.py:311-326."""
    wy1, wx1, wy2, wx2 = [np.float32(v) for v in window]
    y1 = np.maximum(np.minimum(boxes[:, 0], wy2), wy1)
    x1 = np.maximum(np.minimum(boxes[:, 1], wx2), wx1)
    y2 = np.maximum(np.minimum(boxes[:, 2], wy2), wy1)
    x2 = np.maximum(np.minimum(boxes[:, 3], wx2), wx1)
    return np.stack([y1, x1, y2, x2], axis=1)


def box_refinement_np(box, gt_box):
    """mrcnn/utils.py:275-298 == box_refinement_graph :249-272."""
    box = box.astype(np.float32)
    gt_box = gt_box.astype(np.float32)
    height = box[:, 2] - box[:, 0]
    width = box[:, 3] - box[:, 1]
    center_y = box[:, 0] + np.float32(0.5) * height
    center_x = box[:, 1] + np.float32(0.5) * width
    gt_height = gt_box[:, 2] - gt_box[:, 0]
    gt_width = gt_box[:, 3] - gt_box[:, 1]
    gt_center_y = gt_box[:, 0] + np.float32(0.5) * gt_height
    gt_center_x = gt_box[:, 1] + np.float32(0.5) * gt_width
    dy = (gt_center_y - center_y) / height
    dx = (gt_center_x - center_x) / width
    dh = np.log(gt_height / height)
    dw = np.log(gt_width / width)
    return np.stack([dy, dx, dh, dw], axis=1)


def trim_zeros(x):
    """mrcnn/utils.py:715-722."""
    return x[~np.all(x == 0, axis=1)]


def extract_bboxes(mask):
    """mrcnn/utils.py:49-72."""
    boxes = np.zeros([mask.shape[-1], 4], dtype=np.int32)
    for i in range(mask.shape[-1]):
        m = mask[:, :, i]
        hor = np.where(np.any(m, axis=0))[0]
        ver = np.where(np.any(m, axis=1))[0]
        if hor.shape[0]:
            x1, x2 = hor[[0, -1]]
            y1, y2 = ver[[0, -1]]
            x2 += 1
            y2 += 1
        else:
            x1, x2, y1, y2 = 0, 0, 0, 0
        boxes[i] = np.array([y1, x1, y2, x2])
    return boxes.astype(np.int32)


def compose_image_meta(image_id, original_image_shape, image_shape, window, scale, active_class_ids):
    """mrcnn/model.py:2891-2913."""
    return np.array([image_id] + list(original_image_shape) + list(image_shape) + list(window) + [scale] +
                    list(active_class_ids))


def build_rpn_targets(anchors, gt_class_ids, gt_boxes, rpn_train_anchors, rpn_bbox_std_dev, rng=np.random):
    """mrcnn/model.py:1536-1644 (rng replaces the module-level np.random for injectable sampling)."""
    rpn_match = np.zeros([anchors.shape[0]], dtype=np.int32)
    rpn_bbox = np.zeros((rpn_train_anchors, 4))
    crowd_ix = np.where(gt_class_ids < 0)[0]
    if crowd_ix.shape[0] > 0:
        non_crowd_ix = np.where(gt_class_ids > 0)[0]
        crowd_boxes = gt_boxes[crowd_ix]
        gt_class_ids = gt_class_ids[non_crowd_ix]
        gt_boxes = gt_boxes[non_crowd_ix]
        crowd_overlaps = compute_overlaps(anchors, crowd_boxes)
        no_crowd_bool = (np.amax(crowd_overlaps, axis=1) < 0.001)
    else:
        no_crowd_bool = np.ones([anchors.shape[0]], dtype=bool)
    overlaps = compute_overlaps(anchors, gt_boxes)
    anchor_iou_argmax = np.argmax(overlaps, axis=1)
    anchor_iou_max = overlaps[np.arange(overlaps.shape[0]), anchor_iou_argmax]
    rpn_match[(anchor_iou_max < 0.3) & (no_crowd_bool)] = -1
    gt_iou_argmax = np.argwhere(overlaps == np.max(overlaps, axis=0))[:, 0]
    rpn_match[gt_iou_argmax] = 1
    rpn_match[anchor_iou_max >= 0.7] = 1
    ids = np.where(rpn_match == 1)[0]
    extra = len(ids) - (rpn_train_anchors // 2)
    if extra > 0:
        ids = rng.choice(ids, extra, replace=False)
        rpn_match[ids] = 0
    ids = np.where(rpn_match == -1)[0]
    extra = len(ids) - (rpn_train_anchors - np.sum(rpn_match == 1))
    if extra > 0:
        ids = rng.choice(ids, extra, replace=False)
        rpn_match[ids] = 0
    ids = np.where(rpn_match == 1)[0]
    ix = 0
    for i, a in zip(ids, anchors[ids]):
        gt = gt_boxes[anchor_iou_argmax[i]]
        gt_h, gt_w = gt[2] - gt[0], gt[3] - gt[1]
        gt_cy, gt_cx = gt[0] + 0.5 * gt_h, gt[1] + 0.5 * gt_w
        a_h, a_w = a[2] - a[0], a[3] - a[1]
        a_cy, a_cx = a[0] + 0.5 * a_h, a[1] + 0.5 * a_w
        rpn_bbox[ix] = [(gt_cy - a_cy) / a_h, (gt_cx - a_cx) / a_w, np.log(gt_h / a_h), np.log(gt_w / a_w)]
        rpn_bbox[ix] /= rpn_bbox_std_dev
        ix += 1
    return rpn_match, rpn_bbox


class KeyedChoice(object):
    """Stand-in for the ``rng`` of build_rpn_targets: ``choice(ids, n, replace=False)`` returns the n members
    of ids with the smallest injected keys (ties -> lower id).  This is the rule of the device kernel
    (csrc/rpn_targets.hip), which replaces the two np.random.choice draws of model.py:1588-1601 the same
    way rand_keys replace tf.random_shuffle in the detection targets."""

    def __init__(self, keys):
        self.keys = np.asarray(keys, dtype=np.float32)

    def choice(self, ids, n, replace=False):
        assert not replace
        ids = np.asarray(ids)
        order = np.lexsort((ids, self.keys[ids]))
        return ids[order[:n]]


# =============================================================================================
#  Restated TensorFlow / Keras op semantics ([3P], parity unpinned)
# =============================================================================================

def tf_top_k_indices(scores, k):
    """tf.nn.top_k(sorted=True).indices: descending, equal values -> lower index first."""
    return np.argsort(-scores.astype(np.float32), kind="stable")[:k]


def tf_iou(boxes, i, js):
    """IOU of tensorflow/core/kernels/non_max_suppression_op.cc (1.13), float32 arithmetic."""
    b = boxes
    ymin_i, xmin_i = np.minimum(b[i, 0], b[i, 2]), np.minimum(b[i, 1], b[i, 3])
    ymax_i, xmax_i = np.maximum(b[i, 0], b[i, 2]), np.maximum(b[i, 1], b[i, 3])
    ymin_j, xmin_j = np.minimum(b[js, 0], b[js, 2]), np.minimum(b[js, 1], b[js, 3])
    ymax_j, xmax_j = np.maximum(b[js, 0], b[js, 2]), np.maximum(b[js, 1], b[js, 3])
    area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i)
    area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j)
    iy0, ix0 = np.maximum(ymin_i, ymin_j), np.maximum(xmin_i, xmin_j)
    iy1, ix1 = np.minimum(ymax_i, ymax_j), np.minimum(xmax_i, xmax_j)
    inter = np.maximum(iy1 - iy0, np.float32(0)) * np.maximum(ix1 - ix0, np.float32(0))
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / (area_i + area_j - inter)
    iou = np.where((area_i <= 0) | (area_j <= 0), np.float32(0), iou)
    return iou


def tf_non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    """tf.image.non_max_suppression: greedy in descending score order (ties: lower index first --
    TF 1.13 leaves it unspecified), suppress when IoU > threshold, stop at max_output_size."""
    boxes = boxes.astype(np.float32)
    order = np.argsort(-scores.astype(np.float32), kind="stable")
    alive = np.ones(len(order), dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    for pos in range(len(order)):
        if not alive[pos]:
            continue
        if len(keep) >= max_output_size:
            break
        i = order[pos]
        keep.append(i)
        rest = np.where(alive[pos + 1:])[0] + pos + 1
        if rest.size:
            iou = tf_iou(boxes, i, order[rest])
            alive[rest[iou > thr]] = False
    return np.array(keep, dtype=np.int64)


def same_pad(size, k, stride):
    """TF SAME: out = ceil(size/stride), pad_before = total // 2, pad_after = total - pad_before."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2, total - total // 2


def conv2d_nhwc(x, w_hwio, bias=None, stride=1, padding="same"):
    """KL.Conv2D on NHWC torch tensors. w_hwio [KH,KW,Cin,Cout]."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    xc = x.permute(0, 3, 1, 2)
    if padding == "same":
        _, pt, pb = same_pad(x.shape[1], kh, stride)
        _, pl, pr = same_pad(x.shape[2], kw, stride)
        xc = F.pad(xc, (pl, pr, pt, pb))
    elif padding != "valid":
        ph, pw = padding
        xc = F.pad(xc, (pw, pw, ph, ph))
    y = F.conv2d(xc, w_hwio.permute(3, 2, 0, 1), bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def batchnorm_frozen(x, gamma, beta, mean, var):
    """BatchNorm.call(training=False), mrcnn/model.py:57-72: gamma*(x-mean)/sqrt(var+eps)+beta."""
    return (x - mean) * (gamma / torch.sqrt(var + BN_EPS)) + beta


def maxpool3x3s2_same(x):
    """KL.MaxPooling2D((3,3), strides=(2,2), padding='same'), mrcnn/model.py:187."""
    xc = x.permute(0, 3, 1, 2)
    _, pt, pb = same_pad(x.shape[1], 3, 2)
    _, pl, pr = same_pad(x.shape[2], 3, 2)
    xc = F.pad(xc, (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(xc, 3, 2).permute(0, 2, 3, 1)


def conv2d_transpose_2x2(x, k_keras, bias):
    """KL.Conv2DTranspose(C, (2,2), strides=2): k_keras (2,2,Cout,Cin); out[n,2i+a,2j+b,co]."""
    w = k_keras.permute(3, 2, 0, 1)     # torch conv_transpose2d weight: [Cin, Cout, kh, kw]
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w, bias, stride=2)
    return y.permute(0, 2, 3, 1)


def crop_and_resize(image, boxes, box_idx, crop_h, crop_w):
    """tf.image.crop_and_resize(method='bilinear', extrapolation_value=0) on NHWC torch `image`;
    boxes [n,4] float32 (NumPy or torch, no gradient), box_idx [n].  Differentiable w.r.t. image."""
    boxes = torch.as_tensor(boxes, dtype=torch.float32)
    box_idx = torch.as_tensor(np.asarray(box_idx), dtype=torch.long)
    H, W = image.shape[1], image.shape[2]
    n = boxes.shape[0]
    y1, x1, y2, x2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    fH, fW = torch.tensor(float(H - 1)), torch.tensor(float(W - 1))
    if crop_h > 1:
        hs = (y2 - y1) * fH / float(crop_h - 1)
        in_y = (y1 * fH)[:, None] + torch.arange(crop_h, dtype=torch.float32)[None, :] * hs[:, None]
    else:
        in_y = (0.5 * (y1 + y2) * fH)[:, None]
    if crop_w > 1:
        ws = (x2 - x1) * fW / float(crop_w - 1)
        in_x = (x1 * fW)[:, None] + torch.arange(crop_w, dtype=torch.float32)[None, :] * ws[:, None]
    else:
        in_x = (0.5 * (x1 + x2) * fW)[:, None]
    vy = ~((in_y < 0) | (in_y > fH))
    vx = ~((in_x < 0) | (in_x > fW))
    iy = torch.where(vy, in_y, torch.zeros_like(in_y))
    ix = torch.where(vx, in_x, torch.zeros_like(in_x))
    top, bot = torch.floor(iy).long(), torch.ceil(iy).long()
    lef, rig = torch.floor(ix).long(), torch.ceil(ix).long()
    yl = (iy - torch.floor(iy))[:, :, None, None]
    xl = (ix - torch.floor(ix))[:, None, :, None]
    bi = box_idx[:, None, None]
    tl = image[bi, top[:, :, None], lef[:, None, :]]
    tr = image[bi, top[:, :, None], rig[:, None, :]]
    bl = image[bi, bot[:, :, None], lef[:, None, :]]
    br = image[bi, bot[:, :, None], rig[:, None, :]]
    t = tl + (tr - tl) * xl
    b = bl + (br - bl) * xl
    out = t + (b - t) * yl
    valid = (vy[:, :, None] & vx[:, None, :])[..., None]
    return torch.where(valid, out, torch.zeros_like(out))


def roi_levels(boxes, image_area):
    """Level assignment of PyramidROIAlign, mrcnn/model.py:465-477 (float32, tf.round = half-even).
    -inf / nan cast to int32 gives INT_MIN in TF on x86, which the clamp maps to level 2."""
    b = torch.as_tensor(boxes, dtype=torch.float32)
    h = b[..., 2] - b[..., 0]
    w = b[..., 3] - b[..., 1]
    area = torch.tensor(float(image_area), dtype=torch.float32)
    v = torch.log(torch.sqrt(h * w) / (torch.tensor(224.0) / torch.sqrt(area))) / torch.log(torch.tensor(2.0))
    lv = torch.where(torch.isfinite(v), 4 + torch.round(v), torch.full_like(v, -1e9))
    return torch.clamp(lv, 2, 5).long()


def pyramid_roi_align(boxes, fms, pool, image_area):
    """PyramidROIAlign.call, mrcnn/model.py:452-531.  boxes [B,R,4]; fms [P2..P5] NHWC torch."""
    boxes_t = torch.as_tensor(boxes, dtype=torch.float32)
    B, R = boxes_t.shape[0], boxes_t.shape[1]
    lv = roi_levels(boxes_t, image_area)
    out = torch.zeros((B, R, pool, pool, fms[0].shape[3]), dtype=torch.float32)
    for i, level in enumerate(range(2, 6)):
        ix = torch.nonzero(lv == level)
        if ix.shape[0] == 0:
            continue
        lb = boxes_t[ix[:, 0], ix[:, 1]]
        pooled = crop_and_resize(fms[i], lb, ix[:, 0], pool, pool)
        out = out.index_put((ix[:, 0], ix[:, 1]), pooled)
    return out


# =============================================================================================
#  The graph (MaskRCNN.build, mrcnn/model.py:1935-2166)
# =============================================================================================

class OracleMaskRCNN(object):
    """weights: {"<layer>/kernel", "<layer>/bias", "<bn>/gamma|beta|moving_mean|moving_variance"} as
    NumPy arrays in the Keras layouts (Conv2D HWIO, Dense [in,out], Conv2DTranspose (2,2,out,in))."""

    def __init__(self, config, weights, requires_grad=False):
        self.config = config
        self.w = {}
        for k, v in weights.items():
            t = torch.tensor(np.asarray(v, dtype=np.float32))
            if requires_grad and not (k.endswith("moving_mean") or k.endswith("moving_variance")):
                t.requires_grad_(True)
            self.w[k] = t

    # ---- building blocks -----------------------------------------------------------------
    def _conv(self, x, name, stride=1, padding="same"):
        k = self.w[name + "/kernel"]
        if k.dim() == 2:
            k = k.reshape(1, 1, k.shape[0], k.shape[1])
        return conv2d_nhwc(x, k, self.w[name + "/bias"], stride, padding)

    def _bn(self, x, name):
        return batchnorm_frozen(x, self.w[name + "/gamma"], self.w[name + "/beta"], self.w[name + "/moving_mean"],
                                self.w[name + "/moving_variance"])

    def _block(self, x, stage, block, has_shortcut, stride):
        """identity_block / conv_block, mrcnn/model.py:99-172."""
        cb, bb = "res%d%s_branch" % (stage, block), "bn%d%s_branch" % (stage, block)
        y = F.relu(self._bn(self._conv(x, cb + "2a", stride, "valid"), bb + "2a"))
        y = F.relu(self._bn(self._conv(y, cb + "2b", 1, "same"), bb + "2b"))
        y = self._bn(self._conv(y, cb + "2c", 1, "valid"), bb + "2c")
        sc = self._bn(self._conv(x, cb + "1", stride, "valid"), bb + "1") if has_shortcut else x
        return F.relu(y + sc)

    def backbone(self, image):
        """resnet_graph / custom_backbone, mrcnn/model.py:175-244."""
        arch = self.config.BACKBONE
        n4 = {"resnet50": 5, "resnet101": 22, "custom": 1}[arch]
        blocks = {2: "abc", 3: "abcd", 4: "a" + "".join(chr(98 + i) for i in range(n4)), 5: "abc"}
        x = self._conv(image, "conv1", 2, (3, 3))
        x = F.relu(self._bn(x, "bn_conv1"))
        x = maxpool3x3s2_same(x)
        outs = []
        for stage in (2, 3, 4, 5):
            for bi, b in enumerate(blocks[stage]):
                x = self._block(x, stage, b, bi == 0, 2 if (bi == 0 and stage > 2) else 1)
            outs.append(x)
        return outs

    def fpn(self, C2, C3, C4, C5):
        """mrcnn/model.py:2005-2026."""
        up = lambda t: t.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)   # UpSampling2D nearest
        P5 = self._conv(C5, "fpn_c5p5", 1, "valid")
        P4 = up(P5) + self._conv(C4, "fpn_c4p4", 1, "valid")
        P3 = up(P4) + self._conv(C3, "fpn_c3p3", 1, "valid")
        P2 = up(P3) + self._conv(C2, "fpn_c2p2", 1, "valid")
        P2 = self._conv(P2, "fpn_p2")
        P3 = self._conv(P3, "fpn_p3")
        P4 = self._conv(P4, "fpn_p4")
        P5 = self._conv(P5, "fpn_p5")
        P6 = P5[:, ::2, ::2, :]                 # MaxPooling2D(pool_size=(1,1), strides=2)
        return [P2, P3, P4, P5, P6]

    def rpn(self, pyramid):
        """rpn_graph + level concat, mrcnn/model.py:916-957, 2040-2055."""
        logits, probs, bbox = [], [], []
        for p in pyramid:
            shared = F.relu(self._conv(p, "rpn_conv_shared"))
            x = self._conv(shared, "rpn_class_raw", 1, "valid")
            lg = x.reshape(x.shape[0], -1, 2)
            logits.append(lg)
            probs.append(torch.softmax(lg, dim=-1))
            x = self._conv(shared, "rpn_bbox_pred", 1, "valid")
            bbox.append(x.reshape(x.shape[0], -1, 4))
        return torch.cat(logits, 1), torch.cat(probs, 1), torch.cat(bbox, 1)

    def proposal_layer(self, rpn_probs, rpn_bbox, anchors, proposal_count, detail=False):
        """ProposalLayer.call, mrcnn/model.py:350-403.  anchors [A,4] normalised (same for every image)."""
        cfg = self.config
        probs = rpn_probs.detach().numpy()
        deltas_all = rpn_bbox.detach().numpy() * np.reshape(cfg.RPN_BBOX_STD_DEV, [1, 1, 4]).astype(np.float32)
        B, A = probs.shape[0], probs.shape[1]
        k = min(cfg.PRE_NMS_LIMIT, A)
        out = np.zeros((B, proposal_count, 4), np.float32)
        details = []
        for b in range(B):
            scores = probs[b, :, 1]
            ix = tf_top_k_indices(scores, k)
            boxes = apply_box_deltas_np(anchors[ix], deltas_all[b, ix])
            boxes = clip_boxes_np(boxes, [0, 0, 1, 1]).astype(np.float32)
            keep = tf_non_max_suppression(boxes, scores[ix], proposal_count, cfg.RPN_NMS_THRESHOLD)
            out[b, :len(keep)] = boxes[keep]
            details.append((ix, boxes, keep))
        return (out, details) if detail else out

    def classifier_head(self, rois, fms, image_area):
        """fpn_classifier_graph, mrcnn/model.py:986-1039."""
        cfg = self.config
        x = pyramid_roi_align(rois, fms, cfg.POOL_SIZE, image_area)
        B, R = x.shape[0], x.shape[1]
        x = x.reshape(B * R, cfg.POOL_SIZE, cfg.POOL_SIZE, -1)
        x = F.relu(self._bn(self._conv(x, "mrcnn_class_conv1", 1, "valid"), "mrcnn_class_bn1"))
        x = F.relu(self._bn(self._conv(x, "mrcnn_class_conv2", 1, "valid"), "mrcnn_class_bn2"))
        logits = self._conv(x, "mrcnn_class_logits", 1, "valid").reshape(B, R, -1)
        probs = torch.softmax(logits, dim=-1)
        bbox = self._conv(x, "mrcnn_bbox_fc", 1, "valid").reshape(B, R, cfg.NUM_CLASSES, 4)
        return logits, probs, bbox

    def mask_head(self, rois, fms, image_area):
        """build_fpn_mask_graph, mrcnn/model.py:1042-1091."""
        cfg = self.config
        x = pyramid_roi_align(rois, fms, cfg.MASK_POOL_SIZE, image_area)
        B, R = x.shape[0], x.shape[1]
        x = x.reshape(B * R, cfg.MASK_POOL_SIZE, cfg.MASK_POOL_SIZE, -1)
        for i in range(1, 5):
            x = F.relu(self._bn(self._conv(x, "mrcnn_mask_conv%d" % i), "mrcnn_mask_bn%d" % i))
        x = F.relu(conv2d_transpose_2x2(x, self.w["mrcnn_mask_deconv/kernel"], self.w["mrcnn_mask_deconv/bias"]))
        x = torch.sigmoid(self._conv(x, "mrcnn_mask", 1, "valid"))
        return x.reshape(B, R, x.shape[1], x.shape[2], x.shape[3])

    def refine_detections(self, rois, probs, deltas, window):
        """refine_detections_graph for ONE image, mrcnn/model.py:770-865 (NumPy)."""
        cfg = self.config
        N = probs.shape[0]
        class_ids = np.argmax(probs, axis=1).astype(np.int32)
        class_scores = probs[np.arange(N), class_ids]
        deltas_specific = deltas[np.arange(N), class_ids] * cfg.BBOX_STD_DEV.astype(np.float32)
        refined = clip_boxes_np(apply_box_deltas_np(rois, deltas_specific.astype(np.float32)), window).astype(np.float32)
        keep = np.where(class_ids > 0)[0]
        if cfg.DETECTION_MIN_CONFIDENCE:
            keep = np.intersect1d(keep, np.where(class_scores >= cfg.DETECTION_MIN_CONFIDENCE)[0])
        pre_ids, pre_scores, pre_rois = class_ids[keep], class_scores[keep], refined[keep]
        nms_keep = []
        for cid in np.unique(pre_ids):
            ixs = np.where(pre_ids == cid)[0]
            ck = tf_non_max_suppression(pre_rois[ixs], pre_scores[ixs], cfg.DETECTION_MAX_INSTANCES,
                                        cfg.DETECTION_NMS_THRESHOLD)
            nms_keep.extend(keep[ixs[ck]].tolist())
        keep = np.intersect1d(keep, np.array(nms_keep, dtype=np.int64))      # sorted ascending
        num_keep = min(len(keep), cfg.DETECTION_MAX_INSTANCES)
        top = tf_top_k_indices(class_scores[keep], num_keep)
        keep = keep[top]
        det = np.zeros((cfg.DETECTION_MAX_INSTANCES, 6), np.float32)
        det[:len(keep), :4] = refined[keep]
        det[:len(keep), 4] = class_ids[keep].astype(np.float32)
        det[:len(keep), 5] = class_scores[keep]
        return det

    # ---- inference graph (mode == "inference", mrcnn/model.py:2133-2159) --------------------------
    def forward_inference(self, images, windows_norm, anchors):
        """images [B,H,W,3] float32 (molded), windows_norm [B,4], anchors [A,4] normalised."""
        cfg = self.config
        with torch.no_grad():
            x = torch.as_tensor(images, dtype=torch.float32)
            image_area = float(x.shape[1] * x.shape[2])
            C2, C3, C4, C5 = self.backbone(x)
            pyr = self.fpn(C2, C3, C4, C5)
            rpn_logits, rpn_probs, rpn_bbox = self.rpn(pyr)
            rois = self.proposal_layer(rpn_probs, rpn_bbox, anchors, cfg.POST_NMS_ROIS_INFERENCE)
            logits, probs, bbox = self.classifier_head(rois, pyr[:4], image_area)
            dets = np.stack([self.refine_detections(rois[b], probs[b].numpy(), bbox[b].numpy(), windows_norm[b])
                             for b in range(x.shape[0])])
            masks = self.mask_head(dets[..., :4], pyr[:4], image_area)
        return {"detections": dets, "mrcnn_class": probs.numpy(), "mrcnn_bbox": bbox.numpy(),
                "mrcnn_mask": masks.numpy(), "rpn_rois": rois, "rpn_class": rpn_probs.numpy(),
                "rpn_bbox": rpn_bbox.numpy(), "pyramid": [p.numpy() for p in pyr]}

    # ---- training graph ------------------------------------------------------------------------------
    def detection_targets(self, proposals, gt_class_ids, gt_boxes, gt_masks, rand_keys):
        """detection_targets_graph for ONE image, mrcnn/model.py:570-705.  tf.random.shuffle is replaced
        by "ascending rand_keys (indexed by proposal row), ties by lower row"; gt_masks [H,W,G] bool."""
        cfg = self.config
        T = cfg.TRAIN_ROIS_PER_IMAGE
        prow = np.where(np.sum(np.abs(proposals), axis=1) != 0)[0]
        props = proposals[prow]
        nz = np.sum(np.abs(gt_boxes), axis=1) != 0
        gtb, gtc, gidx = gt_boxes[nz], gt_class_ids[nz], np.where(nz)[0]
        crowd = gtb[gtc < 0]
        inst = gtc > 0
        gtb, gtc, gidx = gtb[inst], gtc[inst], gidx[inst]

        def overlaps(b1, b2):      # overlaps_graph, model.py:541-567 (float32)
            if b1.shape[0] == 0 or b2.shape[0] == 0:
                return np.zeros((b1.shape[0], b2.shape[0]), np.float32)
            y1 = np.maximum(b1[:, None, 0], b2[None, :, 0]); x1 = np.maximum(b1[:, None, 1], b2[None, :, 1])
            y2 = np.minimum(b1[:, None, 2], b2[None, :, 2]); x2 = np.minimum(b1[:, None, 3], b2[None, :, 3])
            inter = np.maximum(x2 - x1, np.float32(0)) * np.maximum(y2 - y1, np.float32(0))
            a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
            a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
            with np.errstate(divide="ignore", invalid="ignore"):
                return inter / (a1[:, None] + a2[None, :] - inter)

        ov = overlaps(props, gtb)
        cov = overlaps(props, crowd)
        crowd_max = cov.max(axis=1) if cov.shape[1] else np.full(props.shape[0], -np.inf, np.float32)
        iou_max = ov.max(axis=1) if ov.shape[1] else np.full(props.shape[0], -np.inf, np.float32)
        pos = np.where(iou_max >= 0.5)[0]
        neg = np.where((iou_max < 0.5) & (crowd_max < 0.001))[0]
        keys = rand_keys[prow]
        pos = pos[np.argsort(keys[pos], kind="stable")][:int(T * cfg.ROI_POSITIVE_RATIO)]
        P = len(pos)
        r = np.float32(1.0 / cfg.ROI_POSITIVE_RATIO)
        N = int(np.float32(r * np.float32(P))) - P
        neg = neg[np.argsort(keys[neg], kind="stable")][:max(N, 0)]
        assign = ov[pos].argmax(axis=1) if (P and ov.shape[1]) else np.zeros(0, np.int64)
        roi_gt = gtb[assign]
        deltas = box_refinement_np(props[pos], roi_gt) / cfg.BBOX_STD_DEV.astype(np.float32) if P else np.zeros((0, 4), np.float32)
        if P:
            m = torch.tensor(gt_masks[:, :, gidx[assign]].astype(np.float32)).permute(2, 0, 1)[..., None]
            boxes = props[pos]
            if cfg.USE_MINI_MASK:
                gh = roi_gt[:, 2] - roi_gt[:, 0]; gw = roi_gt[:, 3] - roi_gt[:, 1]
                boxes = np.stack([(boxes[:, 0] - roi_gt[:, 0]) / gh, (boxes[:, 1] - roi_gt[:, 1]) / gw,
                                  (boxes[:, 2] - roi_gt[:, 0]) / gh, (boxes[:, 3] - roi_gt[:, 1]) / gw], 1)
            masks = torch.round(crop_and_resize(m, boxes, np.arange(P), cfg.MASK_SHAPE[0], cfg.MASK_SHAPE[1]))[..., 0].numpy()
        else:
            masks = np.zeros((0, cfg.MASK_SHAPE[0], cfg.MASK_SHAPE[1]), np.float32)
        rois = np.zeros((T, 4), np.float32)
        rois[:P] = props[pos]
        rois[P:P + len(neg)] = props[neg]
        cls = np.zeros(T, np.int32); cls[:P] = gtc[assign]
        tb = np.zeros((T, 4), np.float32); tb[:P] = deltas
        tm = np.zeros((T, cfg.MASK_SHAPE[0], cfg.MASK_SHAPE[1]), np.float32); tm[:P] = masks
        return rois, cls, tb, tm, (P, len(neg))

    def losses(self, rpn_match, rpn_bbox_t, rpn_logits, rpn_bbox, tcls, tbbox, tmask, active_class_ids, logits,
               mbbox, mmask):
        """The five loss graphs, mrcnn/model.py:1098-1270 (torch, differentiable)."""
        cfg = self.config
        rm = torch.as_tensor(rpn_match).reshape(rpn_logits.shape[0], -1)
        sel = rm != 0
        if sel.any():
            l1 = F.cross_entropy(rpn_logits[sel], (rm[sel] == 1).long(), reduction="mean")
        else:
            l1 = torch.tensor(0.0)
        pos = rm == 1
        if pos.any():
            tgt = torch.cat([torch.as_tensor(rpn_bbox_t[b], dtype=torch.float32)[:int(pos[b].sum())]
                             for b in range(rm.shape[0])], 0)
            l2 = smooth_l1(tgt, rpn_bbox[pos]).mean()
        else:
            l2 = torch.tensor(0.0)
        tc = torch.as_tensor(tcls).reshape(-1).long()
        lg = logits.reshape(-1, logits.shape[-1])
        pred_active = torch.as_tensor(active_class_ids[0], dtype=torch.float32)[lg.argmax(dim=1)]
        ce = F.cross_entropy(lg, tc, reduction="none")
        l3 = (ce * pred_active).sum() / pred_active.sum()
        prow = torch.nonzero(tc > 0)[:, 0]
        if prow.numel():
            pb = mbbox.reshape(-1, mbbox.shape[2], 4)[prow, tc[prow]]
            l4 = smooth_l1(torch.as_tensor(tbbox, dtype=torch.float32).reshape(-1, 4)[prow], pb).mean()
            yt = torch.as_tensor(tmask, dtype=torch.float32).reshape(-1, tmask.shape[-2], tmask.shape[-1])[prow]
            yp = mmask.reshape(-1, mmask.shape[2], mmask.shape[3], mmask.shape[4])[prow, :, :, tc[prow]]
            if cfg.MASK_LOSS_FUNCTION == "dice_coef_loss":
                sm = 1e-7
                inter = (yt * yp).sum()
                l5 = 1 - (2. * inter + sm) / (yt.sum() + yp.sum() + sm)
            else:
                l5 = keras_binary_crossentropy(yt, yp).mean()
        else:
            l4 = torch.tensor(0.0)
            l5 = torch.tensor(0.0)
        return [l1, l2, l3, l4, l5]

    def forward_training(self, images, rpn_match, rpn_bbox_t, gt_class_ids, gt_boxes, gt_masks, active_class_ids,
                         anchors, rand_keys, forced=None):
        """Training graph, mrcnn/model.py:2068-2132.  gt_boxes in pixels; gt_masks [B,H,W,G].
        `forced` = {"rois", "target_class_ids", "target_bbox", "target_mask"} replaces the (discrete,
        gradient-free) proposal + target stages, so a test can compare the differentiable part on
        identical ROIs."""
        cfg = self.config
        x = torch.as_tensor(images, dtype=torch.float32)
        H, W = x.shape[1], x.shape[2]
        image_area = float(H * W)
        C2, C3, C4, C5 = self.backbone(x)
        pyr = self.fpn(C2, C3, C4, C5)
        rpn_logits, rpn_probs, rpn_bbox = self.rpn(pyr)
        if forced is None:
            rpn_rois = self.proposal_layer(rpn_probs, rpn_bbox, anchors, cfg.POST_NMS_ROIS_TRAINING)
            # norm_boxes_graph, model.py:3003-3017
            gtn = ((gt_boxes.astype(np.float32) - np.array([0., 0., 1., 1.], np.float32)) /
                   (np.array([H, W, H, W], np.float32) - np.float32(1.0))).astype(np.float32)
            tg = [self.detection_targets(rpn_rois[b], gt_class_ids[b], gtn[b], gt_masks[b], rand_keys[b])
                  for b in range(x.shape[0])]
            rois = np.stack([t[0] for t in tg]); tcls = np.stack([t[1] for t in tg])
            tbbox = np.stack([t[2] for t in tg]); tmask = np.stack([t[3] for t in tg])
        else:
            rpn_rois, tg = None, []
            rois, tcls = forced["rois"], forced["target_class_ids"]
            tbbox, tmask = forced["target_bbox"], forced["target_mask"]
        logits, probs, mbbox = self.classifier_head(rois, pyr[:4], image_area)
        mmask = self.mask_head(rois, pyr[:4], image_area)
        ls = self.losses(rpn_match, rpn_bbox_t, rpn_logits, rpn_bbox, tcls, tbbox, tmask, active_class_ids, logits,
                         mbbox, mmask)
        return {"losses": ls, "rpn_class_logits": rpn_logits, "rpn_class": rpn_probs, "rpn_bbox": rpn_bbox,
                "rpn_rois": rpn_rois, "rois": rois, "target_class_ids": tcls, "target_bbox": tbbox,
                "target_mask": tmask, "mrcnn_class_logits": logits, "mrcnn_class": probs, "mrcnn_bbox": mbbox,
                "mrcnn_mask": mmask, "pyramid": pyr, "counts": [t[4] for t in tg]}

    def total_loss(self, losses, trainable=None):
        """MaskRCNN.compile, mrcnn/model.py:2267-2291: weighted losses + sum_w l2(WD)(w)/size(w) over the
        trainable weights whose names lack gamma/beta (biases are regularised)."""
        cfg = self.config
        names = ["rpn_class_loss", "rpn_bbox_loss", "mrcnn_class_loss", "mrcnn_bbox_loss", "mrcnn_mask_loss"]
        total = torch.tensor(0.0)
        for n, l in zip(names, losses):
            if cfg.USE_LOSSES.get(n, True):
                total = total + l * cfg.LOSS_WEIGHTS.get(n, 1.)
        reg = torch.tensor(0.0)
        for k, w in self.w.items():
            if "gamma" in k or "beta" in k or "moving_" in k:
                continue
            if trainable is not None and not trainable(k):
                continue
            reg = reg + cfg.WEIGHT_DECAY * (w ** 2).sum() / float(w.numel())
        return total + reg


def smooth_l1(y_true, y_pred):
    """mrcnn/model.py:1098-1105."""
    diff = torch.abs(y_true - y_pred)
    less = (diff < 1.0).float()
    return less * 0.5 * diff ** 2 + (1 - less) * (diff - 0.5)


def keras_binary_crossentropy(target, output):
    """K.binary_crossentropy(from_logits=False) of Keras 2.2.4 / TF backend [3P]: clip to
    [eps, 1-eps], logit, tf.nn.sigmoid_cross_entropy_with_logits."""
    eps = 1e-7
    o = torch.clamp(output, eps, 1 - eps)
    x = torch.log(o / (1 - o))
    return torch.relu(x) - x * target + torch.log1p(torch.exp(-torch.abs(x)))   # relu'(0)=0, sign(0)=0 as in TF


def sgd_step(params, grads, velocity, lr, momentum, clipnorm):
    """keras.optimizers.SGD.get_updates with clipnorm (Keras 2.2.4 [3P]): global norm over all grads,
    g *= clipnorm/norm when norm >= clipnorm; v = momentum*v - lr*g; p += v.  Dicts of NumPy arrays."""
    norm = np.sqrt(sum(float(np.sum(np.square(g.astype(np.float64)))) for g in grads.values()))
    scale = clipnorm / norm if (clipnorm > 0 and norm >= clipnorm) else 1.0
    for k in params:
        g = grads[k] * np.float32(scale)
        velocity[k] = np.float32(momentum) * velocity[k] - np.float32(lr) * g
        params[k] = params[k] + velocity[k]
    return norm
