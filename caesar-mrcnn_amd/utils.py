"""Host-side (NumPy) helpers around the device graph -- the pre/post steps that the reference also
runs on the host in mrcnn/utils.py and mrcnn/model.py (anchors, box normalisation, image molding and
un-molding of detections).  Function names and argument meaning follow the reference so callers of
``mrcnn.utils`` can switch; the bodies are written for this code base.
"""
import math

import numpy as np


# ---- anchors (mrcnn/utils.py:652-708, mrcnn/model.py:75-89, 2764-2784) ---------------------------
def compute_backbone_shapes(config, image_shape):
    if callable(config.BACKBONE):
        return config.COMPUTE_BACKBONE_SHAPE(image_shape)
    return np.array([[int(math.ceil(image_shape[0] / s)), int(math.ceil(image_shape[1] / s))]
                     for s in config.BACKBONE_STRIDES])


def generate_anchors(scales, ratios, shape, feature_stride, anchor_stride):
    """Anchors of one pyramid level, ordered (y, x, ratio) row-major like the RPN head output."""
    s, r = np.meshgrid(np.array(scales), np.array(ratios))
    s, r = s.flatten(), r.flatten()
    hs, ws = s / np.sqrt(r), s * np.sqrt(r)
    ys = np.arange(0, shape[0], anchor_stride) * feature_stride
    xs = np.arange(0, shape[1], anchor_stride) * feature_stride
    xs, ys = np.meshgrid(xs, ys)
    bw, cx = np.meshgrid(ws, xs)
    bh, cy = np.meshgrid(hs, ys)
    ctr = np.stack([cy, cx], axis=2).reshape([-1, 2])
    size = np.stack([bh, bw], axis=2).reshape([-1, 2])
    return np.concatenate([ctr - 0.5 * size, ctr + 0.5 * size], axis=1)


def generate_pyramid_anchors(scales, ratios, feature_shapes, feature_strides, anchor_stride):
    return np.concatenate([generate_anchors(scales[i], ratios, feature_shapes[i], feature_strides[i], anchor_stride)
                           for i in range(len(scales))], axis=0)


def norm_boxes(boxes, shape):
    """pixel (y2,x2 exclusive) -> normalised (inclusive) coordinates, float32 (mrcnn/utils.py:923)."""
    h, w = shape
    return np.divide(boxes - np.array([0, 0, 1, 1]), np.array([h - 1, w - 1, h - 1, w - 1])).astype(np.float32)


def denorm_boxes(boxes, shape):
    h, w = shape
    return np.around(np.multiply(boxes, np.array([h - 1, w - 1, h - 1, w - 1])) + np.array([0, 0, 1, 1])).astype(np.int32)


def get_anchors(config, image_shape):
    shapes = compute_backbone_shapes(config, image_shape)
    a = generate_pyramid_anchors(config.RPN_ANCHOR_SCALES, config.RPN_ANCHOR_RATIOS, shapes, config.BACKBONE_STRIDES,
                                 config.RPN_ANCHOR_STRIDE)
    return norm_boxes(a, image_shape[:2])


# ---- boxes ----------------------------------------------------------------------------------------
def extract_bboxes(mask):
    """[H,W,N] masks -> [N,(y1,x1,y2,x2)] int32, zeros for empty masks (mrcnn/utils.py:49)."""
    n = mask.shape[-1]
    boxes = np.zeros([n, 4], dtype=np.int32)
    for i in range(n):
        m = mask[:, :, i]
        cols = np.where(np.any(m, axis=0))[0]
        rows = np.where(np.any(m, axis=1))[0]
        if cols.shape[0]:
            boxes[i] = [rows[0], cols[0], rows[-1] + 1, cols[-1] + 1]
    return boxes


def compute_iou(box, boxes, box_area, boxes_area):
    y1 = np.maximum(box[0], boxes[:, 0])
    y2 = np.minimum(box[2], boxes[:, 2])
    x1 = np.maximum(box[1], boxes[:, 1])
    x2 = np.minimum(box[3], boxes[:, 3])
    inter = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0)
    return inter / (box_area + boxes_area[:] - inter[:])


def compute_overlaps(boxes1, boxes2):
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    out = np.zeros((boxes1.shape[0], boxes2.shape[0]))
    for i in range(out.shape[1]):
        out[:, i] = compute_iou(boxes2[i], boxes1, a2[i], a1)
    return out


def trim_zeros(x):
    assert len(x.shape) == 2
    return x[~np.all(x == 0, axis=1)]


# ---- image resize ----------------------------------------------------------------------------------
def resize(image, output_shape, order=1, mode='constant', cval=0, clip=True, preserve_range=False,
           anti_aliasing=False, anti_aliasing_sigma=None):
    """Restatement of skimage.transform.resize as the reference calls it (mrcnn/utils.py:957-978:
    order=1, mode='constant', cval=0, no anti-aliasing) [3P, skimage <= 0.15]: half-pixel-centre
    bilinear warp, neighbours outside the image contribute `cval`, output clipped to the input range.
    Integer/bool inputs are scaled to [0,1] floats unless preserve_range (img_as_float)."""
    assert order == 1 and mode == 'constant' and not anti_aliasing
    img = np.asarray(image)
    if not preserve_range:
        if img.dtype == np.bool_:
            img = img.astype(np.float64)
        elif img.dtype == np.uint8:
            img = img.astype(np.float64) / 255.0
        elif img.dtype.kind in "iu":
            img = img.astype(np.float64) / float(np.iinfo(img.dtype).max)
    img = img.astype(np.float64)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W = img.shape[:2]
    oh, ow = int(output_shape[0]), int(output_shape[1])
    if oh == 0 or ow == 0:
        out = np.zeros((oh, ow, img.shape[2]))
        return out[:, :, 0] if squeeze else out
    ry = (np.arange(oh) + 0.5) * (H / float(oh)) - 0.5
    rx = (np.arange(ow) + 0.5) * (W / float(ow)) - 0.5
    y0 = np.floor(ry).astype(np.int64); x0 = np.floor(rx).astype(np.int64)
    fy = (ry - y0)[:, None, None]; fx = (rx - x0)[None, :, None]
    pad = np.full((H + 2, W + 2, img.shape[2]), float(cval))
    pad[1:-1, 1:-1] = img
    yi0 = np.clip(y0 + 1, 0, H + 1); yi1 = np.clip(y0 + 2, 0, H + 1)
    xi0 = np.clip(x0 + 1, 0, W + 1); xi1 = np.clip(x0 + 2, 0, W + 1)
    tl = pad[yi0][:, xi0]; tr = pad[yi0][:, xi1]; bl = pad[yi1][:, xi0]; br = pad[yi1][:, xi1]
    out = (tl * (1 - fx) + tr * fx) * (1 - fy) + (bl * (1 - fx) + br * fx) * fy
    if clip:
        out = np.clip(out, img.min() if img.size else 0, img.max() if img.size else 0)
    return out[:, :, 0] if squeeze else out


def resize_plan(shape, min_dim=None, max_dim=None, min_scale=None, mode="square"):
    """The scalar half of resize_image (mrcnn/utils.py:456-561) for the modes without randomness: scale, the scaled size
    (round(h * scale), round(w * scale)), padding [(top, bottom), (left, right), (0, 0)] and window (y1, x1, y2, x2) of an image
    of `shape`.  The device path of MaskRCNN.mold_inputs does the pixel work from these numbers (ops.mold_image_u8)."""
    h, w = shape[:2]
    scale = 1
    if mode == "none":
        return 1, (h, w), [(0, 0), (0, 0), (0, 0)], (0, 0, h, w)
    if min_dim:
        scale = max(1, min_dim / min(h, w))
    if min_scale and scale < min_scale:
        scale = min_scale
    if max_dim and mode == "square":
        longest = max(h, w)
        if round(longest * scale) > max_dim:
            scale = max_dim / longest
    if scale != 1:
        h, w = round(h * scale), round(w * scale)
    if mode == "square":
        top, left = (max_dim - h) // 2, (max_dim - w) // 2
        return scale, (h, w), [(top, max_dim - h - top), (left, max_dim - w - left), (0, 0)], (top, left, h + top, w + left)
    if mode == "pad64":
        assert min_dim % 64 == 0, "Minimum dimension must be a multiple of 64"
        pads = []
        for size in (h, w):
            if size % 64 > 0:
                full = size - (size % 64) + 64
                before = (full - size) // 2
                pads.append((before, full - size - before))
            else:
                pads.append((0, 0))
        return scale, (h, w), [pads[0], pads[1], (0, 0)], (pads[0][0], pads[1][0], h + pads[0][0], w + pads[1][0])
    raise Exception("Mode {} not supported".format(mode))


def resize_image(image, min_dim=None, max_dim=None, min_scale=None, mode="square"):
    """Scale (up only) and zero-pad to the network input (mrcnn/utils.py:456-561).
    Returns image, window (y1,x1,y2,x2), scale, padding, crop."""
    dtype = image.dtype
    h, w = image.shape[:2]
    window, scale, padding, crop = (0, 0, h, w), 1, [(0, 0), (0, 0), (0, 0)], None
    if mode == "none":
        return image, window, scale, padding, crop
    if min_dim:
        scale = max(1, min_dim / min(h, w))
    if min_scale and scale < min_scale:
        scale = min_scale
    if max_dim and mode == "square":
        longest = max(h, w)
        if round(longest * scale) > max_dim:
            scale = max_dim / longest
    if scale != 1:
        image = resize(image, (round(h * scale), round(w * scale)), preserve_range=True)
    if mode == "square":
        h, w = image.shape[:2]
        top = (max_dim - h) // 2
        left = (max_dim - w) // 2
        padding = [(top, max_dim - h - top), (left, max_dim - w - left), (0, 0)]
        image = np.pad(image, padding, mode='constant', constant_values=0)
        window = (top, left, h + top, w + left)
    elif mode == "pad64":
        h, w = image.shape[:2]
        assert min_dim % 64 == 0, "Minimum dimension must be a multiple of 64"
        pads = []
        for size in (h, w):
            if size % 64 > 0:
                full = size - (size % 64) + 64
                before = (full - size) // 2
                pads.append((before, full - size - before))
            else:
                pads.append((0, 0))
        padding = [pads[0], pads[1], (0, 0)]
        image = np.pad(image, padding, mode='constant', constant_values=0)
        window = (pads[0][0], pads[1][0], h + pads[0][0], w + pads[1][0])
    elif mode == "crop":
        import random
        h, w = image.shape[:2]
        y = random.randint(0, (h - min_dim))
        x = random.randint(0, (w - min_dim))
        crop = (y, x, min_dim, min_dim)
        image = image[y:y + min_dim, x:x + min_dim]
        window = (0, 0, min_dim, min_dim)
    else:
        raise Exception("Mode {} not supported".format(mode))
    return image.astype(dtype), window, scale, padding, crop


def resize_mask(mask, scale, padding, crop=None):
    """Nearest-neighbour zoom of [H,W,N] masks by `scale` + the image's padding/crop
    (mrcnn/utils.py:564-583; scipy.ndimage.zoom(order=0))."""
    if scale != 1:                  # zoom by 1 with order 0 is the identity: the 256-pixel tiles of run.py skip the call
        import scipy.ndimage
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mask = scipy.ndimage.zoom(mask, zoom=[scale, scale, 1], order=0)
    if crop is not None:
        y, x, h, w = crop
        return mask[y:y + h, x:x + w]
    return np.pad(mask, padding, mode='constant', constant_values=0)


def unmold_mask(mask, bbox, image_shape):
    """28x28 float mask -> full-size boolean mask placed at bbox (mrcnn/utils.py:629-645)."""
    y1, x1, y2, x2 = bbox
    m = resize(mask, (y2 - y1, x2 - x1))
    full = np.zeros(image_shape[:2], dtype=bool)
    full[y1:y2, x1:x2] = m >= 0.5
    return full


def minimize_mask(bbox, mask, mini_shape):
    mini = np.zeros(tuple(mini_shape) + (mask.shape[-1],), dtype=bool)
    for i in range(mask.shape[-1]):
        y1, x1, y2, x2 = bbox[i][:4]
        m = mask[:, :, i].astype(bool)[y1:y2, x1:x2]
        if m.size == 0:
            raise Exception("Invalid bounding box with area of zero")
        mini[:, :, i] = np.around(resize(m, mini_shape)).astype(bool)
    return mini


def expand_mask(bbox, mini_mask, image_shape):
    mask = np.zeros(tuple(image_shape[:2]) + (mini_mask.shape[-1],), dtype=bool)
    for i in range(mask.shape[-1]):
        y1, x1, y2, x2 = bbox[i][:4]
        mask[y1:y2, x1:x2, i] = np.around(resize(mini_mask[:, :, i], (y2 - y1, x2 - x1))).astype(bool)
    return mask


# ---- image meta (mrcnn/model.py:2891-2969) ---------------------------------------------------------
def compose_image_meta(image_id, original_image_shape, image_shape, window, scale, active_class_ids):
    return np.array([image_id] + list(original_image_shape) + list(image_shape) + list(window) + [scale] +
                    list(active_class_ids))


def parse_image_meta(meta):
    return {"image_id": meta[:, 0].astype(np.int32), "original_image_shape": meta[:, 1:4].astype(np.int32),
            "image_shape": meta[:, 4:7].astype(np.int32), "window": meta[:, 7:11].astype(np.int32),
            "scale": meta[:, 11].astype(np.float32), "active_class_ids": meta[:, 12:].astype(np.int32)}


def mold_image(images, config):
    return images.astype(np.float32) - config.MEAN_PIXEL


def unmold_image(normalized_images, config):
    return (normalized_images + config.MEAN_PIXEL).astype(np.uint8)


# ---- evaluation helpers used by the test driver (mrcnn/utils.py:725-862) ---------------------------
def compute_overlaps_masks(masks1, masks2):
    if masks1.shape[-1] == 0 or masks2.shape[-1] == 0:
        return np.zeros((masks1.shape[-1], masks2.shape[-1]))
    m1 = np.reshape(masks1 > .5, (-1, masks1.shape[-1])).astype(np.float32)
    m2 = np.reshape(masks2 > .5, (-1, masks2.shape[-1])).astype(np.float32)
    inter = np.dot(m1.T, m2)
    union = np.sum(m1, axis=0)[:, None] + np.sum(m2, axis=0)[None, :] - inter
    return inter / union


def compute_recall(pred_boxes, gt_boxes, iou):
    ov = compute_overlaps(pred_boxes, gt_boxes)
    iou_max, iou_arg = np.max(ov, axis=1), np.argmax(ov, axis=1)
    positive_ids = np.where(iou_max >= iou)[0]
    return len(set(iou_arg[positive_ids])) / gt_boxes.shape[0], positive_ids


def compute_matches(gt_boxes, gt_class_ids, gt_masks, pred_boxes, pred_class_ids, pred_scores, pred_masks,
                    iou_threshold=0.5, score_threshold=0.0):
    """Greedy one-to-one matching of predictions (best score first) to ground truth by MASK IoU
    (mrcnn/utils.py:725-785).  A prediction takes the unmatched GT instance of highest IoU -- candidates are walked
    in descending IoU, the walk ends at the first IoU below the threshold -- provided the class ids agree; on a class
    mismatch the walk goes on to the next ground-truth instance (as the reference does).  Returns (gt_match [G], pred_match [N] as float arrays of indices or -1
    in score order, overlaps [N, G])."""
    keep_g = np.any(gt_boxes != 0, axis=1)
    n_gt = int(keep_g.sum())
    gt_masks = gt_masks[..., :n_gt]
    n_pred = int(np.any(pred_boxes != 0, axis=1).sum())
    order = np.argsort(pred_scores[:n_pred])[::-1]
    pred_class_ids = pred_class_ids[order]
    overlaps = compute_overlaps_masks(pred_masks[..., order], gt_masks)
    gt_match = np.full(n_gt, -1.0)
    pred_match = np.full(n_pred, -1.0)
    for i in range(n_pred):
        ranked = np.argsort(overlaps[i])[::-1]
        below = np.flatnonzero(overlaps[i, ranked] < score_threshold)
        if below.size:
            ranked = ranked[:below[0]]
        for j in ranked:
            if gt_match[j] >= 0:
                continue
            if overlaps[i, j] < iou_threshold:
                break
            if pred_class_ids[i] == gt_class_ids[j]:
                gt_match[j], pred_match[i] = i, j
                break
    return gt_match, pred_match, overlaps


def compute_ap(gt_boxes, gt_class_ids, gt_masks, pred_boxes, pred_class_ids, pred_scores, pred_masks, iou_threshold=0.5):
    """VOC-style average precision at one IoU threshold (mrcnn/utils.py:788-828): precision / recall after every
    prediction in score order, precision made non-increasing from the right, area under the stepped curve."""
    gt_match, pred_match, overlaps = compute_matches(gt_boxes, gt_class_ids, gt_masks, pred_boxes, pred_class_ids,
                                                     pred_scores, pred_masks, iou_threshold)
    hits = np.cumsum(pred_match > -1)
    precisions = np.concatenate([[0], hits / (np.arange(len(pred_match)) + 1), [0]])
    recalls = np.concatenate([[0], hits.astype(np.float32) / len(gt_match), [1]])
    precisions = np.maximum.accumulate(precisions[::-1])[::-1]
    steps = np.flatnonzero(recalls[:-1] != recalls[1:]) + 1
    mAP = np.sum((recalls[steps] - recalls[steps - 1]) * precisions[steps])
    return mAP, precisions, recalls, overlaps
