"""CPU input pipeline of training (mrcnn/model.py:1277-1377, 1536-1644, 1721-1904): per-image ground
truth loading, RPN target building and batch assembly.  NumPy on the host, exactly where the reference
runs it (Keras generator workers); the device step consumes the seven arrays it yields.
"""
import logging

import numpy as np

from . import utils

logger = logging.getLogger("mrcnn")


class EmptyShareError(RuntimeError):
    """A rank's share of the dataset can never yield a batch (fewer images than ranks, or none with instances)."""


def load_image_gt(dataset, config, image_id, augment=False, augmentation=None, use_mini_mask=False):
    """image, image_meta, class_ids, bbox, mask for one dataset entry (model.py:1277-1377).
    `augmentation` may be an imgaug augmenter (if imgaug is installed) or any callable
    ``f(image, mask) -> (image, mask)`` that preserves shapes."""
    image = dataset.load_image(image_id)
    mask, class_ids = dataset.load_mask(image_id)
    original_shape = image.shape
    image, window, scale, padding, crop = utils.resize_image(
        image, min_dim=config.IMAGE_MIN_DIM, min_scale=config.IMAGE_MIN_SCALE, max_dim=config.IMAGE_MAX_DIM,
        mode=config.IMAGE_RESIZE_MODE)
    mask = utils.resize_mask(mask, scale, padding, crop)
    if augment:
        logging.warning("'augment' is deprecated. Use 'augmentation' instead.")
        import random
        if random.randint(0, 1):
            image, mask = np.fliplr(image), np.fliplr(mask)
    if augmentation is not None:
        shape_i, shape_m = image.shape, mask.shape
        if hasattr(augmentation, "to_deterministic"):
            import imgaug
            safe = ["Sequential", "SomeOf", "OneOf", "Sometimes", "Fliplr", "Flipud", "CropAndPad", "Affine",
                    "PiecewiseAffine"]
            det = augmentation.to_deterministic()
            image = det.augment_image(image)
            mask = det.augment_image(mask.astype(np.uint8), hooks=imgaug.HooksImages(
                activator=lambda images, augmenter, parents, default: augmenter.__class__.__name__ in safe))
        else:
            image, mask = augmentation(image, mask)
        assert image.shape == shape_i, "Augmentation shouldn't change image size"
        assert mask.shape == shape_m, "Augmentation shouldn't change mask size"
        mask = mask.astype(bool)
    keep = np.sum(mask, axis=(0, 1)) > 0
    mask = mask[:, :, keep]
    class_ids = class_ids[keep]
    bbox = utils.extract_bboxes(mask)
    active_class_ids = np.zeros([dataset.num_classes], dtype=np.int32)
    source_class_ids = dataset.source_class_ids[dataset.image_info[image_id]["source"]]
    active_class_ids[source_class_ids] = 1
    if use_mini_mask:
        mask = utils.minimize_mask(bbox, mask, config.MINI_MASK_SHAPE)
    image_meta = utils.compose_image_meta(image_id, original_shape, image.shape, window, scale, active_class_ids)
    return image, image_meta, class_ids, bbox, mask


def build_rpn_targets(image_shape, anchors, gt_class_ids, gt_boxes, config, rng=np.random):
    """rpn_match [A] in {-1,0,1} and rpn_bbox [RPN_TRAIN_ANCHORS_PER_IMAGE, 4] (model.py:1536-1644)."""
    n_train = config.RPN_TRAIN_ANCHORS_PER_IMAGE
    rpn_match = np.zeros([anchors.shape[0]], dtype=np.int32)
    rpn_bbox = np.zeros((n_train, 4))
    crowd_ix = np.where(gt_class_ids < 0)[0]
    if crowd_ix.shape[0] > 0:
        non_crowd_ix = np.where(gt_class_ids > 0)[0]
        crowd_boxes = gt_boxes[crowd_ix]
        gt_class_ids, gt_boxes = gt_class_ids[non_crowd_ix], gt_boxes[non_crowd_ix]
        no_crowd = np.amax(utils.compute_overlaps(anchors, crowd_boxes), axis=1) < 0.001
    else:
        no_crowd = np.ones([anchors.shape[0]], dtype=bool)
    overlaps = utils.compute_overlaps(anchors, gt_boxes)
    arg = np.argmax(overlaps, axis=1)
    best = overlaps[np.arange(overlaps.shape[0]), arg]
    rpn_match[(best < 0.3) & no_crowd] = -1
    rpn_match[np.argwhere(overlaps == np.max(overlaps, axis=0))[:, 0]] = 1      # every GT keeps its best anchors
    rpn_match[best >= 0.7] = 1
    ids = np.where(rpn_match == 1)[0]
    extra = len(ids) - (n_train // 2)
    if extra > 0:
        rpn_match[rng.choice(ids, extra, replace=False)] = 0
    ids = np.where(rpn_match == -1)[0]
    extra = len(ids) - (n_train - np.sum(rpn_match == 1))
    if extra > 0:
        rpn_match[rng.choice(ids, extra, replace=False)] = 0
    ids = np.where(rpn_match == 1)[0]
    for ix, (i, a) in enumerate(zip(ids, anchors[ids])):
        gt = gt_boxes[arg[i]]
        gh, gw = gt[2] - gt[0], gt[3] - gt[1]
        gcy, gcx = gt[0] + 0.5 * gh, gt[1] + 0.5 * gw
        ah, aw = a[2] - a[0], a[3] - a[1]
        acy, acx = a[0] + 0.5 * ah, a[1] + 0.5 * aw
        rpn_bbox[ix] = [(gcy - acy) / ah, (gcx - acx) / aw, np.log(gh / ah), np.log(gw / aw)]
        rpn_bbox[ix] /= config.RPN_BBOX_STD_DEV
    return rpn_match, rpn_bbox


def data_generator(dataset, config, shuffle=True, augment=False, augmentation=None, batch_size=1,
                   no_augmentation_sources=None, rank=0, world_size=1, seed=None, device_targets=False):
    """Yields ([images, image_meta, rpn_match, rpn_bbox, gt_class_ids, gt_boxes, gt_masks], []) forever
    (model.py:1721-1904).  With world_size > 1 each rank walks its own stride of the (identically
    shuffled) image list -- the data-parallel replacement of tf.split in parallel_model.py:60-62."""
    b = 0
    image_index = -1
    image_ids = np.copy(dataset.image_ids)
    error_count = 0
    no_augmentation_sources = no_augmentation_sources or []
    rng = np.random.RandomState(seed) if seed is not None else np.random
    backbone_shapes = utils.compute_backbone_shapes(config, config.IMAGE_SHAPE)
    anchors = utils.generate_pyramid_anchors(config.RPN_ANCHOR_SCALES, config.RPN_ANCHOR_RATIOS, backbone_shapes,
                                             config.BACKBONE_STRIDES, config.RPN_ANCHOR_STRIDE)
    if len(image_ids) < world_size:
        raise EmptyShareError("data_generator: %d images for %d ranks -- rank %d would never receive one (and the other "
                         "ranks would hang in the gradient all-reduce)" % (len(image_ids), world_size, rank))
    barren = 0                                      # consecutive own-stride images without a usable instance
    while True:
        try:
            image_index = (image_index + 1) % len(image_ids)
            if shuffle and image_index == 0:
                rng.shuffle(image_ids)
            if image_index % world_size != rank:
                continue
            barren += 1
            if barren > 2 * len(image_ids):
                raise EmptyShareError("data_generator: rank %d found no image with instances in two passes over its share of "
                                 "the %d images" % (rank, len(image_ids)))
            image_id = image_ids[image_index]
            if dataset.image_info[image_id]['source'] in no_augmentation_sources:
                image, image_meta, gt_class_ids, gt_boxes, gt_masks = load_image_gt(
                    dataset, config, image_id, augment=augment, augmentation=None, use_mini_mask=config.USE_MINI_MASK)
            else:
                image, image_meta, gt_class_ids, gt_boxes, gt_masks = load_image_gt(
                    dataset, config, image_id, augment=augment, augmentation=augmentation,
                    use_mini_mask=config.USE_MINI_MASK)
            if not np.any(gt_class_ids > 0):
                continue
            barren = 0
            if not device_targets:
                rpn_match, rpn_bbox = build_rpn_targets(image.shape, anchors, gt_class_ids, gt_boxes, config)
            if b == 0:
                batch_image_meta = np.zeros((batch_size,) + image_meta.shape, dtype=image_meta.dtype)
                if not device_targets:
                    batch_rpn_match = np.zeros([batch_size, anchors.shape[0], 1], dtype=rpn_match.dtype)
                    batch_rpn_bbox = np.zeros([batch_size, config.RPN_TRAIN_ANCHORS_PER_IMAGE, 4], dtype=rpn_bbox.dtype)
                else:
                    batch_rpn_match = batch_rpn_bbox = None
                batch_images = np.zeros((batch_size,) + image.shape, dtype=np.float32)
                batch_gt_class_ids = np.zeros((batch_size, config.MAX_GT_INSTANCES), dtype=np.int32)
                batch_gt_boxes = np.zeros((batch_size, config.MAX_GT_INSTANCES, 4), dtype=np.int32)
                # the reference's batch carries MAX_GT_INSTANCES mask planes per image ([B, 256, 256, 300] bool = 78.6 MB on the run.py
                # path, almost all padding): allocating, zeroing and freeing that array cost the consumer ~3 ms per step.  With
                # device_targets (the product's own loop) the batch keeps only as many planes as its fullest image has (rounded up to
                # 8: they cross PCIe bit-packed); the padding up to MAX_GT_INSTANCES is written on the device (MaskRCNN._to_device).
                batch_gt_masks = None if device_targets else \
                    np.zeros((batch_size, gt_masks.shape[0], gt_masks.shape[1], config.MAX_GT_INSTANCES), dtype=gt_masks.dtype)
                mask_list = []
            if gt_boxes.shape[0] > config.MAX_GT_INSTANCES:
                ids = np.random.choice(np.arange(gt_boxes.shape[0]), config.MAX_GT_INSTANCES, replace=False)
                gt_class_ids, gt_boxes, gt_masks = gt_class_ids[ids], gt_boxes[ids], gt_masks[:, :, ids]
            batch_image_meta[b] = image_meta
            if not device_targets:
                batch_rpn_match[b] = rpn_match[:, np.newaxis]
                batch_rpn_bbox[b] = rpn_bbox
            batch_images[b] = utils.mold_image(image.astype(np.float32), config)
            batch_gt_class_ids[b, :gt_class_ids.shape[0]] = gt_class_ids
            batch_gt_boxes[b, :gt_boxes.shape[0]] = gt_boxes
            if device_targets:
                mask_list.append(gt_masks)
            else:
                batch_gt_masks[b, :, :, :gt_masks.shape[-1]] = gt_masks
            b += 1
            if b >= batch_size:
                if device_targets:
                    planes = max(8, (max(m.shape[-1] for m in mask_list) + 7) // 8 * 8)
                    batch_gt_masks = np.zeros((batch_size,) + mask_list[0].shape[:2] + (planes,), dtype=mask_list[0].dtype)
                    for k, m in enumerate(mask_list):
                        batch_gt_masks[k, :, :, :m.shape[-1]] = m
                yield [batch_images, batch_image_meta, batch_rpn_match, batch_rpn_bbox, batch_gt_class_ids,
                       batch_gt_boxes, batch_gt_masks], []
                b = 0
        except (GeneratorExit, KeyboardInterrupt, EmptyShareError):
            raise
        except Exception:
            logger.exception("Error processing image {}".format(dataset.image_info[image_id]))
            error_count += 1
            if error_count > 5:
                raise


class Prefetcher(object):
    """Background producer threads in front of one or more batch generators (the reference hands its generator
    to Keras ``fit_generator(workers=n, max_queue_size=100)``, model.py:2497-2510).  Worker k advances its own
    generator and the batches are handed out in arrival order; with a single worker the order is the generator's.
    Threads, not processes: the loaders are NumPy / file I/O (which release the GIL for the heavy parts) and a
    process that has initialised the GPU must neither fork-and-exec nor share its HIP context with children."""

    _END = object()

    def __init__(self, generators, depth=8):
        import queue
        import threading
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._threads = []
        self._alive = len(generators)
        self._lock = threading.Lock()
        for g in generators:
            t = threading.Thread(target=self._run, args=(g,), daemon=True)
            t.start()
            self._threads.append(t)

    def _put(self, item):
        import queue
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self, gen):
        try:
            for item in gen:
                if not self._put(item):
                    return
        except BaseException as e:                   # handed to the consumer, raised by __next__
            self._put(e)
            return
        finally:
            with self._lock:
                self._alive -= 1
                last = self._alive == 0
            if last:
                self._put(self._END)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is self._END:
            self._q.put(item)
            raise StopIteration
        if isinstance(item, BaseException):
            self.close()
            raise item
        return item

    def close(self):
        self._stop.set()
        for t in self._threads:
            t.join(timeout=2.0)
