"""``MaskRCNN(mode, config, model_dir)`` -- the drop-in boundary of the reference
(mrcnn/model.py:1917-2784; members used by run.py / analyze.py / sfinder.py: SURVEY.md section 8b).

Same constructor, ``train`` / ``detect`` / ``detect_molded`` / ``load_weights`` / ``find_last`` /
``get_anchors`` / ``mold_inputs`` / ``unmold_detections`` / ``set_trainable`` / ``compile`` signatures,
same result dictionaries and the same assertion behaviour; the Keras graph behind it is replaced by
``engine.MaskRCNNEngine`` (HIP kernels on the local MI355X).  Multi-GPU is one process per GPU
(torchrun), not in-graph towers: ``config.GPU_COUNT`` is the world size.
"""
import datetime
import logging
import os
import re

import numpy as np

from . import utils
from .datagen import data_generator

logger = logging.getLogger("mrcnn")


def log(text, array=None):
    if array is not None:
        text = text.ljust(25) + "shape: {:20}  ".format(str(array.shape))
        if array.size:
            text += "min: {:10.5f}  max: {:10.5f}".format(array.min(), array.max())
        else:
            text += "min: {:10}  max: {:10}".format("", "")
        text += "  {}".format(array.dtype)
    print(text)


class _GraphHandle(object):
    """Stands where callers expect ``model.keras_model``: predict() on molded inputs and summary()."""

    def __init__(self, owner):
        self._o = owner

    def predict(self, inputs, verbose=0):
        molded_images, image_metas, anchors = inputs
        return self._o._predict_molded(molded_images, image_metas)

    def summary(self):
        lines = ["%-28s %-8s %s" % (l.name, l.kind, l.shape) for l in self._o.engine.layout.layers]
        return "\n".join(lines + ["total trainable parameters: %d" % sum(s[2] for s in self._o.engine.layout.segments)])


class MaskRCNN(object):
    def __init__(self, mode, config, model_dir, device=None, weights=None, seed=0):
        assert mode in ['training', 'inference']
        self.mode = mode
        self.config = config
        self.model_dir = model_dir
        self.set_log_dir()
        self.engine = self.build(mode=mode, config=config, device=device, weights=weights, seed=seed)
        self.keras_model = _GraphHandle(self)
        self._lr, self._momentum = config.LEARNING_RATE, config.LEARNING_MOMENTUM
        self.use_hip_graph = True       # detect(): replay the inference graph instead of ~450 eager launches
        # detect() returns `masks` as ordinary NumPy arrays copied out of a pinned staging buffer that is reused call after
        # call.  True hands out views of freshly pinned memory instead (no host copy: -0.5 ms per 256 x 256 x 100 result), for
        # callers that drop each result before asking for many more -- results that are KEPT then keep pinned memory
        self.detect_zero_copy = False
        self._mask_staging = {}

    def print_model(self):
        print(self.keras_model.summary())

    def build(self, mode, config, device=None, weights=None, seed=0):
        assert mode in ['training', 'inference']
        h, w = config.IMAGE_SHAPE[:2]
        if h / 2 ** 6 != int(h / 2 ** 6) or w / 2 ** 6 != int(w / 2 ** 6):
            raise Exception("Image size must be dividable by 2 at least 6 times "
                            "to avoid fractions when downscaling and upscaling."
                            "For example, use 256, 320, 384, 448, 512, ... etc. ")
        if getattr(config, "TRAIN_BN", False) is not False:
            # the reference passes training=config.TRAIN_BN to every BatchNorm (model.py:57-72): None / True would
            # normalise with batch statistics and update the moving averages; this engine implements the frozen form
            # only (run.py never changes the default False) -- refuse instead of silently training something else
            raise NotImplementedError("TRAIN_BN=%r: only frozen BatchNorm (TRAIN_BN=False, the run.py default) is "
                                      "implemented" % (config.TRAIN_BN,))
        import torch
        from . import _hip
        from .engine import MaskRCNNEngine
        _hip.lib()       # fail loudly if the kernel library is missing
        if device is None:
            if not torch.cuda.is_available():
                raise _hip.HipPathError("MaskRCNN needs an MI355X (torch.cuda.is_available() is False); "
                                        "there is no CPU execution path")
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
            torch.cuda.set_device(device)
        eng = MaskRCNNEngine(config, device, weights=weights, seed=seed)
        hd = getattr(config, "HEAD_DTYPE", None)
        if hd is not None:
            assert hd in ("float16", "bfloat16"), "HEAD_DTYPE must be None, 'float16' or 'bfloat16'"
            eng.head_dtype = getattr(torch, hd)
            eng.loss_scale = float(getattr(config, "HEAD_LOSS_SCALE", 4096.0))
        return eng

    # ---- checkpoints ------------------------------------------------------------------------------
    def find_last(self):
        """Last checkpoint of the last run under model_dir (mrcnn/model.py:2168-2195)."""
        import errno
        key = self.config.NAME.lower()
        dir_names = sorted(d for d in next(os.walk(self.model_dir))[1] if d.startswith(key))
        if not dir_names:
            raise FileNotFoundError(errno.ENOENT, "Could not find model directory under {}".format(self.model_dir))
        dir_name = os.path.join(self.model_dir, dir_names[-1])
        checkpoints = sorted(f for f in next(os.walk(dir_name))[2] if f.startswith("mask_rcnn"))
        if not checkpoints:
            raise FileNotFoundError(errno.ENOENT, "Could not find weight files in {}".format(dir_name))
        return os.path.join(dir_name, checkpoints[-1])

    def load_weights(self, filepath, by_name=False, exclude=None):
        """Keras-HDF5 (or .npz) weights by layer name (mrcnn/model.py:2197-2239).  ``exclude`` keeps
        the reference's quirk: callers pass a *string* ('conv1'), so it is a substring test on the
        layer name."""
        from . import weights_io
        if exclude:
            by_name = True
        tensors = weights_io.load(filepath)
        if exclude:
            tensors = {k: v for k, v in tensors.items() if k.split("/")[0] not in exclude}
        known = self.engine.layout.offsets
        stats = ("moving_mean", "moving_variance")
        picked = {k: v for k, v in tensors.items() if k in known or k.split("/")[-1] in stats}
        if not by_name and len(picked) != len(tensors):
            raise ValueError("weight file does not match the model topology; use by_name=True")
        self.engine.set_weights(picked, strict=False)
        # by-name loading skips what does not match, silently in Keras; a naming / layout mismatch would leave layers at
        # their random initialisation without a trace, so say what happened (and keep the lists for callers / tests)
        wanted = set(known) | {"%s/%s" % (l.bn, s_) for l in self.engine.layout.bn_layers for s_ in stats}
        self.last_load = {"loaded": sorted(set(picked) & wanted),
                          "missing_in_file": sorted(n for n in wanted if n not in picked
                                                    and not (exclude and n.split("/")[0] in exclude)),
                          "unused_in_file": sorted(n for n in tensors if n not in wanted)}
        log("load_weights(%s): %d tensors loaded, %d model tensors not in the file, %d file tensors not used" % (
            os.path.basename(str(filepath)), len(self.last_load["loaded"]), len(self.last_load["missing_in_file"]),
            len(self.last_load["unused_in_file"])))
        for key in ("missing_in_file", "unused_in_file"):
            if self.last_load[key]:
                log("  %s: %s%s" % (key, ", ".join(self.last_load[key][:6]), " ..." if len(self.last_load[key]) > 6 else ""))
        self.set_log_dir(filepath)

    def save_weights(self, filepath):
        from . import weights_io
        weights_io.save(filepath, self.engine.get_weights(), self.engine.layout)

    def get_imagenet_weights(self):
        raise RuntimeError("get_imagenet_weights downloads from the network (mrcnn/model.py:2241-2253); "
                           "not available offline -- pass a local weights file to load_weights()")

    # ---- training set-up ----------------------------------------------------------------------------
    def compile(self, learning_rate, momentum):
        """Optimiser set-up (mrcnn/model.py:2255-2318): SGD(lr, momentum, clipnorm=GRADIENT_CLIP_NORM),
        weighted losses per USE_LOSSES/LOSS_WEIGHTS and the L2/numel regulariser, all applied by
        ``engine.apply_gradients``."""
        self._lr, self._momentum = learning_rate, momentum
        for name in ("rpn_class_loss", "rpn_bbox_loss", "mrcnn_class_loss", "mrcnn_bbox_loss", "mrcnn_mask_loss"):
            if not self.config.USE_LOSSES.get(name, True):
                print('Will not include %s in total loss ...' % name)
        print("Optimizer SGD: lr=%f, momentum=%f, clipnorm=%f" % (learning_rate, momentum, self.config.GRADIENT_CLIP_NORM))
        print("weight_decay=%f" % (self.config.WEIGHT_DECAY))

    def set_trainable(self, layer_regex, keras_model=None, indent=0, verbose=1):
        if verbose > 0:
            log("Selecting layers to train")
        self.engine.set_trainable(layer_regex)
        if verbose > 0:
            seen = set()
            for (name, _, _, _, _), t in zip(self.engine.layout.segments, self.engine.trainable_host):
                base = name.split("/")[0]
                if t and base not in seen:
                    seen.add(base)
                    log("{}{:20}".format(" " * indent, base))

    def get_trainable_layers(self):
        return [l for l in self.engine.layout.layers]

    def set_log_dir(self, model_path=None):
        """Log directory + epoch counter from a checkpoint path (mrcnn/model.py:2357-2393)."""
        self.epoch = 0
        now = datetime.datetime.now()
        if model_path:
            regex = r".*[/\\][\w-]+(\d{4})(\d{2})(\d{2})T(\d{2})(\d{2})[/\\]mask\_rcnn\_[\w-]+(\d{4})\.(h5|npz)"
            m = re.match(regex, model_path)
            if m:
                now = datetime.datetime(int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)),
                                        int(m.group(5)))
                self.epoch = int(m.group(6)) - 1 + 1
                print('Re-starting from epoch %d' % self.epoch)
        name = (self.config.NAME or "mrcnn").lower()
        self.log_dir = os.path.join(self.model_dir, "{}{:%Y%m%dT%H%M}".format(name, now))
        self.checkpoint_path = os.path.join(self.log_dir, "mask_rcnn_{}_*epoch*.h5".format(name))
        self.checkpoint_path = self.checkpoint_path.replace("*epoch*", "{epoch:04d}")

    # ---- one optimisation step (device) -------------------------------------------------------------
    def _to_device(self, inputs, rand_keys=None, rpn_keys=None):
        """Generator batch -> device tensors of engine.forward_backward.  With rpn_match / rpn_bbox None the RPN
        targets are built on the GPU from the GT boxes (``rpn_keys`` [B, A] uniform floats, drawn here when not
        given).  Only the GT-mask planes of real instances cross PCIe, bit-packed (8 instances per byte); the unpacking and
        the MAX_GT_INSTANCES padding are written on the device."""
        import torch
        from . import ops
        dev = self.engine.dev
        images, image_meta, rpn_match, rpn_bbox, gt_class_ids, gt_boxes, gt_masks = inputs
        B, H, W = images.shape[0], images.shape[1], images.shape[2]
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev, non_blocking=True)
        # norm_boxes_graph (model.py:1971, 3003-3017), float32
        gtn = (gt_boxes.astype(np.float32) - np.array([0., 0., 1., 1.], np.float32)) / \
              (np.array([H, W, H, W], np.float32) - np.float32(1.0))
        active = np.asarray(image_meta)[0, 12:].astype(np.int32)
        if rand_keys is None:
            rand_keys = np.random.uniform(0, 1, (B, self.config.POST_NMS_ROIS_TRAINING))
        cls_d = t(gt_class_ids, np.int32)
        if rpn_match is None:
            self.get_anchors((H, W, 3))                                    # sets self.anchors (pixels, float64)
            if getattr(self, "_anchors_px_dev_key", None) != (H, W):
                self._anchors_px_dev = t(self.anchors, np.float64)
                self._anchors_px_dev_key = (H, W)
            A = self._anchors_px_dev.shape[0]
            keys_d = t(rpn_keys, np.float32) if rpn_keys is not None else torch.rand((B, A), device=dev)
            rpn_match_d, rpn_bbox_d = ops.rpn_targets(self._anchors_px_dev, cls_d, t(gt_boxes, np.int32), keys_d,
                                                      self.config.RPN_TRAIN_ANCHORS_PER_IMAGE,
                                                      self.config.RPN_BBOX_STD_DEV)
        else:
            rpn_match_d, rpn_bbox_d = t(rpn_match, np.int32), t(rpn_bbox, np.float32)
        G = np.asarray(gt_class_ids).shape[1]                             # planes on the device: MAX_GT_INSTANCES, like the class ids
        have = gt_masks.shape[-1]                                         # planes in the batch (data_generator(device_targets=True) trims the padding)
        used = np.flatnonzero(np.any(np.asarray(gt_class_ids) != 0, axis=0))
        n_used = min(int(used[-1]) + 1 if used.size else 0, have)
        if n_used < have and gt_masks[..., n_used:].any():
            n_used = min(have, G)                                         # masks without a class id: upload them all
        # bit-packed across PCIe (8 instances per byte), unpacked on the device; the padding planes are written there
        self.last_mask_h2d_bytes = 0
        if n_used:
            packed = np.packbits(np.asarray(gt_masks[..., :n_used]) != 0, axis=-1, bitorder="little")
            self.last_mask_h2d_bytes = packed.nbytes
            masks_d = ops.unpack_mask_bits(t(packed, np.uint8), n_used, G)
        else:
            masks_d = torch.zeros(tuple(gt_masks.shape[:-1]) + (G,), dtype=torch.uint8, device=dev)
        return (t(images, np.float32), rpn_match_d, rpn_bbox_d, cls_d, t(gtn, np.float32), masks_d,
                t(active, np.int32), t(rand_keys, np.float32))

    def train_on_batch(self, inputs, rand_keys=None, reducer=None, world_size=1, apply=True, keep_outputs=False):
        """Forward + backward + (all-reduce) + SGD step on one generator batch.  Returns the five
        losses (host array, this rank's batch means)."""
        eng = self.engine
        dev_inputs = self._to_device(inputs, rand_keys)
        eng.grad_ready = reducer.ready if reducer is not None else None
        if apply and reducer is None and world_size == 1 and not keep_outputs and getattr(self.config, "TRAIN_HIP_GRAPH", False):
            return eng.step_graphed(dev_inputs, self._lr, self._momentum)      # the whole step replayed from one HIP graph
        if apply and not keep_outputs and getattr(self.config, "TRAIN_LAUNCH_TAPE", False) and \
                ((reducer is None and world_size == 1) or (reducer is not None and not reducer.timing)):
            # the same launches on the same streams, re-issued from a recording without the engine's Python per launch: the device
            # time is unchanged, the main thread holds the interpreter lock for a fraction of the time -- which the loader threads
            # of MaskRCNN.train() need (feed-inclusive loop).  Under data parallelism the gradient hooks (hand-off
            # events, mrcnn_allreduce_grad, the join) are part of the recording.
            return eng.step_taped(dev_inputs, self._lr, self._momentum, world_size, reducer)
        losses = eng.forward_backward(*dev_inputs, keep_outputs=keep_outputs)
        if reducer is not None:
            reducer.finish()
        if apply:
            eng.apply_gradients(self._lr, self._momentum, world_size)
        return losses

    def train(self, train_dataset, val_dataset, learning_rate, epochs, layers, augmentation=None,
              custom_callbacks=None, no_augmentation_sources=None, n_worker_threads=-1, class_weights=None,
              draw_loss=False):
        """Training loop (mrcnn/model.py:2395-2517): ``epochs`` is the total target epoch, one weight
        file per epoch under log_dir, validation losses per epoch; returns None."""
        assert self.mode == "training", "Create model in training mode."
        import torch
        from .parallel import GradReducer, allreduce_mean_scalars, init_distributed
        rank, _, world = init_distributed()
        cfg = self.config
        # n_worker_threads loader threads (<= 0: one per host core, at most 8) prefetch batches while the GPU works
        # (fit_generator(workers=..., max_queue_size=100), model.py:2497-2510); worker 0 walks the data like the
        # unthreaded generator, further workers use their own shuffles
        from .datagen import Prefetcher
        nw = int(n_worker_threads) if n_worker_threads and n_worker_threads > 0 else min(8, os.cpu_count() or 1)
        dev_t = bool(getattr(cfg, "DEVICE_RPN_TARGETS", False))
        if getattr(cfg, "DEVICE_FITS", False):
            for ds in (train_dataset, val_dataset):
                if hasattr(ds, "load_image") and hasattr(ds, "zscale_contrasts"):     # SourceDataset: FITS -> uint8 RGB on the GPU
                    ds.device = self.engine.dev
        train_gen = Prefetcher([data_generator(train_dataset, cfg, shuffle=True, augmentation=augmentation,
                                               batch_size=cfg.IMAGES_PER_GPU, no_augmentation_sources=no_augmentation_sources,
                                               rank=rank, world_size=world, seed=1234 + 1000 * k, device_targets=dev_t)
                                for k in range(nw)], depth=2 * nw + 2)
        val_gen = Prefetcher([data_generator(val_dataset, cfg, shuffle=True, batch_size=cfg.IMAGES_PER_GPU, rank=rank,
                                             world_size=world, seed=4321, device_targets=dev_t)], depth=2)
        if rank == 0 and not os.path.exists(self.log_dir):
            os.makedirs(self.log_dir)
        log("\nStarting at epoch {}. LR={}\n".format(self.epoch, learning_rate))
        log("Checkpoint Path: {}".format(self.checkpoint_path))
        self.set_trainable(layers, verbose=1 if rank == 0 else 0)
        self.compile(learning_rate, cfg.LEARNING_MOMENTUM)
        reducer = GradReducer(self.engine.grads, world) if world > 1 else None
        names = ["rpn_class_loss", "rpn_bbox_loss", "mrcnn_class_loss", "mrcnn_bbox_loss", "mrcnn_mask_loss"]
        self.history = {"loss": [], "val_loss": []}
        for epoch in range(self.epoch, epochs):
            acc = torch.zeros(5, device=self.engine.dev)
            for step in range(cfg.STEPS_PER_EPOCH):
                inputs, _ = next(train_gen)
                acc += self.train_on_batch(inputs, reducer=reducer, world_size=world)
            tr = allreduce_mean_scalars(acc / cfg.STEPS_PER_EPOCH, world).cpu().numpy()
            skipped = self.engine.adapt_loss_scale()          # float16 mode: overflowed steps were skipped on the device
            if skipped and rank == 0:
                print("%d step(s) skipped (non-finite float16 gradients); loss scale now %g" % (skipped, self.engine.loss_scale))
            vacc = torch.zeros(5, device=self.engine.dev)
            for step in range(cfg.VALIDATION_STEPS):
                inputs, _ = next(val_gen)
                vacc += self.train_on_batch(inputs, reducer=None, world_size=1, apply=False)
            va = allreduce_mean_scalars(vacc / max(cfg.VALIDATION_STEPS, 1), world).cpu().numpy()
            w = np.array(self.engine.loss_weights())
            self.history["loss"].append(float((tr * w).sum()))
            self.history["val_loss"].append(float((va * w).sum()))
            if rank == 0:
                print("Epoch %d/%d - loss: %.4f - %s - val_loss: %.4f" % (
                    epoch + 1, epochs, self.history["loss"][-1],
                    " - ".join("%s: %.4f" % (n, v) for n, v in zip(names, tr)), self.history["val_loss"][-1]))
                self.save_weights(self.checkpoint_path.format(epoch=epoch + 1))
                for cb in (custom_callbacks or []):
                    if callable(cb):
                        cb(epoch, dict(zip(names, tr.tolist())))
        self.epoch = max(self.epoch, epochs)
        train_gen.close()
        val_gen.close()

    # ---- inference ------------------------------------------------------------------------------------
    def mold_inputs(self, images):
        """Resize/pad, subtract MEAN_PIXEL, build image_meta (mrcnn/model.py:2519-2556)."""
        molded_images, image_metas, windows = [], [], []
        for image in images:
            molded, window, scale, padding, crop = utils.resize_image(
                image, min_dim=self.config.IMAGE_MIN_DIM, min_scale=self.config.IMAGE_MIN_SCALE,
                max_dim=self.config.IMAGE_MAX_DIM, mode=self.config.IMAGE_RESIZE_MODE)
            molded = utils.mold_image(molded, self.config)
            meta = utils.compose_image_meta(0, image.shape, molded.shape, window, scale,
                                            np.zeros([self.config.NUM_CLASSES], dtype=np.int32))
            molded_images.append(molded)
            windows.append(window)
            image_metas.append(meta)
        return np.stack(molded_images), np.stack(image_metas), np.stack(windows)

    def _mold_inputs_device(self, images):
        """mold_inputs for uint8 images with the pixel work on the GPU (ops.mold_image_u8): each image crosses PCIe as its
        uint8 bytes and is scaled, padded and mean-subtracted there.  Returns the molded batch as a device tensor
        [B, H, W, C] float32, image_metas and windows as mold_inputs does -- or None when an image is not uint8 HxWxC or the
        resize mode draws random numbers ("crop"): the caller then takes the host path."""
        import torch
        from . import ops
        cfg = self.config
        if cfg.IMAGE_RESIZE_MODE not in ("square", "pad64", "none"):
            return None
        plans = []
        for image in images:
            if not (isinstance(image, np.ndarray) and image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] <= 4):
                return None
            scale, (oh, ow), padding, window = utils.resize_plan(image.shape, cfg.IMAGE_MIN_DIM, cfg.IMAGE_MAX_DIM, cfg.IMAGE_MIN_SCALE,
                                                                 cfg.IMAGE_RESIZE_MODE)
            canvas = (oh + padding[0][0] + padding[0][1], ow + padding[1][0] + padding[1][1])
            plans.append((scale, (oh, ow), padding, window, canvas))
        if any(p[4] != plans[0][4] for p in plans) or any(im.shape[2] != images[0].shape[2] for im in images):
            return None                                              # the host path raises the reference's assertion
        OH, OW = plans[0][4]
        C_ = images[0].shape[2]
        dev = self.engine.dev
        batch = torch.empty((len(images), OH, OW, C_), dtype=torch.float32, device=dev)
        metas, windows = [], []
        for i, (image, (scale, out_hw, padding, window, canvas)) in enumerate(zip(images, plans)):
            src = torch.from_numpy(np.ascontiguousarray(image)).to(dev, non_blocking=True)
            ops.mold_image_u8(src, out_hw, (padding[0][0], padding[1][0]), canvas, cfg.MEAN_PIXEL, out=batch[i])
            metas.append(utils.compose_image_meta(0, image.shape, (OH, OW, C_), window, scale,
                                                  np.zeros([cfg.NUM_CLASSES], dtype=np.int32)))
            windows.append(window)
        return batch, np.stack(metas), np.stack(windows)

    def _unmold_boxes(self, detections, original_image_shape, image_shape, window):
        """Host half of unmold_detections (mrcnn/model.py:2578-2605): the rows before the first class id 0, boxes from
        normalised window coordinates to pixels of the original image, zero-area boxes dropped.  Returns boxes [n,4] int32,
        class_ids [n] int32, scores [n] float32 and the rows of `detections` they came from."""
        zero_ix = np.where(detections[:, 4] == 0)[0]
        N = zero_ix[0] if zero_ix.shape[0] > 0 else detections.shape[0]
        boxes = detections[:N, :4]
        class_ids = detections[:N, 4].astype(np.int32)
        scores = detections[:N, 5]
        rows = np.arange(N, dtype=np.int32)
        window = utils.norm_boxes(window, image_shape[:2])
        wy1, wx1, wy2, wx2 = window
        shift = np.array([wy1, wx1, wy1, wx1])
        wh, ww = wy2 - wy1, wx2 - wx1
        scale = np.array([wh, ww, wh, ww])
        boxes = np.divide(boxes - shift, scale)
        boxes = utils.denorm_boxes(boxes, original_image_shape[:2])
        keep = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) > 0
        if not keep.all():
            boxes, class_ids, scores, rows = boxes[keep], class_ids[keep], scores[keep], rows[keep]
        return boxes, class_ids, scores, rows

    def _unmold_masks_device(self, boxes, class_ids, rows, mrcnn_mask_dev, original_image_shape, slot=0):
        """Device half (mrcnn/model.py:2607-2619): every detection's class mask resized to its box, thresholded and pasted
        (ops.unmold_masks); returns the pinned host tensor the [H, W, n] uint8 result is being copied into -- valid after the
        next synchronisation of the current stream."""
        import torch
        from . import ops
        H, W = int(original_image_shape[0]), int(original_image_shape[1])
        n = boxes.shape[0]
        C_ = mrcnn_mask_dev.shape[-1]
        if n and (class_ids.min() < 0 or class_ids.max() >= C_ or (boxes[:, 0] < 0).any() or (boxes[:, 1] < 0).any() or
                  (boxes[:, 2] > H).any() or (boxes[:, 3] > W).any()):
            # the reference's paste `full_mask[y1:y2, x1:x2] = mask` raises on such a box too (shape mismatch)
            raise ValueError("detection box outside the %dx%d image or class id outside [0, %d)" % (H, W, C_))
        if self.detect_zero_copy:
            host = torch.empty((H, W, n), dtype=torch.uint8, pin_memory=True)
        else:                                                      # one pinned buffer per slot of the batch, grown as needed, reused
            buf = self._mask_staging.get(slot)
            if buf is None or buf.numel() < H * W * n:
                buf = self._mask_staging[slot] = torch.empty(max(H * W * n, 1), dtype=torch.uint8, pin_memory=True)
            host = buf[:H * W * n].view(H, W, n)
        if n:
            dets = np.empty((n, 6), np.int32)
            dets[:, :4], dets[:, 4], dets[:, 5] = boxes, class_ids, rows
            d = torch.from_numpy(dets).to(mrcnn_mask_dev.device, non_blocking=True)
            host.copy_(ops.unmold_masks(mrcnn_mask_dev, d, (H, W)), non_blocking=True)
        return host

    def unmold_detections(self, detections, mrcnn_mask, original_image_shape, image_shape, window):
        """Network output -> boxes/class ids/scores/full-size masks of one image (mrcnn/model.py:2558-2621).  `mrcnn_mask`
        [N, mh, mw, num_classes] may be a host array (uploaded) or a device tensor; the masks are resized and pasted on the
        GPU (utils.unmold_mask is the host statement of the same arithmetic, tests/test_kernels_gpu.py::test_unmold_masks)."""
        import torch
        detections = np.asarray(detections)
        boxes, class_ids, scores, rows = self._unmold_boxes(detections, original_image_shape, image_shape, window)
        if not torch.is_tensor(mrcnn_mask):
            mrcnn_mask = torch.from_numpy(np.ascontiguousarray(mrcnn_mask, dtype=np.float32))
        mrcnn_mask = mrcnn_mask.to(self.engine.dev).contiguous()
        if boxes.shape[0] == 0:
            return boxes, class_ids, scores, np.empty(tuple(original_image_shape[:2]) + (0,))
        host = self._unmold_masks_device(boxes, class_ids, rows, mrcnn_mask, original_image_shape)
        torch.cuda.current_stream(self.engine.dev).synchronize()
        return boxes, class_ids, scores, self._masks_out(host)

    def _masks_out(self, host):
        """The [H, W, n] uint8 planes in pinned memory as the bool array the caller gets: a copy (default; the staging buffer is
        reused by the next call) or, with detect_zero_copy, a view that keeps its own pinned block alive."""
        m = host.numpy().view(np.bool_)
        return m if self.detect_zero_copy else m.copy()

    def _run_graph(self, molded_images, image_metas):
        """The inference graph on molded inputs (model.py:2156-2159); outputs stay on the device."""
        import torch
        eng = self.engine
        x = molded_images if torch.is_tensor(molded_images) else torch.from_numpy(np.ascontiguousarray(molded_images, dtype=np.float32))
        meta = np.asarray(image_metas)
        shape = meta[0, 4:6].astype(np.float32)          # image_shape of the first image (model.py:891-893)
        win = (meta[:, 7:11].astype(np.float32) - np.array([0., 0., 1., 1.], np.float32)) / \
              (np.array([shape[0], shape[1], shape[0], shape[1]], np.float32) - np.float32(1.0))
        wt = torch.from_numpy(np.ascontiguousarray(win, dtype=np.float32))
        if self.use_hip_graph:
            return eng.infer_graphed(x, wt)
        return eng.infer(x.to(eng.dev), wt.to(eng.dev))

    def _predict_molded(self, molded_images, image_metas):
        """The seven outputs of the inference graph as host arrays (model.py:2156-2159)."""
        import torch
        out = self._run_graph(molded_images, image_metas)
        torch.cuda.synchronize(self.engine.dev)
        return [out[k].cpu().numpy() for k in ("detections", "mrcnn_class", "mrcnn_bbox", "mrcnn_mask", "rpn_rois",
                                               "rpn_class", "rpn_bbox")]

    def _detect_results(self, out, shapes, molded_shapes, windows, timing=None):
        """detections (host, 2.4 KB per image) -> box arithmetic on the host -> masks resized / pasted on the device -> one
        pinned D2H copy of the [H, W, n] planes per image.  mrcnn_mask never leaves the device."""
        import time
        import torch
        stream = torch.cuda.current_stream(self.engine.dev)
        detections = out["detections"].cpu().numpy()                       # synchronises: the graph is done here
        if timing is not None:
            timing["graph_done"] = time.perf_counter()
        pending = []
        for i in range(len(shapes)):
            boxes, class_ids, scores, rows = self._unmold_boxes(detections[i], shapes[i], molded_shapes[i], windows[i])
            host = self._unmold_masks_device(boxes, class_ids, rows, out["mrcnn_mask"][i], shapes[i], slot=i) if boxes.shape[0] else None
            pending.append((boxes, class_ids, scores, host))
        stream.synchronize()
        results = []
        for (boxes, class_ids, scores, host), shp in zip(pending, shapes):
            masks = self._masks_out(host) if host is not None else np.empty(tuple(shp[:2]) + (0,))
            results.append({"rois": boxes, "class_ids": class_ids, "scores": scores, "masks": masks})
        return results

    def detect(self, images, verbose=0, timing=None):
        """List of [h,w,3] images -> list of {rois, class_ids, scores, masks} (mrcnn/model.py:2623-2704).  `timing`: an
        optional dict that receives perf_counter() marks after the molding, the graph and the un-molding."""
        assert self.mode == "inference", "Create model in inference mode."
        assert len(images) == self.config.BATCH_SIZE, "len(images) must be equal to BATCH_SIZE"
        if verbose:
            log("Processing {} images".format(len(images)))
            for image in images:
                log("image", image)
        if timing is not None:
            import time
            timing["start"] = time.perf_counter()
        molded = self._mold_inputs_device(images)                  # uint8 images: scaled / padded / mean-subtracted on the GPU
        if molded is None:
            molded = self.mold_inputs(images)
        molded_images, image_metas, windows = molded
        image_shape = molded_images[0].shape
        for g in molded_images[1:]:
            assert g.shape == image_shape, \
                "After resizing, all images must have the same size. Check IMAGE_RESIZE_MODE and image sizes."
        if verbose:
            log("molded_images", molded_images if isinstance(molded_images, np.ndarray) else molded_images.cpu().numpy())
            log("image_metas", image_metas)
        if timing is not None:
            timing["molded"] = time.perf_counter()
        out = self._run_graph(molded_images, image_metas)
        results = self._detect_results(out, [im.shape for im in images], [tuple(m.shape) for m in molded_images], windows, timing)
        if timing is not None:
            timing["end"] = time.perf_counter()
        return results

    def detect_molded(self, molded_images, image_metas, verbose=0):
        """As detect() but on already molded inputs (mrcnn/model.py:2706-2762)."""
        assert self.mode == "inference", "Create model in inference mode."
        assert len(molded_images) == self.config.BATCH_SIZE, "Number of images must be equal to BATCH_SIZE"
        image_shape = molded_images[0].shape
        for g in molded_images[1:]:
            assert g.shape == image_shape, "Images must have the same size"
        out = self._run_graph(np.asarray(molded_images), image_metas)
        shapes = [im.shape for im in molded_images]
        return self._detect_results(out, shapes, shapes, [[0, 0, s[0], s[1]] for s in shapes])

    def get_anchors(self, image_shape):
        """Normalised anchor pyramid for an image shape, cached (mrcnn/model.py:2764-2784)."""
        if not hasattr(self, "_anchor_cache"):
            self._anchor_cache = {}
        key = tuple(image_shape)
        if key not in self._anchor_cache:
            shapes = utils.compute_backbone_shapes(self.config, image_shape)
            a = utils.generate_pyramid_anchors(self.config.RPN_ANCHOR_SCALES, self.config.RPN_ANCHOR_RATIOS, shapes,
                                               self.config.BACKBONE_STRIDES, self.config.RPN_ANCHOR_STRIDE)
            self.anchors = a
            self._anchor_cache[key] = utils.norm_boxes(a, image_shape[:2])
        return self._anchor_cache[key]
