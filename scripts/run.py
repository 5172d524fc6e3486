#!/usr/bin/env python3
"""Command line entry point with the reference's interface (scripts/run.py:1263-1384, 1448-1760):

    run.py train  --datalist train.dat [--datalist_val val.dat] --nepochs 10 --backbone resnet101 ...
    run.py test   --datalist test.dat  --weights w.h5 --scoreThr 0.7 --iouThr 0.6
    run.py detect --image map.fits --weights w.h5 [--xmin/--xmax/--ymin/--ymax] [--split_img_in_tiles ...]

Same flag names and defaults; the model behind them is caesar_mrcnn_amd.model.MaskRCNN (MI355X).
Multi-GPU training: ``torchrun --nproc-per-node NGPU scripts/run.py train --ngpu NGPU ...``.
`test` reports completeness / reliability per class from mask-IoU matching; `detect` runs the Analyzer
post-processing (caesar_mrcnn_amd/analyze.py: score filter, connected-mask merging, best-of-overlapping
selection) on every tile and writes the objects as JSON (`out_<image>.json`: image_id, objs[name, x1, x2,
y1, y2, class_id, class_name, score, pixels, vertexes, edge]) and DS9 regions.  Plots are not produced.
"""
import argparse
import json
import logging
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

logger = logging.getLogger("mrcnn")
DEFAULT_LOGS_DIR = os.path.join(os.getcwd(), "logs")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Train Mask R-CNN to detect radio sources.')
    p.add_argument("command", metavar="<command>", help="'train', 'test' or 'detect'")
    p.add_argument('--imgsize', type=int, default=256)
    p.add_argument('--grayimg', action='store_true', default=False)
    p.add_argument('--no_uint8', dest='to_uint8', action='store_false', default=True)
    p.add_argument('--no_zscale', dest='zscale', action='store_false', default=True)
    p.add_argument('--zscale_contrasts', type=str, default='0.25,0.25,0.25')
    p.add_argument('--biascontrast', action='store_true', default=False)
    p.add_argument('--bias', type=float, default=0.5)
    p.add_argument('--contrast', type=float, default=1.0)
    p.add_argument('--no_norm_img', dest='norm_img', action='store_false', default=True)
    p.add_argument('--classdict', type=str, default='{"sidelobe":1,"source":2,"galaxy":3}')
    p.add_argument('--classdict_model', type=str, default='')
    p.add_argument('--dataloader', type=str, default='datalist', help='{datalist,datalist_json,datadir_json}')
    p.add_argument('--datalist', default=None)
    p.add_argument('--datalist_train', default=None)
    p.add_argument('--datalist_val', default=None)
    p.add_argument('--datadir', default=None)
    p.add_argument('--validation_data_fract', type=float, default=0.1)
    p.add_argument('--maxnimgs', type=int, default=-1)
    p.add_argument('--weights', default=None, help="Path to weights .h5 (Keras) or .npz file")
    p.add_argument('--logs', default=DEFAULT_LOGS_DIR)
    p.add_argument('--nthreads', type=int, default=1)
    p.add_argument('--ngpu', type=int, default=1)
    p.add_argument('--nimg_per_gpu', type=int, default=1)
    p.add_argument('--nepochs', type=int, default=1)
    p.add_argument('--epoch_length', type=int, default=None)
    p.add_argument('--nvalidation_steps', type=int, default=None)
    p.add_argument('--rpn_anchor_scales', type=str, default='4,8,16,32,64')
    p.add_argument('--max_gt_instances', type=int, default=300)
    p.add_argument('--backbone', type=str, default='resnet101')
    p.add_argument('--backbone_strides', type=str, default='4,8,16,32,64')
    p.add_argument('--rpn_nms_threshold', type=float, default=0.7)
    p.add_argument('--rpn_train_anchors_per_image', type=int, default=512)
    p.add_argument('--train_rois_per_image', type=int, default=512)
    p.add_argument('--rpn_anchor_ratios', type=str, default='0.5,1,2')
    for n in ("rpn_class", "rpn_bbox", "mrcnn_class", "mrcnn_bbox", "mrcnn_mask"):
        p.add_argument('--%s_loss_weight' % n, type=float, default=1.0)
        p.add_argument('--%s_loss' % n, dest='%s_loss' % n, action='store_true', default=True)
        p.add_argument('--no_%s_loss' % n, dest='%s_loss' % n, action='store_false')
    p.add_argument('--mask_loss_function', type=str, default='binary_crossentropy',
                   choices=['binary_crossentropy', 'dice_coef_loss'])
    p.add_argument('--weight_classes', action='store_true', default=False)
    p.add_argument('--exclude_first_layer_weights', action='store_true', default=False)
    p.add_argument('--head_dtype', type=str, default="", help="extension: float16 | bfloat16 mask-head convolutions")
    p.add_argument('--no_augmentation', dest='use_augmentation', action='store_false', default=True)
    p.add_argument('--scoreThr', type=float, default=0.7)
    p.add_argument('--iouThr', type=float, default=0.6)
    p.add_argument('--image', type=str, default=None)
    for n in ("xmin", "xmax", "ymin", "ymax"):
        p.add_argument('--' + n, type=int, default=-1)
    p.add_argument('--detect_outfile', type=str, default="")
    p.add_argument('--detect_outfile_json', type=str, default="")
    p.add_argument('--split_img_in_tiles', action='store_true', default=False)
    p.add_argument('--tile_xsize', type=int, default=512)
    p.add_argument('--tile_ysize', type=int, default=512)
    p.add_argument('--tile_xstep', type=float, default=1.0)
    p.add_argument('--tile_ystep', type=float, default=1.0)
    return p.parse_args(argv)


def build_config(args, class_dict_model):
    """SDetectorConfig + the CLI overrides of scripts/run.py:1628-1706."""
    from caesar_mrcnn_amd.config import SDetectorConfig
    cfg = SDetectorConfig()
    cfg.NUM_CLASSES = len(class_dict_model) + 1
    cfg.CLASS_NAMES = ["bkg"] + [k for k, _ in sorted(class_dict_model.items(), key=lambda kv: kv[1])]
    cfg.CLASS_DICT = class_dict_model
    cfg.IMAGE_META_SIZE = 1 + 3 + 3 + 4 + 1 + cfg.NUM_CLASSES
    inference = args.command in ("test", "detect")
    cfg.GPU_COUNT = 1 if inference else args.ngpu
    cfg.IMAGES_PER_GPU = 1 if inference else args.nimg_per_gpu
    cfg.LOSS_WEIGHTS = {"%s_loss" % n: getattr(args, "%s_loss_weight" % n)
                        for n in ("rpn_class", "rpn_bbox", "mrcnn_class", "mrcnn_bbox", "mrcnn_mask")}
    cfg.USE_LOSSES = {"%s_loss" % n: getattr(args, "%s_loss" % n)
                      for n in ("rpn_class", "rpn_bbox", "mrcnn_class", "mrcnn_bbox", "mrcnn_mask")}
    cfg.MASK_LOSS_FUNCTION = args.mask_loss_function
    cfg.RPN_ANCHOR_SCALES = tuple(int(v) for v in args.rpn_anchor_scales.split(","))
    cfg.MAX_GT_INSTANCES = args.max_gt_instances
    cfg.BACKBONE = args.backbone
    cfg.BACKBONE_STRIDES = [int(v) for v in args.backbone_strides.split(",")]
    cfg.RPN_NMS_THRESHOLD = args.rpn_nms_threshold
    cfg.RPN_TRAIN_ANCHORS_PER_IMAGE = args.rpn_train_anchors_per_image
    cfg.TRAIN_ROIS_PER_IMAGE = args.train_rois_per_image
    cfg.RPN_ANCHOR_RATIOS = [float(v) for v in args.rpn_anchor_ratios.split(",")]
    cfg.IMAGE_MIN_DIM = cfg.IMAGE_MAX_DIM = args.imgsize
    cfg.IMAGE_SHAPE = np.array([args.imgsize, args.imgsize, cfg.IMAGE_CHANNEL_COUNT])
    cfg.ZSCALE_STRETCH, cfg.IMG_TO_UINT8, cfg.NORMALIZE_IMG = args.zscale, args.to_uint8, args.norm_img
    cfg.ZSCALE_CONTRASTS = [float(v) for v in args.zscale_contrasts.split(",")]
    cfg.BIAS_CONTRAST_STRETCH, cfg.IMG_BIAS, cfg.IMG_CONTRAST = args.biascontrast, args.bias, args.contrast
    cfg.IOU_THR, cfg.SCORE_THR = args.iouThr, args.scoreThr
    cfg.IMG_PATH = args.image or ""
    cfg.IMG_XMIN, cfg.IMG_XMAX, cfg.IMG_YMIN, cfg.IMG_YMAX = args.xmin, args.xmax, args.ymin, args.ymax
    cfg.SPLIT_IMG_IN_TILES = args.split_img_in_tiles
    cfg.TILE_XSIZE, cfg.TILE_YSIZE, cfg.TILE_XSTEP, cfg.TILE_YSTEP = args.tile_xsize, args.tile_ysize, args.tile_xstep, args.tile_ystep
    cfg.HEAD_DTYPE = args.head_dtype or None
    return cfg


def make_dataset(args, path):
    from caesar_mrcnn_amd.dataset import SourceDataset
    ds = SourceDataset()
    if ds.set_class_dict(args.classdict) < 0:
        return None
    ds.apply_zscale, ds.convert_to_uint8 = args.zscale, args.to_uint8
    ds.zscale_contrasts = [float(v) for v in args.zscale_contrasts.split(",")]
    ds.apply_biascontrast, ds.bias, ds.contrast = args.biascontrast, args.bias, args.contrast
    if args.dataloader == 'datalist':
        rc = ds.load_data_from_list(path, args.maxnimgs)
    elif args.dataloader == 'datalist_json':
        rc = ds.load_data_from_json_list(path, args.maxnimgs)
    elif args.dataloader == 'datadir_json':
        rc = ds.load_data_from_json_search(path, args.maxnimgs)
    else:
        logger.error("Invalid/unknown dataloader (%s)!" % args.dataloader)
        return None
    if rc < 0:
        logger.error("Failed to load dataset %s" % path)
        return None
    ds.prepare()
    return ds


def train(args, model, cfg):
    src = args.datadir if args.dataloader == 'datadir_json' else (args.datalist_train or args.datalist)
    ds_train = make_dataset(args, src)
    ds_val = make_dataset(args, args.datalist_val) if args.datalist_val else ds_train
    if ds_train is None or ds_val is None:
        return -1
    world = max(1, cfg.GPU_COUNT)
    cfg.STEPS_PER_EPOCH = args.epoch_length or max(1, len(ds_train.image_ids) // (cfg.IMAGES_PER_GPU * world))
    cfg.VALIDATION_STEPS = args.nvalidation_steps or max(1, len(ds_val.image_ids) // (cfg.IMAGES_PER_GPU * world))
    aug = None
    if args.use_augmentation:
        rng = np.random.RandomState(0)

        def aug(image, mask):        # flips + 90-degree rotations (imgaug SomeOf in run.py:1091-1100)
            if rng.rand() < 0.5:
                image, mask = np.fliplr(image), np.fliplr(mask)
            if rng.rand() < 0.5:
                image, mask = np.flipud(image), np.flipud(mask)
            k = rng.randint(0, 4)
            if image.shape[0] == image.shape[1]:
                image, mask = np.rot90(image, k), np.rot90(mask, k)
            return np.ascontiguousarray(image), np.ascontiguousarray(mask)
    model.train(ds_train, ds_val, learning_rate=cfg.LEARNING_RATE, epochs=args.nepochs, layers='all', augmentation=aug,
                n_worker_threads=args.nthreads, class_weights=ds_train.compute_class_weights() if args.weight_classes else None)
    return 0


def _mask_iou(a, b):
    inter = np.logical_and(a, b).sum()
    union = np.logical_or(a, b).sum()
    return inter / union if union else 0.0


def test(args, model, cfg):
    ds = make_dataset(args, args.datadir if args.dataloader == 'datadir_json' else args.datalist)
    if ds is None:
        return -1
    C = cfg.NUM_CLASSES
    n_true, n_true_det = np.zeros(C), np.zeros(C)
    n_det, n_det_ok = np.zeros(C), np.zeros(C)
    for image_id in ds.image_ids:
        image = ds.load_image(image_id)
        gt_masks, gt_ids = ds.load_mask(image_id)
        r = model.detect([image], verbose=0)[0]
        keep = r["scores"] >= cfg.SCORE_THR
        det_masks, det_ids = r["masks"][:, :, keep], r["class_ids"][keep]
        matched_det = set()
        for g in range(gt_masks.shape[-1]):
            n_true[gt_ids[g]] += 1
            best, bj = 0.0, -1
            for j in range(det_masks.shape[-1]):
                iou = _mask_iou(gt_masks[:, :, g], det_masks[:, :, j])
                if iou > best:
                    best, bj = iou, j
            if best >= cfg.IOU_THR:
                n_true_det[gt_ids[g]] += 1
                matched_det.add(bj)
        for j in range(det_masks.shape[-1]):
            n_det[det_ids[j]] += 1
            if j in matched_det:
                n_det_ok[det_ids[j]] += 1
    for c in range(1, C):
        name = cfg.CLASS_NAMES[c]
        comp = n_true_det[c] / n_true[c] if n_true[c] else float("nan")
        rel = n_det_ok[c] / n_det[c] if n_det[c] else float("nan")
        print("class %s: completeness %.3f (%d/%d) reliability %.3f (%d/%d)" % (name, comp, n_true_det[c], n_true[c], rel,
                                                                                 n_det_ok[c], n_det[c]))
    return 0


def detect(args, model, cfg):
    from caesar_mrcnn_amd import fits
    if not args.image:
        logger.error("No input image given (--image)")
        return -1
    tiles = [(args.xmin, args.xmax, args.ymin, args.ymax)]
    if cfg.SPLIT_IMG_IN_TILES:
        size = fits.get_fits_size(args.image)
        if size is None:
            return -1
        nx, ny = size
        x0, x1 = (args.xmin, args.xmax) if args.xmin >= 0 and args.xmax >= 0 else (0, nx - 1)
        y0, y1 = (args.ymin, args.ymax) if args.ymin >= 0 and args.ymax >= 0 else (0, ny - 1)
        tiles = fits.generate_tiles(x0, x1, y0, y1, cfg.TILE_XSIZE, cfg.TILE_YSIZE, cfg.TILE_XSTEP, cfg.TILE_YSTEP) or tiles
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    from caesar_mrcnn_amd.analyze import Analyzer
    base = os.path.splitext(os.path.basename(args.image))[0]
    analyzer = Analyzer(model, cfg)
    analyzer.score_thr, analyzer.iou_thr = args.scoreThr, args.iouThr
    analyzer.write_to_json = analyzer.write_to_ds9 = False          # one file for all tiles, written below
    objs, regions = [], []
    for t, (xmin, xmax, ymin, ymax) in enumerate(tiles):
        if t % world != rank:            # tiles are independent: replicas only (sfinder.py:1235-1251 round-robin)
            continue
        res = fits.read_fits(args.image, xmin, xmax, ymin, ymax, stretch=cfg.ZSCALE_STRETCH,
                             zscale_contrasts=cfg.ZSCALE_CONTRASTS, normalize=cfg.NORMALIZE_IMG, convertToRGB=True,
                             to_uint8=cfg.IMG_TO_UINT8, stretch_biascontrast=cfg.BIAS_CONTRAST_STRETCH, bias=cfg.IMG_BIAS,
                             contrast=cfg.IMG_CONTRAST)
        if res is None:
            return -1
        image, header = res
        analyzer.obj_name_tag = "t%d" % t if len(tiles) > 1 else ""
        if analyzer.predict(image, image_id=base, header=header, xmin=max(xmin, 0), ymin=max(ymin, 0)) < 0:
            logger.error("Failed to run model prediction on image %s!" % args.image)
            return -1
        if analyzer.results:
            objs.extend(analyzer.results["objs"])
            regions.extend(analyzer.obj_regions)
    suffix = "" if world == 1 else "_rank%d" % rank
    out = args.detect_outfile_json or ("out_%s%s.json" % (base, suffix))
    analyzer.results = {"image_id": base, "objs": objs}
    analyzer.write_json_results(out)
    analyzer.obj_regions = regions
    analyzer.write_ds9_regions(os.path.splitext(out)[0] + ".reg")
    print("%d sources written to %s" % (len(objs), out))
    return 0


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s - %(message)s")
    args = parse_args(argv)
    if args.command not in ("train", "test", "detect"):
        logger.error("Unknown command %s" % args.command)
        return 1
    if args.command in ("train", "test") and not (args.datalist or args.datalist_train or args.datadir):
        logger.error("Argument --datalist/--datadir is required for training/testing")
        return 1
    try:
        class_dict = json.loads(args.classdict)
        class_dict_model = json.loads(args.classdict_model) if args.classdict_model else class_dict
    except Exception:
        logger.error("Failed to convert class dict string to dict!")
        return 1
    from caesar_mrcnn_amd import model as modellib
    cfg = build_config(args, class_dict_model)
    cfg.display()
    mode = "training" if args.command == "train" else "inference"
    model = modellib.MaskRCNN(mode=mode, config=cfg, model_dir=args.logs)
    if args.weights:
        if args.exclude_first_layer_weights:
            model.load_weights(args.weights, by_name=True, exclude='conv1')
        else:
            model.load_weights(args.weights, by_name=True)
    else:
        logger.info("No weights given: starting from random initialisation")
    rc = {"train": train, "test": test, "detect": detect}[args.command](args, model, cfg)
    return 0 if rc == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
