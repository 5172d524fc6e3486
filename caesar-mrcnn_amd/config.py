"""Hyper-parameter surface of the reference (mrcnn/config.py:17-247 and the SDetectorConfig overrides
of scripts/run.py:93-239).  Attribute names, defaults and the three derived fields are the contract;
everything else here is ours.  Unlike the reference, ``BATCH_SIZE`` is a live property so overriding
IMAGES_PER_GPU / GPU_COUNT on an instance (as run.py:1632-1633 does) stays consistent.
"""
import os

import numpy as np

_LOSS_NAMES = ("rpn_class_loss", "rpn_bbox_loss", "mrcnn_class_loss", "mrcnn_bbox_loss", "mrcnn_mask_loss")


class Config(object):
    NAME = None
    GPU_COUNT = 1
    IMAGES_PER_GPU = 2
    STEPS_PER_EPOCH = 1000
    VALIDATION_STEPS = 50
    BACKBONE = "resnet101"                      # "resnet50" | "resnet101" | "custom"
    COMPUTE_BACKBONE_SHAPE = None
    BACKBONE_STRIDES = [4, 8, 16, 32, 64]
    FPN_CLASSIF_FC_LAYERS_SIZE = 1024
    TOP_DOWN_PYRAMID_SIZE = 256
    NUM_CLASSES = 1
    RPN_ANCHOR_SCALES = (32, 64, 128, 256, 512)
    RPN_ANCHOR_RATIOS = [0.5, 1, 2]
    RPN_ANCHOR_STRIDE = 1
    RPN_NMS_THRESHOLD = 0.7
    RPN_TRAIN_ANCHORS_PER_IMAGE = 256
    PRE_NMS_LIMIT = 6000
    POST_NMS_ROIS_TRAINING = 2000
    POST_NMS_ROIS_INFERENCE = 1000
    USE_MINI_MASK = True
    MINI_MASK_SHAPE = (56, 56)
    IMAGE_RESIZE_MODE = "square"
    IMAGE_MIN_DIM = 800
    IMAGE_MAX_DIM = 1024
    IMAGE_MIN_SCALE = 0
    IMAGE_CHANNEL_COUNT = 3
    MEAN_PIXEL = np.array([123.7, 116.8, 103.9])
    TRAIN_ROIS_PER_IMAGE = 200
    ROI_POSITIVE_RATIO = 0.33
    POOL_SIZE = 7
    MASK_POOL_SIZE = 14
    MASK_SHAPE = [28, 28]
    MAX_GT_INSTANCES = 100
    RPN_BBOX_STD_DEV = np.array([0.1, 0.1, 0.2, 0.2])
    BBOX_STD_DEV = np.array([0.1, 0.1, 0.2, 0.2])
    DETECTION_MAX_INSTANCES = 100
    DETECTION_MIN_CONFIDENCE = 0.7
    DETECTION_NMS_THRESHOLD = 0.3
    LEARNING_RATE = 0.001
    LEARNING_MOMENTUM = 0.9
    WEIGHT_DECAY = 0.0001
    LOSS_WEIGHTS = {n: 1. for n in _LOSS_NAMES}
    USE_LOSSES = {n: True for n in _LOSS_NAMES}
    USE_RPN_ROIS = True
    TRAIN_BN = False
    GRADIENT_CLIP_NORM = 5.0
    MASK_LOSS_FUNCTION = 'binary_crossentropy'  # or 'dice_coef_loss'
    # extension (not in the reference): build rpn_match / rpn_bbox on the GPU (csrc/rpn_targets.hip) instead of in
    # the NumPy generator; False restores the host path and its np.random.choice stream (model.py:1536-1644)
    DEVICE_RPN_TARGETS = True
    # extension: single-rank training steps (forward, backward, optimiser) replayed from one HIP graph
    # (engine.step_graphed); data-parallel runs keep eager launches (the gradient hooks are not capturable).
    # Off by default: measured 2-3 ms SLOWER per step than eager launches in every mode (ResNet-101, 4 images:
    # float32 dense 60.2 -> 62.5 ms, 16-bit sparse 14.1 -> 16.7 ms; 512x512 f16 27.9 -> 31.6) -- the replay runs the
    # three forked streams of the step with less overlap than the eager queues do, which costs more than the host
    # time it saves
    TRAIN_HIP_GRAPH = False
    # extension: single-rank training steps re-issued from a launch recording (engine.step_taped): the same launches on the same
    # streams as the eager step, without the engine's Python per launch.  Device time is unchanged; what it buys is the
    # interpreter lock -- the loader threads of MaskRCNN.train() share it with the main thread, and the positive-quota loop went
    # from 121-127 to 168-182 images/s (ResNet-101, 4 images, 8 loader threads; the dense loop is device-bound either way).
    # MRCNN_TRAIN_TAPE=0 switches it off.  Data-parallel runs keep eager launches (gradient hooks are not recorded).
    TRAIN_LAUNCH_TAPE = os.environ.get("MRCNN_TRAIN_TAPE", "1") != "0"
    # FITS tiles of the training / validation datasets: NaN fill, zscale, normalisation and the uint8 RGB conversion of
    # utils.read_fits (mrcnn/utils.py:1088-1157) on the GPU (mrcnn_fits_to_rgb; byte-identical images), from the loader threads
    DEVICE_FITS = os.environ.get("MRCNN_DEVICE_FITS", "1") != "0"
    # extension: None (float32 everywhere, the reference's precision) | "float16" | "bfloat16": run the 3x3
    # convolutions of the mask head on the 16-bit matrix cores (csrc/conv_h16.hip; float32 master weights,
    # accumulation and gradients; HEAD_LOSS_SCALE guards float16 gradients)
    HEAD_DTYPE = None
    HEAD_LOSS_SCALE = 4096.0

    def __init__(self):
        side = self.IMAGE_MIN_DIM if self.IMAGE_RESIZE_MODE == "crop" else self.IMAGE_MAX_DIM
        self.IMAGE_SHAPE = np.array([side, side, self.IMAGE_CHANNEL_COUNT])
        self.IMAGE_META_SIZE = 1 + 3 + 3 + 4 + 1 + self.NUM_CLASSES

    @property
    def BATCH_SIZE(self):
        return self.IMAGES_PER_GPU * self.GPU_COUNT

    def display(self):
        print("\nConfigurations:")
        for a in dir(self):
            if not a.startswith("__") and not callable(getattr(self, a)):
                print("{:30} {}".format(a, getattr(self, a)))
        print("\n")


class SDetectorConfig(Config):
    """Effective defaults of the radio-source detector (scripts/run.py:93-239)."""
    NAME = "rg-dataset"
    GPU_COUNT = 1
    IMAGES_PER_GPU = 2
    NUM_CLASSES = 1
    CLASS_NAMES = ["bkg"]
    VALIDATION_STEPS = max(1, 200 // 2)
    STEPS_PER_EPOCH = (16439 - 200) // 2
    DETECTION_MIN_CONFIDENCE = 0
    DETECTION_NMS_THRESHOLD = 0.3
    RPN_ANCHOR_SCALES = (4, 8, 16, 32, 64)
    MAX_GT_INSTANCES = 300
    BACKBONE = "resnet101"
    IMAGE_RESIZE_MODE = "square"
    IMAGE_MIN_DIM = 256
    IMAGE_MAX_DIM = 256
    MEAN_PIXEL = np.array([0, 0, 0])
    RPN_NMS_THRESHOLD = 0.9
    RPN_TRAIN_ANCHORS_PER_IMAGE = 512
    TRAIN_ROIS_PER_IMAGE = 512
    LEARNING_RATE = 0.0005
    OPTIMIZER = "ADAM"          # never read by the reference either (model.py:2260 always builds SGD)
    USE_MINI_MASK = False
    IMG_PATH = ""
    IMG_XMIN = IMG_XMAX = IMG_YMIN = IMG_YMAX = 0
    OUTFILE = ""
    OUTFILE_JSON = ""
    ZSCALE_STRETCH = True
    ZSCALE_CONTRASTS = [0.25, 0.25, 0.25]
    NORMALIZE_IMG = True
    IMG_TO_UINT8 = True
    IMG_TO_RGB = True
    BIAS_CONTRAST_STRETCH = False
    IMG_BIAS = 0.5
    IMG_CONTRAST = 1.0
    IOU_THR = 0.6
    SCORE_THR = 0.7
    MPI = None
    SPLIT_IMG_IN_TILES = False
    TILE_XSIZE = 512
    TILE_YSIZE = 512
    TILE_XSTEP = 1.0
    TILE_YSTEP = 1.0
    MAX_NTASKS_PER_WORKER = 100


def run_py_config(num_classes=4, imgsize=256, backbone="resnet101", images_per_gpu=2, gpu_count=1,
                  rpn_nms_threshold=0.7, mode="training"):
    """The instance run.py ends up with after its CLI overrides (scripts/run.py:1628-1706)."""
    cfg = SDetectorConfig()
    cfg.NUM_CLASSES = num_classes
    cfg.IMAGE_META_SIZE = 12 + num_classes
    cfg.IMAGE_MIN_DIM = cfg.IMAGE_MAX_DIM = imgsize
    cfg.IMAGE_SHAPE = np.array([imgsize, imgsize, cfg.IMAGE_CHANNEL_COUNT])
    cfg.BACKBONE = backbone
    cfg.RPN_NMS_THRESHOLD = rpn_nms_threshold
    if mode == "inference":
        images_per_gpu, gpu_count = 1, 1
    cfg.IMAGES_PER_GPU = images_per_gpu
    cfg.GPU_COUNT = gpu_count
    return cfg
