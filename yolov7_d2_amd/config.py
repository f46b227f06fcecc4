"""Config surface for the YOLOX path: a yacs-free CfgNode that loads the reference's YAMLs
(`_BASE_` inheritance + `--opts` style overrides) with the defaults the YOLOX meta-arch reads.

Keys and default values restate yolov7/config.py:11-114 (add_yolo_config) for the sections the hot
path consumes, on top of the detectron2 defaults listed in SURVEY.md Appendix D.  Unknown keys in a
YAML are accepted (the reference's YAMLs carry many keys for other model families).
If detectron2 is installed, use its get_cfg() + this module's add_yolo_config() instead.
"""
import copy
import os

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_other(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_other(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, path):
        self.merge_from_other(_load_yaml_with_base(path))

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, CfgNode())
            if isinstance(v, str):
                try:
                    v = yaml.safe_load(v)
                except Exception:
                    pass
            node[parts[-1]] = v

    def freeze(self):
        return self


CN = CfgNode


def _load_yaml_with_base(path):
    with open(path) as f:
        cfg = yaml.unsafe_load(f) or {}
    base = cfg.pop("_BASE_", None)
    if base is not None:
        if not os.path.isabs(base):
            base = os.path.join(os.path.dirname(path), base)
        if not os.path.exists(base):
            # the reference's yolox_s.yaml names "../Base-YoloV7.yaml" while the file is Base-YOLOv7.yaml
            d = os.path.dirname(base)
            cands = [f for f in os.listdir(d) if f.lower() == os.path.basename(base).lower()]
            if cands:
                base = os.path.join(d, cands[0])
        b = CfgNode(_load_yaml_with_base(base))
        b.merge_from_other(cfg)
        return b
    return cfg


def get_cfg():
    """detectron2 default keys the path touches (d2 upstream defaults)."""
    c = CN()
    c.VERSION = 2
    c.OUTPUT_DIR = "./output"
    c.SEED = -1
    c.CUDNN_BENCHMARK = False
    c.VIS_PERIOD = 0
    c.MODEL = CN()
    c.MODEL.DEVICE = "cuda"
    c.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
    c.MODEL.WEIGHTS = ""
    c.MODEL.MASK_ON = False
    c.MODEL.KEYPOINT_ON = False
    c.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
    c.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
    c.MODEL.BACKBONE = CN({"NAME": "build_resnet_backbone", "FREEZE_AT": 2})
    c.MODEL.FPN = CN({"IN_FEATURES": [], "OUT_CHANNELS": 256, "NORM": "", "FUSE_TYPE": "sum"})
    c.MODEL.RESNETS = CN({"DEPTH": 50, "NORM": "FrozenBN", "STRIDE_IN_1X1": True, "OUT_FEATURES": ["res4"]})
    c.DATASETS = CN({"TRAIN": (), "TEST": ()})
    c.DATALOADER = CN({"NUM_WORKERS": 4, "FILTER_EMPTY_ANNOTATIONS": True})
    c.INPUT = CN({"FORMAT": "BGR", "MIN_SIZE_TRAIN": (800,), "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800,
                  "MAX_SIZE_TEST": 1333, "MASK_FORMAT": "polygon",
                  "CROP": {"ENABLED": False, "TYPE": "relative_range", "SIZE": [0.9, 0.9]}})
    c.SOLVER = CN({"IMS_PER_BATCH": 16, "BASE_LR": 0.001, "STEPS": (30000,), "MAX_ITER": 40000,
                   "WARMUP_FACTOR": 1.0 / 1000, "WARMUP_ITERS": 1000, "WEIGHT_DECAY": 0.0001,
                   "WEIGHT_DECAY_NORM": 0.0, "WEIGHT_DECAY_BIAS": None, "BIAS_LR_FACTOR": 1.0, "MOMENTUM": 0.9,
                   "NESTEROV": False, "LR_SCHEDULER_NAME": "WarmupMultiStepLR", "CHECKPOINT_PERIOD": 5000,
                   "AMP": {"ENABLED": False}, "REFERENCE_WORLD_SIZE": 0,
                   "CLIP_GRADIENTS": {"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0}})
    c.TEST = CN({"EVAL_PERIOD": 0, "PRECISE_BN": {"ENABLED": False, "NUM_ITER": 200}})
    return c


def add_yolo_config(cfg):
    """restates the YOLOX-relevant defaults of yolov7/config.py:11-114 (+ utils/get_default_cfg.py:9-12)."""
    _C = cfg
    assert _C.SOLVER.REFERENCE_WORLD_SIZE == 0
    _C.SOLVER.REFERENCE_WORLD_SIZE = 8
    _C.SOLVER.OPTIMIZER = "sgd"
    # yolov7/config.py:19: the reference's add_yolo_config itself calls add_sparse_inst_config, so train_inseg.py:42-49
    # (get_cfg + add_yolo_config + merge_from_file) finds every MODEL.SPARSE_INST default the YAMLs do not spell out
    # (ENCODER.NUM_CHANNELS, DECODER.*, LOSS.*, MATCHER.*).  Its side effects are the reference's too: MODEL.MASK_ON True and
    # MODEL.DEVICE "cuda" until a YAML says otherwise (detr_256_6_6_torchvision.yaml sets MASK_ON False), SOLVER.AMSGRAD
    add_sparse_inst_config(_C)
    _C.DATASETS.CLASS_NAMES = []
    _C.MODEL.NMS_TYPE = "normal"
    _C.MODEL.ONNX_EXPORT = False
    _C.MODEL.PADDED_VALUE = 114.0
    _C.MODEL.FPN.REPEAT = 2
    _C.MODEL.FPN.OUT_CHANNELS_LIST = [256, 512, 1024]
    _C.MODEL.BIFPN = CN({"NUM_LEVELS": 5, "NUM_BIFPN": 6, "NORM": "GN", "OUT_CHANNELS": 160, "SEPARABLE_CONV": False})
    _C.INPUT.INPUT_SIZE = [640, 640]
    _C.MODEL.YOLO = CN({
        "NUM_BRANCH": 3, "VARIANT": "yolov3", "ANCHOR_MASK": [], "CLASSES": 80, "MAX_BOXES_NUM": 100,
        "IN_FEATURES": ["dark3", "dark4", "dark5"], "CONF_THRESHOLD": 0.01, "NMS_THRESHOLD": 0.5,
        "IGNORE_THRESHOLD": 0.07, "NORMALIZE_INPUT": False, "WIDTH_MUL": 1.0, "DEPTH_MUL": 1.0, "IOU_TYPE": "ciou",
        "LOSS_TYPE": "v4",
        "LOSS": {"LAMBDA_XY": 1.0, "LAMBDA_WH": 1.0, "LAMBDA_CLS": 1.0, "LAMBDA_CONF": 1.0, "LAMBDA_IOU": 1.1,
                 "USE_L1": True, "ANCHOR_RATIO_THRESH": 4.0, "BUILD_TARGET_TYPE": "default"},
        "NECK": {"TYPE": "yolov3", "WITH_SPP": False}, "HEAD": {"TYPE": "yolox"},
    })
    _C.MODEL.DARKNET = CN({"DEPTH": 53, "WITH_CSP": True, "RES5_DILATION": 1, "NORM": "BN", "STEM_OUT_CHANNELS": 32,
                           "OUT_FEATURES": ["dark3", "dark4", "dark5"], "WEIGHTS": "", "DEPTH_WISE": False})
    _C.INPUT.MOSAIC_AND_MIXUP = CN({"ENABLED": False, "DEBUG_VIS": False, "ENABLE_MIXUP": False,
                                    "DISABLE_AT_ITER": 120000})
    # DETR (yolov7/config.py:199-245)
    _C.MODEL.DETR = CN({
        "NUM_CLASSES": 80, "FROZEN_WEIGHTS": "", "DEFORMABLE": False, "USE_FOCAL_LOSS": False,
        "CENTERED_POSITION_ENCODIND": False, "CLS_WEIGHT": 1.0, "NUM_FEATURE_LEVELS": 1, "GIOU_WEIGHT": 2.0,
        "L1_WEIGHT": 5.0, "DEEP_SUPERVISION": True, "NO_OBJECT_WEIGHT": 0.1, "WITH_BOX_REFINE": False, "TWO_STAGE": False,
        "DECODER_BLOCK_GRAD": True, "ATTENTION_TYPE": "DETR", "NHEADS": 8, "DROPOUT": 0.1, "DIM_FEEDFORWARD": 2048,
        "ENC_LAYERS": 6, "DEC_LAYERS": 6, "PRE_NORM": False, "BBOX_EMBED_NUM_LAYERS": 3, "HIDDEN_DIM": 256,
        "NUM_OBJECT_QUERIES": 100, "NUM_QUERY_POSITION": 300, "NUM_QUERY_PATTERN": 3, "SPATIAL_PRIOR": "learned",
    })
    _C.MODEL.BACKBONE.SIMPLE = False
    _C.MODEL.BACKBONE.STRIDE = 1
    _C.MODEL.BACKBONE.CHANNEL = 0
    _C.SOLVER.OPTIMIZER = "ADAMW"          # yolov7/config.py:243-244 (train_det.py's d2 build_optimizer does not read it)
    _C.SOLVER.BACKBONE_MULTIPLIER = 0.1
    return _C


def add_sparse_inst_config(cfg):
    """yolov7/configs/config_sparseinst.py:6-68"""
    cfg.MODEL.DEVICE = "cuda"
    cfg.MODEL.MASK_ON = True
    cfg.MODEL.SPARSE_INST = CN({
        "CLS_THRESHOLD": 0.005, "MASK_THRESHOLD": 0.45, "MAX_DETECTIONS": 100,
        "ENCODER": {"NAME": "FPNPPMEncoder", "NORM": "", "IN_FEATURES": ["res3", "res4", "res5"], "NUM_CHANNELS": 256},
        "DECODER": {"NAME": "BaseIAMDecoder", "NUM_MASKS": 100, "NUM_CLASSES": 80, "KERNEL_DIM": 128, "SCALE_FACTOR": 2.0,
                    "OUTPUT_IAM": False, "GROUPS": 4, "INST": {"DIM": 256, "CONVS": 4}, "MASK": {"DIM": 256, "CONVS": 4}},
        "LOSS": {"NAME": "SparseInstCriterion", "ITEMS": ("labels", "masks"), "CLASS_WEIGHT": 2.0,
                 "MASK_PIXEL_WEIGHT": 5.0, "MASK_DICE_WEIGHT": 2.0, "OBJECTNESS_WEIGHT": 1.0},
        "MATCHER": {"NAME": "SparseInstMatcher", "ALPHA": 0.8, "BETA": 0.2},
        "DATASET_MAPPER": "SparseInstDatasetMapper",
    })
    cfg.SOLVER.OPTIMIZER = "ADAMW"
    cfg.SOLVER.BACKBONE_MULTIPLIER = 1.0
    cfg.SOLVER.AMSGRAD = False
    return cfg


def sparse_inst_r50_giam_cfg(device="cuda", **over):
    """configs/coco/sparseinst/Base-SparseInst.yaml + sparse_inst_r50_giam.yaml without needing the files"""
    cfg = add_yolo_config(get_cfg())          # (includes add_sparse_inst_config, as the reference's does)
    cfg.MODEL.DEVICE = device
    cfg.MODEL.META_ARCHITECTURE = "SparseInst"
    cfg.MODEL.PIXEL_MEAN = [123.675, 116.280, 103.530]
    cfg.MODEL.PIXEL_STD = [58.395, 57.120, 57.375]
    cfg.MODEL.BACKBONE.FREEZE_AT = 0
    cfg.MODEL.BACKBONE.NAME = "build_resnet_backbone"
    cfg.MODEL.RESNETS.NORM = "FrozenBN"
    cfg.MODEL.RESNETS.DEPTH = 50
    cfg.MODEL.RESNETS.STRIDE_IN_1X1 = False
    cfg.MODEL.RESNETS.OUT_FEATURES = ["res3", "res4", "res5"]
    cfg.MODEL.SPARSE_INST.ENCODER.NAME = "InstanceContextEncoder"
    cfg.MODEL.SPARSE_INST.DECODER.NAME = "GroupIAMDecoder"
    cfg.SOLVER.BASE_LR = 0.00005
    cfg.SOLVER.WEIGHT_DECAY = 0.05
    cfg.INPUT.FORMAT = "RGB"
    cfg.INPUT.MASK_FORMAT = "bitmask"
    for k, v in over.items():
        cfg.merge_from_list([k, v])
    return cfg


def detr_r50_cfg(device="cuda", **over):
    """the settings of configs/coco/detr/detr_256_6_6_torchvision.yaml without needing the file"""
    cfg = add_yolo_config(get_cfg())
    cfg.MODEL.DEVICE = device
    cfg.MODEL.META_ARCHITECTURE = "Detr"
    cfg.MODEL.PIXEL_MEAN = [123.675, 116.280, 103.530]
    cfg.MODEL.PIXEL_STD = [58.395, 57.120, 57.375]
    cfg.MODEL.MASK_ON = False
    cfg.MODEL.BACKBONE.NAME = "build_resnet_backbone"
    cfg.MODEL.RESNETS.DEPTH = 50
    cfg.MODEL.RESNETS.STRIDE_IN_1X1 = False
    cfg.MODEL.RESNETS.OUT_FEATURES = ["res2", "res3", "res4", "res5"]
    cfg.MODEL.DETR.GIOU_WEIGHT, cfg.MODEL.DETR.L1_WEIGHT = 2.0, 5.0
    cfg.MODEL.DETR.NUM_OBJECT_QUERIES = 100
    cfg.MODEL.DETR.ENC_LAYERS = cfg.MODEL.DETR.DEC_LAYERS = 6
    cfg.MODEL.DETR.HIDDEN_DIM = 256
    cfg.SOLVER.OPTIMIZER = "ADAMW"
    cfg.SOLVER.BASE_LR = 0.0001
    cfg.INPUT.FORMAT = "RGB"
    for k, v in over.items():
        cfg.merge_from_list([k, v])
    return cfg


def get_yolox_cfg(config_file=None, opts=()):
    cfg = add_yolo_config(get_cfg())
    if config_file:
        cfg.merge_from_file(config_file)
    if opts:
        cfg.merge_from_list(list(opts))
    return cfg


def yolox_s_cfg(device="cuda", **over):
    """the settings of configs/coco/yolox_s.yaml (+ Base-YOLOv7.yaml) without needing the file."""
    cfg = add_yolo_config(get_cfg())
    cfg.MODEL.DEVICE = device
    cfg.MODEL.META_ARCHITECTURE = "YOLOX"
    cfg.MODEL.PIXEL_MEAN = [0.485, 0.456, 0.406]
    cfg.MODEL.PIXEL_STD = [0.229, 0.224, 0.225]
    cfg.MODEL.BACKBONE.NAME = "build_cspdarknetx_backbone"
    cfg.MODEL.YOLO.CONF_THRESHOLD = 0.001
    cfg.MODEL.YOLO.NMS_THRESHOLD = 0.65
    cfg.MODEL.YOLO.WIDTH_MUL = 0.50
    cfg.MODEL.YOLO.DEPTH_MUL = 0.33
    cfg.MODEL.YOLO.LOSS_TYPE = "v7"
    cfg.SOLVER.MAX_ITER = 230000
    cfg.SOLVER.BASE_LR = 0.027
    cfg.SOLVER.IMS_PER_BATCH = 112
    cfg.SOLVER.AMP.ENABLED = True
    cfg.SOLVER.LR_SCHEDULER_NAME = "WarmupCosineLR"
    cfg.SOLVER.WARMUP_ITERS = 1200
    cfg.SOLVER.WARMUP_FACTOR = 0.00033333
    for k, v in over.items():
        cfg.merge_from_list([k, v])
    return cfg
