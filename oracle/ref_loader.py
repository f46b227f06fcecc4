"""ORACLE support (test infrastructure): import the reference's OWN hot-path source files by path.

Works only where /root/reference exists (this container, not the GPU box).  The reference package
cannot be imported normally (yolov7/__init__.py pulls detectron2, timm, alfred, torchvision ... none of
which are installed), so the un-installed third-party names are stubbed and the hot-path files are
loaded as sub-modules of namespace packages whose __init__.py is NOT executed.
Used by oracle/gen_golden.py to pin oracle/yolox_oracle.py and to produce tests/golden/*.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("MI355_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "yolov7"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """returns a namespace with the reference classes/functions of the YOLOX path"""
    if not available():
        raise RuntimeError(f"reference not found under {REF}")
    import torch
    from torch import nn

    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import yolox_oracle as oracle  # our torchvision-semantics NMS backs the torchvision stub

    class Registry(dict):
        def register(self, obj=None):
            if obj is None:
                return lambda o: self.register(o)
            self[obj.__name__] = obj
            return obj

    class Backbone(nn.Module):
        size_divisibility = 0

    class ShapeSpec:
        def __init__(self, channels=None, height=None, width=None, stride=None):
            self.channels, self.height, self.width, self.stride = channels, height, width, stride

    if "detectron2" not in sys.modules:
        _stub("detectron2")
        _stub("detectron2.layers", ShapeSpec=ShapeSpec)
        _stub("detectron2.layers.batch_norm", get_norm=lambda *a, **k: None)
        _stub("detectron2.modeling", META_ARCH_REGISTRY=Registry(), BACKBONE_REGISTRY=Registry(), Backbone=Backbone)
        _stub("detectron2.modeling.backbone", Backbone=Backbone, BACKBONE_REGISTRY=sys.modules["detectron2.modeling"].BACKBONE_REGISTRY)
        _stub("detectron2.modeling.backbone.build", BACKBONE_REGISTRY=sys.modules["detectron2.modeling"].BACKBONE_REGISTRY)
        _stub("detectron2.utils")
        _stub("detectron2.utils.comm")
    if "loguru" not in sys.modules:
        class _Logger:
            def __getattr__(self, k):
                return lambda *a, **kw: None
        _stub("loguru", logger=_Logger())
    if "omegaconf" not in sys.modules:
        _stub("omegaconf")
        _stub("omegaconf.base")
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision", __version__="0.12.0", _is_tracing=lambda: False)
        ops = _stub("torchvision.ops", nms=oracle.nms, batched_nms=oracle.batched_nms)
        _stub("torchvision.ops.boxes", nms=oracle.nms,
              box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
        tv.ops = ops
        _stub("torchvision.models")
        _stub("torchvision.models._utils", IntermediateLayerGetter=object)
    if "cv2" not in sys.modules:
        _stub("cv2")
    if "pycocotools" not in sys.modules:
        _stub("pycocotools")
        _stub("pycocotools.mask")

    def ns(name, path):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m

    y = os.path.join(REF, "yolov7")
    ns("yolov7", y)
    ns("yolov7.utils", os.path.join(y, "utils"))
    ns("yolov7.modeling", os.path.join(y, "modeling"))
    ns("yolov7.modeling.backbone", os.path.join(y, "modeling", "backbone"))
    ns("yolov7.modeling.backbone.layers", os.path.join(y, "modeling", "backbone", "layers"))
    ns("yolov7.modeling.neck", os.path.join(y, "modeling", "neck"))
    ns("yolov7.modeling.head", os.path.join(y, "modeling", "head"))
    out = types.SimpleNamespace()
    out.wrappers = importlib.import_module("yolov7.modeling.backbone.layers.wrappers")
    out.boxes = importlib.import_module("yolov7.utils.boxes")
    out.darknetx = importlib.import_module("yolov7.modeling.backbone.darknetx")
    out.pafpn = importlib.import_module("yolov7.modeling.neck.yolo_pafpn")
    out.head = importlib.import_module("yolov7.modeling.head.yolox_head")
    out.detr_utils = importlib.import_module("yolov7.utils.detr_utils")   # HungarianMatcher (needs scipy: installed)
    return out


def _fvcore(nn_names=None, weight_init=None):
    """fvcore (un-vendored, not installed): get-or-create the stub packages and ADD whatever names are missing - every
    loader asks for its own, in any order (a loader that installed a bare `fvcore.nn` first used to make a later one's
    `from fvcore.nn import giou_loss` fail: the one-shot golden recipe died at gold_detr)"""
    pk = sys.modules.get("fvcore") or _stub("fvcore")
    if not hasattr(pk, "__path__"):
        pk.__path__ = []
    fn = sys.modules.get("fvcore.nn") or _stub("fvcore.nn")
    if not hasattr(fn, "__path__"):
        fn.__path__ = []
    pk.nn = fn
    for k, v in (nn_names or {}).items():
        if getattr(fn, k, None) is None:
            setattr(fn, k, v)
    wi = sys.modules.get("fvcore.nn.weight_init") or _stub("fvcore.nn.weight_init")
    for k, v in (weight_init or {}).items():
        setattr(wi, k, v)
    for k in ("c2_msra_fill", "c2_xavier_fill"):
        if not hasattr(wi, k):
            setattr(wi, k, lambda m: None)
    fn.weight_init = wi
    return fn


def _stub_alfred():
    """`alfred` (the reference author's utility package, un-vendored): the debug print and the logger the loaded files
    import at module level"""
    if "alfred" not in sys.modules:
        a = _stub("alfred", print_shape=lambda *a, **k: None)
        a.__path__ = []
        _stub("alfred.utils").__path__ = []
        _stub("alfred.utils.log", logger=sys.modules["loguru"].logger)
    elif not hasattr(sys.modules["alfred"], "print_shape"):
        sys.modules["alfred"].print_shape = lambda *a, **k: None


def load_detr():
    """the reference's DETR meta-arch file (meta_arch/detr.py: SetCriterion, PostProcess, MLP) loaded by path; the
    un-installed names it imports at module level (detectron2.structures, fvcore, alfred) are stubbed - none of them is
    touched by SetCriterion's labels / cardinality / boxes losses"""
    load()
    comm = sys.modules["detectron2.utils.comm"]
    comm.get_world_size = lambda: 1
    sys.modules["detectron2.utils"].comm = comm
    dm = sys.modules["detectron2.modeling"]
    dm.build_backbone = dm.detector_postprocess = None
    if "detectron2.structures" not in sys.modules:
        _stub("detectron2.structures", Boxes=object, ImageList=object, Instances=object, BitMasks=object, PolygonMasks=object)
        _stub("detectron2.utils.logger", log_first_n=lambda *a, **k: None)
    fn = _fvcore()
    for k in ("giou_loss", "smooth_l1_loss"):     # imported at module level, never called by the set criterion
        if not hasattr(fn, k):
            setattr(fn, k, None)
    _stub_alfred()
    name = "yolov7.modeling.meta_arch"
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, "yolov7", "modeling", "meta_arch")]
        sys.modules[name] = m
    return importlib.import_module("yolov7.modeling.meta_arch.detr")


def build_reference_yolox(depth=0.33, width=0.5, num_classes=80, seed=0, depthwise=False):
    """CSPDarknet + YOLOPAFPN + YOLOXHead assembled the way YOLOX.__init__ does (yolox.py:60-83)"""
    import torch
    from torch import nn
    r = load()
    torch.manual_seed(seed)

    class RefYOLOX(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = r.darknetx.CSPDarknet(depth, width, depthwise=depthwise)   # MODEL.DARKNET.DEPTH_WISE
            self.neck = r.pafpn.YOLOPAFPN(depth=depth, width=width)
            self.head = r.head.YOLOXHead(num_classes, width=width)
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eps, m.momentum = 1e-3, 0.03
            self.head.initialize_biases(1e-2)

        def forward(self, x, labels=None):
            f = self.neck(self.backbone(x))
            if self.training:
                return self.head(f, labels, x)
            return self.head(f)

    return RefYOLOX(), r


def fvcore_sigmoid_focal_loss(inputs, targets, alpha=-1, gamma=2, reduction="none"):
    """stand-in for fvcore.nn.sigmoid_focal_loss_jit (fvcore is neither vendored in the reference nor installed): the
    RetinaNet focal loss restated from its published formula.  Not pinned to fvcore itself; cross-checked against the one
    independent implementation of the same formula that IS installed (transformers' `sigmoid_focal_loss`,
    tests/test_oracle_golden.py::test_focal_loss_restatement_against_an_independent_implementation)."""
    import torch
    import torch.nn.functional as F
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.sum() if reduction == "sum" else (loss.mean() if reduction == "mean" else loss)


def load_sparseinst():
    """the reference's SparseInst files loaded by path: transcoders/encoder_sparseinst.py, transcoders/decoder_sparseinst.py,
    loss/sparseinst_loss.py.  Un-installed names they import at module level are stubbed: fvcore's weight-init helpers
    (initialisation only), fvcore.nn.sigmoid_focal_loss_jit (`fvcore_sigmoid_focal_loss` above: restated from its published formula, PARITY
    UNPINNED against fvcore, cross-checked against transformers' implementation), detectron2.layers.Conv2d (= nn.Conv2d when no norm / activation is passed, which is how these files
    use it), detectron2.utils.registry.Registry, alfred's logger."""
    load()
    import torch
    from torch import nn
    import torch.nn.functional as F

    class Registry(dict):
        def __init__(self, name=""):
            super().__init__()

        def register(self, obj=None):
            if obj is None:
                return lambda o: self.register(o)
            self[obj.__name__] = obj
            return obj

    def c2_msra_fill(m):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    def c2_xavier_fill(m):
        nn.init.kaiming_uniform_(m.weight, a=1)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    fn = _fvcore(weight_init=dict(c2_msra_fill=c2_msra_fill, c2_xavier_fill=c2_xavier_fill))
    fn.sigmoid_focal_loss_jit = fvcore_sigmoid_focal_loss
    _stub("detectron2.utils.registry", Registry=Registry)
    sys.modules["detectron2.layers"].Conv2d = nn.Conv2d
    if "alfred" not in sys.modules:
        _stub("alfred", print_shape=lambda *a, **k: None)
        _stub("alfred.utils")
        _stub("alfred.utils.log", logger=sys.modules["loguru"].logger)
    y = os.path.join(REF, "yolov7")
    for name, sub in (("yolov7.modeling.transcoders", ("modeling", "transcoders")), ("yolov7.modeling.loss", ("modeling", "loss"))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(y, *sub)]
            sys.modules[name] = m
    out = types.SimpleNamespace()
    out.encoder = importlib.import_module("yolov7.modeling.transcoders.encoder_sparseinst")
    out.decoder = importlib.import_module("yolov7.modeling.transcoders.decoder_sparseinst")
    out.loss = importlib.import_module("yolov7.modeling.loss.sparseinst_loss")
    return out


def load_sparseinst_meta():
    """meta_arch/sparseinst.py (the SparseInst META_ARCH: rescoring_mask, inference) loaded by path on top of
    load_sparseinst().  detectron2.structures.Instances - un-vendored - is a plain attribute bag with image_size here
    (its published behaviour as far as `inference` uses it: construct, set fields)."""
    load_sparseinst()
    load_detr()          # detectron2.structures / detectron2.modeling stubs, yolov7.modeling.meta_arch package

    class Instances:
        def __init__(self, image_size):
            self.image_size = tuple(image_size)

    sys.modules["detectron2.structures"].Instances = Instances
    m = importlib.import_module("yolov7.modeling.meta_arch.sparseinst")
    m.Instances = Instances
    return m


def load_nms_family():
    """the reference's NMS variants loaded by path: meta_arch/utils.py (softnms / cluster_nms / generalized_batched_nms) and
    utils/solov2_utils.py (matrix_nms).  Their un-vendored imports are stubbed: detectron2.layers.nms.batched_nms and
    torchvision.ops.boxes.box_iou are restated here (torchvision semantics; *parity unpinned* for those two, as for NMS
    itself) - the soft / cluster / matrix arithmetic under test is the reference's own."""
    import torch
    load()
    import yolox_oracle as oracle

    def box_iou(b1, b2):      # torchvision.ops.boxes.box_iou
        a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
        a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
        lt = torch.max(b1[:, None, :2], b2[:, :2])
        rb = torch.min(b1[:, None, 2:], b2[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, :, 0] * wh[:, :, 1]
        return inter / (a1[:, None] + a2 - inter)

    _stub("detectron2.layers.nms", batched_nms=oracle.batched_nms)
    sys.modules["torchvision.ops.boxes"].box_iou = box_iou
    sys.modules["torchvision.ops"].boxes = sys.modules["torchvision.ops.boxes"]
    cv2 = sys.modules["cv2"]
    for k in ("INTER_NEAREST", "INTER_LINEAR", "INTER_CUBIC", "INTER_AREA", "INTER_LANCZOS4"):
        if not hasattr(cv2, k):
            setattr(cv2, k, 0)
    y = os.path.join(REF, "yolov7")
    if "yolov7.modeling.meta_arch" not in sys.modules:
        m = types.ModuleType("yolov7.modeling.meta_arch")
        m.__path__ = [os.path.join(y, "modeling", "meta_arch")]
        sys.modules["yolov7.modeling.meta_arch"] = m
    out = types.SimpleNamespace()
    out.nms_utils = importlib.import_module("yolov7.modeling.meta_arch.utils")
    out.solov2_utils = importlib.import_module("yolov7.utils.solov2_utils")
    return out


def load_transforms():
    """yolov7/data/transforms/transform.py loaded by path (YOLOFDistortTransform, YOLOFShiftTransform).  cv2 is not installed:
    cv2.cvtColor is provided by oracle/augment_oracle.py's restatement of OpenCV's 8-bit RGB <-> HSV conversions, so the
    reference's own numpy arithmetic and random draws run as they are (this pins THOSE, not cv2).  fvcore / detectron2's
    Transform base (un-vendored) is a stub with the two members the module uses: `_set_attributes` and `register_type`."""
    load()
    import augment_oracle as A
    cv2 = sys.modules["cv2"]
    cv2.COLOR_RGB2HSV, cv2.COLOR_HSV2RGB = 41, 55
    cv2.cvtColor = lambda img, code: A.rgb2hsv_u8(img) if code == 41 else A.hsv2rgb_u8(img)

    class Transform:
        def _set_attributes(self, params=None):
            for k, v in (params or {}).items():
                if k != "self" and not k.startswith("_"):
                    setattr(self, k, v)

        @classmethod
        def register_type(cls, data_type, func=None):
            return (lambda f: f) if func is None else func
    _stub("fvcore.transforms.transform", Transform=Transform, TransformList=list, NoOpTransform=Transform)
    _stub("fvcore.transforms", transform=sys.modules["fvcore.transforms.transform"])
    if "fvcore" not in sys.modules:
        _stub("fvcore")
    sys.modules["fvcore"].transforms = sys.modules["fvcore.transforms"]
    _stub("detectron2.data.transforms", Transform=Transform, HFlipTransform=type("HFlipTransform", (Transform,), {}),
          VFlipTransform=type("VFlipTransform", (Transform,), {}), ResizeTransform=type("ResizeTransform", (Transform,), {}))
    if "PIL" not in sys.modules:
        try:
            import PIL.Image  # noqa: F401
        except ImportError:
            _stub("PIL", Image=None)
    y = os.path.join(REF, "yolov7")
    for name, sub in (("yolov7.data", ("data",)), ("yolov7.data.transforms", ("data", "transforms"))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(y, *sub)]
            sys.modules[name] = m
    return importlib.import_module("yolov7.data.transforms.transform")


def load_yolov6_loss():
    """head/yolov6_head.py loaded by path for its ComputeLoss (SimOTA + IOUlossV6 + L1).  The EfficientRep `Conv` import
    (unused by the loss) and `alfred.print_shape` (a debug print) are stubbed."""
    from torch import nn
    load()
    _stub_alfred()
    _stub("yolov7.modeling.backbone.efficientrep", Conv=nn.Module)
    return importlib.import_module("yolov7.modeling.head.yolov6_head")


def load_bifpn():
    """neck/bifpn.py loaded by path.  Stubbed module-level imports it never calls on this path: fvcore's weight init,
    detectron2's ResNet builder, the reference's EfficientNet / DLA builders and `MaxPool2d` wrapper (used only by the
    RetinaNet-style LastLevelP6P7).  detectron2.layers.Conv2d is nn.Conv2d when no norm / activation is passed (how bifpn.py
    uses it) and get_norm("GN", C) is nn.GroupNorm(32, C) (detectron2/layers/batch_norm.py, published behaviour)."""
    from torch import nn
    load()
    _fvcore()
    sys.modules["detectron2.layers"].Conv2d = nn.Conv2d
    sys.modules["detectron2.layers.batch_norm"].get_norm = lambda norm, c: nn.GroupNorm(32, c) if norm == "GN" else None
    _stub("detectron2.modeling.backbone.resnet", build_resnet_backbone=None)
    sys.modules["yolov7.modeling.backbone.layers"].MaxPool2d = nn.MaxPool2d
    _stub("yolov7.modeling.backbone.efficientnet", build_efficientnet_backbone=None)
    _stub("yolov7.modeling.backbone.dlafpn", dla34=None)
    m = importlib.import_module("yolov7.modeling.neck.bifpn")
    m.Backbone = sys.modules["detectron2.modeling.backbone"].Backbone
    return m


def load_data_augment():
    """yolov7/data/transforms/data_augment.py loaded by path (random_perspective, box_candidates).  cv2 is not installed:
    the three cv2 functions it calls are provided by oracle/augment_oracle.py's restatement of OpenCV (getRotationMatrix2D
    exactly; warpAffine as the restated fixed-point algorithm) - the reference's own numpy label / matrix code runs as is."""
    load()
    import augment_oracle as A
    cv2 = sys.modules["cv2"]
    cv2.getRotationMatrix2D = lambda angle, center, scale: A.rotation_matrix_2d(angle, scale)
    cv2.warpAffine = lambda img, M, dsize, borderValue=(0, 0, 0): A.warp_affine_u8(img, M, dsize, borderValue[0])
    cv2.INTER_LINEAR = 1
    y = os.path.join(REF, "yolov7")
    for name, sub in (("yolov7.data", ("data",)), ("yolov7.data.transforms", ("data", "transforms"))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(y, *sub)]
            sys.modules[name] = m
    return importlib.import_module("yolov7.data.transforms.data_augment")
