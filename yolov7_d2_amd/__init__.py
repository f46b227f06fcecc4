"""yolov7_d2_amd — MI355X-native hot path of yolov7_d2's YOLOX data-parallel training step.

Public surface mirrors the reference: META_ARCH `YOLOX`, backbone builder
`build_cspdarknetx_backbone`, `postprocess` / `batched_nms`; compute lives in libmi355det.so
(include/mi355_det.h).  Importing registers the classes into the (detectron2 or shim) registries.
"""
from . import _lib
from .config import add_sparse_inst_config, add_yolo_config, detr_r50_cfg, sparse_inst_r50_giam_cfg, get_cfg, get_yolox_cfg, yolox_s_cfg
from .d2shim import BACKBONE_REGISTRY, META_ARCH_REGISTRY, build_backbone, build_model
from .modeling import YOLOX, Detr, SparseInst, build_cspdarknetx_backbone, build_resnet_backbone, batched_nms, postprocess
from . import ops  # noqa: F401  (registers torch.ops.mi355.*)
from .ops import patch_base_convs
from .export_onnx import export_yolox_onnx, export_sparseinst_onnx, export_detr_onnx, export_onnx

__all__ = ["export_yolox_onnx", "export_sparseinst_onnx", "export_detr_onnx", "export_onnx", "YOLOX", "build_cspdarknetx_backbone", "batched_nms", "postprocess", "build_model", "build_backbone",
           "patch_base_convs", "get_cfg", "add_yolo_config", "get_yolox_cfg", "yolox_s_cfg", "META_ARCH_REGISTRY", "BACKBONE_REGISTRY"]
