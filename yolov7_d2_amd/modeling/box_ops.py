"""Box utilities of the DETR path behind the reference's names (yolov7/utils/boxes.py:28-37,85-122): HIP entries
mi_box_convert / mi_box_iou_pairwise (csrc/detr_ops.hip, the reference's float operation order).  The training step does
not call them (matching cost and GIoU loss are fused inside mi_hungarian_match / mi_detr_set_loss_*); they serve target
preparation, inference and callers of the reference's API.  Device tensors only: there is no CPU fallback."""
import torch

from .. import _lib as L


def _dev4(x, what):
    if not x.is_cuda:
        raise L.MI355Error(f"{what}: the MI355X path needs device tensors (no CPU fallback)")
    if x.shape[-1] != 4:
        raise ValueError(f"{what}: boxes [..., 4] expected, got {tuple(x.shape)}")
    return x.to(torch.float32).contiguous()


def _convert(x, to_cxcywh, what):
    a = _dev4(x, what)
    out = torch.empty_like(a)
    L.check(L.lib().mi_box_convert(a.data_ptr(), out.data_ptr(), a.numel() // 4, to_cxcywh, L.stream_ptr()), what)
    return out.to(x.dtype) if x.dtype != torch.float32 else out


def box_cxcywh_to_xyxy(x):
    return _convert(x, 0, "box_cxcywh_to_xyxy")


def box_xyxy_to_cxcywh(x):
    return _convert(x, 1, "box_xyxy_to_cxcywh")


def box_area(boxes):
    b = box_xyxy_to_cxcywh(boxes)
    return b[:, 2] * b[:, 3]


def _pairwise(boxes1, boxes2, want_giou):
    a, b = _dev4(boxes1, "box_iou"), _dev4(boxes2, "box_iou")
    n, m = a.shape[0], b.shape[0]
    iou = torch.empty((n, m), dtype=torch.float32, device=a.device)
    uni = torch.empty_like(iou)
    giou = torch.empty_like(iou) if want_giou else None
    flag = torch.zeros(1, dtype=torch.int32, device=a.device) if want_giou else None
    L.check(L.lib().mi_box_iou_pairwise(a.data_ptr(), n, b.data_ptr(), m, iou.data_ptr(), uni.data_ptr(),
                                        giou.data_ptr() if want_giou else None, flag.data_ptr() if want_giou else None,
                                        L.stream_ptr()), "box_iou_pairwise")
    return iou, uni, giou, flag


def box_iou(boxes1, boxes2):
    """-> (iou [N, M], union [N, M])"""
    iou, uni, _, _ = _pairwise(boxes1, boxes2, False)
    return iou, uni


def generalized_box_iou(boxes1, boxes2):
    """[N, M] GIoU of xyxy boxes; degenerate boxes raise like the reference's assertion (the same host sync)"""
    _, _, giou, flag = _pairwise(boxes1, boxes2, True)
    f = int(flag.item()) if boxes1.shape[0] and boxes2.shape[0] else 0
    assert not (f & 1), "generalized_box_iou: boxes1 has x1 < x0 or y1 < y0"
    assert not (f & 2), "generalized_box_iou: boxes2 has x1 < x0 or y1 < y0"
    return giou
