"""IOUlossV6 — drop-in for yolov7/utils/boxes.py:666-752 (CIoU / DIoU / GIoU / SIoU box loss of the YOLOv6 head):
same constructor and call convention (box1 = predictions TRANSPOSED [4, N] exactly as yolov6_head.py:512 passes
them, box2 = targets [N, 4]), differentiable with respect to box1; forward and gradient come from one HIP kernel."""
import torch

from .. import _lib as L

_TYPES = {"iou": 0, "giou": 1, "diou": 2, "ciou": 3, "siou": 4}


class _IouLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_n4, target, iou_type, xyxy, eps):
        if not pred_n4.is_cuda:
            raise L.MI355Error("IOUlossV6: the MI355X path needs device tensors (no CPU fallback)")
        p = pred_n4.detach().float().contiguous()
        t = target.detach().float().contiguous()
        n = p.shape[0]
        loss = torch.empty(n, device=p.device)
        dpred = torch.empty(n, 4, device=p.device)
        L.check(L.lib().mi_iou_loss_v6(p.data_ptr(), t.data_ptr(), n, iou_type, xyxy, eps, None, loss.data_ptr(),
                                       dpred.data_ptr(), L.stream_ptr()), "mi_iou_loss_v6")
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g.unsqueeze(1), None, None, None, None


class IOUlossV6:
    def __init__(self, box_format="xywh", iou_type="ciou", reduction="none", eps=1e-7):
        self.box_format, self.iou_type, self.reduction, self.eps = box_format, iou_type.lower(), reduction, eps

    def __call__(self, box1, box2):
        loss = _IouLossFn.apply(box1.T, box2, _TYPES[self.iou_type], 1 if self.box_format == "xyxy" else 0, self.eps)
        if self.reduction == "sum":
            return loss.sum()
        if self.reduction == "mean":
            return loss.mean()
        return loss


class _YoloxIouLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, loss_type):
        if not pred.is_cuda:
            raise L.MI355Error("IOUloss: the MI355X path needs device tensors (no CPU fallback)")
        p = pred.detach().float().contiguous()
        t = target.detach().float().contiguous()
        n = p.shape[0]
        loss = torch.empty(n, device=p.device)
        dpred = torch.empty(n, 4, device=p.device)
        L.check(L.lib().mi_yolox_iou_loss(p.data_ptr(), t.data_ptr(), n, loss_type, None, loss.data_ptr(), dpred.data_ptr(),
                                          L.stream_ptr()), "mi_yolox_iou_loss")
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g.unsqueeze(1), None, None


class IOUloss(torch.nn.Module):
    """drop-in for yolov7/utils/boxes.py:125-168 (the YOLOX head's box loss as a module): pred / target (cx, cy, w, h),
    loss_type "iou" (1 - iou^2) or "giou", reduction none / mean / sum; differentiable with respect to pred."""

    def __init__(self, reduction="none", loss_type="iou"):
        super().__init__()
        if loss_type not in ("iou", "giou"):
            raise ValueError(f"IOUloss: loss_type {loss_type!r} (the reference implements 'iou' and 'giou')")
        self.reduction, self.loss_type = reduction, loss_type

    def forward(self, pred, target):
        assert pred.shape[0] == target.shape[0]
        loss = _YoloxIouLossFn.apply(pred.view(-1, 4), target.view(-1, 4), 0 if self.loss_type == "iou" else 1)
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss


def pairwise_bbox_iou(box1, box2, box_format="xywh"):
    """utils/boxes.py:755-779 -> [N, M] (no gradient: the reference calls it inside torch.no_grad() assignment code)"""
    if box_format not in ("xywh", "xyxy"):
        raise ValueError(box_format)
    if not box1.is_cuda:
        raise L.MI355Error("pairwise_bbox_iou: the MI355X path needs device tensors (no CPU fallback)")
    a, b = box1.detach().float().contiguous(), box2.detach().float().contiguous()
    out = torch.empty(a.shape[0], b.shape[0], device=a.device)
    L.check(L.lib().mi_pairwise_bbox_iou(a.data_ptr(), b.data_ptr(), a.shape[0], b.shape[0], int(box_format == "xyxy"),
                                         out.data_ptr(), L.stream_ptr()), "mi_pairwise_bbox_iou")
    return out


def bboxes_iou(bboxes_a, bboxes_b, xyxy=True):
    """utils/boxes.py:57-81"""
    if bboxes_a.shape[1] != 4 or bboxes_b.shape[1] != 4:
        raise IndexError
    return pairwise_bbox_iou(bboxes_a, bboxes_b, "xyxy" if xyxy else "xywh")
