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
