"""ComputeLoss of the YOLOv6 head — drop-in for yolov7/modeling/head/yolov6_head.py:315-531 (SimOTA assignment with
configurable centre radius / cost weights, IOUlossV6 box loss, L1 on the raw box outputs, BCE objectness and class
losses).  Same constructor and call convention: `outputs` is the list of per-level head outputs [B, n_anchors, h, w,
5 + nc] (raw, undecoded), `targets` [B, max_boxes, 5] = (class, cx, cy, w, h) NORMALISED to the input size (the reference
scales them by (W, H, W, H) in place, :410-412; so does this).  Returns (total_loss, tensor([reg_weight * iou, l1, obj,
cls]).detach()) like the reference; total_loss is differentiable with respect to the head outputs.

It is the YOLOX head's loss with other constants, so it runs on the same HIP kernels (csrc/yolox_loss.hip):
assignment, the four losses and the gradient with respect to the raw predictions."""
import ctypes as C

import torch

from .. import _lib as L
from .iou_loss import _TYPES


class _V6LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, labels, anchors, cfg):
        B, A, nch = raw.shape
        ML = labels.shape[1]
        dev = raw.device
        t = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        rd, ld, ad = raw.detach().float().contiguous(), labels.detach().float().contiguous(), anchors.float().contiguous()
        nb = B * ((A + 255) // 256)
        ws = dict(cost=t(B, ML, A), iou=t(B, ML, A), match=t(B, ML, A, dt=torch.uint8), ngt=t(B, dt=torch.int32),
                  fg=t(B, A, dt=torch.uint8), matched_gt=t(B, A, dt=torch.int32), matched_iou=t(B, A), partial=t(nb, 4),
                  out=t(8), partial_l1=t(nb))
        d = L.mi_yolox_loss_desc()
        d.preds, d.labels, d.anchors = rd.data_ptr(), ld.data_ptr(), ad.data_ptr()
        d.B, d.A, d.ncls, d.max_labels, d.gmax = B, A, nch - 5, ML, ML
        for k, v in ws.items():
            setattr(d, k, v.data_ptr())
        d.use_l1 = 1
        d.center_radius, d.cls_weight, d.iou_weight, d.reg_weight, d.iou_type = cfg
        L.check(L.lib().mi_yolox_loss_fwd(C.byref(d), L.stream_ptr()), "mi_yolox_loss_fwd (yolov6)")
        ctx.keep = (d, ws, rd, ld, ad)
        ctx.shape = (B, A, nch)
        return ws["out"][:6].clone()

    @staticmethod
    def backward(ctx, g):
        d, ws, rd, ld, ad = ctx.keep
        B, A, nch = ctx.shape
        # g: upstream of (total, reg_weight * iou, obj, cls, l1, num_fg ratio) -> the kernel's (total, iou, obj, cls, l1)
        gw = g[:5].float().contiguous()
        dpreds = torch.empty(B, A, nch, dtype=torch.float32, device=rd.device)
        L.check(L.lib().mi_yolox_loss_bwd(C.byref(d), gw.data_ptr(), dpreds.data_ptr(), L.stream_ptr()),
                "mi_yolox_loss_bwd (yolov6)")
        return dpreds, None, None, None


class ComputeLoss:
    def __init__(self, reg_weight=5.0, iou_weight=3.0, cls_weight=1.0, center_radius=2.5, eps=1e-7,
                 in_channels=[256, 512, 1024], strides=[8, 16, 32], n_anchors=1, iou_type="ciou"):
        if n_anchors != 1:
            raise NotImplementedError("ComputeLoss: n_anchors = 1 (what EffiDeHead builds, yolov6_head.py:25-150)")
        if iou_type.lower() not in ("giou", "diou", "ciou", "siou"):
            raise ValueError(f"iou_type {iou_type!r}")
        if eps != 1e-7:
            raise NotImplementedError("ComputeLoss: eps is not forwarded to IOUlossV6 by the reference either (1e-7)")
        self.reg_weight, self.iou_weight, self.cls_weight = reg_weight, iou_weight, cls_weight
        self.center_radius, self.eps, self.n_anchors, self.strides = center_radius, eps, n_anchors, list(strides)
        self.iou_type = iou_type.lower()
        self._anchors = {}

    def anchors_for(self, hw, device):
        key = (tuple(hw), str(device))
        a = self._anchors.get(key)
        if a is None:
            rows = []
            for (h, w), s in zip(hw, self.strides):
                yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
                rows.append(torch.stack([xv.reshape(-1).float(), yv.reshape(-1).float(), torch.full((h * w,), float(s))], 1))
            a = torch.cat(rows, 0).to(device)
            self._anchors[key] = a
        return a

    def __call__(self, outputs, targets):
        if not outputs[0].is_cuda:
            raise L.MI355Error("ComputeLoss: the MI355X path needs device tensors (no CPU fallback)")
        B = outputs[0].shape[0]
        hw = [tuple(o.shape[2:4]) for o in outputs]
        raw = torch.cat([o.reshape(B, -1, o.shape[-1]) for o in outputs], 1)          # [B, A, 5 + nc]
        feat_h, feat_w = hw[-1][0] * self.strides[-1], hw[-1][1] * self.strides[-1]
        scale = torch.tensor([[feat_w, feat_h, feat_w, feat_h]], dtype=targets.dtype, device=targets.device)
        ngt = (targets.sum(dim=2) > 0).sum(dim=1)
        for b in range(B):                                    # yolov6_head.py:410-412: scaled IN PLACE, valid rows only
            n = int(ngt[b])
            if n:
                targets[b, :n, 1:5].mul_(scale)
        out = _V6LossFn.apply(raw, targets, self.anchors_for(hw, raw.device),
                              (self.center_radius, self.cls_weight, self.iou_weight, self.reg_weight, _TYPES[self.iou_type]))
        return out[0], torch.stack([out[1], out[4], out[2], out[3]]).detach()
