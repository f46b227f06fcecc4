"""SetCriterion — drop-in for yolov7/modeling/meta_arch/detr.py:475-647 (the DETR training loss, config 4).

Same constructor (num_classes, matcher, weight_dict, eos_coef, losses) and forward(outputs, targets) contract; returns
the reference's dict (loss_ce, class_error, cardinality_error, loss_bbox, loss_giou and the `_i` aux variants) with
autograd attached to pred_logits / pred_boxes.  Per decoder level the whole criterion is four launches on the device:
mi_hungarian_match (cost matrix + assignment) and mi_detr_set_loss_fwd (rows + reduction); the backward is one launch
(mi_detr_set_loss_bwd).  With our HungarianMatcher nothing is copied to the host; the reference copies the cost
matrix out and runs scipy per image, per level.

'masks' (the segmentation variant's focal / dice losses, detr.py:558-583) is out of this path's scope and raises.
"""
import ctypes as C

import torch
from torch import nn

from .. import _lib as L
from .detr_matcher import HungarianMatcher, PackedTargets


class _SetLossFn(torch.autograd.Function):
    """(logits, boxes) -> tensor[5] = loss_ce, class_error, cardinality_error, loss_bbox, loss_giou"""

    @staticmethod
    def forward(ctx, logits, boxes, m, eos_coef, num_boxes):
        lg = logits.detach().float().contiguous()
        bx = boxes.detach().float().contiguous()
        B, Q, NC = lg.shape
        dev = lg.device
        losses = torch.empty(8, dtype=torch.float32, device=dev)
        rows = torch.empty(B * Q * 16, dtype=torch.float32, device=dev)
        d = L.mi_detr_loss_desc()
        d.logits, d.boxes = lg.data_ptr(), bx.data_ptr()
        d.tgt_labels, d.tgt_boxes, d.tgt_off = m["tgt_labels"].data_ptr(), m["tgt_boxes"].data_ptr(), m["tgt_off"].data_ptr()
        d.match_q, d.match_t, d.nmatch = m["match_q"].data_ptr(), m["match_t"].data_ptr(), m["nmatch"].data_ptr()
        d.B, d.Q, d.NC, d.gmax = B, Q, NC, m["gmax"]
        d.eos_coef, d.num_boxes = float(eos_coef), float(num_boxes)
        d.losses, d.rowstate = losses.data_ptr(), rows.data_ptr()
        L.check(L.lib().mi_detr_set_loss_fwd(C.byref(d), L.stream_ptr()), "mi_detr_set_loss_fwd")
        ctx.desc, ctx.keep = d, (lg, bx, m, losses, rows)
        ctx.in_dtypes = (logits.dtype, boxes.dtype)
        return losses[:5].clone()

    @staticmethod
    def backward(ctx, g):
        lg, bx = ctx.keep[0], ctx.keep[1]
        gw = torch.stack((g[0], g[3], g[4])).float().contiguous()     # (no index tensor: nothing is copied from the host)
        dl, db = torch.empty_like(lg), torch.empty_like(bx)
        L.check(L.lib().mi_detr_set_loss_bwd(C.byref(ctx.desc), gw.data_ptr(), dl.data_ptr(), db.data_ptr(),
                                             L.stream_ptr()), "mi_detr_set_loss_bwd")
        return dl.to(ctx.in_dtypes[0]), db.to(ctx.in_dtypes[1]), None, None, None


def _pack_indices(indices, targets, dev):
    """(index_i, index_j) lists from a foreign matcher -> the padded device layout"""
    sizes = [len(v["boxes"]) for v in targets]
    gmax = max(max(sizes), 1)
    B = len(targets)
    mq = torch.zeros(B, gmax, dtype=torch.int64)
    mt = torch.zeros(B, gmax, dtype=torch.int64)
    nm = torch.zeros(B, dtype=torch.int32)
    for b, (i, j) in enumerate(indices):
        n = len(i)
        mq[b, :n], mt[b, :n], nm[b] = i.cpu(), j.cpu(), n
    off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32)
    tl = torch.cat([v["labels"] for v in targets]).to(torch.int64)
    tb = torch.cat([v["boxes"] for v in targets]).to(torch.float32)
    if tl.numel() == 0:
        tl, tb = torch.zeros(1, dtype=torch.int64), torch.zeros(1, 4)
    return dict(match_q=mq.to(dev), match_t=mt.to(dev), nmatch=nm.to(dev), tgt_off=off.to(dev),
                tgt_labels=tl.to(dev).contiguous(), tgt_boxes=tb.to(dev).contiguous(), gmax=gmax, num_targets=sum(sizes))


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict, self.eos_coef, self.losses = (num_classes, matcher, weight_dict,
                                                                                       eos_coef, losses)
        for l_ in losses:
            if l_ == "masks":
                raise NotImplementedError("SetCriterion: 'masks' losses (DETR segmentation variant) are out of scope")
            assert l_ in ("labels", "cardinality", "boxes"), f"do you really want to compute {l_} loss?"
        empty_weight = torch.ones(num_classes + 1)
        empty_weight[-1] = eos_coef
        self.register_buffer("empty_weight", empty_weight)   # same buffer as the reference (state_dict compatible)

    def _match(self, outputs, targets):
        if isinstance(self.matcher, HungarianMatcher):
            return self.matcher.match_device(outputs, targets)
        return _pack_indices(self.matcher(outputs, targets), targets, outputs["pred_logits"].device)

    def _level(self, outputs, targets, num_boxes, log, inv_nb=None):
        if not outputs["pred_logits"].is_cuda:
            raise L.MI355Error("SetCriterion: the MI355X path needs device tensors (no CPU fallback)")
        if outputs["pred_logits"].shape[-1] != self.num_classes + 1:
            raise ValueError("pred_logits must have num_classes + 1 channels")
        m = self._match(outputs, targets)
        v = _SetLossFn.apply(outputs["pred_logits"], outputs["pred_boxes"], m, self.eos_coef, num_boxes)
        out = {}
        if "labels" in self.losses:
            out["loss_ce"] = v[0]
            if log:
                out["class_error"] = v[1].detach()
        if "cardinality" in self.losses:
            out["cardinality_error"] = v[2].detach()
        if "boxes" in self.losses:
            out["loss_bbox"], out["loss_giou"] = v[3], v[4]
            if inv_nb is not None:     # the kernels ran with num_boxes = 1: both sums are linear in 1 / num_boxes
                out["loss_bbox"], out["loss_giou"] = v[3] * inv_nb[0], v[4] * inv_nb[0]
        return out

    def weighted_packed(self, outputs, targets):
        """The PackedTargets path for a captured step, with the scalar bookkeeping of all decoder levels as a few vector
        ops: -> the reference's loss dict ALREADY multiplied by weight_dict (what Detr.forward returns, detr.py:262-267) plus
        "total" = the sum of the weighted entries.  Level by level (`v[3] * inv`, `* weight_dict[k]`, 17 additions and their
        backward nodes) the same arithmetic is ~100 one-element launches per DETR step."""
        inv = targets.inv_num_boxes
        levels = [{k: v for k, v in outputs.items() if k != "aux_outputs"}] + list(outputs.get("aux_outputs", []))
        for lv in levels:
            if not lv["pred_logits"].is_cuda:
                raise L.MI355Error("SetCriterion: the MI355X path needs device tensors (no CPU fallback)")
            if lv["pred_logits"].shape[-1] != self.num_classes + 1:
                raise ValueError("pred_logits must have num_classes + 1 channels")
        lvt = getattr(targets, "lv", None)
        import os
        if (isinstance(self.matcher, HungarianMatcher) and lvt is not None and lvt["n"] == len(levels)
                and os.environ.get("MI_DETR_MATCH_LEVELS", "1") == "1"):       # (=0: one matching per level, A/B switch)
            ms = self.matcher.match_device_levels(levels, targets)          # all levels: one cost + one assignment launch
        else:
            ms = [self._match(lv, targets) for lv in levels]
        vs = [_SetLossFn.apply(lv["pred_logits"], lv["pred_boxes"], m, self.eos_coef, 1.0) for lv, m in zip(levels, ms)]
        V = torch.stack(vs)                                               # [levels, 5]: ce, class_error, cardinality, bbox, giou
        names = ("loss_ce", "class_error", "cardinality_error", "loss_bbox", "loss_giou")
        need = ("labels", "labels", "cardinality", "boxes", "boxes")
        key = (len(levels), V.device)
        c = self.__dict__.get("_packed_consts")
        if c is None or c[0] != key:
            sfx = [""] + [f"_{i}" for i in range(len(levels) - 1)]
            w = torch.tensor([[float(self.weight_dict.get(n + s_, 1.0)) for n in names] for s_ in sfx])
            m = torch.tensor([[float(n + s_ in self.weight_dict and q in self.losses) for n, q in zip(names, need)] for s_ in sfx])
            a, b = torch.tensor([1.0, 1.0, 1.0, 0.0, 0.0]), torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0])
            c = self.__dict__["_packed_consts"] = (key, w.to(V.device), m.to(V.device), a.to(V.device), b.to(V.device), sfx)
        _, w, m, a, b, sfx = c
        R = V * (w * torch.addcmul(a, b, inv[0]))                         # bbox / giou sums ran with num_boxes = 1 (linear in 1 / n)
        out = {}
        for l_, s_ in enumerate(sfx):
            for j, (n, q) in enumerate(zip(names, need)):
                if q in self.losses and (n != "class_error" or l_ == 0):
                    out[n + s_] = R[l_, j] if n.startswith("loss_") else R[l_, j].detach()
        out["total"] = (R * m).sum()
        return out

    def forward(self, outputs, targets):
        if isinstance(targets, PackedTargets):
            # device-resident targets (Detr.prepare_batch): 1 / num_boxes is a device scalar (already averaged over the
            # ranks there), so that a captured step serves batches with any number of boxes
            inv = targets.inv_num_boxes
            losses = self._level({k: v for k, v in outputs.items() if k != "aux_outputs"}, targets, 1.0, True, inv)
            for i, aux in enumerate(outputs.get("aux_outputs", [])):
                losses.update({k + f"_{i}": v for k, v in self._level(aux, targets, 1.0, False, inv).items()})
            return losses
        num_boxes = float(sum(len(t["labels"]) for t in targets))
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():   # detr.py:616-619
            nb = torch.tensor([num_boxes], dtype=torch.float, device=outputs["pred_logits"].device)
            torch.distributed.all_reduce(nb)
            world = torch.distributed.get_world_size()
            num_boxes = float(nb.item())
        num_boxes = max(num_boxes / world, 1.0)
        losses = self._level({k: v for k, v in outputs.items() if k != "aux_outputs"}, targets, num_boxes, True)
        for i, aux in enumerate(outputs.get("aux_outputs", [])):
            losses.update({k + f"_{i}": v for k, v in self._level(aux, targets, num_boxes, False).items()})
        return losses
