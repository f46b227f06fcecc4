"""HungarianMatcher — drop-in for yolov7/utils/detr_utils.py:12-91 (DETR set-prediction matching, config 4).

Same constructor (cost_class, cost_bbox, cost_giou), same forward(outputs, targets) contract and return value
(list of (index_i, index_j) int64 tensor pairs, rows sorted) — but the cost matrix AND the linear sum assignment
run on the GPU (libmi355det: mi_hungarian_match); the reference copies the cost matrix to the host and calls
scipy.optimize.linear_sum_assignment per image (detr_utils.py:84-90), six times per step.
"""
import torch
from torch import nn

from .. import _lib as L


class PackedTargets(list):
    """the ground truth of a batch in the criterion's DEVICE layout, with a fixed capacity per image: a list of the
    reference's {"labels", "boxes"} dicts (so foreign code still sees detr.py:199-213's targets) that also carries
    tgt_off int32 [B + 1], tgt_labels int64 [B * cap], tgt_boxes fp32 [B * cap, 4] (compact, indexed through tgt_off) and
    inv_num_boxes fp32 [1] = 1 / max(sum of boxes / world, 1).  Nothing in it lives on the host: a step captured as a
    hipGraph reads the SAME tensors for every batch (Detr.prepare_batch refills them in place)."""

    def __init__(self, dicts, cap, tgt_off, tgt_labels, tgt_boxes, inv_num_boxes, levels=1):
        super().__init__(dicts)
        self.cap, self.tgt_off, self.tgt_labels, self.tgt_boxes, self.inv_num_boxes = cap, tgt_off, tgt_labels, tgt_boxes, inv_num_boxes
        # the same ground truth `levels` times over (level l's copy starts where level l - 1's ends): what ONE matching launch
        # over (decoder level, image) pairs indexes - HungarianMatcher.match_device_levels.  Filled by Detr.prepare_batch.
        self.lv = None
        if levels > 1:
            B, dev = len(dicts), tgt_off.device
            self.lv = dict(n=levels, off=torch.zeros(levels * B + 1, dtype=torch.int32, device=dev),
                           labels=torch.zeros(levels * B * cap, dtype=torch.int64, device=dev),
                           boxes=torch.zeros(levels * B * cap, 4, dtype=torch.float32, device=dev))

    def fill_levels(self, off, labels, boxes):
        """host lists of one batch (prefix offsets, concatenated labels / boxes or None) -> the replicated device arrays"""
        if self.lv is None:
            return
        n, B, ntot = self.lv["n"], len(self), off[-1]
        from ..ops import HostRing
        HostRing.upload(self.lv["off"], [l * ntot + off[b] for l in range(n) for b in range(B)] + [n * ntot])
        if ntot:
            HostRing.upload(self.lv["labels"][:n * ntot], labels.repeat(n))
            HostRing.upload(self.lv["boxes"][:n * ntot], boxes.repeat(n, 1))


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def match_device(self, outputs, targets):
        """the matching without any host round trip: returns a dict of device tensors (match_q / match_t int64
        [B][gmax], nmatch int32 [B], tgt_off int32 [B+1], concatenated tgt_labels / tgt_boxes, gmax) in the layout
        mi_detr_set_loss_* consumes"""
        logits = outputs["pred_logits"].detach().float().contiguous()
        boxes = outputs["pred_boxes"].detach().float().contiguous()
        if not logits.is_cuda:
            raise L.MI355Error("HungarianMatcher: the MI355X path needs device tensors (no CPU fallback)")
        bs, nq, nc = logits.shape
        dev = logits.device
        if isinstance(targets, PackedTargets):     # device-resident, fixed capacity: no host value enters the launch
            gmax, off, tl, tb, ntot = targets.cap, targets.tgt_off, targets.tgt_labels, targets.tgt_boxes, -1
        else:
            sizes = [len(v["boxes"]) for v in targets]
            gmax, ntot = max(max(sizes), 1), sum(sizes)
            off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=dev)
            tl = torch.cat([v["labels"] for v in targets]).to(dev, torch.int64).contiguous()
            tb = torch.cat([v["boxes"] for v in targets]).to(dev, torch.float32).contiguous()
            if tl.numel() == 0:   # no targets at all: keep the pointers valid
                tl, tb = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, 4, device=dev)
        cost = torch.empty(bs, nq, gmax, device=dev)
        mq = torch.empty(bs, gmax, dtype=torch.int64, device=dev)
        mt = torch.empty(bs, gmax, dtype=torch.int64, device=dev)
        nm = torch.zeros(bs, dtype=torch.int32, device=dev)
        L.check(L.lib().mi_hungarian_match(logits.data_ptr(), boxes.data_ptr(), tl.data_ptr(), tb.data_ptr(),
                                           off.data_ptr(), bs, nq, nc, gmax, float(self.cost_class),
                                           float(self.cost_bbox), float(self.cost_giou), cost.data_ptr(), mq.data_ptr(),
                                           mt.data_ptr(), nm.data_ptr(), L.stream_ptr()), "mi_hungarian_match")
        self.last_cost = cost
        return dict(match_q=mq, match_t=mt, nmatch=nm, tgt_off=off, tgt_labels=tl, tgt_boxes=tb, gmax=gmax,
                    num_targets=ntot)

    @torch.no_grad()
    def match_device_levels(self, levels, targets):
        """match_device for ALL decoder levels of a step in one launch pair: the (level, image) pairs are the batch of one
        mi_hungarian_match call over the level-replicated ground truth (PackedTargets.lv) - one cost launch and one
        assignment launch of levels x B blocks instead of `levels` launches of B blocks each (the assignment kernel is one
        wave per image and ~65 us long: six of them in series were 0.4 ms of a DETR step).  Returns one dict per level in
        match_device's layout (row blocks of the shared match arrays; targets indexed through the ORIGINAL tgt_off)."""
        lv = targets.lv
        n = len(levels)
        assert lv is not None and lv["n"] == n
        logits = torch.stack([x["pred_logits"].detach() for x in levels]).float().contiguous()
        boxes = torch.stack([x["pred_boxes"].detach() for x in levels]).float().contiguous()
        if not logits.is_cuda:
            raise L.MI355Error("HungarianMatcher: the MI355X path needs device tensors (no CPU fallback)")
        _, bs, nq, nc = logits.shape
        dev, gmax = logits.device, targets.cap
        cost = torch.empty(n * bs, nq, gmax, device=dev)
        mq = torch.empty(n * bs, gmax, dtype=torch.int64, device=dev)
        mt = torch.empty(n * bs, gmax, dtype=torch.int64, device=dev)
        nm = torch.zeros(n * bs, dtype=torch.int32, device=dev)
        L.check(L.lib().mi_hungarian_match(logits.data_ptr(), boxes.data_ptr(), lv["labels"].data_ptr(), lv["boxes"].data_ptr(),
                                           lv["off"].data_ptr(), n * bs, nq, nc, gmax, float(self.cost_class),
                                           float(self.cost_bbox), float(self.cost_giou), cost.data_ptr(), mq.data_ptr(),
                                           mt.data_ptr(), nm.data_ptr(), L.stream_ptr()), "mi_hungarian_match (all levels)")
        self.last_cost = cost[:bs]
        return [dict(match_q=mq[l * bs:(l + 1) * bs], match_t=mt[l * bs:(l + 1) * bs], nmatch=nm[l * bs:(l + 1) * bs],
                     tgt_off=targets.tgt_off, tgt_labels=targets.tgt_labels, tgt_boxes=targets.tgt_boxes, gmax=gmax,
                     num_targets=-1) for l in range(n)]

    @torch.no_grad()
    def forward(self, outputs, targets):
        m = self.match_device(outputs, targets)
        n = m["nmatch"].tolist()   # one small D2H (the reference moves the whole cost matrix instead)
        if any(v < 0 for v in n):  # the kernel's flag for NaN / inf costs: scipy raises ValueError in the same situation
            raise ValueError("matrix contains invalid numeric entries")
        return [(m["match_q"][b, : n[b]].clone(), m["match_t"][b, : n[b]].clone()) for b in range(len(n))]
