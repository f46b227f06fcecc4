"""ORACLE (test infrastructure, not product): CPU restatement of the reference's DETR set matching.

Only tests/ may import this module.  Restates, in torch fp32 + scipy (test infrastructure may use scipy; the
product may not):

  box_cxcywh_to_xyxy, box_iou, generalized_box_iou      yolov7/utils/boxes.py:28-31,85-122
  HungarianMatcher.forward                              yolov7/utils/detr_utils.py:37-91

`linear_sum_assignment` itself is scipy's (un-vendored dependency of the reference, version unpinned: this image has
scipy 1.15.3); the GPU kernel follows the same published algorithm (shortest augmenting path, fp64 duals).
Pinning: oracle/gen_golden.py runs the reference's own HungarianMatcher (loaded by path) on seeded inputs and stores
its outputs in tests/golden/hungarian.npz; tests/test_oracle_golden.py checks this restatement against them.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment


def box_cxcywh_to_xyxy(x):
    xc, yc, w, h = x.unbind(-1)
    return torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], dim=-1)


def box_iou(b1, b2):
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    return inter / union, union


def generalized_box_iou(b1, b2):
    iou, union = box_iou(b1, b2)
    lt = torch.min(b1[:, None, :2], b2[:, :2])
    rb = torch.max(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


def matching_cost(logits, boxes, targets, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0):
    """[bs, nq, sum(G)] cost matrix exactly as detr_utils.py:58-82 builds it"""
    bs, nq = logits.shape[:2]
    out_prob = logits.flatten(0, 1).softmax(-1)
    out_bbox = boxes.flatten(0, 1)
    tgt_ids = torch.cat([v["labels"] for v in targets])
    tgt_bbox = torch.cat([v["boxes"] for v in targets])
    cc = -out_prob[:, tgt_ids]
    cb = torch.cdist(out_bbox, tgt_bbox, p=1)
    cg = -generalized_box_iou(box_cxcywh_to_xyxy(out_bbox), box_cxcywh_to_xyxy(tgt_bbox))
    C = cost_bbox * cb + cost_class * cc + cost_giou * cg
    return C.view(bs, nq, -1)


def hungarian_match(logits, boxes, targets, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0):
    C = matching_cost(logits, boxes, targets, cost_class, cost_bbox, cost_giou)
    sizes = [len(v["boxes"]) for v in targets]
    idx = [linear_sum_assignment(c[i]) for i, c in enumerate(C.split(sizes, -1))]
    return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)) for i, j in idx], C


def synth_detr(bs, nq, ncls, seed, max_gt=20, sizes=None):
    """seeded DETR-shaped inputs: logits [bs,nq,ncls+1], normalised cxcywh boxes, per-image targets"""
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(bs, nq, ncls + 1, generator=g)
    c = 0.1 + 0.8 * torch.rand(bs, nq, 2, generator=g)
    wh = 0.02 + 0.3 * torch.rand(bs, nq, 2, generator=g)
    boxes = torch.cat([c, wh], -1)
    targets = []
    for b in range(bs):
        n = int(torch.randint(1, max_gt + 1, (1,), generator=g)) if sizes is None else sizes[b]
        tc = 0.1 + 0.8 * torch.rand(n, 2, generator=g)
        twh = 0.02 + 0.3 * torch.rand(n, 2, generator=g)
        targets.append(dict(labels=torch.randint(0, ncls, (n,), generator=g), boxes=torch.cat([tc, twh], -1)))
    return logits, boxes, targets
