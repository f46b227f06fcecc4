"""ORACLE (test infrastructure, not product): CPU restatement of the reference's DETR set matching.

Only tests/ may import this module.  Restates, in torch fp32 + scipy (test infrastructure may use scipy; the
product may not):

  box_cxcywh_to_xyxy, box_iou, generalized_box_iou      yolov7/utils/boxes.py:28-31,85-122
  HungarianMatcher.forward                              yolov7/utils/detr_utils.py:37-91
  SetCriterion (labels / cardinality / boxes losses,    yolov7/modeling/meta_arch/detr.py:475-647
   aux_outputs loop), accuracy                          yolov7/utils/misc.py:212-227

`linear_sum_assignment` itself is scipy's (un-vendored dependency of the reference, version unpinned: this image has
scipy 1.15.3); the GPU kernel follows the same published algorithm (shortest augmenting path, fp64 duals).
Pinning: oracle/gen_golden.py runs the reference's own HungarianMatcher / SetCriterion (loaded by path) on seeded
inputs and stores their outputs in tests/golden/{hungarian,set_criterion}.npz; tests/test_oracle_golden.py checks
this restatement against them.
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


def set_losses(logits, boxes, targets, indices, num_classes, eos_coef, num_boxes, log=True):
    """labels + cardinality + boxes losses of one output level for GIVEN match indices (detr.py:504-556).
    logits / boxes may require grad; returns a dict of 0-d tensors"""
    bs, nq = logits.shape[:2]
    bi = torch.cat([torch.full_like(s_, i) for i, (s_, _) in enumerate(indices)])
    si = torch.cat([s_ for (s_, _) in indices])
    tco = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
    tcls = torch.full((bs, nq), num_classes, dtype=torch.int64)
    tcls[bi, si] = tco
    w = torch.ones(num_classes + 1)
    w[-1] = eos_coef
    # weighted cross entropy, 'mean' reduction = sum(w_t * nll) / sum(w_t)   (detr.py:518)
    lsm = torch.log_softmax(logits, -1)
    nll = -lsm.gather(-1, tcls[..., None])[..., 0]
    wt = w[tcls]
    out = {"loss_ce": (wt * nll).sum() / wt.sum()}
    if log:
        if tco.numel() == 0:
            out["class_error"] = torch.tensor(100.0)
        else:
            out["class_error"] = 100 - 100.0 * (logits[bi, si].argmax(-1) == tco).float().sum() / tco.numel()
    lens = torch.tensor([len(t["labels"]) for t in targets], dtype=torch.float32)
    card = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1).float()
    out["cardinality_error"] = (card - lens).abs().mean()
    sb = boxes[bi, si]
    tb = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)], 0)
    out["loss_bbox"] = (sb - tb).abs().sum() / num_boxes
    g = torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(sb), box_cxcywh_to_xyxy(tb)))
    out["loss_giou"] = (1 - g).sum() / num_boxes
    return out


def set_criterion(outputs, targets, num_classes, eos_coef, cost_class=1.0, cost_bbox=5.0, cost_giou=2.0, world_size=1):
    """SetCriterion.forward (detr.py:599-647) with losses ['labels', 'boxes', 'cardinality']: match the last level and
    every aux level independently; aux losses get the suffix _i and no class_error"""
    num_boxes = max(float(sum(len(t["labels"]) for t in targets)) / world_size, 1.0)
    idx, _ = hungarian_match(outputs["pred_logits"].detach(), outputs["pred_boxes"].detach(), targets, cost_class,
                             cost_bbox, cost_giou)
    losses = set_losses(outputs["pred_logits"], outputs["pred_boxes"], targets, idx, num_classes, eos_coef, num_boxes)
    for i, aux in enumerate(outputs.get("aux_outputs", [])):
        idx, _ = hungarian_match(aux["pred_logits"].detach(), aux["pred_boxes"].detach(), targets, cost_class, cost_bbox,
                                 cost_giou)
        l_ = set_losses(aux["pred_logits"], aux["pred_boxes"], targets, idx, num_classes, eos_coef, num_boxes, log=False)
        losses.update({k + f"_{i}": v for k, v in l_.items()})
    return losses


def synth_detr_levels(bs, nq, ncls, seed, levels=3, sizes=None):
    """seeded decoder-level outputs (last + aux) sharing one target set"""
    logits, boxes, targets = synth_detr(bs, nq, ncls, seed, sizes=sizes)
    outs = []
    for lv in range(levels):
        l2, b2, _ = synth_detr(bs, nq, ncls, seed + 1000 * (lv + 1), sizes=[1] * bs)
        outs.append((0.5 * logits + l2, (0.5 * boxes + 0.5 * b2)))
    return outs, targets
