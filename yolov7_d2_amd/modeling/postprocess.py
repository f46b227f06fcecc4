"""Eval post-processing: `postprocess` of yolov7/utils/boxes.py:171-210 with the NMS on the GPU
(libmi355det mi_batched_nms, torchvision.ops.batched_nms semantics)."""
import ctypes as C

import torch

from .. import _lib as L


def batched_nms(boxes, scores, idxs, iou_threshold):
    """drop-in for torchvision.ops.batched_nms(boxes[n,4] xyxy, scores[n], idxs[n], thr) -> int64 keep
    (descending score).  HIP tensors only."""
    if not boxes.is_cuda:
        raise L.MI355Error("batched_nms: HIP tensors required (no CPU fallback; the CPU oracle is oracle/nms_ref)")
    n = boxes.shape[0]
    dev = boxes.device
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dev)
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    idxs = idxs.contiguous().float()
    nw = (n + 63) // 64
    order = torch.empty(n, dtype=torch.int32, device=dev)
    mask = torch.empty(n * nw, dtype=torch.int64, device=dev)
    sboxes = torch.empty(5 * n + 1, dtype=torch.float32, device=dev)
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(L.lib().mi_batched_nms(boxes.data_ptr(), scores.data_ptr(), idxs.data_ptr(), n, float(iou_threshold),
                                   order.data_ptr(), mask.data_ptr(), sboxes.data_ptr(), keep.data_ptr(),
                                   nkeep.data_ptr(), L.stream_ptr()), "mi_batched_nms")
    return keep[: int(nkeep.item())]


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45):
    """prediction [B, A, 5+ncls] decoded (cx,cy,w,h,obj,cls...) -> list of [n_i, 7] or None
    rows (x1,y1,x2,y2,obj,cls_conf,cls_idx) in descending score order; mutates `prediction` like the
    reference does (cxcywh -> xyxy in place)."""
    box_corner = prediction.new_empty(prediction.shape)
    box_corner[:, :, 0] = prediction[:, :, 0] - prediction[:, :, 2] / 2
    box_corner[:, :, 1] = prediction[:, :, 1] - prediction[:, :, 3] / 2
    box_corner[:, :, 2] = prediction[:, :, 0] + prediction[:, :, 2] / 2
    box_corner[:, :, 3] = prediction[:, :, 1] + prediction[:, :, 3] / 2
    prediction[:, :, :4] = box_corner[:, :, :4]
    output = [None for _ in range(len(prediction))]
    for i, image_pred in enumerate(prediction):
        if not image_pred.size(0):
            continue
        class_conf, class_pred = torch.max(image_pred[:, 5: 5 + num_classes], 1, keepdim=True)
        conf_mask = (image_pred[:, 4] * class_conf.squeeze() >= conf_thre).squeeze()
        detections = torch.cat((image_pred[:, :5], class_conf, class_pred.float()), 1)
        detections = detections[conf_mask]
        if not detections.size(0):
            continue
        keep = batched_nms(detections[:, :4], detections[:, 4] * detections[:, 5], detections[:, 6], nms_thre)
        output[i] = detections[keep]
    return output


# ------------------------------------------------------------------------------------------------------------------
# the NMS family of the sibling heads: yolov7/modeling/meta_arch/utils.py:33-113 and
# yolov7/utils/solov2_utils.py:160-206
def _nms_ex(boxes, scores, idxs, iou_threshold, arithmetic):
    n, dev = boxes.shape[0], boxes.device
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dev)
    boxes, scores, idxs = boxes.contiguous().float(), scores.contiguous().float(), idxs.contiguous().float()
    nw = (n + 63) // 64
    order = torch.empty(n, dtype=torch.int32, device=dev)
    mask = torch.empty(n * nw, dtype=torch.int64, device=dev)
    sboxes = torch.empty(5 * n + 1, dtype=torch.float32, device=dev)
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(L.lib().mi_batched_nms_ex(boxes.data_ptr(), scores.data_ptr(), idxs.data_ptr(), n, float(iou_threshold),
                                      arithmetic, order.data_ptr(), mask.data_ptr(), sboxes.data_ptr(), keep.data_ptr(),
                                      nkeep.data_ptr(), L.stream_ptr()), "mi_batched_nms_ex")
    return keep[: int(nkeep.item())]


def batched_softnms(boxes, scores, idxs, iou_threshold, score_threshold=0.001, soft_mode="gaussian"):
    """meta_arch/utils.py:47-63: class-aware Soft-NMS; `scores` is rescaled IN PLACE (as the reference does) and the
    indices with score > score_threshold come back in descending score order"""
    assert soft_mode in ["linear", "gaussian"]
    assert boxes.shape[-1] == 4
    if not boxes.is_cuda:
        raise L.MI355Error("batched_softnms: HIP tensors required (no CPU fallback)")
    n, dev = boxes.shape[0], boxes.device
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dev)
    if not (scores.dtype == torch.float32 and scores.is_contiguous()):
        raise ValueError("batched_softnms rescales `scores` in place: pass a contiguous float32 tensor")
    b, ix = boxes.contiguous().float(), idxs.contiguous().float()
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(L.lib().mi_batched_softnms(b.data_ptr(), scores.data_ptr(), ix.data_ptr(), n, float(iou_threshold),
                                       float(score_threshold), int(soft_mode == "linear"), keep.data_ptr(), nkeep.data_ptr(),
                                       L.stream_ptr()), "mi_batched_softnms")
    return keep[: int(nkeep.item())]


def batched_clusternms(boxes, scores, idxs, iou_threshold):
    """meta_arch/utils.py:66-95.  Cluster-NMS iterates keep = (max_i keep_i * iou_ij <= thr) over the score-sorted,
    upper-triangular IoU matrix of one class until nothing changes; that fixed point IS the greedy NMS result (box j is
    dropped iff a KEPT higher-scoring box overlaps it by more than thr), so the class-by-class NMS kernel computes it in
    one pass."""
    assert boxes.shape[-1] == 4
    if not boxes.is_cuda:
        raise L.MI355Error("batched_clusternms: HIP tensors required (no CPU fallback)")
    return _nms_ex(boxes, scores, idxs, iou_threshold, 0)


def generalized_batched_nms(boxes, scores, idxs, iou_threshold, score_threshold=0.001, nms_type="normal"):
    """meta_arch/utils.py:98-113 (MODEL.NMS_TYPE): "normal", "softnms-linear", "softnms-gaussian", "cluster" """
    assert boxes.shape[-1] == 4
    if nms_type == "normal":
        return batched_nms(boxes, scores, idxs, iou_threshold)
    if nms_type.startswith("softnms"):
        return batched_softnms(boxes, scores, idxs, iou_threshold, score_threshold=score_threshold,
                               soft_mode=nms_type.lstrip("softnms-"))      # (sic) the reference's way to drop the prefix
    if nms_type == "cluster":
        return batched_clusternms(boxes, scores, idxs, iou_threshold)
    raise NotImplementedError("NMS type not implemented: \"{}\"".format(nms_type))


def _mask_inter(seg_masks, n):
    """n x n mask intersections = masks @ masks^T, one MFMA pass over the pixels (0 / 1 masks: exact)"""
    from .sparseinst import pixel_outer
    dev = seg_masks.device
    m = seg_masks.reshape(n, -1)
    P = m.shape[1]
    npad, ppad = (n + 31) // 32 * 32, (P + 7) // 8 * 8
    mt = torch.zeros(ppad, npad, dtype=torch.bfloat16, device=dev)      # pixels x candidates, zero padded
    mt[:P, :n] = m.t().to(torch.bfloat16)
    return pixel_outer(mt, mt)[:n, :n].contiguous()


def mask_nms(cate_labels, seg_masks, sum_masks, cate_scores, nms_thr=0.5):
    """utils/solov2_utils.py:209-236: greedy NMS on mask IoU (candidates sorted by descending score) -> keep [n]
    (1.0 / 0.0 in the masks' dtype, like the reference's `seg_masks.new_ones`)"""
    n = len(cate_scores)
    if n == 0:
        return []
    if not seg_masks.is_cuda:
        raise L.MI355Error("mask_nms: HIP tensors required (no CPU fallback)")
    inter = _mask_inter(seg_masks, n)
    keep = torch.empty(n, dtype=torch.uint8, device=seg_masks.device)
    L.check(L.lib().mi_mask_nms(inter.data_ptr(), sum_masks.contiguous().float().data_ptr(),
                                cate_labels.contiguous().float().data_ptr(), n, float(nms_thr), keep.data_ptr(), L.stream_ptr()),
            "mi_mask_nms")
    return keep.to(seg_masks.dtype if seg_masks.dtype != torch.bool else torch.float32)


def matrix_nms(cate_labels, seg_masks, sum_masks, cate_scores, sigma=2.0, kernel="gaussian"):
    """utils/solov2_utils.py:160-206 (SOLOv2): decayed scores of n mask candidates sorted by descending score.
    The n x n mask intersections are one MFMA pass over the pixels (masks are 0 / 1: exact in bf16 with fp32 sums)."""
    n = len(cate_labels)
    if n == 0:
        return []
    if not seg_masks.is_cuda:
        raise L.MI355Error("matrix_nms: HIP tensors required (no CPU fallback)")
    dev = seg_masks.device
    inter = _mask_inter(seg_masks, n)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    comp = torch.empty(n, dtype=torch.float32, device=dev)
    L.check(L.lib().mi_matrix_nms(inter.data_ptr(), sum_masks.contiguous().float().data_ptr(),
                                  cate_labels.contiguous().float().data_ptr(), cate_scores.contiguous().float().data_ptr(), n,
                                  float(sigma), int(kernel == "linear"), comp.data_ptr(), out.data_ptr(), L.stream_ptr()),
            "mi_matrix_nms")
    return out
