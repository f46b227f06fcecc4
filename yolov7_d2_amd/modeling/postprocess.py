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
