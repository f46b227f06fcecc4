"""COCO-format evaluation output — drop-in for yolov7/evaluation/coco_evaluation.py:14-100
(`instances_to_coco_json`, `COCOMaskEvaluator.process`): Instances -> list of COCO json dicts, masks as COCO RLE.

The reference encodes every mask on the CPU with pycocotools (`mask_util.encode` on an np.array per mask).  Here the
run-length encoding runs on the GPU (`mi_rle_encode`: one block per mask, column-major scan) and only the run lengths
(a few hundred integers per mask instead of H x W bytes) cross PCIe; the compact ASCII `counts` string is produced by
host code in the same library (`mi_rle_to_string`).  pycocotools is un-vendored (and absent from this image): the byte
format is restated from cocoapi's maskApi.c and checked against an independent numpy restatement + a decoder round trip
(tests/test_gpu_eval.py) - *parity unpinned* against pycocotools itself."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .d2shim import Boxes


def rle_encode(masks):
    """masks: bool / uint8 tensor [n, H, W] on the HIP device -> list of {"size": [H, W], "counts": str} (COCO RLE)"""
    if masks.dim() != 3:
        raise ValueError("rle_encode: masks [n, H, W]")
    n, H, W = masks.shape
    if n == 0:
        return []
    if not masks.is_cuda:
        raise L.MI355Error("rle_encode: the MI355X path needs device tensors (no CPU fallback)")
    m = masks.to(torch.uint8).contiguous()
    max_runs = 2048
    while True:
        counts = torch.empty(n, max_runs, dtype=torch.int32, device=m.device)
        nruns = torch.empty(n, dtype=torch.int32, device=m.device)
        L.check(L.lib().mi_rle_encode(m.data_ptr(), n, H, W, max_runs, counts.data_ptr(), nruns.data_ptr(), L.stream_ptr()),
                "mi_rle_encode")
        nr = nruns.cpu().numpy()
        if (nr > 0).all():
            break
        max_runs = int(-nr.min()) + 1          # a very fragmented mask: once more with room for it
    cnt = counts.cpu().numpy().view(np.uint32)
    out, buf = [], C.create_string_buffer(7 * int(nr.max()) + 8)
    for k in range(n):
        row = np.ascontiguousarray(cnt[k, : nr[k]])
        ln = L.check(L.lib().mi_rle_to_string(row.ctypes.data, int(nr[k]), buf, len(buf)), "mi_rle_to_string")
        out.append({"size": [int(H), int(W)], "counts": buf.raw[:ln].decode("ascii")})
    return out


def instances_to_coco_json(instances, img_id):
    """coco_evaluation.py:14-76: an Instances object (boxes optional: SparseInst predicts masks only) -> COCO json dicts"""
    num_instance = len(instances)
    if num_instance == 0:
        return []
    has_box = instances.has("pred_boxes")
    if has_box:
        b = instances.pred_boxes.tensor.detach().float().cpu().numpy().copy()
        b[:, 2] -= b[:, 0]                      # BoxMode.convert(XYXY_ABS -> XYWH_ABS)
        b[:, 3] -= b[:, 1]
        boxes = b.tolist()
    scores = instances.scores.tolist()
    classes = instances.pred_classes.tolist()
    has_mask = instances.has("pred_masks")
    if has_mask:
        rles = rle_encode(instances.pred_masks)
    has_keypoints = instances.has("pred_keypoints")
    if has_keypoints:
        keypoints = instances.pred_keypoints.detach().float().cpu().clone()
    results = []
    for k in range(num_instance):
        result = {"image_id": img_id, "category_id": classes[k], "score": scores[k]}
        if has_box:
            result["bbox"] = boxes[k]
        if has_mask:
            result["segmentation"] = rles[k]
        if has_keypoints:
            keypoints[k][:, :2] -= 0.5          # :66-71: predictions are float coordinates, annotations pixel indices
            result["keypoints"] = keypoints[k].flatten().tolist()
        results.append(result)
    return results


class COCOMaskEvaluator:
    """the `process` override of coco_evaluation.py:79-100 (the COCOEvaluator base class - dataset loading, COCOeval - is
    detectron2's and stays there): collects {"image_id", "instances": [json dicts]} per image"""

    def __init__(self):
        self._predictions = []

    def reset(self):
        self._predictions = []

    def process(self, inputs, outputs):
        for inp, output in zip(inputs, outputs):
            prediction = {"image_id": inp["image_id"]}
            if "instances" in output:
                prediction["instances"] = instances_to_coco_json(output["instances"], inp["image_id"])
            if "proposals" in output:
                prediction["proposals"] = output["proposals"]
            if len(prediction) > 1:
                self._predictions.append(prediction)
