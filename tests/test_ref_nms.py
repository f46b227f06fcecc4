"""The reference's own native greedy NMS (deploy/trt_cc/demo_yolox.cc:53-135, compiled by oracle/Makefile into
oracle/_ref/libref_nms.so) pins the NMS restatement of the oracle (CPU) and the HIP kernel (GPU) on tie-free inputs:
torchvision - whose batched_nms the eval path calls (utils/boxes.py:199) - is neither vendored nor installed, this loop
is the only statement of the suppression rule the reference tree holds (descending score, suppress iff IoU > thr)."""
import os
import sys

import numpy as np
import pytest
import torch

import yolox_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ref_nms as R   # noqa: E402

needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref_nms.so not built (make -C oracle, needs /root/reference)")


def _tie_free(n, ncls, thr, seed):
    """clustered boxes with unique scores and no pairwise IoU within 2e-4 of the threshold (the reference computes the
    areas from (w, h), the torchvision form from (x2 - x1): at an exact tie the float roundings could differ) - boxes of
    a near-threshold pair are dropped, so slightly fewer than n come back"""
    g = np.random.default_rng(seed)
    ctr = g.uniform(40, 600, size=(max(n // 25, 1), 2))
    c = ctr[g.integers(0, len(ctr), n)] + g.normal(0, 12, size=(n, 2))
    wh = g.uniform(20, 120, size=(n, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    scores = (g.permutation(n).astype(np.float32) / n * 0.98 + 0.01).astype(np.float32)
    idxs = g.integers(0, ncls, n)
    x1 = np.maximum(boxes[:, None, 0], boxes[None, :, 0]); y1 = np.maximum(boxes[:, None, 1], boxes[None, :, 1])
    x2 = np.minimum(boxes[:, None, 2], boxes[None, :, 2]); y2 = np.minimum(boxes[:, None, 3], boxes[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    iou = inter / (area[:, None] + area[None, :] - inter)
    near = np.abs(iou - thr) <= 2e-4
    drop = np.unique(np.nonzero(np.triu(near, 1))[1])
    keep = np.setdiff1d(np.arange(n), drop)
    assert len(keep) >= max(1, n - n // 10)
    return boxes[keep], scores[keep], idxs[keep]


@needs_ref
@pytest.mark.parametrize("n,thr", [(1, 0.45), (2, 0.45), (64, 0.45), (65, 0.65), (300, 0.65), (1000, 0.65), (1500, 0.3)])
def test_oracle_nms_against_reference_native(n, thr):
    boxes, scores, _ = _tie_free(n, 1, thr, seed=n)
    ref = R.nms_xyxy(boxes, scores, thr)
    got = O.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
    assert len(ref) >= 1 and np.array_equal(got, ref)


@needs_ref
@pytest.mark.parametrize("n,ncls", [(500, 5), (900, 80), (1200, 80)])   # 900 x 4 <= 4000 elements: the coordinate trick; 1200: per class
def test_oracle_batched_nms_against_reference_native(n, ncls):
    boxes, scores, idxs = _tie_free(n, ncls, 0.65, seed=7 * n)
    ref = R.batched_nms_xyxy(boxes, scores, idxs, 0.65)
    got = O.batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(idxs).float(), 0.65).numpy()
    assert np.array_equal(got, ref)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("n,ncls,thr", [(1, 1, 0.45), (63, 1, 0.45), (64, 3, 0.45), (65, 80, 0.65), (1000, 80, 0.65), (3000, 80, 0.65), (8400, 80, 0.65)])
def test_hip_batched_nms_against_reference_native(n, ncls, thr):
    from yolov7_d2_amd.modeling.postprocess import batched_nms
    boxes, scores, idxs = _tie_free(n, ncls, thr, seed=3 * n + ncls)
    ref = R.batched_nms_xyxy(boxes, scores, idxs, thr)
    got = batched_nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(idxs).float().cuda(), thr)
    assert torch.equal(got.cpu(), torch.from_numpy(ref))
