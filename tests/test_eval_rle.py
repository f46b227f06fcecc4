"""COCO evaluation output (f4): RLE masks + instances_to_coco_json.  CPU: the oracle's restatement of cocoapi's RLE is
self-consistent (loop form == vector form, string round trip, hand-checked vectors).  GPU: the HIP encoder is
bit-exact against it, including empty / full / single-pixel / odd-sized / very fragmented masks."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import coco_rle_oracle as R  # noqa: E402


def _masks(n, H, W, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[:H, :W]
    ms = []
    for k in range(n):
        cy, cx, r = g.uniform(0, H), g.uniform(0, W), g.uniform(1, max(H, W) / 3 + 1)
        ms.append(((yy - cy) ** 2 + (xx - cx) ** 2) < r * r)
    ms = np.stack(ms).astype(np.uint8)
    ms[0] = 0                                  # empty
    if n > 1:
        ms[1] = 1                              # full
    if n > 2:
        ms[2] = 0; ms[2, H // 2, W // 3] = 1   # one pixel
    if n > 3:
        ms[3] = 0; ms[3, 0, 0] = 1             # starts with a one: an empty leading run of zeros
    if n > 4:
        ms[4] = (g.uniform(size=(H, W)) < 0.5)  # noise: ~H*W/2 runs
    return ms


def test_rle_oracle_hand_checked_vectors_and_round_trip():
    m = np.array([[0, 1], [1, 1]], dtype=np.uint8)          # column-major scan: 0 1 1 1
    assert R.rle_counts(m).tolist() == [1, 3]
    assert R.rle_counts(np.ones((2, 3))).tolist() == [0, 6]
    assert R.rle_counts(np.zeros((2, 3))).tolist() == [6]
    # rleToString by hand: 1 -> '1' (0x01 + 48), 3 -> '3'; 6 -> '6'; 0 -> '0'; 40 = 0b01000 | (1 << 5): low group 8 with
    # "more" (8 | 0x20 = 40 -> chr 88 'X'), then 1 -> '1'
    assert R.rle_to_string([1, 3]) == "13" and R.rle_to_string([0, 6]) == "06" and R.rle_to_string([40]) == "X1"
    # third and later counts are differences to the count two places before (may be negative: sign bit 0x10)
    s = R.rle_to_string([5, 7, 9, 2])
    assert R.rle_from_string(s).tolist() == [5, 7, 9, 2]
    for k, msk in enumerate(_masks(8, 37, 53, 3)):
        c = R.rle_counts(msk)
        assert np.array_equal(c, R.rle_counts_fast(msk))
        assert int(c.sum()) == msk.size
        s = R.rle_to_string(c)
        assert np.array_equal(R.rle_from_string(s), c.astype(np.int64))
        assert np.array_equal(R.rle_decode(c, 37, 53), msk)


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,W", [(8, 37, 53), (100, 160, 160), (5, 1, 1), (6, 480, 640), (5, 3, 1000)])
def test_rle_encode_bit_exact(n, H, W):
    from yolov7_d2_amd.evaluation import rle_encode
    ms = _masks(n, H, W, H * W + n)
    got = rle_encode(torch.from_numpy(ms).to("cuda"))
    assert len(got) == n
    for k in range(n):
        c = R.rle_counts_fast(ms[k])
        assert got[k]["size"] == [H, W]
        assert got[k]["counts"] == R.rle_to_string(c), k
        assert np.array_equal(R.rle_decode(R.rle_from_string(got[k]["counts"]), H, W), ms[k])
    assert rle_encode(torch.zeros(0, H, W, dtype=torch.bool, device="cuda")) == []


@pytest.mark.gpu
def test_instances_to_coco_json_and_evaluator():
    from yolov7_d2_amd.d2shim import Boxes, Instances
    from yolov7_d2_amd.evaluation import COCOMaskEvaluator, instances_to_coco_json
    ms = _masks(6, 48, 64, 9)
    inst = Instances((48, 64))
    inst.pred_boxes = Boxes(torch.tensor([[1.0, 2.0, 11.0, 22.0]] * 6, device="cuda"))
    inst.scores = torch.linspace(0.9, 0.4, 6, device="cuda")
    inst.pred_classes = torch.arange(6, device="cuda")
    inst.pred_masks = torch.from_numpy(ms).bool().to("cuda")
    res = instances_to_coco_json(inst, 42)
    assert len(res) == 6 and res[0]["image_id"] == 42 and res[3]["category_id"] == 3
    assert res[0]["bbox"] == [1.0, 2.0, 10.0, 20.0]                         # XYXY_ABS -> XYWH_ABS
    assert res[2]["segmentation"]["counts"] == R.rle_to_string(R.rle_counts_fast(ms[2]))
    json.dumps(res)                                                         # serialisable (counts is a str)
    # masks only (SparseInst): no bbox key
    inst2 = Instances((48, 64))
    inst2.scores, inst2.pred_classes, inst2.pred_masks = inst.scores, inst.pred_classes, inst.pred_masks
    assert "bbox" not in instances_to_coco_json(inst2, 1)[0]
    ev = COCOMaskEvaluator()
    ev.process([{"image_id": 7}, {"image_id": 8}], [{"instances": inst}, {}])
    assert len(ev._predictions) == 1 and ev._predictions[0]["image_id"] == 7 and len(ev._predictions[0]["instances"]) == 6
