"""child process of tests/test_gpu_augment.py::test_dataset_mapper_batches_equal_the_oracle: `GpuDatasetMapper.make_batch`
(front + mosaic + random_perspective + mixup + the mixed pad-to-batch) on the GPU against the oracle's `mapper_call` +
`preprocess_batch` on the same two random streams.  Exit code 0 = every batch bit-identical (pixels and label rows)."""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import augment_oracle as A  # noqa: E402
from yolov7_d2_amd.data_pipeline import GpuDatasetMapper  # noqa: E402

FRONT = dict(MIN_SIZE_TRAIN=(96, 128, 160), MAX_SIZE_TRAIN=224, SHIFT_PIXELS=12)
MOSAIC = dict(MOSAIC_WIDTH_RANGE=(128, 224), MOSAIC_HEIGHT_RANGE=(128, 224))
OFRONT = dict(min_sizes=(96, 128, 160), max_size=224, max_shifts=12)


def data(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        h, w = int(rs.randint(80, 260)), int(rs.randint(80, 260))
        yy, xx = np.mgrid[0:h, 0:w]
        base = (127 + 90 * np.sin(xx / 11.0 + rs.uniform(0, 6)) * np.cos(yy / 13.0 + rs.uniform(0, 6)))[..., None]
        img = np.clip(base + rs.randint(-30, 31, (h, w, 3)), 0, 255).astype(np.uint8)
        m = int(rs.randint(0, 7))
        x1 = rs.uniform(0, w - 30, m); y1 = rs.uniform(0, h - 30, m)
        lab = np.stack([x1, y1, np.minimum(x1 + rs.uniform(8, 150, m), w), np.minimum(y1 + rs.uniform(8, 150, m), h),
                        rs.randint(0, 80, m).astype(np.float64)], 1)
        out.append((img, lab))
    return out


bad, seen = [], dict(plain=0, mosaic=0, mixed_batches=0)
# distort: the colour entries of configs/coco/yolox_s.yaml:46-50 (RandomSaturation, RandomBrightness, YOLOFRandomDistortion) on -
# every loaded image is then float32 in the reference and the mosaic / mixup resizes take cv2's float path
for mixup, distort in ((False, False), (True, False), (False, True), (True, True)):
    FRONT = dict(FRONT, SATURATION=distort, BRIGHTNESS=distort, DISTORTION=distort)
    OFRONT = dict(OFRONT, saturation=distort, brightness=distort, distortion=(0.1, 1.5, 1.5) if distort else None)
    mp = GpuDatasetMapper(device="cuda", enable_mixup=mixup, front_cfg=FRONT, mosaic_cfg=MOSAIC)
    r1n, r1p, r2n, r2p = np.random.RandomState(17), random.Random(18), np.random.RandomState(17), random.Random(18)
    pool = []
    samples = data(40 + mixup, 24)
    for k in range(0, 24, 6):
        chunk = samples[k: k + 6]
        out, rows, sizes = mp.make_batch([(torch.from_numpy(i), l) for i, l in chunk], r1n, r1p)
        torch.cuda.synchronize()
        ref = [A.mapper_call(pool, (i, l), r2n, r2p, mcfg=MOSAIC, front_kw=OFRONT, enable_mixup=mixup) for i, l in chunk]
        ref_img, ref_rows = A.preprocess_batch([(r[0], r[1]) for r in ref])
        kinds = [r[2] for r in ref]
        seen["plain"] += kinds.count(False); seen["mosaic"] += kinds.count(True); seen["mixed_batches"] += len(set(kinds)) == 2
        got = out.cpu().numpy()
        if got.shape != ref_img.shape:
            bad.append(("shape", mixup, distort, k, got.shape, ref_img.shape))
            continue
        for b in range(len(chunk)):
            n = int((got[b] != ref_img[b]).sum())
            if n:
                bad.append(("pixels", mixup, distort, k, b, kinds[b], n))
        if not np.array_equal(rows.cpu().numpy(), ref_rows):
            bad.append(("rows", mixup, distort, k))
        if any(tuple(s) != r[0].shape[:2] for s, r in zip(sizes, ref) if not r[2]):     # (a mosaic sample reports its input_dim)
            bad.append(("sizes", mixup, distort, k))
if seen["mosaic"] < 8 or seen["plain"] < 16 or seen["mixed_batches"] < 4:
    bad.append(("coverage", seen))
print("dataset mapper on the GPU:", "bit-identical " + str(seen) if not bad else bad[:8])
sys.exit(1 if bad else 0)
