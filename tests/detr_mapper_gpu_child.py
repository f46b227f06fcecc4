"""child process of tests/test_gpu_augment.py::test_detr_mapper_equals_the_oracle: `GpuDetrMapper.make_batch` on the GPU against
the oracle's `detr_mapper_call` (whose pixels the CPU suite holds to the real Pillow) on the same random stream, at the
reference's DETR sizes (MIN_SIZE_TRAIN 480..832, max 1333, crop (384, 600)).  Exit code 0 = bit-identical."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import augment_oracle as A  # noqa: E402
from yolov7_d2_amd.data_pipeline import GpuDetrMapper  # noqa: E402

rs = np.random.RandomState(21)
data = []
for (h, w) in [(480, 640), (427, 640), (640, 480), (375, 500), (333, 500), (612, 612), (500, 375), (360, 640)]:
    img = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
    m = int(rs.randint(0, 7))
    x1 = rs.uniform(0, w - 30, m); y1 = rs.uniform(0, h - 30, m)
    lab = np.stack([x1, y1, np.minimum(x1 + rs.uniform(8, 300, m), w), np.minimum(y1 + rs.uniform(8, 300, m), h),
                    rs.randint(0, 80, m).astype(np.float64)], 1)
    data.append((img, lab))
mp = GpuDetrMapper(device="cuda")
r1, r2 = np.random.RandomState(5), np.random.RandomState(5)
res = mp.make_batch([torch.from_numpy(i).cuda() for i, _ in data], [l for _, l in data], r1)
torch.cuda.synchronize()
bad, crops = [], 0
for k, ((img, lab), (o, box, cls_)) in enumerate(zip(data, res)):
    ref, rbox, rcls, rec = A.detr_mapper_call(img, lab, r2)
    crops += rec["crop"] is not None
    got = o.cpu().numpy()
    if got.shape != ref.transpose(2, 0, 1).shape or not np.array_equal(got, ref.transpose(2, 0, 1)):
        bad.append(("pixels", k, got.shape, ref.shape, rec))
    if not (np.array_equal(box, rbox) and np.array_equal(cls_, rcls)):
        bad.append(("boxes", k))
if crops == 0 or crops == len(data):
    bad.append(("coverage: crop branch taken %d of %d" % (crops, len(data)),))
print("detr mapper on the GPU:", "bit-identical (%d crop / %d plain)" % (crops, len(data) - crops) if not bad else bad[:6])
sys.exit(1 if bad else 0)
