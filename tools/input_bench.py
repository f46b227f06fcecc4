#!/usr/bin/env python
"""Throughput of the device input pipeline pieces (images / s, one GPU, synthetic COCO-sized data): JPEG decode (host Huffman
threads + two launches), the T.* front, the whole YOLOX mapper (mosaic / plain mix, with and without mixup) and DETR's mapper.
NOT yet run on a device (written after round 3's GPU minutes were spent).  usage: input_bench.py [batch] [iters]"""
import io, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from yolov7_d2_amd.data_pipeline import GpuDatasetMapper, GpuDetrMapper, GpuFrontAugment, GpuJpegDecoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rs = np.random.RandomState(0)
files, labels = [], []
for k in range(B):
    h, w = [(480, 640), (427, 640), (640, 480), (375, 500)][k % 4]
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.clip(np.stack([127 + 100 * np.sin(xx / 9.0 + yy / 17.0), 127 + 100 * np.cos(xx / 13.0), 127 + 100 * np.sin(yy / 7.0)], -1)
                  + rs.randint(-25, 26, (h, w, 3)), 0, 255).astype(np.uint8)
    buf = io.BytesIO(); Image.fromarray(img).save(buf, format="JPEG", quality=90, subsampling=2)
    files.append(buf.getvalue())
    m = int(rs.randint(1, 12))
    x1 = rs.uniform(0, w - 30, m); y1 = rs.uniform(0, h - 30, m)
    labels.append(np.stack([x1, y1, np.minimum(x1 + rs.uniform(8, 300, m), w), np.minimum(y1 + rs.uniform(8, 300, m), h),
                            rs.randint(0, 80, m).astype(np.float64)], 1))


def timed(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(IT):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / IT
    print(f"{name:46s} {dt * 1e3:8.2f} ms / batch of {B}  = {B / dt:9.0f} images/s")


dec = GpuJpegDecoder(format="BGR", workers=min(32, os.cpu_count() or 8))
timed("GpuJpegDecoder.decode (BGR, EXIF)", lambda: dec.decode(files))
images = dec.decode(files)
front = GpuFrontAugment()
rn, rp = np.random.RandomState(1), random.Random(2)
timed("GpuFrontAugment.make_batch (mapper, mosaic off)", lambda: front.make_batch(images, labels, [front.draw(tuple(i.shape[:2]), rn) for i in images]))
for mix in (False, True):
    mp = GpuDatasetMapper(enable_mixup=mix)
    for _ in range(2):
        mp.make_batch(list(zip(images, labels)), rn, rp)           # fills the pool: mosaic samples from here on
    timed(f"GpuDatasetMapper.make_batch (mixup {mix})", lambda: mp.make_batch(list(zip(images, labels)), rn, rp))
dm = GpuDetrMapper()
timed("GpuDetrMapper.make_batch", lambda: dm.make_batch(images, labels, rn))
timed("decode + GpuDatasetMapper (end to end)", lambda: GpuDatasetMapper.make_batch(mp, list(zip(dec.decode(files), labels)), rn, rp))
