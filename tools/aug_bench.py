#!/usr/bin/env python
"""throughput of the GPU input pipeline (mosaic + warp + pad) at the benchmark batch: B = 16 samples of 4 COCO-sized images"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from yolov7_d2_amd.data_pipeline import GpuMosaicMapper, MosaicPool
rs = np.random.RandomState(0)
pool = MosaicPool("cuda")
for k in range(64):
    h, w = (480, 640) if k % 3 else (640, 427)
    n = rs.randint(1, 15)
    x1 = rs.uniform(0, w - 40, n); y1 = rs.uniform(0, h - 40, n)
    pool.append(torch.from_numpy(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)),
                np.stack([x1, y1, x1 + rs.uniform(8, 200, n), y1 + rs.uniform(8, 200, n), rs.randint(0, 80, n).astype(float)], 1))
for rng in ((512, 800), (640, 640)):
    mapper = GpuMosaicMapper(dict(MOSAIC_WIDTH_RANGE=rng, MOSAIC_HEIGHT_RANGE=rng), device="cuda")
    rng_np, rng_py = np.random.RandomState(1), random.Random(2)
    B, iters = 16, 30
    def one():
        groups = [tuple(int(i) for i in rng_np.randint(0, len(pool), 4)) for _ in range(B)]
        return mapper.make_batch(pool, groups, [mapper.draw(rng_np, rng_py) for _ in range(B)])
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = one(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    gpu = np.median([a.elapsed_time(b) for a, b in ev])
    print(f"mosaic size range {rng}: {B / wall:8.0f} images/s end to end ({wall * 1e3:.2f} ms per batch of {B}; GPU span {gpu:.2f} ms, "
          f"last batch padded to {tuple(out[0].shape[2:])})")
