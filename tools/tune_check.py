#!/usr/bin/env python
"""GPU box: run the in-sequence conv autotuner verbosely on the bench plan and report the step time before/after"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd.engine import NativeTrainer
from bench import synth_batch_device
torch.manual_seed(0)
model = M.build_model(M.yolox_s_cfg(device="cuda"))
tr = NativeTrainer(model, lr=0.0025, tune=False)
imgs, labels = synth_batch_device(16, 640, 640, 1234, "cuda")
st = tr.load_batch(imgs, labels)
def bench(n=30):
    for _ in range(3): tr.step(st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(st)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("heuristic ms/step", bench(), bench())
t0 = time.perf_counter()
with torch.cuda.stream(tr.stream):
    auto, tuned = st["plan"].autotune_convs(stream=tr.stream, verbose=("-v" in sys.argv))
torch.cuda.synchronize()
print(f"tuning took {time.perf_counter() - t0:.1f}s: conv sum {auto:.0f} -> {tuned:.0f} us")
st["graphs"] = None; st["warm"] = True
print("tuned ms/step", bench(), bench())
