#!/usr/bin/env python
"""where the HOST spends a captured DETR / SparseInst step: prepare_batch (eager launches + blocking H2D copies) against the
graph launch.  The device idles whenever the two together take longer than the graph runs (in-graph traces: 1.7 ms of gaps
per SparseInst step, all of them in the eager prefix).  usage: host_step_probe.py [detr|sparseinst] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "sparseinst"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
import bench
import yolov7_d2_amd as M
from yolov7_d2_amd.graph_step import GraphedTrainStep
from yolov7_d2_amd.optim import MultiTensorAdamW
dev = torch.device("cuda", 0)
torch.manual_seed(0)
inputs, model = bench.synthetic_batch(which) if hasattr(bench, "synthetic_batch") else (None, None)
if inputs is None:
    from yolov7_d2_amd.d2shim import Boxes, Instances
    g = torch.Generator().manual_seed(1234)
    if which == "detr":
        model = M.build_model(M.detr_r50_cfg(device="cuda:0"))
        B, H_, W_ = 4, 800, 1333
    else:
        model = M.build_model(M.sparse_inst_r50_giam_cfg(device="cuda:0"))
        B, H_, W_ = 8, 640, 640
    inputs = []
    for b in range(B):
        h, w = H_, W_
        n = 5
        wh = 16 + torch.rand(n, 2, generator=g) * 128
        xy = torch.rand(n, 2, generator=g) * (torch.tensor([w, h]) - wh).clamp(min=1)
        if which == "detr":
            inst = Instances((h, w), gt_boxes=Boxes(torch.cat([xy, xy + wh], 1)), gt_classes=torch.randint(0, 80, (n,), generator=g))
        else:
            m = torch.zeros(n, h, w)
            for k in range(n):
                x0, y0, x1, y1 = [int(v) for v in torch.cat([xy[k], xy[k] + wh[k]])]
                m[k, y0:y1, x0:x1] = 1
            inst = Instances((h, w), gt_classes=torch.randint(0, 80, (n,), generator=g).to(dev), gt_masks=m.to(dev))
        inputs.append(dict(image=torch.randint(0, 256, (3, h, w), generator=g).float().to(dev), instances=inst, height=h, width=w))
model.train()
params = [p for p in model.parameters() if p.requires_grad]
opt = MultiTensorAdamW(params, lr=1e-5, weight_decay=1e-4)
gs = GraphedTrainStep(model, opt)
for _ in range(3):
    gs(inputs)
torch.cuda.synchronize()

# instrument
tp = [0.0]
orig_prepare = model.prepare_batch


def timed_prepare(*a, **k):
    t = time.perf_counter()
    r = orig_prepare(*a, **k)
    tp[0] += time.perf_counter() - t
    return r


model.prepare_batch = timed_prepare
tr = [0.0]
orig_replay = gs.replay_backward


def timed_replay(ent):
    t = time.perf_counter()
    r = orig_replay(ent)
    tr[0] += time.perf_counter() - t
    return r


gs.replay_backward = timed_replay
t0 = time.perf_counter()
for _ in range(steps):
    gs(inputs)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{which}: {steps} steps; per step: wall {t_all / steps * 1e3:.3f} ms, host loop {t_host / steps * 1e3:.3f} ms, "
      f"prepare_batch (host) {tp[0] / steps * 1e3:.3f} ms, graph launch (host) {tr[0] / steps * 1e3:.3f} ms")
# the graph alone, back to back (no prepare): the device-side length of the captured step
key = model.batch_key(inputs)
ent = gs.graphs[key]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    orig_replay(ent)
    if ent[1] is not None:
        ent[1].replay()
t_l = time.perf_counter() - t0
torch.cuda.synchronize()
t_g = time.perf_counter() - t0
print(f"graph replays alone: {t_g / steps * 1e3:.3f} ms per step (host launch {t_l / steps * 1e3:.3f} ms)")
# prepare alone, device idle
t0 = time.perf_counter()
for _ in range(steps):
    orig_prepare(inputs, static=ent[2])
torch.cuda.synchronize()
print(f"prepare_batch alone (device otherwise idle): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per call")
