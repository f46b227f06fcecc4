#!/usr/bin/env python
"""can the eager DETR / SparseInst training step be captured as one hipGraph (torch.cuda.graph)?  timing eager vs replay"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd.d2shim import Boxes, Instances
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, H_, W_ = 4, 800, 1333
model = M.build_model(M.detr_r50_cfg(device="cuda:0")); model.train()
g = torch.Generator().manual_seed(1234)
inputs = []
for b in range(B):
    h, w = (H_, W_) if b == 0 else (H_ - 32 * (b % 3), W_ - 64 * (b % 4))
    n = int(torch.randint(1, 21, (1,), generator=g))
    wh = 16 + torch.rand(n, 2, generator=g) * 256
    xy = torch.rand(n, 2, generator=g) * (torch.tensor([w, h]) - wh).clamp(min=1)
    inst = Instances((h, w), gt_boxes=Boxes(torch.cat([xy, xy + wh], 1)), gt_classes=torch.randint(0, 80, (n,), generator=g))
    inputs.append(dict(image=torch.randint(0, 256, (3, h, w), generator=g).float().to(dev), instances=inst))
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, capturable=True)
def step():
    losses = model(inputs)
    total = sum(v for k, v in losses.items() if k in model.criterion.weight_dict)
    opt.zero_grad(set_to_none=False)
    total.backward()
    opt.step()
    return total
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 5 * 1e3, flush=True)
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr):
        static_total = step()
except Exception as e:
    print("CAPTURE FAILED:", type(e).__name__, str(e)[:2000]); sys.exit(0)
torch.cuda.synchronize()
for _ in range(2): gr.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): gr.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 10 * 1e3, "loss", float(static_total))
