#!/usr/bin/env python
"""which lines of the eager DETR / SparseInst step launch the torch elementwise kernels (fill / copy / add / mul ...)?
torch.profiler with stacks, one eager step at the bench shape; prints device time and launch count per (aten op, innermost
frames inside yolov7_d2_amd).  usage: torch_ops_probe.py [detr|sparseinst]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd.d2shim import Boxes, Instances
from torch.profiler import ProfilerActivity, profile

dev = torch.device("cuda", 0)
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "detr"
B, H_, W_ = (4, 800, 1333) if which == "detr" else (8, 640, 640)
model = M.build_model(M.detr_r50_cfg(device="cuda:0") if which == "detr" else M.sparse_inst_r50_giam_cfg(device="cuda:0"))
model.train()
g = torch.Generator().manual_seed(1234)
inputs = []
for b in range(B):
    h, w = (H_, W_) if b == 0 else (H_ - 32 * (b % 2), W_ - 64 * (b % 3))
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
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, capturable=True)


def step():
    losses = model(inputs)
    wd = getattr(getattr(model, "criterion", None), "weight_dict", None) if which == "detr" else None
    total = sum(v for k, v in losses.items() if wd is None or k in wd)
    opt.zero_grad(set_to_none=True)
    total.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
tot = 0.0
for e in prof.events():
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0 or not e.name.startswith("aten::"):
        continue
    ours = [f for f in (e.stack or []) if "yolov7_d2_amd/" in f or "tools/" in f]
    site = " < ".join(s.split("yolov7_d2_amd/")[-1][:70] for s in ours[:2]) or "(autograd engine / optimizer)"
    k = (e.name, site)
    agg[k][0] += dt
    agg[k][1] += 1
    tot += dt
print(f"{which}: aten:: device time of one eager step {tot / 1e3:.2f} ms")
for (name, site), (dt, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{dt:9.1f} us {n:5d}  {name:28s} {site}")
