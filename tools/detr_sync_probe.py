#!/usr/bin/env python
"""where does the eager DETR training step synchronise with the host?  (torch.cuda.set_sync_debug_mode('warn'))"""
import os, sys, time, warnings, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd.d2shim import Boxes, Instances
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, H_, W_ = 2, 512, 640
which = sys.argv[1] if len(sys.argv) > 1 else "detr"
if which == "detr":
    model = M.build_model(M.detr_r50_cfg(device="cuda:0"))
else:
    model = M.build_model(M.sparse_inst_r50_giam_cfg(device="cuda:0"))
model.train()
g = torch.Generator().manual_seed(1234)
inputs = []
for b in range(B):
    h, w = (H_, W_) if b == 0 else (H_ - 32, W_ - 64)
    n = 3
    wh = 16 + torch.rand(n, 2, generator=g) * 128
    xy = torch.rand(n, 2, generator=g) * (torch.tensor([w, h]) - wh).clamp(min=1)
    inst = Instances((h, w), gt_boxes=Boxes(torch.cat([xy, xy + wh], 1)), gt_classes=torch.randint(0, 80, (n,), generator=g))
    if which != "detr":
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
    opt.zero_grad(set_to_none=False)
    total.backward()
    opt.step()
    return total
for _ in range(2): step()
torch.cuda.synchronize()
seen = set()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = traceback.extract_stack()
    ours = [f for f in st if "/yolov7_d2_amd/" in f.filename or f.filename.endswith("detr_sync_probe.py")]
    key = tuple((f.filename, f.lineno) for f in ours[-3:])
    if key in seen: return
    seen.add(key)
    print("SYNC:", str(message)[:80], "<-", " | ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.line}" for f in ours[-3:]))
warnings.showwarning = showwarning
torch.cuda.set_sync_debug_mode("warn")
step()
torch.cuda.set_sync_debug_mode("default")
print("distinct sync sites:", len(seen))
