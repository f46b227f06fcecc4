#!/usr/bin/env python
"""which lines of the eager DETR / SparseInst step launch torch's own elementwise kernels (fill / copy / add / mul ...)?
A TorchDispatchMode records every aten op on a device tensor with the innermost frames inside yolov7_d2_amd (backward runs
on the calling thread so that the mode sees it); prints launches and output bytes per (op, call site) for ONE eager step at
the bench shape.  usage: torch_ops_probe.py [detr|sparseinst]"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import yolov7_d2_amd as M
from yolov7_d2_amd.d2shim import Boxes, Instances

dev = torch.device("cuda", 0)
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "detr"
B, H_, W_ = (4, 800, 1333) if which == "detr" else (8, 640, 640)
model = M.build_model(M.detr_r50_cfg(device="cuda:0") if which == "detr" else M.sparse_inst_r50_giam_cfg(device="cuda:0"))
model.train()
g = torch.Generator().manual_seed(1234)
inputs = []
for b in range(B):
    h, w = (H_, W_) if (b == 0 or which != "detr") else (H_ - 32 * (b % 2), W_ - 64 * (b % 3))
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
from yolov7_d2_amd.optim import MultiTensorAdamW
opt = MultiTensorAdamW(params, lr=1e-4, weight_decay=1e-4)


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
NOKERNEL = ("view", "permute", "reshape", "slice", "select", "expand", "transpose", "detach", "alias", "as_strided", "unsqueeze",
            "squeeze", "empty", "t.default", "unbind", "split", "_unsafe_view", "unfold", "narrow", "flatten", "chunk", "lift_fresh",
            "is_", "sym_", "_local_scalar", "stride", "size", "numel", "dim", "result_type", "_has_compatible")
agg = collections.defaultdict(lambda: [0, 0])


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if any(s in name for s in NOKERNEL):
            return out
        t = out if isinstance(out, torch.Tensor) else (out[0] if isinstance(out, (tuple, list)) and out and isinstance(out[0], torch.Tensor) else None)
        if t is None or not t.is_cuda:
            return out
        st = traceback.extract_stack()
        ours = [f for f in st if "/yolov7_d2_amd/" in f.filename]
        site = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(ours[-2:])) or "(native autograd node / probe)"
        a = agg[(name, site)]
        a[0] += 1
        a[1] += t.numel() * t.element_size()
        return out


with torch.autograd.set_multithreading_enabled(False), Rec():
    step()
torch.cuda.synchronize()
n_all = sum(v[0] for v in agg.values())
print(f"{which}: {n_all} torch kernel-launching ops in one eager step")
print("--- by output bytes")
for (name, site), (n, byt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{byt / 1e6:9.2f} MB {n:5d}  {name:26s} {site}")
print("--- by count")
for (name, site), (n, byt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{byt / 1e6:9.2f} MB {n:5d}  {name:26s} {site}")
