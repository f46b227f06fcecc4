#!/usr/bin/env python
"""which PYTHON lines of one eager DETR / SparseInst step launch copies, casts and fills?  torch_ops_probe.py cannot look
inside the mi355 custom ops and autograd Functions (the dispatch mode is off in there); this one wraps the Python entry
points (contiguous / to / float / clone / copy_ / zeros / zeros_like / new_zeros / zero_ / fill_ / cat / stack /
__setitem__ / add / mul ...) and counts the calls that really launch something, per call site inside yolov7_d2_amd.
usage: py_kernel_sites.py [detr|sparseinst]"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "detr"
agg = collections.defaultdict(lambda: [0, 0])
ON = [False]


def site():
    st = traceback.extract_stack()[:-2]
    ours = [f for f in st if "/yolov7_d2_amd/" in f.filename]
    return " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(ours[-3:])) or "(outside the package)"


def rec(name, t):
    if ON[0] and isinstance(t, torch.Tensor) and t.is_cuda:
        a = agg[(name, site())]
        a[0] += 1
        a[1] += t.numel() * t.element_size()


def wrap_method(name, launches):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        try:
            if launches(self, out, a, k):
                rec(name, out if isinstance(out, torch.Tensor) else self)
        except Exception:
            pass
        return out
    setattr(torch.Tensor, name, f)


def wrap_fn(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        out = orig(*a, **k)
        rec(name, out)
        return out
    setattr(mod, name, f)


moved = lambda s, o, a, k: isinstance(o, torch.Tensor) and o.data_ptr() != s.data_ptr()
always = lambda s, o, a, k: True
for n in ("contiguous", "to", "float", "bfloat16", "half", "long", "int", "reshape", "flatten"):
    wrap_method(n, moved)
for n in ("clone", "copy_", "zero_", "fill_", "__setitem__", "new_zeros", "new_full", "new_ones",
          "__add__", "__mul__", "__sub__", "__truediv__", "__radd__", "__rmul__", "add_", "mul_", "sum", "mean", "masked_fill",
          "masked_fill_", "index_put_", "sigmoid", "exp", "sqrt", "clamp"):
    wrap_method(n, always)
for n in ("zeros", "zeros_like", "ones", "ones_like", "full", "cat", "stack", "where", "arange"):
    wrap_fn(torch, n)

# ---- the model + one eager step, as torch_ops_probe.py builds them
import yolov7_d2_amd as M
from yolov7_d2_amd.d2shim import Boxes, Instances
dev = torch.device("cuda", 0)
torch.manual_seed(0)
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
static = model.prepare_batch(inputs)


def step():
    losses = model.forward_prepared(static)
    wd = getattr(getattr(model, "criterion", None), "weight_dict", None) if which == "detr" else None
    total = losses["total"] if "total" in losses else sum(v for k, v in losses.items() if wd is None or k in wd)
    opt.zero_grad(set_to_none=True)
    total.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
ON[0] = True
with torch.autograd.set_multithreading_enabled(False):
    step()
ON[0] = False
torch.cuda.synchronize()
print(f"{which}: {sum(v[0] for v in agg.values())} python-level launching calls in forward_prepared + backward + step")
for (name, s), (c, b) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{c:5d} {b / 1e6:9.2f} MB  {name:14s} {s}")
