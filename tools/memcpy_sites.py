#!/usr/bin/env python
"""which calls of one eager DETR / SparseInst step end in a MEMCPY (hipMemcpyAsync: a graph memcpy node when the step is
captured - the in-graph traces show ~10 us of idle device on either side of each one)?  torch.profiler with stacks: every CPU
op that owns a device-side Memcpy / copyBuffer activity, by Python call site inside yolov7_d2_amd; the C library's own
copies (mi_upload_async, COPY commands) have no aten op and are counted by wrapping the ctypes entry points.
usage: memcpy_sites.py [detr|sparseinst]"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "detr"

import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
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

# the library's own copies
lib_calls = collections.Counter()
ON = [False]


def site():
    st = traceback.extract_stack()[:-2]
    ours = [f for f in st if "/yolov7_d2_amd/" in f.filename]
    return " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(ours[-4:])) or "(outside the package)"


lib = L.lib()
for name in ("mi_upload_async",):
    orig = getattr(lib, name)

    def f(*a, _o=orig, _n=name):
        if ON[0]:
            lib_calls[(_n, site())] += 1
        return _o(*a)
    setattr(lib, name, f)

from torch.profiler import profile, ProfilerActivity
ON[0] = True
with torch.autograd.set_multithreading_enabled(False):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
ON[0] = False
agg = collections.defaultdict(lambda: [0, 0.0, ""])
ndev = 0
for e in prof.events():
    ks = [k for k in (e.kernels or []) if "emcpy" in k.name or "copyBuffer" in k.name or "emset" in k.name]
    if not ks:
        continue
    ours = [s for s in (e.stack or []) if "/yolov7_d2_amd/" in s]
    where = " < ".join(s.split("/yolov7_d2_amd/")[-1].split(" ")[0].replace(".py(", ".py:").rstrip(")") for s in ours[:4]) or "(no package frame)"
    a = agg[(e.name, ks[0].name[:24], where)]
    a[0] += len(ks)
    a[1] += sum(k.duration for k in ks)
    a[2] = str(e.input_shapes)[:80]
    ndev += len(ks)
print(f"{which}: {ndev} device-side memcpy / memset activities owned by aten ops in one eager step (forward_prepared + backward + step)")
for (n, kn, w), (c, us, shp) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{c:4d} {us:8.1f} us  {n:22s} {kn:24s} {shp:80s} {w}")
print(f"library copies (no aten op): {sum(lib_calls.values())}")
for (n, w), c in lib_calls.most_common():
    print(f"{c:4d}  {n:18s} {w}")
