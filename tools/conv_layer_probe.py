#!/usr/bin/env python
"""per-call table of every convolution-kernel launch (mi_conv2d: forward / data gradient; mi_conv2d_wgrad) of ONE eager DETR /
SparseInst step: geometry from the descriptor, duration from HIP events around the call, TFLOP/s and algorithmic GB/s.
usage: conv_layer_probe.py [detr|sparseinst] [top]"""
import collections, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.d2shim import Boxes, Instances
which = sys.argv[1] if len(sys.argv) > 1 else "detr"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, H_, W_ = (4, 800, 1333) if which == "detr" else (8, 640, 640)
model = M.build_model(M.detr_r50_cfg(device="cuda:0") if which == "detr" else M.sparse_inst_r50_giam_cfg(device="cuda:0"))
model.train()
g = torch.Generator().manual_seed(1234)
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
from yolov7_d2_amd.optim import MultiTensorAdamW
opt = MultiTensorAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
static = model.prepare_batch(inputs)


def step():
    losses = model.forward_prepared(static)
    total = losses["total"] if "total" in losses else sum(losses.values())
    opt.zero_grad(set_to_none=True)
    total.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
lib = L.lib()
rows = []
orig_conv, orig_wg = lib.mi_conv2d, lib.mi_conv2d_wgrad


def timed(fn, args):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = fn(*args)
    b.record()
    return rc, (a, b)


def conv(dref, st):
    d = dref._obj
    rc, ev = timed(orig_conv, (dref, st))
    px = d.N * d.gridH * d.gridW
    fl = 2.0 * px * d.Cout * d.K8 * 8 * d.ntaps
    by = 2.0 * (d.N * d.H * d.W * d.K8 * 8 + px * d.Cout)
    rows.append(("conv", f"N{d.N} {d.H}x{d.W}->{d.outH}x{d.outW} K{d.K8 * 8} Co{d.Cout} t{d.ntaps} s{d.in_stride}{d.out_stride} f{d.flags}", fl, by, ev))
    return rc


def wg(dref, st):
    d = dref._obj
    rc, ev = timed(orig_wg, (dref, st))
    px = d.N * d.outH * d.outW
    fl = 2.0 * px * d.Cout * d.Cin * d.ntaps
    by = 2.0 * (d.N * d.H * d.W * d.Cin + px * d.Cout) + 4.0 * d.Cout * d.Cin * d.ntaps
    rows.append(("wgrad", f"N{d.N} {d.H}x{d.W}->{d.outH}x{d.outW} Ci{d.Cin} Co{d.Cout} t{d.ntaps} s{d.stride}", fl, by, ev))
    return rc


lib.mi_conv2d, lib.mi_conv2d_wgrad = conv, wg
with torch.autograd.set_multithreading_enabled(False):
    step()
torch.cuda.synchronize()
lib.mi_conv2d, lib.mi_conv2d_wgrad = orig_conv, orig_wg
agg = collections.OrderedDict()
for kind, desc, fl, by, (a, b) in rows:
    e = agg.setdefault((kind, desc), [0, 0.0, fl, by])
    e[0] += 1
    e[1] += a.elapsed_time(b) * 1e3
tot = sum(v[1] for v in agg.values())
print(f"# {which}: {len(rows)} conv-kernel calls in one eager step, {tot / 1e3:.2f} ms between their events "
      f"(each includes ~5-10 us of eager launch overhead); total {sum(v[2] * v[0] for v in agg.values()) / 1e12:.3f} TFLOP")
print(f"# {'kind':5s} {'geometry':58s} {'n':>3s} {'us each':>8s} {'TFLOP/s':>8s} {'GB/s':>7s} {'% time':>6s}")
for (kind, desc), (n, t, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    each = t / n
    print(f"{kind:7s} {desc:58s} {n:3d} {each:8.1f} {fl / each / 1e6:8.1f} {by / each / 1e3:7.0f} {100 * t / tot:6.1f}")
