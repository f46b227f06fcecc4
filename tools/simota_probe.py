#!/usr/bin/env python
"""GPU box: mi_yolox_loss_fwd alone (HIP events) with the round-5 / round-6 forms of the cost and dynamic-k kernels, on a
random head (the bench's: k = 1 mostly) and on a half-trained one (high IoUs, k up to 10)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import yolox_oracle as O
from yolov7_d2_amd import _lib as L
DEV = "cuda"
B, H, W = 16, 640, 640
hw = [(H // s, W // s) for s in (8, 16, 32)]
_, labels = O.synth_batch(B, H, W, seed=1234, max_gt=20)
for name, lab_for_raw in (("random head", None), ("half-trained head", labels)):
    raw, anchors = O.synth_raw(B, hw, 35, labels=lab_for_raw)
    A, nch = raw.shape[1], raw.shape[2]
    t = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)
    ws = dict(cost=t(B, 100, A), iou=t(B, 100, A), match=t(1, dt=torch.uint8), ngt=t(B, dt=torch.int32), fg=t(B, A, dt=torch.uint8),
              matched_gt=t(B, A, dt=torch.int32), matched_iou=t(B, A), partial=t(B * ((A + 255) // 256), 4), out=t(8))
    rd, ld, ad = raw.to(DEV), labels.to(DEV), anchors.to(DEV)
    d = L.mi_yolox_loss_desc()
    d.preds, d.labels, d.anchors = rd.data_ptr(), ld.data_ptr(), ad.data_ptr()
    d.B, d.A, d.ncls, d.max_labels, d.gmax = B, A, nch - 5, 100, 100
    for k in ws: setattr(d, k, ws[k].data_ptr())
    for c, pf, dbg in (("0", "0", "0"), ("1", "0", "0"), ("2", "0", "0"), ("0", "1", "0"), ("1", "1", "0"), ("2", "1", "0"), ("2", "1", "1"), ("2", "1", "2"), ("2", "1", "3"), ("2", "1", "4"), ("2", "1", "5"), ("2", "1", "6")):
        os.environ["MI_SIMOTA_COMPACT"], os.environ["MI_SIMOTA_PREFILTER"], os.environ["MI_SIMOTA_DBG"] = c, pf, dbg
        call = lambda: L.check(L.lib().mi_yolox_loss_fwd(C.byref(d), L.stream_ptr()), "loss")
        for _ in range(5): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): call()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:18s} compact={c} prefilter={pf} dbg={dbg}: {e0.elapsed_time(e1) * 20:7.1f} us per loss forward (4 launches), num_fg {float(ws['out'][6]):.0f}", flush=True)
