#!/usr/bin/env python
"""per-block phase timeline of selected convolutions of the step plan (diagnostic build: tools/build_variant.sh tl
-DMI_CONV_TIMELINE; run with MI355_LIB=yolov7_d2_amd/libmi355det_tl.so).  wall_clock64() ticks are 10 ns.
phases: 0 entry, 1 first loads issued, 2 first step landed (wait + barrier), 3 main loop done, 4 tile staged in LDS,
5 stores issued, 6 statistics done"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from bench import synth_batch_device
B, S = 16, 640
torch.manual_seed(0)
model = M.build_model(M.yolox_s_cfg(device="cuda")); model.train()
ps = model.plan_for(B, S, S, True)
imgs, labels = synth_batch_device(B, S, S, 1234, "cuda")
ps.image.copy_(imgs); ps.labels.copy_(labels); ps.gw().fill_(1.0)
plan = ps.plan
plan.run("fwd"); plan.run("bwd"); torch.cuda.synchronize()
lib = L.lib()
lib.mi_debug_conv_timeline.argtypes = [C.c_void_p]
lib.mi_debug_conv_timeline.restype = None
tl = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
want = sys.argv[1:] or ["head.stems.0.conv", "backbone.dark3.1.conv3.conv", "head.cls_convs.0.1.conv", "backbone.dark2.1.conv3.conv",
                        "backbone.dark4.1.m.0.conv2.conv", "backbone.dark2.0.conv", "neck.C3_p4.conv3.conv", "backbone.dark5.1.conv1.conv",
                        "backbone.dark3.0.conv"]
def t(d):
    cmd = (L.mi_cmd * 1)(); cmd[0].op = L.OP["CONV"]; cmd[0].p[0] = C.cast(C.pointer(d), C.c_void_p).value
    per = (C.c_float * 1)(); tot = C.c_float(0)
    L.check(lib.mi_cmdlist_time(cmd, 1, 10, C.byref(tot), per, L.stream_ptr()), "time")
    return tot.value * 1e3
for which in ("fwd", "bwd"):
    tags = plan.fwd_tags if which == "fwd" else plan.bwd_tags
    for k, tag in enumerate(tags):
        base = tag.split(".dgrad")[0]
        if not isinstance(plan.cmd_descs[which][k], L.mi_conv_desc):
            continue
        if "--list" in want:
            print(which, tag)
            continue
        if base in want or "--all" in want:
            d = L.mi_conv_desc.from_buffer_copy(plan.cmd_descs[which][k])
            lib.mi_debug_conv_timeline(None)
            us = t(d)
            tl.zero_()
            lib.mi_debug_conv_timeline(tl.data_ptr())
            t(d)
            torch.cuda.synchronize()
            lib.mi_debug_conv_timeline(None)
            a = tl.cpu().numpy().reshape(-1, 8)
            a = a[a[:, 0] > 0]
            if not len(a):
                continue
            t0 = a[:, 0].min()
            r = (a[:, :7] - t0) * 0.01            # us
            last = [k for k in range(7) if (a[:, k] > 0).all()]
            med = lambda v: float(np.median(v))
            ph = " ".join(f"{k}:{med(r[:, k]):5.2f}" for k in last)
            dur = " ".join(f"{last[i]}-{last[i+1]}:{med(r[:, last[i+1]] - r[:, last[i]]):5.2f}" for i in range(len(last) - 1))
            print(f"{which} {tag:44s} {us:6.1f}us blocks {len(a):5d} KC{d.KC} BN{d.BN} start p50/p90/max {med(r[:,0]):.2f}/{np.percentile(r[:,0],90):.2f}/{r[:,0].max():.2f}"
                  f" end p50/max {med(r[:,last[-1]]):.2f}/{r[:,last[-1]].max():.2f}\n      median t(phase) {ph}\n      median durations {dur}")
