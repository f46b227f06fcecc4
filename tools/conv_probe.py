#!/usr/bin/env python
"""time selected conv commands of the step plan with debug flags (256: no main loop, 512: no epilogue)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
want = sys.argv[1:] or ["head.stems.0.conv", "backbone.dark3.1.conv3.conv", "head.cls_convs.0.1.conv", "backbone.dark2.1.conv3.conv",
                        "backbone.dark4.1.m.0.conv2.conv", "backbone.stem.conv.conv", "neck.C3_p4.conv3.conv", "backbone.dark5.1.conv1.conv"]
def t(d):
    cmd = (L.mi_cmd * 1)(); cmd[0].op = L.OP["CONV"]; cmd[0].p[0] = C.cast(C.pointer(d), C.c_void_p).value
    per = (C.c_float * 1)(); tot = C.c_float(0)
    L.check(lib.mi_cmdlist_time(cmd, 1, 10, C.byref(tot), per, L.stream_ptr()), "time")
    return tot.value * 1e3
for which in ("fwd", "bwd"):
    tags = plan.fwd_tags if which == "fwd" else plan.bwd_tags
    for k, tag in enumerate(tags):
        if tag in want:
            d0 = plan.cmd_descs[which][k]
            out = []
            for fl in (0, 256, 512):
                d = L.mi_conv_desc.from_buffer_copy(d0); d.flags |= fl
                out.append(t(d))
            d = L.mi_conv_desc.from_buffer_copy(d0); d.stats_acc = None
            out.append(t(d))
            var = []
            for (kc, bn, th, tw) in ((64, 128, 8, 16), (64, 64, 8, 16), (32, 128, 8, 16), (64, 64, 8, 8), (64, 128, 8, 8), (64, 32, 8, 16), (32, 64, 8, 16)):
                d = L.mi_conv_desc.from_buffer_copy(d0); d.KC, d.BN, d.TH, d.TW = kc, bn, th, tw
                d.stats_acc = None
                try:
                    var.append(f"KC{kc}/BN{bn}/{th}x{tw}:{t(d):.1f}")
                except Exception as e:
                    var.append(f"KC{kc}/BN{bn}/{th}x{tw}:ERR")
            print(f"{tag:36s} full {out[0]:.1f}us  no-mainloop {out[1]:.1f}us  no-epilogue {out[2]:.1f}us  no-stats {out[3]:.1f}us | " + " ".join(var[:0]))
