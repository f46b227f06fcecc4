#!/usr/bin/env python
"""BN_ACT_FWD / BN_BWD_APPLY with and without the statistics prologue, per layer"""
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
def t(cmd, it=20):
    arr = (L.mi_cmd * 1)(); C.memmove(arr, C.byref(cmd), C.sizeof(L.mi_cmd))
    per = (C.c_float * 1)(); tot = C.c_float(0)
    L.check(lib.mi_cmdlist_time(arr, 1, it, C.byref(tot), per, L.stream_ptr()), "time")
    return tot.value * 1e3
arr, n = plan.fwd_cmds
tot_a = tot_b = 0
rows = []
for k in range(n):
    if L.OPS[arr[k].op] != "BN_ACT_FWD": continue
    c = L.mi_cmd(); C.memmove(C.byref(c), C.byref(arr[k]), C.sizeof(L.mi_cmd))
    ta = t(c)
    c.p[1] = None   # eval mode: no prologue (scale/shift were published by the train-mode run)
    tb = t(c)
    rows.append((plan.fwd_tags[k], c.l[1], c.i[3], ta, tb))
    tot_a += ta; tot_b += tb
for r in sorted(rows, key=lambda r: -r[3])[:12]:
    print(f"{r[0]:40s} npix {r[1]:8d} C {r[2]:4d}: with prologue {r[3]:6.1f}us  without {r[4]:6.1f}us")
print(f"total with {tot_a/1e3:.3f} ms, without {tot_b/1e3:.3f} ms")
