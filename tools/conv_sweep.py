#!/usr/bin/env python
"""sweep (KC, BN, tile) of mi_conv2d over the distinct conv shapes of the YOLOX-s step; prints auto vs best"""
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
def t(d, it=5):
    cmd = (L.mi_cmd * 1)(); cmd[0].op = L.OP["CONV"]; cmd[0].p[0] = C.cast(C.pointer(d), C.c_void_p).value
    per = (C.c_float * 1)(); tot = C.c_float(0)
    rc = lib.mi_cmdlist_time(cmd, 1, it, C.byref(tot), per, L.stream_ptr())
    return tot.value * 1e3 if rc >= 0 else None
seen = {}
for which in ("fwd", "bwd"):
    arr, n = plan.fwd_cmds if which == "fwd" else plan.bwd_cmds
    for k in range(n):
        if L.OPS[arr[k].op] != "CONV": continue
        d = plan.cmd_descs[which][k]
        key = (d.H, d.W, d.K8 * 8, d.CoutPad, d.ntaps, d.in_stride, d.out_stride, bool(d.stats_acc), d.flags)
        seen.setdefault(key, []).append((which, k))
ta = tb = 0.0
def divisors(n):
    return [x for x in range(n, 0, -1) if n % x == 0]
for key, lst in sorted(seen.items(), key=lambda kv: -len(kv[1])):
    which, k = lst[0]
    d0 = plan.cmd_descs[which][k]
    dd = L.mi_conv_desc.from_buffer_copy(d0); lib.mi_conv2d_plan(C.byref(dd))
    auto = t(d0)
    res = []
    small = d0.gridH * d0.gridW <= 1600
    tiles = [(8, 16), (4, 32), (8, 8), (4, 16)] + ([(3, 40), (6, 20), (3, 20), (2, 20), (5, 20)] if small else [])
    for (th, tw) in tiles:
        if tw > d0.gridW * 2: continue
        for kc in (16, 32, 64, 128):
            if (d0.K8 * 8) % kc: continue
            for bn in (32, 64, 128):
                if d0.CoutPad % bn: continue
                for tps in divisors(d0.ntaps):
                    d = L.mi_conv_desc.from_buffer_copy(d0); d.KC, d.BN, d.TH, d.TW, d.TPS = kc, bn, th, tw, tps
                    r = t(d, 3)
                    if r is not None and r > 0: res.append((r, kc, bn, th, tw, tps))
    best = min(res) if res else (auto, 0, 0, 0, 0, 0)
    ta += auto * len(lst); tb += min(best[0], auto) * len(lst)
    print(f"{key} x{len(lst)}: auto {auto:.1f}us (KC{dd.KC} BN{dd.BN} {dd.TH}x{dd.TW} T{dd.TPS}) | best " +
          " ".join(f"{r[0]:.1f}:KC{r[1]}/BN{r[2]}/{r[3]}x{r[4]}/T{r[5]}" for r in sorted(res)[:6]), flush=True)
print(f"total auto {ta/1e3:.3f} ms, best {tb/1e3:.3f} ms")
