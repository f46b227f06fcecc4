#!/usr/bin/env python
"""per-command HIP-event timings of the YOLOX-s 640x640 B=16 step plan (runs on the GPU box)"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from bench import synth_batch_device, conv_algorithmic

B = int(os.environ.get("B", 16)); S = int(os.environ.get("S", 640))
torch.manual_seed(0)
model = M.build_model(M.yolox_s_cfg(device="cuda"))
model.train()
ps = model.plan_for(B, S, S, True)
imgs, labels = synth_batch_device(B, S, S, 1234, "cuda")
ps.image.copy_(imgs); ps.labels.copy_(labels); ps.gw().fill_(1.0)
plan = ps.plan
plan.run("fwd"); plan.run("bwd"); torch.cuda.synchronize()
rows = []
for which in ("fwd", "bwd"):
    tot, per = plan.time_cmds(which, iters=5)
    arr, n = plan.fwd_cmds if which == "fwd" else plan.bwd_cmds
    descs = plan.cmd_descs[which]
    for k in range(n):
        op = L.OPS[arr[k].op]; tag, ms = per[k]
        info = ""; fl = 0; byt = 0
        if op == "CONV":
            d = descs[k]
            dd = L.mi_conv_desc.from_buffer_copy(d); L.lib().mi_conv2d_plan(C.byref(dd))
            byt, fl = conv_algorithmic(d)
            info = f"N{d.N} {d.H}x{d.W} K{d.K8*8} Co{d.Cout} t{d.ntaps} is{d.in_stride} os{d.out_stride} tile{dd.TH}x{dd.TW} KC{dd.KC} BN{dd.BN}"
        elif op == "WGRAD":
            d = descs[k]
            npx = d.N * d.outH * d.outW
            fl = 2.0 * npx * d.CoutPad * d.CinPad * d.ntaps
            byt = d.N * d.H * d.W * d.CinPad * 2 + npx * d.CoutPad * 2
            info = f"N{d.N} {d.H}x{d.W} Ci{d.CinPad} Co{d.CoutPad} t{d.ntaps} s{d.stride}"
        rows.append((which, tag, op, ms, fl, byt, info))
    print(which, "total ms", tot)
rows.sort(key=lambda r: -r[3])
tot = sum(r[3] for r in rows)
print("sum of per-cmd ms", tot)
for r in rows[:int(os.environ.get("TOP", 70))]:
    tf = r[4] / (r[3] * 1e-3) / 1e12 if r[4] else 0
    gb = r[5] / (r[3] * 1e-3) / 1e9 if r[5] else 0
    print(f"{r[0]} {r[1]:42s} {r[2]:14s} {r[3]*1e3:8.1f}us {tf:6.1f}TF {gb:7.0f}GB/s  {r[6]}")
byop = {}
for r in rows:
    byop.setdefault(r[2], [0, 0]); byop[r[2]][0] += r[3]; byop[r[2]][1] += 1
for k, v in sorted(byop.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:16s} {v[0]:8.3f} ms {v[1]:4d}")
