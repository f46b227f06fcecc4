#!/usr/bin/env python
"""GPU box: the YOLOX-s B=16 640^2 step plan's command tags in launch order (for tools/trace_table.py)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
model = M.build_model(M.yolox_s_cfg(device="cuda")); model.train()
ps = model.plan_for(16, 640, 640, True)
plan = ps.plan
for which in ("fwd", "bwd"):
    arr, n = plan.fwd_cmds if which == "fwd" else plan.bwd_cmds
    tags = plan.fwd_tags if which == "fwd" else plan.bwd_tags
    for k in range(n):
        op = L.OPS[arr[k].op]
        info = ""
        if op == "CONV":
            d = plan.cmd_descs[which][k]
            dd = L.mi_conv_desc.from_buffer_copy(d); L.lib().mi_conv2d_plan(C.byref(dd))
            info = f"{d.H}x{d.W} K{d.K8*8} Co{d.Cout} t{d.ntaps} s{d.in_stride}{d.out_stride} tile{dd.TH}x{dd.TW} BN{dd.BN} KC{dd.KC} T{dd.TPS}"
        elif op == "BN_ACT_FWD":
            info = f"count{arr[k].l[1]} C{arr[k].i[3]} res{int(arr[k].i[1] > 0)}"
        elif op == "BN_BWD_REDUCE":
            info = f"count{arr[k].l[0]} C{arr[k].i[3]} res0"
        elif op == "BN_BWD_APPLY":
            info = f"count{arr[k].l[0]} C{arr[k].i[2]} res{int(arr[k].i[3] > 0)}"
        print(which, k, op, tags[k], info)
