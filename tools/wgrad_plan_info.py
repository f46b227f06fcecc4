#!/usr/bin/env python
"""(CPU) sizes of the grouped wgrad plan of the YOLOX-s B=16 640x640 step: workspace, groups, blocks"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.modeling.yolox import _PlanState
from yolov7_d2_amd.params import ParamArena
model = M.build_model(M.yolox_s_cfg(device="cpu")); model.train()
model.params = ParamArena(model, "cpu")
ps = _PlanState(model, 16, 640, 640, True, materialize=False)
b = ps.builder
wg = [c for c in b.bwd if c.op == L.OP["WGRAD"]]
descs = (L.mi_wgrad_desc * len(wg))()
for d, c in zip(descs, wg):
    C.memmove(C.byref(d), C.byref(b._wgrad_desc(c.desc)), C.sizeof(L.mi_wgrad_desc))
meta = L.mi_wgrad_group()
L.check(L.lib().mi_conv2d_wgrad_group_plan(descs, len(wg), None, None, 0, C.byref(meta)), "plan")
print("ws MB", meta.ws_bytes / 1e6, "groups", meta.ngroups, "reduce blocks", meta.red_blocks)
for g in range(meta.ngroups):
    gg = meta.g[g]
    print(list(gg.cfg), "jobs", gg.njobs, "blocks", gg.nblocks, "lds", gg.lds_bytes)
