#!/usr/bin/env python
"""time the weight-stationary 3x3 kernel alone (C-ABI, HIP events): full / without halo traffic / without main loop"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from yolov7_d2_amd import _lib as L
from test_gpu_conv3x3_ws import _pack, _desc, sp
DEV = "cuda"
def run(K, jobs, mode, iters=20):
    g = torch.Generator().manual_seed(0)
    descs = (L.mi_conv_desc * len(jobs))(); keep = []
    fl = 0.0
    for j, (N, H, W) in enumerate(jobs):
        x = torch.randn(N, H, W, K, generator=g).to(DEV, torch.bfloat16)
        w = (torch.randn(K, K, 3, 3, generator=g) / (3 * K ** .5)).to(DEV)
        img = _pack(w, False); y = torch.empty(N, H, W, K, dtype=torch.bfloat16, device=DEV)
        st = torch.zeros(16, K, 2, dtype=torch.float64, device=DEV) if mode == "stats" else None
        d = _desc(x, K, 0, N, H, W, K, img, y, K, 0, False, st, L.MI_CONV_ACCUM if mode == "accum" else 0)
        C.memmove(C.byref(descs[j]), C.byref(d), C.sizeof(d)); keep += [x, w, img, y, st]
        fl += 2.0 * N * H * W * K * K * 9
    out = []
    for env in ({"MI_CONV_WS": "0"}, {}, {"MI_W3_DBG": "2"}, {"MI_W3_DBG": "4"}, {"MI_W3_DBG": "6"}):
        for k in ("MI_CONV_WS", "MI_W3_DBG"): os.environ.pop(k, None)
        os.environ.update(env)
        def call():
            if env.get("MI_CONV_WS") == "0":
                for j in range(len(jobs)): L.check(L.lib().mi_conv2d(C.byref(descs[j]), sp()), "conv2d")
            else:
                L.check(L.lib().mi_conv3x3_ws(descs, len(jobs), sp()), "ws")
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out.append(f"{'tile' if env.get('MI_CONV_WS') else 'ws' + env.get('MI_W3_DBG', ''):5s} {us:7.1f}us {fl / us / 1e6:6.0f}TF")
    for k in ("MI_CONV_WS", "MI_W3_DBG"): os.environ.pop(k, None)
    print(f"K{K} {jobs} {mode}: " + " | ".join(out), flush=True)
for mode in ("plain", "stats"):
    run(128, [(16, 80, 80)], mode)
    run(128, [(16, 40, 40)], mode)
    run(128, [(16, 80, 80), (16, 40, 40), (16, 20, 20)] * 2, mode)
    run(64, [(16, 80, 80)], mode)
    run(32, [(16, 160, 160)], mode)
    run(32, [(16, 320, 320)], mode)
run(128, [(64, 80, 80)], "plain")
