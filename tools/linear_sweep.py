#!/usr/bin/env python
"""sweep (KC, BN, pixel tile) of mi_conv2d over the token-row GEMMs of DETR's transformer (nn.Linear as a 1x1 convolution
over T = L * B rows; modeling/transformer.py::_conv1x1): auto configuration vs the best of the sweep, per shape."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.modeling.transformer import _factor
lib = L.lib()
dev = "cuda"


def t(d, it=20):
    cmd = (L.mi_cmd * 1)(); cmd[0].op = L.OP["CONV"]; cmd[0].p[0] = C.cast(C.pointer(d), C.c_void_p).value
    per = (C.c_float * 1)(); tot = C.c_float(0)
    rc = lib.mi_cmdlist_time(cmd, 1, it, C.byref(tot), per, L.stream_ptr())
    return tot.value * 1e3 if rc >= 0 else None


shapes = [(4368, 256, 256, True), (400, 256, 256, True), (4368, 256, 2048, True), (4368, 2048, 256, True), (400, 256, 2048, True),
          (400, 2048, 256, True), (4368, 256, 256, False), (400, 256, 256, False), (4200, 256, 256, True)]
for T, K, Cout, bias in shapes:
    H, W = _factor(T)
    x = torch.randn(T, K, device=dev).to(torch.bfloat16)
    w = torch.randn(K // 8 * Cout * 8, device=dev).to(torch.bfloat16)
    y = torch.empty(T, Cout, dtype=torch.bfloat16, device=dev)
    b = torch.randn(Cout, device=dev)
    d0 = L.mi_conv_desc()
    d0.x, d0.w, d0.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    d0.bias = b.data_ptr() if bias else None
    d0.ldx, d0.ldy = K, Cout
    d0.N, d0.H, d0.W, d0.outH, d0.outW, d0.gridH, d0.gridW = 1, H, W, H, W, H, W
    d0.in_stride = d0.out_stride = 1
    d0.K8, d0.Cout, d0.CoutPad, d0.ntaps = K // 8, Cout, Cout, 1
    dd = L.mi_conv_desc.from_buffer_copy(d0); lib.mi_conv2d_plan(C.byref(dd))
    auto = t(d0)
    res = []
    for (th, tw) in [(8, 16), (4, 16), (2, 16), (1, 16), (16, 16), (4, 32), (8, 8), (3, 40), (6, 20)]:
        if tw > W * 2 and tw != 16: continue
        for kc in (16, 32, 64, 128):
            if K % kc: continue
            for bn in (32, 64, 128):
                if Cout % bn: continue
                d = L.mi_conv_desc.from_buffer_copy(d0); d.KC, d.BN, d.TH, d.TW, d.TPS = kc, bn, th, tw, 1
                r = t(d, 10)
                if r is not None and r > 0: res.append((r, kc, bn, th, tw))
    print(f"T{T} ({H}x{W}) K{K} Co{Cout} bias{int(bias)}: auto {auto:.1f}us (KC{dd.KC} BN{dd.BN} {dd.TH}x{dd.TW}) | best " +
          " ".join(f"{r[0]:.1f}:KC{r[1]}/BN{r[2]}/{r[3]}x{r[4]}" for r in sorted(res)[:5]), flush=True)
