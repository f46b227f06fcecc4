#!/usr/bin/env python
"""what does ONE pixel-tile step of the 1x1 weight-gradient kernel cost, and what is the launch's fixed part?  A single
mi_conv2d_wgrad on ResNet-shaped 1x1 layers with forced split counts: time against tiles per block -> slope and intercept.
usage: wgrad_kstep_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolov7_d2_amd import _lib as L
lib = L.lib()
dev = "cuda"
ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def run(N, H, W, Cin, Cout, stride, splits, tp=0, ns=0):
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
    gw = torch.empty(Cout, Cin, dtype=torch.float32, device=dev)
    out = []
    for sk in splits:
        d = L.mi_wgrad_desc()
        d.x, d.dy, d.gw = x.data_ptr(), dy.data_ptr(), gw.data_ptr()
        d.ldx, d.ldy, d.N, d.H, d.W, d.outH, d.outW, d.stride = Cin, Cout, N, H, W, Ho, Wo, stride
        d.Cin, d.Cout, d.CinPad, d.CoutPad, d.ntaps = Cin, Cout, Cin, Cout, 1
        d.splitk, d.cfg_tp, d.cfg_ns = sk, tp, ns
        need = lib.mi_conv2d_wgrad_plan(C.byref(d))
        if need < 0 or need > ws.numel():
            continue
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
        cmd = (L.mi_cmd * 1)()
        cmd[0].op = L.OP["WGRAD"]
        cmd[0].p[0] = C.cast(C.pointer(d), C.c_void_p).value
        per = (C.c_float * 1)(); tot = C.c_float(0)
        if lib.mi_cmdlist_time(cmd, 1, 10, C.byref(tot), per, L.stream_ptr()) < 0:
            continue
        npix = N * Ho * Wo
        tiles = (npix + 63) // 64
        pairs = (Cout // 128) * (Cin // 128)
        out.append((sk, pairs * sk, (tiles + sk - 1) // sk, per[0] * 1e3))
    return out


for name, args in (("res5 identity conv1 2048->512 @4x25x42", (4, 25, 42, 2048, 512, 1)),
                   ("res5 identity conv3 512->2048 @4x25x42", (4, 25, 42, 512, 2048, 1)),
                   ("res5.0 conv1 1024->512 @4x50x84", (4, 50, 84, 1024, 512, 1)),
                   ("res5.0 shortcut 1024->2048 s2 @4x50x84", (4, 50, 84, 1024, 2048, 2)),
                   ("encoder linear1 256->2048 T=4368", (1, 52, 84, 256, 2048, 1))):
    for ns in (0, 2, 4):
        r = run(*args, splits=(1, 2, 4, 8, 16, 32), ns=ns)
        print(f"{name:44s} ns={ns}: " + "  ".join(f"s{sk}: {b} blk x {t} tiles {us:6.1f}us" for sk, b, t, us in r))
