#!/usr/bin/env python
"""GPU box, diagnostic build (tools/build_variant.sh tl "-DMI_WG_TIMELINE" conv_wgrad; MI355_LIB=.../libmi355det_tl.so):
where does a 64-pixel step of the 1x1 weight-gradient kernel spend its time?  Wave 0 of block 0 stamps five points per step
(wall_clock64, 100 MHz): loop top -> counted wait -> barrier -> refill issued -> multiplied."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolov7_d2_amd import _lib as L
lib = L.lib()
dev = "cuda"
ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

def run(name, N, H, W, Cin, Cout, stride, sk, ns):
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
    gw = torch.empty(Cout, Cin, dtype=torch.float32, device=dev)
    d = L.mi_wgrad_desc()
    d.x, d.dy, d.gw = x.data_ptr(), dy.data_ptr(), gw.data_ptr()
    d.ldx, d.ldy, d.N, d.H, d.W, d.outH, d.outW, d.stride = Cin, Cout, N, H, W, Ho, Wo, stride
    d.Cin, d.Cout, d.CinPad, d.CoutPad, d.ntaps = Cin, Cout, Cin, Cout, 1
    d.splitk, d.cfg_tp, d.cfg_ns = sk, 0, ns
    need = lib.mi_conv2d_wgrad_plan(C.byref(d))
    assert 0 <= need <= ws.numel(), need
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
    for _ in range(3):
        L.check(lib.mi_conv2d_wgrad(C.byref(d), L.stream_ptr()), "wgrad")
    torch.cuda.synchronize()
    buf = (C.c_longlong * 1000)()
    L.check(lib.mi_debug_wg_timeline(buf), "timeline")
    for role, o in (("wave 0", 0),):
        t = [[buf[o + i * 5 + k] for k in range(5)] for i in range(200)]
        steps = [r for r in t[4:60] if r[4] > r[0] > 0]
        if len(steps) < 3:
            continue
        names = ("counted wait", "barrier", "refill issue", "multiply")
        avg = [sum(r[k + 1] - r[k] for r in steps) / len(steps) * 10 for k in range(4)]       # ns (100 MHz ticks)
        gap = sum(steps[i + 1][0] - steps[i][4] for i in range(len(steps) - 1)) / (len(steps) - 1) * 10
        tot = sum(steps[i + 1][0] - steps[i][0] for i in range(len(steps) - 1)) / (len(steps) - 1) * 10
        print(f"{name} splits {sk} ns {ns} {role}: step {tot:6.0f} ns = " + "  ".join(f"{n} {a:5.0f}" for n, a in zip(names, avg)) + f"  loop edge {gap:4.0f}", flush=True)

for ns in (0, 3, 4):
    run("res5.0 conv1 1024->512 @4x50x84", 4, 50, 84, 1024, 512, 1, 1, ns)
    run("encoder linear1 256->2048 T=4368", 1, 52, 84, 256, 2048, 1, 1, ns)
