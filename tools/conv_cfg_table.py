#!/usr/bin/env python
"""CPU-only: the launcher's (tile, BN, KC, TPS) choice for every conv of the last layer table (gpurun_out/layers*.log)"""
import ctypes as C, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolov7_d2_amd import _lib as L
lib = L.lib()
log = sys.argv[1] if len(sys.argv) > 1 else sorted([f for f in os.listdir(os.path.join(ROOT, "gpurun_out")) if f.startswith("layers")],
                                                    key=lambda f: int(re.findall(r"\d+", f)[0]))[-1]
seen = {}
for l in open(os.path.join(ROOT, "gpurun_out", log) if not os.path.isabs(log) else log):
    m = re.match(r"(\w+) (\S+)\s+CONV\s+([\d.]+)us.*N(\d+) (\d+)x(\d+) K(\d+) Co(\d+) t(\d+) is(\d) os(\d)", l)
    if not m:
        continue
    which, tag, us, N, H, W, K, Co, t, is_, os_ = m.groups()
    N, H, W, K, Co, t, is_, os_ = map(int, (N, H, W, K, Co, t, is_, os_))
    d = L.mi_conv_desc()
    d.x = d.w = d.y = 4096
    d.ldx, d.ldy, d.N, d.H, d.W = K, (Co + 31) // 32 * 32, N, H, W
    gh, gw = (H // is_, W // is_)
    d.outH, d.outW, d.gridH, d.gridW = gh * os_, gw * os_, gh, gw
    d.in_stride, d.out_stride, d.K8, d.Cout, d.CoutPad, d.ntaps = is_, os_, K // 8, Co, (Co + 31) // 32 * 32, t
    offs = {9: [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)], 4: [(a, b) for a in (0, 1) for b in (0, 1)], 2: [(0, 0), (0, 1)], 1: [(0, 0)]}[t]
    for i, (a, b) in enumerate(offs):
        d.tap_dy[i], d.tap_dx[i], d.tap_w[i] = a, b, i
    if which == "fwd" and Co not in (80, 4, 1):
        d.stats_acc = 4096
    else:
        d.flags = 0
    tiles = lib.mi_conv2d_plan(C.byref(d))
    tp = 64 if d.TH * d.TW <= 64 else 128
    steps = (K // d.KC) * (t // d.TPS)
    print(f"{which} {tag:36s} {float(us):6.1f}us {H:3d}x{W:<3d} K{K:<4d} Co{Co:<4d} t{t} s{is_}{os_} | tile {d.TH}x{d.TW} BN{d.BN} KC{d.KC} TPS{d.TPS} "
          f"steps {steps:3d} blocks {tiles * (d.CoutPad // d.BN)}")
