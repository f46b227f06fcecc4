#!/usr/bin/env python
"""BatchNorm backward, two-pass (reduce + apply) vs the one-launch fused kernel, per layer shape of YOLOX-s at 16x640x640.
Rotates over NSET buffer sets so that every launch reads from HBM.  MI_BN_FUSED_MODE selects the retention variant."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolov7_d2_amd import _lib as L

lib = L.lib()
DEV = "cuda"
shapes = [(32, 16 * 320 * 320), (64, 16 * 160 * 160), (32, 16 * 160 * 160), (128, 16 * 80 * 80), (64, 16 * 80 * 80),
          (256, 16 * 40 * 40), (128, 16 * 40 * 40), (512, 16 * 20 * 20), (256, 16 * 20 * 20)]
NSET = int(os.environ.get("NSET", 4)); IT = int(os.environ.get("IT", 24))
sp = L.stream_ptr()
print("capacity", lib.mi_bn_fused_set_capacity(int(os.environ.get("CAP", 0))), "mode", os.environ.get("MI_BN_FUSED_MODE", "1"))
for C, npix in shapes:
    ld = (C + 31) // 32 * 32
    sets = []
    for _ in range(NSET):
        da = torch.randn(npix, ld, device=DEV).to(torch.bfloat16)
        y = torch.randn(npix, ld, device=DEV).to(torch.bfloat16)
        dy = torch.empty(npix, ld, dtype=torch.bfloat16, device=DEV)
        sets.append((da, y, dy))
    mean = torch.zeros(C, device=DEV); invstd = torch.ones(C, device=DEV); gamma = torch.ones(C, device=DEV)
    scale = torch.ones(C, device=DEV); shift = torch.zeros(C, device=DEV)
    dacc = torch.zeros(L.MI_BN_SLOTS * ld * 2, dtype=torch.float64, device=DEV)
    dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
    bar = torch.zeros(L.MI_BN_BAR_WORDS, dtype=torch.int32, device=DEV)
    nblk = max(1, min(1024, math.ceil(npix / (256 // (C // 8)) / 4)))

    def two(s):
        da, y, dy = s
        cm = (da.data_ptr(), ld, y.data_ptr(), ld, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr())
        lib.mi_bn_act_bwd_reduce(*cm, dacc.data_ptr(), L.MI_BN_SLOTS, nblk, npix, C, 1, sp)
        lib.mi_bn_act_bwd_apply(*cm, gamma.data_ptr(), dacc.data_ptr(), L.MI_BN_SLOTS, npix, dg.data_ptr(), db.data_ptr(),
                                dy.data_ptr(), ld, None, 0, 0, npix, C, 1, sp)

    def fus(s):
        da, y, dy = s
        cm = (da.data_ptr(), ld, y.data_ptr(), ld, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr())
        L.check(lib.mi_bn_act_bwd_fused(*cm, gamma.data_ptr(), dacc.data_ptr(), L.MI_BN_SLOTS, npix, dg.data_ptr(), db.data_ptr(),
                                        dy.data_ptr(), ld, None, 0, 0, npix, C, 1, bar.data_ptr(), sp), "fused")

    res = {}
    for name, fn in (("two", two), ("fused", fus), ("two2", two), ("fused2", fus)):
        for i in range(3):
            fn(sets[i % NSET])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(IT):
            fn(sets[i % NSET])
        b.record(); torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) * 1000 / IT
    mb = npix * ld * 2 / 1e6
    print(f"C {C:4d} npix {npix:8d} tensor {mb:6.1f} MB  two-pass {res['two']:6.1f} {res['two2']:6.1f} us   fused {res['fused']:6.1f} {res['fused2']:6.1f} us"
          f"   fused GB/s {3 * mb / res['fused2'] * 1e-3 * 1e6 / 1e3:7.0f}   flag {int(bar[2])}", flush=True)
