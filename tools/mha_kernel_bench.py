#!/usr/bin/env python
"""kernel-only throughput of the attention core through the C-ABI (20 launches between HIP events, no Python autograd)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov7_d2_amd import _lib as L
H, E = 8, 256
for (Lq, Lk, B) in ((1050, 1050, 4), (1050, 1050, 16), (100, 1050, 4), (100, 100, 4)):
    dev = "cuda"
    q, k, v, do = (torch.randn(l, B, E, device=dev).to(torch.bfloat16) for l in (Lq, Lk, Lk, Lq))
    o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v); dws = torch.empty(B, H, Lq, device=dev)
    sp = L.stream_ptr(); lib = L.lib(); sc = 1.0 / 32 ** 0.5
    fwd = lambda: L.check(lib.mi_mha_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, o.data_ptr(), lse.data_ptr(), B, H, Lq, Lk, E, sc, sp), "fwd")
    bwd = lambda: L.check(lib.mi_mha_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, o.data_ptr(), lse.data_ptr(), do.data_ptr(), dws.data_ptr(),
                                         dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, Lq, Lk, E, sc, sp), "bwd")
    res = []
    for fn in (fwd, bwd):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 4.0 * Lq * Lk * 32 * B * H
    print(f"Lq {Lq} Lk {Lk} B {B}: fwd {res[0]:.1f} us ({fl/res[0]/1e6:.1f} TF)  bwd {res[1]:.1f} us ({2.5*fl/res[1]/1e6:.1f} TF)", flush=True)
