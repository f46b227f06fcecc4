#!/usr/bin/env python
"""throughput of the DETR attention core (fwd, bwd) at encoder / decoder shapes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov7_d2_amd.modeling import mha_core
for (Lq, Lk, B) in ((1050, 1050, 4), (1050, 1050, 16), (100, 1050, 4), (100, 100, 4)):
    H, E = 8, 256
    q = torch.randn(Lq, B, E, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(Lk, B, E, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(Lk, B, E, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    go = torch.randn(Lq, B, E, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        o = mha_core(q, k, v, None, H); o.backward(go)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    n = 20
    tf = tb = 0.0
    for _ in range(n):
        e0.record(); o = mha_core(q, k, v, None, H); e1.record(); o.backward(go); e2.record()
        torch.cuda.synchronize()
        tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
    fl = 4.0 * Lq * Lk * 32 * B * H
    print(f"Lq {Lq} Lk {Lk} B {B}: fwd {tf/n*1e3:.1f} us ({fl/(tf/n*1e-3)/1e12:.1f} TF)  bwd {tb/n*1e3:.1f} us ({2.5*fl/(tb/n*1e-3)/1e12:.1f} TF)")
