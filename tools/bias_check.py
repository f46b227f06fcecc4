import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov7_d2_amd import _lib as L
B, A, nch = 4, 336, 85
torch.manual_seed(0)
dp = torch.randn(B, A, nch, device="cuda")
jobs = (L.mi_bias_job * 9)()
outs = []
a0 = 0; k = 0
for HW in (256, 64, 16):
    for (c0, nc) in ((5, 80), (0, 4), (4, 1)):
        o = torch.zeros(nc, device="cuda"); outs.append((o, a0, HW, c0, nc))
        jobs[k].out, jobs[k].a0, jobs[k].HW, jobs[k].c0, jobs[k].nc = o.data_ptr(), a0, HW, c0, nc; k += 1
    a0 += HW
ws = torch.zeros(16 * 512 * 128, device="cuda")
L.check(L.lib().mi_yolox_bias_grads(dp.data_ptr(), B, A, nch, jobs, 9, ws.data_ptr(), L.stream_ptr()), "bias")
torch.cuda.synchronize()
for (o, a0, HW, c0, nc) in outs:
    ref = dp[:, a0:a0 + HW, c0:c0 + nc].sum((0, 1))
    print(a0, HW, c0, nc, float((o - ref).abs().max()), float(ref.abs().max()))
