#!/usr/bin/env python
"""one shape of the weight-stationary 3x3 kernel a few times (for rocprofv3 --pmc passes)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from yolov7_d2_amd import _lib as L
from test_gpu_conv3x3_ws import _pack, _desc, sp
K = int(os.environ.get("K", 128)); N = int(os.environ.get("N", 64)); HW = int(os.environ.get("HW", 80))
g = torch.Generator().manual_seed(0)
x = torch.randn(N, HW, HW, K, generator=g).to("cuda", torch.bfloat16)
w = (torch.randn(K, K, 3, 3, generator=g) / (3 * K ** .5)).to("cuda")
img = _pack(w, False); y = torch.empty(N, HW, HW, K, dtype=torch.bfloat16, device="cuda")
d = _desc(x, K, 0, N, HW, HW, K, img, y, K, 0, False)
for _ in range(5):
    L.check(L.lib().mi_conv3x3_ws(C.byref(d), 1, sp()), "ws")
torch.cuda.synchronize()
