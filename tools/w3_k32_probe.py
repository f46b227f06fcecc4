#!/usr/bin/env python
"""GPU box: the K = 32 weight-stationary 3x3 launches of the YOLOX-s step alone (stem 320^2 stride 1, dark2.0 stride 2,
dark2 bottleneck 160^2) - HIP-event time per launch and achieved bytes/s; under rocprofv3 --pmc the same calls give the
L1 / L2 request counters (tools/w3_k32_pmc.sh)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from yolov7_d2_amd import _lib as L
from test_gpu_conv3x3_ws import _pack, _desc, _desc_s2, sp
DEV = "cuda"
iters = int(os.environ.get("ITERS", 20))
g = torch.Generator().manual_seed(0)
def timeit(call):
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (K, Co, N, H, W, S) in ((32, 32, 16, 320, 320, 1), (32, 64, 16, 320, 320, 2), (32, 32, 16, 160, 160, 1), (64, 128, 16, 160, 160, 2),
                            (64, 64, 16, 80, 80, 1)):
    x = torch.randn(N, H, W, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(Co, K, 3, 3, generator=g) / (3 * K ** .5)).to(DEV)
    Ho, Wo = (H - 1) // S + 1, (W - 1) // S + 1
    y = torch.empty(N, Ho, Wo, Co, dtype=torch.bfloat16, device=DEV)
    st = torch.zeros(16, Co, 2, dtype=torch.float64, device=DEV)
    img = _pack(w, False)
    d = _desc(x, K, 0, N, H, W, K, img, y, K, 0, False, st) if S == 1 else _desc_s2(x, N, H, W, K, img, y, Co, st)
    us = timeit(lambda: L.check(L.lib().mi_conv3x3_ws(C.byref(d), 1, sp()), "ws"))
    mb = (x.numel() + y.numel()) * 2 / 1e6
    print(f"K{K}->{Co} {N}x{H}x{W} s{S}: {us:7.1f} us  {mb:6.1f} MB  {mb / us * 1e-3 * 1e3:6.2f} GB/ms = {mb / us:5.2f} TB/s", flush=True)
