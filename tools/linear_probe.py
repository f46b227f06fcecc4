import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov7_d2_amd.modeling.transformer import _LinearFn
torch.manual_seed(0)
for (T, Cin, Cout) in ((2100, 2048, 256), (2048, 2048, 256), (2100, 256, 2048), (2100, 512, 2048), (336, 2048, 256)):
    x = torch.randn(T, Cin, device="cuda").to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(Cout, Cin, device="cuda") / Cin ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, device="cuda").requires_grad_(True)
    y = _LinearFn.apply(x, w, b)
    dy = torch.randn(T, Cout, device="cuda").to(torch.bfloat16)
    y.backward(dy)
    wb = w.detach().to(torch.bfloat16).float()
    yr = x.detach().float() @ wb.t() + b.detach()
    dxr = dy.float() @ wb
    gwr = dy.float().t() @ x.detach().float()
    rel = lambda a, r: float((a.float() - r).norm() / r.norm())
    print(T, Cin, Cout, "y", rel(y, yr), "dx", rel(x.grad, dxr), "gw", rel(w.grad, gwr), "gb", rel(b.grad, dy.float().sum(0)), "dx norm ratio", float(x.grad.float().norm() / dxr.norm()))
