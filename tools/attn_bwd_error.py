#!/usr/bin/env python
"""local: error analysis of the attention backward on operands dumped from the device (MI_MHA_DUMP, modeling/attention.py).
For each dump: fp64 autograd on the SAME bf16 operands = truth; the device's dq / dk / dv against it; and CPU emulations of
the kernel arithmetic with delta = rowsum(dO o O) taken from the bf16 O, from an fp32 O, and from the backward's own P."""
import glob, math, sys
import torch

bf = lambda x: x.to(torch.bfloat16).to(torch.float32)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))
cos = lambda a, b: float(torch.dot(a.double().flatten(), b.double().flatten()) / (a.double().norm() * b.double().norm() + 1e-300))

for f in sorted(glob.glob(sys.argv[1] + "/mha_bwd_*.pt")):
    d = torch.load(f)
    q, k, v, do = (d[n].float() for n in ("q", "k", "v", "do"))
    H, scale = d["heads"], d["scale"]
    Lq, B, E = q.shape
    Lk = k.shape[0]
    hd = E // H
    sp = lambda t: t.view(t.shape[0], B, H, hd).permute(1, 2, 0, 3)          # [B, H, L, hd]
    Q, K, V, DO = sp(q), sp(k), sp(v), sp(do)
    mask = d["mask"]
    bias = torch.zeros(B, 1, 1, Lk)
    if mask is not None:
        bias = bias.masked_fill(mask.bool()[:, None, None, :], float("-inf"))
    Q64, K64, V64 = (t.double().requires_grad_(True) for t in (Q, K, V))
    P = torch.softmax(Q64 @ K64.transpose(-1, -2) * scale + bias.double(), -1)
    ((P @ V64) * DO.double()).sum().backward()
    tq, tk, tv = Q64.grad, K64.grad, V64.grad
    print(f"{f.split('/')[-1]}: Lq {Lq} Lk {Lk} B {B} drop {d['drop']}  P max {float(P.max()):.4f} (uniform {1 / Lk:.4f})")
    print("   device      dq rel %.4f cos %.5f | dk rel %.4f cos %.5f | dv rel %.4f" % (rel(sp(d["dq"].float()), tq), cos(sp(d["dq"].float()), tq),
          rel(sp(d["dk"].float()), tk), cos(sp(d["dk"].float()), tk), rel(sp(d["dv"].float()), tv)))
    # kernel arithmetic
    s = Q @ K.transpose(-1, -2) * scale + bias
    m = s.max(-1, keepdim=True).values
    Pt = bf(torch.exp(s - m)); l = Pt.sum(-1, keepdim=True)
    O32 = (Pt @ V) / l; O16 = bf(O32)
    pe = torch.exp(s - (m + torch.log(l)))
    dP = DO @ V.transpose(-1, -2)
    for name in ("bf16_O", "fp32_O", "own_P"):
        if name == "own_P":
            a, n = bf(pe * dP * scale), bf(pe)
            delta = a.sum(-1, keepdim=True) / n.sum(-1, keepdim=True)
            dq = a @ K - delta * (n @ K)
            dS = bf(pe * (dP * scale - delta))
        else:
            delta = (DO * (O16 if name == "bf16_O" else O32)).sum(-1, keepdim=True)
            dS = bf(pe * (dP - delta) * scale)
            dq = dS @ K
        dk = dS.transpose(-1, -2) @ Q
        print("   %-10s  dq rel %.4f cos %.5f | dk rel %.4f cos %.5f   (bf16 outputs: dq %.4f dk %.4f)" % (name, rel(dq, tq), cos(dq, tq), rel(dk, tk), cos(dk, tk),
              rel(bf(dq), tq), rel(bf(dk), tk)))
    print("   |dq| %.3e |dk| %.3e |dv| %.3e ; |delta| %.3e |dP - delta| rms %.3e" % (float(tq.norm()), float(tk.norm()), float(tv.norm()),
          float(delta.norm()), float(((dP - (DO * O32).sum(-1, keepdim=True)) ** 2).mean().sqrt())))
