"""ORACLE support: seeded input generators shared by gen_golden.py (reference side, needs /root/reference) and the
tests (which must not import gen_golden.py's reference loader on the GPU box)."""
import torch


def synth_box_pairs(n, seed):
    g = torch.Generator().manual_seed(seed)
    c = 50 + 400 * torch.rand(n, 2, generator=g)
    wh = 8 + 120 * torch.rand(n, 2, generator=g)
    pred = torch.cat([c, wh], 1)
    tgt = torch.cat([c + 30 * (torch.rand(n, 2, generator=g) - 0.5), wh * (0.6 + 0.8 * torch.rand(n, 2, generator=g))], 1)
    tgt[: n // 8] = torch.cat([c[: n // 8] + 300, wh[: n // 8]], 1)     # some disjoint pairs
    return pred, tgt
