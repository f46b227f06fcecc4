"""Optimizer-side entry points over flat fp32 tensors: fused AdamW and full-model gradient clipping.

`FlatAdamW` mirrors torch.optim.AdamW (the optimizer the reference's DETR / SparseInst trainers build,
train_transformer.py / optimizer/build.py) for a parameter arena: one launch per step, per-segment lr / weight decay.
`clip_grad_norm_flat_` mirrors FullModelGradientClippingOptimizer (optimizer/build.py:206-223 = clip_grad_norm_ over all
parameters) with no host synchronisation.
"""
import ctypes as C

import torch

from . import _lib as L


class FlatAdamW:
    def __init__(self, params, grads, segments, betas=(0.9, 0.999), eps=1e-8):
        """params / grads: flat fp32 device tensors of equal size; segments: [(offset, count, lr, weight_decay)]"""
        assert params.is_cuda and params.dtype == torch.float32 and params.numel() == grads.numel()
        self.params, self.grads = params, grads
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(params), torch.zeros_like(params)
        self.betas, self.eps, self.steps = betas, eps, 0
        self.set_segments(segments)

    def set_segments(self, segments, chunk=16384):
        # one 256-thread block per table entry: natural per-lr-group segments (2-3 for DETR, ~41 M parameters) would run
        # the whole update on 2-3 CUs -> cut them into 16 k-element chunks like ParamArena.build_sgd_segments does
        cut = []
        for (off, cnt, lr, wd) in segments:
            k = 0
            while k < cnt:
                c = min(chunk, cnt - k)
                cut.append((off + k, c, lr, wd))
                k += c
        segments = cut
        segs = (L.mi_sgd_seg * len(segments))()
        for s, (off, cnt, lr, wd) in zip(segs, segments):
            s.offset, s.count, s.lr, s.weight_decay = int(off), int(cnt), float(lr), float(wd)
        self.nseg = len(segments)
        self.segs = torch.frombuffer(bytearray(bytes(segs)), dtype=torch.uint8).to(self.params.device)

    def step(self, grad_scale=1.0):
        self.steps += 1
        L.check(L.lib().mi_adamw_step(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), self.segs.data_ptr(), self.nseg, self.betas[0],
                                      self.betas[1], self.eps, self.steps, float(grad_scale), L.stream_ptr()),
                "mi_adamw_step")


def clip_grad_norm_flat_(grads, max_norm, ws=None):
    """grads (flat fp32 device tensor) *= min(1, max_norm / (||grads||_2 + 1e-6)); returns the norm as a 0-dim DEVICE
    tensor (reading it synchronises; the clipping itself does not)"""
    assert grads.is_cuda and grads.dtype == torch.float32
    if ws is None:
        ws = torch.empty(1024, dtype=torch.float64, device=grads.device)
    norm = torch.empty((), dtype=torch.float32, device=grads.device)
    L.check(L.lib().mi_grad_clip_full_model(grads.data_ptr(), grads.numel(), float(max_norm), ws.data_ptr(),
                                            norm.data_ptr(), L.stream_ptr()), "mi_grad_clip_full_model")
    return norm
