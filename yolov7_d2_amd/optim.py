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


class MultiTensorAdamW:
    """torch.optim.AdamW over an eager module tree's separately allocated parameters as ONE launch per step
    (mi_adamw_step_multi), usable inside a captured hipGraph: the tables of pointers and the update count live on the
    device and are read at run time.  Same constructor shape as torch.optim.AdamW (params or param groups with their own
    lr / weight_decay), same update rule (decoupled weight decay, bias-corrected moments), `param_groups` with a mutable
    "lr" (re-uploaded by refresh()).  Inside a capture the gradient tensors are the graph pool's: step() records the
    launch and remembers them, finish_capture() uploads the table once the capture has ended."""

    CHUNK = 16384

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = list(params)
        if params and not isinstance(params[0], dict):
            params = [dict(params=params)]
        self.param_groups = []
        for g in params:
            g = dict(g)
            g["params"] = [p for p in g["params"]]
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            self.param_groups.append(g)
        self.betas, self.eps = betas, eps
        self.params = [p for g in self.param_groups for p in g["params"]]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params)
        dev = self.params[0].device
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        chunks = []
        for ti, p in enumerate(self.params):
            n, k = p.numel(), 0
            while k < n:
                c = min(self.CHUNK, n - k)
                chunks.append((ti, c, k))
                k += c
        arr = (L.mi_adamw_chunk * len(chunks))()
        for a, (ti, c, k) in zip(arr, chunks):
            a.tensor, a.count, a.offset = ti, c, k
        self.nchunks = len(chunks)
        self.chunks = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.table = torch.zeros(len(self.params) * C.sizeof(L.mi_adamw_tensor), dtype=torch.uint8, device=dev)
        self._grad_ptrs = None

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _host_table(self):
        arr = (L.mi_adamw_tensor * len(self.params))()
        k = 0
        for g in self.param_groups:
            for p in g["params"]:
                a = arr[k]
                assert p.grad is not None and p.grad.is_contiguous() and p.is_contiguous(), "MultiTensorAdamW: missing / strided gradient"
                a.p, a.g, a.m, a.v = p.data_ptr(), p.grad.data_ptr(), self.exp_avg[k].data_ptr(), self.exp_avg_sq[k].data_ptr()
                a.count, a.lr, a.weight_decay = p.numel(), float(g["lr"]), float(g["weight_decay"])
                k += 1
        return arr

    def refresh(self):
        """upload the table (new gradient addresses, changed learning rates); outside a capture only"""
        arr = self._host_table()
        self.table.copy_(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8))
        self._grad_ptrs = [p.grad.data_ptr() for p in self.params]

    def finish_capture(self):
        self.refresh()

    def step(self, grad_scale=1.0):
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            ptrs = [p.grad.data_ptr() if p.grad is not None else 0 for p in self.params]
            if ptrs != self._grad_ptrs:
                self.refresh()
        else:
            self._pending = [p.grad for p in self.params]       # (keeps the graph pool's gradient tensors referenced)
        self.step_count += 1
        L.check(L.lib().mi_adamw_step_multi(self.table.data_ptr(), self.chunks.data_ptr(), self.nchunks, self.betas[0],
                                            self.betas[1], self.eps, self.step_count.data_ptr(), float(grad_scale),
                                            L.stream_ptr()), "mi_adamw_step_multi")

    # ---- what GraphedTrainStep snapshots around its warm-up steps
    def state_tensors(self):
        return self.exp_avg + self.exp_avg_sq + [self.step_count]


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
