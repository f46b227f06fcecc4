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
    "lr".

    `clip_norm`: FullModelGradientClippingOptimizer (yolov7/optimizer/build.py:206-223; the DETR configs set
    SOLVER.CLIP_GRADIENTS full_model 0.01 / 0.1): the global gradient norm is taken on the device over the same tables
    (mi_grad_norm_multi) and the update kernel multiplies by the coefficient it left there - no host value, capturable.

    Captured steps: every capture gets ITS OWN device table (`begin_capture` / `finish_capture`): the gradient tensors of a
    captured graph live in that graph's private pool, so two captured batch shapes have two sets of gradient addresses and
    a shared table would make the replays of the first graph read the second graph's (stale) gradients.  A learning-rate
    change (`param_groups[i]["lr"] = ...`, an LR scheduler) is carried into every table by `sync_lr()`, which
    GraphedTrainStep calls before each replay: the captured launch reads lr from its table at run time."""

    CHUNK = 16384

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, clip_norm=None):
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
        dev = self.device = self.params[0].device
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
        self._chunks_host, self._sub_chunks = chunks, {}
        self.chunks = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.table_bytes = len(self.params) * C.sizeof(L.mi_adamw_tensor)
        self.table = torch.zeros(self.table_bytes, dtype=torch.uint8, device=dev)   # the eager steps' table
        self._grad_ptrs = None
        self._eager_lrs = None
        self._cap_table, self._cap_used = None, False
        self.captures = []          # [device table, host ctypes array, lrs it was uploaded with, -]
        self.clip_norm = None if clip_norm is None or clip_norm <= 0 else float(clip_norm)
        if self.clip_norm is not None:
            self.norm_partial = torch.zeros(self.nchunks, dtype=torch.float64, device=dev)
            self.clip_out = torch.ones(2, dtype=torch.float32, device=dev)      # [coefficient, norm] of the last step

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _lrs(self):
        return [float(g["lr"]) for g in self.param_groups]

    def _host_table(self, partial=False):
        """partial: a table for a launch that reads only some tensors' gradients (one backward stage's gather): the
        others may not exist yet, their g stays null"""
        arr = (L.mi_adamw_tensor * len(self.params))()
        k = 0
        for g in self.param_groups:
            for p in g["params"]:
                a = arr[k]
                assert (p.grad is not None or partial) and p.is_contiguous(), "MultiTensorAdamW: missing gradient"
                assert p.grad is None or p.grad.is_contiguous(), "MultiTensorAdamW: strided gradient"
                a.p, a.m, a.v = p.data_ptr(), self.exp_avg[k].data_ptr(), self.exp_avg_sq[k].data_ptr()
                a.g = p.grad.data_ptr() if p.grad is not None else 0
                a.count, a.lr, a.weight_decay = p.numel(), float(g["lr"]), float(g["weight_decay"])
                k += 1
        return arr

    @staticmethod
    def _upload(table, arr):
        table.copy_(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8))

    def refresh(self, partial=False):
        """upload the eager table (new gradient addresses, changed learning rates); outside a capture only"""
        self._upload(self.table, self._host_table(partial))
        self._grad_ptrs = [p.grad.data_ptr() if p.grad is not None else 0 for p in self.params]
        self._eager_lrs = self._lrs()

    def begin_capture(self):
        """a fresh device table for the capture that follows (allocated outside the graph's pool)"""
        self._cap_table = torch.zeros(self.table_bytes, dtype=torch.uint8, device=self.device)
        self._cap_used = self._cap_partial = False

    def finish_capture(self):
        """the capture has ended: the gradient addresses of the graph's pool are final -> fill THIS capture's table"""
        if self._cap_table is None or not self._cap_used:      # (a captured segment that read no gradient table: the
            self._cap_table = None                              #  data-parallel update reads the flat buffer's table)
            return None
        arr = self._host_table(partial=self._cap_partial)
        self._upload(self._cap_table, arr)
        # (no reference to the gradient tensors is kept: they belong to the graph's memory pool, which GraphedTrainStep
        #  shares between captures - a later capture may lay its own tensors over them, replays never overlap)
        ent = [self._cap_table, arr, self._lrs(), None]
        self.captures.append(ent)
        self._cap_table = None
        return ent            # the capture's handle: release_capture(handle) when its graph is dropped

    def release_capture(self, handle):
        """the graph that reads this table is gone (GraphedTrainStep's LRU eviction / close()): drop the device table and its
        host mirror, so that a long multi-scale run neither grows by one table per re-capture nor re-uploads dead ones"""
        if handle is None:
            return
        self.captures[:] = [e for e in self.captures if e is not handle]

    def sync_lr(self, only=None):
        """carry changed learning rates into the captured tables (cheap no-op when nothing changed).  only: the handles
        of the graph about to be replayed (+ the flat buffer's table, always) instead of every live table"""
        lrs = self._lrs()
        ents = self.captures if only is None else [e for e in self.captures if any(e is h for h in only) or e[3] == "flat"]
        for ent in ents:
            if ent[2] != lrs:
                k = 0
                for g in self.param_groups:
                    for _ in g["params"]:
                        ent[1][k].lr = float(g["lr"])
                        k += 1
                self._upload(ent[0], ent[1])
                ent[2] = lrs

    def _grad_table(self, partial=False):
        """the table whose g fields are the parameters' CURRENT .grad tensors: this capture's, or the eager one
        (partial: the caller reads only tensors that have a gradient - gather_grads(only=...))"""
        if torch.cuda.is_current_stream_capturing():
            if self._cap_table is None:      # (a caller without begin_capture: allocated from the graph's pool, kept alive here)
                self._cap_table = torch.zeros(self.table_bytes, dtype=torch.uint8, device=self.device)
            self._cap_used = True
            return self._cap_table
        ptrs = [p.grad.data_ptr() if p.grad is not None else 0 for p in self.params]
        if ptrs != self._grad_ptrs or self._eager_lrs != self._lrs():
            self.refresh(partial)
        return self.table

    # ---- data parallel: gradients gathered into one flat buffer, all-reduced in a few large buckets, updated from there
    def enable_flat_grads(self, bucket_bytes=64 << 20):
        """allocate the flat gradient buffer (tensor k at element offset flat_off[k], 64-element aligned), a table whose g
        fields point INTO it (addresses that no captured graph's pool can change) and the all-reduce buckets
        [(lo, hi)] - contiguous element ranges of ~bucket_bytes cut at tensor boundaries: a few large messages, the shape
        RCCL's ring over xGMI wants (not the 25 MiB NCCL-on-NVSwitch habit)"""
        if getattr(self, "flat", None) is not None:
            return
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.flat_off = torch.tensor(offs, dtype=torch.int64, device=self.device)
        arr = (L.mi_adamw_tensor * len(self.params))()
        k = 0
        for g in self.param_groups:
            for p in g["params"]:
                a = arr[k]
                a.p, a.g = p.data_ptr(), self.flat.data_ptr() + 4 * offs[k]
                a.m, a.v = self.exp_avg[k].data_ptr(), self.exp_avg_sq[k].data_ptr()
                a.count, a.lr, a.weight_decay = p.numel(), float(g["lr"]), float(g["weight_decay"])
                k += 1
        self.flat_table = torch.zeros(self.table_bytes, dtype=torch.uint8, device=self.device)
        self._upload(self.flat_table, arr)
        self.captures.append([self.flat_table, arr, self._lrs(), "flat"])    # (sync_lr keeps its lr fields current too)
        per = max(1, int(bucket_bytes) // 4)
        self.buckets, lo = [], 0
        for k in range(len(offs)):
            end = offs[k + 1] if k + 1 < len(offs) else n
            if end - lo >= per or k + 1 == len(offs):
                self.buckets.append((lo, end))
                lo = end

    def gather_grads(self, only=None):
        """.grad of every parameter -> the flat buffer (one launch; capturable: reads this capture's own table).
        only: parameter indices - the tensors of one backward stage (GraphedTrainStep cuts the backward into stages whose
        slices of the flat buffer are all-reduced while the later stages still compute); the launch then walks just their
        chunks, and the captured table may hold null gradients for the rest"""
        chunks, n = self.chunks, self.nchunks
        if only is not None:
            key = tuple(only)
            if key not in self._sub_chunks:
                want = set(key)
                sub = [c for c in self._chunks_host if c[0] in want]
                arr = (L.mi_adamw_chunk * max(1, len(sub)))()
                for a, (ti, c, k) in zip(arr, sub):
                    a.tensor, a.count, a.offset = ti, c, k
                self._sub_chunks[key] = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(sub))
            chunks, n = self._sub_chunks[key]
            if torch.cuda.is_current_stream_capturing():
                self._cap_partial = True
            if n == 0:
                return
        L.check(L.lib().mi_grad_gather_multi(self._grad_table(partial=only is not None).data_ptr(), chunks.data_ptr(), n,
                                             self.flat_off.data_ptr(), self.flat.data_ptr(), L.stream_ptr()),
                "mi_grad_gather_multi")

    def step(self, grad_scale=1.0, from_flat=False):
        """from_flat: the gradients are the flat buffer's (gather_grads + all-reduce happened); grad_scale = 1 / world there"""
        if from_flat:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_lr()
            table = self.flat_table
        else:
            table = self._grad_table()
        self.step_count += 1
        coef = None
        if self.clip_norm is not None:
            L.check(L.lib().mi_grad_norm_multi(table.data_ptr(), self.chunks.data_ptr(), self.nchunks,
                                               self.norm_partial.data_ptr(), self.clip_norm, float(grad_scale),
                                               self.clip_out.data_ptr(), L.stream_ptr()), "mi_grad_norm_multi")
            coef = self.clip_out.data_ptr()
        L.check(L.lib().mi_adamw_step_multi_clip(table.data_ptr(), self.chunks.data_ptr(), self.nchunks, self.betas[0],
                                                 self.betas[1], self.eps, self.step_count.data_ptr(), float(grad_scale),
                                                 coef, L.stream_ptr()), "mi_adamw_step_multi_clip")

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
