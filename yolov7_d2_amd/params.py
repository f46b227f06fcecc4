"""Flat fp32 arenas for parameters / gradients / momentum.

The nn.Parameters of the module tree keep their names, shapes and OIHW layout (so state_dict keys,
DetectionCheckpointer and external optimizers see the reference's parameters) but
their storage is re-pointed into one contiguous buffer, so that
  * every weight gradient is written by the wgrad kernels straight into one flat gradient buffer
    (a single RCCL all-reduce range per bucket, no flatten/unflatten copies),
  * the fused SGD update is one launch over the arena.
"""
import torch

from . import _lib as L


class ParamArena:
    def __init__(self, module: torch.nn.Module, device):
        self.device = torch.device(device)
        self.entries = []  # (name, param, offset, numel)
        off = 0
        named = list(module.named_parameters())
        # a module may ask for pairs of its parameters to lie back to back (`arena_adjacent()` -> [(first, second)], names
        # relative to the module): the YOLOX head's reg / obj prediction convs then read as ONE [5, C] weight / [5] bias (and
        # gradient) view and run as one convolution (yolox_net.YOLOXHead.emit).  Names, shapes and state_dict keys are untouched.
        order = [n for n, _ in named]
        for mname, m in module.named_modules():
            fn = getattr(m, "arena_adjacent", None)
            if fn is None:
                continue
            pre = mname + "." if mname else ""
            for a, b in fn():
                a, b = pre + a, pre + b
                if a in order and b in order:
                    order.remove(b)
                    order.insert(order.index(a) + 1, b)
        byname = dict(named)
        for name in order:
            p = byname[name]
            n = p.numel()
            self.entries.append((name, p, off, n))
            off += (n + 3) // 4 * 4
        self.total = off
        self.data = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.mom = torch.zeros(off, dtype=torch.float32, device=self.device)
        self._grad_views = {}
        with torch.no_grad():
            for name, p, o, n in self.entries:
                v = self.data[o:o + n].view(p.shape)
                v.copy_(p.data.to(self.device, torch.float32))
                p.data = v
                self._grad_views[id(p)] = self.grad[o:o + n].view(p.shape)
        self._segs = None

    def grad_of(self, p):
        return self._grad_views[id(p)]

    def bind_grads(self):
        """expose the flat gradient buffer as .grad of each parameter (no copy)"""
        for name, p, o, n in self.entries:
            p.grad = self._grad_views[id(p)]

    # ---- fused SGD segment table (detectron2 build_optimizer semantics: WEIGHT_DECAY_NORM for norm params)
    def build_sgd_segments(self, lr, weight_decay, weight_decay_norm, norm_param_ids, chunk=16384):
        segs = []
        for name, p, o, n in self.entries:
            wd = weight_decay_norm if id(p) in norm_param_ids else weight_decay
            k = 0
            while k < n:
                c = min(chunk, n - k)
                segs.append((o + k, c, wd, lr))
                k += c
        arr = (L.mi_sgd_seg * len(segs))()
        for i, (o, c, wd, l_) in enumerate(segs):
            arr[i].offset, arr[i].count, arr[i].weight_decay, arr[i].lr = o, c, wd, l_
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        self._segs_host = arr
        self._segs = host.to(self.device)
        self._nseg = len(segs)
        return self._segs, self._nseg

    def set_lr(self, lr):
        """rewrite the lr column of the device segment table (graph-replay safe: same pointer)"""
        arr = self._segs_host
        for i in range(self._nseg):
            arr[i].lr = lr
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self._segs.copy_(host, non_blocking=False)
