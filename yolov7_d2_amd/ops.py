"""`torch.ops.mi355.*` — the per-operator boundary ("PyTorch-ROCm custom ops ... TORCH_LIBRARY(mi355) +
register_autograd so that loss.backward() works unchanged").

The whole-step plan (plan.py) is the fast path; these ops expose the SAME kernels one operator at a time, with autograd,
for a model that keeps the reference's own nn.Module tree and swaps single operators:

    torch.ops.mi355.conv2d(x, weight, bias, stride, padding)                -> nn.Conv2d (1x1 / 3x3, stride 1 / 2)
    torch.ops.mi355.conv_bn_silu(x, weight, gamma, beta, running_mean, running_var, stride, eps, training)
                                                                            -> BaseConv.forward (wrappers.py:60-83)
    torch.ops.mi355.batched_nms(boxes, scores, idxs, iou_threshold)         -> torchvision.ops.batched_nms (boxes.py:199)
    torch.ops.mi355.yolox_loss(raw, labels, anchors, num_classes)           -> YOLOXHead.get_losses (yolox_head.py:274-441)
    torch.ops.mi355.mha(q, k, v, key_padding_mask, num_heads)               -> attention core of nn.MultiheadAttention
    torch.ops.mi355.iou_loss_v6(pred, target, iou_type, xyxy, eps)          -> IOUlossV6 (boxes.py:666-752)

`patch_base_convs(module)` re-points the forward of every BaseConv-shaped sub-module (children `conv`: bias-free
nn.Conv2d, `bn`: nn.BatchNorm2d, `act`: nn.SiLU - the reference's own class qualifies unmodified) to
torch.ops.mi355.conv_bn_silu on its own parameters.

Tensors: activations NCHW bf16 in channels_last memory (= the kernels' NHWC; anything else is converted on entry),
weights / BatchNorm parameters fp32 as nn.Conv2d / nn.BatchNorm2d hold them.  HIP device tensors only: there is no CPU
implementation registered (a CPU call fails with PyTorch's "no kernel for backend" error, by design).
"""
import ctypes as C
import math

import torch

from . import _lib as L

_LIBDEF = torch.library.Library("mi355", "DEF")


def _rup(a, b):
    return (a + b - 1) // b * b


# ------------------------------------------------------------------------------------------------ weight images
class WeightImages:
    """The packed bf16 images (forward + data gradient) of every PARAMETER-backed weight of a training step, written by ONE
    launch (mi_pack_conv_weights_batch, the YOLOX plan's PACK_W_BATCH) at the start of the step instead of one pack launch
    per layer call: 143 of the ~1900 launches of a captured DETR-R50 step, 131 of SparseInst-R50's ~1330
    (graph_step.GraphedTrainStep owns one; `active` is set only inside its step body, right after run()).

    A weight qualifies when its fp32 storage lies inside one of the model's parameters (a row block of in_proj_weight
    does; `weight * scale`, a space-to-depth rearrangement or an activation used as a weight do not: those are packed by
    their own launch as before).  The first step RECORDS: every qualifying pack call allocates persistent images, packs
    them with its own launch and registers the job; freeze() uploads the job table; from then on run() re-packs all
    images from the current parameter values and the pack calls return them without a launch.  A call that was not
    recorded (a layer that did not run in the first step, or a FrozenBatchNorm factor that was recomputed since - its key
    carries the factor's address, and the recorded tensor is kept alive so that the address cannot be reused) is served by
    its own launch: correct, one launch slower."""
    active = None

    def __init__(self, params):
        import bisect
        self._bisect = bisect.bisect_right
        rs = sorted((p.data_ptr(), p.data_ptr() + p.numel() * 4) for p in params if p.dtype == torch.float32 and p.is_contiguous())
        self._lo, self._hi = [r[0] for r in rs], [r[1] for r in rs]
        # the job table (and every captured replay of it) reads these ADDRESSES: verify() notices a parameter whose storage
        # was replaced afterwards (model.to(), load_state_dict(assign=True), an EMA swap through .data)
        self._params = [(p, p.data_ptr()) for p in params if p.dtype == torch.float32 and p.is_contiguous()]
        self.images = {}        # key -> (wf, wd)
        self._jobs, self._hold = [], []
        self.table = None       # device job table once frozen
        self.launch = None

    def _stable(self, ptr, nbytes):
        i = self._bisect(self._lo, ptr) - 1
        return i >= 0 and ptr + nbytes <= self._hi[i]

    def get(self, w32, Cout, Cin, KK, CinP, CoutP, CoutPK, CinPN, fwd, dgrad, scale, pack_now):
        """(wf, wd) or None when the weight does not qualify / was not recorded; pack_now(wf, wd): the single-layer launch"""
        if not self._stable(w32.data_ptr(), Cout * Cin * KK * 4):
            return None
        key = (w32.data_ptr(), Cout, Cin, KK, CinP, CoutP, CoutPK, CinPN, bool(fwd), bool(dgrad), 0 if scale is None else scale.data_ptr())
        hit = self.images.get(key)
        if hit is not None:
            if self.table is None:
                pack_now(*hit)              # (still recording: a second use of the layer in the first step)
            return hit
        if self.table is not None:
            return None
        dev = w32.device
        wf = torch.empty(KK * CinP * CoutP, dtype=torch.bfloat16, device=dev) if fwd else None
        wd = torch.empty(KK * CoutPK * CinPN, dtype=torch.bfloat16, device=dev) if dgrad else None
        pack_now(wf, wd)
        j = L.mi_pack_job()
        j.w, j.wf, j.wd = w32.data_ptr(), L.ptr(wf), L.ptr(wd)
        j.Cout, j.Cin, j.KK, j.CinPad, j.CoutPad, j.CoutPadK, j.CinPadN = Cout, Cin, KK, CinP, CoutP, CoutPK, CinPN
        j.scale = L.ptr(scale)
        self._jobs.append(j)
        self._hold.append((w32, scale))     # (their addresses are in the table: never freed, never reused)
        self.images[key] = (wf, wd)
        return wf, wd

    def freeze(self):
        if self.table is not None or not self._jobs:
            return
        n = len(self._jobs)
        jobs = (L.mi_pack_job * n)(*self._jobs)
        nblk = L.check(L.lib().mi_pack_jobs_layout(jobs, n), "mi_pack_jobs_layout")
        self.table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(self._hold[0][0].device)
        self.launch = (n, nblk, max(j.KK for j in self._jobs))

    def verify(self):
        """raise if a parameter's storage moved since construction: the recorded jobs (and the graphs that replay them)
        would keep packing the OLD storage - training on stale weights without any error"""
        moved = [i for i, (p, ptr) in enumerate(self._params) if p.data_ptr() != ptr]
        if moved:
            raise L.MI355Error(f"WeightImages: the storage of {len(moved)} parameter(s) was replaced after the step was "
                               "recorded (model.to() / load_state_dict(assign=True) / .data swap): the captured weight "
                               "re-pack reads the old addresses - build a new GraphedTrainStep")

    def run(self):
        """all recorded images from the current parameter values, one launch (a no-op while recording)"""
        if self.table is not None:
            n, nblk, kk = self.launch
            L.check(L.lib().mi_pack_conv_weights_batch(self.table.data_ptr(), n, nblk, kk, L.stream_ptr()), "mi_pack_conv_weights_batch")


# ------------------------------------------------------------------------------------------------ small host values
class HostRing:
    """Small host values on their way to the device (the `prepare_batch` halves of the DETR / SparseInst steps: image sizes,
    packed labels / boxes, prefix offsets, 1 / num_boxes) WITHOUT blocking the host.  `t.copy_(torch.tensor(...))` from
    pageable memory blocks until the stream has drained - i.e. until the PREVIOUS step's graph has finished - so the rest of
    the host half and the graph launch ran with the device idle: 0.6 - 0.7 ms of every 12.5 ms step
    (tools/host_step_probe.py).  Here the value is written into a slot of one page-locked ring and copied from there
    asynchronously on the current stream; the host runs a step ahead, the device runs the host half's launches back to
    back.  The ring is reused in order; when it wraps the stream is synchronised once (4 MB: every few thousand steps)."""
    BYTES = 4 << 20
    _buf, _off = None, 0

    @classmethod
    def _take(cls, nbytes):
        nbytes = _rup(nbytes, 256)
        if cls._buf is None or nbytes > cls._buf.numel():
            cls._buf = torch.empty(max(cls.BYTES, nbytes), dtype=torch.uint8).pin_memory()
            cls._off = 0
        if cls._off + nbytes > cls._buf.numel():
            torch.cuda.current_stream().synchronize()      # every copy issued from the ring has been executed
            cls._off = 0
        o = cls._off
        cls._off += nbytes
        return cls._buf[o:o + nbytes]

    @classmethod
    def upload(cls, dst, src):
        """dst (device tensor, dense) <- src (CPU tensor or nested list of numbers of dst's shape), asynchronously"""
        src = torch.as_tensor(src, dtype=dst.dtype, device="cpu").reshape(dst.shape).contiguous()
        n = src.numel()
        if n == 0:
            return dst
        if not dst.is_cuda:                 # (host-side unit tests of the packing logic)
            dst.copy_(src)
            return dst
        if torch.cuda.is_current_stream_capturing():
            raise L.MI355Error("HostRing.upload inside a graph capture: host values belong to the eager half (prepare_batch)")
        slot = cls._take(n * src.element_size())[: n * src.element_size()].view(dst.dtype).view(dst.shape)
        slot.copy_(src)
        dst.copy_(slot, non_blocking=True)
        return dst


def feed_batch_enabled():
    """MI_FEED_BATCH=0: the per-image torch calls of the reference in prepare_batch (A/B, tests)"""
    import os
    return os.environ.get("MI_FEED_BATCH", "1") != "0"


def _feed_dtype(t):
    return 0 if t.dtype == torch.float32 else (1 if t.dtype in (torch.uint8, torch.bool) else -1)


def normalize_pad_batch(images, dst, mean, std):
    """dst[b] = zero-padded (images[b] - mean) / std for the whole batch in ONE launch (mi_normalize_pad_batch: detr.py:273-278 /
    sparseinst.py:95-98 of the reference's meta architectures, which normalise every image and let ImageList.from_tensors
    copy it into a zero-filled batch tensor).  images: CHW tensors (any device / dtype; moved to dst's device, anything but
    fp32 / uint8 converted to fp32), dst fp32 [B, 3, Hp, Wp] dense, mean / std: three numbers each."""
    B, _, Hp, Wp = dst.shape
    assert dst.is_cuda and dst.dtype == torch.float32 and dst.is_contiguous() and len(images) == B
    jobs = (L.mi_image_job * B)()
    keep = []
    for b, im in enumerate(images):
        im = im.to(dst.device)
        if _feed_dtype(im) < 0 or im.dtype == torch.bool:
            im = im.float()
        im = im.contiguous()
        keep.append(im)
        assert im.dim() == 3 and im.shape[0] == 3, "normalize_pad_batch: CHW images with three channels"
        jobs[b].src, jobs[b].h, jobs[b].w, jobs[b].dtype = im.data_ptr(), int(im.shape[1]), int(im.shape[2]), _feed_dtype(im)
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    L.check(L.lib().mi_normalize_pad_batch(jobs, B, dst.data_ptr(), Hp, Wp, m3, s3, L.stream_ptr()), "mi_normalize_pad_batch")
    return dst


def mask_targets_batch(masks, labels, cap, in_shape, out_shape, tgt, tgtT=None, labels_out=None, t2=None):
    """the batch's ground-truth masks zero-extended to `in_shape`, resized (bilinear, align_corners=False) to `out_shape` and
    packed at fixed capacity in ONE launch (mi_mask_targets_batch: utils/misc.py:148-170 + sparseinst_loss.py:149-151 of the
    reference, which pad, stack and F.interpolate per image).  masks: per image [M, h, w] (fp32 / bool / uint8, on any device),
    labels: per image int64 [M] or None; tgt fp32 [B * cap, Ho * Wo], tgtT bf16 [B, Ho * Wo, cap] or None, labels_out int64
    [B, cap] or None - all written completely (unused rows zero); t2 fp32 [B, cap] or None: every row's sum of squares over the
    pixels (the dice denominators' target term), in a fixed order."""
    B = len(masks)
    dev = tgt.device
    t2ws = None
    if t2 is not None:
        nchunk = (int(out_shape[0]) * int(out_shape[1]) + 63) // 64
        t2ws = torch.empty(B * cap * nchunk, dtype=torch.float32, device=dev)
    jobs = (L.mi_mask_job * B)()
    keep = []
    for b, m in enumerate(masks):
        M = int(m.shape[0])
        jobs[b].M, jobs[b].dtype = M, 0
        if M == 0:
            continue
        m = m.to(dev)
        if _feed_dtype(m) < 0:
            m = m.float()
        m = m.contiguous()
        lab = None
        if labels_out is not None:
            lab = labels[b].to(dev).to(torch.int64).contiguous()
            assert lab.numel() == M
        keep.append((m, lab))
        jobs[b].masks, jobs[b].labels = m.data_ptr(), L.ptr(lab)
        jobs[b].h, jobs[b].w, jobs[b].dtype = int(m.shape[1]), int(m.shape[2]), _feed_dtype(m)
    L.check(L.lib().mi_mask_targets_batch(jobs, B, cap, int(in_shape[0]), int(in_shape[1]), int(out_shape[0]), int(out_shape[1]),
                                          tgt.data_ptr(), L.ptr(tgtT), L.ptr(labels_out), L.ptr(t2), L.ptr(t2ws), L.stream_ptr()),
            "mi_mask_targets_batch")


# ------------------------------------------------------------------------------------------------ grouped weight gradients
class WgradBatch:
    """Weight gradients of the eager module trees (transformer Linears, ResNet convolutions), deferred and issued ONE
    GROUPED LAUNCH PER LAYER (round 6).  A captured DETR-R50 step ran 116 single weight-gradient launches of 18 us + 131
    split-K reductions of 6.3 us - 20 % of the step - almost all of them on the launch floor (a 256 x 256 Linear over 4 368
    rows is 0.6 GFLOP).  Here a backward node only REGISTERS its job (x, dy, the gradient's address); `flush()` - called from
    an identity autograd node at every transformer layer / ResNet block input and once more when the backward pass ends -
    plans the pending jobs as one group (mi_conv2d_wgrad_group_plan: one grid per tile configuration + one reduce grid,
    bias gradients included), uploads the job table and launches it.  x / dy live one layer longer, not the whole step
    (deferring to the END of backward was measured in round 4: -13 %, every activation kept alive and read cold).

    The gradient tensor a node returns to autograd is written LATER by the group's reduce grid; autograd takes such a
    tensor over as the parameter's .grad without touching it (AccumulateGrad steals a fresh contiguous gradient when .grad
    is None - the trainers zero with set_to_none=True), so the job keeps only its ADDRESS: holding the tensor would make
    autograd clone it before it is written.  A parameter that receives gradient twice in one backward would be accumulated
    by a torch kernel before the deferred write - `add()` refuses to defer when .grad is already set.
    MI_WGRAD_LAYER_GROUP=0: every weight gradient as its own launch at its own node (round 5's form)."""
    pending = []        # (mi_wgrad_desc without workspace, keep-alive tensors)
    armed = False       # an end-of-backward flush is queued for the running backward pass ...
    armed_task = -1     # ... of this autograd graph task
    owners = []         # (parameter, address of its deferred gradient): checked when the backward pass ends
    # pinned host memory the job tables are copied from.  Eager steps use a RING (the stream is synchronised when it wraps:
    # once per ~2 000 groups); a graph capture takes its tables from append-only blocks that live as long as the process -
    # every replay of the graph copies from them again
    _ring, _ring_off = None, 0
    _graph_blocks, _graph_off = [], 0
    _graph_dev = []             # device twin of every capture block (None for a block that had to be allocated inside a capture)
    _in_step_capture, _graph_pending = False, []
    ARENA_BYTES = 8 << 20
    stats = dict(flushes=0, jobs=0)

    @staticmethod
    def enabled():
        import os
        return os.environ.get("MI_WGRAD_LAYER_GROUP", "1") != "0"

    @classmethod
    def add(cls, desc, keep, owner=None):
        """owner: the parameter whose gradient this job writes (checked at the end of the backward pass: autograd must have
        taken the returned tensor over as .grad, not copied it)"""
        # one end-of-backward flush per autograd pass.  A pass that died with jobs pending (an exception between a node and
        # its flush point) never ran its callback: its jobs point at gradients that no longer exist - dropped here, when the
        # next pass registers its first job, instead of being written into whatever owns that memory now
        task = torch._C._current_graph_task_id()
        if cls.armed and task != cls.armed_task:
            cls.pending, cls.owners, cls.armed = [], [], False
        cls.pending.append((desc, keep))
        if owner is not None:
            cls.owners.append((owner, int(desc.gw)))
        if not cls.armed:
            cls.armed, cls.armed_task = True, task
            torch.autograd.Variable._execution_engine.queue_callback(cls._end_of_backward)

    @classmethod
    def _end_of_backward(cls):
        cls.armed = False
        cls.flush()
        owners, cls.owners = cls.owners, []
        for p, ptr in owners:
            g = p.grad
            if g is None or not (g.data_ptr() <= ptr < g.data_ptr() + g.numel() * g.element_size()):
                raise L.MI355Error("WgradBatch: autograd did not take a deferred weight gradient over as the parameter's .grad "
                                   f"(parameter {tuple(p.shape)}: a hook or a second use copied it before it was written); "
                                   "set MI_WGRAD_LAYER_GROUP=0")

    @classmethod
    def _pinned(cls, nbytes):
        """(page-locked slot, device slot or None).  Device slots exist inside a `step_capture()` only: the table of a captured
        group never changes between replays, so it is uploaded ONCE when the capture ends instead of by a memcpy node in every
        replay (an in-graph copy from host memory: ~5 us of copy kernel + up to 10 us of idle device on either side, 24 per
        DETR step)."""
        nbytes = _rup(nbytes, 256)
        if torch.cuda.is_current_stream_capturing():
            if not cls._graph_blocks or cls._graph_off + nbytes > cls._graph_blocks[-1].numel():
                # (page-locked allocation inside a capture is not a stream operation: thread-local capture mode allows it)
                cls._graph_blocks.append(torch.empty(max(cls.ARENA_BYTES, nbytes), dtype=torch.uint8).pin_memory())
                cls._graph_dev.append(None)
                cls._graph_off = 0
            o = cls._graph_off
            cls._graph_off += nbytes
            dev = cls._graph_dev[-1]
            return cls._graph_blocks[-1][o:o + nbytes], (dev[o:o + nbytes] if dev is not None and cls._in_step_capture else None)
        if cls._ring is None or nbytes > cls._ring.numel():
            cls._ring = torch.empty(max(cls.ARENA_BYTES, nbytes), dtype=torch.uint8).pin_memory()
            cls._ring_off = 0
        if cls._ring_off + nbytes > cls._ring.numel():
            torch.cuda.current_stream().synchronize()      # every copy issued from the ring has been executed
            cls._ring_off = 0
        o = cls._ring_off
        cls._ring_off += nbytes
        return cls._ring[o:o + nbytes], None

    @classmethod
    def reserve_for_capture(cls):
        """make sure a graph capture finds a pinned block and its device twin (called from eager code before a capture starts:
        allocating page-locked memory INSIDE a global-mode capture is an error, and a device allocation in there would come
        from the graph's private pool)"""
        if not cls._graph_blocks or cls._graph_dev[-1] is None or cls._graph_off + (1 << 20) > cls._graph_blocks[-1].numel():
            cls._graph_blocks.append(torch.empty(cls.ARENA_BYTES, dtype=torch.uint8).pin_memory())
            cls._graph_dev.append(torch.empty(cls.ARENA_BYTES, dtype=torch.uint8, device="cuda"))
            cls._graph_off = 0

    @classmethod
    def step_capture(cls):
        """context manager around the capture of a training step whose owner promises to replay it only after the context has
        ended (graph_step.GraphedTrainStep): the groups' job tables go to persistent device slots, written once on exit.  Any
        other capture keeps the in-graph upload (a memcpy node per group), which needs no such promise."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            cls.reserve_for_capture()
            cls._in_step_capture, cls._graph_pending = True, []
            try:
                yield
            finally:
                cls._in_step_capture = False
                pend, cls._graph_pending = cls._graph_pending, []
                for dev, host in pend:          # (eager, after the capture: ordinary stream-ordered copies, then one wait)
                    dev.copy_(host, non_blocking=True)
                if pend:
                    torch.cuda.current_stream().synchronize()
        return ctx()

    @classmethod
    def run_now(cls, jobs):
        """`jobs` [(mi_wgrad_desc, keep-alive tensors)] as one grouped launch, at once (forward-pass users of the
        weight-gradient kernel: the per-image outer products of SparseInst)"""
        saved, cls.pending = cls.pending, list(jobs)
        try:
            cls.flush()
        finally:
            cls.pending = saved + cls.pending

    @classmethod
    def flush(cls):
        jobs, cls.pending = cls.pending, []
        n = len(jobs)
        if n == 0:
            return
        lib, dev = L.lib(), jobs[0][1][0].device
        descs = (L.mi_wgrad_desc * n)()
        for d, (src, _) in zip(descs, jobs):
            C.memmove(C.byref(d), C.byref(src), C.sizeof(L.mi_wgrad_desc))
        meta = L.mi_wgrad_group()
        L.check(lib.mi_conv2d_wgrad_group_plan(descs, n, None, None, 0, C.byref(meta)), "wgrad_group_plan (sizes)")
        ws = torch.empty(max(int(meta.ws_bytes), 256), dtype=torch.uint8, device=dev)
        nb = int(meta.table_bytes)
        host, table = cls._pinned(nb)
        L.check(lib.mi_conv2d_wgrad_group_plan(descs, n, ws.data_ptr(), host.data_ptr(), nb, C.byref(meta)), "wgrad_group_plan")
        sp = L.stream_ptr()
        if table is None:
            table = torch.empty(_rup(nb, 256), dtype=torch.uint8, device=dev)
            L.check(lib.mi_upload_async(table.data_ptr(), host.data_ptr(), nb, sp), "upload_async")
        else:
            cls._graph_pending.append((table, host))      # written when the capture ends (step_capture)
        L.check(lib.mi_conv2d_wgrad_group_run(C.byref(meta), table.data_ptr(), sp), "wgrad_group_run")
        cls.stats["flushes"] += 1
        cls.stats["jobs"] += n


def wgrad_can_defer(*params):
    """the weight gradient may be written later only into a tensor autograd takes over untouched: every parameter is a leaf
    whose .grad is None (else AccumulateGrad adds the returned tensor to .grad right away, and a non-leaf weight's producer
    reads it right away)"""
    return WgradBatch.enabled() and all(p is None or (p.is_leaf and p.grad is None) for p in params)


class _WgradFlushFn(torch.autograd.Function):
    """identity; its backward runs when every consumer of the tensor inside the layer has produced its input gradient -
    i.e. when the layer's weight-gradient jobs are all registered - and issues them as one group"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        WgradBatch.flush()
        return g


def wgrad_flush_point(x):
    """mark a layer boundary: the weight gradients registered by the layer's backward are launched when the backward
    pass reaches this tensor"""
    if WgradBatch.enabled() and torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled():
        return _WgradFlushFn.apply(x)
    return x


def pack_images(w32, Cout, Cin, kh, kw, CinP, CoutP, CoutPK, CinPN, fwd=True, dgrad=True, scale=None):
    """(forward image wf[tap][ci/8][co][ci%8] or None, data-gradient image wd[tap][co/8][ci][co%8] or None) of the fp32
    OIHW weight w32 (contiguous); scale: fp32 [Cout] folded in (W * scale[co], fp32 product then the bf16 rounding)"""
    KK = kh * kw

    def pack_now(wf, wd):
        if scale is None:
            L.check(L.lib().mi_pack_conv_weight(w32.data_ptr(), Cout, Cin, kh, kw, L.ptr(wf), CinP, CoutP, L.ptr(wd), CoutPK, CinPN,
                                                L.stream_ptr()), "mi_pack_conv_weight")
        else:
            assert scale.dtype == torch.float32 and scale.is_contiguous() and scale.numel() == Cout
            L.check(L.lib().mi_pack_conv_weight_scaled(w32.data_ptr(), scale.data_ptr(), Cout, Cin, kh, kw, L.ptr(wf), CinP, CoutP,
                                                       L.ptr(wd), CoutPK, CinPN, L.stream_ptr()), "mi_pack_conv_weight_scaled")

    reg = WeightImages.active
    if reg is not None:
        hit = reg.get(w32, Cout, Cin, KK, CinP, CoutP, CoutPK, CinPN, fwd, dgrad, scale, pack_now)
        if hit is not None:
            return hit
    dev = w32.device
    wf = torch.empty(KK * CinP * CoutP, dtype=torch.bfloat16, device=dev) if fwd else None
    wd = torch.empty(KK * CoutPK * CinPN, dtype=torch.bfloat16, device=dev) if dgrad else None
    pack_now(wf, wd)
    return wf, wd


# ------------------------------------------------------------------------------------------------ descriptors
def _nhwc(x):
    """NCHW tensor -> (bf16 [N,H,W,C] contiguous view/copy)"""
    return x.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()


def _nhwc_v(x):
    """NCHW-shaped tensor -> NHWC-shaped bf16 tensor WITHOUT a copy when x's memory already is an NHWC image whose pixel
    stride may exceed its channel count: a channel slice of a wider map (the gradient torch.cat hands each of its inputs,
    one group of a grouped convolution).  Every kernel takes (pointer, pixel stride); `_ld(t)` is that stride.  Anything
    else gets the contiguous copy of _nhwc.  (MI_NHWC_VIEWS=0: always copy - the round-4 form, A/B switch.)
    SparseInst-R50's step made 18 such copies (251 MB) + 6 more in its resize backward."""
    if x.dtype == torch.bfloat16 and x.dim() == 4 and _NHWC_VIEWS:
        N, Cc, H, W = x.shape
        sn, sc, sh, sw = x.stride()
        if (W > 1 and H > 1 and sc == 1 and Cc % 8 == 0 and sw >= Cc and sw % 8 == 0 and sh == W * sw
                and (N == 1 or sn == H * W * sw) and x.data_ptr() % 16 == 0):
            return x.permute(0, 2, 3, 1)
    return _nhwc(x)


def nhwc_strided_ok(t):
    """an NHWC-SHAPED bf16 tensor the kernels can read in place through (data_ptr, _ld): dense, or a channel slice"""
    if t.dtype != torch.bfloat16 or t.dim() != 4:
        return False
    if t.is_contiguous():
        return True
    N, H, W, Cc = t.shape
    sn, sh, sw, sc = t.stride()
    return bool(_NHWC_VIEWS and W > 1 and H > 1 and sc == 1 and Cc % 8 == 0 and sw >= Cc and sw % 8 == 0 and sh == W * sw
                and (N == 1 or sn == H * W * sw) and t.data_ptr() % 16 == 0)


def _ld(t):
    """pixel stride (elements) of an NHWC-shaped tensor that is dense or came from _nhwc_v"""
    return t.shape[-1] if t.is_contiguous() else t.stride(2)


import os as _os
_NHWC_VIEWS = _os.environ.get("MI_NHWC_VIEWS", "1") != "0"


def _nchw(y, C):
    """bf16 [N,H,W,Cld] -> NCHW view (channels_last memory) of the first C channels"""
    return y.permute(0, 3, 1, 2)[:, :C]


def _conv_desc(x, ldx, N, H, W, w_img, K, y, ldy, outH, outW, Cout, CoutPad, taps, in_stride=1, out_stride=1, oy=0, ox=0,
               gridH=None, gridW=None, bias=None, stats=None, nslots=0, flags=0, aux=None):
    """aux = (address, pixel stride) of the second tensor of the MI_CONV_RELUMASK / MI_CONV_ADDRELU epilogues"""
    d = L.mi_conv_desc()
    d.x, d.w, d.y = x, w_img.data_ptr(), y
    d.bias, d.stats_acc = L.ptr(bias), L.ptr(stats)
    d.ldx, d.ldy = ldx, ldy
    d.N, d.H, d.W, d.outH, d.outW = N, H, W, outH, outW
    d.gridH, d.gridW = outH if gridH is None else gridH, outW if gridW is None else gridW
    d.in_stride, d.out_stride, d.out_oy, d.out_ox = in_stride, out_stride, oy, ox
    d.K8, d.Cout, d.CoutPad, d.ntaps = K // 8, Cout, CoutPad, len(taps)
    for t, (dy, dx, wi) in enumerate(taps):
        d.tap_dy[t], d.tap_dx[t], d.tap_w[t] = dy, dx, wi
    d.flags, d.stats_slots = flags, nslots
    if aux is not None:
        d.bn_y, d.bn_ldy = aux
    return d


def _run_conv(d, what):
    L.check(L.lib().mi_conv2d(C.byref(d), L.stream_ptr()), what)


class _ConvGeom:
    """everything shape-dependent of one k x k / stride s convolution on the implicit-GEMM kernels"""

    def __init__(self, x_shape, w_shape, stride, padding):
        self.N, self.Cin, self.H, self.W = x_shape
        self.Cout, cin_w, self.k, kw = w_shape
        if cin_w != self.Cin or kw != self.k or self.k not in (1, 3) or stride not in (1, 2) or padding != (self.k - 1) // 2:
            raise L.MI355Error(f"mi355 conv: unsupported geometry weight {tuple(w_shape)} stride {stride} padding {padding} "
                               "(served: 1x1 and 3x3, stride 1 or 2, padding (k-1)/2, groups 1)")
        self.s, self.pad = stride, padding
        self.Ho = (self.H + 2 * padding - self.k) // stride + 1
        self.Wo = (self.W + 2 * padding - self.k) // stride + 1
        self.CinP, self.CoutP = _rup(self.Cin, 32), _rup(self.Cout, 32)
        self.KK = self.k * self.k

    def pack(self, weight, fwd=True, dgrad=True, scale=None):
        """(forward image, data-gradient image) of an OIHW weight; an image that is not asked for is None.
        scale: fp32 [Cout] folded into the images (W * scale[co], fp32 product then the bf16 rounding)"""
        return pack_images(weight.detach().float().contiguous(), self.Cout, self.Cin, self.k, self.k, self.CinP, self.CoutP,
                           self.CoutP, self.CinP, fwd, dgrad, scale)

    def wgrad_scaled(self, xh, dyh, scale, defer=False, owner=None):
        """weight gradient of a layer whose image carried a folded per-Cout factor: scale[co] * dW' (the factor is applied
        to the fp32 sums in the split-K reduction: mi_wgrad_desc.row_scale)"""
        assert scale.dtype == torch.float32 and scale.is_contiguous() and scale.numel() == self.Cout
        return self.wgrad(xh, dyh, row_scale=scale, defer=defer, owner=owner)

    def pad_in(self, x):
        """NCHW -> bf16 [N,H,W,CinP] (zero pad channels)"""
        xh = _nhwc_v(x) if self.CinP == self.Cin else _nhwc(x)
        if self.CinP != self.Cin:
            xp = torch.zeros(self.N, self.H, self.W, self.CinP, dtype=torch.bfloat16, device=x.device)
            xp[..., : self.Cin] = xh
            xh = xp
        return xh

    def fwd(self, xh, wf, y, bias=None, stats=None, nslots=0, relu=False, add_relu=None):
        """add_relu: a bf16 NHWC tensor of y's shape: y = relu(bf16(conv + bias) + add_relu) in the epilogue (MI_CONV_ADDRELU)"""
        taps = [(r - self.pad, s - self.pad, r * self.k + s) for r in range(self.k) for s in range(self.k)]
        fl = L.MI_CONV_RELU if relu else 0
        aux = None
        if add_relu is not None:
            fl, aux = L.MI_CONV_ADDRELU, (add_relu.data_ptr(), _ld(add_relu))
        _run_conv(_conv_desc(xh.data_ptr(), _ld(xh), self.N, self.H, self.W, wf, self.CinP, y.data_ptr(), y.shape[-1],
                             self.Ho, self.Wo, self.Cout, self.CoutP, taps, in_stride=self.s, bias=bias, stats=stats,
                             nslots=nslots, flags=fl, aux=aux), "mi_conv2d (forward)")

    def dgrad(self, dyh, wd, dx, accum=False, relu_mask=None):
        """dyh bf16 [N,Ho,Wo,CoutP] (zero pad channels) -> dx bf16 [N,H,W,CinP] (real channels written).
        accum: dx += (MI_CONV_ACCUM; the strided forms then touch only the pixels that receive a gradient, so dx need not be
        zeroed for them)"""
        k, pad = self.k, self.pad
        fl = L.MI_CONV_ACCUM if accum else 0
        aux = None
        if relu_mask is not None:       # dx *= (relu_mask > 0) in the epilogue (MI_CONV_RELUMASK): relu_mask = the ReLU OUTPUT that was this conv's input
            # (with accum: the mask applies to the accumulated sum - the last data gradient into a block input; stride 1 only:
            #  the strided forms visit only the pixels that receive a gradient)
            assert tuple(relu_mask.shape[:3]) == (self.N, self.H, self.W) and not (accum and self.s != 1)
            fl, aux = fl | L.MI_CONV_RELUMASK, (relu_mask.data_ptr(), _ld(relu_mask))
        ldy = _ld(dyh)
        if self.s == 1:
            taps = [(pad - r, pad - s, r * k + s) for r in range(k) for s in range(k)]
            _run_conv(_conv_desc(dyh.data_ptr(), ldy, self.N, self.Ho, self.Wo, wd, self.CoutP, dx.data_ptr(),
                                 self.CinP, self.H, self.W, self.Cin, self.CinP, taps, flags=fl, aux=aux), "mi_conv2d (dgrad)")
            return
        if k == 1:      # 1x1 stride 2 (ResNet shortcut): only the even pixels receive a gradient; dx arrives zeroed
            _run_conv(_conv_desc(dyh.data_ptr(), ldy, self.N, self.Ho, self.Wo, wd, self.CoutP, dx.data_ptr(),
                                 self.CinP, self.H, self.W, self.Cin, self.CinP, [(0, 0, 0)], out_stride=2, oy=0, ox=0,
                                 gridH=(self.H + 1) // 2, gridW=(self.W + 1) // 2, flags=fl, aux=aux), "mi_conv2d (dgrad 1x1 s2)")
            return
        cls_taps = {0: [(1, 0)], 1: [(0, 1), (2, 0)]}   # output-pixel parity -> [(kernel row, dy offset)]
        for py in (0, 1):
            for px in (0, 1):
                taps = [(oy, ox, r * 3 + s) for (r, oy) in cls_taps[py] for (s, ox) in cls_taps[px]]
                gh, gw = (self.H - py + 1) // 2, (self.W - px + 1) // 2
                _run_conv(_conv_desc(dyh.data_ptr(), ldy, self.N, self.Ho, self.Wo, wd, self.CoutP, dx.data_ptr(),
                                     self.CinP, self.H, self.W, self.Cin, self.CinP, taps, out_stride=2, oy=py, ox=px,
                                     gridH=gh, gridW=gw, flags=fl, aux=aux), "mi_conv2d (dgrad s2)")

    def wgrad(self, xh, dyh, row_scale=None, gbias=None, defer=False, owner=None):
        """gbias: fp32 [Cout] tensor that receives the bias gradient (column sums of dyh) from the same two launches.
        defer: only register the job with WgradBatch (the caller flushes: one grouped launch for several layers); xh / dyh
        must not be modified before that flush"""
        gw = torch.empty(self.Cout, self.Cin, self.k, self.k, dtype=torch.float32, device=xh.device)
        d = L.mi_wgrad_desc()
        d.x, d.dy, d.gw = xh.data_ptr(), dyh.data_ptr(), gw.data_ptr()
        d.row_scale, d.gbias = L.ptr(row_scale), L.ptr(gbias)
        d.ldx, d.ldy, d.N, d.H, d.W, d.outH, d.outW, d.stride = _ld(xh), _ld(dyh), self.N, self.H, self.W, self.Ho, self.Wo, self.s
        d.Cin, d.Cout, d.CinPad, d.CoutPad, d.ntaps = self.Cin, self.Cout, self.CinP, self.CoutP, self.KK
        for t in range(self.KK):
            d.tap_dy[t], d.tap_dx[t] = t // self.k - self.pad, t % self.k - self.pad
        if defer:
            WgradBatch.add(d, (xh, dyh, row_scale), owner)
            return gw
        need = L.lib().mi_conv2d_wgrad_plan(C.byref(d))
        L.check(need, "mi_conv2d_wgrad_plan")
        ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=xh.device)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
        L.check(L.lib().mi_conv2d_wgrad(C.byref(d), L.stream_ptr()), "mi_conv2d_wgrad")
        return gw


def wgrad_bias_fused(npix=0):
    """the bias gradient from the weight-gradient launches (mi_wgrad_desc.gbias) instead of a column-sum launch of its own.
    Pays where the layer is a handful of pixel tiles per block - the transformer's token-row GEMMs (T = 4 200: DETR-R50 267 ->
    272 images/s, same box) - and LOSES on the long pixel ranges of SparseInst's 80 x 80 convolutions (-1.2 %: the blocks of
    input-channel tile 0 re-read every dy tile while the others wait for them; profiles/r05_wgrad_bias_ab.txt): above
    MI_WGRAD_BIAS_MAXPIX (16 384) pixels the separate launch stays.  MI_WGRAD_BIAS=0: never (round 4's form); =2: always."""
    import os
    m = os.environ.get("MI_WGRAD_BIAS", "1")
    if m == "0":
        return False
    return m == "2" or npix <= int(os.environ.get("MI_WGRAD_BIAS_MAXPIX", "16384"))


def _colsum(dyh, C_):
    """fp32 column sums over all pixels of a bf16 [..., CP] map (bias gradient)"""
    CP = dyh.shape[-1]
    T = dyh.numel() // CP
    out = torch.empty(CP, dtype=torch.float32, device=dyh.device)
    ws = torch.empty(128 * CP, dtype=torch.float32, device=dyh.device)
    L.check(L.lib().mi_colsum_bf16_wide(dyh.data_ptr(), _ld(dyh), T, CP, out.data_ptr(), 0, ws.data_ptr(), L.stream_ptr()), "mi_colsum_bf16_wide")
    return out[:C_]


def _pad_last(t, CP):
    """bf16 [N,H,W,C] -> [N,H,W,CP] with zero pad channels"""
    if t.shape[-1] == CP:
        return t
    out = torch.zeros(*t.shape[:-1], CP, dtype=t.dtype, device=t.device)
    out[..., : t.shape[-1]] = t
    return out


# ------------------------------------------------------------------------------------------------ mi355::conv2d
_LIBDEF.define("conv2d(Tensor x, Tensor weight, Tensor? bias, int stride, int padding) -> Tensor")
_LIBDEF.define("conv2d_backward(Tensor grad, Tensor x, Tensor weight, bool has_bias, int stride, int padding, bool defer=False) -> (Tensor, Tensor, Tensor)")


def _conv2d_cuda(x, weight, bias, stride, padding, relu=False):
    g = _ConvGeom(x.shape, weight.shape, stride, padding)
    wf, _ = g.pack(weight, dgrad=False)
    y = torch.empty(g.N, g.Ho, g.Wo, g.CoutP, dtype=torch.bfloat16, device=x.device)
    b32 = None
    if bias is not None:
        if g.CoutP == g.Cout and bias.dtype == torch.float32 and bias.is_contiguous():
            b32 = bias.detach()        # (no padded copy: a fill + a copy launch per call otherwise)
        else:
            b32 = torch.zeros(g.CoutP, dtype=torch.float32, device=x.device)
            b32[: g.Cout] = bias.detach().float()
    g.fwd(g.pad_in(x), wf, y, bias=b32, relu=relu)
    return _nchw(y, g.Cout)


def _conv2d_backward_cuda(grad, x, weight, has_bias, stride, padding, defer=False):
    """defer: the weight gradient only REGISTERS with WgradBatch (`weight` is the parameter that will own it) and leaves with
    the next grouped launch - a later layer's flush point or the end of the backward pass"""
    g = _ConvGeom(x.shape, weight.shape, stride, padding)
    _, wd = g.pack(weight, fwd=False)
    dyh = _pad_last(_nhwc_v(grad), g.CoutP)
    xh = g.pad_in(x)
    # the data gradient writes every real channel of every pixel - except the 1x1 stride-2 form (odd pixels receive
    # nothing) and pad channels: only those need the zero fill (it was a full-tensor pass per convolution)
    full = g.CinP == g.Cin and not (g.k == 1 and g.s == 2)
    dx = (torch.empty if full else torch.zeros)(g.N, g.H, g.W, g.CinP, dtype=torch.bfloat16, device=x.device)
    g.dgrad(dyh, wd, dx)
    own = weight if defer else None
    if has_bias and wgrad_bias_fused(g.N * g.Ho * g.Wo):
        gb = torch.empty(g.Cout, dtype=torch.float32, device=x.device)
        gw = g.wgrad(xh, dyh, gbias=gb, defer=defer, owner=own)       # (the bias gradient leaves with the weight gradient: no column-sum launch)
    else:
        gw = g.wgrad(xh, dyh, defer=defer, owner=own)
        gb = _colsum(dyh, g.Cout) if has_bias else torch.zeros(0, device=x.device)
    return _nchw(dx, g.Cin), gw, gb


torch.library.impl(_LIBDEF, "conv2d", "CUDA")(_conv2d_cuda)
torch.library.impl(_LIBDEF, "conv2d_backward", "CUDA")(_conv2d_backward_cuda)


def _conv2d_setup(ctx, inputs, output):
    x, weight, bias, stride, padding = inputs
    ctx.save_for_backward(x, weight)
    ctx.has_bias, ctx.stride, ctx.padding = bias is not None, stride, padding
    ctx.params = (weight, bias)        # (the parameter objects themselves: whether their .grad is still unset decides the deferral)


def _conv_op_defer(weight, bias=None):
    """the weight gradient of a mi355::conv2d(_relu) node may join the next grouped launch (WgradBatch) when autograd will
    take the returned tensor over untouched: fp32 leaf parameter without a .grad yet (wgrad_can_defer).  MI_WGRAD_CONV_DEFER=0:
    every convolution's weight gradient as its own launch + reduce (round 5's form)"""
    import os
    return (os.environ.get("MI_WGRAD_CONV_DEFER", "1") != "0" and weight.dtype == torch.float32 and weight.requires_grad
            and wgrad_can_defer(weight, bias))


def _conv2d_bwd(ctx, grad):
    x, weight = ctx.saved_tensors
    need_gb = ctx.has_bias and ctx.needs_input_grad[2]     # (a frozen-norm shift passed as bias has no gradient to compute)
    w_, b_ = ctx.params
    dx, gw, gb = torch.ops.mi355.conv2d_backward(grad, x, w_, need_gb, ctx.stride, ctx.padding,
                                                 ctx.needs_input_grad[1] and _conv_op_defer(w_, b_ if need_gb else None))
    return dx.to(x.dtype), gw.to(weight.dtype), (gb if need_gb else None), None, None


torch.library.register_autograd("mi355::conv2d", _conv2d_bwd, setup_context=_conv2d_setup)

class ConvPaddedFn(torch.autograd.Function):
    """conv2d (+ ReLU) on an input that ALREADY is the kernels' operand: bf16 NHWC, channels zero-padded to a multiple of
    32 (xh [N, H, W, CinP]; `cin` real channels).  mi355::conv2d builds that operand from an NCHW tensor on every call -
    forward AND backward - which for a 258-channel input (SparseInst's decoder: 256 features + 2 coordinates, two branches)
    was a 26 MB layout copy + a 29 MB zero fill + a 29 MB strided copy, four times per step.  Here the caller builds it
    once (sparseinst._CoordCat) and both branches, forward and backward, read it.  Returns NCHW (channels_last memory);
    backward returns the gradient for xh with its pad channels untouched (never read: the producer slices them off)."""

    @staticmethod
    def forward(ctx, xh, weight, bias, cin, stride, padding, relu):
        N, H, W, CP = xh.shape
        g = _ConvGeom((N, cin, H, W), weight.shape, stride, padding)
        if CP != g.CinP or xh.dtype != torch.bfloat16 or not xh.is_contiguous():
            raise L.MI355Error(f"ConvPaddedFn: operand {tuple(xh.shape)} {xh.dtype} is not the padded bf16 NHWC image of {cin} channels")
        wf, wd = g.pack(weight, dgrad=ctx.needs_input_grad[0])
        y = torch.empty(N, g.Ho, g.Wo, g.CoutP, dtype=torch.bfloat16, device=xh.device)
        b32 = None
        if bias is not None:
            if g.CoutP == g.Cout and bias.dtype == torch.float32 and bias.is_contiguous():
                b32 = bias.detach()
            else:
                b32 = torch.zeros(g.CoutP, dtype=torch.float32, device=xh.device)
                b32[: g.Cout] = bias.detach().float()
        g.fwd(xh, wf, y, bias=b32, relu=relu)
        ctx.g, ctx.relu, ctx.has_bias = g, relu, bias is not None
        ctx.params = (weight, bias)
        ctx.save_for_backward(xh, wd, y if relu else None)
        return _nchw(y, g.Cout)

    @staticmethod
    def backward(ctx, grad):
        xh, wd, y = ctx.saved_tensors
        g = ctx.g
        dyh = _nhwc_v(grad)
        if ctx.relu:
            dyh = dyh.contiguous()
            yh = y[..., : g.Cout].contiguous()
            gm = torch.empty_like(dyh)
            L.check(L.lib().mi_ew_bf16(dyh.data_ptr(), yh.data_ptr(), gm.data_ptr(), dyh.numel(), 2, L.stream_ptr()), "mi_ew_bf16 relu'")
            dyh = gm
        dyh = _pad_last(dyh, g.CoutP)
        dx = gw = gb = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(g.N, g.H, g.W, g.CinP, dtype=torch.bfloat16, device=xh.device)
            if g.CinP != g.Cin:
                dx[..., g.Cin:].zero_()          # (the pad channels: a few bytes per pixel, so sums of such gradients stay finite)
            g.dgrad(dyh, wd, dx)
        need_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            df = _conv_op_defer(ctx.params[0], ctx.params[1] if need_gb else None)
            own = ctx.params[0] if df else None
            if need_gb and wgrad_bias_fused(g.N * g.Ho * g.Wo):
                gb = torch.empty(g.Cout, dtype=torch.float32, device=xh.device)
                gw = g.wgrad(xh, dyh, gbias=gb, defer=df, owner=own)
            else:
                gw = g.wgrad(xh, dyh, defer=df, owner=own)
        if need_gb and gb is None:
            gb = _colsum(dyh, g.Cout)
        return dx, gw, gb, None, None, None, None


# conv2d + ReLU in the convolution's epilogue (MI_CONV_RELU): detectron2's Conv2d(norm=FrozenBN, activation=relu) as ONE
# launch - the separate ReLU was a read + write of the whole map per convolution.  Backward: the ReLU mask comes from the
# saved OUTPUT (out > 0), then the plain convolution backward.
_LIBDEF.define("conv2d_relu(Tensor x, Tensor weight, Tensor? bias, int stride, int padding) -> Tensor")
torch.library.impl(_LIBDEF, "conv2d_relu", "CUDA")(lambda x, weight, bias, stride, padding: _conv2d_cuda(x, weight, bias, stride, padding, relu=True))


def _conv2d_relu_setup(ctx, inputs, output):
    x, weight, bias, stride, padding = inputs
    ctx.save_for_backward(x, weight, output)
    ctx.has_bias, ctx.stride, ctx.padding = bias is not None, stride, padding
    ctx.params = (weight, bias)


def _conv2d_relu_bwd(ctx, grad):
    x, weight, out = ctx.saved_tensors
    gh, oh = _nhwc(grad), _nhwc(out)
    gm = torch.empty_like(gh)
    L.check(L.lib().mi_ew_bf16(gh.data_ptr(), oh.data_ptr(), gm.data_ptr(), gh.numel(), 2, L.stream_ptr()), "mi_ew_bf16 relu'")
    need_gb = ctx.has_bias and ctx.needs_input_grad[2]
    w_, b_ = ctx.params
    dx, gw, gb = torch.ops.mi355.conv2d_backward(gm.permute(0, 3, 1, 2), x, w_, need_gb, ctx.stride, ctx.padding,
                                                 ctx.needs_input_grad[1] and _conv_op_defer(w_, b_ if need_gb else None))
    return dx.to(x.dtype), gw.to(weight.dtype), (gb if need_gb else None), None, None


torch.library.register_autograd("mi355::conv2d_relu", _conv2d_relu_bwd, setup_context=_conv2d_relu_setup)


# ------------------------------------------------------------------------------------------------ mi355::conv_bn_silu
# functional (register_autograd requires it): in training mode the op RETURNS the batch statistics (stats[2] = mean,
# stats[3] = 1/sqrt(var + eps)) and `base_conv_forward` below applies nn.BatchNorm2d's running-statistics update
_LIBDEF.define("conv_bn_silu(Tensor x, Tensor weight, Tensor gamma, Tensor beta, Tensor running_mean, "
               "Tensor running_var, int stride, float eps, bool training) -> (Tensor, Tensor, Tensor)")
_LIBDEF.define("conv_bn_silu_backward(Tensor grad, Tensor x, Tensor weight, Tensor gamma, Tensor y, Tensor stats, "
               "int stride) -> (Tensor, Tensor, Tensor, Tensor)")


def _conv_bn_silu_cuda(x, weight, gamma, beta, running_mean, running_var, stride, eps, training):
    """-> (out NCHW bf16, y = raw conv output bf16 [N,Ho,Wo,Cout], stats fp32 [4, Cout] = scale, shift, mean, invstd)"""
    g = _ConvGeom(x.shape, weight.shape, stride, (weight.shape[2] - 1) // 2)
    if g.Cout % 8:
        raise L.MI355Error("mi355::conv_bn_silu: Cout must be a multiple of 8")
    dev = x.device
    wf, _ = g.pack(weight, dgrad=False)
    y = torch.empty(g.N, g.Ho, g.Wo, g.Cout, dtype=torch.bfloat16, device=dev)
    out = torch.empty(g.N, g.Ho, g.Wo, g.Cout, dtype=torch.bfloat16, device=dev)
    stats = torch.empty(4, g.Cout, dtype=torch.float32, device=dev)
    npix = g.N * g.Ho * g.Wo
    ga, be = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    lib = L.lib()
    if training:
        acc = torch.zeros(L.MI_BN_SLOTS * g.CoutP * 2, dtype=torch.float64, device=dev)
        g.fwd(g.pad_in(x), wf, y, stats=acc, nslots=L.MI_BN_SLOTS)
        L.check(lib.mi_bn_act_fwd(y.data_ptr(), g.Cout, acc.data_ptr(), L.MI_BN_SLOTS, npix, ga.data_ptr(), be.data_ptr(),
                                  eps, 0.0, None, None, None, stats[0].data_ptr(), stats[1].data_ptr(),
                                  stats[2].data_ptr(), stats[3].data_ptr(), None, 0, out.data_ptr(), g.Cout, npix, g.Cout, 1,
                                  L.stream_ptr()), "mi_bn_act_fwd")
    else:
        g.fwd(g.pad_in(x), wf, y)
        L.check(lib.mi_bn_eval_affine(ga.data_ptr(), be.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), eps,
                                      g.Cout, stats[0].data_ptr(), stats[1].data_ptr(), L.stream_ptr()), "mi_bn_eval_affine")
        L.check(lib.mi_bn_act_fwd(y.data_ptr(), g.Cout, None, 0, 0, None, None, eps, 0.0, None, None, None,
                                  stats[0].data_ptr(), stats[1].data_ptr(), None, None, None, 0, out.data_ptr(), g.Cout,
                                  npix, g.Cout, 1, L.stream_ptr()), "mi_bn_act_fwd (eval)")
    return _nchw(out, g.Cout), y, stats


def _conv_bn_silu_backward_cuda(grad, x, weight, gamma, y, stats, stride):
    g = _ConvGeom(x.shape, weight.shape, stride, (weight.shape[2] - 1) // 2)
    dev = x.device
    lib = L.lib()
    npix = g.N * g.Ho * g.Wo
    da = _nhwc(grad)
    ga = gamma.detach().float().contiguous()
    dacc = torch.zeros(L.MI_BN_SLOTS * g.CoutP * 2, dtype=torch.float64, device=dev)
    C8 = g.Cout // 8
    nblk = max(1, min(1024, math.ceil(npix / (256 // C8) / 4)))
    L.check(lib.mi_bn_act_bwd_reduce(da.data_ptr(), g.Cout, y.data_ptr(), g.Cout, stats[0].data_ptr(), stats[1].data_ptr(),
                                     stats[2].data_ptr(), stats[3].data_ptr(), dacc.data_ptr(), L.MI_BN_SLOTS, nblk, npix,
                                     g.Cout, 1, L.stream_ptr()), "mi_bn_act_bwd_reduce")
    dyh = torch.zeros(g.N, g.Ho, g.Wo, g.CoutP, dtype=torch.bfloat16, device=dev)
    dgamma = torch.empty(g.Cout, dtype=torch.float32, device=dev)
    dbeta = torch.empty(g.Cout, dtype=torch.float32, device=dev)
    L.check(lib.mi_bn_act_bwd_apply(da.data_ptr(), g.Cout, y.data_ptr(), g.Cout, stats[0].data_ptr(), stats[1].data_ptr(),
                                    stats[2].data_ptr(), stats[3].data_ptr(), ga.data_ptr(), dacc.data_ptr(), L.MI_BN_SLOTS,
                                    npix, dgamma.data_ptr(), dbeta.data_ptr(), dyh.data_ptr(), g.CoutP, None, 0, 0, npix,
                                    g.Cout, 1, L.stream_ptr()), "mi_bn_act_bwd_apply")
    _, wd = g.pack(weight, fwd=False)
    dx = torch.zeros(g.N, g.H, g.W, g.CinP, dtype=torch.bfloat16, device=dev)
    g.dgrad(dyh, wd, dx)
    gw = g.wgrad(g.pad_in(x), dyh)
    return _nchw(dx, g.Cin), gw, dgamma, dbeta


torch.library.impl(_LIBDEF, "conv_bn_silu", "CUDA")(_conv_bn_silu_cuda)
torch.library.impl(_LIBDEF, "conv_bn_silu_backward", "CUDA")(_conv_bn_silu_backward_cuda)


def _cbs_setup(ctx, inputs, output):
    x, weight, gamma = inputs[0], inputs[1], inputs[2]
    out, y, stats = output
    ctx.save_for_backward(x, weight, gamma, y, stats)
    ctx.stride, ctx.training = inputs[6], inputs[8]
    ctx.mark_non_differentiable(y, stats)


def _cbs_bwd(ctx, grad, _gy, _gs):
    x, weight, gamma, y, stats = ctx.saved_tensors
    if not ctx.training:
        raise L.MI355Error("mi355::conv_bn_silu: backward through the eval-mode (running statistics) form is not implemented")
    dx, gw, dgamma, dbeta = torch.ops.mi355.conv_bn_silu_backward(grad, x, weight, gamma, y, stats, ctx.stride)
    return dx.to(x.dtype), gw.to(weight.dtype), dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), None, None, None, None, None


torch.library.register_autograd("mi355::conv_bn_silu", _cbs_bwd, setup_context=_cbs_setup)


# ------------------------------------------------------------------------------------------------ mi355::batched_nms
_LIBDEF.define("batched_nms(Tensor boxes, Tensor scores, Tensor idxs, float iou_threshold) -> Tensor")


def _batched_nms_cuda(boxes, scores, idxs, iou_threshold):
    from .modeling.postprocess import batched_nms
    return batched_nms(boxes, scores, idxs, iou_threshold)


torch.library.impl(_LIBDEF, "batched_nms", "CUDA")(_batched_nms_cuda)


# ------------------------------------------------------------------------------------------------ mi355::yolox_loss
_LIBDEF.define("yolox_loss(Tensor raw, Tensor labels, Tensor anchors, int num_classes) -> (Tensor, Tensor)")


def _yolox_loss_cuda(raw, labels, anchors, num_classes):
    """raw [B,A,5+nc] fp32 head output (undecoded), labels [B,L,5] (cls,cx,cy,w,h), anchors [A,3] (gx,gy,stride) ->
    (losses fp32 [8] = total, 5*iou, obj, cls, l1, num_fg/num_gt, num_fg, num_gt;  d(sum of the 4 losses)/d(raw))"""
    B, A, nch = raw.shape
    ML = labels.shape[1]
    dev = raw.device
    t = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
    rd, ld, ad = raw.detach().float().contiguous(), labels.detach().float().contiguous(), anchors.detach().float().contiguous()
    ws = dict(cost=t(B, ML, A), iou=t(B, ML, A), match=t(B, ML, A, dt=torch.uint8), ngt=t(B, dt=torch.int32),
              fg=t(B, A, dt=torch.uint8), matched_gt=t(B, A, dt=torch.int32), matched_iou=t(B, A),
              partial=t(B * ((A + 255) // 256), 4), out=t(8))
    d = L.mi_yolox_loss_desc()
    d.preds, d.labels, d.anchors = rd.data_ptr(), ld.data_ptr(), ad.data_ptr()
    d.B, d.A, d.ncls, d.max_labels, d.gmax = B, A, num_classes, ML, ML
    for k in ("cost", "iou", "match", "ngt", "fg", "matched_gt", "matched_iou", "partial", "out"):
        setattr(d, k, ws[k].data_ptr())
    L.check(L.lib().mi_yolox_loss_fwd(C.byref(d), L.stream_ptr()), "mi_yolox_loss_fwd")
    gw = torch.ones(4, dtype=torch.float32, device=dev)
    dpreds = t(B, A, nch)
    L.check(L.lib().mi_yolox_loss_bwd(C.byref(d), gw.data_ptr(), dpreds.data_ptr(), L.stream_ptr()), "mi_yolox_loss_bwd")
    return ws["out"], dpreds


torch.library.impl(_LIBDEF, "yolox_loss", "CUDA")(_yolox_loss_cuda)


class _YoloxLossFn(torch.autograd.Function):
    """losses[0..3] = (total, 5*iou, obj, cls); the gradient of ANY weighting of the four follows from d(total)/d(raw) =
    d(5 iou)/d + d(obj)/d + d(cls)/d only when the weights are equal, so the per-loss gradients are taken from the kernel
    by unit weights one loss at a time when the incoming gradient is not uniform."""

    @staticmethod
    def forward(ctx, raw, labels, anchors, num_classes):
        out, dsum = torch.ops.mi355.yolox_loss(raw, labels, anchors, num_classes)
        ctx.save_for_backward(dsum)
        return out[:6].clone()

    @staticmethod
    def backward(ctx, g):
        (dsum,) = ctx.saved_tensors
        w = g[:4]
        if not bool(torch.all(w == w[0])):
            raise L.MI355Error("mi355.yolox_loss: weight the four losses equally (detectron2 sums the loss dict); "
                               "other weightings go through the whole-step plan (loss weights `gw`)")
        return dsum * w[0], None, None, None


def yolox_loss(raw, labels, anchors, num_classes=80):
    """differentiable wrapper of torch.ops.mi355.yolox_loss: returns the reference's (total, 5*iou, obj, cls, l1,
    num_fg/num_gt) as a [6] tensor attached to autograd through `raw`"""
    return _YoloxLossFn.apply(raw, labels, anchors, num_classes)


# ------------------------------------------------------------------------------------------------ mi355::mha / iou_loss_v6
_LIBDEF.define("mha(Tensor q, Tensor k, Tensor v, Tensor? key_padding_mask, int num_heads) -> Tensor")
_LIBDEF.define("iou_loss_v6(Tensor pred, Tensor target, str iou_type, bool xyxy, float eps) -> Tensor")


def _mha_any(q, k, v, key_padding_mask, num_heads):
    from .modeling.attention import mha_core
    return mha_core(q, k, v, key_padding_mask, num_heads)


def _iou_v6_any(pred, target, iou_type, xyxy, eps):
    from .modeling.iou_loss import _IouLossFn, _TYPES
    return _IouLossFn.apply(pred, target, _TYPES[iou_type.lower()], 1 if xyxy else 0, eps)


# these two are torch.autograd.Functions already (fused forward + saved gradient): registered as composite ops, autograd
# sees through them
torch.library.impl(_LIBDEF, "mha", "CompositeImplicitAutograd")(_mha_any)
torch.library.impl(_LIBDEF, "iou_loss_v6", "CompositeImplicitAutograd")(_iou_v6_any)


# ------------------------------------------------------------------------------------------------ module patching
def _is_base_conv(m):
    conv, bn, act = getattr(m, "conv", None), getattr(m, "bn", None), getattr(m, "act", None)
    return (isinstance(conv, torch.nn.Conv2d) and isinstance(bn, torch.nn.BatchNorm2d) and isinstance(act, torch.nn.SiLU)
            and conv.bias is None and conv.groups == 1 and conv.kernel_size[0] == conv.kernel_size[1]
            and conv.kernel_size[0] in (1, 3) and conv.stride[0] in (1, 2) and conv.out_channels % 8 == 0)


def base_conv_forward(m, x):
    """BaseConv.forward (wrappers.py:82-83) of module `m` (children conv / bn / act) through torch.ops.mi355.conv_bn_silu,
    including nn.BatchNorm2d's train-mode side effects (running statistics with momentum, unbiased running variance,
    num_batches_tracked)"""
    bn = m.bn
    out, _, stats = torch.ops.mi355.conv_bn_silu(x, m.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                  m.conv.stride[0], bn.eps, m.training)
    if m.training and bn.track_running_stats:
        with torch.no_grad():
            n = out.numel() // out.shape[1]
            mean = stats[2]
            var = (1.0 / (stats[3] * stats[3]) - bn.eps).clamp_(min=0.0)        # biased batch variance
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
            bn.running_mean.mul_(1 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1 - mom).add_((var * (n / max(n - 1, 1))).to(bn.running_var.dtype), alpha=mom)
            bn.num_batches_tracked += 1
    return out


def patch_base_convs(module):
    """re-point every BaseConv-shaped sub-module (wrappers.py:60-83: conv -> bn -> SiLU; the reference's own class
    qualifies as it is) to torch.ops.mi355.conv_bn_silu on its own parameters and buffers.  Returns the number of
    patched modules.  state_dict, optimizers and checkpoints are untouched: only `forward` changes."""
    n = 0
    for m in module.modules():
        if _is_base_conv(m):
            m.forward = (lambda x, _m=m: base_conv_forward(_m, x))
            n += 1
    return n
