"""Native training step driver: forward + loss + backward + gradient all-reduce + fused SGD as
command lists / hipGraphs on one stream, no per-launch Python.

Replaces, for the YOLOX path, what detectron2's SimpleTrainer/AMPTrainer.run_step does around
model(data) (train_det.py:21-50,73-75 -> DefaultTrainer; d2 upstream): sum of the returned loss dict
(Q1: all four values, i.e. 2x total), backward, optimizer step (d2 build_optimizer: SGD momentum 0.9,
weight decay 1e-4, WEIGHT_DECAY_NORM 0 for norm layers).  bf16 needs no GradScaler.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from .parallel import GradReducer, broadcast_params, grad_write_ranges, parallel_regions, plan_buckets


def ddp_schedule(plan, builder, params, world, n_buckets=3, group=None):
    """gradient buckets + the cut points of the backward command list for data parallelism: (GradReducer, segments).
    With the staged weight-gradient groups (PlanBuilder.wgrad_split, on iff world > 1) the buckets are cut where the
    stages end in the flat arena (parameters are laid out backbone.stem .. dark5, neck, head): the neck + head bucket is
    complete - and on the wire - while the backbone's backward runs, the dark4 + dark5 bucket while dark3 .. stem run.
    Each segment is (first cmd, one past last cmd, bucket to all-reduce after it or None)."""
    writes = grad_write_ranges(plan, params.grad)
    bounds = None
    if world > 1 and getattr(builder, "wgrad_split", False):
        bounds = []
        for pre in builder.wgrad_stages:
            offs = [o for (name, p, o, n) in params.entries if name.startswith(tuple(pre))]
            if offs:
                bounds.append(min(offs))
    buckets = plan_buckets(params.total, writes, n_buckets if world > 1 else 1, bounds=bounds or None)
    red = GradReducer(params.grad, buckets, group=group)
    barr, bn = plan.bwd_cmds
    return red, red.segments(bn, parallel_regions(plan))


class NativeTrainer:
    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=1e-4, weight_decay_norm=0.0, n_buckets=3,
                 use_graph=True, loss_weights=(1.0, 1.0, 1.0, 1.0), tune=None, input_u8=False):
        self.model = model
        # input_u8: batches arrive as uint8 [B,3,H,W] (what a data loader produces); the plan's Focus packer converts
        self.input_u8 = bool(input_u8)
        self.copy_stream = None
        self._feed = None
        self.feed_direct = os.environ.get("MI_FEED_DIRECT", "1") != "0"
        self.one_graph = os.environ.get("MI_STEP_ONE_GRAPH", "1") != "0"
        # MI_WGRAD_SIDE=G (experiment, round 6; default off - profiles/r06_wgrad_cumask_ab.txt): the weight gradients as G
        # grouped launches on a SIDE QUEUE beside the backward chain, each issued when the last out-gradient of its layers
        # exists (PlanBuilder.wgrad_async cuts the groups), joined before the optimizer.  MI_WGRAD_CUMASK=n[x] confines
        # the side queue to n compute units (hipExtStreamCreateWithCUMask; plain n: n / 8 CUs of every XCD, "nx": n / 32
        # whole XCDs); MI_MAIN_CUMASK=1 gives the chain's stream the complement.  A captured hipGraph does not carry a
        # stream's CU mask into its branches, so under use_graph the chain pieces and the groups are SEPARATE graphs
        # launched on their own streams with event edges (_side_pieces).
        self.side_groups = int(os.environ.get("MI_WGRAD_SIDE", "0") or 0) if self.world_is_one() else 0
        self.side_stream = None
        self.stream = None
        if self.side_groups > 0:
            self._make_side_queue()      # (MI_WGRAD_ASYNC is set around the plan build only: _state)
        model.train()
        self.params = model.ensure_params()
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # Data parallel: which gradient-exchange schedule (train_det.py:73 -> d2 create_ddp_model).
        #   "overlap": three staged weight-gradient groups, each bucket's all-reduce launched from its cut in the backward
        #              list (ddp_schedule); RCCL's kernels then share the device with the rest of backward, so the plan takes
        #              the two-pass BatchNorm backward (the one-launch form's grid barrier needs every block resident) - the
        #              step itself is ~4 % longer than the single-GPU step, and a resident foreign kernel costs another 3 - 4 %
        #              (profiles/r05_ddp_collective_footprint.txt);
        #   "exposed": the single-GPU backward unchanged (fused BatchNorm backward selected on the device, one weight-gradient
        #              group) and ONE all-reduce of the whole 35.9 MB arena after it: 0.06 - 0.41 ms on the xGMI mesh.
        # Which is faster is a property of the node, not of this code: MI_DDP_OVERLAP=auto (default) times both backward
        # schedules WITH their collectives on the device at the first capture, takes the maximum over ranks of each, and
        # keeps the faster (every rank decides on the same two numbers, so all agree on the collective sequence);
        # MI_DDP_OVERLAP=1 / 0 force one.  bench.py prints the choice and both times in its `ddp` block.
        ov = os.environ.get("MI_DDP_OVERLAP", "auto")
        self.ddp_mode = None if self.world == 1 else {"1": "overlap", "0": "exposed"}.get(ov)     # None: single GPU / undecided
        self.ddp_auto = self.world > 1 and self.ddp_mode is None
        self.ddp_choice = None
        self._bn_capacity_full = L.lib().mi_bn_fused_set_capacity(0) if self.world > 1 else 0
        broadcast_params(self.params.data)
        norm_ids = set()
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                norm_ids.update(id(p) for p in m.parameters())
        self.segs, self.nseg = self.params.build_sgd_segments(lr, weight_decay, weight_decay_norm, norm_ids)
        self.momentum, self.n_buckets, self.use_graph = momentum, n_buckets, use_graph
        self.loss_weights = loss_weights
        self._states = {}
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        # optional on-device selection of the conv tile configurations before the graphs are captured (MI_CONV_TUNE=1).
        # Off by default: on the YOLOX-s step the launcher's cost model is within measurement noise of the tuned result.
        self.tune = (os.environ.get("MI_CONV_TUNE", "0") == "1") if tune is None else bool(tune)
        self.tune_report = None

    @staticmethod
    def world_is_one():
        return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    @staticmethod
    def cu_mask_words(n, whole_xcds=False, total=256, xcds=8, invert=False):
        """the 32-bit mask words for hipExtStreamCreateWithCUMask: logical CU i belongs to XCD i % xcds (the driver deals
        the mask round-robin over the XCDs: tools/cu_mask_probe.py), so the low n bits are n / xcds CUs of every XCD and
        the bits with i % xcds < n / (total / xcds) are whole XCDs"""
        bits = [False] * total
        per = total // xcds
        for i in range(total):
            bits[i] = (i % xcds) < max(1, n // per) if whole_xcds else i < n
        if invert:
            bits = [not b for b in bits]
        words = [0] * (total // 32)
        for i, b in enumerate(bits):
            if b:
                words[i // 32] |= 1 << (i % 32)
        return words

    def _masked_stream(self, words):
        arr = (C.c_uint32 * len(words))(*words)
        out = C.c_void_p()
        L.check(L.lib().mi_stream_create_cu_mask(arr, len(words), C.byref(out)), "stream_create_cu_mask")
        return torch.cuda.ExternalStream(out.value)

    def _make_side_queue(self):
        spec = os.environ.get("MI_WGRAD_CUMASK", "")
        if spec:
            whole = spec.endswith("x")
            n = int(spec.rstrip("x"))
            self.side_stream = self._masked_stream(self.cu_mask_words(n, whole))
            if os.environ.get("MI_MAIN_CUMASK", "0") == "1":
                self.stream = self._masked_stream(self.cu_mask_words(n, whole, invert=True))
        else:
            self.side_stream = torch.cuda.Stream(priority=0)       # torch: 0 = the low priority
        # the executor's own FORK / STREAM / JOIN commands (eager replay, plan-time timing) use the same queue
        L.check(L.lib().mi_aux_stream_set(L.MI_WGRAD_STREAM, self.side_stream.cuda_stream), "aux_stream_set")

    def _side_pieces(self, barr, bn):
        """the backward list cut at the side-queue markers the plan emitted (FORK, STREAM sid, <group>, STREAM 0 ... JOIN):
        [("main", lo, hi) | ("side", lo, hi) | ("join", k, k + 1)]"""
        sid, OP = L.MI_WGRAD_STREAM, L.OP
        out, k, lo = [], 0, 0
        while k < bn:
            c = barr[k]
            if c.op == OP["FORK"] and c.i[0] == sid:
                if not (k + 3 < bn and barr[k + 1].op == OP["STREAM"] and barr[k + 1].i[0] == sid
                        and barr[k + 3].op == OP["STREAM"] and barr[k + 3].i[0] == 0):
                    raise L.MI355Error("side queue: unexpected command pattern after FORK")
                if k > lo:
                    out.append(("main", lo, k))
                out.append(("side", k + 2, k + 3))
                k += 4
                lo = k
            elif c.op == OP["JOIN"] and c.i[0] == sid:
                if k > lo:
                    out.append(("main", lo, k))
                out.append(("join", k, k + 1))
                k += 1
                lo = k
            else:
                if c.op in (OP["FORK"], OP["JOIN"]) or (c.op == OP["STREAM"] and c.i[0] != 0):
                    raise L.MI355Error("side queue: the backward list has other parallel regions (MI_MULTI_STREAM)")
                k += 1
        if bn > lo:
            out.append(("main", lo, bn))
        return out

    def set_lr(self, lr):
        # the lr table is read by the SGD graph on self.stream: update it on that stream (stream order = no race with
        # an in-flight step)
        with torch.cuda.stream(self.stream):
            self.params.set_lr(lr)

    def _ddp_build_env(self, mode):
        """the build-time switches of a data-parallel schedule (PlanBuilder / Plan read them while the plan is built) and
        the resident-block budget of the one-launch BatchNorm backward: an eighth of the device is left to the collective
        while it runs beside backward"""
        full = self._bn_capacity_full
        if mode == "exposed":
            L.lib().mi_bn_fused_set_capacity(0)
            return {"MI_WGRAD_SPLIT": "0", "MI_BN_FUSED": os.environ.get("MI_BN_FUSED", "auto")}
        L.lib().mi_bn_fused_set_capacity(full - full // 8)
        return {}

    def _state(self, B, H, W, mode=None):
        primary = self.ddp_mode or ("overlap" if self.world > 1 else None)    # (undecided: the overlap candidate comes first)
        mode = mode or primary
        key = (B, H, W) if mode == primary else (B, H, W, mode)
        st = self._states.get(key)
        if st is not None:
            return st
        env = self._ddp_build_env(mode) if self.world > 1 else {}
        if self.side_groups > 0:
            env = dict(env, MI_WGRAD_ASYNC=str(self.side_groups))      # PlanBuilder cuts the weight gradients into that many groups
        prev = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            variant = "ddp-exposed" if mode == "exposed" else (f"side{self.side_groups}" if self.side_groups > 0 else "")
            ps = self.model.plan_for(B, H, W, True, input_u8=self.input_u8, variant=variant)
        finally:
            for k, v in prev.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        ps.gw().copy_(torch.tensor(self.loss_weights, dtype=torch.float32))
        plan = ps.plan
        # SGD command
        sgd = (L.mi_cmd * 1)()
        sgd[0].op = L.OP["SGD"]
        sgd[0].p[0], sgd[0].p[1], sgd[0].p[2], sgd[0].p[3] = (self.params.data.data_ptr(), self.params.grad.data_ptr(),
                                                             self.params.mom.data_ptr(), self.segs.data_ptr())
        sgd[0].i[0], sgd[0].i[1] = self.nseg, 0
        sgd[0].f[0], sgd[0].f[1] = self.momentum, 1.0 / self.world
        red, segs = ddp_schedule(plan, ps.builder, self.params, self.world, 1 if mode == "exposed" else self.n_buckets)
        st = dict(ps=ps, plan=plan, sgd=sgd, red=red, segs=segs, graphs=None, ddp_mode=mode)
        self._states[key] = st
        return st

    def _run_backward(self, st):
        """the backward of a step as step() issues it: segments (graphs when captured) with each bucket's all-reduce"""
        lib, sp = L.lib(), L.stream_ptr(self.stream)
        barr, bn = st["plan"].bwd_cmds
        gs = st["graphs"] if self.use_graph else None
        for i, (lo, hi, bucket) in enumerate(st["segs"]):
            if gs:
                if gs["bwd"][i] is not None:
                    L.check(lib.mi_graph_launch(gs["bwd"][i], sp), "launch bwd")
            else:
                self._run_cmds(barr, lo, hi, sp)
            st["red"].reduce_bucket(bucket)
        st["red"].wait()

    def _choose_ddp_schedule(self, st, rounds=3, reps=3):
        """MI_DDP_OVERLAP=auto: `st` is the overlap candidate after its first (eager) step.  Builds the exposed candidate for
        the same batch, runs its forward + backward once, captures both, then times `reps` backward passes WITH their
        all-reduces per round, candidates alternating; each candidate's best round, maximum over ranks, decides.  Only
        buffers every step overwrites are written (activations, gradients) plus the BatchNorm running statistics the extra
        forward advances - saved and restored."""
        lib, sp = L.lib(), L.stream_ptr(self.stream)
        B, H, W = st["ps"].B, st["ps"].H, st["ps"].W
        saved = [(t, t.clone()) for t in getattr(st["ps"].builder, "tune_restore", [])]
        with torch.cuda.stream(self.stream):
            st2 = self._state(B, H, W, mode="exposed")
            st2["ps"].image.copy_(st["ps"].image)
            st2["ps"].labels.copy_(st["ps"].labels)
            farr, fn = st2["plan"].fwd_cmds
            barr, bn = st2["plan"].bwd_cmds
            self._run_cmds(farr, 0, fn, sp)
            self._run_cmds(barr, 0, bn, sp)            # (eager once: kernel attributes, lazy initialisation)
            cands = {"overlap": st, "exposed": st2}
            best = {}
            for mode, s_ in cands.items():
                self._ddp_build_env(mode)              # the BatchNorm barrier budget the captured launches are sized with
                if self.use_graph and s_["graphs"] is None:
                    self._capture(s_)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for r in range(rounds):
                for mode, s_ in cands.items():
                    self._ddp_build_env(mode)
                    self.stream.synchronize()
                    dist.barrier()
                    ev0.record(self.stream)
                    for _ in range(reps):
                        self._run_backward(s_)
                    ev1.record(self.stream)
                    ev1.synchronize()
                    ms = ev0.elapsed_time(ev1) / reps
                    best[mode] = ms if mode not in best else min(best[mode], ms)
            t = torch.tensor([best["overlap"], best["exposed"]], device=self.params.data.device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_ov, t_ex = float(t[0]), float(t[1])
            for tns, v in saved:
                tns.copy_(v)
        mode = "overlap" if t_ov <= t_ex else "exposed"
        self.ddp_choice = dict(mode=mode, selected_on_device=dict(overlap_backward_ms=round(t_ov, 4), exposed_backward_ms=round(t_ex, 4)),
                               rule="max over ranks of the best of %d rounds x %d backward passes incl. all-reduce" % (rounds, reps))
        self.ddp_mode, self.ddp_auto = mode, False
        self._ddp_build_env(mode)
        # states are looked up by shape under the chosen mode from here on
        for k in [k for k in self._states if len(k) == 4]:
            self._states.pop(k)
        if mode == "exposed":
            st2["warm"] = True
            # (the host-feed staging buffers are inputs, not plan state: they move over; the staged command lists, which
            #  quote the other plan's forward list, are rebuilt on demand)
            keep = {k: st[k] for k in ("stage", "stage_k") if k in st}
            st.clear()
            st.update(st2)                             # the caller's handle now IS the chosen state
            st.update(keep)
        self._states[(B, H, W)] = st

    def _run_cmds(self, arr, lo, hi, sp):
        if hi > lo:
            ptr = C.cast(C.byref(arr, lo * C.sizeof(L.mi_cmd)), C.POINTER(L.mi_cmd))
            L.check(L.lib().mi_cmdlist_run(ptr, hi - lo, sp), "cmdlist_run")

    def _capture(self, st):
        lib, s = L.lib(), self.stream
        sp = L.stream_ptr(s)
        plan = st["plan"]
        farr, fn = plan.fwd_cmds
        barr, bn = plan.bwd_cmds
        gs = {"fwd": L.check(lib.mi_graph_capture(farr, fn, sp), "capture fwd"), "bwd": [], "fwd_stage": {}}
        if self.side_groups > 0:
            gs["side"] = []
            for (kind, lo, hi) in self._side_pieces(barr, bn):
                h = None
                if kind != "join":
                    ptr = C.cast(C.byref(barr, lo * C.sizeof(L.mi_cmd)), C.POINTER(L.mi_cmd))
                    on = sp if kind == "main" else L.stream_ptr(self.side_stream)
                    h = L.check(lib.mi_graph_capture(ptr, hi - lo, on), f"capture bwd {kind} piece")
                gs["side"].append((kind, h, torch.cuda.Event()))
        for (lo, hi, bucket) in st["segs"]:
            h = None
            if hi > lo and self.side_groups == 0:
                ptr = C.cast(C.byref(barr, lo * C.sizeof(L.mi_cmd)), C.POINTER(L.mi_cmd))
                h = L.check(lib.mi_graph_capture(ptr, hi - lo, sp), "capture bwd segment")
            gs["bwd"].append(h)
        gs["sgd"] = L.check(lib.mi_graph_capture(st["sgd"], 1, sp), "capture sgd")
        # single GPU: forward + backward + optimizer as ONE graph - the two graph-to-graph boundaries inside the step cost
        # 8 - 14 us of idle device each (profiles/r06_h2d_where_the_gap_was.txt, the gaps after loss_final and the reduce grid)
        gs["step"], gs["step_stage"] = None, {}
        if self.one_graph and self.world == 1 and self.side_groups == 0:
            arr, n = self._whole_step_cmds(st, farr, fn)
            gs["step"] = L.check(lib.mi_graph_capture(arr, n, sp), "capture whole step")
        st["graphs"] = gs

    def _whole_step_cmds(self, st, farr, fn):
        """[forward list `farr`] + backward list + the SGD command as one array (kept alive in the state)"""
        barr, bn = st["plan"].bwd_cmds
        sz = C.sizeof(L.mi_cmd)
        arr = (L.mi_cmd * (fn + bn + 1))()
        C.memmove(arr, farr, fn * sz)
        C.memmove(C.byref(arr, fn * sz), barr, bn * sz)
        C.memmove(C.byref(arr, (fn + bn) * sz), st["sgd"], sz)
        st.setdefault("_step_arrays", []).append(arr)
        return arr, fn + bn + 1

    def load_batch(self, images, labels):
        """images float [B,3,H,W] (0..255, already padded to /32), labels [B,max_boxes,5] — device tensors"""
        B, _, H, W = images.shape
        st = self._state(B, H, W)
        # order the copies after whatever produced the inputs on the caller's stream (no device-wide synchronise)
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            st["ps"].image.copy_(images, non_blocking=True)
            st["ps"].labels.copy_(labels, non_blocking=True)
        # the copies are queued behind the previous step on self.stream while the sources belong to the caller's stream:
        # without this the caching allocator may hand their blocks to the caller's next batch before the copies have run
        for t in (images, labels):
            if t.is_cuda:
                t.record_stream(self.stream)
        return st

    def feed(self, st, images_host, labels_host):
        """asynchronous host -> device copy of the NEXT batch (pinned host tensors: images in the plan's input dtype,
        labels [B,max_boxes,5] float) on a dedicated copy stream into one of two staging buffers, so that the PCIe
        transfer of batch i+1 runs under the compute of batch i.  The following step() waits for the copy on the compute
        stream and replays the forward graph OF THAT STAGING BUFFER: its Focus packer reads the staged image in place and
        its first command copies the 38 KB of labels into the plan's label buffer (which the backward reads too), so the
        only operation outside the graphs is the event wait (MI_FEED_DIRECT=0: the round-5 form, two device copies on the
        compute stream in front of the one forward graph).  Replaces the synchronous .to(device) of yolox.py:96,183."""
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream()
        # two staging buffers PER input shape (multi-scale training feeds several (B, H, W) states)
        if "stage" not in st:
            ps = st["ps"]
            st["stage"] = []
            for k in range(2):
                lab_flat = torch.zeros_like(ps.labels_flat)
                st["stage"].append(dict(img=torch.empty_like(ps.image), lab_flat=lab_flat,
                                        lab=lab_flat[:ps.labels.numel()].view(ps.labels.shape), k=k,
                                        ready=torch.cuda.Event(), free=torch.cuda.Event(), st=st))
            st["stage_k"] = 0
            for b in st["stage"]:
                b["free"].record(self.stream)
        b = st["stage"][st["stage_k"]]
        st["stage_k"] ^= 1
        if tuple(images_host.shape) != tuple(b["img"].shape) or tuple(labels_host.shape) != tuple(b["lab"].shape):
            raise ValueError(f"feed(): batch {tuple(images_host.shape)} / {tuple(labels_host.shape)} does not match the "
                             f"state's input buffers {tuple(b['img'].shape)} / {tuple(b['lab'].shape)}")
        # the step that consumed this buffer must have read it.  Waited for ON THE HOST: the event is one and a half steps
        # old (two buffers), so the call returns at once or - when the host has run ahead - holds it back while the device
        # still has more than a whole step queued.  A device-side wait of the copy stream on an event of the compute
        # stream costs the STEP 0.12 ms (2.3 %) on this runtime, the copy itself nothing: tools/h2d_probe.py,
        # profiles/r06_h2d_where_the_gap_was.txt (e2 against e1 / e4).
        b["free"].synchronize()
        with torch.cuda.stream(self.copy_stream):
            b["img"].copy_(images_host, non_blocking=True)
            b["lab"].copy_(labels_host, non_blocking=True)
            b["ready"].record(self.copy_stream)
        self._feed = b

    def _staged_fwd(self, st, k):
        """the forward command list reading staging buffer k: [COPY staged labels -> plan labels] + the plan's forward
        list with the Focus packer's source replaced.  Returns (array, n); built once per (state, buffer)."""
        key = ("fwd_stage_cmds", k)
        if key in st:
            return st[key]
        ps, b = st["ps"], st["stage"][k]
        farr, fn = st["plan"].fwd_cmds
        arr = (L.mi_cmd * (fn + 1))()
        c = arr[0]
        c.op = L.OP["COPY"]                                   # mi_copy_bf16: npix rows of 8 bf16 = 16-byte pieces
        c.p[0], c.p[1] = b["lab_flat"].data_ptr(), ps.labels_flat.data_ptr()
        c.i[0], c.i[1], c.i[2], c.i[3] = 8, 8, 0, 8
        c.l[0] = ps.labels_flat.numel() // 4
        C.memmove(C.byref(arr, C.sizeof(L.mi_cmd)), farr, fn * C.sizeof(L.mi_cmd))
        img = ps.image.data_ptr()
        hits = [j for j in range(1, fn + 1) if arr[j].op == L.OP["FOCUS"] and arr[j].p[0] == img]
        if len(hits) != 1:
            raise L.MI355Error(f"staged forward: expected one Focus command reading the plan's image, found {len(hits)}")
        arr[hits[0]].p[0] = b["img"].data_ptr()
        st[key] = (arr, fn + 1)
        return st[key]

    def step(self, st):
        """one optimisation step on the batch resident in the plan's input buffers (or on the batch handed to feed()
        since the last step); returns nothing (no host sync)."""
        lib = L.lib()
        plan, red = st["plan"], st["red"]
        with torch.cuda.stream(self.stream):
            sp = L.stream_ptr(self.stream)
            staged = None
            if self._feed is not None:
                b, self._feed = self._feed, None
                if b["st"] is not st:
                    raise ValueError("step(): the batch handed to feed() belongs to another input shape's state")
                self.stream.wait_event(b["ready"])
                if self.feed_direct:
                    staged = b
                else:
                    st["ps"].image.copy_(b["img"], non_blocking=True)
                    st["ps"].labels.copy_(b["lab"], non_blocking=True)
                    b["free"].record(self.stream)
            if self.ddp_auto and st.get("warm"):
                self._choose_ddp_schedule(st)          # (second call of the first shape: both schedules timed, one kept)
                plan, red = st["plan"], st["red"]
            if self.use_graph and st["graphs"] is None:
                # first call runs eagerly (sets kernel attributes, warms allocators), second call captures
                if st.get("warm"):
                    if self.tune:
                        self.tune_report = plan.autotune_convs(stream=self.stream)
                    self._capture(st)
                st["warm"] = True
            gs = st["graphs"] if self.use_graph else None
            farr, fn = plan.fwd_cmds
            barr, bn = plan.bwd_cmds
            if gs and gs.get("step") is not None:
                # the whole step is one graph (of the staging buffer the batch sits in, when it was fed)
                if staged is not None:
                    h = gs["step_stage"].get(staged["k"])
                    if h is None:
                        sarr, sn = self._staged_fwd(st, staged["k"])
                        arr, n = self._whole_step_cmds(st, sarr, sn)
                        h = gs["step_stage"][staged["k"]] = L.check(lib.mi_graph_capture(arr, n, sp), "capture staged whole step")
                        gs["fwd_stage"][staged["k"]] = h
                    L.check(lib.mi_graph_launch(h, sp), "launch staged whole step")
                    staged["free"].record(self.stream)
                else:
                    L.check(lib.mi_graph_launch(gs["step"], sp), "launch whole step")
                return
            if staged is not None:
                sarr, sn = self._staged_fwd(st, staged["k"])
                if gs:
                    h = gs["fwd_stage"].get(staged["k"])
                    if h is None:
                        h = gs["fwd_stage"][staged["k"]] = L.check(lib.mi_graph_capture(sarr, sn, sp), "capture staged fwd")
                    L.check(lib.mi_graph_launch(h, sp), "launch staged fwd")
                else:
                    self._run_cmds(sarr, 0, sn, sp)
                staged["free"].record(self.stream)      # only the forward list reads the staging buffer
            elif gs:
                L.check(lib.mi_graph_launch(gs["fwd"], sp), "launch fwd")
            else:
                self._run_cmds(farr, 0, fn, sp)
            if gs and self.side_groups > 0:
                ssp = L.stream_ptr(self.side_stream)
                for (kind, h, ev) in gs["side"]:
                    if kind == "main":
                        L.check(lib.mi_graph_launch(h, sp), "launch bwd chain piece")
                    elif kind == "side":
                        ev.record(self.stream)
                        self.side_stream.wait_event(ev)
                        L.check(lib.mi_graph_launch(h, ssp), "launch wgrad group (side queue)")
                    else:
                        ev.record(self.side_stream)
                        self.stream.wait_event(ev)
            for i, (lo, hi, bucket) in enumerate(st["segs"]):
                if gs and self.side_groups > 0:
                    continue
                if gs:
                    if gs["bwd"][i] is not None:
                        L.check(lib.mi_graph_launch(gs["bwd"][i], sp), "launch bwd")
                else:
                    self._run_cmds(barr, lo, hi, sp)
                red.reduce_bucket(bucket)
            red.wait()
            if gs:
                L.check(lib.mi_graph_launch(gs["sgd"], sp), "launch sgd")
            else:
                self._run_cmds(st["sgd"], 0, 1, sp)

    def losses(self, st):
        """host copy of (total, 5*iou, obj, cls, l1, num_fg/num_gt, num_fg, num_gt) — synchronises.  Also the point
        where a grid barrier of the one-launch BatchNorm backward that gave up (a block that never became resident) is
        reported: the blocks that gave up poisoned their outputs with NaN, here it becomes an error."""
        self.stream.synchronize()
        st["plan"].check_bn_barriers()
        return st["ps"].loss_out().cpu()
