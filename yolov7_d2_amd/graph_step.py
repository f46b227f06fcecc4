"""One training step of an eager module tree as ONE hipGraph.

The YOLOX path has its own step plan (plan.py: a command list built per input shape, replayed as hipGraphs).  DETR and
SparseInst run module by module through the per-op autograd functions - ~3 000 launches and as many Python calls per
step, host-bound (26 ms of kernels in a 37 ms DETR step).  GraphedTrainStep captures forward + backward + optimizer of such
a model with torch.cuda.graph, once per padded batch shape, and replays it: no per-op Python, no launch gaps.

Contract with the model (yolov7_d2_amd.modeling.detr_meta.Detr implements it):
  batch_key(batched_inputs) -> hashable   what a captured graph is specialised for (batch size, padded image size)
  prepare_batch(batched_inputs, static=None) -> static
        everything that touches the host - image padding, ground truth -> device - into tensors that are REFILLED IN
        PLACE when `static` is passed back (the graph reads the same addresses for every batch)
  forward_prepared(static) -> {name: loss}   device work only: no host value of the batch may enter a launch argument;
        an entry "total" (with autograd) is taken as the objective instead of the sum over loss_keys

Dropout: seeds are launch arguments, i.e. constants of the captured graph.  The library adds a device word to every
dropout seed at run time (mi_dropout_seed_offset); the captured step advances it after the backward, so every replay draws
fresh masks and each backward still recomputes its own forward's.
"""
import torch

from . import _lib as L


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_keys=None, warmup=2, max_graphs=24, bucket_bytes=64 << 20, batch_packs=None,
                 backward_stages=None):
        """optimizer: optim.MultiTensorAdamW (one launch per step; `clip_norm=` for the reference DETR configs' full-model
        gradient clipping, learning-rate changes honoured per replay) or a torch optimizer created with capturable=True (its
        lr must then be a device tensor for a scheduler to have any effect; no clipping; single process only).
        loss_keys: the entries of the loss dict that are summed into the objective (default: model.criterion.weight_dict).
        max_graphs: captured batch shapes kept (least recently used evicted; all captures share ONE memory pool, so the
        count costs instantiated graphs, not activation memory).

        Data parallel (torch.distributed initialised with world_size > 1; train_transformer.py:188-203 and
        train_inseg.py:63-77 reach d2's create_ddp_model for this): rank 0's parameters and buffers are broadcast here; a
        step is then graphs A0 .. Ak (forward + the backward cut into stages, each stage's gradients gathered into the
        optimizer's flat buffer) -> the bucketed all-reduce of that buffer (a few messages of ~bucket_bytes; RCCL over
        xGMI, or gloo in rehearsal) -> graph B (clip + AdamW reading the flat buffer with 1 / world_size).  The collectives
        sit BETWEEN graphs: ranks may capture different padded shapes at different steps without ever disagreeing on the
        sequence of collectives, and the warm-up / capture passes issue none.

        Overlap (what DDP's bucket hooks do for the reference): `model.grad_cut_modules()` names modules with a single
        tensor output, in forward order (the ResNet stages for DETR / SparseInst).  Their outputs are detached during
        the forward, so the backward falls into stages - A0: forward + everything after the last cut, A1: the last cut
        module's backward, ... - each its own graph.  Stage j's slice of the flat buffer is all-reduced (async) as soon
        as Aj is ENQUEUED, so it travels while A(j+1).. compute; only the last stage's message is exposed.  The stage a
        parameter belongs to is found, not declared: whatever gained a .grad during that stage's backward.
        MI_DDP_STAGES=0: one backward graph, all-reduce after it (the round-4 form).
        backward_stages: None = staged exactly when data parallel (and MI_DDP_STAGES is not 0); True = also in a single
        process (what the tests use: the staged graphs must reproduce the one-graph step); False = never."""
        import os
        import torch.distributed as dist
        from .ops import WeightImages
        # the bf16 images of every parameter-backed weight from ONE launch at the start of the step (ops.WeightImages;
        # MI_BATCH_PACK=0: one pack launch per layer call, as the eager modules do)
        if batch_packs is None:
            batch_packs = os.environ.get("MI_BATCH_PACK", "1") == "1"
        self.images = WeightImages(list(model.parameters())) if batch_packs else None
        self.model, self.opt, self.warmup = model, optimizer, max(warmup, 2 if batch_packs else 1)
        self.loss_keys = loss_keys
        self.graphs = {}            # key -> [[graphs A0..Ak], graph B or None, static, out, optimizer table handles]; insertion order = recency
        self.max_graphs = max_graphs
        self.pool = None
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.cut_modules = []
        self.stage_params = None    # [[parameter indices (optimizer order)] per backward stage], found at the first pass
        self.stage_buckets = None   # [[(lo, hi) element ranges of the flat buffer] per stage]
        self.bucket_bytes = int(bucket_bytes)
        env = os.environ.get("MI_DDP_STAGES", "1")          # 0: never, 1: when data parallel, 2: always (measures what the cuts cost)
        want = ((self.world > 1 and env != "0") or env == "2") if backward_stages is None else bool(backward_stages)
        cuts = getattr(model, "grad_cut_modules", None)
        if want and cuts is not None:
            self.cut_modules = list(cuts())
        if self.world > 1:
            if not hasattr(optimizer, "enable_flat_grads"):
                raise L.MI355Error("GraphedTrainStep: data parallel needs optim.MultiTensorAdamW (flat gradient buckets)")
            optimizer.enable_flat_grads(bucket_bytes)
            with torch.no_grad():       # DDP's construction-time _sync_module_states
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, 0)
        self.seed_word = torch.zeros(1, dtype=torch.int64, device=model.device)
        L.check(L.lib().mi_dropout_seed_offset(self.seed_word.data_ptr()), "mi_dropout_seed_offset")

    def _drop(self, ent):
        """a capture leaves the cache: its optimizer pointer tables go with it (optim.MultiTensorAdamW.release_capture)"""
        if hasattr(self.opt, "release_capture"):
            for h in (ent[4] if len(ent) > 4 else []):
                self.opt.release_capture(h)

    def close(self):
        L.check(L.lib().mi_dropout_seed_offset(None), "mi_dropout_seed_offset")
        for ent in self.graphs.values():
            self._drop(ent)
        self.graphs.clear()

    def _keys(self, losses):
        if self.loss_keys is not None:
            return self.loss_keys
        wd = getattr(getattr(self.model, "criterion", None), "weight_dict", None)
        return [k for k in losses if wd is None or k in wd]

    def _stage_fns(self, static):
        """the step's forward + backward as closures, one per backward stage (run in order; each may be captured as its
        own graph).  Stage 0 = forward + the backward down to the last cut, returns the loss dict; stage j = the backward
        of the j-th cut module from the end.  No cut modules: one entry."""
        from .ops import WeightImages
        cuts = []           # (output of a cut module as autograd produced it, the detached leaf its consumers saw)
        seen, found = set(), []
        plist = [p for g in self.opt.param_groups for p in g["params"]]

        def detach_output(_m, _inp, out):
            if not (torch.is_tensor(out) and out.requires_grad):       # (frozen prefix: nothing flows back through it)
                return None
            leaf = out.detach().requires_grad_(True)
            cuts.append((out, leaf))
            return leaf

        marks = {}          # parameter index -> (address, version) of its .grad when its stage handed it over

        def finish(last):
            # a parameter belongs to the FIRST stage that leaves it a gradient, and that stage's slice of the flat buffer
            # goes on the wire at once: a later stage that adds to the same .grad (a weight shared across a cut) would
            # train on a partial gradient without any error - refuse it instead
            for k, mk in marks.items():
                g = plist[k].grad
                if g is None or (g.data_ptr(), g._version) != mk:
                    raise L.MI355Error(f"GraphedTrainStep: parameter #{k} ({tuple(plist[k].shape)}) received gradient in two "
                                       "backward stages (a weight shared across a grad_cut_modules() boundary): its first "
                                       "stage's all-reduce would carry a partial sum; remove that cut")
            idx = [k for k, p in enumerate(plist) if p.grad is not None and k not in seen]
            seen.update(idx)
            found.append(idx)
            for k in idx:
                marks[k] = (plist[k].grad.data_ptr(), plist[k].grad._version)
            if last:
                missing = [k for k, p in enumerate(plist) if p.requires_grad and k not in seen]
                if missing:
                    # torch's optimizers skip parameters without a gradient; the one-launch AdamW reads every slot of its
                    # table (weight decay and moment decay would run on a stale or zero slot): refuse rather than differ
                    raise L.MI355Error(f"GraphedTrainStep: {len(missing)} optimizer parameter(s) received no gradient in any "
                                       f"backward stage (first: #{missing[0]}, shape {tuple(plist[missing[0]].shape)}); freeze "
                                       "them (requires_grad_(False)) or leave them out of the optimizer")
            if self.world > 1:
                self.opt.gather_grads(only=None if (last and len(found) == 1) else idx)
            if last:
                WeightImages.active = None
                if self.images is not None:
                    self.images.freeze()        # (after the first pass: the recorded jobs become the step's one pack launch)
                self._note_stages(found)
                self.seed_word += 1
                cuts.clear()

        def abort():
            WeightImages.active = None
            cuts.clear()

        def stage0():
            if self.images is not None:
                self.images.run()
            WeightImages.active = self.images
            try:
                hooks = [m.register_forward_hook(detach_output) for m in self.cut_modules]
                try:
                    losses = self.model.forward_prepared(static)
                finally:
                    for h in hooks:
                        h.remove()
                if len(cuts) != len(nstage) - 1:
                    raise L.MI355Error("GraphedTrainStep: grad_cut_modules() lists a module whose output carries no gradient")
                total = losses["total"] if "total" in losses else sum(losses[k] for k in self._keys(losses))
                self.opt.zero_grad(set_to_none=True)
                total.backward()
                finish(last=not cuts)
            except BaseException:
                abort()
                raise
            out = {k: v.detach() for k, v in losses.items()}
            out["total"] = total.detach()
            return out

        def later(j):
            def run():
                try:
                    out, leaf = cuts[len(cuts) - j]         # stage 1 = the LAST cut module's backward
                    g, leaf.grad = leaf.grad, None
                    if g is None:
                        raise L.MI355Error("GraphedTrainStep: a cut module's output received no gradient")
                    out.backward(g)
                    finish(last=j == len(cuts))
                except BaseException:
                    abort()
                    raise
            return run

        nstage = [stage0] + [later(j) for j in range(1, len(self._live_cut_modules()) + 1)]
        return nstage

    def _live_cut_modules(self):
        """the cut modules that have a trainable parameter at or before them in forward order would be the exact rule;
        what is checked here is cheaper and enough for the backbones in use: a cut module with no trainable parameter
        of its own AND none in any earlier cut module carries no gradient (d2's FREEZE_AT prefix)"""
        live, any_before = [], False
        for m in self.cut_modules:
            any_before = any_before or any(p.requires_grad for p in m.parameters())
            if any_before:
                live.append(m)
        self.cut_modules = live
        return live

    def _note_stages(self, found):
        """the parameters per backward stage (same for every batch shape and every rank: a property of the module tree)
        and, under data parallel, each stage's all-reduce messages: contiguous runs of its tensors in the flat buffer,
        cut at ~bucket_bytes"""
        if self.stage_params is not None:
            if self.stage_params != found:
                raise L.MI355Error("GraphedTrainStep: the backward stages changed between passes")
            return
        self.stage_params = [list(f) for f in found]
        if self.world == 1:
            return
        offs, n = [int(v) for v in self.opt.flat_off.tolist()], self.opt.flat.numel()
        end = lambda k: offs[k + 1] if k + 1 < len(offs) else n
        per = max(1, self.bucket_bytes // 4)
        self.stage_buckets = []
        for idx in self.stage_params:
            runs, b = [], []
            for k in sorted(idx):
                if runs and runs[-1][1] == offs[k]:
                    runs[-1][1] = end(k)
                else:
                    runs.append([offs[k], end(k)])
            for lo, hi in runs:
                while hi - lo > per + per // 2:
                    b.append((lo, lo + per))
                    lo += per
                b.append((lo, hi))
            self.stage_buckets.append(b)

    def _body_opt(self):
        if self.world > 1:
            self.opt.step(grad_scale=1.0 / self.world, from_flat=True)
        else:
            self.opt.step()

    def _body(self, static):
        fns = self._stage_fns(static)
        out = fns[0]()
        for f in fns[1:]:
            f()
        self._body_opt()
        return out

    def _allreduce(self, stage=None, async_op=False):
        """stage None: every stage's messages, blocking (tests take the step apart with it); stage j: that backward
        stage's messages, as work handles when async_op"""
        import torch.distributed as dist
        if stage is None:           # (the SAME message sequence as replay_backward: a rank taking its step apart and a
            for b in self.stage_buckets:        # rank running it whole must agree on the collectives)
                for lo, hi in b:
                    dist.all_reduce(self.opt.flat[lo:hi])
            return []
        return [dist.all_reduce(self.opt.flat[lo:hi], async_op=async_op) for lo, hi in self.stage_buckets[stage]]

    def replay_backward(self, ent, reduce=True):
        """graphs A0 .. Ak of a captured step.  reduce: stage j's all-reduce is issued right after Aj is enqueued and is
        waited for (by the stream, not the host, under RCCL) only before graph B: it overlaps the later stages' backward"""
        works = []
        for j, g in enumerate(ent[0]):
            g.replay()
            if reduce and self.world > 1:
                works += self._allreduce(stage=j, async_op=True)
        for w in works:
            if w is not None:
                w.wait()

    def _snapshot(self):
        st = [p.detach().clone() for g in self.opt.param_groups for p in g["params"]]
        # module buffers too (train-mode BatchNorm running statistics / num_batches_tracked of a model that has them: the
        # warm-up passes run the whole training forward and would advance them by `warmup` steps per captured shape)
        self._buf_snap = [(b, b.detach().clone()) for b in self.model.buffers()]
        if hasattr(self.opt, "state_tensors"):       # optim.MultiTensorAdamW
            os_ = [t.clone() for t in self.opt.state_tensors()]
        else:
            os_ = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in s.items()} for p, s in self.opt.state.items()}
        return st, os_, self.seed_word.clone()

    def _restore(self, snap):
        st, os_, sw = snap
        with torch.no_grad():
            for b, v in getattr(self, "_buf_snap", []):
                b.copy_(v)
            self._buf_snap = []
            for p, v in zip((p for g in self.opt.param_groups for p in g["params"]), st):
                p.copy_(v)
            if hasattr(self.opt, "state_tensors"):
                for t, v in zip(self.opt.state_tensors(), os_):
                    t.copy_(v)
                self.seed_word.copy_(sw)
                return
            for p, s in self.opt.state.items():
                old = os_.get(id(p))
                for k, v in s.items():
                    if torch.is_tensor(v):
                        if old is not None and k in old:
                            v.copy_(old[k])
                        else:
                            v.zero_()      # state created by the warm-up steps: back to its initial value
            self.seed_word.copy_(sw)

    def _capture(self, fn):
        from .ops import WgradBatch
        g = torch.cuda.CUDAGraph()
        if hasattr(self.opt, "begin_capture"):
            self.opt.begin_capture()        # a device table of its own for this graph (its pool has its own gradients)
        # (the job tables of the grouped weight gradients: pinned + device slots allocated outside the capture, the device
        #  slots written once when it ends - no replay happens before this function returns)
        with WgradBatch.step_capture():
            if self.pool is None:
                with torch.cuda.graph(g):
                    out = fn()
                self.pool = g.pool()
            else:
                with torch.cuda.graph(g, pool=self.pool):
                    out = fn()
        handle = None
        if hasattr(self.opt, "finish_capture"):
            handle = self.opt.finish_capture()       # the gradient addresses of the graph's pool -> that table
        return g, out, handle

    def __call__(self, batched_inputs):
        """one optimizer step on the batch; returns the loss dict (device scalars of the step just run; valid until the
        next call)"""
        key = self.model.batch_key(batched_inputs)
        if self.images is not None:
            self.images.verify()                # (replays read recorded parameter addresses: they must still be the parameters')
        ent = self.graphs.pop(key, None)
        if ent is None:
            while len(self.graphs) >= self.max_graphs:             # least recently used capture goes, with its tables
                self._drop(self.graphs.pop(next(iter(self.graphs))))
            static = self.model.prepare_batch(batched_inputs)
            # warm-up on a side stream (lazy initialisation inside the library, allocator pools), undone afterwards:
            # the first real step on this batch is the first replay.  No collective in here (see __init__)
            snap = self._snapshot()
            try:                                # (a failed warm-up / capture must not leave its updates behind)
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(self.warmup):
                        self._body(static)
                torch.cuda.current_stream().wait_stream(s)
                if self.world > 1 or self.cut_modules:
                    ga, hs, out = [], [], None
                    for j, f in enumerate(self._stage_fns(static)):     # one graph per backward stage (see __init__)
                        g, o, h = self._capture(f)
                        ga.append(g)
                        hs.append(h)
                        out = o if j == 0 else out
                    gb, _, hb = self._capture(self._body_opt)
                    hs.append(hb)
                else:
                    g, out, ha = self._capture(lambda: self._body(static))
                    ga, gb, hs = [g], None, [ha]
            finally:
                self._restore(snap)
            ent = [ga, gb, static, out, [h for h in hs if h is not None]]
        else:
            self.model.prepare_batch(batched_inputs, static=ent[2])
        self.graphs[key] = ent                                     # (most recent last)
        if hasattr(self.opt, "sync_lr"):
            # an LR scheduler's new param_groups[i]["lr"] -> the tables THIS replay reads (not every table ever captured)
            self.opt.sync_lr(only=ent[4]) if hasattr(self.opt, "release_capture") else self.opt.sync_lr()
        self.replay_backward(ent)
        if ent[1] is not None:
            ent[1].replay()
        return ent[3]
