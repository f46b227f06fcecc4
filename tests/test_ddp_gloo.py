"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (bucket planning from the backward command
list, bucketed all-reduce, parameter broadcast).  One process per rank, rendezvous on 127.0.0.1."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolov7_d2_amd.parallel import GradReducer, broadcast_params, plan_buckets


def test_plan_buckets_orders_by_completion():
    # 3 commands; cmd0 writes the tail of the arena (head grads), cmd2 writes the front (stem grads)
    writes = [[(800 * 4, 1000 * 4)], [(300 * 4, 800 * 4)], [(0, 300 * 4)]]
    b = plan_buckets(1000, writes, 3)
    assert [x[2] for x in b] == sorted(x[2] for x in b)
    last = {(lo, hi): k for lo, hi, k in b}
    tail = [k for (lo, hi), k in last.items() if hi == 1000][0]
    front = [k for (lo, hi), k in last.items() if lo == 0][0]
    assert tail < front and b[-1][0] == 0              # head gradients complete first, the stem's bucket last
    assert sorted((x[0], x[1]) for x in b)[0][0] == 0 and sorted((x[0], x[1]) for x in b)[-1][1] == 1000
    red = GradReducer(torch.zeros(1000), b)
    segs = red.segments(3)
    assert segs[0][0] == 0 and segs[-1][1] == 3
    assert all(s[1] >= s[0] for s in segs)
    covered = sorted((s[2] for s in segs if s[2]), key=lambda t: t[0])
    assert covered[0][0] == 0 and covered[-1][1] == 1000


def test_plan_buckets_with_explicit_bounds():
    """the split weight-gradient groups cut the arena where the early (neck + head) group starts: that bucket completes
    at the early group's command, the backbone bucket at the end"""
    writes = [[(600 * 4, 1000 * 4)], [], [(0, 600 * 4)]]          # cmd0 = early wgrad group, cmd2 = late group
    b = plan_buckets(1000, writes, 3, bounds=[600])
    assert b == [(600, 1000, 0), (0, 600, 2)]
    red = GradReducer(torch.zeros(1000), b)
    assert red.segments(3) == [(0, 1, (600, 1000)), (1, 3, (0, 600))]
    assert plan_buckets(1000, writes, 2, bounds=[0, 1000, 5000]) == plan_buckets(1000, writes, 1)   # degenerate bounds


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1003
    params = torch.full((n,), float(rank + 1))
    broadcast_params(params)
    assert float(params[0]) == 1.0 and float(params[-1]) == 1.0     # rank 0's values everywhere
    grad = torch.arange(n, dtype=torch.float32) * (rank + 1)
    writes = [[(600 * 4, n * 4)], [(0, 600 * 4)]]
    red = GradReducer(grad, plan_buckets(n, writes, 2))
    for (lo, hi, bucket) in red.segments(2):
        red.reduce_bucket(bucket)
    red.wait()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    assert torch.equal(grad, expect)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)


# ------------------------------------------------------------------------------------------------------------------
# The REAL data-parallel schedule of a YOLOX-s step, executed: engine.ddp_schedule() on the materialised backward list
# (staged weight-gradient groups, three buckets, cut points), the command list run by the CPU plan interpreter segment by
# segment with the gloo all-reduce of each bucket issued exactly where NativeTrainer.step issues the RCCL one.
def _members(plan, builder, k):
    """symbolic commands (PlanBuilder._Cmd) behind materialised backward command k"""
    from yolov7_d2_amd import _lib as L
    tag = plan.bwd_tags[k]
    by_tag = {}
    for c in builder.bwd:
        by_tag.setdefault(c.tag, []).append(c)
    if tag.startswith("wgrad_group"):
        idx = 0 if tag == "wgrad_group.early" else (int(tag.split("stage")[1]) if "stage" in tag else len(plan.wgrad_stage_tags) - 1)
        out = []
        for t in plan.wgrad_stage_tags[idx]:
            out += [c for c in by_tag[t] if c.op == L.OP["WGRAD"]]
        return out
    arr, _ = plan.bwd_cmds
    if L.OPS[arr[k].op] in ("FORK", "JOIN", "STREAM", "NOP"):
        return []
    merged = plan.cmd_members["bwd"][k]
    if merged is not None:                       # a merged launch: the builder commands it was made of (merges nest)
        def flat(cs):
            return [x for c in cs for x in (flat(c.members) if getattr(c, "members", None) else [c])]
        return flat(merged)
    return [c for c in by_tag[tag] if c.op != L.OP["WGRAD"]]


def _step_worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p_ in (here, os.path.join(here, "..", "oracle")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import yolox_oracle as O
    from plan_interp import Interp
    import yolov7_d2_amd as M
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.engine import ddp_schedule
    from yolov7_d2_amd.modeling.yolox import _PlanState
    from yolov7_d2_amd.params import ParamArena
    from yolov7_d2_amd.plan import Plan

    def fresh(seed):
        cfg = M.yolox_s_cfg(device="cpu")
        model = M.build_model(cfg)
        model.load_state_dict(O.init_state_dict(0.33, 0.5, 80, seed=seed))
        model.params = ParamArena(model, "cpu")
        return model

    B, H, W = 2, 64, 96
    imgs, labels = O.synth_batch(B, H, W, seed=100 + rank, max_gt=4)      # a different batch per rank

    def scheduled_step(exposed):
        """the overlapped schedule (staged weight-gradient groups, three buckets cut into the backward list) or the exposed
        one (engine.NativeTrainer, MI_DDP_OVERLAP=0: the single-GPU backward list, ONE all-reduce after its last writer)"""
        model = fresh(seed=rank)                     # different weights per rank ...
        broadcast_params(model.params.data)          # ... until rank 0's are broadcast (DDP construction semantics)
        if exposed:
            os.environ["MI_WGRAD_SPLIT"] = "0"       # what NativeTrainer._ddp_build_env sets while the plan is built
        try:
            ps = _PlanState(model, B, H, W, True, materialize=False)
        finally:
            os.environ.pop("MI_WGRAD_SPLIT", None)
        b = ps.builder
        assert b.wgrad_split == (not exposed)        # on by itself under torch.distributed with world_size > 1
        plan = Plan(b, dry_run=True)
        red, segs = ddp_schedule(plan, b, model.params, world, 1 if exposed else 3)
        nb = 1 if exposed else 3
        assert len(red.buckets) == nb and len([s for s in segs if s[2] is not None]) == nb
        ps.image.copy_(imgs); ps.labels.copy_(labels)
        it = Interp(b, torch.float32)
        it.run(b.prologue + b.fwd)
        it.raw(ps.loss["gw"]).view(torch.float32)[:4] = 1.0
        ran = set()
        for (lo, hi, bucket) in segs:                # exactly NativeTrainer.step's loop
            for k in range(lo, hi):
                for c in _members(plan, b, k):
                    assert id(c) not in ran
                    ran.add(id(c))
                    it.run([c])
            red.reduce_bucket(bucket)
        red.wait()
        assert ran == {id(c) for c in b.bwd if L.OPS[c.op] != "NOP"}   # every backward command ran exactly once
        return model, plan, segs

    model_x, plan_x, segs_x = scheduled_step(exposed=True)
    assert [plan_x.bwd_tags[s[1] - 1] for s in segs_x if s[2]] == ["wgrad_group"]      # the one bucket leaves after the last writer
    G_exposed = model_x.params.grad.clone()
    model, plan, segs = scheduled_step(exposed=False)
    G = model.params.grad.clone()                # sum over ranks of the local gradients
    schedules_agree = bool(torch.equal(G, G_exposed))      # same commands, same two-rank sums: bit for bit

    # reference: every rank's LOCAL gradient from a plain (un-partitioned) run of the same plan, summed
    m2 = fresh(seed=0)                           # = the broadcast weights
    ps2 = _PlanState(m2, B, H, W, True, materialize=False)
    ps2.image.copy_(imgs); ps2.labels.copy_(labels)
    it2 = Interp(ps2.builder, torch.float32)
    it2.run(ps2.builder.prologue + ps2.builder.fwd)
    it2.raw(ps2.loss["gw"]).view(torch.float32)[:4] = 1.0
    it2.run(ps2.builder.bwd)
    local = m2.params.grad.clone()
    parts = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    ref = sum(parts)
    err = float((G - ref).abs().max() / ref.abs().max())
    # the optimizer step (sgd_kernel semantics: d = g / world + wd * p; m = d on the first step; p -= lr * m)
    lr, wd = 0.01, 1e-4
    p0 = model.params.data.clone()
    model.params.data.sub_(lr * (G / world + wd * p0))
    mine = model.params.data.clone()
    allp = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allp, mine)
    same = all(torch.equal(allp[0], a) for a in allp[1:])
    moved = float((mine - p0).abs().max())
    if rank == 0:
        q.put(dict(err=err, same=same, moved=moved, nseg=len(segs), tags=[plan.bwd_tags[s[1] - 1] for s in segs if s[2]],
                   schedules_agree=schedules_agree))
    dist.barrier()
    dist.destroy_process_group()


def test_real_step_schedule_world2_gloo():
    """rank-0 broadcast + one step on DIFFERENT batches through the real bucket / segment schedule: the reduced gradient
    equals the sum of the two ranks' local gradients (every range reduced exactly once, after its last writer - a
    bucket sent too early would miss the late contribution), and both ranks hold identical parameters afterwards"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_step_worker, args=(2, port, q), nprocs=2, join=True)
    r = q.get()
    assert r["err"] < 1e-6, r
    assert r["same"] and r["moved"] > 0, r
    assert r["tags"] == ["wgrad_group.early", "wgrad_group.stage1", "wgrad_group"], r
    assert r["schedules_agree"], "the exposed and the overlapped schedule reduced different gradients"
