"""worker of tests/test_gpu_detr_graph.py::test_graphed_step_data_parallel_two_ranks: one of two gloo ranks that SHARE cuda:0
(1-GPU boxes; RCCL cannot run there).  The data-parallel form of GraphedTrainStep executes on a device: rank-0 broadcast of
the parameters, graphs A0..Ak (forward + the backward in stages cut at the ResNet stages, each with its gradient gather),
the per-stage all-reduce of the flat buffer, graph B (full-model clip + AdamW from the flat buffer with 1 / world), two
alternating padded shapes, different batches per rank.  Checked besides the sums and the update: the staged backward's
flat gradient == an eager un-staged backward's, and the whole step (all-reduces issued between the stage graphs, async)
lands on the same parameters as the step taken apart.  Writes a JSON verdict for the parent."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

if os.environ.get("MI_TEST_STACKS"):                    # a hang shows where: every thread's stack after that many seconds
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ["MI_TEST_STACKS"]), exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
out_path = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("gloo")

from test_gpu_detr_graph import _batch, _model          # noqa: E402  (the small 2 + 2-layer DETR-R50 of the graph tests)
from yolov7_d2_amd.graph_step import GraphedTrainStep   # noqa: E402
from yolov7_d2_amd.optim import MultiTensorAdamW        # noqa: E402

model = _model(0.0)
if rank == 1:                                            # rank 1 starts from OTHER weights: the broadcast must fix that
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01)
params = [p for p in model.parameters() if p.requires_grad]
opt = MultiTensorAdamW(params, lr=1e-4, weight_decay=1e-4, clip_norm=1.0)
step = GraphedTrainStep(model, opt, bucket_bytes=32 << 20)
res = dict(rank=rank, buckets=len(opt.buckets), flat=opt.flat.numel())
small = lambda s: _batch(s, ((256, 320), (224, 288)), (3, 2))
large = lambda s: _batch(s, ((320, 384), (300, 352)), (2, 4))
# ranks see different batches AND, at step 1, different padded shapes (rank 0 captures `large` while rank 1 replays `small`)
seq = [small(10 + rank), (large if rank == 0 else small)(20 + rank), large(30 + rank), small(40 + rank), large(50 + rank)]
ok_sum, ok_upd, worst_upd = True, True, 0.0
ok_stage, worst_stage, ok_whole = True, 0.0, True
kept, trace = {}, []
for it, b in enumerate(seq):
    key = model.batch_key(b)
    if key not in step.graphs:
        step(b)                                          # (capture + first replay: checked from the second visit on)
        continue
    # the same step, taken apart: graph A, the local flat gradient, the all-reduce, graph B
    before = [p.detach().clone() for p in params]
    m0 = [t.clone() for t in opt.exp_avg]; v0 = [t.clone() for t in opt.exp_avg_sq]; cnt = int(opt.step_count)
    ent = step.graphs[key]
    both_apart = it >= 3                                 # (at 1 and 2 one of the ranks is capturing)
    snap = step._snapshot() if both_apart else None
    model.prepare_batch(b, static=ent[2])
    opt.sync_lr()
    # an eager backward of the same batch with no cut anywhere: what the staged graphs must reproduce
    if os.environ.get("DDP_SKIP_EAGER") == "1":
        eager = None
    else:
        losses = model.forward_prepared(ent[2])
        eager = torch.autograd.grad(losses["total"] if "total" in losses else sum(losses[k] for k in step._keys(losses)), params)
    step.replay_backward(ent, reduce=False)
    local = opt.flat.clone()
    badp = []
    for k, (g, p) in enumerate(zip(eager or [], params)):
        off = int(opt.flat_off[k])
        d = float((local[off: off + p.numel()].view_as(p) - g).abs().max()) / (float(g.abs().max()) + 1e-30)
        if not d <= 1e-5:
            badp.append((k, [j for j, st in enumerate(step.stage_params) if k in st][0], d))
        worst_stage = max(worst_stage, d)
    if eager is None:
        losses = g = None
        eager = []
    ok_stage = ok_stage and worst_stage <= 1e-5
    trace.append(dict(it=it, key=list(key), worst_stage=worst_stage, local_finite=bool(torch.isfinite(local).all()),
                      eager_finite=all(bool(torch.isfinite(g).all()) for g in eager), nbad=len(badp), bad=badp[:6]))
    del losses, eager, g        # (an autograd graph left alive would pin the parameters' AccumulateGrad nodes to THIS stream,
                                #  and the next capture's backward would have to synchronise with it)
    step._allreduce()
    summed = opt.flat.clone()
    ent[1].replay()
    torch.cuda.synchronize()
    kept[it] = (local, summed)
    trace[-1].update(summed_finite=bool(torch.isfinite(summed).all()), clip=[float(x) for x in opt.clip_out.tolist()],
                     params_finite=all(bool(torch.isfinite(p).all()) for p in params))
    if both_apart:
        # the same step WHOLE (stage graphs with their all-reduces in flight between them) from the same state
        apart = [p.detach().clone() for p in params]
        step._restore(snap)
        step(b)
        torch.cuda.synchronize()
        ok_whole = ok_whole and all(bool(torch.equal(a, p.detach())) for a, p in zip(apart, params))
    # the update = clip_grad_norm_(mean gradient, 1.0) + AdamW, by torch on copies
    ref = [t.clone().requires_grad_(True) for t in before]
    ro = torch.optim.AdamW(ref, lr=1e-4, weight_decay=1e-4)
    for k, (r, p) in enumerate(zip(ref, params)):
        off = int(opt.flat_off[k])
        r.grad = (summed[off: off + p.numel()] / world).view_as(p).clone()
        ro.state[r] = dict(step=torch.tensor(float(cnt)), exp_avg=m0[k].clone(), exp_avg_sq=v0[k].clone())
    torch.nn.utils.clip_grad_norm_(ref, 1.0)
    ro.step()
    for r, p in zip(ref, params):
        d = float((r.detach() - p.detach()).abs().max())
        worst_upd = max(worst_upd, d)
        ok_upd = ok_upd and d <= 2e-6
torch.cuda.synchronize()
for it in (3, 4):            # the steps both ranks took apart (at 1 and 2 one of them was capturing): sum of the local gradients
    local, summed = kept[it]
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    ok_sum = ok_sum and bool(torch.equal(summed, both[0] + both[1]))
# identical parameters on both ranks after the five steps
flat_p = torch.cat([p.detach().flatten() for p in params])
gathered = [torch.empty_like(flat_p) for _ in range(world)]
dist.all_gather(gathered, flat_p)
res.update(stages=len(step.stage_params), stage_messages=[len(b) for b in step.stage_buckets],
           stage_params=[len(x) for x in step.stage_params], staged_equals_eager=ok_stage, worst_stage_diff=worst_stage,
           whole_equals_apart=ok_whole)
res.update(trace=trace)
res.update(sum_exact=ok_sum, update_ok=ok_upd, worst_update_diff=worst_upd, graphs=len(step.graphs),
           params_equal=bool(torch.equal(gathered[0], gathered[1])), finite=bool(torch.isfinite(flat_p).all()),
           clip=[float(x) for x in opt.clip_out.tolist()], steps=int(opt.step_count))
step.close()
with open(out_path, "w") as f:
    json.dump(res, f)
dist.barrier()
dist.destroy_process_group()
