"""worker of tests/test_gpu_ddp_rehearsal.py: one of two gloo ranks that SHARE cuda:0 (the pool has 1-GPU boxes; RCCL
itself cannot run there).  Everything device-side of NativeTrainer's world > 1 path executes: three staged weight-gradient
groups, three captured backward hipGraph segments, a bucket all-reduce after each on the communication path, 1 / world
folded into the SGD kernel.  Writes a JSON verdict for the parent."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
import yolox_oracle as O
from yolov7_d2_amd.engine import NativeTrainer

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
out_path = sys.argv[1]
torch.cuda.set_device(0)
B, H, W = 4, 160, 192
sd = O.init_state_dict(0.33, 0.5, 80, seed=0)
imgs, labels = O.synth_batch(B, H, W, seed=100 + rank, max_gt=6)
res = dict(rank=rank)


def fresh():
    m = M.build_model(M.yolox_s_cfg(device="cuda"))
    m.load_state_dict(sd)
    return m


# (1) this rank's LOCAL gradient: a world-1 trainer (built before the process group exists), single backward list, lr 0
os.environ["MI_BN_FUSED"] = "0"       # the same BatchNorm backward kernels in both runs (world > 1 defaults to two-pass)
tr1 = NativeTrainer(fresh(), lr=0.0, use_graph=False)
assert tr1.world == 1
st = tr1.load_batch(imgs.cuda(), labels.cuda())
tr1.step(st)
tr1.stream.synchronize()
g_local = tr1.params.grad.detach().cpu().clone()
res["single_segments"] = len(st["segs"])

# (2) the data-parallel trainer: graphs on, lr 0 (parameters and hence gradients stay comparable over the three steps)
dist.init_process_group("gloo", rank=rank, world_size=world)
del os.environ["MI_BN_FUSED"]
os.environ["MI_DDP_OVERLAP"] = "1"    # parts (2) / (3): the overlapped schedule, forced (the default "auto" is part (5))
tr2 = NativeTrainer(fresh(), lr=0.0, use_graph=True)
assert tr2.world == 2
st2 = tr2.load_batch(imgs.cuda(), labels.cuda())
res["bn_fused_under_ddp"] = bool(st2["plan"].bn_fused)
res["segments"] = [[lo, hi, list(b) if b else None] for lo, hi, b in st2["segs"]]
res["wgrad_groups"] = sum(1 for k in range(st2["plan"].bwd_cmds[1]) if L.OPS[st2["plan"].bwd_cmds[0][k].op] == "WGRAD_GROUP")
for it in range(3):                   # eager, capture + replay, replay
    tr2.step(st2)
tr2.stream.synchronize()
res["graphs"] = st2["graphs"] is not None and sum(h is not None for h in st2["graphs"]["bwd"])
g_red = tr2.params.grad.detach().cpu().clone()
both = [torch.zeros_like(g_local) for _ in range(world)]
dist.all_gather(both, g_local)
want = both[0] + both[1]
res["reduced_vs_sum_rel"] = float((g_red - want).norm() / want.norm())
res["reduced_vs_sum_max"] = float((g_red - want).abs().max() / want.abs().max())
res["local_norm"], res["sum_norm"] = float(g_local.norm()), float(want.norm())

# (3) a real update: identical parameters on both ranks afterwards, and they moved
tr3 = NativeTrainer(fresh(), lr=0.01, use_graph=True)
st3 = tr3.load_batch(imgs.cuda(), labels.cuda())
p0 = tr3.params.data.detach().cpu().clone()
for it in range(3):
    tr3.step(st3)
tr3.stream.synchronize()
p = tr3.params.data.detach().cpu().clone()
ps = [torch.zeros_like(p) for _ in range(world)]
dist.all_gather(ps, p)
res["params_equal"] = bool(torch.equal(ps[0], ps[1]))
res["params_moved"] = float((p - p0).norm() / p0.norm())
res["finite"] = bool(torch.isfinite(p).all())
def run_schedule(tag, steps=3, ref=None):
    tr = NativeTrainer(fresh(), lr=0.01, use_graph=True)
    s_ = tr.load_batch(imgs.cuda(), labels.cuda())
    for it in range(steps):
        tr.step(s_)
    tr.stream.synchronize()
    q = tr.params.data.detach().cpu().clone()
    p_overlap = p if ref is None else ref
    qs = [torch.zeros_like(q) for _ in range(world)]
    dist.all_gather(qs, q)
    res[tag + "_params_equal"] = bool(torch.equal(qs[0], qs[1]))
    res[tag + "_finite"] = bool(torch.isfinite(q).all())
    res[tag + "_vs_overlap_rel"] = float((q - p_overlap).norm() / ((p_overlap - p0).norm() + 1e-30))     # relative to the update itself
    res[tag + "_buckets"] = len(s_["red"].buckets)
    res[tag + "_wgrad_groups"] = sum(1 for k in range(s_["plan"].bwd_cmds[1]) if L.OPS[s_["plan"].bwd_cmds[0][k].op] == "WGRAD_GROUP")
    res[tag + "_bn_fused"] = bool(s_["plan"].bn_fused)
    res[tag + "_mode"] = tr.ddp_mode
    res[tag + "_choice"] = tr.ddp_choice
    return tr


# (4) the exposed schedule, forced: the single-GPU backward + ONE all-reduce after it lands on the same parameters.  With
# the SAME BatchNorm backward kernels as the overlapped run (two-pass) the schedules differ only in the split-K summation
# order of the weight gradients (one group instead of three) and three steps stay together; with the BatchNorm form chosen
# on the device (possibly the one-launch kernel: dy differs by bf16 roundings) the comparison is made after ONE step - a
# randomly initialised YOLOX amplifies a rounding into other SimOTA assignments within a few updates
os.environ["MI_DDP_OVERLAP"] = "1"
tr_one = NativeTrainer(fresh(), lr=0.01, use_graph=True)
st_one = tr_one.load_batch(imgs.cuda(), labels.cuda())
tr_one.step(st_one)
tr_one.stream.synchronize()
p_overlap_1 = tr_one.params.data.detach().cpu().clone()
os.environ["MI_DDP_OVERLAP"] = "0"
run_schedule("exposed_bnauto_1step", steps=1, ref=p_overlap_1)
os.environ["MI_BN_FUSED"] = "0"
run_schedule("exposed")
if os.environ.get("MI_REHEARSAL_PROBE") == "1":        # (diagnostic: the exposed schedule with each BatchNorm backward form forced)
    for f in ("0", "1"):
        os.environ["MI_BN_FUSED"] = f
        run_schedule("exposed_bn" + f)
    del os.environ["MI_BN_FUSED"]
# (5) auto: both schedules timed with their collectives at the first capture, the same one kept on both ranks
os.environ["MI_DDP_OVERLAP"] = "auto"
run_schedule("auto")
del os.environ["MI_BN_FUSED"]
dist.barrier()
dist.destroy_process_group()
json.dump(res, open(out_path, "w"))
