import os
"""CPU: host logic of the product (plan builder: buffer layout, concat slices, gradient fan-in flags, dgrad parity
classes, weight packing) checked by interpreting the symbolic plan with torch ops (tests/plan_interp.py) against
the fp32 oracle.  Storage is bf16 like the product's, so tolerances are the bf16 ones stated in SURVEY §8c."""
import numpy as np
import pytest
import torch

import yolox_oracle as O
from plan_interp import Interp

import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.modeling.yolox import _PlanState
from yolov7_d2_amd.params import ParamArena


def _model(seed=0):
    cfg = M.yolox_s_cfg(device="cpu")
    model = M.build_model(cfg)
    sd = O.init_state_dict(0.33, 0.5, 80, seed=seed)
    model.load_state_dict(sd)
    model.params = ParamArena(model, "cpu")
    return model, sd


@pytest.mark.parametrize("fuse_bn_bwd", ["0", "1"])
def test_plan_step_matches_oracle(monkeypatch, fuse_bn_bwd):
    # "1": the BatchNorm-backward sums are taken by the data-gradient convs (MI_CONV_BNBWD) wherever the latest writer
    # of a layer's output gradient is such a conv; the remaining layers keep their BN_BWD_REDUCE command
    monkeypatch.setenv("MI_FUSE_BN_BWD", fuse_bn_bwd)
    model, sd = _model()
    B, H, W = 2, 64, 96
    imgs, labels = O.synth_batch(B, H, W, seed=11, max_gt=4)
    ps = _PlanState(model, B, H, W, True, materialize=False)
    b = ps.builder
    ps.image.copy_(imgs)
    ps.labels.copy_(labels)
    # fp32 storage: isolates the host logic (wiring, flags, packing, tap tables) from bf16 storage noise, which
    # the oracle's own bf16 emulation shows to be ~50% on single-step gradients of this random-init network
    # (gradient condition number ~500 w.r.t. per-layer relative perturbations; see DESIGN.md "Precision").
    nfused = sum(c.tag.endswith("+bnred") for c in b.bwd)
    nred = sum(c.op == L.OP["BN_BWD_REDUCE"] for c in b.bwd)
    assert (nfused, nred) == ((71, 21) if fuse_bn_bwd == "1" else (0, 74))
    it = Interp(b, torch.float32)
    it.run(b.prologue + b.fwd)
    out = it.raw(ps.loss["out"]).view(torch.float32)[:8].clone()
    # oracle
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    res = O.train_step_losses(sd, imgs, labels)
    ref = torch.tensor([float(x) for x in res[:4]])
    np.testing.assert_allclose(out[:4].numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    # backward with detectron2's sum-of-dict weights
    it.raw(ps.loss["gw"]).view(torch.float32)[:4] = 1.0
    it.run(b.bwd)
    (res[0] + res[1] + res[2] + res[3]).backward()
    bad = []
    for name, p in model.named_parameters():
        g = model.params.grad_of(p).detach().float()
        r = sd[name].grad
        denom = float(r.norm()) + 1e-6
        rel = float((g - r).norm()) / denom
        if rel > 2e-3:
            bad.append((name, rel, float(r.norm())))
    assert not bad, bad[:10]
    # BN running statistics were updated through the plan
    rm = model.state_dict()["backbone.stem.conv.bn.running_mean"]
    np.testing.assert_allclose(rm.numpy(), sd["backbone.stem.conv.bn.running_mean"].detach().numpy(), rtol=1e-4, atol=1e-4)
    assert int(model.state_dict()["head.stems.2.bn.num_batches_tracked"]) == 1


def test_plan_step_with_l1_loss_matches_oracle():
    """the step plan with head.use_l1 on (what YOLOX.forward builds once update_iter() has passed
    INPUT.MOSAIC_AND_MIXUP.DISABLE_AT_ITER): five losses and every parameter gradient, detectron2 summing the loss dict
    including l1_loss"""
    model, sd = _model(seed=2)
    B, H, W = 2, 64, 96
    imgs, labels = O.synth_batch(B, H, W, seed=9, max_gt=4)
    ps = _PlanState(model, B, H, W, True, materialize=False, use_l1=True)
    b = ps.builder
    ps.image.copy_(imgs)
    ps.labels.copy_(labels)
    it = Interp(b, torch.float32)
    it.run(b.prologue + b.fwd)
    out = it.raw(ps.loss["out"]).view(torch.float32)[:8].clone()
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    res = O.train_step_losses(sd, imgs, labels, use_l1=True)
    assert float(res[4]) > 0.1
    np.testing.assert_allclose(out[:5].numpy(), np.array([float(x) for x in res[:5]]), rtol=1e-4, atol=1e-4)
    it.raw(ps.loss["gw"]).view(torch.float32)[:5] = 1.0
    it.run(b.bwd)
    (res[0] + res[1] + res[2] + res[3] + res[4]).backward()
    bad = []
    for name, p in model.named_parameters():
        g = model.params.grad_of(p).detach().float()
        r = sd[name].grad
        rel = float((g - r).norm()) / (float(r.norm()) + 1e-6)
        if rel > 2e-3:
            bad.append((name, rel))
    assert not bad, bad[:10]


def test_plan_step_depthwise_backbone_matches_oracle():
    """MODEL.DARKNET.DEPTH_WISE True: the backbone's 3x3 convs become DWConv (depthwise 3x3 + pointwise) - built from the
    config key, same state_dict keys as the reference, the plan (DWCONV_FWD / DGRAD / WGRAD commands between the usual
    BatchNorm passes) interpreted on the CPU against the oracle: losses and every parameter gradient"""
    cfg = M.yolox_s_cfg(device="cpu")
    cfg.MODEL.DARKNET.DEPTH_WISE = True
    model = M.build_model(cfg)
    sd = O.init_state_dict(0.33, 0.5, 80, seed=3, depthwise=True)
    model.load_state_dict(sd)          # strict: the DWConv keys (dconv.conv / dconv.bn / pconv.conv / pconv.bn) line up
    model.params = ParamArena(model, "cpu")
    B, H, W = 2, 64, 96
    imgs, labels = O.synth_batch(B, H, W, seed=12, max_gt=4)
    ps = _PlanState(model, B, H, W, True, materialize=False)
    b = ps.builder
    ops = [L.OPS[c.op] for c in b.fwd + b.bwd]
    assert ops.count("DWCONV_FWD") == 12 and ops.count("DWCONV_WGRAD") == 12 and ops.count("DWCONV_DGRAD") == 12   # 4 stage stems + 8 bottlenecks
    ps.image.copy_(imgs)
    ps.labels.copy_(labels)
    it = Interp(b, torch.float32)
    it.run(b.prologue + b.fwd)
    out = it.raw(ps.loss["out"]).view(torch.float32)[:8].clone()
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    res = O.train_step_losses(sd, imgs, labels, depthwise=True)
    np.testing.assert_allclose(out[:4].numpy(), np.array([float(x) for x in res[:4]]), rtol=1e-4, atol=1e-4)
    it.raw(ps.loss["gw"]).view(torch.float32)[:4] = 1.0
    it.run(b.bwd)
    (res[0] + res[1] + res[2] + res[3]).backward()
    bad = []
    for name, p in model.named_parameters():
        g = model.params.grad_of(p).detach().float()
        r = sd[name].grad
        rel = float((g - r).norm()) / (float(r.norm()) + 1e-6)
        if rel > 2e-3:
            bad.append((name, rel))
    assert not bad, bad[:10]
    # the materialised (dry-run) plan knows the depthwise weight gradients' place in the flat gradient arena
    from yolov7_d2_amd.plan import Plan
    from yolov7_d2_amd.parallel import grad_write_ranges
    plan = Plan(b, dry_run=True)
    covered = np.zeros(model.params.total, dtype=bool)
    for rs in grad_write_ranges(plan, model.params.grad):
        for (b0, b1) in rs:
            covered[b0 // 4: b1 // 4] = True
    for name, p, off, cnt in model.params.entries:
        assert covered[off: off + cnt].all(), name


def test_eval_plan_matches_oracle():
    model, sd = _model(seed=3)
    model.eval()
    B, H, W = 1, 64, 64
    imgs, _ = O.synth_batch(B, H, W, seed=5)
    ps = _PlanState(model, B, H, W, False, materialize=False)
    b = ps.builder
    assert not b.bwd
    ps.image.copy_(imgs)
    it = Interp(b)   # bf16 storage, like the product
    it.run(b.prologue + b.fwd)
    got = it.raw(ps.preds_buf).view(torch.float32)[: B * ps.A * 85].view(B, ps.A, 85)
    with torch.no_grad():
        net = O.Net(sd, 0.33, 0.5, 80, training=False)
        raw, hw = net.forward_raw(imgs)
        ref = O.decode_eval(raw, O.make_anchors(hw))
    np.testing.assert_allclose(got[..., :4].numpy(), ref[..., :4].numpy(), rtol=5e-2, atol=0.5)
    np.testing.assert_allclose(got[..., 4:].numpy(), ref[..., 4:].numpy(), rtol=5e-2, atol=5e-3)


def test_gradient_fanin_flags_and_buffers():
    model, _ = _model()
    ps = _PlanState(model, 2, 64, 64, True, materialize=False)
    b = ps.builder
    tags = [c.tag for c in b.bwd]
    # the loss gradient comes first, the stem's weight gradient last; the image never gets a gradient
    assert tags[0] == "bn_acc_zero.bwd" and tags[1] == "loss.bwd" and "backbone.stem.conv" in tags[-1]
    assert not any(t.startswith("backbone.stem.conv.dgrad") for t in tags)
    # stride-2 data gradients are 4 parity-class launches
    assert sum(t.startswith("backbone.dark3.0.dgrad") for t in tags) == 4
    # no NHWC tensor is copied for a concat: there is no COPY command at all
    from yolov7_d2_amd import _lib as L
    assert all(c.op != L.OP["COPY"] for c in b.fwd + b.bwd)
    n_conv = sum(c.op == L.OP["CONV"] for c in b.fwd)
    # 83 convolutions in YOLOX-s; reg_preds + obj_preds of a level run as ONE 5-channel convolution
    # (round 6: their parameters lie back to back in the arena, YOLOXHead.arena_adjacent): 83 - 3
    fused_head = os.environ.get("MI_HEAD_FUSE_REGOBJ", "1") != "0"
    assert n_conv == (80 if fused_head else 83)
    assert (sum("obj_preds" in t for t in tags) == 0) == fused_head and sum(t.startswith("head.reg_preds") for t in tags) > 0


def test_lane_scheduling_invariants():
    """Plan._group_lanes' pure part (plan.schedule_lanes) on the real YOLOX-s step plan: every parallel region (head
    stems, head level x branch chains, CSP conv1 / conv2) is scheduled into issue sets that (1) contain every command
    exactly once, (2) keep each lane's order, (3) never put two writers of one tensor into the same set and keep their
    original order, (4) are homogeneous in op, and (5) actually group: the head's 12 3x3 convs per direction come out
    as sets of 6"""
    from yolov7_d2_amd.plan import schedule_lanes, lane_out_key
    model, _ = _model()
    ps = _PlanState(model, 2, 64, 96, True, materialize=False)
    b = ps.builder
    NOP, CONV = L.OP["NOP"], L.OP["CONV"]
    nregions = 0
    sizes = {"fwd": [], "bwd": []}
    for which, cmds in (("fwd", b.fwd), ("bwd", b.bwd)):
        k = 0
        while k < len(cmds):
            if cmds[k].op == NOP and cmds[k].tag.endswith(".begin"):
                e = k + 1
                while not (cmds[e].op == NOP and cmds[e].tag.endswith(".end")):
                    e += 1
                region = cmds[k + 1: e]
                nregions += 1
                order = {id(c): i for i, c in enumerate(region)}
                sets = schedule_lanes(region)
                flat = [c for s_ in sets for c in s_]
                assert sorted(map(id, flat)) == sorted(map(id, region))                       # (1)
                for ln in {c.lane for c in region}:                                            # (2)
                    seq = [order[id(c)] for c in flat if c.lane == ln]
                    assert seq == sorted(seq)
                last = {}
                for si, s_ in enumerate(sets):
                    assert len({c.op for c in s_}) == 1                                        # (4)
                    keys = [lane_out_key(c) for c in s_]
                    assert len(set(keys)) == len(keys)                                         # (3a)
                    for c, kk in zip(s_, keys):
                        if kk in last:
                            assert last[kk][0] < si and last[kk][1] < order[id(c)]             # (3b)
                        last[kk] = (si, order[id(c)])
                    if s_[0].op == CONV and len(s_[0].desc.taps) == 9 and "head" in s_[0].tag:
                        sizes[which].append(len(s_))
                k = e + 1
            else:
                k += 1
    assert nregions == 2 * (2 + 8)           # head stems + head chains + 8 CSP layers, forward and backward
    assert sizes["fwd"] == [6, 6]             # cls/reg conv 0 and conv 1 of the three levels
    # backward: conv-1 data gradients of the 6 chains together; the conv-0 data gradients write the 3 stem gradients
    # twice (cls + reg branch) -> two ordered sets of 3
    assert sorted(sizes["bwd"]) == [3, 3, 6]


def test_wgrad_split_is_opt_in_and_ordered(monkeypatch):
    """MI_WGRAD_SPLIT (default: only under torch.distributed with world_size > 1): the head + neck weight gradients form
    an early group placed after the last head / neck backward command and before any backbone backward command"""
    model, _ = _model()
    b0 = _PlanState(model, 2, 64, 96, True, materialize=False).builder
    assert b0.wgrad_split is False
    monkeypatch.setenv("MI_WGRAD_SPLIT", "1")
    b1 = _PlanState(model, 2, 64, 96, True, materialize=False).builder
    assert b1.wgrad_split is True and b1.wgrad_early_prefixes == ("head.", "neck.")
    tags = [c.tag for c in b1.bwd]
    wg = [i for i, c in enumerate(b1.bwd) if c.op == L.OP["WGRAD"]]
    early = [i for i in wg if tags[i].startswith(("head.", "neck."))]
    late = [i for i in wg if i not in early]
    # 24 + 24 convs, 35 in the backbone (reg_preds + obj_preds as one convolution per level: 48 - 3)
    assert len(early) == (45 if os.environ.get("MI_HEAD_FUSE_REGOBJ", "1") != "0" else 48) and len(late) == 35
    first_backbone = min(i for i, t in enumerate(tags) if t.startswith("backbone."))
    assert max(early) < first_backbone                            # the early group can be issued before the backbone's backward


def test_plan_step_matches_oracle_yolox_l():
    """the same plan builder on YOLOX-l (depth 1.0, width 1.0: 54.2 M parameters, 9/9/9/3-bottleneck CSP stages, up to
    1024 channels): losses and every parameter gradient of the interpreted plan against the fp32 oracle - the host logic
    is not specialised to the YOLOX-s shapes of the benchmark"""
    cfg = M.yolox_s_cfg(device="cpu")
    cfg.MODEL.YOLO.DEPTH_MUL, cfg.MODEL.YOLO.WIDTH_MUL = 1.0, 1.0
    model = M.build_model(cfg)
    sd = O.init_state_dict(1.0, 1.0, 80, seed=0)
    model.load_state_dict(sd)
    model.params = ParamArena(model, "cpu")
    assert sum(p.numel() for p in model.parameters()) == 54208895
    B, H, W = 2, 128, 128
    imgs, labels = O.synth_batch(B, H, W, seed=11, max_gt=4)
    ps = _PlanState(model, B, H, W, True, materialize=False)
    b = ps.builder
    ps.image.copy_(imgs)
    ps.labels.copy_(labels)
    it = Interp(b, torch.float32)
    it.run(b.prologue + b.fwd)
    out = it.raw(ps.loss["out"]).view(torch.float32)[:4].clone()
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    res = O.train_step_losses(sd, imgs, labels, depth=1.0, width=1.0)
    np.testing.assert_allclose(out.numpy(), np.array([float(x.detach()) for x in res[:4]]), rtol=1e-5, atol=1e-5)
    it.raw(ps.loss["gw"]).view(torch.float32)[:4] = 1.0
    it.run(b.bwd)
    (res[0] + res[1] + res[2] + res[3]).backward()
    bad = []
    for name, p in model.named_parameters():
        g, r = model.params.grad_of(p).detach().float(), sd[name].grad
        rel = float((g - r).norm()) / (float(r.norm()) + 1e-6)
        if rel > 3e-3:
            bad.append((name, rel))
    assert not bad, bad[:8]


@pytest.mark.parametrize("split", ["0", "1"])
def test_gradient_buckets_are_complete_before_their_allreduce(monkeypatch, split):
    """data-parallel schedule on the REAL YOLOX-s backward list (materialised against a host arena: grouped conv /
    BatchNorm / weight-gradient launches, job tables and all): every byte of the flat gradient arena is written by some
    backward command, the buckets tile the arena, and each bucket's all-reduce is issued only after the LAST command that
    writes into it - with the split weight-gradient groups the neck + head bucket goes out before the first backbone
    command"""
    from yolov7_d2_amd.plan import Plan
    from yolov7_d2_amd.parallel import GradReducer, grad_write_ranges, plan_buckets
    monkeypatch.setenv("MI_WGRAD_SPLIT", split)
    model, _ = _model()
    ps = _PlanState(model, 2, 64, 96, True, materialize=False)
    plan = Plan(ps.builder, dry_run=True)
    arena = model.params
    writes = grad_write_ranges(plan, arena.grad)
    n = plan.bwd_cmds[1]
    assert len(writes) == n
    # (1) every parameter's gradient is written by some command
    covered = np.zeros(arena.total, dtype=bool)
    for rs in writes:
        for (b0, b1) in rs:
            covered[b0 // 4: b1 // 4] = True
    for name, p, off, cnt in arena.entries:
        assert covered[off: off + cnt].all(), name
    # (2) buckets (the engine's choice for world > 1) tile the arena and are reduced after their last writer
    bounds = None
    if split == "1":
        bounds = [min(o for (nm, p, o, c) in arena.entries if nm.startswith(("head.", "neck.")))]
    buckets = plan_buckets(arena.total, writes, 3, bounds=bounds)
    assert sorted((lo, hi) for lo, hi, _ in buckets)[0][0] == 0
    assert sum(hi - lo for lo, hi, _ in buckets) == arena.total
    segs = GradReducer(arena.grad, buckets).segments(n)
    assert segs[0][0] == 0 and segs[-1][1] == n and all(a[1] == b_[0] for a, b_ in zip(segs, segs[1:]))
    for (c0, c1, bucket) in segs:
        if bucket is None:
            continue
        lo, hi = bucket
        for k, rs in enumerate(writes):
            if any(b0 < hi * 4 and lo * 4 < b1 for (b0, b1) in rs):
                assert k < c1, (bucket, k, c1, plan.bwd_tags[k])
    tags = plan.bwd_tags
    if split == "1":
        first_backbone = min(i for i, t in enumerate(tags) if t.startswith("backbone."))
        assert len(buckets) == 2 and segs[0][1] <= first_backbone and "wgrad_group.early" in tags
    else:
        assert "wgrad_group.early" not in tags and tags[-1] == "wgrad_group"


def test_grouped_launches_of_the_640_plan(monkeypatch):
    """regression guard for the launch structure of the benchmark-sized plan (dry-run, 640x640): the head's level x branch
    chains and the CSP conv1 / conv2 pairs really come out as grouped launches (a group planner that silently falls back
    to one launch per layer costs ~6 % of the step and no other test would notice), and no grouped conv launch exceeds
    the 80 KB LDS footprint that keeps two blocks per CU"""
    import collections
    import ctypes as C
    from yolov7_d2_amd.plan import Plan
    monkeypatch.setenv("MI_BN_IN_CONSUMER", "0")      # (the launch structure WITHOUT the BatchNorm fold; the fold has its own test below)
    model, _ = _model()
    ps = _PlanState(model, 2, 640, 640, True, materialize=False)
    plan = Plan(ps.builder, dry_run=True)
    # (round 6: reg_preds + obj_preds of a level are ONE convolution - YOLOXHead.arena_adjacent -, so the three obj_preds jobs
    #  and their accumulating data gradients no longer form launches of their own; MI_HEAD_FUSE_REGOBJ=0: round 5's structure)
    fh = os.environ.get("MI_HEAD_FUSE_REGOBJ", "1") != "0"
    want = {"fwd": dict(CONV_GROUP=12 if fh else 13, BN_GROUP=11, CONV=43, BN_ACT_FWD=43),
            "bwd": dict(CONV_GROUP=11 if fh else 12, BN_GROUP=11, BN_BWD_FUSED=43, BN_BWD_REDUCE=0, BN_BWD_APPLY=0, WGRAD_GROUP=1,
                        SPLIT_DPREDS_BATCH=0, SPLIT_DPREDS=0, LOSS_BWD_FUSED=1, LOSS_BWD=0, BIAS_GRADS=0)}
    jobs = {"fwd": [2] * 8 + ([3, 6, 6, 6] if fh else [3, 6, 6, 6, 3]),
            "bwd": ([6, 6, 3, 3, 3] if fh else [6, 3, 6, 3, 3, 3]) + [4] * 6}    # + the parity classes of the six stride-2 data gradients
    for which in ("fwd", "bwd"):
        arr, n = plan.fwd_cmds if which == "fwd" else plan.bwd_cmds
        ops = collections.Counter(L.OPS[arr[k].op] for k in range(n))
        for op, cnt in want[which].items():
            assert ops[op] == cnt, (which, op, ops[op], cnt)
        metas = [C.cast(arr[k].p[0], C.POINTER(L.mi_conv_group)).contents for k in range(n) if L.OPS[arr[k].op] == "CONV_GROUP"]
        assert [m.njobs for m in metas] == jobs[which]
        # CSP conv1 + conv2 pairs read one tensor: they leave as ONE streaming 1x1 launch (KC == -1, csrc/conv1x1_stream.h;
        # persistent blocks with an LDS ring, one or two per CU by design) wherever N*H*W is a multiple of its pixel tile
        stream = [m.njobs for m in metas if m.KC == -1]
        assert stream == ([2] * 6 if which == "fwd" else []), (which, stream)
        # the head's 3x3 128 -> 128 towers (six jobs per launch) run on the weight-stationary kernel (KC == -2): one
        # persistent block per CU by design
        ws = [m.njobs for m in metas if m.KC == -2]
        assert ws == ([6, 6] if which == "fwd" else [6, 3, 3]), (which, ws)
        assert all(m.lds_bytes <= 80 * 1024 for m in metas if m.KC > 0)


@pytest.mark.parametrize("depth,width", [(0.33, 0.375), (0.67, 0.75), (1.33, 1.25)], ids=["tiny", "m", "x"])
def test_plan_step_other_widths(depth, width):
    """YOLOX-tiny / -m / -x: channel counts that are not multiples of 32 (24, 48, 80 ...; C/8 not a power of two) - padded
    pixel strides, zero-padded weight images, real-channel BatchNorm - through the same plan builder, interpreted on
    the CPU in fp32 against the oracle: the four losses and every parameter gradient"""
    cfg = M.yolox_s_cfg(device="cpu")
    cfg.MODEL.YOLO.DEPTH_MUL, cfg.MODEL.YOLO.WIDTH_MUL = depth, width
    model = M.build_model(cfg)
    sd = O.init_state_dict(depth, width, 80, seed=1)
    model.load_state_dict(sd)
    model.params = ParamArena(model, "cpu")
    B, H, W = 2, 64, 96
    imgs, labels = O.synth_batch(B, H, W, seed=11, max_gt=4)
    ps = _PlanState(model, B, H, W, True, materialize=False)
    b = ps.builder
    ps.image.copy_(imgs)
    ps.labels.copy_(labels)
    it = Interp(b, torch.float32)
    it.run(b.prologue + b.fwd)
    out = it.raw(ps.loss["out"]).view(torch.float32)[:8].clone()
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    res = O.train_step_losses(sd, imgs, labels, depth=depth, width=width)
    ref = torch.tensor([float(x) for x in res[:4]])
    np.testing.assert_allclose(out[:4].numpy(), ref.numpy(), rtol=1e-3, atol=1e-4)   # few anchors: bf16 weight images move a loss by ~3e-4
    it.raw(ps.loss["gw"]).view(torch.float32)[:4] = 1.0
    it.run(b.bwd)
    (res[0] + res[1] + res[2] + res[3]).backward()
    bad = []
    for name, p in model.named_parameters():
        g = model.params.grad_of(p).detach().float()
        r = sd[name].grad
        rel = float((g - r).norm()) / (float(r.norm()) + 1e-6)
        if rel > 5e-3:   # (a mis-wired slice / pad channel is an O(1) error; fp32 summation order with 12 samples per
            #  channel in the deepest BatchNorm layers is ~1e-3)
            bad.append((name, rel, float(r.norm())))
    assert not bad, bad[:10]


def test_batchnorm_in_the_consumer_pass_of_the_640_plan(monkeypatch):
    """plan.Plan._defer_bn on the benchmark-sized plan (dry run): MI_BN_IN_CONSUMER=1 drops the BN_ACT_FWD job of every layer
    whose first reader is a forward convolution on the streaming 1x1 / weight-stationary 3x3 kernel (never one with a
    residual, never a reader on the tile kernel), the reader's descriptor then reads the RAW conv output and carries the
    device record; launches of one input share the record and exactly one of them writes the activated tensor; `auto` (the
    default) keeps the measured winners only; the builder's symbolic lists are untouched."""
    import collections
    import ctypes as C
    from yolov7_d2_amd.plan import Plan
    model, _ = _model()
    ps = _PlanState(model, 16, 640, 640, True, materialize=False)
    nfwd = len(ps.builder.fwd)
    counts = {}
    for mode in ("0", "auto", "1"):
        monkeypatch.setenv("MI_BN_IN_CONSUMER", mode)
        plan = Plan(ps.builder, dry_run=True)
        assert len(ps.builder.fwd) == nfwd and sum(c.op == L.OP["BN_ACT_FWD"] for c in ps.builder.fwd) == 74
        arr, n = plan.fwd_cmds
        members = plan.cmd_members["fwd"]
        nbn = sum((len(members[k]) if members[k] else 1) for k in range(n) if L.OPS[arr[k].op] in ("BN_ACT_FWD", "BN_GROUP"))
        assert nbn == 74 - len(plan.deferred_bn)
        counts[mode] = len(plan.deferred_bn)
        assert not any(".conv2.bnact" in t and ".m." in t and t.startswith("backbone.dark2") for t in plan.deferred_bn)   # (shortcut: has a residual)
        xf = collections.defaultdict(list)
        for k in range(n):
            ds = plan.cmd_descs["fwd"][k]
            if L.OPS[arr[k].op] == "CONV":
                ds = [ds]
            elif L.OPS[arr[k].op] != "CONV_GROUP":
                continue
            for d in ds:
                if d.xf:
                    assert d.stats_acc and not d.flags and d.xf_C == d.K8 * 8
                    assert L.lib().mi_conv2d_route(C.byref(d)) in (1, 2)
                    xf[(k, d.xf)].append(d.xf_write)
        assert all(sum(w) == 1 for w in xf.values()), xf
        assert len(xf) == len(plan.deferred_bn)
    assert counts["0"] == 0 and counts["1"] >= 35 and 1 <= counts["auto"] <= 10, counts


def test_wgrad_fixup_plan_is_opt_in(monkeypatch):
    """MI_WG_FIXUP=1 (csrc/conv_wgrad.hip wg_fixup: the split-K reduction inside the grouped weight-gradient launches):
    the same grids and the same split-K workspace, no reduce grid, every group flagged, the table longer by the tile
    counters (256 bytes per output tile, ahead of the job records); the default plan keeps the reduce grid"""
    import ctypes as C
    from yolov7_d2_amd.plan import Plan
    model, _ = _model()
    metas = {}
    monkeypatch.setenv("MI_WG_MULTI", "0")     # (the fix-up form keeps one grid per tile configuration: compare like with like)
    for mode in ("0", "1"):
        monkeypatch.setenv("MI_WG_FIXUP", mode)
        ps = _PlanState(model, 2, 640, 640, True, materialize=False)
        plan = Plan(ps.builder, dry_run=True)
        arr, n = plan.bwd_cmds
        grp = [k for k in range(n) if L.OPS[arr[k].op] == "WGRAD_GROUP"]
        assert len(grp) == 1
        metas[mode] = L.mi_wgrad_group.from_buffer_copy(C.cast(arr[grp[0]].p[0], C.POINTER(L.mi_wgrad_group)).contents)
    monkeypatch.delenv("MI_WG_FIXUP")
    a, b = metas["0"], metas["1"]
    assert a.red_blocks > 0 and b.red_blocks == 0 and b.red9_blocks == 0 and b.nred == 0
    assert a.ngroups == b.ngroups and a.ws_bytes == b.ws_bytes
    for i in range(a.ngroups):
        assert a.g[i].fixup == 0 and b.g[i].fixup == 1
        assert list(a.g[i].cfg) == list(b.g[i].cfg) and a.g[i].nblocks == b.g[i].nblocks and a.g[i].njobs == b.g[i].njobs
        assert b.g[i].lds_bytes >= a.g[i].lds_bytes
    assert b.table_bytes > a.table_bytes and (b.g[0].job_off % 256) == 0 and b.g[0].job_off >= 256


def test_wgrad_fixup_table_layout(monkeypatch):
    """the job table of a MI_WG_FIXUP=1 plan, read back through a mirror of csrc/conv_wgrad.hip's job record: the tile
    counters come first and are zero, every job points at its own 256 bytes per (cout, cin) output tile - inside the counter
    region, overlapping no other job's - and carries the split-K reduction's operands (gradient pointer, channel counts)"""
    import ctypes as C
    from yolov7_d2_amd.plan import PlanBuilder

    class Wg2K(C.Structure):          # (struct Wg2K; a changed record shows up in the size check below)
        _fields_ = ([("x", C.c_void_p), ("dy", C.c_void_p), ("part", C.c_void_p)] +
                    [(n, C.c_int32) for n in ("ldx", "lddy", "N", "H", "W", "outH", "outW", "is_", "TH", "TW", "tilesY", "tilesX",
                                              "ntiles", "tps", "nsplit", "dymin", "dxmin", "haloW", "npixh", "nqx", "stage", "ns")] +
                    [("toff", C.c_int32 * L.MI_MAX_TAPS), ("nco", C.c_int32), ("nci", C.c_int32), ("nrx", C.c_int32),
                     ("xmap", C.c_int32), ("mTW", C.c_uint32), ("mHW", C.c_uint32), ("V", C.c_longlong), ("bpart", C.c_void_p),
                     ("bld", C.c_int32), ("fix", C.c_int32), ("g", C.c_void_p), ("row_scale", C.c_void_p), ("Cout", C.c_int32),
                     ("Cin", C.c_int32), ("accumulate", C.c_int32), ("pad_", C.c_int32), ("ra", C.c_int32), ("rb", C.c_int32),
                         ("cnt_rel", C.c_longlong)])

    model, _ = _model()
    monkeypatch.setenv("MI_WG_FIXUP", "1")
    ps = _PlanState(model, 2, 320, 320, True, materialize=False)
    wg = [c for c in ps.builder.bwd if c.op == L.OP["WGRAD"]]
    descs = (L.mi_wgrad_desc * len(wg))()
    for i, (d, c) in enumerate(zip(descs, wg)):
        C.memmove(C.byref(d), C.byref(PlanBuilder._wgrad_desc(c.desc)), C.sizeof(L.mi_wgrad_desc))
        d.gw = 4096 * (i + 1)                      # (a recognisable gradient address per layer)
    lib, meta = L.lib(), L.mi_wgrad_group()
    L.check(lib.mi_conv2d_wgrad_group_plan(descs, len(wg), None, None, 0, C.byref(meta)), "plan (sizes)")
    host = (C.c_char * int(meta.table_bytes))()
    L.check(lib.mi_conv2d_wgrad_group_plan(descs, len(wg), 1 << 30, host, meta.table_bytes, C.byref(meta)), "plan")
    raw = bytes(host)
    first_job = min(meta.g[i].job_off for i in range(meta.ngroups))
    jobs, ranges = [], []
    for gi in range(meta.ngroups):
        g = meta.g[gi]
        assert g.fixup == 1
        # the record size the library used: the starts array follows the job array (16-byte granules)
        assert (C.sizeof(Wg2K) * g.njobs + 15) // 16 * 16 == g.starts_off - g.job_off, "struct Wg2K changed: update this mirror"
        NT, MI, NJ, WCO, WCI, TP = list(g.cfg)
        assert g.lds_bytes >= 16 + 3 * 3 * (WCO * WCI * 16) * 16
        for j in range(g.njobs):
            k = Wg2K.from_buffer_copy(raw[g.job_off + j * C.sizeof(Wg2K): g.job_off + (j + 1) * C.sizeof(Wg2K)])
            assert k.fix == 1 and k.nsplit >= 1 and k.nco * k.nci >= 1
            lo = g.job_off + k.cnt_rel
            ranges.append((lo, lo + k.nco * k.nci * 256))
            jobs.append(k)
    assert len(jobs) == len(wg)
    ranges.sort()
    assert ranges[0][0] == 0 and ranges[-1][1] <= first_job                       # inside the counter region, ahead of every record
    assert all(a[1] <= b[0] for a, b in zip(ranges[:-1], ranges[1:]))             # no two jobs share a counter line
    assert raw[:ranges[-1][1]] == bytes(ranges[-1][1])                            # uploaded as zeros
    # the reduction's operands travel with the job: every layer's gradient address and channel counts exactly once
    assert sorted(k.g for k in jobs) == [4096 * (i + 1) for i in range(len(wg))]
    by_g = {k.g: k for k in jobs}
    for i, d in enumerate(descs):
        k = by_g[4096 * (i + 1)]
        assert (k.Cout, k.Cin, k.accumulate) == (d.Cout, d.Cin, d.accumulate)
        assert k.nsplit * k.V * 16 < 2 ** 31                                      # the fix-up's 32-bit slab offsets
