"""GPU (-m gpu): the DETR training step (configs[3]: meta_arch/detr.py:136-279 forward + SetCriterion + backward + AdamW)
captured as ONE hipGraph (yolov7_d2_amd/graph_step.py) against the eager module-by-module step: same losses, same
parameters after several optimizer steps; a second batch with other images, other box counts and other image sizes inside
the same padded shape REPLAYS the same graph (masks, targets and 1 / num_boxes live on the device); dropout draws a fresh
mask on every replay (mi_dropout_seed_offset) although seeds are launch constants of the captured graph."""
import copy

import pytest
import torch

import yolov7_d2_amd as M
from yolov7_d2_amd.d2shim import Boxes, Instances
from yolov7_d2_amd.graph_step import GraphedTrainStep
from gen_golden_inputs import seeded_tensor_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batch(seed, sizes, counts):
    g = torch.Generator().manual_seed(seed)
    out = []
    for (h, w), n in zip(sizes, counts):
        wh = 16 + torch.rand(n, 2, generator=g) * 96
        xy = torch.rand(n, 2, generator=g) * (torch.tensor([w, h]) - wh).clamp(min=1)
        inst = Instances((h, w), gt_boxes=Boxes(torch.cat([xy, xy + wh], 1)), gt_classes=torch.randint(0, 80, (n,), generator=g))
        out.append(dict(image=torch.randint(0, 256, (3, h, w), generator=g).float().to(DEV), instances=inst))
    return out


def _model(dropout):
    cfg = M.detr_r50_cfg(device=DEV)
    cfg.MODEL.DETR.ENC_LAYERS, cfg.MODEL.DETR.DEC_LAYERS, cfg.MODEL.DETR.NUM_OBJECT_QUERIES = 2, 2, 30
    cfg.MODEL.DETR.DROPOUT = dropout
    model = M.build_model(cfg)
    sd = model.state_dict()
    model.load_state_dict(seeded_tensor_dict({k: v.shape for k, v in sd.items()}, seed=203), strict=False)
    with torch.no_grad():
        model.detr.input_proj.weight.mul_(1e-3)     # (encoder tokens of trained magnitude: see test_gpu_detr_meta real-size test)
    model.train()
    return model


def _opt(model, lr=1e-4, kind="torch"):
    params = [p for p in model.parameters() if p.requires_grad]
    if kind == "multi":
        from yolov7_d2_amd.optim import MultiTensorAdamW
        return MultiTensorAdamW(params, lr=lr, weight_decay=1e-4)
    return torch.optim.AdamW(params, lr=lr, weight_decay=1e-4, capturable=True)


def test_multi_tensor_adamw_equals_torch_adamw():
    """optim.MultiTensorAdamW (one launch over separately allocated tensors, pointers and update count in device tables)
    against torch.optim.AdamW: five steps with two parameter groups"""
    from yolov7_d2_amd.optim import MultiTensorAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 256), (2048,), (64, 3, 7, 7), (1,), (300, 33)]
    a = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    groups = lambda ps: [dict(params=ps[:2], lr=1e-3, weight_decay=1e-2), dict(params=ps[2:], lr=1e-4, weight_decay=0.0)]
    oa, ob = torch.optim.AdamW(groups(a), betas=(0.9, 0.999), eps=1e-8), MultiTensorAdamW(groups(b), betas=(0.9, 0.999), eps=1e-8)
    for it in range(5):
        for x, y in zip(a, b):
            gr = torch.randn(x.shape, generator=g).to(DEV)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        torch.testing.assert_close(y.detach(), x.detach(), rtol=1e-5, atol=1e-6)
    assert int(ob.step_count) == 5


@pytest.mark.parametrize("kind", ["torch", "multi"])
def test_graphed_step_equals_eager_step_and_serves_other_batches(kind):
    batches = [_batch(1, ((256, 320), (224, 288)), (3, 2)),
               _batch(2, ((256, 320), (256, 256)), (1, 5)),      # other sizes inside the same padded shape, other box counts
               _batch(3, ((240, 320), (256, 300)), (4, 4))]
    eager, graphed = _model(0.0), _model(0.0)
    # the eager side uses the SAME optimizer implementation: the two AdamW kernels agree to 1 ulp per update
    # (test_multi_tensor_adamw_equals_torch_adamw), and one ulp in a weight is enough to flip a Hungarian assignment of this
    # randomly initialised model at the next step - a comparison across implementations would measure that, not the graph
    oe, og = _opt(eager, kind=kind), _opt(graphed, kind=kind)
    step = GraphedTrainStep(graphed, og)
    try:
        for it, b in enumerate(batches):
            losses = eager(b)
            total = sum(v for k, v in losses.items() if k in eager.criterion.weight_dict)
            oe.zero_grad(set_to_none=True)
            total.backward()
            oe.step()
            out = step(b)
            assert len(step.graphs) == 1                      # one capture serves all three batches
            dev_ = {k: float((out[k].float() - v.detach().float()).abs() / (v.detach().float().abs() + 1e-3)) for k, v in losses.items()}
            print(f"step {it} [{kind}] worst loss deviation", max(dev_, key=dev_.get), max(dev_.values()))
            for k, v in losses.items():
                torch.testing.assert_close(out[k].float(), v.detach().float(), rtol=2e-3, atol=2e-3, msg=f"step {it} {k}")
        torch.cuda.synchronize()
        worst = 0.0
        for (n, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
            if p.requires_grad:
                d = float((p.detach() - q.detach()).abs().max())
                worst = max(worst, d)
                # three AdamW steps of lr 1e-4 move a weight by <= 3e-4; both paths must have moved it the same way
                assert d <= 1.5e-4, (n, d)
        assert int(step.seed_word) == len(batches)
    finally:
        step.close()


def test_graphed_step_draws_fresh_dropout_masks_per_replay():
    model = _model(0.1)
    opt = _opt(model, lr=0.0)                                   # parameters stay put: only the masks can change the loss
    step = GraphedTrainStep(model, opt)
    try:
        b = _batch(5, ((256, 320), (224, 288)), (3, 2))
        a1 = float(step(b)["total"])
        a2 = float(step(b)["total"])
        a3 = float(step(b)["total"])
        assert len({a1, a2, a3}) == 3, (a1, a2, a3)             # a captured seed constant alone would repeat the same mask
        assert abs(a1 - a2) < 0.2 * abs(a1)
        step.seed_word.fill_(0)                                  # the word back at 0: the first replay's mask again
        assert float(step(b)["total"]) == a1
    finally:
        step.close()


def test_multi_tensor_adamw_clip_equals_torch_clip_then_adamw():
    """FullModelGradientClippingOptimizer.step (yolov7/optimizer/build.py:206-223): torch.nn.utils.clip_grad_norm_ over ALL
    parameters, then AdamW - against MultiTensorAdamW(clip_norm=): norm and coefficient on the device, gradients scaled
    inside the update kernel.  Steps with the norm above and below the threshold."""
    from yolov7_d2_amd.optim import MultiTensorAdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(256, 256), (2048,), (64, 3, 7, 7), (1,), (300, 33), (40000,)]
    a = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    groups = lambda ps: [dict(params=ps[:2], lr=1e-3, weight_decay=1e-2), dict(params=ps[2:], lr=1e-4, weight_decay=0.0)]
    oa = torch.optim.AdamW(groups(a))
    ob = MultiTensorAdamW(groups(b), clip_norm=0.1)
    for it, mag in enumerate([1.0, 1e-5, 3.0, 1e-4, 0.5]):          # norms ~ 4e2 .. 4e-3 around the 0.1 threshold
        for x, y in zip(a, b):
            gr = torch.randn(x.shape, generator=g).to(DEV) * mag
            x.grad, y.grad = gr.clone(), gr.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_(a, 0.1)
        oa.step()
        ob.step()
        coef, norm = ob.clip_out.tolist()
        assert abs(norm - float(ref_norm)) <= 1e-5 * float(ref_norm), (it, norm, float(ref_norm))
        assert abs(coef - min(1.0, 0.1 / (float(ref_norm) + 1e-6))) <= 1e-5 * coef
    for x, y in zip(a, b):
        torch.testing.assert_close(y.detach(), x.detach(), rtol=2e-5, atol=2e-6)


def test_two_captured_shapes_alternate_and_lr_changes_reach_the_replays():
    """every captured graph has its own gradient pool, so every capture needs its own pointer table - alternating
    two padded shapes (DETR's multi-scale input does this almost every batch) must keep equalling the eager step; and a
    learning rate written into param_groups after the captures (the reference's LR drop / warm-up schedule) must reach the
    replays of BOTH graphs.  Full-model clipping on, as the DETR YAMLs have it (CLIP_VALUE 0.01 there; 1.0 here so that
    some steps clip and some do not)."""
    from yolov7_d2_amd.optim import MultiTensorAdamW
    small = lambda s: _batch(s, ((256, 320), (224, 288)), (3, 2))
    large = lambda s: _batch(s, ((320, 384), (300, 352)), (2, 4))
    seq = [small(1), large(2), small(3), large(4), small(5), large(6)]
    eager, graphed = _model(0.0), _model(0.0)
    mk = lambda m: MultiTensorAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4, clip_norm=1.0)
    oe, og = mk(eager), mk(graphed)
    step = GraphedTrainStep(graphed, og)
    try:
        for it, b in enumerate(seq):
            if it == 4:                                          # LR drop after both shapes were captured
                for o in (oe, og):
                    for grp in o.param_groups:
                        grp["lr"] = 3e-5
            losses = eager(b)
            total = sum(v for k, v in losses.items() if k in eager.criterion.weight_dict)
            oe.zero_grad(set_to_none=True)
            total.backward()
            oe.step()
            out = step(b)
            for k, v in losses.items():
                torch.testing.assert_close(out[k].float(), v.detach().float(), rtol=2e-3, atol=2e-3, msg=f"step {it} {k}")
            torch.testing.assert_close(og.clip_out, oe.clip_out, rtol=2e-3, atol=1e-6, msg=f"step {it} clip")
        assert len(step.graphs) == 2 and len(og.captures) == 2
        assert og.captures[0][0].data_ptr() != og.captures[1][0].data_ptr()
        torch.cuda.synchronize()
        for (n, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
            if p.requires_grad:
                d = float((p.detach() - q.detach()).abs().max())
                assert d <= 2.5e-4, (n, d)                       # six AdamW steps of lr <= 1e-4 move a weight by <= 6e-4
        # the LR change took effect in the replays: steps 5 and 6 at 3e-5 instead of 1e-4.  A graph that kept its captured
        # lr would leave the two models a full (1e-4 - 3e-5) * 2 apart on most weights - far outside the bound above
    finally:
        step.close()


def test_evicted_captures_release_their_optimizer_tables():
    """the LRU drops a graph -> its optimizer pointer table goes too (a multi-scale DETR run re-captures
    endlessly: tables must not pile up, and an LR change must be uploaded only into the table of the graph about to replay).
    max_graphs = 2, three padded shapes in rotation: never more than two live tables, results still equal the eager step."""
    from yolov7_d2_amd.optim import MultiTensorAdamW
    shapes = [((256, 320), (224, 288)), ((320, 384), (300, 352)), ((192, 256), (160, 224))]
    seq = [_batch(10 + i, shapes[i % 3], (2, 3)) for i in range(7)]
    eager, graphed = _model(0.0), _model(0.0)
    mk = lambda m: MultiTensorAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4, clip_norm=1.0)
    oe, og = mk(eager), mk(graphed)
    step = GraphedTrainStep(graphed, og, max_graphs=2)
    try:
        for it, b in enumerate(seq):
            if it == 5:
                for o in (oe, og):
                    for grp in o.param_groups:
                        grp["lr"] = 5e-5
            losses = eager(b)
            total = sum(v for k, v in losses.items() if k in eager.criterion.weight_dict)
            oe.zero_grad(set_to_none=True)
            total.backward()
            oe.step()
            out = step(b)
            for k, v in losses.items():
                torch.testing.assert_close(out[k].float(), v.detach().float(), rtol=2e-3, atol=2e-3, msg=f"step {it} {k}")
            assert len(step.graphs) <= 2 and len(og.captures) == len(step.graphs), (it, len(step.graphs), len(og.captures))
        torch.cuda.synchronize()
        for (n, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
            if p.requires_grad:
                assert float((p.detach() - q.detach()).abs().max()) <= 3e-4, n
    finally:
        step.close()
    assert og.captures == []


def test_graphed_step_data_parallel_two_ranks(tmp_path):
    """GraphedTrainStep under torch.distributed (train_transformer.py:188-203 / train_inseg.py:63-77 -> d2 create_ddp_model):
    two gloo ranks on cuda:0 (tests/detr_ddp_worker.py).  Rank 1 starts from other weights (the construction-time
    broadcast must overwrite them), the ranks see different batches and at one step different padded shapes (one captures
    while the other replays: the collectives sit between the graphs, so their order never diverges).  Asserted on the
    device: all-reduced flat gradient == sum of the two local ones (bit-exact), the update == clip_grad_norm_(mean
    gradient) + torch AdamW on copies, both ranks hold identical parameters after five steps."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs, outs = [], []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = str(tmp_path / f"rank{r}.json")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "detr_ddp_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    for out in outs:
        r = json.load(open(out))
        assert r["buckets"] >= 2 and r["graphs"] == 2 and r["steps"] == 5, r
        assert r["sum_exact"] and r["update_ok"] and r["params_equal"] and r["finite"], r
        # the backward in stages (transformer + heads, res5, res4, res3), each with messages of its own
        assert r["stages"] == 4 and all(n >= 1 for n in r["stage_messages"]) and all(n > 0 for n in r["stage_params"]), r
        assert r["staged_equals_eager"] and r["whole_equals_apart"], r


def test_shape_buckets_serve_nearby_shapes_with_one_capture():
    """Detr.shape_bucket = 64: batches whose exact padded shapes differ ((250, 310), (241, 300), (256, 320)) share ONE key and
    one capture; against the exact padding (shape_bucket = 1) the losses move only by what the stated consequence allows -
    the largest image's border column / row sees in-tensor padding like a smaller image's - a fraction of a percent"""
    model = _model(0.0)
    bs = [_batch(7, ((250, 310), (224, 288)), (3, 2)), _batch(8, ((241, 300), (230, 260)), (2, 2)), _batch(9, ((256, 320), (200, 300)), (1, 4))]
    assert {model.batch_key(b) for b in bs} == {(2, 256, 320)}
    with torch.no_grad():
        for b in bs[:2]:
            lb = model.forward_prepared(model.prepare_batch(b))
            model.shape_bucket = 1
            assert model.batch_key(b) != (2, 256, 320)
            le = model.forward_prepared(model.prepare_batch(b))
            model.shape_bucket = 64
            for k in le:
                if k in model.criterion.weight_dict:
                    a, e = float(lb[k]), float(le[k])
                    assert abs(a - e) <= 2e-2 * abs(e) + 1e-3, (k, a, e)
    opt = _opt(model, kind="multi")
    step = GraphedTrainStep(model, opt)
    try:
        for b in bs:
            assert bool(torch.isfinite(step(b)["total"]))
        assert len(step.graphs) == 1
    finally:
        step.close()


def test_all_decoder_levels_matched_in_one_launch_equal_the_per_level_matching():
    """HungarianMatcher.match_device_levels ((level, image) pairs as the batch of ONE mi_hungarian_match call over the
    level-replicated ground truth) against match_device level by level: the same assignments, bit for bit, and the same
    weighted losses from SetCriterion.weighted_packed with and without it"""
    model = _model(0.0)
    b = _batch(7, ((256, 320), (224, 288)), (3, 5))
    static = model.prepare_batch(b)
    targets = static["targets"]
    assert targets.lv is not None and targets.lv["n"] == 2
    with torch.no_grad():
        out = model.detr(static["images"])
    levels = [{k: v for k, v in out.items() if k != "aux_outputs"}] + list(out["aux_outputs"])
    matcher = model.criterion.matcher
    one = matcher.match_device_levels(levels, targets)
    for lv, m1 in zip(levels, one):
        m0 = matcher.match_device(lv, targets)
        assert torch.equal(m0["nmatch"], m1["nmatch"])
        for bi, n in enumerate(m0["nmatch"].tolist()):
            assert n == len(b[bi]["instances"])
            assert torch.equal(m0["match_q"][bi, :n], m1["match_q"][bi, :n]) and torch.equal(m0["match_t"][bi, :n], m1["match_t"][bi, :n])
    with torch.no_grad():
        a = model.criterion.weighted_packed(out, targets)
        lv_saved, targets.lv = targets.lv, None          # per-level matching
        c = model.criterion.weighted_packed(out, targets)
        targets.lv = lv_saved
    for k in a:
        assert torch.equal(a[k], c[k]), k
    # a batch without any box: every level matches nothing, the level offsets stay valid
    e = _batch(8, ((256, 320), (224, 288)), (0, 0))
    static = model.prepare_batch(e, static=static)
    with torch.no_grad():
        out = model.detr(static["images"])
    levels = [{k: v for k, v in out.items() if k != "aux_outputs"}] + list(out["aux_outputs"])
    for m in matcher.match_device_levels(levels, static["targets"]):
        assert int(m["nmatch"].abs().sum()) == 0
