"""GPU (-m gpu): YOLOX head loss / SimOTA, NMS / postprocess and the full training + eval step, through the
C-ABI, against the oracle and the golden vectors produced by the reference's own code."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import yolox_oracle as O
from plan_interp import Interp
from yolov7_d2_amd import _lib as L
import yolov7_d2_amd as M
from yolov7_d2_amd.modeling.yolox import _PlanState
from yolov7_d2_amd.params import ParamArena

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _loss_call(raw, labels, anchors, gw=(1, 1, 1, 1), gmax=None, use_l1=False):
    B, A, nch = raw.shape
    ML = labels.shape[1]
    gmax = ML if gmax is None else gmax
    t = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)
    ws = dict(cost=t(B, gmax, A), iou=t(B, gmax, A), match=t(B, gmax, A, dt=torch.uint8), ngt=t(B, dt=torch.int32),
              fg=t(B, A, dt=torch.uint8), matched_gt=t(B, A, dt=torch.int32), matched_iou=t(B, A),
              partial=t(B * ((A + 255) // 256), 4), out=t(8), dpreds=t(B, A, nch),
              partial_l1=t(B * ((A + 255) // 256)), gw=torch.tensor(gw, dtype=torch.float32, device=DEV))
    rd, ld, ad = raw.to(DEV).contiguous(), labels.to(DEV).contiguous(), anchors.to(DEV).contiguous()
    d = L.mi_yolox_loss_desc()
    d.preds, d.labels, d.anchors = rd.data_ptr(), ld.data_ptr(), ad.data_ptr()
    d.B, d.A, d.ncls, d.max_labels, d.gmax = B, A, nch - 5, ML, gmax
    for k in ("cost", "iou", "match", "ngt", "fg", "matched_gt", "matched_iou", "partial", "out"):
        setattr(d, k, ws[k].data_ptr())
    if use_l1:
        d.use_l1, d.partial_l1 = 1, ws["partial_l1"].data_ptr()
    L.check(L.lib().mi_yolox_loss_fwd(C.byref(d), L.stream_ptr()), "loss_fwd")
    L.check(L.lib().mi_yolox_loss_bwd(C.byref(d), ws["gw"].data_ptr(), ws["dpreds"].data_ptr(), L.stream_ptr()), "loss_bwd")
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in ws.items()}


def test_simota_loss_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "simota_160.npz"))
    B, H, W = 3, 160, 160
    _, labels = O.synth_batch(B, H, W, seed=21, max_gt=12, min_gt=6)
    labels[1] = 0.0                      # an image without ground truth
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    raw, anchors = O.synth_raw(B, hw, 22, labels=labels)
    ws = _loss_call(raw, labels, anchors)
    # integer results: bit-exact against the reference's get_assignments
    assert int(ws["ngt"][1]) == 0 and int(ws["fg"][1].sum()) == 0
    for b in (0, 2):
        fg = ws["fg"][b].bool()
        assert np.array_equal(fg.numpy(), g[f"fg{b}"])
        assert np.array_equal(ws["matched_gt"][b][fg].numpy(), g[f"matched_gt{b}"])
        np.testing.assert_allclose(ws["matched_iou"][b][fg].numpy(), g[f"matched_iou{b}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ws["out"][:6].numpy(), g["losses"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ws["dpreds"].numpy(), g["draw"], rtol=2e-4, atol=1e-6)


def test_l1_loss_against_reference_golden(golden_dir):
    """head.use_l1 (get_l1_target + L1 on the raw regression outputs, yolox_head.py:389-448) against the reference head
    run with the switch on: the six returned values and d(total + iou + conf + cls + l1)/d(raw)"""
    g = np.load(os.path.join(golden_dir, "simota_160_l1.npz"))
    B, H, W = 3, 160, 160
    _, labels = O.synth_batch(B, H, W, seed=21, max_gt=12, min_gt=6)
    labels[1] = 0.0
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    raw, anchors = O.synth_raw(B, hw, 22, labels=labels)
    ws = _loss_call(raw, labels, anchors, gw=(1, 1, 1, 1, 1), use_l1=True)
    np.testing.assert_allclose(ws["out"][:6].numpy(), g["losses"], rtol=1e-5, atol=1e-5)
    assert float(ws["out"][4]) > 0.5
    np.testing.assert_allclose(ws["dpreds"].numpy(), g["draw"], rtol=2e-4, atol=1e-6)
    # the switch off leaves the l1 slot at 0 and the total without it
    ws0 = _loss_call(raw, labels, anchors)
    assert float(ws0["out"][4]) == 0.0
    np.testing.assert_allclose(float(ws0["out"][0]) + float(ws["out"][4]), float(ws["out"][0]), rtol=1e-6)


def test_simota_full_size_properties():
    """8400 anchors, B=4, up to 100 labels: invariants that hold at any size + parity with the oracle"""
    B, H, W = 4, 640, 640
    _, labels = O.synth_batch(B, H, W, seed=33, max_gt=40)
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    raw, anchors = O.synth_raw(B, hw, 34, labels=labels)
    ws = _loss_call(raw, labels, anchors, gw=(1, 0, 0, 0))
    res, assigns = O.yolox_losses(raw.clone().requires_grad_(True), labels, anchors, 80, return_assign=True)
    nmis = 0
    for b in range(B):
        fg = ws["fg"][b].bool()
        nmis += int((fg != assigns[b]["fg"]).sum())
        # every foreground anchor is matched to exactly one valid gt; background anchors to none
        assert (ws["matched_gt"][b][fg] >= 0).all() and (ws["matched_gt"][b][~fg] == -1).all()
        assert int(ws["matched_gt"][b].max()) < int(ws["ngt"][b])
    assert nmis == 0
    np.testing.assert_allclose(ws["out"][:4].numpy(), np.array([float(x) for x in res[:4]]), rtol=2e-5)
    assert abs(float(ws["out"][6]) - sum(a["num_fg"] for a in assigns)) == 0


def _forms_equal(raw, labels, anchors, monkeypatch, **kw):
    monkeypatch.setenv("MI_SIMOTA_COMPACT", "0")
    monkeypatch.setenv("MI_SIMOTA_PREFILTER", "0")
    ref = _loss_call(raw, labels, anchors, **kw)
    for c, pf in (("1", "0"), ("0", "1"), ("2", "0"), ("2", "1")):
        monkeypatch.setenv("MI_SIMOTA_COMPACT", c)
        monkeypatch.setenv("MI_SIMOTA_PREFILTER", pf)
        got = _loss_call(raw, labels, anchors, **kw)
        for k in ("cost", "iou", "ngt", "fg", "matched_gt", "matched_iou", "out", "dpreds"):
            assert torch.equal(ref[k], got[k]), (c, pf, k)
    monkeypatch.delenv("MI_SIMOTA_COMPACT")
    monkeypatch.delenv("MI_SIMOTA_PREFILTER")
    return ref


def test_simota_compacted_and_prefiltered_forms_are_bit_identical(monkeypatch):
    """round 6: candidate compaction (1) + class terms across the lanes (2) in the cost kernel and the pre-filtered dynamic-k selection against the round-5 kernels:
    every output word equal - at the bench size, with exact ties (all logits equal: the orders fall back on the anchor index),
    with lists that overflow the LDS capacity (the block-wide rounds take over), with <= 9 candidates, with 2 100 anchors"""
    B, H, W = 16, 640, 640
    _, labels = O.synth_batch(B, H, W, seed=1234, max_gt=20)
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    raw, anchors = O.synth_raw(B, hw, 35, labels=labels)
    ws = _forms_equal(raw, labels, anchors, monkeypatch)
    assert int(ws["fg"].sum()) > 100
    # exact ties everywhere: constant predictions (equal IoU along rows / columns of the grid, equal class terms), one huge
    # box per image so that thousands of anchors are candidates of equal cost: the lists overflow -> fallback path
    raw0 = torch.zeros(2, anchors.shape[0], 85)
    lab0 = torch.zeros(2, 100, 5)
    lab0[0, 0] = torch.tensor([3.0, 320, 320, 600, 600])
    lab0[0, 1] = torch.tensor([7.0, 100, 100, 64, 64])
    lab0[1, 0] = torch.tensor([1.0, 320, 320, 16, 16])
    ws = _forms_equal(raw0, lab0, anchors, monkeypatch)
    assert int(ws["fg"][0].sum()) >= 2 and int(ws["fg"][1].sum()) >= 1
    # few candidates (a 4-pixel box: only its centre-radius anchors) and the small-map instantiation (NV = 9)
    hw2 = [(40, 40), (20, 20), (10, 10)]
    anchors2 = O.make_anchors(hw2)
    _, lab2 = O.synth_batch(3, 320, 320, seed=5, max_gt=8)
    lab2[2] = 0.0
    lab2[2, 0] = torch.tensor([2.0, 7, 9, 4, 4])
    raw2, _ = O.synth_raw(3, hw2, 36, labels=lab2)
    _forms_equal(raw2, lab2, anchors2, monkeypatch, use_l1=True, gw=(1, 1, 1, 1, 1))


def test_loss_edge_cases():
    hw = [(8, 8), (4, 4), (2, 2)]
    anchors = O.make_anchors(hw)
    raw = torch.zeros(2, anchors.shape[0], 85)
    labels = torch.zeros(2, 100, 5)     # no labels at all: only the objectness loss, num_fg clamps to 1
    ws = _loss_call(raw, labels, anchors)
    assert float(ws["out"][6]) == 0 and abs(float(ws["out"][2]) - 2 * 84 * np.log(2.0)) < 1e-3
    assert float(ws["out"][1]) == 0 and float(ws["out"][3]) == 0
    labels[0, 0] = torch.tensor([3.0, 32, 32, 20, 24])
    labels[0, 1] = torch.tensor([0.0, 0, 0, 0, 0])       # class-0 zero box vanishes (Q5)
    labels[0, 2] = torch.tensor([5.0, 40, 40, 10, 10])   # ... and everything after the prefix count is ignored
    ws = _loss_call(raw, labels, anchors)
    assert int(ws["ngt"][0]) == 2 and float(ws["out"][6]) >= 1


def test_nms_and_postprocess_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    for name, n, seed in (("small", 300, 31), ("large", 2500, 32)):   # coordinate-trick and per-class branches
        pred = O.synth_decoded(2, n, seed)
        out = M.postprocess(pred.to(DEV), 80, 0.3, 0.65)
        for b in range(2):
            got = out[b].cpu().numpy()
            assert got.shape == g[f"{name}_out{b}"].shape
            np.testing.assert_array_equal(got[:, 6], g[f"{name}_out{b}"][:, 6])     # integer class ids exact
            np.testing.assert_allclose(got, g[f"{name}_out{b}"], rtol=1e-6, atol=1e-5)


def test_nms_full_size_and_edges():
    gen = torch.Generator().manual_seed(7)
    n = 8400
    ctr = torch.rand(n, 2, generator=gen) * 600
    wh = 10 + torch.rand(n, 2, generator=gen) * 80
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    scores = torch.rand(n, generator=gen)
    idxs = torch.randint(0, 80, (n,), generator=gen).float()
    keep = M.batched_nms(boxes.to(DEV), scores.to(DEV), idxs.to(DEV), 0.65).cpu()
    ref = O.batched_nms(boxes, scores, idxs, 0.65)
    assert torch.equal(keep, ref)
    assert (scores[keep][:-1] >= scores[keep][1:]).all()          # sortedness
    again = M.batched_nms(boxes[keep].to(DEV), scores[keep].to(DEV), idxs[keep].to(DEV), 0.65).cpu()
    assert torch.equal(again, torch.arange(len(keep)))            # idempotence
    assert M.batched_nms(torch.empty(0, 4, device=DEV), torch.empty(0, device=DEV), torch.empty(0, device=DEV), 0.5).numel() == 0
    one = M.batched_nms(boxes[:1].to(DEV), scores[:1].to(DEV), idxs[:1].to(DEV), 0.5).cpu()
    assert one.tolist() == [0]
    for m in (63, 64, 65, 129):                                    # chunk-boundary sizes
        k = M.batched_nms(boxes[:m].to(DEV), scores[:m].to(DEV), idxs[:m].to(DEV) * 0, 0.3).cpu()
        assert torch.equal(k, O.batched_nms(boxes[:m], scores[:m], idxs[:m] * 0, 0.3))


# ------------------------------------------------------------------------------------------ full model
def _gpu_model(seed=0):
    cfg = M.yolox_s_cfg(device=DEV)
    model = M.build_model(cfg)
    sd = O.init_state_dict(0.33, 0.5, 80, seed=seed)
    model.load_state_dict(sd)
    return model, sd


def test_many_classes_loss_and_bias_gradients():
    """MODEL.YOLO.CLASSES = 200 (5 + 200 prediction channels: two 128-channel chunks in the bias-gradient kernel, class
    loops of the loss kernels beyond one batch): losses against the oracle on the head outputs the HIP path produced, the
    prediction convs' bias gradients against the oracle's autograd through the same loss"""
    cfg = M.yolox_s_cfg(device=DEV)
    cfg.MODEL.YOLO.CLASSES = 200
    model = M.build_model(cfg)
    model.load_state_dict(O.init_state_dict(0.33, 0.5, 200, seed=5))
    model.train()
    imgs, labels = O.synth_batch(2, 64, 96, seed=12, max_gt=4)
    labels[..., 0] = (labels[..., 0] * 2.4).floor() * (labels.sum(-1) > 0)       # classes up to 189
    loss_dict = model(_batched_inputs(imgs, labels))
    sum(loss_dict.values()).backward()
    torch.cuda.synchronize()
    ps = model.plan_for(2, 64, 96, True)
    raw = ps.preds().float().cpu().clone().requires_grad_(True)
    chk = O.yolox_losses(raw, labels, ps.anchors.float().cpu(), 200)
    got = np.array([float(loss_dict[k]) for k in ("total_loss", "iou_loss", "conf_loss", "cls_loss")])
    np.testing.assert_allclose(got, np.array([float(x) for x in chk[:4]]), rtol=1e-4, atol=1e-5)
    (chk[0] + chk[1] + chk[2] + chk[3]).backward()
    d = raw.grad                                                                  # [B, A, 205]
    a0 = 0
    for k, (h, w) in enumerate(((8, 12), (4, 6), (2, 3))):
        seg = d[:, a0:a0 + h * w].sum((0, 1))
        a0 += h * w
        for name, sl in (("reg_preds", slice(0, 4)), ("obj_preds", slice(4, 5)), ("cls_preds", slice(5, 205))):
            g = model.params.grad_of(getattr(model.head, name)[k].bias).float().cpu()
            np.testing.assert_allclose(g.numpy(), seg[sl].numpy(), rtol=2e-4, atol=2e-6, err_msg=f"{name}[{k}].bias")


def _batched_inputs(imgs, labels):
    from yolov7_d2_amd.d2shim import Boxes, Instances
    out = []
    for b in range(imgs.shape[0]):
        n = int((labels[b].sum(1) > 0).sum())
        l = labels[b, :n]
        xyxy = torch.stack([l[:, 1] - l[:, 3] / 2, l[:, 2] - l[:, 4] / 2, l[:, 1] + l[:, 3] / 2, l[:, 2] + l[:, 4] / 2], 1)
        inst = Instances(tuple(imgs.shape[-2:]), gt_boxes=Boxes(xyxy), gt_classes=l[:, 0].long())
        out.append({"image": imgs[b].to(torch.uint8), "instances": inst})
    return out


# Full-network comparisons cannot be tighter than the bf16 storage noise: any 1-ulp difference between two
# bf16-storage executions (accumulation order, exp implementation) decorrelates them to the bf16 noise level
# within a few layers (a flipped rounding perturbs the next layer's sums, which flips more roundings).  The
# oracle's own bf16 emulation shows: raw head outputs 1.5-1.9 % (relative L2) from fp32, losses 0.3-2 % when the
# SimOTA assignment flips for a few anchors.  Tight checks therefore live at kernel level (test_gpu_kernels.py) and
# block level (below); the whole network is checked to the noise level and functionally (loss decreases).
def test_training_step_drop_in(golden_dir):
    """YOLOX(cfg).forward(batched_inputs) -> loss dict -> sum().backward(): the reference's contract; values against
    the reference golden (fp32) within the bf16-storage tolerance."""
    g = np.load(os.path.join(golden_dir, "yolox_s_step_64x96.npz"))
    model, sd = _gpu_model()
    model.train()
    imgs, labels = O.synth_batch(2, 64, 96, seed=11, max_gt=4)
    loss_dict = model(_batched_inputs(imgs, labels))
    assert set(loss_dict) == {"total_loss", "iou_loss", "conf_loss", "cls_loss"}
    losses = sum(loss_dict.values())
    losses.backward()
    torch.cuda.synchronize()
    got = np.array([float(loss_dict[k]) for k in ("total_loss", "iou_loss", "conf_loss", "cls_loss")])
    # (a) tight: the loss kernels against the oracle evaluated on the SAME raw head outputs the HIP path produced
    ps = model.plan_for(2, 64, 96, True)
    chk = O.yolox_losses(ps.preds().float().cpu(), labels, ps.anchors.float().cpu(), 80)
    np.testing.assert_allclose(got, np.array([float(x) for x in chk[:4]]), rtol=1e-4, atol=1e-5)
    # (b) whole network against the reference golden (fp32): bounded by bf16 storage noise; on this tiny 64x96 batch a
    # single flipped SimOTA assignment moves the class loss by several percent, the total by < 2 %
    np.testing.assert_allclose(got[0], g["losses"][0], rtol=3e-2)
    np.testing.assert_allclose(got, g["losses"][:4], rtol=1.5e-1, atol=5e-2)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    # raw head output against the fp32 oracle
    net = O.Net({k: v.clone() for k, v in sd.items()}, 0.33, 0.5, 80, training=True)
    with torch.no_grad():
        raw_ref, _ = net.forward_raw(imgs)
    raw = ps.preds().cpu()
    assert float((raw - raw_ref).norm() / raw_ref.norm()) < 4e-2
    # running statistics: fp32 path
    st = model.state_dict()
    np.testing.assert_allclose(st["backbone.stem.conv.bn.running_mean"].cpu().numpy(),
                               g["rm:backbone.stem.conv.bn.running_mean"], rtol=1e-2, atol=1e-2)
    assert int(st["head.stems.2.bn.num_batches_tracked"]) == 1
    # gradient direction vs the reference gradients (golden): at least as aligned as bf16 storage allows
    gn = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
    ratios = [float(p.grad.norm()) / gn[n] for n, p in model.named_parameters() if gn[n] > 1e-6]
    assert 0.5 < float(np.median(ratios)) < 2.0


def test_backward_keeps_autograd_accumulation_semantics():
    """.grad hand-over of the meta-arch's autograd bridge: after zero_grad(set_to_none=True) the arena view is
    bound without a copy; a second backward WITHOUT zero_grad accumulates (2 x the single gradient - BatchNorm running
    statistics aside, the two forwards see the same batch statistics); zero_grad(set_to_none=False) zeroes the arena in
    place and the next backward leaves exactly one gradient"""
    model, _ = _gpu_model(seed=5)
    model.train()
    imgs, labels = O.synth_batch(2, 64, 96, seed=13, max_gt=4)

    def step():
        sum(model(_batched_inputs(imgs, labels)).values()).backward()
        torch.cuda.synchronize()

    step()
    arena = model.params.grad
    assert all(p.grad.data_ptr() == model.params.grad_of(p).data_ptr() for p in model.parameters())   # zero-copy bind
    g1 = arena.clone()
    assert float(g1.norm()) > 0
    step()                                      # no zero_grad: accumulate
    g2 = arena.clone()
    rel = float((g2 - 2 * g1).norm() / (2 * g1).norm())
    assert rel < 2e-2, rel                      # (bf16 activations: the second forward differs by running-stat-free noise only)
    for p in model.parameters():
        p.grad.zero_()                          # zero_grad(set_to_none=False)
    step()
    g3 = arena.clone()
    assert float((g3 - g1).norm() / g1.norm()) < 2e-2
    for p in model.parameters():
        p.grad = None                           # zero_grad(set_to_none=True): bound again without a copy
    step()
    assert all(p.grad.data_ptr() == model.params.grad_of(p).data_ptr() for p in model.parameters())
    assert float((arena - g1).norm() / g1.norm()) < 2e-2


def _block_case(kind, seed):
    """a multi-layer block (fan-out, concat slices, residual adds, pools) as its own plan: GPU vs fp32 autograd"""
    from yolov7_d2_amd.modeling.blocks import CSPLayer, SPPBottleneck, EmitCtx
    from yolov7_d2_amd.plan import PlanBuilder
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    N, H, W, C = 2, 16, 20, 64

    class Holder(nn.Module):
        def __init__(self):
            super().__init__()
            self.blk = CSPLayer(C, C, n=2, shortcut=True) if kind == "csp" else SPPBottleneck(C, C)
    m = Holder()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.15 if p.dim() == 4 else 0.3) + (1.0 if p.dim() == 1 else 0.0))
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.eps, mod.momentum = 1e-3, 0.03
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(DEV)
    arena = ParamArena(m, DEV)
    b = PlanBuilder(DEV, training=True)
    x = b.new_act(N, H, W, C, "x")
    out = m.blk.emit(EmitCtx(b, arena), x, "blk")
    b.grad_mode(out)
    plan = b.finalize()
    xin = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).float()
    gout = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).float()
    plan.view(x).copy_(xin.to(DEV)); plan.view(out.grad).copy_(gout.to(DEV))
    plan.run("fwd"); plan.run("bwd"); torch.cuda.synchronize()
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    class _Q(torch.autograd.Function):   # the product stores activations as bf16: the reference sees the same values
        @staticmethod                        # (matters for max-pool ties, which route gradients by first-max)
        def forward(ctx, t):
            return t.to(torch.bfloat16).float()

        @staticmethod
        def backward(ctx, gr):
            return gr
    net = O.Net(sd, training=True, quant=_Q.apply)
    xr = xin.clone().requires_grad_(True)
    ref = net.csp("blk", xr, 2, True) if kind == "csp" else net.spp("blk", xr)
    ref.backward(gout)
    e_out = float((plan.view(out).float().cpu() - ref.detach()).norm() / ref.detach().norm())
    e_dx = float((plan.view(x.grad).float().cpu() - xr.grad).norm() / xr.grad.norm())
    e_p = {n: float((arena.grad_of(p).cpu() - sd[n].grad).norm() / (sd[n].grad.norm() + 1e-9)) for n, p in m.named_parameters()}
    return e_out, e_dx, e_p


@pytest.mark.parametrize("kind", ["csp", "spp"])
def test_block_level_fwd_bwd(kind):
    e_out, e_dx, e_p = _block_case(kind, 4)
    assert e_out < 2e-2 and e_dx < 5e-2, (e_out, e_dx)
    assert max(e_p.values()) < 8e-2, sorted(e_p.items(), key=lambda kv: -kv[1])[:4]


def test_overfit_one_batch_loss_decreases():
    """functional end-to-end: 30 native steps (fwd + SimOTA + bwd + fused SGD) on one fixed batch reduce the loss the
    way the fp32 CPU oracle's SGD does"""
    from yolov7_d2_amd.engine import NativeTrainer
    model, sd = _gpu_model(seed=0)
    B, H, W = 4, 128, 128
    imgs, labels = O.synth_batch(B, H, W, seed=17, max_gt=5)
    tr = NativeTrainer(model, lr=0.002, use_graph=True)
    st = tr.load_batch(imgs.to(DEV), labels.to(DEV))
    hist = []
    for it in range(30):
        tr.step(st)
        hist.append(float(tr.losses(st)[0]))
    assert all(np.isfinite(hist))
    # oracle trajectory (same init, same batch, same optimiser; detectron2 optimises the SUM of the loss dict = 2x total)
    params = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    norm = {id(v) for k, v in sd.items() if ".bn." in k}
    opt = torch.optim.SGD([{"params": [p for p in params if id(p) not in norm], "weight_decay": 1e-4},
                           {"params": [p for p in params if id(p) in norm], "weight_decay": 0.0}], lr=0.002, momentum=0.9)
    ref = []
    for it in range(30):
        res = O.train_step_losses(sd, imgs, labels)
        opt.zero_grad()
        (res[0] + res[1] + res[2] + res[3]).backward()
        opt.step()
        ref.append(float(res[0]))
    assert hist[-1] < 0.88 * hist[0], (hist, ref)   # the fp32 oracle itself reaches ~0.8 on this batch
    assert abs(hist[0] - ref[0]) / ref[0] < 5e-2
    print("overfit: hip", [round(h, 3) for h in hist[::5]], "oracle", [round(h, 3) for h in ref[::5]])
    assert abs(np.mean(hist[-5:]) - np.mean(ref[-5:])) / np.mean(ref[-5:]) < 0.25, (hist[-5:], ref[-5:])


def test_eval_forward_and_instances(golden_dir):
    g = np.load(os.path.join(golden_dir, "yolox_s_step_64x96.npz"))
    model, sd = _gpu_model()
    model.train()
    imgs, labels = O.synth_batch(2, 64, 96, seed=11, max_gt=4)
    sum(model(_batched_inputs(imgs, labels)).values()).backward()     # same step the golden took (updates BN stats)
    model.eval()
    ps = model.plan_for(2, 64, 96, False)
    ps.image.copy_(imgs.to(DEV))
    ps.plan.run("fwd")
    torch.cuda.synchronize()
    ev = ps.preds().cpu().numpy()
    ref = g["eval_out"]
    np.testing.assert_allclose(ev[..., :4], ref[..., :4], rtol=5e-2, atol=1.0)      # boxes, px
    np.testing.assert_allclose(ev[..., 4:], ref[..., 4:], rtol=5e-2, atol=5e-3)     # probabilities
    with torch.no_grad():
        res = model(_batched_inputs(imgs, labels))
    assert len(res) == 2 and all("instances" in r for r in res)
    assert all(len(r["instances"]) == 0 for r in res)   # random-init net: nothing above conf 0.001


def test_onnx_export_layout_and_l1_switch(golden_dir):
    """(a13) model.onnx_export: forward(NHWC tensor) -> decode_outputs' export layout (xy, wh, conf, argmax class, probs)
    against the reference head run with onnx_export = True; the class-index column is exactly argmax of the probability
    columns; onnx_vis returns the post-processed detections.  (a15) update_iter() past DISABLE_AT_ITER switches the l1
    loss on: the loss dict gains l1_loss and the step still back-propagates."""
    g = np.load(os.path.join(golden_dir, "yolox_s_onnx_layout_64x96.npz"))["out"]
    model, sd = _gpu_model()
    model.eval()
    imgs, labels = O.synth_batch(2, 64, 96, seed=11, max_gt=4)
    model.onnx_export = True
    with torch.no_grad():
        out = model(imgs.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert tuple(out.shape) == (2, 126, 86)
    o = out.cpu().numpy()
    np.testing.assert_allclose(o[..., :4], g[..., :4], rtol=5e-2, atol=1.0)
    np.testing.assert_allclose(o[..., 4], g[..., 4], rtol=5e-2, atol=5e-3)
    np.testing.assert_allclose(o[..., 6:], g[..., 6:], rtol=5e-2, atol=5e-3)
    assert np.array_equal(o[..., 5], o[..., 6:].argmax(-1).astype(np.float32))      # integer column: exact on our probs
    assert (o[..., 5] == g[..., 5]).mean() > 0.5     # random-init class scores are near ties: most, not all, agree
    model.onnx_vis = True
    with torch.no_grad():
        det = model(imgs.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert isinstance(det, list) and len(det) == 2
    model.onnx_export = model.onnx_vis = False
    with pytest.raises(RuntimeError):
        model.train(); model.onnx_export = True; model(_batched_inputs(imgs, labels))
    model.onnx_export = False
    # l1 switch
    assert not model.use_l1
    out0 = model(_batched_inputs(imgs, labels))
    assert "l1_loss" not in out0
    model.update_iter(model.enable_l1_loss_at + 1)
    out1 = model(_batched_inputs(imgs, labels))
    assert model.use_l1 and model.head.use_l1 and set(out1) == {"total_loss", "iou_loss", "conf_loss", "cls_loss", "l1_loss"}
    assert float(out1["l1_loss"]) > 0
    np.testing.assert_allclose(float(out1["total_loss"]) - float(out1["l1_loss"]),
                               float(out1["iou_loss"]) + float(out1["conf_loss"]) + float(out1["cls_loss"]), rtol=1e-4)
    sum(out1.values()).backward()
    gn = model.params.grad.norm()
    assert torch.isfinite(gn) and float(gn) > 0


def test_graph_replay_equals_eager():
    model, _ = _gpu_model(seed=2)
    model.train()
    B, H, W = 2, 64, 64
    imgs, labels = O.synth_batch(B, H, W, seed=13, max_gt=3)
    ps = model.plan_for(B, H, W, True)
    ps.image.copy_(imgs.to(DEV)); ps.labels.copy_(labels.to(DEV))
    ps.gw().fill_(1.0)
    snap = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    ps.plan.run("fwd"); ps.plan.run("bwd"); torch.cuda.synchronize()
    l0, g0 = ps.loss_out().clone(), model.params.grad.clone()
    model.load_state_dict(snap, strict=False)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ps.plan.capture("fwd", s); ps.plan.capture("bwd", s)
        model.load_state_dict(snap, strict=False)
        model.params.grad.zero_()
        ps.plan.launch("fwd", s); ps.plan.launch("bwd", s)
    s.synchronize()
    assert torch.equal(ps.loss_out()[:4], l0[:4])
    assert float((model.params.grad - g0).norm() / g0.norm()) < 1e-3   # atomics in wgrad: order-dependent rounding only


@pytest.mark.parametrize("B,H,W", [(2, 96, 128), (16, 640, 640)], ids=["2x96x128", "bench_16x640x640"])
def test_grouped_launches_equal_separate_launches(monkeypatch, B, H, W):
    """the lane scheduler (grouped CONV / BatchNorm launches for the head's level x branch chains and the CSP conv1 / conv2
    pairs, with their ordering rules for accumulating data gradients) must not change the step: losses and EVERY
    parameter gradient against the same plan with one launch per command (MI_GROUP_LEVELS=0)"""
    monkeypatch.setenv("MI_BN_FUSED", "1")     # (not "auto": both plans must make the same choice)
    imgs, labels = O.synth_batch(B, H, W, seed=17, max_gt=4)
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MI_GROUP_LEVELS", mode)
        model, _ = _gpu_model(seed=3)
        model.train()
        ps = model.plan_for(B, H, W, True)
        ops = [L.OPS[ps.plan.bwd_cmds[0][k].op] for k in range(ps.plan.bwd_cmds[1])]
        assert ("CONV_GROUP" in ops) == (mode == "1") and ("BN_GROUP" in ops) == (mode == "1")
        ps.image.copy_(imgs.to(DEV)); ps.labels.copy_(labels.to(DEV))
        ps.gw().fill_(1.0)
        ps.plan.run("fwd")
        torch.cuda.synchronize()
        runs[mode] = (model, ps)
    l0, l1 = runs["0"][1].loss_out()[:4].cpu().clone(), runs["1"][1].loss_out()[:4].cpu().clone()
    # forward: same kernels, same math, but a grouped member runs on the group's tile shape - its fp32 BatchNorm partial sums
    # are taken in another order - and this randomly initialised network doubles a rounding difference per layer (measured:
    # 4e-7 at the first grouped CSP layer, 6e-3 at the head), which flips a few SimOTA assignments: the class loss moves by
    # ~1e-3 relative.  A dropped contribution is >> 1 %.
    np.testing.assert_allclose(l1.numpy(), l0.numpy(), rtol=1e-2)
    # backward: with that sensitivity two un-pinned executions decorrelate to 3 - 60 % in the gradients whatever the kernels
    # (DESIGN 5), so the backward lists are compared on ONE forward state: every buffer of the ungrouped plan's forward is
    # copied into the grouped plan (same names, same sizes) before both run their backward.  What is left is the order of
    # the statistics / gradient partial sums and the bf16 roundings it flips on the way down: measured <= 2.4 % on the
    # earliest layers at 16 x 640 x 640, <= 1 % on the small case.  A mis-ordered or dropped gradient contribution is >> 10 %.
    p0, p1 = runs["0"][1].plan, runs["1"][1].plan
    b0 = {bf.name: bf for bf in runs["0"][1].builder.bufs}
    ncopied = 0
    for bf in runs["1"][1].builder.bufs:
        src = b0.get(bf.name)
        if src is not None and src.nbytes == bf.nbytes and not bf.name.endswith(".bar"):
            p1.buf_view(bf, torch.uint8).copy_(p0.buf_view(src, torch.uint8))
            ncopied += 1
    assert ncopied > 200
    grads = {}
    for mode in ("0", "1"):
        model, ps = runs[mode]
        ps.plan.run("bwd")
        torch.cuda.synchronize()
        grads[mode] = {n: model.params.grad_of(p).detach().float().cpu().clone() for n, p in model.named_parameters()}
    bad = []
    for n, g0 in grads["0"].items():
        r = float((grads["1"][n] - g0).norm() / (g0.norm() + 1e-12))
        if r > 5e-2:
            bad.append((n, r))
    assert not bad, bad[:8]


@pytest.mark.parametrize("B,H,W", [(2, 96, 128), (16, 640, 640)], ids=["2x96x128", "bench_16x640x640"])
def test_fused_loss_backward_equals_three_passes(monkeypatch, B, H, W):
    """LOSS_BWD_FUSED (gradient tensor + prediction-conv out-gradient maps + bias-gradient block sums in one pass) against
    LOSS_BWD + BIAS_GRADS + SPLIT_DPREDS_BATCH: the fp32 gradient bit-identical, hence every activation / weight gradient
    downstream; the bias gradients differ in summation order only"""
    res = {}
    monkeypatch.setenv("MI_BN_FUSED", "1")
    monkeypatch.setenv("MI_LOSS_DPREDS", "1")
    imgs, labels = O.synth_batch(B, H, W, seed=23, max_gt=6)
    for mode in ("0", "1"):
        monkeypatch.setenv("MI_LOSS_BWD_FUSED", mode)
        model, _ = _gpu_model(seed=6)
        model.train()
        ps = model.plan_for(B, H, W, True)
        ops = [L.OPS[ps.plan.bwd_cmds[0][k].op] for k in range(ps.plan.bwd_cmds[1])]
        assert ("LOSS_BWD_FUSED" in ops) == (mode == "1") and ("SPLIT_DPREDS_BATCH" in ops) == (mode == "0")
        ps.image.copy_(imgs.to(DEV)); ps.labels.copy_(labels.to(DEV))
        ps.gw().fill_(1.0)
        ps.plan.run("fwd"); ps.plan.run("bwd"); torch.cuda.synchronize()
        nch = 85
        dp = ps.plan.buf_view(ps.loss["dpreds"], torch.float32, B * ps.A * nch).cpu().clone()
        grads = {n: model.params.grad_of(p).detach().float().cpu().clone() for n, p in model.named_parameters()}
        res[mode] = (dp, grads)
    assert torch.equal(res["0"][0], res["1"][0]) and float(res["0"][0].abs().max()) > 0
    for n, g0 in res["0"][1].items():
        g1 = res["1"][1][n]
        if "_preds" in n and n.endswith("bias"):
            np.testing.assert_allclose(g1.numpy(), g0.numpy(), rtol=2e-5, atol=1e-7, err_msg=n)
        elif "head." in n and "_preds" in n:
            assert torch.equal(g0, g1), n                      # (the prediction convs' weight gradients read the maps directly)
    # everything below the head: same maps in, same kernels -> only the order of BatchNorm atomics can differ
    worst = max(float((res["1"][1][n] - g0).norm() / (g0.norm() + 1e-12)) for n, g0 in res["0"][1].items())
    assert worst < 5e-2, worst


def test_wgrad_split_groups_equal_single_group(monkeypatch):
    """data-parallel layout of the weight gradients (MI_WGRAD_SPLIT=1: head + neck group mid-backward, backbone group at
    the end, so the first gradient bucket can be all-reduced under the backbone's backward) against the single group:
    same kernels on the same operands - only the split-K partition of a layer may differ (fp32 summation order)"""
    res = {}
    monkeypatch.setenv("MI_BN_FUSED", "1")
    imgs, labels = O.synth_batch(2, 96, 128, seed=19, max_gt=4)
    for mode in ("0", "1"):
        monkeypatch.setenv("MI_WGRAD_SPLIT", mode)
        model, _ = _gpu_model(seed=4)
        model.train()
        ps = model.plan_for(2, 96, 128, True)
        tags = ps.plan.bwd_tags
        assert ("wgrad_group.early" in tags) == (mode == "1") and tags[-1] == "wgrad_group"
        if mode == "1":
            k = tags.index("wgrad_group.early")
            # every head / neck out-gradient (written by the layer's BatchNorm backward) exists before the early group
            # and no backbone backward command has been issued yet (only data gradients of the first neck layers follow)
            assert any("bnbwd" in t for t in tags[:k])
            assert not any(t.startswith(("head.", "neck.")) and ("bnapply" in t or "bnbwd" in t) for t in tags[k + 1:])
            assert not any(t.startswith("backbone.") for t in tags[:k])
        ps.image.copy_(imgs.to(DEV)); ps.labels.copy_(labels.to(DEV))
        ps.gw().fill_(1.0)
        ps.plan.run("fwd"); ps.plan.run("bwd"); torch.cuda.synchronize()
        res[mode] = (ps.loss_out()[:4].cpu().clone(), model.params.grad.detach().cpu().clone())
    assert torch.equal(res["0"][0], res["1"][0]) or torch.allclose(res["0"][0], res["1"][0], rtol=1e-5)
    g0, g1 = res["0"][1], res["1"][1]
    assert float((g1 - g0).norm() / g0.norm()) < 2e-3


def test_async_wgrad_branch_equals_single_group(monkeypatch):
    """MI_WGRAD_ASYNC=G: the weight gradients as G grouped launches on the low-priority auxiliary stream beside the
    backward chain (eager list and captured hipGraph with the parallel branch) against the single group at the end of
    backward: same kernels on the same operands - only the split-K partition of a layer may differ"""
    res = {}
    monkeypatch.setenv("MI_BN_FUSED", "0")     # the fused BatchNorm backward is not used beside an auxiliary stream
    imgs, labels = O.synth_batch(2, 96, 128, seed=19, max_gt=4)
    for mode in ("0", "3"):
        monkeypatch.setenv("MI_WGRAD_ASYNC", mode)
        model, _ = _gpu_model(seed=4)
        model.train()
        ps = model.plan_for(2, 96, 128, True)
        tags = ps.plan.bwd_tags
        assert (sum(t.startswith("wgrad_group.async") for t in tags) == 3) == (mode == "3")
        if mode == "3":
            assert tags[-1] == "wgrad.join"
            # a group is issued only after every out-gradient it reads exists (BatchNorm backward of its layers)
            for gi in range(3):
                k = tags.index(f"wgrad_group.async{gi}")
                cmd = ps.plan.cmd_descs["bwd"][k]
                assert len(cmd) >= 2
        ps.image.copy_(imgs.to(DEV)); ps.labels.copy_(labels.to(DEV))
        ps.gw().fill_(1.0)
        ps.plan.run("fwd"); ps.plan.run("bwd"); torch.cuda.synchronize()
        g_eager = model.params.grad.detach().cpu().clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ps.plan.capture("bwd", s)
            model.params.grad.zero_()
            ps.plan.launch("bwd", s)
        s.synchronize()
        g_graph = model.params.grad.detach().cpu().clone()
        assert float((g_graph - g_eager).norm() / g_eager.norm()) < 1e-3
        res[mode] = (ps.loss_out()[:4].cpu().clone(), g_eager)
    assert torch.allclose(res["0"][0], res["3"][0], rtol=1e-5)
    g0, g1 = res["0"][1], res["3"][1]
    assert float((g1 - g0).norm() / g0.norm()) < 2e-3


@pytest.mark.parametrize("B,H,W", [(2, 96, 128), (16, 640, 640)], ids=["2x96x128", "bench_16x640x640"])
def test_bn_backward_fused_step_equals_two_pass(monkeypatch, B, H, W):
    """BatchNorm backward as ONE launch per layer (registers / LDS across a grid-wide barrier, also inside the grouped
    launches of the head and the CSP pairs) against reduce + apply on the whole step: same losses, every parameter
    gradient within the bf16 decorrelation level, no barrier wait gave up; "auto" picks one of the two by timing them"""
    res = {}
    imgs, labels = O.synth_batch(B, H, W, seed=23, max_gt=4)
    for mode in ("0", "1", "auto"):
        monkeypatch.setenv("MI_BN_FUSED", mode)
        model, _ = _gpu_model(seed=5)
        model.train()
        ps = model.plan_for(B, H, W, True)
        plan = ps.plan
        arr, n = plan.bwd_cmds
        ops = [L.OPS[arr[k].op] for k in range(n)]
        kinds = [arr[k].i[0] for k in range(n) if L.OPS[arr[k].op] == "BN_GROUP"]
        if mode == "auto":
            assert plan.bn_fused_timing is not None and set(plan.bn_fused_timing) == {"fused_ms", "two_pass_ms"}
            assert plan.bn_fused == (plan.bn_fused_timing["fused_ms"] <= plan.bn_fused_timing["two_pass_ms"])
        else:
            assert plan.bn_fused == (mode == "1") and plan.bn_fused_timing is None
        assert ("BN_BWD_FUSED" in ops) == plan.bn_fused and ("BN_BWD_REDUCE" in ops) == (not plan.bn_fused)
        assert all(k == 3 for k in kinds) if plan.bn_fused else all(k in (1, 2) for k in kinds)
        ps.image.copy_(imgs.to(DEV)); ps.labels.copy_(labels.to(DEV))
        ps.gw().fill_(1.0)
        for _ in range(2):          # twice: the barrier words carry their generation over
            plan.run("fwd"); plan.run("bwd")
        torch.cuda.synchronize()
        for c in ps.builder.bwd:
            if L.OPS[c.op] == "BN_BWD_APPLY":
                words = plan.buf_view(c.p[12].obj, torch.int32)
                assert int(words[2]) == 0 and int(words[0]) == 0, c.tag
        grads = {n_: model.params.grad_of(p).detach().float().cpu().clone() for n_, p in model.named_parameters()}
        res[mode] = (ps.loss_out()[:4].cpu().clone(), grads)
    for mode in ("1", "auto"):
        np.testing.assert_allclose(res[mode][0].numpy(), res["0"][0].numpy(), rtol=1e-5)
        bad = []
        for n_, g0 in res["0"][1].items():
            g1 = res[mode][1][n_]
            r = float((g1 - g0).norm() / (g0.norm() + 1e-12))
            if r > 5e-2:
                bad.append((n_, r))
        assert not bad, bad[:8]


# ------------------------------------------------------------------------------------------------ NMS family (f3)
NMS_CASES = dict(a=(300, 5, 71, 14.0, 0.001), b=(1500, 80, 72, 14.0, 0.001), c=(7, 1, 73, 14.0, 0.001), d=(400, 2, 74, 4.0, 0.05))


@pytest.mark.parametrize("case", sorted(NMS_CASES))
@pytest.mark.parametrize("nms_type", ["softnms-linear", "softnms-gaussian", "cluster"])
def test_generalized_batched_nms_against_reference_golden(golden_dir, case, nms_type):
    """MODEL.NMS_TYPE variants (meta_arch/utils.py:33-113) against the reference's own functions run on the same seeded
    candidates: Soft-NMS rescales the scores in place (fp32 tolerance) and keeps what stays above the threshold, in
    descending score order; Cluster-NMS's fixed point equals class-by-class greedy NMS (index sets identical)"""
    import sys
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from gen_golden_inputs import synth_nms_case
    from yolov7_d2_amd.modeling import generalized_batched_nms
    g = np.load(os.path.join(golden_dir, "nms_family.npz"))
    n, ncls, seed, spread, thr = NMS_CASES[case]
    boxes, scores, idxs = synth_nms_case(n, ncls, seed, spread)
    sc = scores.clone().to(DEV)
    keep = generalized_batched_nms(boxes.to(DEV), sc, idxs.to(DEV), 0.5, score_threshold=thr, nms_type=nms_type).cpu().numpy()
    ref_keep = g[f"{case}_{nms_type}_keep"]
    if nms_type == "cluster":
        assert sorted(keep.tolist()) == sorted(ref_keep.tolist())
        assert np.all(np.diff(scores.numpy()[keep]) <= 0)                       # descending score
        assert torch.equal(sc.cpu(), scores)                                     # scores untouched
        return
    ref_sc = g[f"{case}_{nms_type}_scores"]
    np.testing.assert_allclose(sc.cpu().numpy(), ref_sc, rtol=2e-5, atol=1e-7)
    # membership: everything clearly above / below the threshold agrees (a score within 1e-5 of it may fall either side)
    clear = np.abs(ref_sc - thr) > 1e-5 * max(thr, 1e-3)
    mine, theirs = np.zeros(n, bool), np.zeros(n, bool)
    mine[keep] = True; theirs[ref_keep] = True
    assert np.array_equal(mine[clear], theirs[clear])
    assert np.all(np.diff(sc.cpu().numpy()[keep]) <= 0)
    if case == "d":
        assert theirs.sum() < 0.8 * n      # the case really retires boxes


def test_generalized_batched_nms_dispatch_and_edges():
    from yolov7_d2_amd.modeling import generalized_batched_nms, batched_nms
    b = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60]], dtype=torch.float32, device=DEV)
    s = torch.tensor([0.9, 0.8, 0.7], device=DEV)
    i = torch.zeros(3, device=DEV)
    assert generalized_batched_nms(b, s.clone(), i, 0.5, nms_type="normal").tolist() == batched_nms(b, s, i, 0.5).tolist() == [0, 2]
    with pytest.raises(NotImplementedError):
        generalized_batched_nms(b, s.clone(), i, 0.5, nms_type="matrix")
    e = generalized_batched_nms(b[:0], s[:0].clone(), i[:0], 0.5, nms_type="softnms-gaussian")
    assert e.numel() == 0 and e.dtype == torch.int64
    with pytest.raises(ValueError):
        generalized_batched_nms(b, s.double(), i, 0.5, nms_type="softnms-linear")


@pytest.mark.parametrize("case,n,H,W,ncls,seed", [("m1", 60, 40, 48, 3, 81), ("m2", 500, 64, 64, 10, 82), ("m3", 1, 8, 8, 1, 83)])
@pytest.mark.parametrize("kernel", ["gaussian", "linear"])
def test_matrix_nms_against_reference_golden(golden_dir, case, n, H, W, ncls, seed, kernel):
    """SOLOv2's Matrix NMS (utils/solov2_utils.py:160-206) against the reference function: the mask-intersection matrix
    comes from the MFMA pixel-sum kernel (exact integers), the decay is fp32"""
    import sys
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from gen_golden_inputs import synth_mask_case
    from yolov7_d2_amd.modeling import matrix_nms
    g = np.load(os.path.join(golden_dir, "nms_family.npz"))
    labels, masks, sums, scores = synth_mask_case(n, H, W, ncls, seed)
    out = matrix_nms(labels.to(DEV), masks.to(DEV), sums.to(DEV), scores.to(DEV), sigma=2.0, kernel=kernel)
    np.testing.assert_allclose(out.cpu().numpy(), g[f"{case}_{kernel}"], rtol=2e-5, atol=1e-7)
    assert matrix_nms(labels[:0], masks[:0], sums[:0], scores[:0]) == []


@pytest.mark.parametrize("case,n,H,W,ncls,seed", [("m1", 60, 40, 48, 3, 81), ("m3", 1, 8, 8, 1, 83)])
def test_mask_nms_against_reference_golden(golden_dir, case, n, H, W, ncls, seed):
    """greedy mask NMS (utils/solov2_utils.py:209-236) against the reference's own double loop: keep flags exact"""
    import sys
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from gen_golden_inputs import synth_mask_case
    from yolov7_d2_amd.modeling import mask_nms
    g = np.load(os.path.join(golden_dir, "nms_family.npz"))
    labels, masks, sums, scores = synth_mask_case(n, H, W, ncls, seed)
    keep = mask_nms(labels.to(DEV), masks.to(DEV), sums.to(DEV), scores.to(DEV), nms_thr=0.3)
    assert np.array_equal(keep.cpu().numpy().astype(np.float32), g[f"{case}_masknms"])
    if n > 1:
        assert 0 < float(keep.sum()) < n
