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


def _loss_call(raw, labels, anchors, gw=(1, 1, 1, 1), gmax=None):
    B, A, nch = raw.shape
    ML = labels.shape[1]
    gmax = ML if gmax is None else gmax
    t = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)
    ws = dict(cost=t(B, gmax, A), iou=t(B, gmax, A), match=t(B, gmax, A, dt=torch.uint8), ngt=t(B, dt=torch.int32),
              fg=t(B, A, dt=torch.uint8), matched_gt=t(B, A, dt=torch.int32), matched_iou=t(B, A),
              partial=t(B * ((A + 255) // 256), 4), out=t(8), dpreds=t(B, A, nch),
              gw=torch.tensor(gw, dtype=torch.float32, device=DEV))
    rd, ld, ad = raw.to(DEV).contiguous(), labels.to(DEV).contiguous(), anchors.to(DEV).contiguous()
    d = L.mi_yolox_loss_desc()
    d.preds, d.labels, d.anchors = rd.data_ptr(), ld.data_ptr(), ad.data_ptr()
    d.B, d.A, d.ncls, d.max_labels, d.gmax = B, A, nch - 5, ML, gmax
    for k in ("cost", "iou", "match", "ngt", "fg", "matched_gt", "matched_iou", "partial", "out"):
        setattr(d, k, ws[k].data_ptr())
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


def test_training_step_drop_in(golden_dir):
    """YOLOX(cfg).forward(batched_inputs) -> loss dict -> sum().backward(): the reference's contract; values against
    the reference golden (fp32) within the bf16-storage tolerance, and tightly against the same algorithm
    interpreted on CPU with bf16 storage."""
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
    # tolerance vs the fp32 reference: bf16 storage of ~190 activation tensors (measured with the oracle's own bf16
    # emulation: 0.3-0.6 % on the losses)
    np.testing.assert_allclose(got, g["losses"][:4], rtol=3e-2, atol=3e-2)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    # same algorithm on the CPU (bf16 storage): tight
    cpu_model = M.build_model(M.yolox_s_cfg(device="cpu"))
    cpu_model.load_state_dict(sd)
    cpu_model.params = ParamArena(cpu_model, "cpu")
    ps = _PlanState(cpu_model, 2, 64, 96, True, materialize=False)
    ps.image.copy_(imgs); ps.labels.copy_(labels)
    it = Interp(ps.builder)
    it.run(ps.builder.prologue + ps.builder.fwd)
    out_cpu = it.raw(ps.loss["out"]).view(torch.float32)[:4].numpy()
    np.testing.assert_allclose(got, out_cpu, rtol=1e-2, atol=1e-2)
    # running statistics: fp32 path
    st = model.state_dict()
    np.testing.assert_allclose(st["backbone.stem.conv.bn.running_mean"].cpu().numpy(),
                               g["rm:backbone.stem.conv.bn.running_mean"], rtol=1e-2, atol=1e-2)
    assert int(st["head.stems.2.bn.num_batches_tracked"]) == 1


def test_backward_with_fixed_head_gradient():
    """network backward (all conv/BN/pool/upsample gradient kernels, fan-in flags) with the loss gradient FIXED, so
    that SimOTA's discrete assignment cannot turn rounding noise into different targets: GPU vs the same algorithm
    interpreted on the CPU with bf16 storage."""
    model, sd = _gpu_model(seed=1)
    model.train()
    B, H, W = 2, 64, 96
    imgs, labels = O.synth_batch(B, H, W, seed=12, max_gt=4)
    ps = model.plan_for(B, H, W, True)
    A, nch = ps.A, ps.nch
    R = (torch.randn(B, A, nch, generator=torch.Generator().manual_seed(5)) / A)
    ps.image.copy_(imgs.to(DEV)); ps.labels.copy_(labels.to(DEV))
    ps.plan.run("fwd")
    ps.plan.buf_view(ps.loss["dpreds"], torch.float32, B * A * nch).copy_(R.reshape(-1).to(DEV))
    arr, n = ps.plan.bwd_cmds
    assert ps.plan.bwd_tags[0] == "loss.bwd"
    L.check(L.lib().mi_cmdlist_run(C.cast(C.byref(arr, C.sizeof(L.mi_cmd)), C.POINTER(L.mi_cmd)), n - 1, L.stream_ptr()), "bwd")
    torch.cuda.synchronize()
    raw_gpu = ps.preds().cpu()
    cpu_model = M.build_model(M.yolox_s_cfg(device="cpu"))
    cpu_model.load_state_dict(sd)
    cpu_model.params = ParamArena(cpu_model, "cpu")
    pc = _PlanState(cpu_model, B, H, W, True, materialize=False)
    pc.image.copy_(imgs); pc.labels.copy_(labels)
    it = Interp(pc.builder)
    it.run(pc.builder.prologue + pc.builder.fwd)
    raw_cpu = it.raw(pc.preds_buf).view(torch.float32)[: B * A * nch].view(B, A, nch)
    assert float((raw_gpu - raw_cpu).norm() / raw_cpu.norm()) < 5e-3
    it.raw(pc.loss["dpreds"]).view(torch.float32)[: B * A * nch] = R.reshape(-1)
    it.run(pc.builder.bwd[1:])
    worst = []
    for (name, p), (_, q) in zip(model.named_parameters(), cpu_model.named_parameters()):
        a, b = model.params.grad_of(p).float().cpu(), cpu_model.params.grad_of(q).float()
        if float(b.norm()) > 0:
            worst.append((float((a - b).norm() / b.norm()), name))
    worst.sort(reverse=True)
    med = float(np.median([w[0] for w in worst]))
    assert med < 3e-2 and worst[0][0] < 0.15, (med, worst[:5])


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
