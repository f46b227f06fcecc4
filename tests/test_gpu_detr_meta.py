"""GPU (-m gpu): the `Detr` meta-architecture end to end (BASELINE.json config 4) - ResNet-50 + padding masks + sine
position encoding + transformer + GPU Hungarian matcher + set criterion - against tests/golden/detr_meta.npz, i.e. the
reference's OWN Detr class executed by path (oracle/gen_golden.py::gold_detr_meta); dropout (elementwise and inside the
fused attention) statistically and, for the mask in use, exactly; one step of the full 6 + 6-layer configuration."""
import os

import numpy as np
import pytest
import torch

import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.d2shim import Boxes, Instances
from yolov7_d2_amd.modeling.attention import mha_core
from yolov7_d2_amd.modeling.transformer import _DropoutFn
from gen_golden_inputs import seeded_tensor_dict, synth_detr_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(batch):
    return [dict(image=b["image"], instances=Instances(b["size"], gt_boxes=Boxes(b["boxes"]), gt_classes=b["classes"]))
            for b in batch]


def test_detr_meta_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "detr_meta.npz"))
    cfg = M.detr_r50_cfg(device=DEV)
    cfg.MODEL.DETR.ENC_LAYERS, cfg.MODEL.DETR.DEC_LAYERS, cfg.MODEL.DETR.NUM_OBJECT_QUERIES = 2, 2, 30
    cfg.MODEL.DETR.DROPOUT = 0.0
    cfg.MODEL.YOLO.CONF_THRESHOLD = 0.02
    model = M.build_model(cfg)
    sd = model.state_dict()
    assert sorted(sd.keys()) == [str(k) for k in g["state_keys"]]          # the reference's own state_dict keys
    model.load_state_dict(seeded_tensor_dict({k: v.shape for k, v in sd.items()}, seed=203), strict=False)
    inputs = _inputs(synth_detr_batch())
    model.train()
    losses = model(inputs)
    assert sorted(losses.keys()) == [str(k) for k in g["loss_keys"]]
    got = {k: float(v) for k, v in losses.items()}
    print({k: (round(got[k], 4), round(float(g["loss:" + k]), 4)) for k in got})
    for k in got:
        ref = float(g["loss:" + k])
        if "error" in k:        # class_error (%) / cardinality_error: integer-valued statistics of arg-maxes
            assert abs(got[k] - ref) <= 0.1 * abs(ref) + 2.0, (k, got[k], ref)
        else:                   # weighted losses: fp32 reference vs the bf16 network
            assert abs(got[k] - ref) <= 3e-2 * abs(ref) + 1e-2, (k, got[k], ref)
    total = sum(v for k, v in losses.items() if k in model.criterion.weight_dict)
    total.backward()
    trainable = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in trainable)
    assert all(p.grad is None for n, p in model.named_parameters() if not p.requires_grad)    # stem + res2 frozen
    model.eval()
    with torch.no_grad():
        images = model.preprocess_image(inputs)
        out = model.detr(images)
        dets = model(inputs)
    lg, bx = out["pred_logits"].float().cpu().numpy(), out["pred_boxes"].float().cpu().numpy()
    assert np.abs(lg - g["eval_logits"]).max() < 0.15 and np.abs(bx - g["eval_boxes"]).max() < 2e-2
    for i, d in enumerate(dets):
        inst = d["instances"]
        assert len(inst) == g[f"det{i}_boxes"].shape[0]
        same = (inst.pred_classes.cpu().numpy() == g[f"det{i}_classes"]).mean()
        assert same > 0.8, same                                            # arg-max classes (near-uniform logits)
        np.testing.assert_allclose(inst.pred_boxes.tensor.cpu().numpy(), g[f"det{i}_boxes"], rtol=5e-2, atol=3.0)
        np.testing.assert_allclose(inst.scores.cpu().numpy(), g[f"det{i}_scores"], rtol=1e-1, atol=2e-3)


def test_elementwise_dropout_statistics_and_backward():
    x = torch.randn(64, 1024, device=DEV).to(torch.bfloat16).requires_grad_(True)
    y = _DropoutFn.apply(x, 0.1, 12345)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.9) < 5e-3, keep
    nz = y != 0
    torch.testing.assert_close(y[nz].float(), (x.detach()[nz].float() / 0.9).to(torch.bfloat16).float(), rtol=1e-2, atol=1e-3)
    y2 = _DropoutFn.apply(x, 0.1, 12345)
    assert torch.equal(y, y2)                                  # a pure function of (seed, index)
    assert not torch.equal(y, _DropoutFn.apply(x, 0.1, 12346))
    g = torch.ones_like(y)
    y.backward(g)
    assert torch.equal(x.grad != 0, nz)                        # the backward applies the SAME mask


def test_residual_add_fused_into_the_dropout_pass_equals_two_launches():
    """_AddDroppedFn (mi_dropout_add_bf16: res + dropout(x) in one pass) against _DropoutFn followed by the elementwise add:
    identical bits forward, identical gradients (g to the residual, the SAME mask applied to g for the branch)"""
    from yolov7_d2_amd.modeling.transformer import _AddDroppedFn, _AddFn
    g = torch.Generator().manual_seed(4)
    mk = lambda: torch.randn(300, 4, 256, generator=g).to(torch.bfloat16).to(DEV)
    res, x, go = mk(), mk(), mk()
    r1, x1 = res.clone().requires_grad_(True), x.clone().requires_grad_(True)
    r2, x2 = res.clone().requires_grad_(True), x.clone().requires_grad_(True)
    a = _AddDroppedFn.apply(r1, x1, 0.1, 777)
    b = _AddFn.apply(r2, _DropoutFn.apply(x2, 0.1, 777))
    assert torch.equal(a, b)
    a.backward(go)
    b.backward(go)
    assert torch.equal(r1.grad, r2.grad) and torch.equal(x1.grad, x2.grad)
    assert 0.08 < float((x1.grad == 0).float().mean()) < 0.12


@pytest.mark.parametrize("p", [0.1, 0.0])
def test_post_norm_residual_as_one_node_equals_add_dropped_then_layernorm(p):
    """_AddDroppedLayerNormFn (mi_dropout_add_layernorm_fwd / mi_layernorm_bwd_dropout) against _AddDroppedFn (or _AddFn for
    p = 0) followed by _LayerNormFn: identical output and identical gradients for the residual, the dropped branch, gamma and
    beta (T = 4200 rows: the DETR encoder at 800 x 1333, and a ragged T)"""
    from yolov7_d2_amd.modeling.transformer import _AddDroppedFn, _AddDroppedLayerNormFn, _AddFn, _LayerNormFn
    g = torch.Generator().manual_seed(9)
    for T in (4200, 403):
        E = 256
        mk = lambda: torch.randn(T, E, generator=g).to(torch.bfloat16).to(DEV)
        res, x, go = mk(), mk(), mk()
        gam, bet = (1 + 0.1 * torch.randn(E, generator=g)).to(DEV), (0.1 * torch.randn(E, generator=g)).to(DEV)
        out = []
        for fused in (True, False):
            r, xx = res.clone().requires_grad_(True), x.clone().requires_grad_(True)
            ga, be = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
            if fused:
                y = _AddDroppedLayerNormFn.apply(r, xx, ga, be, 1e-5, p, 4242)
            else:
                sm = _AddDroppedFn.apply(r, xx, p, 4242) if p > 0 else _AddFn.apply(r, xx)
                y = _LayerNormFn.apply(sm, ga, be, 1e-5)
            y.backward(go)
            out.append((y.detach(), r.grad, xx.grad, ga.grad, be.grad))
        for a, c in zip(*out):
            assert torch.equal(a, c), (T, p, float((a.float() - c.float()).abs().max()))


def test_in_projection_as_one_node_equals_three_sliced_linears(monkeypatch):
    """_InProjFn (q, k, v from the WHOLE in_proj_weight / in_proj_bias in one autograd node, the three weight-gradient
    launches writing their row blocks of one [3E, E] tensor) against three _LinearFn calls on parameter slices (autograd's
    SliceBackward + accumulation): identical outputs, input gradients and parameter gradients.  (One launch per weight
    gradient on both sides: the per-layer grouped form, ops.WgradBatch, picks other split-K counts - its own test is
    test_gpu_detr.py::test_layer_grouped_weight_gradients_equal_single_launches.)"""
    monkeypatch.setenv("MI_WGRAD_LAYER_GROUP", "0")
    from yolov7_d2_amd.modeling.transformer import _InProjFn, _LinearFn
    g = torch.Generator().manual_seed(6)
    E, Tq, Tk = 256, 400, 1040
    w = (torch.randn(3 * E, E, generator=g) * 0.05).to(DEV)
    b = (torch.randn(3 * E, generator=g) * 0.1).to(DEV)
    xq, xk, xv = (torch.randn(t, E, generator=g).to(torch.bfloat16).to(DEV) for t in (Tq, Tk, Tk))
    gos = [torch.randn(t, E, generator=g).to(torch.bfloat16).to(DEV) for t in (Tq, Tk, Tk)]
    res = []
    for one in (True, False):
        ws, bs = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        xs = [t.clone().requires_grad_(i != 2) for i, t in enumerate((xq, xk, xv))]       # (value without a gradient: memory)
        if one:
            outs = _InProjFn.apply(xs[0], xs[1], xs[2], ws, bs)
        else:
            outs = [_LinearFn.apply(xs[i], ws[i * E:(i + 1) * E], bs[i * E:(i + 1) * E]) for i in range(3)]
        torch.autograd.backward(list(outs), gos)
        res.append([o.detach() for o in outs] + [xs[0].grad, xs[1].grad, ws.grad, bs.grad])
        assert xs[2].grad is None
    for a, c in zip(*res):
        assert torch.equal(a, c), float((a.float() - c.float()).abs().max())


@pytest.mark.parametrize("L_,B,masked,p", [(1050, 2, True, 0.0), (100, 4, False, 0.0), (333, 3, True, 0.1)])
def test_self_attention_with_packed_q_k_projection_equals_two_projections(monkeypatch, L_, B, masked, p):
    """a self-attention whose query and key are the same tensor (q = k = x + pos, detr_backbone.py:155-157,222-224) projects
    q | k with ONE [T, 2E] launch and the attention kernels read / write them with row stride 2E (mi_mha_*_ld): against the
    two-projection form (MI_MHA_QK_PACKED=0) - same dot products, so the outputs agree to bf16 rounding of a differently
    tiled GEMM and the gradients to the tolerance of the bf16 kernels (the packed form adds dq W_q + dk W_k inside one
    reduction where the other rounds twice)"""
    from yolov7_d2_amd.modeling.transformer import MultiheadAttention
    E, H = 256, 8
    g = torch.Generator().manual_seed(L_ + B)
    x = torch.randn(L_, B, E, generator=g).to(torch.bfloat16)
    val = torch.randn(L_, B, E, generator=g).to(torch.bfloat16)
    go = torch.randn(L_, B, E, generator=g).to(torch.bfloat16)
    kpm = None
    if masked:
        kpm = torch.zeros(B, L_, dtype=torch.bool)
        kpm[0, L_ - L_ // 4:] = True
        kpm = kpm.to(DEV)
    sd = seeded_tensor_dict({"in_proj_weight": (3 * E, E), "in_proj_bias": (3 * E,), "out_proj.weight": (E, E), "out_proj.bias": (E,)}, seed=5)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MI_MHA_QK_PACKED", mode)
        m = MultiheadAttention(E, H, dropout=p).to(DEV)
        m.load_state_dict(sd)
        m.train()
        torch.manual_seed(11)                              # (the dropout seed comes from torch's generator)
        xq = x.to(DEV).requires_grad_(True)
        xv = val.to(DEV).requires_grad_(True)
        out, _ = m(xq, xq, xv, key_padding_mask=kpm)
        out.backward(go.to(DEV))
        torch.cuda.synchronize()
        res[mode] = dict(out=out.detach().float().cpu(), dx=xq.grad.float().cpu(), dv=xv.grad.float().cpu(),
                         gw=m.in_proj_weight.grad.cpu(), gb=m.in_proj_bias.grad.cpu(), go=m.out_proj.weight.grad.cpu())
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    for k in res["0"]:
        assert rel(res["1"][k], res["0"][k]) < 2e-2, (k, rel(res["1"][k], res["0"][k]))
    assert rel(res["1"]["out"], res["0"]["out"]) < 5e-3


def test_linear_with_relu_epilogue_equals_linear_then_relu():
    """_LinearFn(relu=True) (MI_CONV_RELU in the 1x1 convolution's epilogue, the mask applied to dy in backward) against
    _LinearFn followed by _ReluFn: identical bits, forward and all three gradients (Cout 2048 and a padded Cout 72)"""
    from yolov7_d2_amd.modeling.transformer import _LinearFn, _ReluFn
    g = torch.Generator().manual_seed(8)
    for T, cin, cout in ((1200, 256, 2048), (400, 256, 72)):
        x = torch.randn(T, cin, generator=g).to(torch.bfloat16).to(DEV)
        w = (torch.randn(cout, cin, generator=g) * 0.05).to(DEV)
        b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
        go = torch.randn(T, cout, generator=g).to(torch.bfloat16).to(DEV)
        res = []
        for fused in (True, False):
            xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = _LinearFn.apply(xs, ws, bs, True) if fused else _ReluFn.apply(_LinearFn.apply(xs, ws, bs).contiguous())
            y.backward(go)
            res.append((y.detach(), xs.grad, ws.grad, bs.grad))
        for a, c in zip(*res):
            assert torch.equal(a, c), (T, cout, float((a.float() - c.float()).abs().max()))
        assert float((res[0][0] == 0).float().mean()) > 0.2


def test_attention_dropout_exact_for_its_mask():
    """mha with attention-weight dropout against fp32 softmax attention that applies the SAME keep mask (exported by
    mi_mha_dropout_mask): forward and dq / dk / dv; plus: unbiased over seeds"""
    gen = torch.Generator().manual_seed(5)
    Lq, Lk, B, E, nh, p, seed = 48, 80, 2, 256, 8, 0.1, 987654321
    bf = lambda t: t.to(torch.bfloat16).float()
    q, k, v = (bf(torch.randn(n, B, E, generator=gen) * 0.7) for n in (Lq, Lk, Lk))
    kpm = torch.zeros(B, Lk, dtype=torch.bool)
    kpm[1, Lk - 9:] = True
    mask = torch.empty(B, nh, Lq, Lk, dtype=torch.uint8, device=DEV)
    L.check(L.lib().mi_mha_dropout_mask(mask.data_ptr(), B, nh, Lq, Lk, p, seed, L.stream_ptr()), "mask")
    keep = mask.float().cpu()
    assert abs(float(keep.mean()) - (1 - p)) < 1e-2
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    d = E // nh
    qh = qr.view(Lq, B, nh, d).permute(1, 2, 0, 3); kh = kr.view(Lk, B, nh, d).permute(1, 2, 0, 3)
    vh = vr.view(Lk, B, nh, d).permute(1, 2, 0, 3)
    s = qh @ kh.transpose(-1, -2) / d ** 0.5
    s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, -1) * keep / (1 - p)
    ref = (pr @ vh).permute(2, 0, 1, 3).reshape(Lq, B, E)
    go = bf(torch.randn(ref.shape, generator=gen))
    ref.backward(go)
    qd, kd, vd = (t.to(DEV, torch.bfloat16).requires_grad_(True) for t in (q, k, v))
    o = mha_core(qd, kd, vd, kpm.to(DEV), nh, p, seed)
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    assert rel(o, ref.detach()) < 2e-2
    o.backward(go.to(DEV, torch.bfloat16))
    assert rel(qd.grad, qr.grad) < 3e-2 and rel(kd.grad, kr.grad) < 3e-2 and rel(vd.grad, vr.grad) < 3e-2
    # unbiased: the mean over seeds approaches the no-dropout output
    with torch.no_grad():
        base = mha_core(qd, kd, vd, kpm.to(DEV), nh).float()
        acc = torch.zeros_like(base)
        for sd_ in range(64):
            acc += mha_core(qd, kd, vd, kpm.to(DEV), nh, p, 1000 + sd_).float()
        assert float((acc / 64 - base).norm() / base.norm()) < 0.08


def test_detr_r50_full_config_step_with_dropout():
    """the real configuration (6 + 6 layers, 100 queries, dropout 0.1, deep supervision, FREEZE_AT 2) on a padded batch
    of different-sized images: loss dict, backward, AdamW step with the backbone lr multiplier on the flat arena-free
    parameters - everything finite, loss decreases over a few steps on the fixed batch"""
    torch.manual_seed(0)
    cfg = M.detr_r50_cfg(device=DEV)
    model = M.build_model(cfg)
    model.train()
    inputs = _inputs(synth_detr_batch(seed=7, sizes=((320, 416), (288, 480))))
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4)
    hist = []
    for it in range(4):
        losses = model(inputs)
        assert len(losses) == 4 * 6 + 1  # (ce, cardinality_error, bbox, giou) x 6 decoder levels + class_error of the last
        total = sum(v for k, v in losses.items() if k in model.criterion.weight_dict)
        opt.zero_grad()
        total.backward()
        assert all(torch.isfinite(p.grad).all() for p in params)
        torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()
        hist.append(float(total))
    print("detr-r50 loss:", [round(h, 3) for h in hist])
    assert np.isfinite(hist).all() and hist[-1] < hist[0]


def test_detr_r50_real_size_against_reference_golden(golden_dir):
    """BASELINE configs[3] at its real size - 6 + 6 layers, 100 queries, deep supervision, a padded batch of an 800 x 1333
    and a 768 x 1205 image, dropout 0 - against the reference's own Detr class run by path on the CPU in fp32
    (oracle/gen_golden.py::gold_detr_real): the 25 entries of the training loss dict, the eval logits / boxes, and every
    trainable parameter's gradient through its fingerprint (norm + 8 seeded +-1 projections: the mean squared projection
    difference estimates |g - g_ref|^2).  Un-forced: the bf16 network runs on its own activations, so ReLU / arg-max /
    matching decisions near a tie may flip; the bounds are what that costs, measured."""
    from gen_golden_inputs import grad_signature
    g = np.load(os.path.join(golden_dir, "detr_real.npz"))
    cfg = M.detr_r50_cfg(device=DEV)
    cfg.MODEL.DETR.DROPOUT = 0.0
    model = M.build_model(cfg)
    sd = model.state_dict()
    model.load_state_dict(seeded_tensor_dict({k: v.shape for k, v in sd.items()}, seed=207), strict=False)
    with torch.no_grad():     # (conditioning, as the golden's generator: encoder tokens of norm O(10) instead of 1.4e4)
        model.detr.input_proj.weight.mul_(1e-3)
    inputs = _inputs(synth_detr_batch(seed=211, sizes=((800, 1333), (768, 1205))))
    model.train()
    feats = {}
    def keep(m, i, o):
        for k, v in o.items():
            if v.requires_grad:
                v.retain_grad()
            feats[k] = v
    hk = model.detr.backbone[0].backbone.register_forward_hook(keep)
    srcs = {}

    def keep_src(m, args):
        args[0].retain_grad()
        srcs["src"] = args[0]
    hk2 = model.detr.transformer.register_forward_pre_hook(keep_src)
    # the assignment is teacher-forced (the reference's matcher answers, in its call order: last level, aux 0..4): with
    # two ground truths per image only two queries per image and level carry box gradients, and WHICH two is a near tie
    # at random initialisation.  Our own matcher's answers on our own outputs are compared first (agreement is reported).
    own = model.criterion.matcher
    calls = {"n": 0, "same": 0, "total": 0}

    def forced(outputs, targets):
        c = calls["n"]
        calls["n"] += 1
        mine = own(outputs, targets)
        ref = [(torch.from_numpy(g[f"match:{c}:{b}:q"]), torch.from_numpy(g[f"match:{c}:{b}:t"])) for b in range(len(targets))]
        for (i, j), (ri, rj) in zip(mine, ref):
            a = {(int(x), int(y)) for x, y in zip(i.tolist(), j.tolist())}
            r = {(int(x), int(y)) for x, y in zip(ri.tolist(), rj.tolist())}
            calls["same"] += len(a & r)
            calls["total"] += len(r)
        return ref
    class _Forced(torch.nn.Module):      # (not a HungarianMatcher: the criterion takes the foreign-matcher path)
        def forward(self, outputs, targets):
            return forced(outputs, targets)
    model.criterion.matcher = _Forced()
    losses = model(inputs)
    model.criterion.matcher = own
    hk.remove()
    hk2.remove()
    assert calls["n"] == int(g["n_match_calls"]) == 6
    print("matcher agreement on own outputs: %d of %d pairs" % (calls["same"], calls["total"]))
    assert sorted(losses.keys()) == [str(k) for k in g["loss_keys"]] and len(losses) == 25
    # the backbone's output maps against the reference's (fingerprints): the forward at 800 x 1333
    fs = {n[5:]: v for n, v in grad_signature([("feat:" + k, v.detach()) for k, v in feats.items()]).items()}
    frel = {k: float(np.sqrt(np.mean((v[1:] - g["fsig:" + k][1:]) ** 2)) / g["fsig:" + k][0]) for k, v in fs.items()}
    print("backbone feature rel err", {k: round(v, 4) for k, v in frel.items()}, {k: round(float(v[0] / g["fsig:" + k][0]), 4) for k, v in fs.items()})
    assert all(v < 5e-2 for v in frel.values()), frel
    got = {k: float(v) for k, v in losses.items()}
    worst = 0.0
    for k in got:
        ref = float(g["loss:" + k])
        if "error" in k:
            assert abs(got[k] - ref) <= 0.1 * abs(ref) + 2.0, (k, got[k], ref)
        else:
            worst = max(worst, abs(got[k] - ref) / (abs(ref) + 1e-6))
            assert abs(got[k] - ref) <= 3e-2 * abs(ref) + 1e-2, (k, got[k], ref)
    total = sum(v for k, v in losses.items() if k in model.criterion.weight_dict)
    total.backward()
    dsig = grad_signature([("dfeat:res5", feats["res5"].grad)])["dfeat:res5"]
    dref = g["dfsig:res5"]
    ssig = grad_signature([("dfeat:src", srcs["src"].grad)])["dfeat:src"]
    sref = g["dfsig:src"]
    print("d(src) rel err", float(np.sqrt(np.mean((ssig[1:] - sref[1:]) ** 2)) / sref[0]), "norm ratio", float(ssig[0] / sref[0]), tuple(srcs["src"].shape))
    print("d(src) per image", [float(srcs["src"].grad[i].float().norm()) for i in range(2)], "total", float(srcs["src"].grad.float().norm()))
    print("d(res5) rel err", float(np.sqrt(np.mean((dsig[1:] - dref[1:]) ** 2)) / dref[0]), "norm ratio", float(dsig[0] / dref[0]))
    named = [(n, p.grad) for n, p in model.named_parameters() if p.requires_grad]
    # (the oracle's ResNet does not freeze: its 11 extra entries are the stem and res2 convolutions, frozen here by FREEZE_AT 2)
    extra = {k[5:] for k in g.files if k.startswith("gsig:")} - {n for n, _ in named}
    assert all(n.startswith(("detr.backbone.0.backbone.stem", "detr.backbone.0.backbone.res2")) for n in extra), sorted(extra)[:5]
    assert all("gsig:" + n in g.files for n, _ in named)
    sig = grad_signature(named)
    rel, ratio = {}, {}
    for n, v in sig.items():
        r = g["gsig:" + n]
        rel[n] = float(np.sqrt(np.mean((v[1:] - r[1:]) ** 2)) / (r[0] + 1e-30))
        ratio[n] = float(v[0] / (r[0] + 1e-30))
    # the first decoder layer attends over tgt = 0: its q / k gradients are mathematically zero (both sides hold rounding
    # residue), and query_embed's gradient is the sum of twelve attention inputs' near-cancelling terms - bounded apart
    special = {n for n in rel if n.startswith("detr.transformer.decoder.layers.0.self_attn.in_proj") or n == "detr.query_embed.weight"}
    rels = np.array(sorted(v for n, v in rel.items() if n not in special))
    worst_names = sorted((n for n in rel if n not in special), key=rel.get)[-5:]
    print("loss worst rel", worst, "grad rel err: median %.4f p90 %.4f max %.4f" % (np.median(rels), rels[int(0.9 * len(rels))], rels[-1]),
          [(n, round(rel[n], 3)) for n in worst_names], {n: round(rel[n], 3) for n in special})
    if os.environ.get("MI_T_DUMP"):
        for n, _ in named:
            print("GS %-70s rel %.3f ratio %.3f" % (n, rel[n], ratio[n]))
    # measured: median 0.046, p90 0.068, max 0.12 (bf16 activations / gradients through 50 + 12 layers, un-forced forward)
    assert np.median(rels) < 0.07 and rels[int(0.9 * len(rels))] < 0.1 and rels[-1] < 0.2
    assert all(0.9 < ratio[n] < 1.1 for n in rel if n not in special), sorted(ratio.items(), key=lambda kv: kv[1])[:3]
    assert rel["detr.query_embed.weight"] < 0.6
    ssrc = float(np.sqrt(np.mean((ssig[1:] - sref[1:]) ** 2)) / sref[0])
    assert ssrc < 0.1 and 0.95 < float(ssig[0] / sref[0]) < 1.05, ssrc
    model.eval()
    with torch.no_grad():
        out = model.detr(model.preprocess_image(inputs))
    lg, bx = out["pred_logits"].float().cpu().numpy(), out["pred_boxes"].float().cpu().numpy()
    print("eval max abs: logits", np.abs(lg - g["eval_logits"]).max(), "boxes", np.abs(bx - g["eval_boxes"]).max())
    assert np.abs(lg - g["eval_logits"]).max() < 0.25 and np.abs(bx - g["eval_boxes"]).max() < 3e-2


@pytest.mark.parametrize("peak", [1.0, 2.83])
def test_detr_r50_real_size_gradients_with_the_forward_state_pinned(monkeypatch, peak):
    """configs[3] to the YOLOX standard (tests/test_gpu_parity_bench.py): DETR-R50 at its real size (6 + 6 layers, 100 queries,
    an 800 x 1333 + 768 x 1205 padded batch), EVERY trainable parameter's gradient against an fp32 restatement whose forward
    is pinned to the HIP network's own activations (teacher forcing: every trainable conv output of res3 .. res5, the input
    projection, all six encoder and all six decoder layer outputs), so that both sides differentiate the same function at the
    same point and the comparison measures the backward kernels - not which ReLU gate or attention row a bf16 rounding tipped.
    Two segments, each fed the HIP network's own upstream gradient: [input_proj + transformer + heads] from d loss / d (class
    logits, boxes) of all six levels (oracle/detr_net_oracle.py, pinned to the reference's Transformer by the CPU suite), and
    [ResNet-50 res3 .. res5] from d loss / d res5 (oracle/resnet_oracle.py).  Replaces the un-forced norm + 8-projection
    fingerprints (median 4.6 %, max 12 %) as the gradient bound of this configuration: cosine >= 0.999, rel L2 <= 0.05."""
    import detr_net_oracle as DN
    import resnet_oracle as R
    from yolov7_d2_amd.modeling.resnet import Conv2d
    from yolov7_d2_amd.modeling.transformer import TransformerDecoderLayer, TransformerEncoderLayer
    monkeypatch.setenv("MI_RESNET_BLOCK_FN", "0")         # per-convolution autograd nodes: their outputs can be hooked
    cfg = M.detr_r50_cfg(device=DEV)
    cfg.MODEL.DETR.DROPOUT = 0.0
    model = M.build_model(cfg)
    sd = model.state_dict()
    model.load_state_dict(seeded_tensor_dict({k: v.shape for k, v in sd.items()}, seed=207), strict=False)
    with torch.no_grad():
        model.detr.input_proj.weight.mul_(1e-3)           # (tokens of trained magnitude, as in the real-size golden test)
        if peak != 1.0:
            # PEAKED attention (a trained network's regime; random initialisation is near-uniform): the q and k rows of every
            # in-projection scaled, i.e. the attention logits by peak^2
            for n_, p_ in model.named_parameters():
                if n_.endswith("in_proj_weight"):
                    p_[: 2 * p_.shape[1]].mul_(peak)
    model.train()
    inputs = _inputs(synth_detr_batch(seed=211, sizes=((800, 1333), (768, 1205))))
    cpu = lambda t: t.detach().float().cpu()
    caps, grads, keep = {}, {}, {}
    bb = model.detr.backbone[0].backbone
    BP = "detr.backbone.0.backbone."
    for name, mod in bb.named_modules():
        if isinstance(mod, Conv2d) and name.startswith(("res3", "res4", "res5")):
            mod.register_forward_hook(lambda m, i, o, name=name: caps.__setitem__(name, cpu(o)))

    def feat_hook(m, i, o):
        o["res5"].retain_grad()
        keep["res5"] = o["res5"]
    bb.register_forward_hook(feat_hook)
    bb.res3.register_forward_pre_hook(lambda m, args: keep.__setitem__("res2", args[0]))      # (the frozen stages' output)

    def tr_pre(m, args):
        args[0].retain_grad()
        keep["src"], keep["mask"], keep["pos"] = args[0], args[1], args[3]
    model.detr.transformer.register_forward_pre_hook(tr_pre)
    for i, layer in enumerate(model.detr.transformer.encoder.layers):
        assert isinstance(layer, TransformerEncoderLayer)
        layer.register_forward_hook(lambda m, inp, o, i=i: caps.__setitem__(f"enc.{i}", cpu(o)))
    for i, layer in enumerate(model.detr.transformer.decoder.layers):
        assert isinstance(layer, TransformerDecoderLayer)
        layer.register_forward_hook(lambda m, inp, o, i=i: caps.__setitem__(f"dec.{i}", cpu(o)))

    def out_hook(m, i, o):
        levels = list(o["aux_outputs"]) + [dict(pred_logits=o["pred_logits"], pred_boxes=o["pred_boxes"])]
        for lv in levels:
            lv["pred_logits"].retain_grad(); lv["pred_boxes"].retain_grad()
        keep["levels"] = levels
    model.detr.register_forward_hook(out_hook)
    losses = model(inputs)
    total = sum(v for k, v in losses.items() if k in model.criterion.weight_dict)
    total.backward()
    torch.cuda.synchronize()
    hip = {n: cpu(p.grad) for n, p in model.named_parameters() if p.requires_grad}
    q = lambda t: t + (t.to(torch.bfloat16).float() - t).detach()

    # ---- segment 1: input_proj + transformer + heads, forward pinned, backward from the HIP d / d (logits, boxes)
    osd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items() if k.startswith("detr.") and not k.startswith(BP)}
    tkeys = [k for k in osd if k in hip]
    for k in tkeys:
        osd[k].requires_grad_(True)
    feat = cpu(keep["res5"]).requires_grad_(True)
    force = {k: v for k, v in caps.items() if k.startswith(("enc.", "dec."))}
    force["src"] = cpu(keep["src"])
    assert len(force) == 13
    # query_embed enters every decoder layer three times (self-attention query and key, cross-attention query): per-use
    # copies give the 18 TERMS of its gradient
    qes = [tuple(osd["detr.query_embed.weight"].detach().clone().requires_grad_(True) for _ in range(3)) for _ in range(6)]
    ref = DN.detr_after_backbone(osd, feat, keep["mask"].cpu(), cpu(keep["pos"]), nhead=8, prefix="detr.", quant=q, force=force,
                                 query_embed_layers=qes)
    dl = torch.stack([cpu(lv["pred_logits"].grad) for lv in keep["levels"]])
    db = torch.stack([cpu(lv["pred_boxes"].grad) for lv in keep["levels"]])
    lg_hip = torch.stack([cpu(lv["pred_logits"]) for lv in keep["levels"]])
    bx_hip = torch.stack([cpu(lv["pred_boxes"]) for lv in keep["levels"]])
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    print("forced forward: logits rel %.2e boxes rel %.2e" % (rel(lg_hip, ref["logits"].detach()), rel(bx_hip, ref["boxes"].detach())))
    assert rel(lg_hip, ref["logits"].detach()) < 2e-2 and rel(bx_hip, ref["boxes"].detach()) < 2e-2
    torch.autograd.backward([ref["logits"], ref["boxes"]], [dl, db])

    def table(keys, get_ref):
        rows = []
        for k in keys:
            a, b = hip[k].flatten(), get_ref(k).flatten()
            na, nb = float(a.norm()), float(b.norm())
            cos = float(torch.dot(a, b) / (na * nb + 1e-30))
            rows.append((k, cos, float((a - b).norm() / (nb + 1e-30)), na, nb))
        return rows
    qflat = [t for tri in qes for t in tri]
    osd["detr.query_embed.weight"].grad = sum(t.grad for t in qflat)
    rows = table(tkeys, lambda k: osd[k].grad)
    # the first decoder layer attends over tgt = 0: its q / k in-projection gradients are mathematically zero (both sides
    # hold rounding residue) - compared against the same parameter's v rows instead.  query_embed: bounded below against
    # the size of its six summed terms.  linear1 (the layer in front of the FFN's ReLU): the hidden activations are not a
    # pinned site, so gates of near-zero pre-activations are each side's own - 0.998 / 0.07 there
    special = {k for k in tkeys if "decoder.layers.0.self_attn.in_proj" in k} | {"detr.query_embed.weight"}
    lim = lambda k: (0.998, 0.07) if ".linear1." in k else (0.999, 0.05)
    bad = [(k, round(c, 5), round(r, 4)) for k, c, r, na, nb in rows if k not in special and (c < lim(k)[0] or r > lim(k)[1])]
    worst = sorted((r for r in rows if r[0] not in special), key=lambda r: r[1])[:4]
    print("transformer segment: %d tensors, worst cosines" % len(rows), [(k[-46:], round(c, 5), round(r, 4)) for k, c, r, _, _ in worst])
    assert not bad, bad[:8]
    qerr = float((hip["detr.query_embed.weight"] - osd["detr.query_embed.weight"].grad).norm())
    qterms = sum(float(t.grad.norm()) for t in qflat)
    print("query_embed: |error| %.3e, |sum| %.3e, sum of the 18 terms' norms %.3e" % (qerr, float(osd["detr.query_embed.weight"].grad.norm()), qterms),
          "per layer (self q, self k, cross q):", [[round(float(t.grad.norm()), 5) for t in tri] for tri in qes])
    # d / d query_embed and the q / k rows of every in-projection are what flows through dS = P o (dP - rowsum(dO o O)): at
    # random initialisation the values of a row's keys are alike, dP ~ rowsum (|delta| is 10 - 100 x |dP - delta|) and the
    # difference cancels.  Rounds 3 - 4 took rowsum from the bf16 O: its 2^-9 rounding, coherent over the row, came out of the
    # difference as 3 - 15 % (encoder) / 18 - 60 % (cross-attention) errors of these rows.  Round 5: the forward also writes O
    # in fp32 for the backward's delta (mi_mha_fwd_dropout_o32; offline analysis of device operands: tools/attn_bwd_error.py,
    # dq rel 0.78 -> 0.0014) - measured here 1.7 - 7.6 % at random initialisation and 1.1 - 3.7 % with peaked attention
    # (logits x 8), on a component that is 0.3 - 10 % of its tensor.  Held to cosine 0.99 / rel 0.1.
    E = 256
    qk = []
    for k in tkeys:
        if k.endswith("in_proj_weight") and "decoder.layers.0.self_attn" not in k:
            a, b = hip[k][: 2 * E].flatten(), osd[k].grad[: 2 * E].flatten()
            qk.append((k[len("detr.transformer."):-len(".in_proj_weight")], float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)),
                       float((a - b).norm() / (b.norm() + 1e-30)), float(b.norm() / osd[k].grad[2 * E:].norm())))
    print("q / k rows of the in-projections (name, cosine, rel L2, |qk rows| / |v rows|):", [(n, round(c, 4), round(r, 3), round(f, 4)) for n, c, r, f in qk])
    # the decoder's SELF-attention q / k gradients are ~0 on both sides (100 near-identical queries: exactly uniform
    # attention) - there the HIP rows must be noise-small against the value rows
    for n, c, r, f in qk:
        if f > 1e-3:
            assert c > 0.99 and r < 0.1, (n, c, r, f)
    for k in tkeys:
        if k.endswith("self_attn.in_proj_weight") and ".decoder." in k and ".layers.0." not in k:      # (layer 0: below)
            assert float(hip[k][: 2 * E].norm()) < 1e-2 * float(hip[k][2 * E:].norm()), k
    assert qerr < 0.1 * qterms
    for k in special - {"detr.query_embed.weight"}:
        E = 256
        a, b = hip[k][2 * E:].flatten(), osd[k].grad[2 * E:].flatten()       # the value projection rows
        if float(b.norm()) > 1e-12:
            assert float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)) > 0.999, k
        assert float(hip[k][: 2 * E].norm()) < 3e-2 * float(hip[k.replace("layers.0", "layers.1")][: 2 * E].norm()), k
    dres5 = cpu(keep["res5"].grad)
    c5 = float(torch.dot(dres5.flatten(), feat.grad.flatten()) / (dres5.norm() * feat.grad.norm() + 1e-30))
    print("d res5: cosine %.5f rel %.4f" % (c5, rel(dres5, feat.grad)))
    # (d res5 = input_proj^T d src, and d src has crossed six encoder layers' q / k paths, see above: measured 0.9988 / 0.049)
    assert c5 > 0.998 and rel(dres5, feat.grad) < 0.07

    # ---- segment 2: ResNet-50 res3 .. res5 (FREEZE_AT 2), every conv output pinned, backward from the HIP d / d res5
    bsd = {k[len(BP):]: v.detach().float().cpu().clone() for k, v in model.state_dict().items() if k.startswith(BP)}
    bkeys = [k for k in hip if k.startswith(BP)]
    assert len(bkeys) == 42 and len([k for k in caps if k.startswith("res")]) == 42
    for k in bkeys:
        bsd[k[len(BP):]].requires_grad_(True)
    bref = R.forward(bsd, None, quant=q, force={k: v for k, v in caps.items() if k.startswith("res")},
                     start=("res3", cpu(keep["res2"])))
    assert rel(cpu(keep["res5"]), bref["res5"].detach()) < 1e-3
    bref["res5"].backward(dres5)
    brows = table(bkeys, lambda k: bsd[k[len(BP):]].grad)
    bbad = [(k, round(c, 5), round(r, 4)) for k, c, r, na, nb in brows if c < 0.999 or r > 0.05]
    print("backbone segment: 42 tensors, worst cosines", [(k[-30:], round(c, 5), round(r, 4)) for k, c, r, _, _ in sorted(brows, key=lambda r: r[1])[:4]])
    assert not bbad, bbad[:8]
    assert len(rows) + len(brows) == len(hip)              # every trainable parameter of the model was compared
