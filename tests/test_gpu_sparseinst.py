"""GPU (-m gpu): SparseInst (BASELINE.json config 5) against tests/golden/sparseinst.npz = the reference's OWN
InstanceContextEncoder + GroupIAMDecoder + SparseInstCriterion / SparseInstMatcher executed by path on seeded ResNet
features and bitmask targets (oracle/gen_golden.py::gold_sparseinst): encoder output, class logits / objectness / masks,
the Hungarian matching (indices exact), the four weighted losses, gradient norms of every parameter and two full
gradients; the op-level pieces (bilinear resize, pixel-sum outer product, mask-loss kernels) against torch; and one
end-to-end training step + inference of the registered META_ARCH around the ResNet-50."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import yolov7_d2_amd as M
from yolov7_d2_amd.d2shim import Instances
from yolov7_d2_amd.modeling import sparseinst as S
from gen_golden_inputs import seeded_tensor_dict, sparseinst_spread, synth_sparseinst_case

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_bilinear_resize_and_pixel_outer_against_torch():
    g = torch.Generator().manual_seed(0)
    bf = lambda t: t.to(torch.bfloat16).float()
    # (the last four: small maps with wide windows - the pyramid-pooling priors and the 20 -> 80 fusion resize - whose backward
    #  runs the block-per-pixel kernel; 104 channels = 13 channel groups, which do not divide the block)
    for (N, C, H, W, Ho, Wo) in ((2, 64, 10, 13, 20, 26), (1, 32, 4, 5, 16, 20), (2, 104, 16, 20, 32, 40), (1, 64, 1, 1, 6, 7),
                                 (2, 128, 1, 1, 20, 20), (2, 128, 6, 6, 20, 20), (2, 104, 3, 3, 20, 20), (2, 256, 20, 20, 80, 80)):
        x = bf(torch.randn(N, C, H, W, generator=g))
        xr = x.clone().requires_grad_(True)
        ref = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=False)
        go = bf(torch.randn(ref.shape, generator=g))
        ref.backward(go)
        xd = x.to(DEV, torch.bfloat16).requires_grad_(True)
        y = S.resize_bilinear(xd, (Ho, Wo))
        assert _rel(y, ref.detach()) < 5e-3
        y.backward(go.to(DEV, torch.bfloat16))
        assert _rel(xd.grad, xr.grad) < 5e-3
    a = bf(torch.rand(1000, 128, generator=g)); b = bf(torch.randn(1000, 256, generator=g))
    ad, bd = a.to(DEV, torch.bfloat16).requires_grad_(True), b.to(DEV, torch.bfloat16).requires_grad_(True)
    out = S._PixelOuterFn.apply(ad, bd)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = ar.t() @ br
    assert _rel(out, ref.detach()) < 2e-3
    gg = torch.randn(ref.shape, generator=g)
    ref.backward(gg); out.backward(gg.to(DEV))
    assert _rel(ad.grad, ar.grad) < 1e-2 and _rel(bd.grad, br.grad) < 1e-2


def _build(cfg):
    shapes = {n: types.SimpleNamespace(channels=c, stride=s) for n, c, s in (("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    enc = S.InstanceContextEncoder(cfg, shapes)
    dec = S.GroupIAMDecoder(cfg)
    net = torch.nn.ModuleDict(dict(encoder=enc, decoder=dec))
    net.load_state_dict(sparseinst_spread(seeded_tensor_dict({k: v.shape for k, v in net.state_dict().items()}, seed=303)))
    return net.to(DEV)


def test_encoder_decoder_criterion_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "sparseinst.npz"))
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    net = _build(cfg)
    assert sorted(dict(net.named_parameters()).keys()) == [str(n) for n in g["param_names"]]     # the reference's keys
    crit = S.build_sparse_inst_criterion(cfg)
    feats, targets, input_shape = synth_sparseinst_case()
    fin = {k: v.to(DEV, torch.bfloat16).requires_grad_(True) for k, v in feats.items()}
    e = net["encoder"](fin)
    out = net["decoder"](e)
    assert _rel(e.detach()[:, ::8], g["enc_out"]) < 2e-2
    assert _rel(out["pred_logits"].detach(), g["pred_logits"]) < 3e-2
    assert _rel(out["pred_scores"].detach(), g["pred_scores"]) < 3e-2
    assert _rel(out["pred_masks"].detach()[:, ::5, ::2, ::2], g["pred_masks"]) < 3e-2
    tg = [dict(labels=t["labels"].to(DEV), masks=t["masks"].to(DEV)) for t in targets]
    indices, mm = crit.matcher(out, tg, input_shape)
    # On a random-weight network the 100 instances are near copies of each other (the reference's own dice scores tie
    # to the 5th digit), so WHICH of the tied queries is picked is rounding noise: check optimality instead of identity -
    # the kernel's assignment must reach the objective of scipy's on the reference's fp32 cost (exact-index equality is
    # checked on a decisive case in test_matcher_indices_exact_on_a_decisive_case)
    from scipy.optimize import linear_sum_assignment
    pm = out["pred_masks"].detach().float()
    for b, (i, j) in enumerate(indices):
        M_ = len(tg[b]["labels"])
        tm = mm.tgt[b * mm.cap: b * mm.cap + M_]                  # (PackedMaskTargets: image b's rows start at b * cap)
        sg = pm[b].flatten(1).sigmoid()
        score = 2 * sg @ tm.t() / ((sg * sg).sum(-1)[:, None] + (tm * tm).sum(-1)[None] + 1e-4)
        Cm = (score ** 0.8) * (out["pred_logits"][b].detach().float().sigmoid()[:, tg[b]["labels"]] ** 0.2)
        ri, rj = linear_sum_assignment(Cm.cpu().numpy(), maximize=True)
        best = float(Cm.cpu().numpy()[ri, rj].sum())
        mine = float(Cm[i, j].sum())
        assert sorted(j.cpu().tolist()) == list(range(M_)) and len(set(i.cpu().tolist())) == M_
        assert mine >= best - 2e-3 * abs(best), (b, mine, best)
    losses = crit(out, tg, input_shape)
    got = {k: float(v.detach()) for k, v in losses.items()}
    print({k: (round(got[k], 4), round(float(g["loss:" + k]), 4)) for k in got})
    for k in got:
        ref = float(g["loss:" + k])
        assert abs(got[k] - ref) <= 3e-2 * abs(ref) + 1e-3, (k, got[k], ref)
    sum(losses.values()).backward()
    for k, v in fin.items():
        assert abs(float(v.grad.float().norm()) - float(g["dfeat_norm:" + k])) < 0.1 * float(g["dfeat_norm:" + k]), k
    params = dict(net.named_parameters())
    ratio = np.array([float(params[str(n)].grad.float().norm()) / (gn + 1e-12) for n, gn in zip(g["param_names"], g["param_grad_norms"])])
    print("parameter gradient norm ratio: min %.3f max %.3f" % (ratio.min(), ratio.max()))
    # (the matched queries may differ from the reference's among the tied instances: the heads' gradients move a little)
    assert ratio.min() > 0.8 and ratio.max() < 1.3 and abs(float(np.median(ratio)) - 1.0) < 0.05
    for n in ("decoder.inst_branch.mask_kernel.weight", "encoder.fusion.weight"):
        a, b = params[n].grad.float().cpu().flatten(), torch.as_tensor(g["g:" + n]).flatten()
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos > 0.99, (n, cos)


def test_matcher_indices_exact_on_a_decisive_case():
    """SparseInstMatcher (dice^0.8 * prob^0.2, maximise) on predictions where every target has a clearly best query:
    indices identical to scipy.optimize.linear_sum_assignment on the reference's formula"""
    from scipy.optimize import linear_sum_assignment
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    matcher = S.SparseInstMatcher(cfg)
    _, targets, input_shape = synth_sparseinst_case(seed=5)
    g = torch.Generator().manual_seed(6)
    B, N, Ho, Wo = 2, 100, input_shape[0] // 4, input_shape[1] // 4
    tg = [dict(labels=t["labels"].to(DEV), masks=t["masks"].to(DEV)) for t in targets]
    logits = torch.randn(B, N, 80, generator=g) * 2 - 2
    masks = torch.randn(B, N, Ho, Wo, generator=g) * 0.5 - 3
    for b, t in enumerate(targets):
        small = F.interpolate(F.pad(t["masks"], (0, input_shape[1] - t["masks"].shape[2], 0, input_shape[0] - t["masks"].shape[1]))[None],
                              size=(Ho, Wo), mode="bilinear", align_corners=False)[0]
        for k in range(small.shape[0]):
            q = int(torch.randint(0, N, (1,), generator=g))
            masks[b, q] += 7 * small[k]                      # query q predicts target k's mask
            logits[b, q, t["labels"][k]] += 4
    nhwc = torch.zeros(B, Ho, Wo, 128)
    nhwc[..., :N] = masks.permute(0, 2, 3, 1)
    out = {"pred_logits": logits.to(DEV), "_masks_nhwc": nhwc.to(DEV, torch.bfloat16)}
    indices, mm = matcher(out, tg, input_shape)
    for b, (i, j) in enumerate(indices):
        M_ = len(tg[b]["labels"])
        tm = mm.tgt[b * mm.cap: b * mm.cap + M_].cpu()
        sg = masks[b].to(torch.bfloat16).float().flatten(1).sigmoid()
        score = 2 * sg @ tm.t() / ((sg * sg).sum(-1)[:, None] + (tm * tm).sum(-1)[None] + 1e-4)
        Cm = (score ** 0.8) * (logits[b].sigmoid()[:, targets[b]["labels"]] ** 0.2)
        ri, rj = linear_sum_assignment(Cm.numpy(), maximize=True)
        assert i.cpu().tolist() == ri.tolist() and j.cpu().tolist() == rj.tolist(), (b, i, j, ri, rj)


def test_sparseinst_meta_arch_step_and_inference():
    torch.manual_seed(0)
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    cfg.MODEL.YOLO.CONF_THRESHOLD = 0.005
    model = M.build_model(cfg)
    assert M.META_ARCH_REGISTRY.get("SparseInst") is M.SparseInst
    _, targets, _ = synth_sparseinst_case(seed=9, H=192, W=224)
    gen = torch.Generator().manual_seed(1)
    inputs = []
    for t in targets:
        h, w = t["size"]
        inst = Instances((h, w), gt_classes=t["labels"], gt_masks=t["masks"])
        inputs.append(dict(image=torch.randint(0, 256, (3, h, w), generator=gen).float(), instances=inst, height=h, width=w))
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    assert any(n.startswith("backbone.stem") for n, p in model.named_parameters() if p.requires_grad)    # FREEZE_AT 0
    opt = torch.optim.AdamW(params, lr=5e-5, weight_decay=0.05)
    hist = []
    for it in range(3):
        losses = model(inputs)
        assert set(losses) == {"loss_ce", "loss_mask", "loss_dice", "loss_objectness"}
        total = sum(losses.values())
        opt.zero_grad()
        total.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
        opt.step()
        hist.append(float(total))
    print("sparseinst loss:", [round(h, 3) for h in hist])
    assert np.isfinite(hist).all() and hist[-1] < hist[0]
    model.eval()
    with torch.no_grad():
        res = model(inputs)
    assert len(res) == 2
    for r, t in zip(res, targets):
        inst = r["instances"]
        assert inst.image_size == t["size"]
        if len(inst):
            assert inst.pred_masks.shape[1:] == t["size"] and inst.pred_masks.dtype == torch.bool
            assert inst.scores.shape[0] == inst.pred_classes.shape[0] == inst.pred_masks.shape[0]


def test_sparseinst_inference_against_reference_golden(golden_dir):
    """SparseInst.inference (meta_arch/sparseinst.py:173-234 + rescoring_mask :24-27) against the REFERENCE'S OWN method run
    by path on the same decoder outputs (golden sparseinst_inference.npz): which queries survive the class threshold and
    their classes exactly, the maskness-rescored scores to fp32 rounding, the thresholded masks after the two bilinear
    resizes (padded batch -> crop -> requested size; one image up-, one down-scaled) pixel for pixel up to the handful
    whose interpolated value sits within rounding of the threshold"""
    import types
    g = np.load(os.path.join(golden_dir, "sparseinst_inference.npz"))
    out = {k: torch.from_numpy(g[k]).to(DEV) for k in ("pred_logits", "pred_scores", "pred_masks")}
    stub = types.SimpleNamespace(cls_threshold=float(g["cls_threshold"]), mask_threshold=float(g["mask_threshold"]))
    batched_inputs = [dict(height=int(h), width=int(w)) for h, w in g["out_sizes"]]
    image_sizes = [tuple(int(v) for v in s) for s in g["image_sizes"]]
    res = M.SparseInst.inference(stub, out, batched_inputs, tuple(int(v) for v in g["max_shape"]), image_sizes)
    assert len(res) == 2
    for b, r in enumerate(res):
        assert r.image_size == tuple(int(v) for v in g["out_sizes"][b])
        assert np.array_equal(r.pred_classes.cpu().numpy(), g[f"classes{b}"])
        np.testing.assert_allclose(r.scores.cpu().numpy(), g[f"scores{b}"], rtol=2e-5, atol=1e-7)
        ref = np.unpackbits(g[f"masks{b}"], axis=-1)[..., : r.pred_masks.shape[-1]].astype(bool)
        got = r.pred_masks.cpu().numpy()
        assert got.shape == ref.shape and got.dtype == np.bool_
        assert (got != ref).mean() < 1e-4, float((got != ref).mean())
        assert np.abs(got.reshape(len(got), -1).sum(1) - g[f"mask_area{b}"]).max() <= 3


def test_encoder_decoder_criterion_at_real_size_against_reference_golden(golden_dir):
    """configs[4] at its real size - the res3 / res4 / res5 maps of a 640 x 640 batch (80 x 80 x 512 ... 20 x 20 x 2048),
    100 instance queries - against the reference's own encoder / decoder / criterion in fp32
    (gen_golden.py::gold_sparseinst_real): class logits and objectness in full, encoder output / mask logits / d features /
    EVERY parameter gradient through grad_signature fingerprints (norm + 8 seeded +-1 projections), the four losses.  The
    assignment is teacher-forced to the reference's (our matcher's own answer is checked for optimality at the small size and
    for exact indices on a decisive case; here its agreement is only reported)."""
    from gen_golden_inputs import grad_signature
    g = np.load(os.path.join(golden_dir, "sparseinst_real.npz"))
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    shapes = {n: types.SimpleNamespace(channels=c, stride=s) for n, c, s in (("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    net = torch.nn.ModuleDict(dict(encoder=S.InstanceContextEncoder(cfg, shapes), decoder=S.GroupIAMDecoder(cfg)))
    net.load_state_dict(sparseinst_spread(seeded_tensor_dict({k: v.shape for k, v in net.state_dict().items()}, seed=307)))
    net = net.to(DEV)
    crit = S.build_sparse_inst_criterion(cfg)
    feats, targets, input_shape = synth_sparseinst_case(seed=311, B=2, H=640, W=640)
    fin = {k: v.to(DEV, torch.bfloat16).requires_grad_(True) for k, v in feats.items()}
    e = net["encoder"](fin)
    out = net["decoder"](e)

    def frel(name, t, key):
        v, r = grad_signature([(name, t)])[name], g[key]
        return float(np.sqrt(np.mean((v[1:] - r[1:]) ** 2)) / r[0]), float(v[0] / r[0])
    fwd = {"enc_out": frel("enc_out", e.detach().float().cpu(), "sig:enc_out"),
           "pred_masks": frel("pred_masks", out["pred_masks"].detach().float().cpu(), "sig:pred_masks"),
           "pred_logits": _rel(out["pred_logits"].detach(), g["pred_logits"]), "pred_scores": _rel(out["pred_scores"].detach(), g["pred_scores"])}
    print("forward", fwd)
    assert fwd["enc_out"][0] < 2e-2 and fwd["pred_masks"][0] < 3e-2 and fwd["pred_logits"] < 3e-2 and fwd["pred_scores"] < 3e-2
    tg = [dict(labels=t["labels"].to(DEV), masks=t["masks"].to(DEV)) for t in targets]
    own = crit.matcher
    agree = {}

    class _Forced(torch.nn.Module):
        """the criterion asks its matcher for (match_q, match_t, nmatch) on the device: hand it the reference's pairs"""
        def match_packed(self, outputs, pk):
            mq, mt, nm = own.match_packed(outputs, pk)
            n = nm.tolist()
            mine = [(mq[b, : n[b]], mt[b, : n[b]]) for b in range(pk.B)]
            ref = [(torch.from_numpy(g[f"match_i{b}"]).to(DEV), torch.from_numpy(g[f"match_j{b}"]).to(DEV)) for b in range(pk.B)]
            agree["same"] = sum(len({(int(a), int(c)) for a, c in zip(i.tolist(), j.tolist())} & {(int(a), int(c)) for a, c in zip(ri.tolist(), rj.tolist())})
                                for (i, j), (ri, rj) in zip(mine, ref))
            agree["total"] = sum(len(ri) for ri, _ in ref)
            fq, ft, fn = torch.zeros_like(mq), torch.zeros_like(mt), torch.zeros_like(nm)
            for b, (ri, rj) in enumerate(ref):
                fq[b, : len(ri)], ft[b, : len(ri)], fn[b] = ri, rj, len(ri)
            return fq, ft, fn
    crit.matcher = _Forced()
    losses = crit(out, tg, input_shape)
    crit.matcher = own
    print("matcher agreement on own outputs: %d of %d pairs" % (agree["same"], agree["total"]))
    got = {k: float(v.detach()) for k, v in losses.items()}
    print({k: (round(got[k], 4), round(float(g["loss:" + k]), 4)) for k in got})
    for k in got:
        ref = float(g["loss:" + k])
        assert abs(got[k] - ref) <= 3e-2 * abs(ref) + 1e-3, (k, got[k], ref)
    sum(losses.values()).backward()
    drel = {k: frel("dfeat:" + k, v.grad.float().cpu(), "sig:dfeat:" + k) for k, v in fin.items()}
    prel = {n: frel(n, p.grad.float().cpu(), "gsig:" + n) for n, p in net.named_parameters()}
    rels = np.array(sorted(v[0] for v in prel.values()))
    worst = sorted(prel.items(), key=lambda kv: -kv[1][0])[:5]
    print("d features", drel)
    print("param grad rel err: median %.4f p90 %.4f max %.4f" % (np.median(rels), rels[int(0.9 * len(rels))], rels[-1]), worst)
    # measured: d res3 0.10, d res4 / res5 0.033 (norm ratios 1.001); parameter gradients median 0.016, p90 0.037, max 0.13
    # (the pooled 1x1 PPM stage); the fingerprint estimate itself carries ~25 % relative noise at 8 projections
    assert all(v[0] < 0.15 and 0.95 < v[1] < 1.05 for v in drel.values()), drel
    assert np.median(rels) < 0.04 and rels[int(0.9 * len(rels))] < 0.08 and rels[-1] < 0.2


def _si_inputs(seed, sizes, counts=None):
    """batched_inputs of the SparseInst meta-arch: uint8-valued images + bitmask instances (rectangles)"""
    gen = torch.Generator().manual_seed(seed)
    out = []
    for k, (h, w) in enumerate(sizes):
        n = counts[k] if counts is not None else int(torch.randint(1, 6, (1,), generator=gen))
        masks = torch.zeros(n, h, w)
        for j in range(n):
            y0, x0 = int(torch.randint(0, h - 40, (1,), generator=gen)), int(torch.randint(0, w - 40, (1,), generator=gen))
            hh, ww = int(torch.randint(24, min(96, h - y0), (1,), generator=gen)), int(torch.randint(24, min(96, w - x0), (1,), generator=gen))
            masks[j, y0: y0 + hh, x0: x0 + ww] = 1
        inst = Instances((h, w), gt_classes=torch.randint(0, 80, (n,), generator=gen), gt_masks=masks)
        out.append(dict(image=torch.randint(0, 256, (3, h, w), generator=gen).float(), instances=inst, height=h, width=w))
    return out


def test_sparseinst_device_half_never_reads_the_host_and_is_captured():
    """round 4: SparseInstCriterion keeps the assignment and the matched-pair bookkeeping on the device (rounds 2-3 read
    `nmatch.tolist()` / `.item()` three times per step), so (1) forward_prepared + backward run under
    torch.cuda.set_sync_debug_mode("error") - any synchronising call raises - and (2) the whole step is ONE hipGraph
    (graph_step.GraphedTrainStep): graphed == eager over three optimizer steps, the second / third batch - other images,
    other sizes inside the same padded shape, other instance counts, one image without instances - replay the capture."""
    from yolov7_d2_amd.graph_step import GraphedTrainStep
    from yolov7_d2_amd.optim import MultiTensorAdamW
    import copy
    torch.manual_seed(0)
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    eager = M.build_model(cfg)
    graphed = copy.deepcopy(eager)
    eager.train(); graphed.train()
    mk = lambda m: MultiTensorAdamW([p for p in m.parameters() if p.requires_grad], lr=5e-5, weight_decay=0.05)
    oe, og = mk(eager), mk(graphed)
    batches = [_si_inputs(1, ((192, 224), (160, 200))), _si_inputs(2, ((180, 210), (192, 224)), counts=(3, 0)),
               _si_inputs(3, ((170, 224), (192, 200)))]
    assert len({eager.batch_key(b) for b in batches}) == 1
    # (1) no synchronisation in the device half (prepare_batch is the host half and may copy)
    st = eager.prepare_batch(batches[0])
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        losses = eager.forward_prepared(st)
        sum(losses.values()).backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    eager.zero_grad(set_to_none=True)
    # (2) graphed == eager
    step = GraphedTrainStep(graphed, og)
    try:
        for it, b in enumerate(batches):
            losses = eager(b)
            total = sum(losses.values())
            oe.zero_grad(set_to_none=True)
            total.backward()
            oe.step()
            out = step(b)
            assert len(step.graphs) == 1
            for k, v in losses.items():
                torch.testing.assert_close(out[k].float(), v.detach().float(), rtol=3e-3, atol=3e-3, msg=f"step {it} {k}")
        torch.cuda.synchronize()
        for (n, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
            if p.requires_grad:
                d = float((p.detach() - q.detach()).abs().max())
                assert d <= 1.0e-4, (n, d)           # three AdamW steps of lr 5e-5 move a weight by <= 1.5e-4
    finally:
        step.close()


def test_sparseinst_captured_step_at_the_bench_size_equals_eager():
    """configs[4] at the size bench.py runs (B = 8, 640 x 640, up to 10 instances per image): captured == eager over four
    optimizer steps, parameters identical.  The small test above never saw what this size shows: memset nodes losing their
    ordering on later replays of the large graph, and torch's multi-block reductions with them (profiles/
    r04_graph_memset_finding.txt) - with either back in the captured path this test fails from the second step on."""
    from yolov7_d2_amd.graph_step import GraphedTrainStep
    from yolov7_d2_amd.optim import MultiTensorAdamW
    import copy
    torch.manual_seed(0)
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    eager = M.build_model(cfg)
    graphed = copy.deepcopy(eager)
    eager.train(); graphed.train()
    mk = lambda m: MultiTensorAdamW([p for p in m.parameters() if p.requires_grad], lr=5e-5, weight_decay=0.05)
    oe, og = mk(eager), mk(graphed)
    b = [dict(x, image=x["image"].to(DEV)) for x in _si_inputs(11, [(640, 640)] * 8, counts=(3, 10, 1, 7, 5, 2, 9, 4))]
    step = GraphedTrainStep(graphed, og)
    try:
        for it in range(4):
            losses = eager(b)
            total = sum(losses.values())
            oe.zero_grad(set_to_none=True)
            total.backward()
            oe.step()
            out = step(b)
            for k, v in losses.items():
                torch.testing.assert_close(out[k].float(), v.detach().float(), rtol=1e-4, atol=1e-4, msg=f"step {it} {k}")
        torch.cuda.synchronize()
        worst = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(eager.parameters(), graphed.parameters()))
        assert worst <= 1e-6, worst
    finally:
        step.close()


def test_encoder_decoder_real_size_gradients_with_the_forward_state_pinned(monkeypatch):
    """configs[4] to the YOLOX standard: SparseInst's encoder + G-IAM decoder at their real size (the res3 / res4 / res5 maps of
    a 640 x 640 batch, 100 instance queries), EVERY parameter's gradient against the fp32 restatement
    (oracle/sparseinst_net_oracle.py, pinned to the reference's own modules by the CPU suite) with the forward PINNED to the
    HIP network's own activations - all 22 convolution outputs (the ReLU gates with them) - and the backward started from the
    HIP network's own d loss / d (class logits, objectness, mask logits).  Both sides then differentiate the same function at
    the same point.  Replaces the un-forced fingerprints (median 1.6 %, max 13 %) as this configuration's gradient bound:
    cosine >= 0.999, rel L2 <= 0.05 on every tensor."""
    import sparseinst_net_oracle as SN
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    shapes = {n: types.SimpleNamespace(channels=c, stride=s) for n, c, s in (("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    net = torch.nn.ModuleDict(dict(encoder=S.InstanceContextEncoder(cfg, shapes), decoder=S.GroupIAMDecoder(cfg)))
    net.load_state_dict(sparseinst_spread(seeded_tensor_dict({k: v.shape for k, v in net.state_dict().items()}, seed=307)))
    net = net.to(DEV)
    crit = S.build_sparse_inst_criterion(cfg)
    feats, targets, input_shape = synth_sparseinst_case(seed=311, B=2, H=640, W=640)
    fin = {k: v.to(DEV, torch.bfloat16).requires_grad_(True) for k, v in feats.items()}
    cpu = lambda t: t.detach().float().cpu()
    names = {m: n for n, m in net.named_modules() if isinstance(m, torch.nn.Conv2d)}
    caps, keep = {}, {}
    real_conv = S._conv

    def rec_conv(x, m, stride=1, relu=False):
        y = real_conv(x, m, stride, relu)
        caps[names[m]] = cpu(y)
        return y
    monkeypatch.setattr(S, "_conv", rec_conv)
    net["decoder"].inst_branch.register_forward_hook(lambda m, i, o: caps.__setitem__("decoder.inst_branch.iam_conv", cpu(o[3]() if callable(o[3]) else o[3])))   # (round 6: a lazy view of the padded map)
    e = net["encoder"](fin)
    out = net["decoder"](e)
    monkeypatch.setattr(S, "_conv", real_conv)
    for k in ("pred_logits", "pred_scores", "_masks_nhwc"):
        out[k].retain_grad()
    tg = [dict(labels=t["labels"].to(DEV), masks=t["masks"].to(DEV)) for t in targets]
    losses = crit(out, tg, input_shape)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    assert len(caps) == 22, sorted(caps)
    N = out["pred_logits"].shape[1]
    hip = {n: cpu(p.grad) for n, p in net.named_parameters()}
    q = lambda t: t + (t.to(torch.bfloat16).float() - t).detach()
    osd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    fr = {k: cpu(v).requires_grad_(True) for k, v in fin.items()}
    eo = SN.encoder(osd, fr, quant=q, force=caps)
    ro = SN.decoder(osd, eo, groups=cfg.MODEL.SPARSE_INST.DECODER.GROUPS, scale_factor=cfg.MODEL.SPARSE_INST.DECODER.SCALE_FACTOR,
                    quant=q, force=caps)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    fw = dict(enc=rel(cpu(e), eo.detach()), logits=rel(cpu(out["pred_logits"]), ro["pred_logits"].detach()),
              scores=rel(cpu(out["pred_scores"]), ro["pred_scores"].detach()),
              masks=rel(cpu(out["pred_masks"]), ro["pred_masks"].detach()))
    print("forced forward", {k: "%.2e" % v for k, v in fw.items()})
    assert all(v < 2e-2 for v in fw.values()), fw
    dm = cpu(out["_masks_nhwc"].grad).permute(0, 3, 1, 2)[:, :N]
    torch.autograd.backward([ro["pred_logits"], ro["pred_scores"], ro["pred_masks"]],
                            [cpu(out["pred_logits"].grad), cpu(out["pred_scores"].grad), dm])
    rows = []
    for n, g in hip.items():
        a, b = g.flatten(), osd[n].grad.flatten()
        rows.append((n, float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), rel(a, b)))
    worst = sorted(rows, key=lambda r: r[1])[:5]
    print("parameter gradients: %d tensors, worst" % len(rows), [(n[-40:], round(c, 5), round(r, 4)) for n, c, r in worst])
    drel = {k: (float(torch.dot(cpu(fin[k].grad).flatten(), fr[k].grad.flatten()) / (cpu(fin[k].grad).norm() * fr[k].grad.norm() + 1e-30)),
                rel(cpu(fin[k].grad), fr[k].grad)) for k in fin}
    print("d features (cosine, rel)", {k: (round(c, 5), round(r, 4)) for k, (c, r) in drel.items()})
    # iam_conv.bias: the aggregated instance features are (sum prob * f) / (sum prob) - invariant to a rescaling of prob, and
    # with the prior-probability bias (-4.6) sigmoid ~ exp, so a uniform shift of the IAM logits (= the bias) IS a rescaling:
    # its gradient is a near-total cancellation of 12 800 per-pixel terms per channel and carries their bf16 rounding at full
    # size (measured cosine 0.9971, rel 0.087); every other tensor: measured >= 0.9990, typically 0.9999 / 1 %
    lim = lambda n: (0.995, 0.12) if n == "decoder.inst_branch.iam_conv.bias" else (0.999, 0.05)
    bad = [(n, round(c, 5), round(r, 4)) for n, c, r in rows if c < lim(n)[0] or r > lim(n)[1]]
    assert not bad, bad[:8]
    assert all(c > 0.999 and r < 0.05 for c, r in drel.values()), drel


def test_mask_statistics_are_exact_sums_in_a_fixed_order():
    """mi_sparseinst_mask_stats at the bench's mask size (P = 160 x 160: 13 blocks per pair): every call returns the same bits
    (block partials summed in a fixed order - the first form added them with fp32 atomics and the captured SparseInst step
    alternated between two final losses run to run), unused rows (image index < 0) are zero, and the sums match fp64."""
    from yolov7_d2_amd import _lib as L
    g = torch.Generator().manual_seed(21)
    B, P, Np, K = 2, 160 * 160, 128, 24
    masks = (torch.randn(B, P, Np, generator=g) * 3).to(torch.bfloat16).to(DEV)
    tgt = (torch.rand(K, P, generator=g) > 0.7).float().to(DEV)
    pairs = torch.stack([torch.randint(0, B, (K,), generator=g), torch.randint(0, 100, (K,), generator=g), torch.arange(K)], 1).int()
    pairs[5, 0] = pairs[17, 0] = -1
    pairs = pairs.to(DEV).contiguous()
    ws = torch.empty(int(L.lib().mi_sparseinst_mask_stats_ws_floats(K, P)), dtype=torch.float32, device=DEV)
    outs = []
    for _ in range(6):
        stats = torch.full((K, 8), float("nan"), dtype=torch.float32, device=DEV)
        L.check(L.lib().mi_sparseinst_mask_stats(masks.data_ptr(), Np, P, tgt.data_ptr(), pairs.data_ptr(), K, stats.data_ptr(),
                                                 ws.data_ptr(), L.stream_ptr()), "mi_sparseinst_mask_stats")
        outs.append(stats.clone())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    st = outs[0].cpu().double()
    assert float(st[5].abs().max()) == 0.0 and float(st[17].abs().max()) == 0.0 and float(st[:, 7].abs().max()) == 0.0
    pc = pairs.cpu()
    for k in (0, 3, 11, 23):
        x = masks[pc[k, 0], :, pc[k, 1]].double().cpu()
        t = tgt[k].double().cpu()
        sg = torch.sigmoid(x)
        ref = [float((x.clamp(min=0) - x * t + torch.log1p(torch.exp(-x.abs()))).sum()), float((sg * t).sum()), float((sg * sg).sum()),
               float((t * t).sum())]
        for e in range(4):
            assert abs(float(st[k, e]) - ref[e]) <= 2e-4 * abs(ref[e]) + 1e-3, (k, e, float(st[k, e]), ref[e])


def test_staged_backward_graphs_reproduce_the_one_graph_step():
    """GraphedTrainStep(backward_stages=True): the backward cut at the ResNet stages into one hipGraph per stage (what the
    data-parallel step does to overlap its all-reduce, graph_step.py) against the one-graph capture, single process.
    SparseInst is the hard case: res3 / res4 / res5 all feed the encoder AND the next stage, so every cut tensor's gradient
    is the sum of two consumers that arrive in different stages.  Same weights, same three batches: identical losses and
    parameters (two-term sums commute)."""
    from yolov7_d2_amd.graph_step import GraphedTrainStep
    from yolov7_d2_amd.optim import MultiTensorAdamW
    import copy
    torch.manual_seed(0)
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    one = M.build_model(cfg)
    cut = copy.deepcopy(one)
    one.train(); cut.train()
    mk = lambda m: MultiTensorAdamW([p for p in m.parameters() if p.requires_grad], lr=5e-5, weight_decay=0.05)
    s1, s2 = GraphedTrainStep(one, mk(one)), GraphedTrainStep(cut, mk(cut), backward_stages=True)
    batches = [_si_inputs(1, ((192, 224), (160, 200))), _si_inputs(2, ((180, 210), (192, 224)), counts=(3, 0)),
               _si_inputs(3, ((170, 224), (192, 200)))]
    try:
        for it, b in enumerate(batches):
            a, c = s1(b), s2(b)
            for k in a:
                assert torch.equal(a[k], c[k]), (it, k, float(a[k]), float(c[k]))
        # encoder + decoder + criterion, res5, res4, res3, res2 + stem (Base-SparseInst.yaml:7 FREEZE_AT 0: everything trains)
        assert len(s2.stage_params) == 5 and all(len(x) > 0 for x in s2.stage_params)
        assert len(next(iter(s2.graphs.values()))[0]) == 5 and len(next(iter(s1.graphs.values()))[0]) == 1
        torch.cuda.synchronize()
        for (n, p), (_, q) in zip(one.named_parameters(), cut.named_parameters()):
            assert torch.equal(p.detach(), q.detach()), n
    finally:
        s1.close(); s2.close()


def test_grouped_iam_conv_on_one_padded_map_equals_the_cat_form(monkeypatch):
    """GroupInstanceBranch (decoder_sparseinst.py:212-242) with the grouped IAM convolution writing channel slices of ONE
    padded NHWC map that the sigmoid, the aggregation and the whole backward read in place (round 6, _GroupIamFn /
    _aggregate_padded) against round 5's form (G convolution ops + torch.cat + per-image zero-padded copies): same kernels on
    the same values - outputs equal, gradients to the split-K summation order of the (now grouped) weight gradients and the
    512- instead of 416-row outer products"""
    from yolov7_d2_amd.modeling.sparseinst import GroupInstanceBranch
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    res = []
    for padded in ("0", "1"):
        monkeypatch.setenv("MI_SI_IAM_PADDED", padded)
        torch.manual_seed(11)
        br = GroupInstanceBranch(cfg, 256).to(DEV)
        with torch.no_grad():
            br.iam_conv.weight.normal_(0, 0.05)
            br.iam_conv.bias.normal_(0, 0.5)
        g = torch.Generator().manual_seed(5)
        x = (torch.randn(2, 256, 24, 40, generator=g) * 0.5).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        logits, kern, obj, iam = br(x)
        gl, gk, go = (torch.randn(t.shape, generator=g).to(DEV) for t in (logits, kern, obj))
        ((logits.float() * gl).sum() + (kern.float() * gk).sum() + (obj.float() * go).sum()).backward()
        torch.cuda.synchronize()
        iam_t = iam() if callable(iam) else iam
        res.append(dict(logits=logits.detach().float(), kern=kern.detach().float(), obj=obj.detach().float(), iam=iam_t.detach().float(),
                        dx=x.grad.float(), **{"g:" + k: p.grad.float() for k, p in br.named_parameters()}))
    a, b = res
    assert set(a) == set(b) and "g:iam_conv.weight" in a and "g:iam_conv.bias" in a
    assert torch.equal(a["iam"], b["iam"])                      # the same four convolutions on the same operands
    for k in a:
        scale = float(a[k].abs().max()) + 1e-30
        tol = 2e-2 if k in ("dx",) or k.startswith("g:inst_convs") else 5e-3       # bf16 maps downstream of a re-ordered fp32 sum
        assert float((a[k] - b[k]).abs().max()) <= tol * scale, (k, float((a[k] - b[k]).abs().max()), scale)


def test_masks_from_predicted_kernels_as_one_node_equal_the_per_image_nodes():
    """decoder_sparseinst.py:141-147 (`torch.bmm(pred_kernel, mask_features.view(B, C, H * W))`): the batch's masks as ONE
    autograd node (_MaskKernelFn: per-image convolutions into row blocks of one map, one grouped weight-gradient launch)
    against round 5's B `_LinearFn` nodes + torch.stack - same kernels on the same operands: the masks and the feature
    gradient bit for bit, the kernels' gradient to the split-K summation order of the grouped launch"""
    from yolov7_d2_amd.modeling.sparseinst import _MaskKernelFn, _pad_rows
    from yolov7_d2_amd.modeling.transformer import _LinearFn
    g = torch.Generator().manual_seed(3)
    B, P, Cc, N, Np = 3, 40 * 24, 128, 100, 128
    mf0 = (torch.randn(B, P, Cc, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    k0 = (torch.randn(B, N, Cc, generator=g) * 0.2).to(torch.bfloat16).to(DEV)
    gy = torch.randn(B, P, Np, generator=g).to(torch.bfloat16).to(DEV)
    gy[:, :, N:] = 0                                  # (the pad instances receive no gradient: the model slices them off)
    res = []
    for one in (False, True):
        mf, k = mf0.clone().requires_grad_(True), k0.clone().requires_grad_(True)
        if one:
            y = _MaskKernelFn.apply(mf, k, Np)
        else:
            y = torch.stack([_LinearFn.apply(mf[b], _pad_rows(k[b].float(), Np), None) for b in range(B)])
        (y.float() * gy.float()).sum().backward()
        torch.cuda.synchronize()
        res.append((y.detach(), mf.grad, k.grad))
    (ya, dma, dka), (yb, dmb, dkb) = res
    assert torch.equal(ya, yb) and float(ya.float().abs().max()) > 0.5
    assert torch.equal(dma, dmb)
    assert dka.dtype == dkb.dtype == torch.bfloat16
    assert float((dka.float() - dkb.float()).abs().max()) <= 1e-2 * float(dka.float().abs().max())
    # ... and against fp32 torch
    ref = torch.bmm(mf0.float(), k0.float().transpose(1, 2))
    assert float((ya[:, :, :N].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    assert float(ya[:, :, N:].float().abs().max()) == 0.0


def test_fused_matcher_cost_and_criterion_equal_the_torch_spelling(monkeypatch):
    """SparseInstCriterion + SparseInstMatcher (loss/sparseinst_loss.py) with the cost matrix, the pair bookkeeping, the four
    losses and their gradients in csrc/sparseinst_loss.hip (default) against the same module spelled in torch calls
    (MI_SI_FUSED_LOSS=0, round 5's form, itself pinned to the reference's golden vectors above): the same assignment, losses
    to 1e-5, gradients to 1e-4 of their scale - on a batch with an image without instances"""
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV)
    crit = S.build_sparse_inst_criterion(cfg).to(DEV)
    g = torch.Generator().manual_seed(21)
    B, N, C_, Ho, Wo, Np, cap = 3, 100, 80, 40, 56, 128, 32
    logits0 = (torch.randn(B, N, C_, generator=g) * 2 - 2).to(DEV)
    scores0 = torch.randn(B, N, 1, generator=g).to(DEV)
    masks0 = (torch.randn(B, Ho, Wo, Np, generator=g) * 2).to(torch.bfloat16).to(DEV)
    tg = []
    for b, n in enumerate((5, 0, 9)):
        m = torch.zeros(n, Ho * 4, Wo * 4)
        for k in range(n):
            y0, x0 = int(torch.randint(0, Ho * 3, (1,), generator=g)), int(torch.randint(0, Wo * 3, (1,), generator=g))
            m[k, y0:y0 + 30 + 7 * k, x0:x0 + 40 + 5 * k] = 1
        tg.append({"labels": torch.randint(0, C_, (n,), generator=g).to(DEV), "masks": m.to(DEV)})
    pk = S.PackedMaskTargets(B, cap, (Ho, Wo), DEV).fill(tg, (Ho * 4, Wo * 4))
    res = []
    for fused in ("0", "1"):
        monkeypatch.setenv("MI_SI_FUSED_LOSS", fused)
        logits, scores, masks = (t.clone().requires_grad_(True) for t in (logits0, scores0, masks0))
        out = {"pred_logits": logits, "pred_scores": scores, "_masks_nhwc": masks}
        with torch.no_grad():
            mq, mt, nm = crit.matcher.match_packed(out, pk)
        losses = crit(out, pk)
        sum(w * losses[k] for w, k in zip((1.0, 0.7, 1.3, 0.9), ("loss_ce", "loss_mask", "loss_dice", "loss_objectness"))).backward()
        torch.cuda.synchronize()
        res.append(dict(mq=mq, mt=mt, nm=nm, losses={k: float(v) for k, v in losses.items()}, dl=logits.grad, ds=scores.grad, dm=masks.grad.float()))
    a, b = res
    assert torch.equal(a["nm"], b["nm"]) and a["nm"].tolist() == [5, 0, 9]
    for i, n in enumerate(a["nm"].tolist()):
        assert torch.equal(a["mq"][i, :n], b["mq"][i, :n]) and torch.equal(a["mt"][i, :n], b["mt"][i, :n])
    assert set(a["losses"]) == set(b["losses"]) == {"loss_ce", "loss_mask", "loss_dice", "loss_objectness"}
    for k in a["losses"]:
        assert a["losses"][k] == pytest.approx(b["losses"][k], rel=1e-5), k
        assert a["losses"][k] > 0
    for k in ("dl", "ds", "dm"):
        scale = float(a[k].abs().max())
        assert scale > 0 and float((a[k] - b[k]).abs().max()) <= 1e-4 * scale, (k, float((a[k] - b[k]).abs().max()), scale)


def test_pyramid_pooling_stages_in_one_launch_equal_the_torch_calls(monkeypatch):
    """PyramidPoolingModule (encoder_sparseinst.py:18-62) with its four MyAdaptiveAvgPool2d stages as ONE launch forward and one
    backward (mi_pyramid_pool_fwd / _bwd) against the same module spelled in torch calls (MI_SI_PPM_FUSED=0): the pooled maps to
    one bf16 rounding, the module's output and gradients to bf16 accuracy - on a map whose size the windows do not divide"""
    res = []
    for fused in ("0", "1"):
        monkeypatch.setenv("MI_SI_PPM_FUSED", fused)
        torch.manual_seed(5)
        ppm = S.PyramidPoolingModule(256, 64).to(DEV)
        g = torch.Generator().manual_seed(6)
        x = torch.randn(2, 256, 20, 23, generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = ppm(x)
        gy = torch.randn(y.shape, generator=g).to(DEV)
        (y.float() * gy).sum().backward()
        torch.cuda.synchronize()
        res.append(dict(y=y.detach().float(), dx=x.grad.float(), **{"g:" + k: p.grad.float() for k, p in ppm.named_parameters()}))
    a, b = res
    for k in a:
        scale = float(a[k].abs().max()) + 1e-30
        assert float((a[k] - b[k]).abs().max()) <= 2e-2 * scale, (k, float((a[k] - b[k]).abs().max()), scale)
    # the pooling itself against torch, stage by stage
    from yolov7_d2_amd.modeling.sparseinst import _PyramidPoolFn
    x = torch.randn(2, 64, 20, 23, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16).to(DEV).requires_grad_(True)
    ks = ((20, 23), (10, 12), (7, 8), (4, 4))
    outs = _PyramidPoolFn.apply(x, ks)
    refs = [F.avg_pool2d(x.detach().float(), kernel_size=k, ceil_mode=False) for k in ks]
    for o, r in zip(outs, refs):
        assert o.shape == r.shape and float((o.float() - r).abs().max()) <= 2 ** -8 * float(r.abs().max())
    gs = [torch.randn(r.shape, generator=torch.Generator().manual_seed(8)).to(torch.bfloat16).to(DEV) for r in refs]
    torch.autograd.backward(outs, gs)
    xr = x.detach().float().requires_grad_(True)
    torch.autograd.backward([F.avg_pool2d(xr, kernel_size=k, ceil_mode=False) for k in ks], [g_.float() for g_ in gs])
    assert float((x.grad.float() - xr.grad).abs().max()) <= 2 ** -7 * float(xr.grad.abs().max())
