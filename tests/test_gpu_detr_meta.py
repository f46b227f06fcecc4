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
