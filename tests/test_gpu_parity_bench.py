"""GPU (-m gpu): whole-step parity AT THE BENCHMARK CONFIGURATION (YOLOX-s, 16 x 3 x 640 x 640, BASELINE.json configs[1])
of the HIP path against the oracle: every layer forward, the SimOTA / loss head, EVERY parameter gradient, and the eval
path with real detections.

Why teacher forcing.  Both sides store activations as bf16.  Two bf16-storage forward passes that differ in a single
rounding decorrelate to ~1 % (relative L2) at the head output, and train-mode BatchNorm's backward (dy = g*(dz - mean(dz)
- xhat*mean(dz*xhat)), a projection that cancels most of dz on a random-init net) amplifies that forward noise to 10-60 %
in the weight gradients: the ORACLE'S OWN two bf16 emulations agree only to cosine 0.77-0.85 on the backbone gradients
(tests/test_oracle_golden.py::test_bf16_storage_noise_floor, CPU).  An end-to-end "cosine >= 0.99 per tensor" bound is
therefore not a property any correct bf16 implementation has.  What IS checkable to that bound - and is strictly
stronger per layer - is the step with the forward state pinned:

  (1) forward, layer by layer   every one of the 74 BaseConv outputs y (pre-BatchNorm, as the HIP path stored it) vs the
                                oracle's conv of the previous layers' outputs, which are themselves forced to the HIP
                                values: per-layer relative L2 at the bf16 rounding level; the raw head output likewise
  (2) SimOTA + losses           HIP losses / fg mask / matched gt / d(loss)/d(raw) vs the oracle ON THE HIP RAW OUTPUT:
                                integers bit-exact, floats 1e-4
  (3) backward, whole network   all 240 parameter gradients (462 state_dict tensors = 240 parameters + 222 buffers) of
                                the HIP step vs the oracle's autograd through the WHOLE network from the same
                                d(loss)/d(raw), forward values pinned to the HIP ones: cosine >= 0.999, rel L2 <= 0.05
  (4) the un-forced comparison  reported, and bounded by 1.5x the oracle's own bf16 noise floor

(3) is the check a dropped / mis-ordered gradient contribution cannot pass: MI_TEST_DROP_ACCUM makes one accumulating
data gradient overwrite instead (test_dropped_accumulation_is_caught) and the bounds must fail.
Reference: yolov7/modeling/head/yolox_head.py:274-441, yolov7/modeling/meta_arch/yolox.py:171-252."""
import os

import numpy as np
import pytest
import torch

import yolox_oracle as O
import yolov7_d2_amd as M

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, H, W = 16, 640, 640

# bounds (measured on MI355X: the tests print the distributions)
COS_MIN = 0.999         # forced-forward backward pass, every parameter tensor
REL_MAX = 0.05
Y_REL_MAX = 1e-3        # a layer's conv output on identical (forced) inputs (measured max 1.4e-4: fp32 accumulation order + bf16 ties)
RAW_REL_MAX = 1e-5      # fp32 prediction convs on forced features (measured 4.5e-7)
RAW_FREE_REL_MAX = 3e-2  # un-forced end-to-end raw output (bf16 storage noise through ~40 layers)


from parity_util import grad_table as _grad_table, hip_step, oracle_backward


def _hip_step(seed_model, imgs, labels, want_y=False):
    sd = O.init_state_dict(0.33, 0.5, 80, seed=seed_model)
    return sd, hip_step(sd, imgs, labels, want_y=want_y)


def _oracle_backward(sd, imgs, dpreds, force):
    return oracle_backward(sd, imgs, dpreds, force)


@pytest.fixture(scope="module")
def bench_case():
    imgs, labels = O.synth_batch(B, H, W, seed=1234, max_gt=20)
    sd, hip = _hip_step(0, imgs, labels, want_y=True)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    ys = hip.pop("y")
    forced = _oracle_backward(sd, imgs, hip["dpreds"], ys)   # (buffers are 256-byte padded: the hook takes the leading part)
    free = _oracle_backward(sd, imgs, hip["dpreds"], None)
    return dict(imgs=imgs, labels=labels, sd=sd, hip=hip, forced=forced, free=free, nlayers=len(ys))


def test_bench_config_forward_every_layer(bench_case):
    """(1) all 74 BaseConv outputs + the raw head output, HIP vs oracle on identical (forced) inputs"""
    fe = bench_case["forced"]["force_err"]
    assert len(fe) == bench_case["nlayers"] == 74
    worst = sorted(fe.items(), key=lambda kv: -kv[1])
    print("per-layer conv output rel L2 (forced inputs): max %.2e median %.2e; worst %s" %
          (worst[0][1], float(np.median([v for v in fe.values()])), worst[:3]))
    assert worst[0][1] < Y_REL_MAX, worst[:5]
    raw, ref = bench_case["hip"]["raw"], bench_case["forced"]["raw"]
    assert raw.shape == (B, 8400, 85)
    rel = float((raw - ref).norm() / ref.norm())
    free = float((raw - bench_case["free"]["raw"]).norm() / bench_case["free"]["raw"].norm())
    print("raw head output rel L2: forced %.2e, un-forced %.2e" % (rel, free))
    assert rel < RAW_REL_MAX and free < RAW_FREE_REL_MAX, (rel, free)
    # train-mode BatchNorm side effect: running statistics after one step (fp32 on both sides)
    w = 0.0
    for k, v in bench_case["hip"]["rm"].items():
        r = bench_case["forced"]["osd"][k]
        w = max(w, float((v - r).abs().max() / (r.abs().max() + 1e-6)))
    assert w < 5e-3, w


def test_bench_config_losses_and_assignment(bench_case):
    """(2) SimOTA + losses at 16 x 8400 anchors on the raw output the HIP network produced: fg mask and matched gt
    bit-exact, the four losses 1e-4, d(sum of the loss dict)/d(raw) 2e-4"""
    hip, labels = bench_case["hip"], bench_case["labels"]
    raw = hip["raw"].clone().requires_grad_(True)
    res, assigns = O.yolox_losses(raw, labels, hip["anchors"], 80, return_assign=True)
    got = hip["losses"][:4].numpy()
    np.testing.assert_allclose(got, np.array([float(x) for x in res[:4]]), rtol=1e-4, atol=1e-5)
    nfg = 0
    for b in range(B):
        fg = hip["fg"][b].bool()
        assert torch.equal(fg, assigns[b]["fg"]), f"image {b}: fg mask differs on {int((fg != assigns[b]['fg']).sum())} anchors"
        assert torch.equal(hip["mgt"][b][fg].long(), assigns[b]["matched_gt"].long())
        nfg += int(fg.sum())
    assert float(hip["losses"][6]) == nfg and nfg > 100
    (res[0] + res[1] + res[2] + res[3]).backward()     # detectron2 sums the whole loss dict
    d = hip["dpreds"]
    err = float((d - raw.grad).norm() / raw.grad.norm())
    assert err < 2e-4, err
    np.testing.assert_allclose(d.numpy(), raw.grad.numpy(), rtol=2e-3, atol=2e-6)


def test_bench_config_every_parameter_gradient(bench_case):
    """(3) all 240 parameter tensors against the oracle's autograd through the whole network (forward values pinned to
    the HIP ones): cosine >= 0.999, relative L2 <= 0.05, norm ratio within 2 %"""
    rows = _grad_table(bench_case["hip"]["grads"], bench_case["forced"]["grads"])
    assert len(rows) == 240 and len(bench_case["sd"]) == 462
    rows.sort(key=lambda r: r[1])
    cos = np.array([r[1] for r in rows]); rel = np.array([r[2] for r in rows])
    print(f"parameter gradients (forced forward): cos min {cos.min():.6f} median {np.median(cos):.7f} | rel L2 max "
          f"{rel.max():.4f} median {np.median(rel):.4f}")
    for r in rows[:4]:
        print("   worst cos  %-44s cos %.6f rel %.4f |g| %.3e |ref| %.3e" % r)
    bad = [r for r in rows if not (r[1] >= COS_MIN and r[2] <= REL_MAX)]
    assert not bad, bad[:8]
    ratio = np.array([r[3] / (r[4] + 1e-30) for r in rows])
    assert ratio.min() > 0.98 and ratio.max() < 1.02, (ratio.min(), ratio.max())


def test_bench_config_unforced_gradients_within_noise_floor(bench_case):
    """(4) the plain end-to-end comparison (no forcing): HIP vs the bf16-emulating oracle.  Bounded by the oracle's own
    bf16 noise floor (two emulations of the same network agree to cosine ~0.77 on the backbone gradients, ~0.97 on the
    head stems: test_bf16_storage_noise_floor) - reported for the record, asserted loosely"""
    rows = _grad_table(bench_case["hip"]["grads"], bench_case["free"]["grads"])
    cos = np.array([r[1] for r in rows])
    print(f"parameter gradients (un-forced): cos min {cos.min():.4f} median {np.median(cos):.4f}")
    assert np.median(cos) > 0.7 and cos.min() > 0.5, sorted(rows, key=lambda r: r[1])[:4]
    ratio = np.array([r[3] / (r[4] + 1e-30) for r in rows])
    assert 0.8 < np.median(ratio) < 1.25


def test_dropped_accumulation_is_caught(bench_case, monkeypatch):
    """mutation check of (3): one accumulating data gradient (the 7th of the backward list) overwrites its target
    instead - the earlier consumers' contribution is lost - and the per-tensor bounds must fail"""
    monkeypatch.setenv("MI_TEST_DROP_ACCUM", "7")
    _, hip = _hip_step(0, bench_case["imgs"], bench_case["labels"])
    monkeypatch.delenv("MI_TEST_DROP_ACCUM")
    # the forward pass is untouched: same loss gradient goes into the backward pass
    assert torch.equal(hip["dpreds"], bench_case["hip"]["dpreds"])
    rows = _grad_table(hip["grads"], bench_case["forced"]["grads"])
    bad = [r for r in rows if not (r[1] >= COS_MIN and r[2] <= REL_MAX)]
    print("mutated run:", len(bad), "tensors out of bounds; worst", sorted(bad, key=lambda r: r[1])[:3])
    assert len(bad) >= 1


def test_eval_path_with_real_detections():
    """YOLOX.forward in eval mode with a head biased so that thousands of anchors pass the confidence threshold and
    the class-aware NMS has work to do: decode (yolox_head.py:247-272) against the oracle on the same raw output, then
    postprocess -> Instances -> detector_postprocess (meta_arch/yolox.py:211-252, utils/boxes.py:171-210) against the
    oracle's restatement on the same decoded output: same detections in the same order, integer classes exact"""
    from test_gpu_step import _batched_inputs
    cfg = M.yolox_s_cfg(device=DEV)
    cfg.MODEL.YOLO.CONF_THRESHOLD = 0.05
    model = M.build_model(cfg)
    sd = O.init_state_dict(0.33, 0.5, 80, seed=5)
    g = torch.Generator().manual_seed(9)
    for k in range(3):
        sd[f"head.obj_preds.{k}.bias"].fill_(0.5)
        sd[f"head.cls_preds.{k}.bias"].fill_(-3.0)
        sd[f"head.cls_preds.{k}.bias"][[3, 17, 56]] = torch.tensor([0.3, 0.1, 0.2])
        sd[f"head.cls_preds.{k}.weight"] *= 6.0          # class / score variety across anchors
        sd[f"head.obj_preds.{k}.weight"] *= 6.0
        sd[f"head.reg_preds.{k}.bias"][2:] = 1.8          # boxes ~6 strides wide: neighbours overlap above the threshold
        sd[f"head.reg_preds.{k}.weight"] *= 3.0
    model.load_state_dict(sd)
    model.eval()
    Bn, Hn, Wn = 2, 320, 416
    imgs, labels = O.synth_batch(Bn, Hn, Wn, seed=23, max_gt=4)
    inputs = _batched_inputs(imgs, labels)
    inputs[1]["height"], inputs[1]["width"] = 480, 624    # detector_postprocess rescales to the requested output size
    with torch.no_grad():
        res = model(inputs)
    ps = model.plan_for(Bn, Hn, Wn, False)
    dec = ps.preds().float().cpu().clone()                # decoded [B, A, 85] the eval plan left behind
    # decode against the oracle: re-run the network part only to get the raw output is not possible after DECODE ran in
    # place, so check decode's invariants on the decoded tensor and the formula on a fresh raw tensor below
    assert dec.shape == (Bn, ps.A, 85) and float(dec[..., 4:].min()) >= 0.0 and float(dec[..., 4:].max()) <= 1.0
    ref = O.postprocess(dec, 80, 0.05, cfg.MODEL.YOLO.NMS_THRESHOLD)
    ntot = 0
    for b in range(Bn):
        inst = res[b]["instances"]
        assert ref[b] is not None
        r = ref[b]
        oh, ow = (480, 624) if b == 1 else (Hn, Wn)
        sx, sy = ow / Wn, oh / Hn
        boxes = r[:, :4].clone()
        boxes[:, 0::2] *= sx; boxes[:, 1::2] *= sy
        boxes[:, 0::2] = boxes[:, 0::2].clamp(0, ow); boxes[:, 1::2] = boxes[:, 1::2].clamp(0, oh)
        keep = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
        boxes, r = boxes[keep], r[keep]
        assert len(inst) == boxes.shape[0], (len(inst), boxes.shape[0])
        assert torch.equal(inst.pred_classes.cpu(), r[:, 6])                       # integer class ids exact
        np.testing.assert_allclose(inst.pred_boxes.tensor.cpu().numpy(), boxes.numpy(), rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(inst.scores.cpu().numpy(), (r[:, 4] * r[:, 5]).numpy(), rtol=1e-6, atol=1e-7)
        assert (inst.scores[:-1] >= inst.scores[1:]).all()
        ntot += len(inst)
        # NMS really suppressed something and really kept something
        ncand = int((dec[b, :, 4] * dec[b, :, 5:].max(1).values >= 0.05).sum())
        assert 50 < len(inst) < ncand, (len(inst), ncand)
    assert ntot > 200
    # decode formula (xy + grid) * stride, exp(wh) * stride, sigmoid(obj, cls) against the oracle on identical raw input
    raw, anchors = O.synth_raw(2, [(40, 52), (20, 26), (10, 13)], 31)
    import ctypes as C
    from yolov7_d2_amd import _lib as L
    rd = raw.to(DEV).contiguous(); ad = anchors.to(DEV).contiguous()
    L.check(L.lib().mi_yolox_decode(rd.data_ptr(), ad.data_ptr(), 2, anchors.shape[0], 80, L.stream_ptr()), "decode")
    torch.cuda.synchronize()
    np.testing.assert_allclose(rd.cpu().numpy(), O.decode_eval(raw, anchors).numpy(), rtol=2e-6, atol=1e-6)
