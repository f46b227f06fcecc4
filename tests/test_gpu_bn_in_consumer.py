"""GPU (-m gpu): BatchNorm(train) + SiLU applied by the consuming convolution launch (csrc/conv_bn.h BnXf,
mi_conv_desc.xf; plan.Plan._defer_bn) against the two-launch form it replaces - BaseConv.forward's norm + act
(yolov7/modeling/backbone/layers/wrappers.py:76-83) followed by the next layer's conv.  The consumer applies the same
expression at the same rounding point, so everything the step writes must be BIT-identical: every activation, every
raw conv output, the recorded scale / shift / mean / invstd, the running statistics, the losses, every gradient."""
import ctypes as C

import numpy as np
import pytest
import torch

import yolox_oracle as O
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.modeling.yolox import _PlanState
from yolov7_d2_amd.params import ParamArena

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(seed=0):
    cfg = M.yolox_s_cfg(device=DEV)
    model = M.build_model(cfg)
    sd = O.init_state_dict(0.33, 0.5, 80, seed=seed)
    model.load_state_dict(sd)
    model.to(DEV)
    model.params = ParamArena(model, DEV)
    return model


def _run(ps, imgs, labels):
    ps.image.copy_(imgs)
    ps.labels.copy_(labels)
    ps.plan.run("fwd")
    ps.gw().fill_(1.0)
    ps.plan.run("bwd")
    torch.cuda.synchronize()


def _snapshot(model, ps):
    snap = {}
    for bf in ps.builder.bufs:
        if bf.name.startswith("scratch.") or bf.name.endswith((".bar", ".wd_pair")) or bf.nbytes == 0:
            continue
        snap[bf.name] = ps.plan.buf_view(bf, torch.uint8).clone()
    snap["param.grad"] = model.params.grad.clone()
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            snap["state." + k] = v.clone()
    return snap


@pytest.mark.parametrize("B,H,W", [(2, 256, 320), (4, 160, 224), (3, 224, 224)])
def test_bn_in_consumer_is_bit_identical_to_the_two_launch_form(monkeypatch, B, H, W):
    imgs, labels = O.synth_batch(B, H, W, seed=5, max_gt=6)
    imgs, labels = imgs.to(DEV), labels.to(DEV)
    snaps, plans = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MI_BN_IN_CONSUMER", mode)
        model = _model(seed=3)
        ps = _PlanState(model, B, H, W, True)
        _run(ps, imgs, labels)
        snaps[mode] = _snapshot(model, ps)
        plans[mode] = ps
    p0, p1 = plans["0"].plan, plans["1"].plan
    assert p0.deferred_bn == [] and len(p1.deferred_bn) >= 10, p1.deferred_bn
    ops = lambda p: [L.OPS[p.fwd_cmds[0][k].op] for k in range(p.fwd_cmds[1])]
    nbn = lambda p: sum(len(m) if m else 1 for m, o in zip(p.cmd_members["fwd"], ops(p)) if o in ("BN_ACT_FWD", "BN_GROUP"))
    assert nbn(p0) == 74 and nbn(p1) == 74 - len(p1.deferred_bn)
    # both consumer kernels took part: a 1x1 reader (CSP conv1 -> Bottleneck conv1) and a 3x3 reader (Bottleneck conv1 -> conv2)
    if B * (H // 4) * (W // 4) % 128 == 0:
        assert any(t.endswith(".conv1.bnact") and ".m." not in t for t in p1.deferred_bn)
    assert any(".m.0.conv1.bnact" in t for t in p1.deferred_bn)
    bad = []
    assert snaps["0"].keys() == snaps["1"].keys()
    for k, a in snaps["0"].items():
        b = snaps["1"][k]
        if not torch.equal(a, b):
            fa, fb = (a.view(torch.bfloat16).float(), b.view(torch.bfloat16).float()) if a.dtype == torch.uint8 and a.numel() % 2 == 0 else (a.float(), b.float())
            bad.append((k, int((a != b).sum()), a.numel(), float((fa - fb).abs().max())))
    if bad:   # (the whole list, in buffer order, for the post-mortem)
        import os
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/bnx_mismatch_{B}_{H}_{W}.txt", "w") as f:
            f.write("\n".join(f"{k} {n}/{tot} max {mx}" for k, n, tot, mx in bad) + "\n")
            f.write("deferred: " + " ".join(p1.deferred_bn) + "\n")
    assert not bad, bad[:12]
    assert float(plans["1"].loss_out()[0]) > 0 and np.isfinite(float(plans["1"].loss_out()[0]))


def test_xf_descriptor_is_refused_off_the_two_kernels():
    """mi_conv2d with an input-BatchNorm record on a shape only the tile kernel runs (K = 256 3x3) fails loudly"""
    d = L.mi_conv_desc()
    x = torch.zeros(2 * 20 * 20 * 256, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(9 * 256 * 256, dtype=torch.bfloat16, device=DEV)
    y = torch.zeros(2 * 20 * 20 * 256, dtype=torch.bfloat16, device=DEV)
    acc = torch.zeros(16 * 256 * 2, dtype=torch.float64, device=DEV)
    d.x, d.w, d.y, d.stats_acc = x.data_ptr(), w.data_ptr(), y.data_ptr(), acc.data_ptr()
    d.ldx = d.ldy = 256
    d.N, d.H, d.W, d.outH, d.outW, d.gridH, d.gridW = 2, 20, 20, 20, 20, 20, 20
    d.in_stride = d.out_stride = 1
    d.K8, d.Cout, d.CoutPad, d.ntaps = 32, 256, 256, 9
    for t, (a, b) in enumerate((a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)):
        d.tap_dy[t], d.tap_dx[t], d.tap_w[t] = a, b, t
    d.xf, d.xf_write, d.xf_C = acc.data_ptr(), 1, 256
    lib = L.lib()
    assert lib.mi_conv2d_route(C.byref(d)) == -1
    assert lib.mi_conv2d(C.byref(d), L.stream_ptr()) < 0
    lib.mi_last_error()
