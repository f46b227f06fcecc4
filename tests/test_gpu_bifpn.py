"""GPU (-m gpu): the BiFPN neck (neck/bifpn.py).  Kernels (GroupNorm, 2x2 max-pool, fast-attention fusion, Swish) against
plain torch fp32 on the same bf16-rounded inputs; the whole neck, forward and backward, against the golden produced by the
reference's own BiFPN (oracle/gen_golden.py: gold_bifpn), with the bf16 activation storage as the only difference."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gen_golden_inputs import BIFPN_CASES, bifpn_state_dict, synth_bifpn_case
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.d2shim import Backbone, ShapeSpec
from yolov7_d2_amd.modeling import bifpn as B

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bf(t):
    return t.to(torch.bfloat16).float()


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def _close(got, ref, rel, what):
    """max |got - ref| <= rel * max |ref|  (bf16 storage: 2^-9 relative per rounding)"""
    err = float((got.double() - ref.double()).abs().max())
    scale = float(ref.double().abs().max())
    assert err <= rel * scale + 1e-6, (what, err, scale)


@pytest.mark.parametrize("N,H,W,C,G", [(2, 16, 20, 160, 32), (3, 7, 5, 64, 32), (1, 40, 40, 288, 32), (2, 1, 2, 64, 32),
                                       (2, 9, 9, 96, 32), (1, 64, 64, 256, 32), (2, 8, 8, 2048, 32)])
def test_groupnorm_against_torch(N, H, W, C, G):
    g = torch.Generator().manual_seed(N * 1000 + H * 31 + C)
    x = _bf(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).to(DEV)
    go = _bf(torch.randn(N, C, H, W, generator=g)).to(DEV)
    gn = B.GroupNorm(G, C).to(DEV)
    with torch.no_grad():
        gn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        gn.bias.copy_(0.2 * torch.randn(C, generator=g))
    xr = x.clone().requires_grad_(True)
    ref = F.group_norm(xr, G, gn.weight, gn.bias, gn.eps)
    gw_ref, gb_ref = torch.autograd.grad((ref * go).sum(), [gn.weight, gn.bias], retain_graph=True)
    (dx_ref,) = torch.autograd.grad((ref * go).sum(), [xr])
    xm = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = gn(xm)
    assert out.dtype == torch.bfloat16 and out.shape == x.shape
    (out.float() * go).sum().backward()
    _close(out.float(), ref, 6e-3, "y")
    _close(xm.grad.float(), dx_ref, 8e-3, "dx")
    _close(gn.weight.grad, gw_ref, 2e-3, "dgamma")     # fp64 sums of bf16 inputs: only the xhat rounding differs
    _close(gn.bias.grad, gb_ref, 1e-4, "dbeta")


@pytest.mark.parametrize("N,H,W,C", [(2, 16, 16, 64), (1, 7, 9, 160), (3, 2, 2, 8), (1, 33, 18, 96)])
def test_maxpool2x2_against_torch(N, H, W, C):
    g = torch.Generator().manual_seed(H * 100 + W)
    x = _bf(torch.randn(N, C, H, W, generator=g)).to(DEV)
    go = _bf(torch.randn(N, C, H // 2, W // 2, generator=g)).to(DEV)
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 2, 2)
    (ref * go).sum().backward()
    xm = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = B.MaxPool2x2()(xm)
    (out.float() * go).sum().backward()
    assert torch.equal(out.float(), ref)                       # a selection: exact
    assert torch.equal(xm.grad.float(), xr.grad)


@pytest.mark.parametrize("nin,ew", [(2, [1.0, 1.0]), (3, [0.7, -0.3, 1.9]), (2, [-1.0, -2.0]), (3, [0.0, 2.0, 0.5])])
def test_fastattn_against_torch(nin, ew):
    g = torch.Generator().manual_seed(nin * 7 + int(ew[0] * 10))
    xs = [_bf(torch.randn(2, 13, 11, 64, generator=g)).to(DEV) for _ in range(nin)]
    go = _bf(torch.randn(2, 13, 11, 64, generator=g)).to(DEV)
    w = torch.tensor(ew, device=DEV, requires_grad=True)
    xr = [x.clone().requires_grad_(True) for x in xs]
    wr = F.relu(w)
    ref = sum(xr[i] * wr[i] / (wr.sum() + 0.0001) for i in range(nin))
    (ref * go).sum().backward()
    w_ref = w.grad.clone()
    w2 = torch.tensor(ew, device=DEV, requires_grad=True)
    xm = [x.to(torch.bfloat16).requires_grad_(True) for x in xs]
    out = B._FastAttnFn.apply(w2, *xm)
    (out.float() * go).sum().backward()
    _close(out.float(), ref, 5e-3, "out")
    for i in range(nin):
        _close(xm[i].grad.float(), xr[i].grad, 5e-3, f"dx{i}") if float(xr[i].grad.abs().max()) > 0 else None
        assert float(xr[i].grad.abs().max()) > 0 or float(xm[i].grad.float().abs().max()) == 0
    tol = 2e-3 * float(sum((go * x).abs().sum() for x in xs)) / max(sum(max(e, 0) for e in ew), 1e-4)
    assert float((w2.grad - w_ref).abs().max()) <= tol + 1e-6, (w2.grad, w_ref)
    assert all(float(w2.grad[i]) == 0.0 for i in range(nin) if ew[i] <= 0)


def test_swish_against_torch():
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(2, 24, 9, 7, generator=g) * 3).to(DEV)
    go = _bf(torch.randn(2, 24, 9, 7, generator=g)).to(DEV)
    xr = x.clone().requires_grad_(True)
    ref = xr * xr.sigmoid()
    (ref * go).sum().backward()
    xm = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = B.swish(xm)
    (out.float() * go).sum().backward()
    _close(out.float(), ref, 5e-3, "swish")
    _close(xm.grad.float(), xr.grad, 5e-3, "swish'")


class _Feats(Backbone):
    def __init__(self, chans):
        super().__init__()
        self.chans = chans

    def output_shape(self):
        return {f"res{i + 3}": ShapeSpec(channels=c, stride=8 << i) for i, c in enumerate(self.chans)}

    def forward(self, x):
        return x


class _Rnd(torch.autograd.Function):
    """bf16 storage of an activation and of its gradient (what the device path does between kernels)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def _twin(net, xs, rnd):
    """the same network through plain torch fp32 ops, walking OUR module tree (so the parameters are shared); `rnd` is the
    identity (must reproduce the reference golden) or _Rnd.apply (the bf16 storage noise the device path is entitled to)"""
    def conv(m, x):
        return rnd(F.conv2d(x, rnd(m.weight), m.bias, m.stride, m.padding, 1, m.groups))

    def cba(m, x):
        x = conv(m.conv_pw, conv(m.conv_dw, x)) if isinstance(m, B.SeparableConv2d) else conv(m.conv, x)
        if m.bn is not None:
            x = rnd(F.group_norm(x, m.bn.num_groups, m.bn.weight, m.bn.bias, m.bn.eps))
        return rnd(x * x.sigmoid()) if m.act is not None else x

    def resample(m, x):
        for nm, c in m.named_children():
            x = cba(c, x) if nm == "conv" else (F.max_pool2d(x, 2, 2) if nm == "downsample" else F.interpolate(x, scale_factor=2.0))
        return x

    x = [xs[f] for f in net.in_features]
    for m in net.resample:
        x.append(resample(m, x[-1]))
    for cell in net.cell:
        x = list(x)
        for node in cell.fnode:
            cmb = node.combine
            nodes = [resample(cmb.resample[str(o)], x[o]) for o in cmb.inputs_offsets]
            w = F.relu(cmb.edge_weights)
            y = rnd(sum(nodes[i] * w[i] / (w.sum() + 0.0001) for i in range(len(nodes))))
            y = rnd(y * y.sigmoid())
            x.append(cba(node.after_combine.conv, y))
        x = x[-net.num_levels:]
    return dict(zip(net._out_features, x))


@pytest.mark.parametrize("name", list(BIFPN_CASES))
def test_bifpn_against_reference_golden(golden_dir, name):
    """three runs of the same parameters and inputs: (a) plain torch fp32 = must BE the reference golden (pins the twin),
    (b) plain torch with every activation / gradient stored in bf16 = the storage noise floor, (c) the kernels.
    (c) is held to the golden within the floor that (b) measures, and to (b) itself tightly."""
    gold = np.load(os.path.join(golden_dir, "bifpn.npz"))
    kw = BIFPN_CASES[name]
    feats, gos = synth_bifpn_case(out_channels=kw["out_channels"])
    net = B.BiFPN(cfg=None, bottom_up=_Feats([v.shape[1] for v in feats.values()]), in_features=list(feats.keys()), norm="GN",
                  num_levels=5, **kw)
    assert [f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items()] == list(gold[name + "_keys"])   # drop-in checkpoints
    net.load_state_dict(bifpn_state_dict(net))
    net.to(DEV)
    params = dict(net.named_parameters())

    def run(kind):
        net.zero_grad()
        if kind == "kernels":
            xs = {k: v.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                  for k, v in feats.items()}
            out = net(xs)
        else:
            xs = {k: v.to(DEV).requires_grad_(True) for k, v in feats.items()}
            out = _twin(net, xs, (lambda t: t) if kind == "fp32" else _Rnd.apply)
        assert list(out.keys()) == ["p3", "p4", "p5", "p6", "p7"]
        sum((out[k].float() * gos[k].to(DEV)).sum() for k in out).backward()
        res = {f"out_{k}": v.detach().float().cpu() for k, v in out.items()}
        res.update({f"dx_{k}": v.grad.float().cpu() for k, v in xs.items()})
        res.update({f"grad_{k}": p.grad.float().cpu() for k, p in params.items() if f"{name}_grad_{k}" in gold.files})
        return res

    fp32, floor, got = run("fp32"), run("bf16"), run("kernels")
    report = []
    for key in fp32:
        ref = torch.from_numpy(gold[f"{name}_{key}"])
        assert got[key].shape == ref.shape
        if key.endswith("edge_weights"):
            scale = float(ref.abs().max()) + 1e-6
            assert float((fp32[key] - ref).abs().max()) <= 2e-3 * scale + 1e-3, key
            e_floor = float((floor[key] - ref).abs().max())
            e_got = float((got[key] - ref).abs().max())
            assert e_got <= 3.0 * e_floor + 0.01 * scale + 0.05, (key, e_got, e_floor, got[key], ref)
            dead = [i for i in range(len(ref)) if float(params[key[5:]][i].detach()) <= 0]
            assert all(float(got[key][i]) == 0.0 == float(ref[i]) for i in dead)
            continue
        if float(ref.abs().max()) == 0.0:          # a branch behind an edge weight <= 0: relu cuts it off exactly
            assert float(got[key].abs().max()) == 0.0 and float(fp32[key].abs().max()) == 0.0, key
            continue
        c32, cfl, cgot, cpair = _cos(fp32[key], ref), _cos(floor[key], ref), _cos(got[key], ref), _cos(got[key], floor[key])
        report.append((key, round(1 - cfl, 6), round(1 - cgot, 6), round(1 - cpair, 6)))
        assert c32 >= 0.99999, (key, c32)                                  # the twin IS the reference
        assert 1 - cgot <= 2.0 * (1 - cfl) + 2e-4, (key, cfl, cgot)        # the kernels stay inside the storage noise floor
        assert cgot >= 0.99, (key, cgot)
    print(name, "1-cos (floor vs golden, kernels vs golden, kernels vs floor):", report)


def test_build_resnet_bifpn_backbone_runs():
    from yolov7_d2_amd.config import add_yolo_config, get_cfg
    from yolov7_d2_amd.d2shim import build_backbone
    cfg = add_yolo_config(get_cfg())
    cfg.MODEL.BACKBONE.NAME = "build_resnet_bifpn_backbone"
    cfg.MODEL.RESNETS.OUT_FEATURES = ["res3", "res4", "res5"]
    cfg.MODEL.FPN.IN_FEATURES = ["res3", "res4", "res5"]
    cfg.MODEL.BIFPN.NUM_BIFPN = 2
    net = build_backbone(cfg).to(DEV)
    assert net.size_divisibility == 128
    shp = net.output_shape()
    assert [shp[k].stride for k in ("p3", "p4", "p5", "p6", "p7")] == [8, 16, 32, 64, 128] and shp["p3"].channels == 160
    x = torch.randn(2, 3, 256, 384, device=DEV)
    out = net(x)
    assert [tuple(out[f"p{l}"].shape) for l in range(3, 8)] == [(2, 160, 256 >> l, 384 >> l) for l in range(3, 8)]
    sum(o.float().square().mean() for o in out.values()).backward()
    grads = [p.grad for p in net.parameters() if p.requires_grad]
    assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads)
    with pytest.raises(L.MI355Error):
        net.cpu()(x.cpu())
