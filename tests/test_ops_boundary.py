"""The per-operator boundary: torch.ops.mi355.* registration (CPU: schemas, loud failure without a HIP device, and
registration into a REAL-shaped detectron2 registry) and, on the GPU, every op with autograd against fp32 torch
references on bf16-rounded operands."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import yolov7_d2_amd  # noqa: F401
from yolov7_d2_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ops_are_registered_and_fail_loudly_on_cpu():
    names = {"conv2d", "conv2d_backward", "conv_bn_silu", "conv_bn_silu_backward", "batched_nms", "yolox_loss", "mha",
             "iou_loss_v6"}
    for n in names:
        assert hasattr(torch.ops.mi355, n), n
    s = str(torch.ops.mi355.conv_bn_silu.default._schema)
    assert "running_mean" in s and "-> (Tensor, Tensor, Tensor)" in s
    with pytest.raises(NotImplementedError):      # no CPU kernel registered: no silent fallback
        torch.ops.mi355.batched_nms(torch.zeros(2, 4), torch.zeros(2), torch.zeros(2), 0.5)
    with pytest.raises(NotImplementedError):
        torch.ops.mi355.conv2d(torch.zeros(1, 32, 8, 8), torch.zeros(32, 32, 1, 1), None, 1, 0)


def test_registration_lands_in_detectron2_registries():
    """with a detectron2 package importable (here: a stand-in exposing detectron2's Registry semantics - fvcore Registry:
    register() as decorator, get(), duplicate names rejected), the classes register into ITS registries, which is what
    makes `cfg.MODEL.META_ARCHITECTURE: YOLOX` / `BACKBONE.NAME: build_cspdarknetx_backbone` of the reference's YAMLs
    resolve through detectron2's own build_model / build_backbone"""
    code = textwrap.dedent('''
        import sys, types
        class Registry:                                     # fvcore.common.registry.Registry semantics
            def __init__(self, name): self._name, self._obj_map = name, {}
            def _do_register(self, name, obj):
                assert name not in self._obj_map, "An object named '%s' was already registered in '%s' registry!" % (name, self._name)
                self._obj_map[name] = obj
            def register(self, obj=None):
                if obj is None:
                    def deco(o):
                        self._do_register(o.__name__, o); return o
                    return deco
                self._do_register(obj.__name__, obj)
            def get(self, name):
                ret = self._obj_map.get(name)
                if ret is None: raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
                return ret
            def __contains__(self, n): return n in self._obj_map
        import torch
        from dataclasses import dataclass
        def mod(name, **kw):
            m = types.ModuleType(name); m.__dict__.update(kw); sys.modules[name] = m; return m
        META, BACK = Registry("META_ARCH"), Registry("BACKBONE")
        class Backbone(torch.nn.Module):
            @property
            def size_divisibility(self): return 0
        @dataclass
        class ShapeSpec:
            channels: int = None; height: int = None; width: int = None; stride: int = None
        class Boxes:
            def __init__(self, t): self.tensor = t
        class Instances:
            def __init__(self, image_size, **kw): self.image_size = image_size
        class ImageList: pass
        mod("detectron2"); mod("detectron2.layers", ShapeSpec=ShapeSpec)
        mod("detectron2.modeling", BACKBONE_REGISTRY=BACK, META_ARCH_REGISTRY=META, Backbone=Backbone)
        mod("detectron2.modeling.postprocessing", detector_postprocess=lambda r, h, w: r)
        mod("detectron2.structures", Boxes=Boxes, ImageList=ImageList, Instances=Instances)
        import yolov7_d2_amd as M
        from yolov7_d2_amd import d2shim
        assert d2shim.HAVE_D2
        assert d2shim.META_ARCH_REGISTRY is META and d2shim.BACKBONE_REGISTRY is BACK
        assert META.get("YOLOX") is M.YOLOX
        assert BACK.get("build_cspdarknetx_backbone") is M.build_cspdarknetx_backbone
        assert issubclass(type(M.build_cspdarknetx_backbone(M.yolox_s_cfg(device="cpu"), ShapeSpec(channels=3))), Backbone)
        print("registered:", sorted(META._obj_map), sorted(BACK._obj_map))
    ''')
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "YOLOX" in r.stdout and "build_cspdarknetx_backbone" in r.stdout


# ------------------------------------------------------------------------------------------------------------ GPU
def _bf(t):
    return t.to(torch.bfloat16).float()


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / (b.float().cpu().norm() + 1e-12))


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,bias", [(2, 20, 24, 64, 96, 3, 1, True), (2, 16, 16, 32, 64, 1, 1, False),
                                                     (1, 22, 18, 48, 80, 3, 2, True), (2, 15, 18, 64, 128, 1, 2, False)])
def test_op_conv2d_with_autograd(N, H, W, Cin, Cout, k, s, bias):
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, k, k, generator=g) * 0.1)
    b = torch.randn(Cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    ref = F.conv2d(xr, wr, br, s, (k - 1) // 2)
    go = _bf(torch.randn(ref.shape, generator=g))
    ref.backward(go)
    xd = x.cuda().requires_grad_(True); wd = w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True) if bias else None
    out = torch.ops.mi355.conv2d(xd, wd, bd, s, (k - 1) // 2)
    assert out.shape == ref.shape and out.dtype == torch.bfloat16
    assert _rel(out, ref) < 1e-2
    out.backward(go.cuda().to(torch.bfloat16))
    assert _rel(xd.grad, xr.grad) < 1e-2 and _rel(wd.grad, wr.grad) < 1e-2
    if bias:
        assert _rel(bd.grad, br.grad) < 1e-2


class BaseConv(nn.Module):
    """shape of the reference's BaseConv (yolov7/modeling/backbone/layers/wrappers.py:60-83), restated for the test"""

    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, (k - 1) // 2, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU(inplace=True)

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


@pytest.mark.gpu
def test_patch_base_convs_trains_like_the_eager_modules():
    """an UNMODIFIED BaseConv-shaped module tree, forward re-pointed to torch.ops.mi355.conv_bn_silu: outputs, input /
    parameter gradients and the BatchNorm buffers after one train-mode step, and the eval-mode output, against the
    eager fp32 modules (bf16 activation storage between the layers on the op side: 2e-2)"""
    torch.manual_seed(0)
    net = nn.Sequential(BaseConv(32, 64, 3, 1), BaseConv(64, 48, 1, 1), BaseConv(48, 96, 3, 2))
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(_bf(p))
    import copy
    dev = copy.deepcopy(net).cuda()
    assert ops.patch_base_convs(dev) == 3
    x = _bf(torch.randn(2, 32, 24, 20))
    xr = x.clone().requires_grad_(True)
    net.train()
    ref = net(xr)
    go = _bf(torch.randn(ref.shape))
    ref.backward(go)
    dev.train()
    xd = x.cuda().requires_grad_(True)
    out = dev(xd)
    assert _rel(out, ref) < 2e-2
    out.backward(go.cuda().to(out.dtype))
    assert _rel(xd.grad, xr.grad) < 5e-2
    for (n, p), (_, q) in zip(net.named_parameters(), dev.named_parameters()):
        assert _rel(q.grad, p.grad) < 5e-2, n
    for (n, b), (_, c) in zip(net.named_buffers(), dev.named_buffers()):
        assert _rel(c, b) < 1e-2 if b.dtype.is_floating_point else int(c) == int(b), n
    net.eval(); dev.eval()
    with torch.no_grad():
        assert _rel(dev(x.cuda()), net(x)) < 2e-2


@pytest.mark.gpu
def test_op_yolox_loss_nms_mha_iou():
    import yolox_oracle as O
    # yolox_loss: value + gradient against the oracle
    _, labels = O.synth_batch(2, 160, 160, seed=5, max_gt=6)
    raw, anchors = O.synth_raw(2, [(20, 20), (10, 10), (5, 5)], 6, labels=labels)
    rr = raw.clone().requires_grad_(True)
    res = O.yolox_losses(rr, labels, anchors, 80)
    (res[0] + res[1] + res[2] + res[3]).backward()
    rd = raw.cuda().requires_grad_(True)
    out = ops.yolox_loss(rd, labels.cuda(), anchors.cuda(), 80)
    np.testing.assert_allclose(out[:4].detach().cpu().numpy(), np.array([float(v.detach()) for v in res[:4]]), rtol=1e-4)
    out[:4].sum().backward()
    assert _rel(rd.grad, rr.grad) < 5e-4
    # batched_nms: keep indices exact
    g = torch.Generator().manual_seed(2)
    ctr = torch.rand(500, 2, generator=g) * 300
    wh = 10 + torch.rand(500, 2, generator=g) * 60
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    scores, idxs = torch.rand(500, generator=g), torch.randint(0, 5, (500,), generator=g).float()
    keep = torch.ops.mi355.batched_nms(boxes.cuda(), scores.cuda(), idxs.cuda(), 0.5).cpu()
    assert torch.equal(keep, O.batched_nms(boxes, scores, idxs, 0.5))
    # mha: against softmax attention in fp32
    Lq, Lk, Bn, E, nh = 40, 56, 2, 256, 8
    q, k, v = (_bf(torch.randn(n, Bn, E, generator=g) * 0.5) for n in (Lq, Lk, Lk))
    qd, kd, vd = (t.cuda().to(torch.bfloat16).requires_grad_(True) for t in (q, k, v))
    o = torch.ops.mi355.mha(qd, kd, vd, None, nh)
    qh = q.view(Lq, Bn * nh, E // nh).transpose(0, 1); kh = k.view(Lk, Bn * nh, E // nh).transpose(0, 1)
    vh = v.view(Lk, Bn * nh, E // nh).transpose(0, 1)
    ref = (torch.softmax(qh @ kh.transpose(1, 2) / (E // nh) ** 0.5, -1) @ vh).transpose(0, 1).reshape(Lq, Bn, E)
    assert _rel(o, ref) < 2e-2
    o.float().sum().backward()
    assert qd.grad is not None and torch.isfinite(qd.grad).all()
    # iou_loss_v6 exists as an op and is differentiable
    p = (torch.rand(64, 4, generator=g) * 50 + 10).cuda().requires_grad_(True)
    t = (torch.rand(64, 4, generator=g) * 50 + 10).cuda()
    l = torch.ops.mi355.iou_loss_v6(p, t, "ciou", False, 1e-7)
    l.sum().backward()
    assert l.shape == (64,) and torch.isfinite(p.grad).all()
