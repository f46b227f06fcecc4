"""ORACLE (test infrastructure, not product): CPU fp32 restatement of detectron2's ResNet (BasicStem + BottleneckBlock
with FrozenBatchNorm2d, eps 1e-5) over a state_dict with detectron2's key names.  detectron2 is an un-vendored, unpinned
dependency of the reference (readme.md:178 "latest"; call sites meta_arch/detr.py:348, sparseinst.py:63): PARITY UNPINNED -
this follows d2 upstream's modeling/backbone/resnet.py semantics (stride in the 3x3 conv when STRIDE_IN_1X1 is False,
shortcut = 1x1 conv with the block's stride when the channel count changes, ReLU after the residual add)."""
import torch
import torch.nn.functional as F

EPS = 1e-5
BLOCKS = {50: (3, 4, 6, 3)}


def init_state_dict(depth=50, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(p, cin, cout, k):
        std = (2.0 / (cout * k * k)) ** 0.5
        sd[p + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std
        sd[p + ".norm.weight"] = 0.5 + torch.rand(cout, generator=g)
        sd[p + ".norm.bias"] = torch.randn(cout, generator=g) * 0.1
        sd[p + ".norm.running_mean"] = torch.randn(cout, generator=g) * 0.1
        sd[p + ".norm.running_var"] = 0.5 + torch.rand(cout, generator=g)

    conv("stem.conv1", 3, 64, 7)
    cin, bc, cout = 64, 64, 256
    for i, nb in enumerate(BLOCKS[depth]):
        for k in range(nb):
            p = f"res{i + 2}.{k}"
            c0 = cin if k == 0 else cout
            if c0 != cout:
                conv(p + ".shortcut", c0, cout, 1)
            conv(p + ".conv1", c0, bc, 1); conv(p + ".conv2", bc, bc, 3); conv(p + ".conv3", bc, cout, 1)
        cin, bc, cout = cout, bc * 2, cout * 2
    return sd


def _cn(sd, p, x, stride, pad, q, force=None):
    scale = sd[p + ".norm.weight"] * (sd[p + ".norm.running_var"] + EPS).rsqrt()
    shift = sd[p + ".norm.bias"] - sd[p + ".norm.running_mean"] * scale
    y = q(F.conv2d(q(x), q(sd[p + ".weight"] * scale.view(-1, 1, 1, 1)), shift, stride, pad))
    if force is not None and p in force:
        # teacher forcing: the VALUE of this conv output becomes the given one (the ReLU gates and everything downstream
        # then see the same forward state as the implementation under test), the gradient path stays the oracle's own
        y = y + (force[p].to(y.dtype) - y).detach()
    return y


def forward(sd, x, depth=50, stride_in_1x1=False, quant=None, force=None, start=None):
    """-> {"res2".."res5"}; quant: optional storage-rounding emulation (bf16) applied where the product stores.
    force: {conv name ("res3.0.conv1", "res4.2.shortcut", ...): tensor} pins those conv outputs (pre-ReLU) to given values.
    start: (stage name, tensor) = begin at that stage with the tensor as its input (skips the stem and earlier stages)."""
    q = quant if quant is not None else (lambda t: t)
    outs = {}
    if start is None:
        x = q(F.relu(_cn(sd, "stem.conv1", x, 2, 3, q, force)))
        x = F.max_pool2d(x, 3, 2, 1)
    for i, nb in enumerate(BLOCKS[depth]):
        if start is not None:
            if f"res{i + 2}" != start[0] and f"res{i + 2}" not in outs and not outs:
                continue
        if start is not None and f"res{i + 2}" == start[0]:
            x = start[1]
        for k in range(nb):
            p = f"res{i + 2}.{k}"
            stride = 2 if (k == 0 and i > 0) else 1
            s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
            out = q(F.relu(_cn(sd, p + ".conv1", x, s1, 0, q, force)))
            out = q(F.relu(_cn(sd, p + ".conv2", out, s3, 1, q, force)))
            out = _cn(sd, p + ".conv3", out, 1, 0, q, force)
            sc = _cn(sd, p + ".shortcut", x, stride, 0, q, force) if (p + ".shortcut.weight") in sd else x
            x = q(F.relu(out + sc))
        outs[f"res{i + 2}"] = x
    return outs


class _Holder(torch.nn.Module):
    pass


class R50Module(torch.nn.Module):
    """the restatement as an nn.Module tree with detectron2's parameter / buffer names (stem.conv1.weight,
    res3.0.conv2.norm.running_var ...) and the Backbone surface the reference's MaskedBackbone uses (forward ->
    {name: feature}, output_shape() -> {name: .channels / .stride}); stands in for detectron2.modeling.build_backbone when
    the reference's own Detr is executed by path (oracle/gen_golden.py)"""

    def __init__(self, depth=50, out_features=("res2", "res3", "res4", "res5"), stride_in_1x1=False):
        super().__init__()
        self.depth, self.out_features, self.stride_in_1x1 = depth, list(out_features), stride_in_1x1
        for k, v in init_state_dict(depth, 0).items():
            parts = k.split(".")
            m = self
            for name in parts[:-1]:
                if not hasattr(m, name):
                    m.add_module(name, _Holder())
                m = getattr(m, name)
            if ".norm." in k:
                m.register_buffer(parts[-1], v.clone())
            else:
                m.register_parameter(parts[-1], torch.nn.Parameter(v.clone()))

    def forward(self, x):
        sd = dict(list(self.named_parameters()) + list(self.named_buffers()))
        outs = forward(sd, x, self.depth, self.stride_in_1x1)
        return {k: outs[k] for k in self.out_features}

    def output_shape(self):
        import types
        ch = {"res2": 256, "res3": 512, "res4": 1024, "res5": 2048}
        st = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
        return {k: types.SimpleNamespace(channels=ch[k], stride=st[k]) for k in self.out_features}
