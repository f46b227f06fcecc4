"""Building blocks of the YOLOX path with the reference's names and state_dict keys
(yolov7/modeling/backbone/layers/wrappers.py:60-220): BaseConv, Bottleneck, CSPLayer, SPPBottleneck,
Focus.  The nn.Conv2d / nn.BatchNorm2d children only HOLD parameters and buffers (same keys as the
reference: `conv.weight`, `bn.weight`, `bn.running_mean`, ...); compute is emitted into a PlanBuilder
(`emit`) and runs in libmi355det.  Calling `forward` directly is an error: there is no eager/CPU path.
"""
import torch
from torch import nn


class EmitCtx:
    """what a module needs to emit itself: the plan builder and the parameter-gradient lookup"""

    def __init__(self, builder, params):
        self.b, self.params = builder, params

    def g(self, p):
        return self.params.grad_of(p) if self.b.training else None


class _NoEager(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(
            f"{type(self).__name__} has no eager forward: it is executed by the MI355X plan "
            "(yolov7_d2_amd.modeling.yolox.YOLOX / CSPDarknet.forward). No CPU fallback by design.")


def _bn_dict(ctx, bn):
    return dict(gamma=bn.weight, beta=bn.bias, rm=bn.running_mean, rv=bn.running_var, nbt=bn.num_batches_tracked,
                eps=bn.eps, momentum=bn.momentum, ggamma=ctx.g(bn.weight), gbeta=ctx.g(bn.bias))


class BaseConv(_NoEager):
    """Conv2d -> BatchNorm2d -> SiLU (wrappers.py:60-83)"""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu"):
        super().__init__()
        if bias or act != "silu":
            raise NotImplementedError("MI355X path: BaseConv without bias, SiLU (what every reference config builds)")
        if groups != 1 and not (groups == in_channels == out_channels and ksize == 3):
            raise NotImplementedError("grouped conv: only depthwise 3x3 (the dconv of DWConv) is built")
        pad = (ksize - 1) // 2
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ksize, stride=stride, padding=pad, groups=groups,
                              bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = nn.SiLU(inplace=True)
        self.ksize, self.stride, self.groups = ksize, stride, groups

    def emit(self, ctx, x, tag, out=None, res=None, dgrad_pair=None):
        return ctx.b.base_conv(tag, x, self.conv.weight, _bn_dict(ctx, self.bn), self.ksize, self.stride,
                               ctx.g(self.conv.weight), out=out, res=res, act=1, groups=self.groups, dgrad_pair=dgrad_pair)


class DWConv(_NoEager):
    """Depthwise Conv + Conv (wrappers.py:86-102): BaseConv(C, C, k, stride, groups=C) -> BaseConv(C, Cout, 1)"""

    def __init__(self, in_channels, out_channels, ksize, stride=1, act="silu"):
        super().__init__()
        self.dconv = BaseConv(in_channels, in_channels, ksize=ksize, stride=stride, groups=in_channels, act=act)
        self.pconv = BaseConv(in_channels, out_channels, ksize=1, stride=1, groups=1, act=act)

    def emit(self, ctx, x, tag, out=None, res=None):
        h = self.dconv.emit(ctx, x, tag + ".dconv")
        return self.pconv.emit(ctx, h, tag + ".pconv", out=out, res=res)


class Bottleneck(_NoEager):
    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        Conv = DWConv if depthwise else BaseConv
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = Conv(hidden, out_channels, 3, stride=1, act=act)
        self.use_add = shortcut and in_channels == out_channels

    def emit(self, ctx, x, tag, out=None):
        h = self.conv1.emit(ctx, x, tag + ".conv1")
        return self.conv2.emit(ctx, h, tag + ".conv2", out=out, res=x if self.use_add else None)


class SPPBottleneck(_NoEager):
    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), activation="silu"):
        super().__init__()
        assert tuple(kernel_sizes) == (5, 9, 13)
        hidden = in_channels // 2
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=activation)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=ks, stride=1, padding=ks // 2) for ks in kernel_sizes])
        self.conv2 = BaseConv(hidden * 4, out_channels, 1, stride=1, act=activation)
        self.hidden = hidden

    def emit(self, ctx, x, tag, out=None):
        h = self.hidden
        cat = ctx.b.new_act(x.N, x.H, x.W, 4 * h, tag + ".cat")
        s0 = cat.slice(0, h)
        self.conv1.emit(ctx, x, tag + ".conv1", out=s0)
        ctx.b.spp_into(tag + ".pool", s0, cat.slice(h, 2 * h), cat.slice(2 * h, 3 * h), cat.slice(3 * h, 4 * h))
        return self.conv2.emit(ctx, cat, tag + ".conv2", out=out)


class CSPLayer(_NoEager):
    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, stride=1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act) for _ in range(n)])
        self.hidden = hidden

    def emit(self, ctx, x, tag, out=None):
        h = self.hidden
        cat = ctx.b.new_act(x.N, x.H, x.W, 2 * h, tag + ".cat")
        n = len(self.m)
        b = ctx.b
        if getattr(b, "csp_lanes", False):
            # conv1 and conv2 read the same x and are independent: two lanes of a parallel region (grouped launches,
            # Plan._group_lanes).  Their data gradients are ONE convolution over [dy1 | dy2] (PlanBuilder.DgradPair; where
            # that does not apply they stay two launches, the second accumulating, in order).  The reference runs conv2
            # after the bottlenecks (same values, it only reads x).
            pair = b.dgrad_pair(tag + ".split", x, [h, h]) if hasattr(b, "dgrad_pair") else None
            b.par_begin(tag + ".split")
            with b.on_lane(0):
                t = self.conv1.emit(ctx, x, tag + ".conv1", out=cat.slice(0, h) if n == 0 else None,
                                    dgrad_pair=(pair, 0) if pair else None)
            with b.on_lane(1):
                self.conv2.emit(ctx, x, tag + ".conv2", out=cat.slice(h, 2 * h), dgrad_pair=(pair, 1) if pair else None)
            b.par_end(tag + ".split")
            for i, blk in enumerate(self.m):
                t = blk.emit(ctx, t, f"{tag}.m.{i}", out=cat.slice(0, h) if i == n - 1 else None)
            return self.conv3.emit(ctx, cat, tag + ".conv3", out=out)
        t = self.conv1.emit(ctx, x, tag + ".conv1", out=cat.slice(0, h) if n == 0 else None)
        for i, blk in enumerate(self.m):
            t = blk.emit(ctx, t, f"{tag}.m.{i}", out=cat.slice(0, h) if i == n - 1 else None)
        self.conv2.emit(ctx, x, tag + ".conv2", out=cat.slice(h, 2 * h))
        return self.conv3.emit(ctx, cat, tag + ".conv3", out=out)


class Focus(_NoEager):
    """space-to-depth + conv (wrappers.py:202-220); the slicing/concat is fused into the input packer"""

    def __init__(self, in_channels, out_channels, ksize=1, stride=1, act="silu"):
        super().__init__()
        assert in_channels == 3
        self.conv = BaseConv(in_channels * 4, out_channels, ksize, stride, act=act)

    def emit(self, ctx, image, N, H, W, tag):
        x = ctx.b.focus(image, N, H, W)
        return self.conv.emit(ctx, x, tag + ".conv")
