"""BiFPN neck on the MI355X kernels - drop-ins, with the reference's names, constructor arguments, config keys and
state_dict keys, for

    get_fpn_config, Swish, ConvBnAct2d, SeparableConv2d, ResampleFeatureMap,
    FpnCombine, BiFpnLayer, BiFPN                   yolov7/modeling/neck/bifpn.py:29-395
    build_resnet_bifpn_backbone                     yolov7/modeling/neck/bifpn.py:459-479   (MODEL.BIFPN.*: config.py:34-39)

How it maps to the kernels (activations are bf16 NCHW tensors in channels_last memory = NHWC):
  * every 1x1 / 3x3 convolution                                        -> torch.ops.mi355.conv2d (implicit-GEMM MFMA)
  * the depthwise 3x3 of SeparableConv2d (MODEL.BIFPN.SEPARABLE_CONV)  -> mi_dwconv3x3_fwd / _dgrad / _wgrad
  * get_norm("GN") = nn.GroupNorm(32, C), the default MODEL.BIFPN.NORM  -> mi_groupnorm_fwd / _bwd
  * Swish                                                              -> mi_ew_bf16 ops 5 / 6
  * nn.MaxPool2d(2, 2) / nn.UpsamplingNearest2d(2) of the resamplers   -> mi_maxpool2x2_* / mi_upsample2x_*
  * the "fastattn" weighted fusion with its learnable edge weights     -> mi_fastattn_fwd / _bwd
No CPU path: device tensors only.  Norms other than "GN" and "" are rejected (the reference's default and only configured
value for BiFPN is GN; BN here would be SyncBN-less per-GPU statistics over 5 tiny levels).
"""
import ctypes as C
import math
from collections import OrderedDict

import torch
from torch import nn

from .. import _lib as L
from ..d2shim import BACKBONE_REGISTRY, Backbone, ShapeSpec
from .resnet import build_resnet_backbone
from .sparseinst import _Up2, _nhwc


def get_fpn_config(base_reduction=8):
    """bifpn.py:29-45: the 8 nodes of one BiFPN cell (top-down p6..p3, then bottom-up p4..p7) with fast attention"""
    r = base_reduction
    return {
        "nodes": [
            {"reduction": r << 3, "inputs_offsets": [3, 4]},
            {"reduction": r << 2, "inputs_offsets": [2, 5]},
            {"reduction": r << 1, "inputs_offsets": [1, 6]},
            {"reduction": r, "inputs_offsets": [0, 7]},
            {"reduction": r << 1, "inputs_offsets": [1, 7, 8]},
            {"reduction": r << 2, "inputs_offsets": [2, 6, 9]},
            {"reduction": r << 3, "inputs_offsets": [3, 5, 10]},
            {"reduction": r << 4, "inputs_offsets": [4, 11]},
        ],
        "weight_method": "fastattn",
    }


# ------------------------------------------------------------------------------------------------ op wrappers (NHWC bf16)
def _need_cuda(x, what):
    if not x.is_cuda:
        raise L.MI355Error(f"{what}: the MI355X path needs device tensors (no CPU fallback)")


class _SwishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        a = a.contiguous()
        y = torch.empty_like(a)
        L.check(L.lib().mi_ew_bf16(a.data_ptr(), None, y.data_ptr(), a.numel(), 5, L.stream_ptr()), "mi_ew_bf16 swish")
        ctx.save_for_backward(a)
        return y

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(g)
        L.check(L.lib().mi_ew_bf16(g.data_ptr(), a.data_ptr(), out.data_ptr(), g.numel(), 6, L.stream_ptr()), "mi_ew_bf16 swish'")
        return out


def swish(x, inplace=False):
    """bifpn.py:48-51 on an NCHW (channels_last) tensor; `inplace` only names the reference's memory optimisation"""
    _need_cuda(x, "swish")
    return _SwishFn.apply(_nhwc(x)).permute(0, 3, 1, 2)


class Swish(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return swish(x, self.inplace)


class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xh, gamma, beta, G, eps):
        N, H, W, Cc = xh.shape
        lib = L.lib()
        y = torch.empty_like(xh)
        mr = torch.empty(N, G, 2, dtype=torch.float32, device=xh.device)
        ws = torch.empty(int(lib.mi_groupnorm_ws_bytes(N, Cc)) // 8, dtype=torch.float64, device=xh.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        L.check(lib.mi_groupnorm_fwd(xh.data_ptr(), Cc, N, H * W, Cc, G, g32.data_ptr(), b32.data_ptr(), float(eps),
                                     y.data_ptr(), Cc, mr.data_ptr(), ws.data_ptr(), L.stream_ptr()), "mi_groupnorm_fwd")
        ctx.save_for_backward(xh, g32, mr)
        ctx.G = G
        return y

    @staticmethod
    def backward(ctx, g):
        xh, g32, mr = ctx.saved_tensors
        N, H, W, Cc = xh.shape
        lib = L.lib()
        g = g.contiguous()
        dx = torch.empty_like(xh)
        dgamma = torch.empty(Cc, dtype=torch.float32, device=xh.device)
        dbeta = torch.empty(Cc, dtype=torch.float32, device=xh.device)
        ws = torch.empty(int(lib.mi_groupnorm_ws_bytes(N, Cc)) // 8, dtype=torch.float64, device=xh.device)
        L.check(lib.mi_groupnorm_bwd(g.data_ptr(), Cc, xh.data_ptr(), Cc, N, H * W, Cc, ctx.G, g32.data_ptr(), mr.data_ptr(),
                                     dx.data_ptr(), Cc, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), L.stream_ptr()),
                "mi_groupnorm_bwd")
        return dx, dgamma, dbeta, None, None


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm (same parameters / state_dict keys) on mi_groupnorm_*: NCHW (channels_last) bf16 in and out"""

    def forward(self, x):
        _need_cuda(x, "GroupNorm")
        if self.num_channels % 8 or not self.affine:
            raise L.MI355Error("GroupNorm: channel count must be a multiple of 8, affine=True")
        return _GroupNormFn.apply(_nhwc(x), self.weight, self.bias, self.num_groups, self.eps).permute(0, 3, 1, 2)


def get_norm(norm, out_channels):
    """detectron2.layers.batch_norm.get_norm restricted to what MODEL.BIFPN.NORM is configured with"""
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return GroupNorm(32, out_channels)
    raise NotImplementedError(f"BiFPN norm {norm!r}: 'GN' (the reference's MODEL.BIFPN.NORM) or ''")


class _MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xh):
        N, H, W, Cc = xh.shape
        y = torch.empty(N, H // 2, W // 2, Cc, dtype=torch.bfloat16, device=xh.device)
        L.check(L.lib().mi_maxpool2x2_fwd(xh.data_ptr(), Cc, y.data_ptr(), Cc, N, H, W, Cc, L.stream_ptr()), "mi_maxpool2x2_fwd")
        ctx.save_for_backward(xh)
        return y

    @staticmethod
    def backward(ctx, g):
        (xh,) = ctx.saved_tensors
        N, H, W, Cc = xh.shape
        g = g.contiguous()
        dx = torch.zeros_like(xh) if (H | W) & 1 else torch.empty_like(xh)
        L.check(L.lib().mi_maxpool2x2_bwd(xh.data_ptr(), Cc, g.data_ptr(), Cc, dx.data_ptr(), Cc, N, H, W, Cc, L.stream_ptr()),
                "mi_maxpool2x2_bwd")
        return dx


class MaxPool2x2(nn.Module):
    """nn.MaxPool2d(kernel_size=2, stride=2)"""

    def forward(self, x):
        _need_cuda(x, "MaxPool2x2")
        return _MaxPool2Fn.apply(_nhwc(x)).permute(0, 3, 1, 2)


class UpsamplingNearest2x(nn.Module):
    """nn.UpsamplingNearest2d(scale_factor=2)"""

    def forward(self, x):
        _need_cuda(x, "UpsamplingNearest2x")
        return _Up2.apply(_nhwc(x)).permute(0, 3, 1, 2)


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])


class _FastAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ew, *xs):
        xs = [x.contiguous() for x in xs]
        ew32 = ew.detach().float().contiguous()
        out = torch.empty_like(xs[0])
        L.check(L.lib().mi_fastattn_fwd(_ptr_array(xs), len(xs), ew32.data_ptr(), out.data_ptr(), out.numel(), L.stream_ptr()),
                "mi_fastattn_fwd")
        ctx.save_for_backward(ew32, *xs)
        return out

    @staticmethod
    def backward(ctx, g):
        ew32, *xs = ctx.saved_tensors
        lib = L.lib()
        g = g.contiguous()
        dxs = [torch.empty_like(x) if ctx.needs_input_grad[1 + i] else None for i, x in enumerate(xs)]
        dew = torch.empty(len(xs), dtype=torch.float32, device=g.device)
        ws = torch.empty(int(lib.mi_fastattn_ws_bytes()) // 4, dtype=torch.float32, device=g.device)
        L.check(lib.mi_fastattn_bwd(_ptr_array(xs), len(xs), ew32.data_ptr(), g.data_ptr(), _ptr_array(dxs), dew.data_ptr(),
                                    ws.data_ptr(), g.numel(), L.stream_ptr()), "mi_fastattn_bwd")
        return (dew, *dxs)


class _DwConvFn(torch.autograd.Function):
    """3x3 depthwise, stride 1, pad 1, no bias, NHWC bf16"""

    @staticmethod
    def forward(ctx, xh, w):
        N, H, W, Cc = xh.shape
        w32 = w.detach().float().contiguous()
        y = torch.empty_like(xh)
        L.check(L.lib().mi_dwconv3x3_fwd(xh.data_ptr(), Cc, w32.data_ptr(), y.data_ptr(), Cc, N, H, W, Cc, 1, H, W, None, 0,
                                         L.stream_ptr()), "mi_dwconv3x3_fwd")
        ctx.save_for_backward(xh, w32)
        return y

    @staticmethod
    def backward(ctx, g):
        xh, w32 = ctx.saved_tensors
        N, H, W, Cc = xh.shape
        lib = L.lib()
        g = g.contiguous()
        dx = torch.empty_like(xh)
        L.check(lib.mi_dwconv3x3_dgrad(g.data_ptr(), Cc, w32.data_ptr(), dx.data_ptr(), Cc, N, H, W, Cc, 1, H, W, 0,
                                       L.stream_ptr()), "mi_dwconv3x3_dgrad")
        nb = int(lib.mi_dwconv3x3_wgrad_ws_bytes(Cc))
        ws = torch.empty(nb // 4, dtype=torch.float32, device=g.device)
        dw = torch.empty(Cc, 1, 3, 3, dtype=torch.float32, device=g.device)
        L.check(lib.mi_dwconv3x3_wgrad(xh.data_ptr(), Cc, g.data_ptr(), Cc, N, H, W, Cc, 1, H, W, ws.data_ptr(), nb,
                                       dw.data_ptr(), L.stream_ptr()), "mi_dwconv3x3_wgrad")
        return dx, dw


# ------------------------------------------------------------------------------------------------ modules
class Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d without norm / activation (how bifpn.py uses it): parameters and keys of nn.Conv2d"""

    def forward(self, x):
        _need_cuda(x, "Conv2d")
        if self.groups == 1:
            return torch.ops.mi355.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0])
        if not (self.groups == self.in_channels == self.out_channels and self.kernel_size == (3, 3) and self.stride == (1, 1)
                and self.padding == (1, 1) and self.bias is None and self.in_channels % 8 == 0):
            raise NotImplementedError("Conv2d: grouped convolutions other than the 3x3 depthwise of SeparableConv2d")
        return _DwConvFn.apply(_nhwc(x), self.weight).permute(0, 3, 1, 2)


class SequentialAppend(nn.Sequential):
    def forward(self, x):
        for module in self:
            x.append(module(x))
        return x


class SequentialAppendLast(nn.Sequential):
    def forward(self, x):
        for module in self:
            x.append(module(x[-1]))
        return x


class ConvBnAct2d(nn.Module):
    """bifpn.py:84-103: conv (bias only without a norm) -> norm -> activation"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, padding="", bias=False, norm="",
                 act_layer=Swish):
        super().__init__()
        self.conv = Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2,
                           bias=(norm == ""))
        self.bn = get_norm(norm, out_channels)
        self.act = None if act_layer is None else act_layer(inplace=True)

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.act is not None:
            x = self.act(x)
        return x


class SeparableConv2d(nn.Module):
    """bifpn.py:106-146: depthwise 3x3 -> pointwise 1x1 -> norm -> activation"""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, padding="", bias=False,
                 channel_multiplier=1.0, pw_kernel_size=1, act_layer=Swish, norm=""):
        super().__init__()
        mid = int(in_channels * channel_multiplier)
        self.conv_dw = Conv2d(in_channels, mid, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2, bias=bias,
                              groups=out_channels)
        self.conv_pw = Conv2d(mid, out_channels, kernel_size=pw_kernel_size, padding=pw_kernel_size // 2, bias=(norm == ""))
        self.bn = get_norm(norm, out_channels)
        self.act = None if act_layer is None else act_layer(inplace=True)

    def forward(self, x):
        x = self.conv_pw(self.conv_dw(x))
        if self.bn is not None:
            x = self.bn(x)
        if self.act is not None:
            x = self.act(x)
        return x


class ResampleFeatureMap(nn.Sequential):
    """bifpn.py:149-190: optional 1x1 projection to the pyramid width, then 2x2 max-pool (coarser) or nearest x2 (finer)"""

    def __init__(self, in_channels, out_channels, reduction_ratio=1.0, pad_type="", pooling_type="max", norm="", apply_bn=False,
                 conv_after_downsample=False, redundant_bias=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.reduction_ratio, self.conv_after_downsample = reduction_ratio, conv_after_downsample
        conv = None
        if in_channels != out_channels:
            conv = ConvBnAct2d(in_channels, out_channels, kernel_size=1, padding=pad_type, norm=norm if apply_bn else "",
                               bias=not apply_bn or redundant_bias, act_layer=None)
        if reduction_ratio > 1:
            if int(reduction_ratio) != 2:
                raise NotImplementedError("ResampleFeatureMap: max-pool stride 2 (adjacent pyramid levels)")
            if conv is not None and not conv_after_downsample:
                self.add_module("conv", conv)
            self.add_module("downsample", MaxPool2x2())
            if conv is not None and conv_after_downsample:
                self.add_module("conv", conv)
        else:
            if conv is not None:
                self.add_module("conv", conv)
            if reduction_ratio < 1:
                if int(1 // reduction_ratio) != 2:
                    raise NotImplementedError("ResampleFeatureMap: nearest upsampling x2 (adjacent pyramid levels)")
                self.add_module("upsample", UpsamplingNearest2x())


class FpnCombine(nn.Module):
    """bifpn.py:193-247"""

    def __init__(self, feature_info, fpn_config, fpn_channels, inputs_offsets, target_reduction, pad_type="", pooling_type="max",
                 norm="", apply_bn_for_resampling=False, conv_after_downsample=False, redundant_bias=False, weight_method="attn"):
        super().__init__()
        self.inputs_offsets = inputs_offsets
        self.weight_method = weight_method
        self.resample = nn.ModuleDict()
        for offset in inputs_offsets:
            in_channels = fpn_channels
            if offset < len(feature_info):
                in_channels = feature_info[offset]["num_chs"]
                input_reduction = feature_info[offset]["reduction"]
            else:
                input_reduction = fpn_config["nodes"][offset - len(feature_info)]["reduction"]
            self.resample[str(offset)] = ResampleFeatureMap(
                in_channels, fpn_channels, reduction_ratio=target_reduction / input_reduction, pad_type=pad_type,
                pooling_type=pooling_type, norm=norm, apply_bn=apply_bn_for_resampling,
                conv_after_downsample=conv_after_downsample, redundant_bias=redundant_bias)
        if weight_method in ("attn", "fastattn"):
            self.edge_weights = nn.Parameter(torch.ones(len(inputs_offsets)), requires_grad=True)
        else:
            self.edge_weights = None

    def forward(self, x):
        nodes = [_nhwc(self.resample[str(o)](x[o])) for o in self.inputs_offsets]
        if self.weight_method != "fastattn":
            raise NotImplementedError("FpnCombine: weight_method 'fastattn' (get_fpn_config's only value)")
        return _FastAttnFn.apply(self.edge_weights, *nodes).permute(0, 3, 1, 2)


class BiFpnLayer(nn.Module):
    """bifpn.py:250-304: the 8 fusion nodes of one cell, each combine -> Swish -> 3x3 conv (+ norm)"""

    def __init__(self, feature_info, fpn_config, fpn_channels, num_levels=5, pad_type="", pooling_type="max", norm="",
                 act_layer=Swish, apply_bn_for_resampling=False, conv_after_downsample=True, conv_bn_relu_pattern=False,
                 separable_conv=True, redundant_bias=False):
        super().__init__()
        self.fpn_config = fpn_config
        self.num_levels = num_levels
        self.conv_bn_relu_pattern = False
        self.feature_info = []
        self.fnode = SequentialAppend()
        for i, fnode_cfg in enumerate(fpn_config["nodes"]):
            reduction = fnode_cfg["reduction"]
            layers = OrderedDict()
            layers["combine"] = FpnCombine(
                feature_info, fpn_config, fpn_channels, fnode_cfg["inputs_offsets"], target_reduction=reduction,
                pad_type=pad_type, pooling_type=pooling_type, norm=norm, apply_bn_for_resampling=apply_bn_for_resampling,
                conv_after_downsample=conv_after_downsample, redundant_bias=redundant_bias,
                weight_method=fpn_config["weight_method"])
            self.feature_info.append(dict(num_chs=fpn_channels, reduction=reduction))
            after = OrderedDict()
            if not conv_bn_relu_pattern:
                after["act"] = act_layer(inplace=True)
                conv_bias, conv_act = redundant_bias, None
            else:
                conv_bias, conv_act = False, act_layer
            kw = dict(in_channels=fpn_channels, out_channels=fpn_channels, kernel_size=3, padding=pad_type, bias=conv_bias,
                      norm=norm, act_layer=conv_act)
            after["conv"] = SeparableConv2d(**kw) if separable_conv else ConvBnAct2d(**kw)
            layers["after_combine"] = nn.Sequential(after)
            self.fnode.add_module(str(i), nn.Sequential(layers))
        self.feature_info = self.feature_info[-num_levels::]

    def forward(self, x):
        x = self.fnode(list(x))
        return x[-self.num_levels::]


class BiFPN(Backbone):
    """bifpn.py:307-395: bottom-up backbone -> extra coarser levels -> num_bifpn cells -> {"p3": .., .., "p7": ..}"""

    def __init__(self, cfg, bottom_up, in_features, out_channels, norm="", num_levels=5, num_bifpn=4, separable_conv=False):
        super().__init__()
        assert isinstance(bottom_up, Backbone)
        shapes = bottom_up.output_shape()
        in_strides = [shapes[f].stride for f in in_features]
        in_channels = [shapes[f].channels for f in in_features]
        self.num_levels, self.num_bifpn = num_levels, num_bifpn
        self.bottom_up = bottom_up
        self.in_features = in_features
        self._size_divisibility = 128
        levels = [int(math.log2(s)) for s in in_strides]
        self._out_feature_strides = {"p{}".format(int(math.log2(s))): s for s in in_strides}
        if len(in_features) < num_levels:
            for l in range(num_levels - len(in_features)):
                s = l + levels[-1]
                self._out_feature_strides["p{}".format(s + 1)] = 2 ** (s + 1)
        self._out_features = list(sorted(self._out_feature_strides.keys()))
        self._out_feature_channels = {k: out_channels for k in self._out_features}

        feature_info = [{"num_chs": in_channels[l], "reduction": in_strides[l]} for l in range(len(in_features))]
        fpn_config = get_fpn_config()
        self.resample = SequentialAppendLast()
        for level in range(num_levels):
            if level < len(feature_info):
                in_chs, reduction = in_channels[level], in_strides[level]
            else:
                self.resample.add_module(str(level), ResampleFeatureMap(
                    in_channels=in_chs, out_channels=out_channels, pad_type="same", pooling_type=None, norm=norm,
                    reduction_ratio=2, apply_bn=True, conv_after_downsample=False, redundant_bias=False))
                in_chs, reduction = out_channels, int(reduction * 2)
                feature_info.append(dict(num_chs=in_chs, reduction=reduction))
        self.cell = nn.Sequential()
        for rep in range(num_bifpn):
            layer = BiFpnLayer(feature_info=feature_info, fpn_config=fpn_config, fpn_channels=out_channels,
                               num_levels=num_levels, pad_type="same", pooling_type=None, norm=norm, act_layer=Swish,
                               separable_conv=separable_conv, apply_bn_for_resampling=True, conv_after_downsample=False,
                               conv_bn_relu_pattern=False, redundant_bias=False)
            self.cell.add_module(str(rep), layer)
            feature_info = layer.feature_info

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def output_shape(self):
        return {k: ShapeSpec(channels=self._out_feature_channels[k], stride=self._out_feature_strides[k])
                for k in self._out_features}

    def forward(self, x):
        feats = self.bottom_up(x)
        x = [feats[f] for f in self.in_features]
        assert len(self.resample) == self.num_levels - len(x)
        x = self.resample(x)
        x = self.cell(x)
        return {f: xx for f, xx in zip(self._out_features, x)}


@BACKBONE_REGISTRY.register()
def build_resnet_bifpn_backbone(cfg, input_shape=None):
    """bifpn.py:459-479"""
    bottom_up = build_resnet_backbone(cfg, input_shape)
    b = cfg.MODEL.BIFPN
    return BiFPN(cfg=cfg, bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=b.OUT_CHANNELS, norm=b.NORM,
                 num_levels=b.NUM_LEVELS, num_bifpn=b.NUM_BIFPN, separable_conv=b.SEPARABLE_CONV)
