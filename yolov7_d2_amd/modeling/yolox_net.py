"""CSPDarknet backbone, YOLOPAFPN neck and YOLOXHead with the reference's module names
(yolov7/modeling/backbone/darknetx.py:103-213, yolov7/modeling/neck/yolo_pafpn.py:13-114,
yolov7/modeling/head/yolox_head.py:24-149) — parameter holders + plan emission.
"""
import math

import os

import torch
from torch import nn

from ..d2shim import BACKBONE_REGISTRY, Backbone, ShapeSpec
from .blocks import BaseConv, CSPLayer, DWConv, Focus, SPPBottleneck, _NoEager


class CSPDarknet(Backbone):
    def __init__(self, dep_mul, wid_mul, out_features=("dark3", "dark4", "dark5"), depthwise=False, act="silu"):
        super().__init__()
        assert out_features, "please provide output features of Darknet"
        Conv = DWConv if depthwise else BaseConv     # darknetx.py:113 (MODEL.DARKNET.DEPTH_WISE)
        self.out_features = out_features
        bc = int(wid_mul * 64)
        bd = max(round(dep_mul * 3), 1)
        self.output_shape_dict = dict()
        self.stem = Focus(3, bc, ksize=3, act=act)
        self.dark2 = nn.Sequential(Conv(bc, bc * 2, 3, 2, act=act), CSPLayer(bc * 2, bc * 2, n=bd, depthwise=depthwise, act=act))
        self.output_shape_dict["dark2"] = ShapeSpec(channels=bc * 2)
        self.dark3 = nn.Sequential(Conv(bc * 2, bc * 4, 3, 2, act=act),
                                   CSPLayer(bc * 4, bc * 4, n=bd * 3, depthwise=depthwise, act=act))
        self.output_shape_dict["dark3"] = ShapeSpec(channels=bc * 4)
        self.dark4 = nn.Sequential(Conv(bc * 4, bc * 8, 3, 2, act=act),
                                   CSPLayer(bc * 8, bc * 8, n=bd * 3, depthwise=depthwise, act=act))
        self.output_shape_dict["dark4"] = ShapeSpec(channels=bc * 8)
        self.dark5 = nn.Sequential(Conv(bc * 8, bc * 16, 3, 2, act=act), SPPBottleneck(bc * 16, bc * 16, activation=act),
                                   CSPLayer(bc * 16, bc * 16, n=bd, shortcut=False, depthwise=depthwise, act=act))
        self.output_shape_dict["dark5"] = ShapeSpec(channels=bc * 16)
        self.channels = {"dark2": bc * 2, "dark3": bc * 4, "dark4": bc * 8, "dark5": bc * 16}

    def emit(self, ctx, image, N, H, W, outs=None, tag="backbone"):
        """outs: optional {feature name: TRef slice} the stage outputs are written into (concat elimination)"""
        outs = outs or {}
        res = {}
        x = self.stem.emit(ctx, image, N, H, W, tag + ".stem")
        for name in ("dark2", "dark3", "dark4", "dark5"):
            seq = getattr(self, name)
            x = seq[0].emit(ctx, x, f"{tag}.{name}.0")
            if name == "dark5":
                x = seq[1].emit(ctx, x, f"{tag}.{name}.1")
                x = seq[2].emit(ctx, x, f"{tag}.{name}.2", out=outs.get(name))
            else:
                x = seq[1].emit(ctx, x, f"{tag}.{name}.1", out=outs.get(name))
            res[name] = x
        return {k: v for k, v in res.items() if k in self.out_features}

    def forward(self, x):
        from .yolox import run_backbone_standalone
        return run_backbone_standalone(self, x)

    def output_shape(self):
        return self.output_shape_dict

    @property
    def size_divisibility(self) -> int:
        return 32


@BACKBONE_REGISTRY.register()
def build_cspdarknetx_backbone(cfg, input_shape=None):
    return CSPDarknet(dep_mul=cfg.MODEL.YOLO.DEPTH_MUL, wid_mul=cfg.MODEL.YOLO.WIDTH_MUL,
                      depthwise=cfg.MODEL.DARKNET.DEPTH_WISE, out_features=cfg.MODEL.DARKNET.OUT_FEATURES, act="silu")


class YOLOPAFPN(_NoEager):
    def __init__(self, depth=1.0, width=1.0, in_features=("dark3", "dark4", "dark5"), in_channels=[256, 512, 1024],
                 depthwise=False, act="silu"):
        super().__init__()
        Conv = DWConv if depthwise else BaseConv     # yolo_pafpn.py:26
        self.in_features, self.in_channels = in_features, in_channels
        c0, c1, c2 = (int(c * width) for c in in_channels)
        self.c = (c0, c1, c2)
        n = round(3 * depth)
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")
        self.lateral_conv0 = BaseConv(c2, c1, 1, 1, act=act)
        self.C3_p4 = CSPLayer(2 * c1, c1, n, False, depthwise=depthwise, act=act)
        self.reduce_conv1 = BaseConv(c1, c0, 1, 1, act=act)
        self.C3_p3 = CSPLayer(2 * c0, c0, n, False, depthwise=depthwise, act=act)
        self.bu_conv2 = Conv(c0, c0, 3, 2, act=act)
        self.C3_n3 = CSPLayer(2 * c0, c1, n, False, depthwise=depthwise, act=act)
        self.bu_conv1 = Conv(c1, c1, 3, 2, act=act)
        self.C3_n4 = CSPLayer(2 * c1, c2, n, False, depthwise=depthwise, act=act)

    def alloc(self, ctx, N, H8, W8):
        """concat buffers; returns the slices the backbone should write dark3 / dark4 into"""
        c0, c1, c2 = self.c
        b = ctx.b
        self.P3cat = b.new_act(N, H8, W8, 2 * c0, "neck.P3cat")             # [up(fpn_out1) | dark3]
        self.P4cat = b.new_act(N, H8 // 2, W8 // 2, 2 * c1, "neck.P4cat")   # [up(fpn_out0) | dark4]
        self.N3cat = b.new_act(N, H8 // 2, W8 // 2, 2 * c0, "neck.N3cat")   # [bu_conv2(pan_out2) | fpn_out1]
        self.N4cat = b.new_act(N, H8 // 4, W8 // 4, 2 * c1, "neck.N4cat")   # [bu_conv1(pan_out1) | fpn_out0]
        return {"dark3": self.P3cat.slice(c0, 2 * c0), "dark4": self.P4cat.slice(c1, 2 * c1)}

    def emit(self, ctx, feats, tag="neck"):
        c0, c1, c2 = self.c
        b = ctx.b
        x0 = feats["dark5"]
        # dark3/dark4 already live inside P3cat / P4cat (written there by the backbone)
        fpn_out0 = self.lateral_conv0.emit(ctx, x0, tag + ".lateral_conv0", out=self.N4cat.slice(c1, 2 * c1))
        b.upsample_into(tag + ".up0", fpn_out0, self.P4cat.slice(0, c1))
        f_out0 = self.C3_p4.emit(ctx, self.P4cat, tag + ".C3_p4")
        fpn_out1 = self.reduce_conv1.emit(ctx, f_out0, tag + ".reduce_conv1", out=self.N3cat.slice(c0, 2 * c0))
        b.upsample_into(tag + ".up1", fpn_out1, self.P3cat.slice(0, c0))
        pan_out2 = self.C3_p3.emit(ctx, self.P3cat, tag + ".C3_p3")
        self.bu_conv2.emit(ctx, pan_out2, tag + ".bu_conv2", out=self.N3cat.slice(0, c0))
        pan_out1 = self.C3_n3.emit(ctx, self.N3cat, tag + ".C3_n3")
        self.bu_conv1.emit(ctx, pan_out1, tag + ".bu_conv1", out=self.N4cat.slice(0, c1))
        pan_out0 = self.C3_n4.emit(ctx, self.N4cat, tag + ".C3_n4")
        return (pan_out2, pan_out1, pan_out0)


class YOLOXHead(_NoEager):
    def __init__(self, num_classes, width=1.0, strides=[8, 16, 32], in_channels=[256, 512, 1024], act="silu",
                 depthwise=False):
        super().__init__()
        Conv = DWConv if depthwise else BaseConv     # yolox_head.py:51
        self.n_anchors = 1
        self.num_classes = num_classes
        self.decode_in_inference = True
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.stems = nn.ModuleList()
        hid = int(256 * width)
        for i in range(len(in_channels)):
            self.stems.append(BaseConv(int(in_channels[i] * width), hid, 1, 1, act=act))
            self.cls_convs.append(nn.Sequential(Conv(hid, hid, 3, 1, act=act), Conv(hid, hid, 3, 1, act=act)))
            self.reg_convs.append(nn.Sequential(Conv(hid, hid, 3, 1, act=act), Conv(hid, hid, 3, 1, act=act)))
            self.cls_preds.append(nn.Conv2d(hid, self.n_anchors * num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(hid, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(hid, self.n_anchors * 1, 1, 1, 0))
        self.use_l1 = False
        self.strides = strides

    def initialize_biases(self, prior_prob):
        """yolox_head.py:140-149"""
        v = -math.log((1 - prior_prob) / prior_prob)
        with torch.no_grad():
            for conv in list(self.cls_preds) + list(self.obj_preds):
                conv.bias.fill_(v)

    @staticmethod
    def anchors_for(hw_list, strides):
        """[A][3] = (grid_x, grid_y, stride); anchor index = y*w + x per level (yolox_head.py:233-241)"""
        rows = []
        for (h, w), s in zip(hw_list, strides):
            yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            rows.append(torch.stack((xv.reshape(-1).float(), yv.reshape(-1).float(),
                                     torch.full((h * w,), float(s))), dim=1))
        return torch.cat(rows, 0).contiguous()

    def arena_adjacent(self):
        """ParamArena: obj_preds[k] right behind reg_preds[k] (weights and biases): see emit"""
        out = []
        for k in range(len(self.reg_preds)):
            out += [(f"reg_preds.{k}.weight", f"obj_preds.{k}.weight"), (f"reg_preds.{k}.bias", f"obj_preds.{k}.bias")]
        return out

    def _reg_obj_views(self, ctx, k):
        """([5, C, 1, 1] weight, [5] bias, their gradient views) when reg_preds[k] / obj_preds[k] are back to back in the
        parameter and gradient arenas, else None (MI_HEAD_FUSE_REGOBJ=0: the two convolutions of round 5)"""
        if os.environ.get("MI_HEAD_FUSE_REGOBJ", "1") == "0":
            return None
        reg, obj = self.reg_preds[k], self.obj_preds[k]
        if reg.bias is None or obj.bias is None:
            return None
        C = reg.weight.shape[1]

        def joined(a, b_, shape):
            if a is None or b_ is None:
                return None
            if a.dtype != torch.float32 or not a.is_contiguous() or not b_.is_contiguous():
                return None
            if b_.data_ptr() != a.data_ptr() + a.numel() * 4 or a.untyped_storage().data_ptr() != b_.untyped_storage().data_ptr():
                return None
            st = [1] * len(shape)
            st[0] = shape[1] if len(shape) > 1 else 1
            if len(shape) > 1:
                st[1] = 1
            return torch.as_strided(a.detach(), shape, st)

        w = joined(reg.weight.data, obj.weight.data, (5, C, 1, 1))
        bi = joined(reg.bias.data, obj.bias.data, (5,))
        if w is None or bi is None:
            return None
        if not ctx.b.training:
            return w, bi, None, None
        gw = joined(ctx.g(reg.weight), ctx.g(obj.weight), (5, C, 1, 1))
        gb = joined(ctx.g(reg.bias), ctx.g(obj.bias), (5,))
        if gw is None or gb is None:
            return None
        return w, bi, gw, gb

    def emit(self, ctx, fpn_outs, preds, A, tag="head"):
        nch = 5 + self.num_classes
        a0 = 0
        # the FPN levels - and inside a level the classification and the regression branch - are independent chains
        # (yolox_head.py:160-172 loops over them): the plan zips them into grouped launches (Plan._group_lanes), or, with
        # MI_MULTI_STREAM, runs the small levels on auxiliary streams
        b = ctx.b
        stems = []
        b.par_begin(tag + ".stems")
        for k, x in enumerate(fpn_outs):
            with b.on_stream(k), b.on_lane(k):
                stems.append(self.stems[k].emit(ctx, x, f"{tag}.stems.{k}"))
        b.par_end(tag + ".stems")
        b.par_begin(tag + ".levels")
        for k, x in enumerate(fpn_outs):
            t = stems[k]
            with b.on_stream(k), b.on_lane(2 * k):
                c = self.cls_convs[k][0].emit(ctx, t, f"{tag}.cls_convs.{k}.0")
                c = self.cls_convs[k][1].emit(ctx, c, f"{tag}.cls_convs.{k}.1")
                b.pred_conv(f"{tag}.cls_preds.{k}", c, self.cls_preds[k].weight, self.cls_preds[k].bias,
                            ctx.g(self.cls_preds[k].weight), ctx.g(self.cls_preds[k].bias), preds, A, a0, 5, nch)
            with b.on_stream(k), b.on_lane(2 * k + 1):
                r = self.reg_convs[k][0].emit(ctx, t, f"{tag}.reg_convs.{k}.0")
                r = self.reg_convs[k][1].emit(ctx, r, f"{tag}.reg_convs.{k}.1")
                fused = self._reg_obj_views(ctx, k)
                if fused is not None:
                    # reg_preds + obj_preds (yolox_head.py:166-168: two 1x1 convs over the same tower output, channels 0..3 and
                    # 4 of the prediction row) as ONE convolution with 5 output channels: their parameters lie back to back
                    # in the arena (arena_adjacent), so the weight / bias / gradients are [5, C] / [5] views.  One launch less
                    # forward, one data gradient instead of a gradient + an accumulating one, one split map instead of two.
                    w, bi, gw, gb = fused
                    b.pred_conv(f"{tag}.reg_preds.{k}", r, w, bi, gw, gb, preds, A, a0, 0, nch)
                else:
                    for name, mod, c0 in (("reg_preds", self.reg_preds[k], 0), ("obj_preds", self.obj_preds[k], 4)):
                        b.pred_conv(f"{tag}.{name}.{k}", r, mod.weight, mod.bias, ctx.g(mod.weight), ctx.g(mod.bias),
                                    preds, A, a0, c0, nch)
            a0 += x.H * x.W
        b.par_end(tag + ".levels")
        assert a0 == A
