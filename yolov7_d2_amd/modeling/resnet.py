"""detectron2's ResNet-50 (`build_resnet_backbone`, BasicStem + BottleneckBlock, FrozenBatchNorm2d) on the MI355X kernels -
the backbone of BASELINE.json configs 4 / 5 (configs/coco/detr/detr_256_6_6_torchvision.yaml:7-10,
configs/coco/sparseinst/Base-SparseInst.yaml:8).  detectron2 is NOT vendored in the reference and not installed here:
module structure, state_dict keys (`stem.conv1.weight`, `stem.conv1.norm.{weight,bias,running_mean,running_var}`,
`res3.0.shortcut.weight`, `res3.0.conv2.norm.*` ...) and semantics are restated from d2 upstream (parity unpinned: the
CPU restatement is oracle/resnet_oracle.py).

How it runs:
  * every Conv2d + FrozenBatchNorm2d pair is ONE implicit-GEMM launch (torch.ops.mi355.conv2d): the frozen affine
    (scale = w / sqrt(var + 1e-5), shift = b - mean * scale) is folded into the packed weight image and the conv bias -
    a frozen norm has no statistics pass at all;
  * the 7x7 stride-2 stem conv (3 -> 64) is a 4x4 conv over the 2x2 space-to-depth image: the Focus packer of the YOLOX
    stem (mi_focus_pack: fp32 NCHW -> bf16 [N, H/2, W/2, 12+4]) followed by a 16-tap conv with the 7x7 weights scattered
    into an 8x8 kernel (row / column -1 are zero) - no 3-channel implicit GEMM with 13/16 of the k-chunk wasted;
  * ReLU / residual add: mi_ew_bf16; MaxPool2d(3, 2, 1): mi_maxpool3x3s2_*;
  * MODEL.BACKBONE.FREEZE_AT (default 2): the stem and res2 run forward-only (no saved activations, no data / weight
    gradients), exactly the layers detectron2 freezes; FREEZE_AT 0 (SparseInst) trains the stem through the 16-tap
    weight-gradient kernel.
Activations are bf16 NCHW tensors in channels_last memory (= NHWC for the kernels).
"""
import ctypes as C
import os

import torch
from torch import nn

from .. import _lib as L
from ..d2shim import BACKBONE_REGISTRY, Backbone, ShapeSpec
from ..ops import WgradBatch, _ConvGeom, _conv_desc, _nchw, _nhwc, _nhwc_v, _pad_last, _run_conv, wgrad_can_defer


class FrozenBatchNorm2d(nn.Module):
    """detectron2.layers.FrozenBatchNorm2d: four BUFFERS, eps 1e-5, y = x * scale + shift"""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def affine(self):
        """(scale, shift), fp32 [C].  The four buffers are constants of a training run: the pair is computed once and kept
        until a buffer is written or replaced (five tiny launches per convolution per step otherwise).

        The two tensors are ADDRESS-STABLE: a changed buffer (load_state_dict, or just a version bump - GraphedTrainStep
        restores every buffer with copy_ after its warm-up passes) recomputes INTO them.  Captured graphs hold their
        addresses as launch arguments; replacing the pair would free memory an earlier capture still reads, and the
        allocator hands it to the next tensor (found with two captured shapes and an eager forward in between: the
        second shape's replays read another tensor's bytes as the BatchNorm shift)."""
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((t.data_ptr(), t._version) for t in bufs)
        c = self.__dict__.get("_affine")
        if c is None or c[0] != key:
            with torch.no_grad():
                scale = (self.weight.float() * (self.running_var.float() + self.eps).rsqrt()).contiguous()
                shift = (self.bias.float() - self.running_mean.float() * scale).contiguous()
                if c is not None and c[1].device == scale.device and c[1].shape == scale.shape:
                    c[1].copy_(scale)
                    c[2].copy_(shift)
                    scale, shift = c[1], c[2]
            c = self.__dict__["_affine"] = (key, scale, shift)
        return c[1], c[2]


def _nhwc_view(x):
    """bf16 NCHW tensor -> its NHWC image (no copy when x is channels_last)"""
    return x.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()


class _EwRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):          # a: contiguous bf16
        y = torch.empty_like(a)
        L.check(L.lib().mi_ew_bf16(a.data_ptr(), None, y.data_ptr(), a.numel(), 1, L.stream_ptr()), "mi_ew_bf16 relu")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(g)
        L.check(L.lib().mi_ew_bf16(g.data_ptr(), y.data_ptr(), out.data_ptr(), g.numel(), 2, L.stream_ptr()), "mi_ew_bf16 relu'")
        return out


class _EwAddRelu(torch.autograd.Function):
    """relu(a + b): the tail of a bottleneck block"""

    @staticmethod
    def forward(ctx, a, b):
        y = torch.empty_like(a)
        L.check(L.lib().mi_ew_bf16(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), 7, L.stream_ptr()), "mi_ew_bf16 add+relu")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(g)
        L.check(L.lib().mi_ew_bf16(g.data_ptr(), y.data_ptr(), out.data_ptr(), g.numel(), 2, L.stream_ptr()), "mi_ew_bf16 relu'")
        return out, out


def _conv_relu_fused():
    """MI_CONV_RELU_FUSE=0 keeps the ReLU after a bottleneck's conv1 / conv2 as its own pass (A/B switch)"""
    return os.environ.get("MI_CONV_RELU_FUSE", "1") != "0"


def _relu(x):
    """NCHW (channels_last) bf16 -> same layout"""
    return _EwRelu.apply(_nhwc_view(x)).permute(0, 3, 1, 2)


def _add_relu(a, b):
    return _EwAddRelu.apply(_nhwc_view(a), _nhwc_view(b)).permute(0, 3, 1, 2)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xh):          # xh bf16 [N,H,W,C] contiguous
        N, H, W, Cc = xh.shape
        y = torch.empty(N, (H + 1) // 2, (W + 1) // 2, Cc, dtype=torch.bfloat16, device=xh.device)
        if not xh.requires_grad:
            L.check(L.lib().mi_maxpool3x3s2_fwd(xh.data_ptr(), Cc, y.data_ptr(), Cc, N, H, W, Cc, L.stream_ptr()), "maxpool fwd")
            return y
        # training through the stem (FREEZE_AT 0): one byte per output element records where its first maximum sits, the
        # backward reads that instead of re-deriving it from x (and x need not be kept)
        code = torch.empty(N, (H + 1) // 2, (W + 1) // 2, Cc, dtype=torch.uint8, device=xh.device)
        L.check(L.lib().mi_maxpool3x3s2_fwd_idx(xh.data_ptr(), Cc, y.data_ptr(), Cc, code.data_ptr(), N, H, W, Cc, L.stream_ptr()),
                "maxpool fwd")
        ctx.save_for_backward(code)
        ctx.in_shape = (N, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, g):
        (code,) = ctx.saved_tensors
        N, H, W, Cc = ctx.in_shape
        g = g.contiguous()
        dx = torch.empty(N, H, W, Cc, dtype=torch.bfloat16, device=g.device)
        L.check(L.lib().mi_maxpool3x3s2_bwd_idx(code.data_ptr(), g.data_ptr(), Cc, dx.data_ptr(), Cc, 0, N, H, W, Cc,
                                                L.stream_ptr()), "maxpool bwd")
        return dx


class Conv2d(nn.Module):
    """detectron2.layers.Conv2d(..., bias=False, norm=FrozenBatchNorm2d) as the reference's configs build it"""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, kernel_size, kernel_size))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")      # c2_msra_fill
        self.norm = FrozenBatchNorm2d(cout)
        self.stride, self.padding = stride, padding

    def forward(self, x, relu=False, add_relu=None):
        """relu=True: the ReLU that follows runs in the convolution's epilogue (MI_CONV_RELU), one launch instead of two.
        The frozen affine is folded into the packed weight image (scale) and the bias (shift).
        add_relu (no-autograd path only): an NCHW tensor of the output's shape; returns relu(bf16(conv + shift) + add_relu) from
        the convolution's epilogue (MI_CONV_ADDRELU): conv3 + shortcut + ReLU of a frozen bottleneck as one launch"""
        scale, shift = self.norm.affine()
        if add_relu is not None and (not _FOLDED_FN() or (self.weight.requires_grad and torch.is_grad_enabled())
                                     or (x.requires_grad and torch.is_grad_enabled())):
            raise L.MI355Error("Conv2d.forward(add_relu=...): only on the frozen / no-autograd path")
        if not _FOLDED_FN():
            w = self.weight * scale.view(-1, 1, 1, 1)
            op = torch.ops.mi355.conv2d_relu if relu else torch.ops.mi355.conv2d
            if not self.weight.requires_grad:
                with torch.no_grad():
                    return op(x, w, shift, self.stride, self.padding)
            return op(x, w, shift, self.stride, self.padding)
        train_w = self.weight.requires_grad and torch.is_grad_enabled()
        if not train_w and not (x.requires_grad and torch.is_grad_enabled()):
            g = _ConvGeom(x.shape, self.weight.shape, self.stride, self.padding)
            with torch.no_grad():
                if self.weight.requires_grad:
                    # a TRAINABLE layer run without autograd (evaluation between training steps): packed per call - its
                    # weight may have been updated by a kernel that torch's version counter does not see
                    # (mi_adamw_step_multi, the arena SGD), so no key could tell a stale image from a fresh one
                    wf = g.pack(self.weight, dgrad=False, scale=scale)[0]
                else:
                    # a frozen layer under a frozen prefix (FREEZE_AT): its packed image is a constant too.  Keyed on the
                    # identity + write counters of the weight and of the four norm buffers (load_state_dict / .to() change
                    # them), not on the address of `scale`: a recomputed scale may be handed the freed one's address
                    key = (self.weight.data_ptr(), self.weight._version, self.norm.__dict__["_affine"][0])
                    c = self.__dict__.get("_image")
                    if c is None or c[0] != key:
                        c = self.__dict__["_image"] = (key, _same_address(c, g.pack(self.weight, dgrad=False, scale=scale)[0]))
                    wf = c[1]
                return _folded_forward(g, x, wf, shift, relu, add_relu)[1]
        return _FoldedConvFn.apply(x, self.weight, scale, shift, self.stride, self.padding, relu)


def _same_address(cached, new):
    """a re-packed constant image goes INTO the tensor the previous one lived in (captured graphs hold its address; see
    FrozenBatchNorm2d.affine)"""
    if cached is not None and cached[1].device == new.device and cached[1].shape == new.shape and cached[1].dtype == new.dtype:
        cached[1].copy_(new)
        return cached[1]
    return new


def _FOLDED_FN():
    """MI_RESNET_FOLDED_FN=0: the round-2 form (mi355::conv2d on `weight * scale`; packs in forward AND backward,
    a bias gradient nobody reads) - A/B switch"""
    return os.environ.get("MI_RESNET_FOLDED_FN", "1") != "0"


def _folded_forward(g, x, wf, shift, relu, add_relu=None):
    xh = g.pad_in(x)
    y = torch.empty(g.N, g.Ho, g.Wo, g.CoutP, dtype=torch.bfloat16, device=x.device)
    b32 = shift
    if g.CoutP != g.Cout:
        b32 = torch.zeros(g.CoutP, dtype=torch.float32, device=x.device)
        b32[: g.Cout] = shift
    ar = None
    if add_relu is not None:
        if g.CoutP != g.Cout or relu:
            raise L.MI355Error("conv + add + ReLU epilogue: channel count must be a multiple of 32 (and no second ReLU)")
        ar = _nhwc_v(add_relu)
    g.fwd(xh, wf, y, bias=b32, relu=relu, add_relu=ar)
    return xh, _nchw(y, g.Cout)


class _FoldedConvFn(torch.autograd.Function):
    """Conv2d + FrozenBatchNorm2d (+ ReLU) of a TRAINABLE layer as one autograd node: y = act(conv(x, W * scale) + shift).
    One pack launch writes the forward and the data-gradient image (the latter is kept for backward), no bias gradient
    (shift is a buffer), the ReLU mask comes from the saved output, gW = scale * gW'."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, stride, padding, relu):
        g = _ConvGeom(x.shape, weight.shape, stride, padding)
        need_dx = x.requires_grad
        wf, wd = g.pack(weight, dgrad=need_dx, scale=scale)
        xh, y = _folded_forward(g, x, wf, shift, relu)
        ctx.g, ctx.relu = g, relu
        ctx.save_for_backward(xh, wd, scale, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, grad):
        xh, wd, scale, y = ctx.saved_tensors
        g = ctx.g
        dyh = _nhwc(grad)
        if ctx.relu:
            yh = _nhwc(y)
            gm = torch.empty_like(dyh)
            L.check(L.lib().mi_ew_bf16(dyh.data_ptr(), yh.data_ptr(), gm.data_ptr(), dyh.numel(), 2, L.stream_ptr()), "mi_ew_bf16 relu'")
            dyh = gm
        dyh = _pad_last(dyh, g.CoutP)
        dx = gw = None
        if ctx.needs_input_grad[0]:
            full = g.CinP == g.Cin and not (g.k == 1 and g.s == 2)
            dxh = (torch.empty if full else torch.zeros)(g.N, g.H, g.W, g.CinP, dtype=torch.bfloat16, device=xh.device)
            g.dgrad(dyh, wd, dxh)
            dx = _nchw(dxh, g.Cin)
        if ctx.needs_input_grad[1]:
            gw = g.wgrad_scaled(xh, dyh, scale)
        return dx, gw, None, None, None, None, None


class _StemFn(torch.autograd.Function):
    """trainable stem (MODEL.BACKBONE.FREEZE_AT 0, configs/coco/sparseinst/Base-SparseInst.yaml:7): forward as
    BasicStem.forward; backward = max-pool routing, ReLU mask and the weight gradient of the 16-tap conv (wgrad kernel,
    NT = 16) mapped back from the 4x4 space-to-depth layout to the 7x7 OIHW gradient.  The input image needs no gradient."""

    @staticmethod
    def forward(ctx, s2d, w7, scale, shift):
        N, Hh, Wh, _ = s2d.shape
        Cout, Cin = w7.shape[0], w7.shape[1]
        w4 = _w7_to_w4(w7 * scale.view(-1, 1, 1, 1))
        wf = torch.empty(16 * 16 * Cout, dtype=torch.bfloat16, device=s2d.device)
        L.check(L.lib().mi_pack_conv_weight(w4.contiguous().data_ptr(), Cout, 4 * Cin, 4, 4, wf.data_ptr(), 16, Cout,
                                            None, 0, 0, L.stream_ptr()), "mi_pack_conv_weight (stem)")
        a = torch.empty(N, Hh, Wh, Cout, dtype=torch.bfloat16, device=s2d.device)
        _run_conv(_conv_desc(s2d.data_ptr(), 16, N, Hh, Wh, wf, 16, a.data_ptr(), Cout, Hh, Wh, Cout, Cout, _STEM_TAPS,
                             bias=shift.float().contiguous(), flags=L.MI_CONV_RELU),
                  "mi_conv2d (7x7 stem as 4x4 over space-to-depth, ReLU in the epilogue)")
        out = torch.empty(N, (Hh + 1) // 2, (Wh + 1) // 2, Cout, dtype=torch.bfloat16, device=s2d.device)
        code = torch.empty(N, (Hh + 1) // 2, (Wh + 1) // 2, Cout, dtype=torch.uint8, device=s2d.device)
        L.check(L.lib().mi_maxpool3x3s2_fwd_idx(a.data_ptr(), Cout, out.data_ptr(), Cout, code.data_ptr(), N, Hh, Wh, Cout,
                                                L.stream_ptr()), "maxpool")
        ctx.save_for_backward(s2d, a, scale, code)
        ctx.shape = (Cout, Cin)
        return out

    @staticmethod
    def backward(ctx, g):
        s2d, a, scale, code = ctx.saved_tensors
        Cout, Cin = ctx.shape
        N, Hh, Wh, _ = s2d.shape
        g = g.contiguous()
        da = torch.empty_like(a)
        lib = L.lib()
        L.check(lib.mi_maxpool3x3s2_bwd_idx(code.data_ptr(), g.data_ptr(), Cout, da.data_ptr(), Cout, 0, N, Hh, Wh, Cout,
                                            L.stream_ptr()), "maxpool bwd")
        dy = torch.empty_like(a)
        L.check(lib.mi_ew_bf16(da.data_ptr(), a.data_ptr(), dy.data_ptr(), a.numel(), 2, L.stream_ptr()), "relu bwd")
        gw4 = torch.empty(Cout, 4 * Cin, 4, 4, dtype=torch.float32, device=a.device)
        d = L.mi_wgrad_desc()
        d.x, d.dy, d.gw = s2d.data_ptr(), dy.data_ptr(), gw4.data_ptr()
        d.ldx, d.ldy, d.N, d.H, d.W, d.outH, d.outW, d.stride = 16, Cout, N, Hh, Wh, Hh, Wh, 1
        d.Cin, d.Cout, d.CinPad, d.CoutPad, d.ntaps = 4 * Cin, Cout, 16, Cout, 16
        for t, (dy_, dx_, _) in enumerate(_STEM_TAPS):
            d.tap_dy[t], d.tap_dx[t] = dy_, dx_
        need = lib.mi_conv2d_wgrad_plan(C.byref(d))
        L.check(need, "mi_conv2d_wgrad_plan (stem)")
        ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=a.device)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
        L.check(lib.mi_conv2d_wgrad(C.byref(d), L.stream_ptr()), "mi_conv2d_wgrad (stem)")
        # [Cout][(px*2+py)*Cin + c][Ay][Ax] -> 8x8 kernel rows R = 2 Ay + py -> drop the zero row / column -> 7x7
        g8 = gw4.view(Cout, 2, 2, Cin, 4, 4).permute(0, 3, 4, 2, 5, 1).reshape(Cout, Cin, 8, 8)
        gw7 = g8[:, :, 1:, 1:] * scale.view(-1, 1, 1, 1)
        dshift = dy.float().sum((0, 1, 2)) if ctx.needs_input_grad[3] else None     # (a FrozenBN shift has none)
        return None, gw7, None, dshift


_STEM_TAPS = [(ay - 2, ax - 2, ay * 4 + ax) for ay in range(4) for ax in range(4)]


def _w7_to_w4(w7s):
    """7x7 kernel -> 8x8 with a zero first row / column -> [Cout][q = py + 2 px][c] x 4x4 taps (ay, ax in -2..1)"""
    Cout, Cin = w7s.shape[:2]
    w8 = w7s.new_zeros(Cout, Cin, 8, 8)
    w8[:, :, 1:, 1:] = w7s
    return w8.view(Cout, Cin, 4, 2, 4, 2).permute(0, 5, 3, 1, 2, 4).reshape(Cout, 4 * Cin, 4, 4)   # ch = (px*2+py)*Cin+c


class BasicStem(nn.Module):
    def __init__(self, cin=3, cout=64):
        super().__init__()
        self.conv1 = Conv2d(cin, cout, 7, stride=2, padding=3)
        self.out_channels = cout

    def forward(self, x):
        """x: fp32 NCHW image (already normalised).  7x7 s2 conv + frozen norm + ReLU + max-pool."""
        w7 = self.conv1.weight
        scale, shift = self.conv1.norm.affine()
        N, Cin, H, W = x.shape
        He, We = H + (H & 1), W + (W & 1)
        with torch.no_grad():
            if (He, We) != (H, W):       # an odd border: one more zero row / column = the conv's own zero padding
                xp = x.new_zeros(N, Cin, He, We)
                xp[:, :, :H, :W] = x
                x = xp
            x = x.float().contiguous()
            s2d = torch.empty(N, He // 2, We // 2, 16, dtype=torch.bfloat16, device=x.device)
            L.check(L.lib().mi_focus_pack(x.data_ptr(), N, He, We, s2d.data_ptr(), 16, L.stream_ptr()), "mi_focus_pack")
        if w7.requires_grad and torch.is_grad_enabled():
            return _StemFn.apply(s2d, w7, scale, shift).permute(0, 3, 1, 2)
        with torch.no_grad():       # frozen stem (FREEZE_AT >= 1): the packed image is a constant, kept across steps
            Cout = w7.shape[0]
            # (a TRAINABLE stem run without autograd is packed per call: kernels that update weights bypass torch's
            #  version counter, see Conv2d.forward)
            key = None if w7.requires_grad else (w7.data_ptr(), w7._version, self.conv1.norm.__dict__["_affine"][0])
            c = self.__dict__.get("_image")
            if key is None or c is None or c[0] != key:
                w4 = _w7_to_w4(w7 * scale.view(-1, 1, 1, 1)).contiguous()
                wf = torch.empty(16 * 16 * Cout, dtype=torch.bfloat16, device=x.device)
                L.check(L.lib().mi_pack_conv_weight(w4.data_ptr(), Cout, 4 * Cin, 4, 4, wf.data_ptr(), 16, Cout, None, 0, 0,
                                                    L.stream_ptr()), "mi_pack_conv_weight (stem)")
                c = (key, _same_address(c, wf) if key is not None else wf)
                if key is not None:
                    self.__dict__["_image"] = c
            Hh, Wh = He // 2, We // 2
            a = torch.empty(N, Hh, Wh, Cout, dtype=torch.bfloat16, device=x.device)
            _run_conv(_conv_desc(s2d.data_ptr(), 16, N, Hh, Wh, c[1], 16, a.data_ptr(), Cout, Hh, Wh, Cout, Cout, _STEM_TAPS,
                                 bias=shift, flags=L.MI_CONV_RELU), "mi_conv2d (7x7 stem as 4x4 over space-to-depth)")
            out = torch.empty(N, (Hh + 1) // 2, (Wh + 1) // 2, Cout, dtype=torch.bfloat16, device=x.device)
            L.check(L.lib().mi_maxpool3x3s2_fwd(a.data_ptr(), Cout, out.data_ptr(), Cout, N, Hh, Wh, Cout, L.stream_ptr()),
                    "maxpool fwd")
            return out.permute(0, 3, 1, 2)


def _BLOCK_FN():
    """MI_RESNET_BLOCK_FN=0: every convolution of a trainable bottleneck block is its own autograd node again (A/B switch)"""
    return _FOLDED_FN() and os.environ.get("MI_RESNET_BLOCK_FN", "1") != "0"


def _EPI_FUSE():
    """MI_RESNET_EPI_FUSE (default 1 since round 4; 0 restores the three elementwise passes per block): the block's conv3 +
    shortcut + ReLU and the two ReLU backward masks run in convolution epilogues (MI_CONV_ADDRELU / MI_CONV_RELUMASK, the tile
    kernel's EPI 2 instantiations).  First device run + same-call A/B in round 4 (profiles/r04_epi_fuse_ab.txt): DETR-R50
    bs 4 213-216 -> 218.6-219.5 images/s, SparseInst bs 8 353-371 -> 379-382; forward output identical, gradients within one
    bf16 ulp at tensor scale where the masked data gradients change kernel family
    (tests/test_gpu_resnet.py::test_bottleneck_epilogue_fusions_equal_the_elementwise_passes)."""
    return os.environ.get("MI_RESNET_EPI_FUSE", "1") == "1"


def _relu_mask(g, a):
    """g * (a > 0) (bf16, same shape)"""
    out = torch.empty_like(g)
    L.check(L.lib().mi_ew_bf16(g.data_ptr(), a.data_ptr(), out.data_ptr(), g.numel(), 2, L.stream_ptr()), "mi_ew_bf16 relu'")
    return out


# Gradients that leave a bottleneck's backward ALREADY multiplied by the ReLU mask of the tensor they belong to (the block's
# input is the previous block's ReLU output: the last data gradient into it applies (x > 0) in its epilogue, MI_CONV_ACCUM |
# MI_CONV_RELUMASK), so that the previous block's backward skips its own mask pass - a read-read-write of the block's largest
# map, 13 launches per DETR-R50 step.  Keyed by (address, autograd graph task): an entry is only believed inside the backward
# pass that made it.  Correct for any consumer: the producer of a ReLU output multiplies its out-gradient by that mask anyway,
# and the mask is idempotent; a gradient that autograd summed from several consumers is a new tensor and is masked as before.
_PREMASKED = {}


def _premask_note(t):
    task = torch._C._current_graph_task_id()
    for k in [k for k, v in _PREMASKED.items() if v != task]:
        del _PREMASKED[k]
    _PREMASKED[t.data_ptr()] = task


def _premask_take(t):
    return _PREMASKED.pop(t.data_ptr(), None) == torch._C._current_graph_task_id()


def _MASK_FUSE():
    """MI_RESNET_MASK_FUSE=0: every block masks its own out-gradient with a separate pass (round 5's form; A/B, tests)"""
    return os.environ.get("MI_RESNET_MASK_FUSE", "1") == "1"


class _BottleneckFn(torch.autograd.Function):
    """A trainable BottleneckBlock (detectron2 resnet.py BottleneckBlock: conv1 1x1 - conv2 3x3 - conv3 1x1, FrozenBN folded,
    + shortcut, ReLU) as ONE autograd node.  What the per-convolution nodes cannot do: the block input's gradient is the
    sum of two paths - autograd materialised both and added them (a three-tensor pass over the block's largest map, and a
    full zero fill under the stride-2 shortcut's data gradient, which writes only the even pixels); here the second path
    ACCUMULATES into the first in the convolution's epilogue (MI_CONV_ACCUM): identity blocks add conv1's data gradient
    into the (masked) output gradient, shortcut blocks add the shortcut's data gradient into conv1's.  Same kernels, same
    roundings as the separate nodes (bf16 + bf16 -> bf16)."""

    @staticmethod
    def forward(ctx, x, s2, ssc, in_relu, *rest):
        n = 4 if ssc else 3
        ws, aff = rest[:n], rest[n:]
        ctx.in_relu = bool(in_relu)
        scales, shifts = aff[0::2], aff[1::2]
        need_dx = x.requires_grad
        N, Cin, H, W = x.shape
        bc = ws[0].shape[0]
        g1 = _ConvGeom((N, Cin, H, W), ws[0].shape, 1, 0)
        g2 = _ConvGeom((N, bc, H, W), ws[1].shape, s2, 1)
        g3 = _ConvGeom((N, bc, g2.Ho, g2.Wo), ws[2].shape, 1, 0)
        gs = _ConvGeom((N, Cin, H, W), ws[3].shape, ssc, 0) if ssc else None
        geoms = [g1, g2, g3] + ([gs] if ssc else [])
        for g in geoms:
            if g.CinP != g.Cin or g.CoutP != g.Cout:
                raise L.MI355Error("ResNet bottleneck: channel counts must be multiples of 32")
        imgs = [g.pack(w, dgrad=(need_dx or i in (1, 2)), scale=sc)
                for i, (g, w, sc) in enumerate(zip(geoms, ws, scales))]
        dev = x.device
        xh = _nhwc(x)

        def run(g, inp, i, relu, add_relu=None):
            y = torch.empty(g.N, g.Ho, g.Wo, g.CoutP, dtype=torch.bfloat16, device=dev)
            g.fwd(inp, imgs[i][0], y, bias=shifts[i], relu=relu, add_relu=add_relu)
            return y
        a1 = run(g1, xh, 0, True)
        a2 = run(g2, a1, 1, True)
        sc = run(gs, xh, 3, False) if ssc else xh
        ctx.epi = _EPI_FUSE()
        if ctx.epi:
            y = run(g3, a2, 2, False, add_relu=sc)            # relu(bf16(conv3 + shift) + shortcut) in conv3's epilogue
        else:
            o = run(g3, a2, 2, False)
            y = torch.empty_like(o)
            L.check(L.lib().mi_ew_bf16(o.data_ptr(), sc.data_ptr(), y.data_ptr(), o.numel(), 7, L.stream_ptr()), "mi_ew_bf16 add+relu")
        ctx.geoms, ctx.has_sc = geoms, bool(ssc)
        ctx.params = ws
        ctx.save_for_backward(xh, a1, a2, y, *[im[1] for im in imgs], *scales)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        n = 4 if ctx.has_sc else 3
        xh, a1, a2, y = ctx.saved_tensors[:4]
        wds, scales = ctx.saved_tensors[4:4 + n], ctx.saved_tensors[4 + n:]
        g1, g2, g3 = ctx.geoms[:3]
        gs = ctx.geoms[3] if ctx.has_sc else None
        dev = xh.device
        gyh = _nhwc(gy)
        if _premask_take(gyh):
            gm = gyh                        # the next block's backward masked it in its last epilogue (and owns no other use of it)
        else:
            gm = torch.empty_like(gyh)      # (a fresh tensor: identity blocks accumulate the input gradient into it)
            L.check(L.lib().mi_ew_bf16(gyh.data_ptr(), y.data_ptr(), gm.data_ptr(), gm.numel(), 2, L.stream_ptr()), "mi_ew_bf16 relu'")
        gws = [None] * n
        # the block's three or four weight gradients as ONE grouped launch (ops.WgradBatch), issued below - before the input
        # gradient is accumulated INTO gm, which two of the jobs read
        df = wgrad_can_defer(*ctx.params)
        gws[2] = g3.wgrad_scaled(a2, gm, scales[2], defer=df, owner=ctx.params[2] if df else None)
        da2 = torch.empty_like(a2)
        if ctx.epi:
            g3.dgrad(gm, wds[2], da2, relu_mask=a2)
        else:
            g3.dgrad(gm, wds[2], da2)
            da2 = _relu_mask(da2, a2)
        gws[1] = g2.wgrad_scaled(a1, da2, scales[1], defer=df, owner=ctx.params[1] if df else None)
        da1 = torch.empty_like(a1)
        if ctx.epi:
            g2.dgrad(da2, wds[1], da1, relu_mask=a1)
        else:
            g2.dgrad(da2, wds[1], da1)
            da1 = _relu_mask(da1, a1)
        gws[0] = g1.wgrad_scaled(xh, da1, scales[0], defer=df, owner=ctx.params[0] if df else None)
        if gs is not None:
            gws[3] = gs.wgrad_scaled(xh, gm, scales[3], defer=df, owner=ctx.params[3] if df else None)
        if df:
            WgradBatch.flush()
        dx = None
        if ctx.needs_input_grad[0]:
            # the block input is a ReLU output (ResNet sets in_relu): its mask goes into the LAST data gradient's epilogue
            # (not for the stride-2 shortcut, whose accumulate visits only the even pixels)
            fuse = ctx.in_relu and ctx.epi and _MASK_FUSE() and (gs is None or gs.s == 1)
            if gs is None:
                dxh = gm
                g1.dgrad(da1, wds[0], dxh, accum=True, relu_mask=xh if fuse else None)
            else:
                dxh = torch.empty_like(xh)
                g1.dgrad(da1, wds[0], dxh)
                gs.dgrad(gm, wds[3], dxh, accum=True, relu_mask=xh if fuse else None)
            if fuse:
                _premask_note(dxh)
            dx = dxh.permute(0, 3, 1, 2)
        return (dx, None, None, None, *gws, *([None] * (2 * n)))


class BottleneckBlock(nn.Module):
    def __init__(self, cin, cout, bottleneck_channels, stride=1, stride_in_1x1=False):
        super().__init__()
        self.shortcut = Conv2d(cin, cout, 1, stride=stride) if cin != cout else None
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(cin, bottleneck_channels, 1, stride=s1)
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, 3, stride=s3, padding=1)
        self.conv3 = Conv2d(bottleneck_channels, cout, 1)
        self.stride = stride

    def forward(self, x):
        convs = [self.conv1, self.conv2, self.conv3] + ([self.shortcut] if self.shortcut is not None else [])
        if (_BLOCK_FN() and torch.is_grad_enabled() and all(c.weight.requires_grad for c in convs) and self.conv1.stride == 1):
            aff = [t for c in convs for t in c.norm.affine()]
            return _BottleneckFn.apply(x, self.conv2.stride, self.shortcut.stride if self.shortcut is not None else 0,
                                       getattr(self, "input_is_relu", False), *[c.weight for c in convs], *aff)
        if _conv_relu_fused():
            out = self.conv2(self.conv1(x, relu=True), relu=True)
        else:
            out = _relu(self.conv2(_relu(self.conv1(x))))
        sc = self.shortcut(x) if self.shortcut is not None else x
        no_grad_path = not (torch.is_grad_enabled() and (x.requires_grad or any(c.weight.requires_grad for c in convs)))
        if no_grad_path and _EPI_FUSE() and _FOLDED_FN() and self.conv3.weight.shape[0] % 32 == 0:
            # a frozen block (FREEZE_AT prefix) / inference: the residual add + ReLU in conv3's epilogue, as _BottleneckFn
            # does for the trainable blocks - the separate pass was a read-read-write of the block's largest map
            # (3 x 73 us per DETR-R50 step for res2 at 800 x 1333)
            return self.conv3(out, add_relu=sc)
        out = self.conv3(out)
        return _add_relu(out, sc)


class ResNet(Backbone):
    def __init__(self, depth=50, out_features=("res5",), freeze_at=2, stride_in_1x1=False):
        super().__init__()
        blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}[depth]
        self.stem = BasicStem(3, 64)
        self._out_features = list(out_features)
        self._shapes = {"stem": ShapeSpec(channels=64, stride=4)}
        cin, bc, cout, stride = 64, 64, 256, 4
        self.stage_names = []
        for i, nb in enumerate(blocks):
            name = f"res{i + 2}"
            first = 1 if i == 0 else 2
            stage = nn.Sequential(*[BottleneckBlock(cin if k == 0 else cout, cout, bc, stride=first if k == 0 else 1,
                                                    stride_in_1x1=stride_in_1x1) for k in range(nb)])
            for blk in stage:
                blk.input_is_relu = True    # every block's input is a ReLU output (the stem's max-pooled ReLU, the previous block)
            self.add_module(name, stage)
            self.stage_names.append(name)
            stride *= first
            self._shapes[name] = ShapeSpec(channels=cout, stride=stride)
            cin, bc, cout = cout, bc * 2, cout * 2
        self.freeze(freeze_at)

    def freeze(self, freeze_at):
        """detectron2 ResNet.freeze: 1 = stem, 2 = stem + res2, ..."""
        if freeze_at >= 1:
            for p in self.stem.parameters():
                p.requires_grad = False
        for idx, name in enumerate(self.stage_names, start=2):
            if freeze_at >= idx:
                for p in getattr(self, name).parameters():
                    p.requires_grad = False
        return self

    def forward(self, x):
        if not x.is_cuda:
            raise L.MI355Error("ResNet: the MI355X path needs device tensors (no CPU fallback)")
        outs = {}
        x = self.stem(x)
        if "stem" in self._out_features:
            outs["stem"] = x
        for name in self.stage_names:
            x = getattr(self, name)(x)
            if name in self._out_features:
                outs[name] = x
        return outs

    def output_shape(self):
        return {k: self._shapes[k] for k in self._out_features}

    def stage_modules(self):
        """res2 .. res5 in forward order: single-tensor outputs that cut the backward into stages
        (graph_step.GraphedTrainStep's data-parallel overlap)"""
        return [getattr(self, n) for n in self.stage_names]

    @property
    def size_divisibility(self):
        return 0


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape=None):
    r = cfg.MODEL.RESNETS
    if r.NORM != "FrozenBN":
        raise NotImplementedError("build_resnet_backbone: MODEL.RESNETS.NORM FrozenBN (the reference's DETR / SparseInst configs)")
    return ResNet(depth=r.DEPTH, out_features=r.OUT_FEATURES, freeze_at=cfg.MODEL.BACKBONE.FREEZE_AT,
                  stride_in_1x1=r.STRIDE_IN_1X1)
