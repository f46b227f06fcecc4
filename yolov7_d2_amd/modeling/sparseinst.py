"""SparseInst (BASELINE.json config 5) on the MI355X kernels - drop-ins, with the reference's names, constructors, config
keys and state_dict keys, for

    SparseInst (META_ARCH)                         yolov7/modeling/meta_arch/sparseinst.py:54-234 (+ rescoring_mask :24-27)
    InstanceContextEncoder, PyramidPoolingModule,
    MyAdaptiveAvgPool2d                            yolov7/modeling/transcoders/encoder_sparseinst.py:18-127
    InstanceBranch, GroupInstanceBranch, MaskBranch,
    BaseIAMDecoder, GroupIAMDecoder                yolov7/modeling/transcoders/decoder_sparseinst.py:18-255
    SparseInstCriterion, SparseInstMatcher,
    dice_score, dice_loss, compute_mask_iou        yolov7/modeling/loss/sparseinst_loss.py:19-354

How it maps to the kernels (activations are bf16 NCHW tensors in channels_last memory = NHWC):
  * every convolution (FPN laterals / outputs, PPM, the two 4 x (3x3 + ReLU) stacks over the 258-channel coordinate-augmented
    map, the grouped IAM conv as its 4 independent groups, projection, fusion)      -> torch.ops.mi355.conv2d
  * IAM aggregation  inst[b, n, c] = sum_p sigmoid(iam)[b, n, p] * feat[b, c, p]   -> the weight-gradient MFMA kernel
    (a "sum over pixels of an outer product" IS a conv wgrad: dy = iam probabilities, x = features), backward through the
    1x1 conv kernel
  * dynamic mask head  mask[b, n, p] = sum_c kernel[b, n, c] * feat[b, c, p]        -> the 1x1 conv kernel with the
    per-image predicted kernels as weights (forward, data gradient, and the weight gradient = d kernel)
  * the matcher's dice-score matmul [B*N, HW] x [HW, M]                            -> the same pixel-sum MFMA kernel;
    scipy's linear_sum_assignment(maximize=True)                                   -> mi_lsap on the negated scores
  * bilinear resizes, nearest x2, sigmoid / ReLU / add                              -> mi_bilinear_resize_*, mi_upsample2x_*,
    mi_ew_bf16
  * mask BCE + dice + thresholded mask IoU of the matched pairs, and their gradient  -> mi_sparseinst_mask_stats / _grad
Small tensors (the [B, 100, 80] class logits' focal loss, the PPM's 1..6-pixel average pools, normalisers) stay torch
device ops.  No CPU path: device tensors only.
"""
import ctypes as C
import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib as L
from ..d2shim import META_ARCH_REGISTRY, ImageList, Instances, build_backbone
from ..ops import ConvPaddedFn, HostRing, WgradBatch, _ConvGeom, _conv_desc, _ld, _nhwc_v, _run_conv, nhwc_strided_ok, wgrad_can_defer
from ..ops import feed_batch_enabled, mask_targets_batch, normalize_pad_batch, pack_images
from .transformer import _LinearFn, _conv1x1, _factor

# ------------------------------------------------------------------------------------------------ small op wrappers


def _nhwc(x):
    return x.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()


class _Ew1(torch.autograd.Function):
    """unary elementwise on a contiguous bf16 tensor: kind 'relu' (ops 1 / 2) or 'sigmoid' (ops 3 / 4)"""

    @staticmethod
    def forward(ctx, a, kind):
        a = a.contiguous()
        y = torch.empty_like(a)
        L.check(L.lib().mi_ew_bf16(a.data_ptr(), None, y.data_ptr(), a.numel(), 1 if kind == "relu" else 3,
                                   L.stream_ptr()), "mi_ew_bf16")
        ctx.save_for_backward(y)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(g)
        L.check(L.lib().mi_ew_bf16(g.data_ptr(), y.data_ptr(), out.data_ptr(), g.numel(), 2 if ctx.kind == "relu" else 4,
                                   L.stream_ptr()), "mi_ew_bf16 (backward)")
        return out, None


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        L.check(L.lib().mi_ew_bf16(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), 0, L.stream_ptr()), "mi_ew_bf16 add")
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


def relu(x):
    """NCHW (channels_last) bf16 -> same"""
    return _Ew1.apply(_nhwc(x), "relu").permute(0, 3, 1, 2)


class _Resize(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=False) on an NHWC bf16 image"""

    @staticmethod
    def forward(ctx, xh, Ho, Wo):
        N, H, W, Cc = xh.shape
        assert Cc % 8 == 0
        y = torch.empty(N, Ho, Wo, Cc, dtype=torch.bfloat16, device=xh.device)
        L.check(L.lib().mi_bilinear_resize_bf16(xh.data_ptr(), Cc, N, H, W, Cc, y.data_ptr(), Cc, Ho, Wo, L.stream_ptr()),
                "mi_bilinear_resize_bf16")
        ctx.shape = (N, H, W, Cc, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, g):
        N, H, W, Cc, Ho, Wo = ctx.shape
        if not nhwc_strided_ok(g):       # (a channel slice of the fusion cat's gradient is read in place: pixel stride _ld(g))
            g = g.contiguous()
        dx = torch.empty(N, H, W, Cc, dtype=torch.bfloat16, device=g.device)
        L.check(L.lib().mi_bilinear_resize_bwd_bf16(g.data_ptr(), _ld(g), N, H, W, Cc, dx.data_ptr(), Cc, Ho, Wo, None,
                                                    L.stream_ptr()), "mi_bilinear_resize_bwd_bf16")
        return dx, None, None


def resize_bilinear(x, size):
    """NCHW (channels_last) -> NCHW (channels_last), size = (Ho, Wo)"""
    if tuple(x.shape[-2:]) == tuple(size):
        return x
    return _Resize.apply(_nhwc(x), int(size[0]), int(size[1])).permute(0, 3, 1, 2)


class _Up2(torch.autograd.Function):
    """F.interpolate(scale_factor=2, mode='nearest') on NHWC bf16"""

    @staticmethod
    def forward(ctx, xh):
        N, H, W, Cc = xh.shape
        y = torch.empty(N, 2 * H, 2 * W, Cc, dtype=torch.bfloat16, device=xh.device)
        L.check(L.lib().mi_upsample2x_fwd(xh.data_ptr(), Cc, y.data_ptr(), Cc, N, H, W, Cc, L.stream_ptr()), "mi_upsample2x_fwd")
        ctx.shape = (N, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, g):
        N, H, W, Cc = ctx.shape
        if not nhwc_strided_ok(g):
            g = g.contiguous()
        dx = torch.empty(N, H, W, Cc, dtype=torch.bfloat16, device=g.device)
        L.check(L.lib().mi_upsample2x_bwd(g.data_ptr(), _ld(g), dx.data_ptr(), Cc, 0, N, H, W, Cc, L.stream_ptr()), "mi_upsample2x_bwd")
        return dx


def _rup(a, b):
    return (a + b - 1) // b * b


def _pad_cols(t, Cp):
    if t.shape[-1] == Cp:
        return t.contiguous()
    out = torch.zeros(*t.shape[:-1], Cp, dtype=t.dtype, device=t.device)
    out[..., : t.shape[-1]] = t
    return out


def pixel_outer(a, b):
    """out[i, j] = sum_p a[p, i] * b[p, j]  (a: bf16 [P, Ca], b: bf16 [P, Cb], Ca / Cb multiples of 32) -> fp32 [Ca, Cb]:
    the conv weight-gradient kernel with dy = a, x = b over the P 'pixels' (split-K over pixel tiles, MFMA)"""
    P, Ca = a.shape
    Cb = b.shape[1]
    assert Ca % 32 == 0 and Cb % 32 == 0 and a.is_contiguous() and b.is_contiguous()
    H, W = _factor(P)
    out = torch.empty(Ca, Cb, dtype=torch.float32, device=a.device)
    d = L.mi_wgrad_desc()
    d.x, d.dy, d.gw = b.data_ptr(), a.data_ptr(), out.data_ptr()
    d.ldx, d.ldy, d.N, d.H, d.W, d.outH, d.outW, d.stride = Cb, Ca, 1, H, W, H, W, 1
    d.Cin, d.Cout, d.CinPad, d.CoutPad, d.ntaps = Cb, Ca, Cb, Ca, 1
    need = L.lib().mi_conv2d_wgrad_plan(C.byref(d))
    L.check(need, "mi_conv2d_wgrad_plan (pixel_outer)")
    ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=a.device)
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
    L.check(L.lib().mi_conv2d_wgrad(C.byref(d), L.stream_ptr()), "mi_conv2d_wgrad (pixel_outer)")
    return out


def pixel_outer_batch(a, b):
    """pixel_outer for every image of a batch as ONE grouped launch (+ one reduce grid) instead of B pairs of launches:
    a bf16 [B, P, Ca], b bf16 [B, P, Cb] (dense) -> fp32 [B, Ca, Cb].  MI_SI_OUTER_BATCH=0: the per-image launches."""
    import os
    B, P, Ca = a.shape
    Cb = b.shape[2]
    assert Ca % 32 == 0 and Cb % 32 == 0 and a.is_contiguous() and b.is_contiguous() and b.shape[:2] == a.shape[:2]
    if os.environ.get("MI_SI_OUTER_BATCH", "1") == "0" or not WgradBatch.enabled():
        return torch.stack([pixel_outer(a[i], b[i]) for i in range(B)])
    H, W = _factor(P)
    out = torch.empty(B, Ca, Cb, dtype=torch.float32, device=a.device)
    jobs = []
    for i in range(B):
        d = L.mi_wgrad_desc()
        d.x, d.dy, d.gw = b[i].data_ptr(), a[i].data_ptr(), out[i].data_ptr()
        d.ldx, d.ldy, d.N, d.H, d.W, d.outH, d.outW, d.stride = Cb, Ca, 1, H, W, H, W, 1
        d.Cin, d.Cout, d.CinPad, d.CoutPad, d.ntaps = Cb, Ca, Cb, Ca, 1
        jobs.append((d, (a, b, out)))
    WgradBatch.run_now(jobs)
    return out


class _PixelOuterBatchFn(torch.autograd.Function):
    """differentiable pixel_outer_batch; backward per image as _PixelOuterFn's (two 1x1-conv launches with the image's own
    gradient matrix as the weight)"""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return pixel_outer_batch(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B, P, Ca = a.shape
        Cb = b.shape[2]
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        g32 = g.float().contiguous()
        gt = g32.transpose(1, 2).contiguous() if db is not None else None
        for i in range(B):        # the 1x1-conv kernel writes each image's rows of the batch gradient in place
            if da is not None:
                wf, _ = pack_images(g32[i], Ca, Cb, 1, 1, Cb, Ca, Ca, Cb, dgrad=False)
                _conv1x1(b[i], wf, da[i], P, Cb, Ca, Ca)
            if db is not None:
                wf, _ = pack_images(gt[i], Cb, Ca, 1, 1, Ca, Cb, Cb, Ca, dgrad=False)
                _conv1x1(a[i], wf, db[i], P, Ca, Cb, Cb)
        return da, db


class _PixelOuterFn(torch.autograd.Function):
    """differentiable pixel_outer: d a[p, i] = sum_j g[i, j] b[p, j], d b[p, j] = sum_i a[p, i] g[i, j] (1x1 conv kernel)"""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return pixel_outer(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        with torch.no_grad():
            da = _LinearFn.apply(b, g.float().contiguous(), None)                 # [P, Ca]
            db = _LinearFn.apply(a, g.float().t().contiguous(), None)             # [P, Cb]
        return da.contiguous(), db.contiguous()


def _conv(x, m, stride=1, relu=False):
    """nn.Conv2d / detectron2 Conv2d (bias, no norm) through the implicit-GEMM op; relu=True applies the ReLU that follows
    in the convolution's epilogue (MI_CONV_RELU) instead of as a second pass over the map"""
    if isinstance(x, PaddedNHWC):        # the kernels' operand built once by the caller (ops.ConvPaddedFn)
        return ConvPaddedFn.apply(x.xh, m.weight, m.bias, x.cin, stride, m.padding[0], relu)
    op = torch.ops.mi355.conv2d_relu if relu else torch.ops.mi355.conv2d
    return op(x, m.weight, m.bias, stride, m.padding[0])


class PaddedNHWC:
    """a feature map as bf16 [N, H, W, CinP] with `cin` real channels and zero pad channels (what the implicit-GEMM kernels
    read); `_conv` takes it in place of an NCHW tensor"""

    def __init__(self, xh, cin):
        self.xh, self.cin = xh, cin

    @property
    def is_cuda(self):
        return self.xh.is_cuda


class _CoordCat(torch.autograd.Function):
    """decoder_sparseinst.py:117-131: torch.cat([coordinates(x), x], 1) - written ONCE as the padded NHWC operand of the two
    branches' first convolutions (PaddedNHWC): a zero fill + two strided copies, where every one of those convolutions
    (two branches, forward and backward) made its own layout copy + zero fill + strided copy of the 258-channel cat"""

    @staticmethod
    def forward(ctx, x, coords):
        N, Cc, H, W = x.shape
        nco = coords.shape[-1]
        CP = _rup(Cc + nco, 32)
        xh = torch.zeros(N, H, W, CP, dtype=torch.bfloat16, device=x.device)
        xh[..., :nco] = coords
        xh[..., nco: nco + Cc] = x.permute(0, 2, 3, 1)
        ctx.sl = (nco, Cc)
        return xh

    @staticmethod
    def backward(ctx, g):
        nco, Cc = ctx.sl
        return g[..., nco: nco + Cc].permute(0, 3, 1, 2), None


def _linear(x, lin):
    shp = x.shape
    y = _LinearFn.apply(x.to(torch.bfloat16).reshape(-1, shp[-1]).contiguous(), lin.weight, lin.bias)
    return y.reshape(*shp[:-1], lin.out_features)


# ------------------------------------------------------------------------------------------------ encoder
class MyAdaptiveAvgPool2d(nn.Module):
    """encoder_sparseinst.py:18-40: avg_pool2d with kernel = ceil(size / sz) (NOT nn.AdaptiveAvgPool2d)"""

    def __init__(self, sz=None):
        super().__init__()
        self.sz = sz

    def forward(self, x):
        kh, kw = x.shape[2], x.shape[3]
        if self.sz is not None:
            sz = (self.sz, self.sz) if isinstance(self.sz, int) else self.sz
            kh, kw = math.ceil(x.shape[2] / sz[0]), math.ceil(x.shape[3] / sz[1])
        xf = x.float()
        # torch's avg_pool2d gives every OUTPUT element one thread that walks its window alone: the 20 x 20 window of the
        # 1-bin stage was a 2 048-thread launch of 400 serial loads (105 us; the 10 x 10 windows 23 us each).  A window that
        # an f x f grid tiles exactly is the mean of its tile means: pool by f first (wide launch), then the rest
        f = next((f for f in (5, 4, 3, 2) if kh * kw >= 64 and kh % f == 0 and kw % f == 0), None)
        if f is not None:
            xf, kh, kw = F.avg_pool2d(xf, kernel_size=(f, f)), kh // f, kw // f
        return F.avg_pool2d(xf, kernel_size=(kh, kw), ceil_mode=False).to(x.dtype)      # 1..36 output pixels


class _PyramidPoolFn(torch.autograd.Function):
    """the pooling stages of PyramidPoolingModule (MyAdaptiveAvgPool2d per stage, encoder_sparseinst.py:18-40) on one map as
    ONE launch, their backward as one launch (mi_pyramid_pool_fwd / _bwd): x NCHW-shaped -> one NCHW-shaped (channels_last
    memory) bf16 map per stage.  As torch calls: float() + two avg_pool2d + to(bf16) per stage, and the mirror image + the
    accumulation of the four input gradients backward - ~35 launches on a [8, 256, 20, 20] map."""

    @staticmethod
    def forward(ctx, x, kernels):
        xh = _nhwc_v(x)
        N, H, W, Cc = xh.shape
        ns = len(kernels)
        ys = [torch.empty(N, H // kh, W // kw, Cc, dtype=torch.bfloat16, device=x.device) for kh, kw in kernels]
        khs, kws = (C.c_int * ns)(*[k[0] for k in kernels]), (C.c_int * ns)(*[k[1] for k in kernels])
        ptrs = (C.c_void_p * ns)(*[y.data_ptr() for y in ys])
        L.check(L.lib().mi_pyramid_pool_fwd(xh.data_ptr(), _ld(xh), N, H, W, Cc, ns, khs, kws, ptrs, L.stream_ptr()), "mi_pyramid_pool_fwd")
        ctx.kernels, ctx.shape, ctx.dtype = kernels, (N, H, W, Cc), x.dtype
        return tuple(y.permute(0, 3, 1, 2) for y in ys)

    @staticmethod
    def backward(ctx, *gs):
        N, H, W, Cc = ctx.shape
        ns = len(ctx.kernels)
        ghs = [None if g is None else _nhwc(g) for g in gs]
        dx = torch.empty(N, H, W, Cc, dtype=torch.bfloat16, device=next(g for g in ghs if g is not None).device)
        khs, kws = (C.c_int * ns)(*[k[0] for k in ctx.kernels]), (C.c_int * ns)(*[k[1] for k in ctx.kernels])
        ptrs = (C.c_void_p * ns)(*[None if g is None else g.data_ptr() for g in ghs])
        L.check(L.lib().mi_pyramid_pool_bwd(ptrs, N, H, W, Cc, ns, khs, kws, dx.data_ptr(), Cc, L.stream_ptr()), "mi_pyramid_pool_bwd")
        return dx.permute(0, 3, 1, 2).to(ctx.dtype), None


def _PPM_FUSED():
    """MI_SI_PPM_FUSED=0: the pooling stages as torch calls (round 5's form; A/B, tests)"""
    import os
    return os.environ.get("MI_SI_PPM_FUSED", "1") != "0"


class PyramidPoolingModule(nn.Module):
    def __init__(self, in_channels, channels=512, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.stages = nn.ModuleList([nn.Sequential(MyAdaptiveAvgPool2d((s, s)), nn.Conv2d(in_channels, channels, 1))
                                     for s in sizes])
        self.bottleneck = nn.Conv2d(in_channels + len(sizes) * channels, in_channels, 1)

    def forward(self, feats):
        h, w = feats.shape[2], feats.shape[3]
        if _PPM_FUSED() and feats.is_cuda and feats.shape[1] % 8 == 0 and len(self.stages) <= 8:
            kernels = []
            for st in self.stages:
                sz = st[0].sz
                sz = (sz, sz) if isinstance(sz, int) else sz
                kernels.append((h, w) if sz is None else (math.ceil(h / sz[0]), math.ceil(w / sz[1])))
            pooled = _PyramidPoolFn.apply(feats.to(torch.bfloat16), tuple(kernels))
        else:
            pooled = [st[0](feats) for st in self.stages]
        priors = [resize_bilinear(_conv(pl, st[1], relu=True), (h, w)) for pl, st in zip(pooled, self.stages)] + [feats]
        return _conv(torch.cat(priors, 1), self.bottleneck, relu=True)


class InstanceContextEncoder(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.num_channels = cfg.MODEL.SPARSE_INST.ENCODER.NUM_CHANNELS
        self.in_features = cfg.MODEL.SPARSE_INST.ENCODER.IN_FEATURES
        self.in_channels = [input_shape[f].channels for f in self.in_features]
        lat, outc = [], []
        for cin in reversed(self.in_channels):
            l_ = nn.Conv2d(cin, self.num_channels, 1)
            o_ = nn.Conv2d(self.num_channels, self.num_channels, 3, padding=1)
            for m in (l_, o_):      # c2_xavier_fill
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)
            lat.append(l_); outc.append(o_)
        self.fpn_laterals, self.fpn_outputs = nn.ModuleList(lat), nn.ModuleList(outc)
        self.ppm = PyramidPoolingModule(self.num_channels, self.num_channels // 4)
        self.fusion = nn.Conv2d(self.num_channels * 3, self.num_channels, 1)
        nn.init.kaiming_normal_(self.fusion.weight, mode="fan_out", nonlinearity="relu")       # c2_msra_fill
        nn.init.constant_(self.fusion.bias, 0)

    def forward(self, features):
        feats = [features[f] for f in self.in_features][::-1]
        prev = self.ppm(_conv(feats[0], self.fpn_laterals[0]))
        outputs = [_conv(prev, self.fpn_outputs[0])]
        for feature, lat_conv, out_conv in zip(feats[1:], self.fpn_laterals[1:], self.fpn_outputs[1:]):
            lat = _conv(feature, lat_conv)
            top = _Up2.apply(_nhwc(prev))
            prev = _Add.apply(_nhwc(lat), top).permute(0, 3, 1, 2)
            outputs.insert(0, _conv(prev, out_conv))
        size = outputs[0].shape[2:]
        fused = [outputs[0]] + [resize_bilinear(x, size) for x in outputs[1:]]
        return _conv(torch.cat(fused, dim=1), self.fusion)


# ------------------------------------------------------------------------------------------------ decoder
def _make_stack_3x3_convs(num_convs, in_channels, out_channels):
    convs = []
    for _ in range(num_convs):
        c = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        nn.init.kaiming_normal_(c.weight, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(c.bias, 0)
        convs += [c, nn.ReLU(True)]
        in_channels = out_channels
    return nn.Sequential(*convs)


def _run_stack(seq, x):
    mods, i = list(seq), 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ReLU):
            x = relu(x)
        else:
            act = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)      # conv + ReLU pair: one launch
            x = _conv(x, m, relu=act)
            i += act
        i += 1
    return x


def _colsums(x, square=False):
    """x: bf16 [B, P, C] contiguous (C % 8 == 0) -> fp32 [B, C]: sum over the P pixels of every image of x (or x^2).  The
    library's column-sum kernel, one launch per image - NOT `x.sum(1, dtype=float32)`: a torch reduction of this shape
    (thousands of rows per output, few outputs) splits each output across blocks, and that multi-block form returned
    wrong sums on every replay after the first inside a captured hipGraph on this stack (PyTorch 2.10 / ROCm 7.x;
    tools/si_graph_debug3.py localised it: norm 1635.7 eager / first replay, 1489.0 on every later replay, all inputs
    identical).  The captured SparseInst step therefore keeps torch reductions to shapes that stay single-block."""
    B, P, Cc = x.shape
    assert x.is_contiguous() and Cc % 8 == 0 and x.dtype == torch.bfloat16
    out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    ws = torch.empty(128 * Cc, dtype=torch.float32, device=x.device)
    fn = L.lib().mi_colsumsq_bf16_wide if square else L.lib().mi_colsum_bf16_wide
    for b in range(B):
        L.check(fn(x[b].data_ptr(), Cc, P, Cc, out[b].data_ptr(), 0, ws.data_ptr(), L.stream_ptr()), "mi_colsum_bf16_wide")
    return out


def _aggregate(iam, features):
    """iam [B, N, H, W] logits, features [B, C, H, W] -> inst [B, N, C] = (sigmoid(iam) @ features^T) / normaliser
    (decoder_sparseinst.py:62-74 / 217-228)"""
    B, N, H, W = iam.shape
    Cc = features.shape[1]
    Np = _rup(N, 32)
    prob = _Ew1.apply(_nhwc(iam), "sigmoid")                          # [B, H, W, N]
    fh = _nhwc(features)
    outs = []
    for b in range(B):
        a = _pad_cols(prob[b].reshape(H * W, N), Np)
        outs.append(_PixelOuterFn.apply(a, fh[b].reshape(H * W, Cc))[:N])
    inst = torch.stack(outs)                                           # fp32 [B, N, C]
    return inst, _ColSumFn.apply(prob.reshape(B, H * W, N))           # normaliser [B, N] (fp32 accumulation, no fp32 copy of the map)


IAM_GS = 128      # channel stride of one group in the padded IAM map: the kernels' output-channel tile (>= masks per group)


class _GroupIamFn(torch.autograd.Function):
    """GroupInstanceBranch's grouped 3x3 `iam_conv` (decoder_sparseinst.py:212-242: nn.Conv2d(dim, masks * G, 3, groups=G)) as
    its G convolutions over channel slices of the feature map, each WRITING ITS SLICE of one bf16 NHWC map [B, H, W, G * 128]
    (group g's `cout` masks at channels g * 128 .., the rest zero).  Round 5 ran G ops on NCHW slices and torch.cat'ed the
    results (41 MB), `_aggregate` then made a zero-padded per-image copy of the sigmoid of that map (8 x (5.3 MB fill +
    strided copy)), and the backward padded every out-gradient slice again (ops._pad_last: 4 x 52 MB): here the map IS the
    operand of everything behind it, forward and backward, and nothing is copied.  Backward: the data gradients write their
    64-channel slices of one feature gradient, the G weight gradients leave as ONE grouped launch (ops.WgradBatch), the bias
    gradient is one column-sum launch over the whole map."""

    @staticmethod
    def forward(ctx, features, weight, bias, G):
        fh = _nhwc_v(features)                                  # [B, H, W, C] (a view when the map is channels_last)
        B, H, W, Cc = fh.shape
        cin, cout = Cc // G, weight.shape[0] // G
        if cin % 32 or cout > IAM_GS or cout % 4 or tuple(weight.shape[1:]) != (cin, 3, 3):
            raise L.MI355Error(f"grouped IAM conv: {tuple(weight.shape)} over {Cc} channels in {G} groups is not served")
        geo = _ConvGeom((B, cin, H, W), (cout, cin, 3, 3), 1, 1)
        assert geo.CoutP == IAM_GS and geo.CinP == cin
        out = torch.empty(B, H, W, G * IAM_GS, dtype=torch.bfloat16, device=fh.device)
        if cout < IAM_GS:
            out.view(B, H, W, G, IAM_GS)[..., cout:].zero_()
        w32 = weight.detach().float().contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        taps = [(r - 1, s_ - 1, r * 3 + s_) for r in range(3) for s_ in range(3)]
        ldx = _ld(fh)
        wds = []
        for g in range(G):
            wf, wd = geo.pack(w32[g * cout:(g + 1) * cout])
            wds.append(wd)
            _run_conv(_conv_desc(fh.data_ptr() + g * cin * 2, ldx, B, H, W, wf, cin, out.data_ptr() + g * IAM_GS * 2, G * IAM_GS,
                                 H, W, cout, IAM_GS, taps, bias=None if b32 is None else b32[g * cout:(g + 1) * cout]),
                      "mi_conv2d (grouped IAM conv)")
        ctx.geo, ctx.G, ctx.dims, ctx.has_bias = geo, G, (B, H, W, Cc, cin, cout), bias is not None
        ctx.params = (weight, bias)
        ctx.save_for_backward(fh, *wds)
        return out

    @staticmethod
    def backward(ctx, gy):
        fh, *wds = ctx.saved_tensors
        geo, G = ctx.geo, ctx.G
        B, H, W, Cc, cin, cout = ctx.dims
        gy = gy.contiguous()                                     # [B, H, W, G * 128]; its pad channels are zero (see _aggregate)
        dev = gy.device
        taps = [(1 - r, 1 - s_, r * 3 + s_) for r in range(3) for s_ in range(3)]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, H, W, Cc, dtype=torch.bfloat16, device=dev)
            for g in range(G):
                _run_conv(_conv_desc(gy.data_ptr() + g * IAM_GS * 2, G * IAM_GS, B, H, W, wds[g], IAM_GS, dx.data_ptr() + g * cin * 2,
                                     Cc, H, W, cin, cin, taps), "mi_conv2d (grouped IAM dgrad)")
            dx = dx.permute(0, 3, 1, 2)
        gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = torch.empty(G * cout, cin, 3, 3, dtype=torch.float32, device=dev)
            defer = wgrad_can_defer(*ctx.params)
            for g in range(G):
                d = L.mi_wgrad_desc()
                d.x, d.dy, d.gw = fh.data_ptr() + g * cin * 2, gy.data_ptr() + g * IAM_GS * 2, gw.data_ptr() + g * cout * cin * 9 * 4
                d.ldx, d.ldy, d.N, d.H, d.W, d.outH, d.outW, d.stride = _ld(fh), G * IAM_GS, B, H, W, H, W, 1
                d.Cin, d.Cout, d.CinPad, d.CoutPad, d.ntaps = cin, cout, cin, IAM_GS, 9
                for t in range(9):
                    d.tap_dy[t], d.tap_dx[t] = t // 3 - 1, t % 3 - 1
                if defer:
                    WgradBatch.add(d, (fh, gy), ctx.params[0])
                else:
                    need = L.lib().mi_conv2d_wgrad_plan(C.byref(d))
                    L.check(need, "mi_conv2d_wgrad_plan")
                    ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=dev)
                    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
                    L.check(L.lib().mi_conv2d_wgrad(C.byref(d), L.stream_ptr()), "mi_conv2d_wgrad (grouped IAM conv)")
            if defer:
                WgradBatch.flush()                               # the G jobs as one grouped launch
        if ctx.has_bias and ctx.needs_input_grad[2]:
            CP = G * IAM_GS
            sums = torch.empty(CP, dtype=torch.float32, device=dev)
            ws = torch.empty(128 * CP, dtype=torch.float32, device=dev)
            L.check(L.lib().mi_colsum_bf16_wide(gy.data_ptr(), CP, B * H * W, CP, sums.data_ptr(), 0, ws.data_ptr(), L.stream_ptr()),
                    "mi_colsum_bf16_wide")
            gb = sums.view(G, IAM_GS)[:, :cout].reshape(G * cout)
        return dx, gw, gb, None


def _aggregate_padded(iam_h, features, G, cout):
    """_aggregate on the padded IAM map of _GroupIamFn: iam_h bf16 [B, H, W, G * 128] -> inst fp32 [B, G * cout, C] and the
    normaliser [B, G * cout].  The outer product and the column sums run over all G * 128 channels - the 0.5s the sigmoid makes
    of the zero pad channels land in rows that are sliced off here, and the slices' backward hands those rows zero gradients,
    so the pad channels of the map's gradient stay exactly zero for the convolution backward."""
    B, H, W, CP = iam_h.shape
    Cc = features.shape[1]
    prob = _Ew1.apply(iam_h, "sigmoid")                                # [B, H, W, G * 128]
    fh = _nhwc(features)
    outer = _PixelOuterBatchFn.apply(prob.reshape(B, H * W, CP), fh.reshape(B, H * W, Cc))      # one grouped launch for the batch
    inst = outer.view(B, G, IAM_GS, Cc)[:, :, :cout].reshape(B, G * cout, Cc)
    norm = _ColSumFn.apply(prob.reshape(B, H * W, CP)).view(B, G, IAM_GS)[:, :, :cout].reshape(B, G * cout)
    return inst, norm


class _ColSumFn(torch.autograd.Function):
    """[B, P, N] bf16 -> fp32 [B, N] sums over P (see _colsums); backward: the [B, N] gradient broadcast over the pixels"""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return _colsums(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        B, P, N = ctx.shape
        return g.to(torch.bfloat16)[:, None, :].expand(B, P, N)


class InstanceBranch(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        d = cfg.MODEL.SPARSE_INST.DECODER
        dim, self.num_classes = d.INST.DIM, d.NUM_CLASSES
        self.inst_convs = _make_stack_3x3_convs(d.INST.CONVS, in_channels, dim)
        self.iam_conv = nn.Conv2d(dim, d.NUM_MASKS, 3, padding=1)
        self.cls_score = nn.Linear(dim, self.num_classes)
        self.mask_kernel = nn.Linear(dim, d.KERNEL_DIM)
        self.objectness = nn.Linear(dim, 1)
        self.prior_prob = 0.01
        _init_inst_heads(self)

    def forward(self, features):
        features = _run_stack(self.inst_convs, features)
        iam = _conv(features, self.iam_conv)
        inst, norm = _aggregate(iam, features)
        inst = inst / norm.clamp(min=1e-6)[:, :, None]
        return _linear(inst, self.cls_score), _linear(inst, self.mask_kernel), _linear(inst, self.objectness), iam


def _init_inst_heads(m):
    bias_value = -math.log((1 - m.prior_prob) / m.prior_prob)
    for module in (m.iam_conv, m.cls_score):
        nn.init.constant_(module.bias, bias_value)
    nn.init.normal_(m.iam_conv.weight, std=0.01)
    nn.init.normal_(m.cls_score.weight, std=0.01)
    nn.init.normal_(m.mask_kernel.weight, std=0.01)
    nn.init.constant_(m.mask_kernel.bias, 0.0)


class GroupInstanceBranch(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        d = cfg.MODEL.SPARSE_INST.DECODER
        dim, self.num_groups, self.num_classes = d.INST.DIM, d.GROUPS, d.NUM_CLASSES
        self.inst_convs = _make_stack_3x3_convs(d.INST.CONVS, in_channels, dim)
        expand_dim = dim * self.num_groups
        self.iam_conv = nn.Conv2d(dim, d.NUM_MASKS * self.num_groups, 3, padding=1, groups=self.num_groups)
        self.fc = nn.Linear(expand_dim, expand_dim)
        self.cls_score = nn.Linear(expand_dim, self.num_classes)
        self.mask_kernel = nn.Linear(expand_dim, d.KERNEL_DIM)
        self.objectness = nn.Linear(expand_dim, 1)
        self.prior_prob = 0.01
        _init_inst_heads(self)
        nn.init.kaiming_uniform_(self.fc.weight, a=1)
        nn.init.constant_(self.fc.bias, 0)

    def forward(self, features):
        features = _run_stack(self.inst_convs, features)
        # grouped 3x3 conv = its `groups` independent convs over channel slices
        G = self.num_groups
        cin, cout = features.shape[1] // G, self.iam_conv.out_channels // G
        if _IAM_PADDED() and cin % 32 == 0 and cout <= IAM_GS and cout % 4 == 0:
            # one padded NHWC map written by the G convolutions in place and read in place by everything behind it
            iam_h = _GroupIamFn.apply(features, self.iam_conv.weight, self.iam_conv.bias, G)
            inst, norm = _aggregate_padded(iam_h, features, G, cout)
            Bq, Hq, Wq = iam_h.shape[:3]
            iam = lambda: iam_h.view(Bq, Hq, Wq, G, IAM_GS)[..., :cout].reshape(Bq, Hq, Wq, G * cout).permute(0, 3, 1, 2)   # (only OUTPUT_IAM reads it)
        else:
            iam = torch.cat([torch.ops.mi355.conv2d(features[:, g * cin:(g + 1) * cin], self.iam_conv.weight[g * cout:(g + 1) * cout],
                                                    self.iam_conv.bias[g * cout:(g + 1) * cout], 1, 1) for g in range(G)], 1)
            inst, norm = _aggregate(iam, features)
        inst = inst / norm.clamp(min=1e-6, max=1e5)[:, :, None]
        B, N = inst.shape[:2]
        d4 = N // 4
        inst = inst.reshape(B, 4, d4, -1).transpose(1, 2).reshape(B, d4, -1)
        inst = _Ew1.apply(_linear(inst, self.fc), "relu")
        return _linear(inst, self.cls_score), _linear(inst, self.mask_kernel), _linear(inst, self.objectness), iam


class MaskBranch(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        d = cfg.MODEL.SPARSE_INST.DECODER
        self.mask_convs = _make_stack_3x3_convs(d.MASK.CONVS, in_channels, d.MASK.DIM)
        self.projection = nn.Conv2d(d.MASK.DIM, d.KERNEL_DIM, kernel_size=1)
        nn.init.kaiming_normal_(self.projection.weight, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(self.projection.bias, 0)

    def forward(self, features):
        return _conv(_run_stack(self.mask_convs, features), self.projection)


class BaseIAMDecoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        in_channels = cfg.MODEL.SPARSE_INST.ENCODER.NUM_CHANNELS + 2
        self.scale_factor = cfg.MODEL.SPARSE_INST.DECODER.SCALE_FACTOR
        self.output_iam = cfg.MODEL.SPARSE_INST.DECODER.OUTPUT_IAM
        self.inst_branch = InstanceBranch(cfg, in_channels)
        self.mask_branch = MaskBranch(cfg, in_channels)

    @torch.no_grad()
    def compute_coordinates(self, x):
        h, w = x.size(2), x.size(3)
        y_loc = torch.linspace(-1, 1, h, device=x.device)
        x_loc = torch.linspace(-1, 1, w, device=x.device)
        y_loc, x_loc = torch.meshgrid(y_loc, x_loc, indexing="ij")
        y_loc = y_loc.expand([x.shape[0], 1, -1, -1])
        x_loc = x_loc.expand([x.shape[0], 1, -1, -1])
        return torch.cat([x_loc, y_loc], 1).to(x)

    def forward(self, features):
        if not features.is_cuda:
            raise L.MI355Error("SparseInst decoder: the MI355X path needs device tensors (no CPU fallback)")
        if _PADDED_COORDS():
            # the coordinate planes are a constant of the map shape: built once per shape, kept (address-stable) across steps
            key = (tuple(features.shape[:1] + features.shape[2:]), features.device, features.dtype)
            cache = self.__dict__.setdefault("_coords", {})
            if key not in cache:
                cache[key] = self.compute_coordinates(features).permute(0, 2, 3, 1).contiguous()     # [B, H, W, 2] (x, y)
            features = PaddedNHWC(_CoordCat.apply(features, cache[key]), features.shape[1] + 2)
        else:
            features = torch.cat([self.compute_coordinates(features), features], dim=1)
        pred_logits, pred_kernel, pred_scores, iam = self.inst_branch(features)
        mask_features = self.mask_branch(features)
        B, Cc, H, W = mask_features.shape
        N = pred_kernel.shape[1]
        Np = _rup(N, 32)
        mf = _nhwc(mask_features)
        # the predicted kernels ARE the weights of a per-image 1x1 conv over the mask features (padded to 32 instances)
        if _MASK_BATCH() and Cc % 32 == 0:
            masks = _MaskKernelFn.apply(mf.reshape(B, H * W, Cc), pred_kernel, Np).reshape(B, H, W, Np)
        else:
            masks = torch.stack([_LinearFn.apply(mf[b].reshape(H * W, Cc), _pad_rows(pred_kernel[b].float(), Np), None)
                                 for b in range(B)]).reshape(B, H, W, Np)
        Ho, Wo = int(H * self.scale_factor), int(W * self.scale_factor)
        masks = _Resize.apply(masks.contiguous(), Ho, Wo)                       # [B, Ho, Wo, Np] (NHWC, instance = channel)
        output = {"pred_logits": pred_logits.float(), "pred_masks": masks.permute(0, 3, 1, 2)[:, :N],
                  "pred_scores": pred_scores.float(), "_masks_nhwc": masks}
        if self.output_iam:
            output["pred_iam"] = resize_bilinear(iam() if callable(iam) else iam, (Ho, Wo))
        return output


def _MASK_BATCH():
    """MI_SI_MASK_BATCH=0: the masks as B `_LinearFn` nodes + torch.stack (round 5's form; A/B, tests)"""
    import os
    return os.environ.get("MI_SI_MASK_BATCH", "1") != "0"


class _MaskKernelFn(torch.autograd.Function):
    """decoder_sparseinst.py:141-147 `pred_masks = torch.bmm(pred_kernel, mask_features.view(B, C, H * W))` as ONE autograd node:
    image b's predicted kernels are the weight of a 1x1 convolution over its mask features, written straight into row block b
    of the batch's mask map.  As B `_LinearFn` nodes (round 5) autograd indexed the two operands per image and stacked the
    results: per image and step a zero fill + a slice copy + a full-size addition for EACH operand's gradient (SelectBackward
    of a [B, P, C] map: 3 x 13 MB), a padded copy of the kernels, and a weight gradient of its own (launch + reduce) - 80 of the
    ~890 launches of a captured step.  Here: forward one pack + one convolution per image (the pack also leaves the
    data-gradient image), backward one convolution per image into its rows of the feature gradient and ONE grouped
    weight-gradient launch for the batch (pixel_outer_batch).  mf bf16 [B, P, C] dense, kern [B, N, C] -> bf16 [B, P, Np]."""

    @staticmethod
    def forward(ctx, mf, kern, Np):
        B, P, Cc = mf.shape
        N = kern.shape[1]
        if N == Np:
            k32 = kern.detach().float().contiguous()
        else:
            k32 = torch.zeros(B, Np, Cc, dtype=torch.float32, device=mf.device)
            k32[:, :N] = kern.detach()
        out = torch.empty(B, P, Np, dtype=torch.bfloat16, device=mf.device)
        wds = []
        for b in range(B):
            wf, wd = pack_images(k32[b], Np, Cc, 1, 1, Cc, Np, Np, Cc, dgrad=ctx.needs_input_grad[0])
            wds.append(wd)
            _conv1x1(mf[b], wf, out[b], P, Cc, Np, Np)
        ctx.save_for_backward(mf, *[w for w in wds if w is not None])
        ctx.dims, ctx.kdtype = (B, P, Cc, N, Np), kern.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        mf, *wds = ctx.saved_tensors
        B, P, Cc, N, Np = ctx.dims
        g = g.contiguous()
        dmf = dk = None
        if ctx.needs_input_grad[0]:
            dmf = torch.empty_like(mf)
            for b in range(B):
                _conv1x1(g[b], wds[b], dmf[b], P, Np, Cc, Cc)
        if ctx.needs_input_grad[1]:
            dk = pixel_outer_batch(g, mf)[:, :N].to(ctx.kdtype)          # [B, Np, C] = sum_p g[b, p, n] * mf[b, p, c]
        return dmf, dk, None


def _pad_rows(w, Np):
    if w.shape[0] == Np:
        return w
    return torch.cat([w, w.new_zeros(Np - w.shape[0], w.shape[1])], 0)


class GroupIAMDecoder(BaseIAMDecoder):
    def __init__(self, cfg):
        super().__init__(cfg)
        in_channels = cfg.MODEL.SPARSE_INST.ENCODER.NUM_CHANNELS + 2
        self.inst_branch = GroupInstanceBranch(cfg, in_channels)


def _IAM_PADDED():
    """MI_SI_IAM_PADDED=0: the grouped IAM conv as G ops + torch.cat and per-image padded copies (round 5's form; A/B, tests)"""
    import os
    return os.environ.get("MI_SI_IAM_PADDED", "1") != "0"


def _PADDED_COORDS():
    """MI_SI_PADDED_COORDS=0: the cat + per-convolution operand building of round 4 (A/B switch)"""
    import os
    return os.environ.get("MI_SI_PADDED_COORDS", "1") != "0"


_ENCODERS = {"InstanceContextEncoder": InstanceContextEncoder}
_DECODERS = {"BaseIAMDecoder": BaseIAMDecoder, "GroupIAMDecoder": GroupIAMDecoder}


def build_sparse_inst_encoder(cfg, input_shape):
    return _ENCODERS[cfg.MODEL.SPARSE_INST.ENCODER.NAME](cfg, input_shape)


def build_sparse_inst_decoder(cfg):
    return _DECODERS[cfg.MODEL.SPARSE_INST.DECODER.NAME](cfg)


# ------------------------------------------------------------------------------------------------ criterion
class PackedMaskTargets:
    """The ground truth of a batch on the DEVICE at fixed capacity (`cap` instances per image, a multiple of 32): target
    masks already resized to the prediction size (nested_masks_from_list, utils/misc.py:148-170, + the bilinear resize of
    sparseinst_loss.py:138-143) as fp32 [B * cap, Ho * Wo] - image b's instance j is row b * cap + j, unused rows zero -,
    labels int64 [B, cap], the instance counts as the cumulative table the assignment kernel reads, and 1 / num_instances
    (already averaged over the ranks, sparseinst_loss.py:206-212) as a device scalar.  Everything the criterion needs without
    a host value: a captured step serves any batch of the same padded shape, and the eager step never synchronises."""

    def __init__(self, B, cap, size, device):
        assert cap % 32 == 0
        self.B, self.cap, self.size = B, cap, tuple(size)
        P = size[0] * size[1]
        self.tgt = torch.zeros(B * cap, P, dtype=torch.float32, device=device)
        self.labels = torch.zeros(B, cap, dtype=torch.int64, device=device)
        self.off = torch.zeros(B + 1, dtype=torch.int32, device=device)
        self.inv_num = torch.ones(1, dtype=torch.float32, device=device)
        self.t2 = torch.zeros(B, cap, dtype=torch.float32, device=device)      # sum_p t^2 of every row (dice denominators)
        # the matcher's operand: every image's targets as bf16 [P, cap] (pixel-major).  Ground truth, so it is laid out
        # HERE, once per batch, not by 2 x B transpose / cast launches inside every captured step
        self.tgtT = torch.zeros(B, P, cap, dtype=torch.bfloat16, device=device)
        self.sizes = [0] * B

    def fill(self, targets, input_shape):
        """targets: per image {"labels": int64 [M], "masks": [M, h, w]} (prepare_targets); refilled IN PLACE"""
        dev = self.tgt.device
        if feed_batch_enabled() and dev.type == "cuda" and self.cap * 128 <= 65536:
            # pad + resize + pack of every image's masks, the bf16 transpose and the labels: one launch
            ms = [t["masks"].tensor if hasattr(t["masks"], "tensor") else t["masks"] for t in targets]
            sizes = [int(m.shape[0]) for m in ms]
            if max(sizes + [0]) > self.cap:
                raise ValueError(f"SparseInst: {max(sizes)} instances in one image, target capacity is {self.cap}")
            mask_targets_batch(ms, [t["labels"] for t in targets], self.cap, input_shape, self.size, self.tgt, self.tgtT, self.labels,
                               t2=self.t2)
            self.sizes = sizes
            return self._fill_counts(sizes)
        self.tgt.zero_()
        self.labels.zero_()
        sizes = []
        for b, t in enumerate(targets):
            m = t["masks"]
            m = m.tensor if hasattr(m, "tensor") else m
            M = int(m.shape[0])
            if M > self.cap:
                raise ValueError(f"SparseInst: {M} instances in one image, target capacity is {self.cap}")
            sizes.append(M)
            if M == 0:
                continue
            pad = torch.zeros(M, input_shape[0], input_shape[1], device=dev)
            pad[:, : m.shape[1], : m.shape[2]] = m.to(dev).float()
            r = F.interpolate(pad[:, None], size=self.size, mode="bilinear", align_corners=False).squeeze(1)     # ground-truth preparation
            self.tgt[b * self.cap: b * self.cap + M] = r.flatten(1)
            self.labels[b, :M] = t["labels"].to(dev)
        self.sizes = sizes
        self.t2.copy_((self.tgt * self.tgt).sum(-1).view(self.B, self.cap))      # (eager, in the host half)
        self.tgtT.copy_(self.tgt.view(self.B, self.cap, -1).transpose(1, 2))
        return self._fill_counts(sizes)

    def _fill_counts(self, sizes):
        dev = self.tgt.device
        # (host values through the page-locked ring: a blocking copy here waits for the previous step's graph, ops.HostRing)
        HostRing.upload(self.off, [0] + [sum(sizes[:i + 1]) for i in range(len(sizes))])
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            # one process: 1 / max(num_instances, 1) is a host value (the same fp32 division, done here)
            import numpy as np
            HostRing.upload(self.inv_num, [float(np.float32(1.0) / np.float32(max(float(sum(sizes)), 1.0)))])
            return self
        num = HostRing.upload(torch.empty(1, device=dev), [float(sum(sizes))])
        torch.distributed.all_reduce(num)                                                # sparseinst_loss.py:210-212
        world = torch.distributed.get_world_size()
        self.inv_num.copy_(1.0 / torch.clamp(num / world, min=1.0))
        return self


def _pack_targets(targets, input_shape, size, device, cap=None):
    if cap is None:
        cap = max(32, _rup(max([len(t["labels"]) for t in targets] + [1]), 32))
    return PackedMaskTargets(len(targets), cap, size, device).fill(targets, input_shape)


def _FUSED_LOSS():
    """MI_SI_FUSED_LOSS=0: the matcher's cost matrix and the criterion's scalar half as torch calls (round 5's form; A/B, tests)"""
    import os
    return os.environ.get("MI_SI_FUSED_LOSS", "1") != "0"


def dice_terms_packed(masks_nhwc, pk):
    """the two device-heavy terms of dice_score_packed: num fp32 [B, Np, cap] = sum_p sigmoid * t and s2 fp32 [B, Np] =
    sum_p sigmoid^2 (the target term is pk.t2)"""
    B, Ho, Wo, Np = masks_nhwc.shape
    sig = _Ew1.apply(masks_nhwc.contiguous(), "sigmoid").reshape(B, Ho * Wo, Np)
    return pixel_outer_batch(sig, pk.tgtT), _colsums(sig, square=True)


def dice_score_packed(masks_nhwc, pk):
    """dice_score (sparseinst_loss.py:31-36) of every prediction against the `cap` target rows OF ITS IMAGE: fp32 [B, Np, cap]
    (columns of unused rows are 0).  masks_nhwc: bf16 logits [B, Ho, Wo, Np]"""
    B, Ho, Wo, Np = masks_nhwc.shape
    P = Ho * Wo
    sig = _Ew1.apply(masks_nhwc.contiguous(), "sigmoid").reshape(B, P, Np)
    s2 = _colsums(sig, square=True)                                                   # [B, Np]  sum_p sigmoid^2
    t2 = pk.t2                                                                        # [B, cap] sum_p t^2 (PackedMaskTargets.fill)
    num = pixel_outer_batch(sig, pk.tgtT)                                             # [B, Np, cap]; tgtT[b] = bf16 [P, cap]
    return (2.0 * num) / (s2[:, :, None] + t2[:, None, :] + 1e-4)


class SparseInstMatcher(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.alpha = cfg.MODEL.SPARSE_INST.MATCHER.ALPHA
        self.beta = cfg.MODEL.SPARSE_INST.MATCHER.BETA

    @torch.no_grad()
    def match_packed(self, outputs, pk):
        """SparseInstMatcher.forward (sparseinst_loss.py:300-354) with nothing on the host: cost = dice^alpha * prob^beta of
        every (prediction, target) pair of an image as [B, N, cap], maximised by the device assignment kernel (mi_lsap on
        the negated matrix; scipy's linear_sum_assignment(maximize=True) in the reference).  Returns match_q / match_t int64
        [B, cap] (the first nmatch[b] entries are the pairs, sorted by prediction index) and nmatch int32 [B]"""
        masks = outputs["_masks_nhwc"]
        B, Ho, Wo, Np = masks.shape
        N = outputs["pred_logits"].shape[1]
        dev = masks.device
        cap = pk.cap
        if _FUSED_LOSS():
            # dice, class probability and the two powers in one launch (mi_sparseinst_match_cost) over the outer products
            num, s2 = dice_terms_packed(masks, pk)
            logits = outputs["pred_logits"].float().contiguous()
            C_ = logits.shape[2]
            cost = torch.empty(B, N, cap, dtype=torch.float32, device=dev)
            L.check(L.lib().mi_sparseinst_match_cost(num.data_ptr(), s2.data_ptr(), pk.t2.data_ptr(), logits.data_ptr(),
                                                     pk.labels.data_ptr(), B, N, num.shape[1], C_, cap, float(self.alpha),
                                                     float(self.beta), cost.data_ptr(), L.stream_ptr()), "mi_sparseinst_match_cost")
            mqt = torch.zeros(2, B, cap, dtype=torch.int64, device=dev)
            mq, mt = mqt[0], mqt[1]
        else:
            prob = outputs["pred_logits"].float().sigmoid()                                          # [B, N, C]
            scores = dice_score_packed(masks, pk)[:, :N]                                             # [B, N, cap]
            pl = prob.gather(2, pk.labels[:, None, :].expand(B, N, cap))
            cost = (-((scores ** self.alpha) * (pl ** self.beta))).contiguous()
            mq = torch.zeros(B, cap, dtype=torch.int64, device=dev)
            mt = torch.zeros(B, cap, dtype=torch.int64, device=dev)
        nm = torch.zeros(B, dtype=torch.int32, device=dev)
        L.check(L.lib().mi_lsap(cost.data_ptr(), pk.off.data_ptr(), B, N, cap, mq.data_ptr(), mt.data_ptr(), nm.data_ptr(),
                                L.stream_ptr()), "mi_lsap")
        return mq, mt, nm

    @torch.no_grad()
    def forward(self, outputs, targets, input_shape):
        """the reference's return value - a list of (prediction indices, target indices) per image - for callers that want
        it on the host (tests; this one synchronises).  A NaN cost matrix raises like scipy does."""
        masks = outputs["_masks_nhwc"]
        pk = _pack_targets(targets, input_shape, masks.shape[1:3], masks.device)
        mq, mt, nm = self.match_packed(outputs, pk)
        n = nm.tolist()
        if any(v < 0 for v in n):
            raise ValueError("matrix contains invalid numeric entries")
        return [(mq[b, : n[b]].clone(), mt[b, : n[b]].clone()) for b in range(len(n))], pk


class _MaskLossFn(torch.autograd.Function):
    """(mask logits NHWC, fixed-capacity pair table) -> stats fp32 [K, 8] of every pair (rows of unused pairs zero; layout in
    csrc/sparseinst_ops.hip); backward: d/d(mask logits) from the upstream gradients of (sum of BCE sums, sum of dice
    losses), handed to the kernel as a DEVICE pair"""

    @staticmethod
    def forward(ctx, masks, tgt, pairs, valid):
        B, Ho, Wo, Np = masks.shape
        P = Ho * Wo
        K = pairs.shape[0]
        stats = torch.empty(K, 8, dtype=torch.float32, device=masks.device)
        ws = torch.empty(int(L.lib().mi_sparseinst_mask_stats_ws_floats(K, P)), dtype=torch.float32, device=masks.device)
        L.check(L.lib().mi_sparseinst_mask_stats(masks.data_ptr(), Np, P, tgt.data_ptr(), pairs.data_ptr(), K,
                                                 stats.data_ptr(), ws.data_ptr(), L.stream_ptr()), "mi_sparseinst_mask_stats")
        bce = stats[:, 0].sum()
        dice = ((1.0 - 2.0 * stats[:, 1] / (stats[:, 2] + stats[:, 3] + 1e-4)) * valid).sum()
        iou = stats[:, 4] / (stats[:, 6] + stats[:, 5] - stats[:, 4] + 1e-6)
        ctx.save_for_backward(masks, tgt, pairs, stats)
        return torch.cat([bce[None], dice[None], iou])

    @staticmethod
    def backward(ctx, g):
        masks, tgt, pairs, stats = ctx.saved_tensors
        B, Ho, Wo, Np = masks.shape
        dm = torch.zeros_like(masks)
        coef = g[:2].float().contiguous()
        L.check(L.lib().mi_sparseinst_mask_grad_dev(masks.data_ptr(), Np, Ho * Wo, tgt.data_ptr(), pairs.data_ptr(),
                                                    pairs.shape[0], stats.data_ptr(), coef.data_ptr(), dm.data_ptr(),
                                                    L.stream_ptr()), "mi_sparseinst_mask_grad_dev")
        return dm, None, None, None


class _CriterionFn(torch.autograd.Function):
    """SparseInstCriterion.forward behind the matcher as ONE autograd node: (class logits fp32 [B, N, C], objectness logits
    fp32 [B, N, 1], mask logits bf16 NHWC, packed targets, the assignment) -> the four WEIGHTED losses fp32 [4] (loss_ce,
    loss_mask, loss_dice, loss_objectness; sparseinst_loss.py:214-297).  Forward: pair table (mi_sparseinst_pairs), the
    pairs' mask statistics (mi_sparseinst_mask_stats), the losses (mi_sparseinst_head_loss).  Backward: d logits, d scores
    and the mask kernels' two coefficients from the four upstream gradients (mi_sparseinst_head_loss_bwd), then
    mi_sparseinst_mask_grad_dev.  Focal loss alpha 0.25 / gamma 2 (sparseinst_loss.py:230-236)."""

    @staticmethod
    def forward(ctx, logits, scores, masks, tgt, labels, inv_num, mq, mt, nm, use, weights):
        B, Ho, Wo, Np = masks.shape
        N, C_ = logits.shape[1], logits.shape[2]
        cap = mq.shape[1]
        P, K = Ho * Wo, B * cap
        dev = masks.device
        logits, scores = logits.contiguous(), scores.contiguous()
        mq, mt = mq.contiguous(), mt.contiguous()
        ibuf = torch.empty(K * 3 + 2 * B * N, dtype=torch.int32, device=dev)
        fbuf = torch.empty(K + 1 + 4, dtype=torch.float32, device=dev)
        stats = torch.empty(K, 8, dtype=torch.float32, device=dev)
        d = L.mi_sparseinst_loss_desc()
        d.logits, d.scores, d.labels = logits.data_ptr(), scores.data_ptr(), labels.data_ptr()
        d.match_q, d.match_t, d.nmatch, d.inv_num = mq.data_ptr(), mt.data_ptr(), nm.data_ptr(), inv_num.data_ptr()
        d.pairs, d.row_cls, d.row_pair = ibuf.data_ptr(), ibuf[K * 3:].data_ptr(), ibuf[K * 3 + B * N:].data_ptr()
        d.valid, d.kdev, d.losses = fbuf.data_ptr(), fbuf[K:].data_ptr(), fbuf[K + 1:].data_ptr()
        d.stats = stats.data_ptr()
        d.B, d.N, d.C, d.cap, d.P, d.use_labels, d.use_masks = B, N, C_, cap, P, use[0], use[1]
        d.alpha, d.gamma = 0.25, 2.0
        d.w_ce, d.w_mask, d.w_dice, d.w_obj = [float(w) for w in weights]
        lib, sp = L.lib(), L.stream_ptr()
        L.check(lib.mi_sparseinst_pairs(C.byref(d), sp), "mi_sparseinst_pairs")
        if use[1]:
            ws = torch.empty(int(lib.mi_sparseinst_mask_stats_ws_floats(K, P)), dtype=torch.float32, device=dev)
            L.check(lib.mi_sparseinst_mask_stats(masks.data_ptr(), Np, P, tgt.data_ptr(), d.pairs, K, stats.data_ptr(), ws.data_ptr(), sp),
                    "mi_sparseinst_mask_stats")
        L.check(lib.mi_sparseinst_head_loss(C.byref(d), sp), "mi_sparseinst_head_loss")
        ctx.desc, ctx.keep = d, (logits, scores, labels, inv_num, mq, mt, nm, ibuf, fbuf, stats)
        ctx.save_for_backward(masks, tgt)
        ctx.scores_shape = scores.shape
        # four scalar outputs (views of one buffer), not one [4] tensor the caller indexes: every index is a select node whose
        # backward is a zero fill + a copy, and the four of them meet in three additions
        return tuple(fbuf[K + 1 + i] for i in range(4))

    @staticmethod
    def backward(ctx, *gs):
        masks, tgt = ctx.saved_tensors
        d = ctx.desc
        B, Ho, Wo, Np = masks.shape
        dev = masks.device
        z = None
        for g in gs:
            if g is None and z is None:
                z = torch.zeros((), dtype=torch.float32, device=dev)
        gup = torch.stack([z if g is None else g.float() for g in gs])
        dlogits = torch.empty(d.B, d.N, d.C, dtype=torch.float32, device=dev)
        dscores = torch.empty(ctx.scores_shape, dtype=torch.float32, device=dev)
        coef = torch.empty(2, dtype=torch.float32, device=dev)
        d.gup, d.dlogits, d.dscores, d.coef = gup.data_ptr(), dlogits.data_ptr(), dscores.data_ptr(), coef.data_ptr()
        lib, sp = L.lib(), L.stream_ptr()
        L.check(lib.mi_sparseinst_head_loss_bwd(C.byref(d), sp), "mi_sparseinst_head_loss_bwd")
        dm = None
        if ctx.needs_input_grad[2]:
            dm = torch.zeros_like(masks)
            if d.use_masks:
                L.check(lib.mi_sparseinst_mask_grad_dev(masks.data_ptr(), Np, Ho * Wo, tgt.data_ptr(), d.pairs, d.B * d.cap,
                                                        d.stats, coef.data_ptr(), dm.data_ptr(), sp), "mi_sparseinst_mask_grad_dev")
        return (dlogits if ctx.needs_input_grad[0] else None, dscores if ctx.needs_input_grad[1] else None, dm,
                None, None, None, None, None, None, None, None)


def sigmoid_focal_loss(inputs, targets, alpha=0.25, gamma=2.0):
    """fvcore.nn.sigmoid_focal_loss_jit(reduction='sum') (un-vendored; published formula)"""
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.sum(-1).sum()        # (rows first: both reductions stay single-block, see _colsums)


class SparseInstCriterion(nn.Module):
    """SparseInstCriterion (loss/sparseinst_loss.py:190-297) with the matching AND the matched-pair bookkeeping on the device
    (round 4; rounds 2-3 built the pair table on the host from `nmatch.tolist()` - three synchronisations per step, which
    also kept the step from being captured).  The assignment leaves (match_q, match_t, nmatch) on the device; every loss is
    written over the FIXED B x cap pair grid with the unused pairs masked:
      labels   the one-hot target gets 1 at (b, match_q, label[match_t]) by an index_put of the pair-validity mask
      masks    pair table rows (b or -1, q, b * cap + t): the kernels skip rows with b < 0
      counts   K = sum nmatch (clamped to 1 where it divides) and 1 / num_instances are device scalars"""

    def __init__(self, cfg, matcher):
        super().__init__()
        self.matcher = matcher
        ls = cfg.MODEL.SPARSE_INST.LOSS
        self.losses = ls.ITEMS
        self.weight_dict = dict(loss_ce=ls.CLASS_WEIGHT, loss_mask=ls.MASK_PIXEL_WEIGHT, loss_dice=ls.MASK_DICE_WEIGHT,
                                loss_objectness=ls.OBJECTNESS_WEIGHT)
        self.num_classes = cfg.MODEL.SPARSE_INST.DECODER.NUM_CLASSES

    def forward(self, outputs, targets, input_shape=None):
        """targets: PackedMaskTargets (SparseInst.prepare_batch) or the reference's list of per-image dicts (packed here)"""
        masks = outputs["_masks_nhwc"]
        pk = targets if isinstance(targets, PackedMaskTargets) else \
            _pack_targets(targets, input_shape, masks.shape[1:3], masks.device)
        mq, mt, nm = self.matcher.match_packed(outputs, pk)
        B, cap = mq.shape
        dev = masks.device
        if _FUSED_LOSS() and outputs["pred_logits"].dtype == torch.float32 and outputs["pred_scores"].dtype == torch.float32:
            # pair table, the four losses and their gradients: four launches + the two mask kernels (csrc/sparseinst_loss.hip)
            use = (int("labels" in self.losses), int("masks" in self.losses))
            wd = self.weight_dict
            r = _CriterionFn.apply(outputs["pred_logits"], outputs["pred_scores"], masks.contiguous(), pk.tgt, pk.labels, pk.inv_num,
                                   mq, mt, nm, use, (wd["loss_ce"], wd["loss_mask"], wd["loss_dice"], wd["loss_objectness"]))
            losses = {}
            if use[0]:
                losses["loss_ce"] = r[0]
            if use[1]:
                losses.update(loss_mask=r[1], loss_dice=r[2], loss_objectness=r[3])
            return losses
        nmc = nm.clamp(min=0)                       # (an invalid cost matrix - nmatch < 0 - contributes no pair; no host raise)
        valid = torch.arange(cap, device=dev)[None, :] < nmc[:, None]                           # [B, cap]
        vf = valid.float()
        inv_num = pk.inv_num[0]
        bi = torch.arange(B, device=dev)[:, None].expand(B, cap)
        losses = {}
        if "labels" in self.losses:
            src_logits = outputs["pred_logits"]
            cls = pk.labels.gather(1, mt)                                                       # label of the matched target
            labels = torch.zeros_like(src_logits)
            labels.index_put_((bi.reshape(-1), mq.reshape(-1), cls.reshape(-1)), vf.reshape(-1).to(labels.dtype),
                              accumulate=True)       # (unused pairs add 0 somewhere in row 0 of their image)
            losses["loss_ce"] = sigmoid_focal_loss(src_logits.flatten(0, 1), labels.flatten(0, 1)) * inv_num
        if "masks" in self.losses:
            kdev = vf.sum().clamp(min=1.0)
            rows = torch.stack([torch.where(valid, bi, torch.full_like(bi, -1)), mq, bi * cap + mt], -1)
            pairs = rows.reshape(-1, 3).to(torch.int32).contiguous()
            r = _MaskLossFn.apply(masks.contiguous(), pk.tgt, pairs, vf.reshape(-1))
            P = masks.shape[1] * masks.shape[2]
            losses["loss_mask"] = r[0] / (kdev * P)                   # BCE 'mean' over the K x P matched elements
            losses["loss_dice"] = r[1] * inv_num
            scr = outputs["pred_scores"][..., 0].gather(1, mq)                                  # [B, cap]
            obj = F.binary_cross_entropy_with_logits(scr, r[2:].detach().view(B, cap).to(scr.dtype), reduction="none")
            losses["loss_objectness"] = (obj.float() * vf).sum() / kdev
        for k in list(losses.keys()):
            if k in self.weight_dict:
                losses[k] = losses[k] * self.weight_dict[k]
        return losses


def build_sparse_inst_criterion(cfg):
    return SparseInstCriterion(cfg, SparseInstMatcher(cfg))


# ------------------------------------------------------------------------------------------------ meta architecture
def rescoring_mask(scores, mask_pred, masks):
    mask_pred_ = mask_pred.float()
    return scores * ((masks * mask_pred_).sum([1, 2]) / (mask_pred_.sum([1, 2]) + 1e-6))


@META_ARCH_REGISTRY.register()
class SparseInst(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.backbone = build_backbone(cfg)
        self.size_divisibility = self.backbone.size_divisibility
        self.encoder = build_sparse_inst_encoder(cfg, self.backbone.output_shape())
        self.decoder = build_sparse_inst_decoder(cfg)
        self.criterion = build_sparse_inst_criterion(cfg)
        self.mask_format = cfg.INPUT.MASK_FORMAT
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1), persistent=False)
        self.pixel_mean_host, self.pixel_std_host = [float(v) for v in cfg.MODEL.PIXEL_MEAN], [float(v) for v in cfg.MODEL.PIXEL_STD]
        self.cls_threshold = cfg.MODEL.YOLO.CONF_THRESHOLD
        self.mask_threshold = cfg.MODEL.SPARSE_INST.MASK_THRESHOLD
        self.max_detections = cfg.MODEL.SPARSE_INST.MAX_DETECTIONS
        # prediction masks: the stride-8 encoder map up-sampled by DECODER.SCALE_FACTOR (2.0 -> stride 4)
        self.mask_stride = int(round(8 / cfg.MODEL.SPARSE_INST.DECODER.SCALE_FACTOR))
        self.to(self.device)

    def normalizer(self, image):
        return (image - self.pixel_mean) / self.pixel_std

    def preprocess_inputs(self, batched_inputs):
        images = [self.normalizer(x["image"].to(self.device).float()) for x in batched_inputs]
        return ImageList.from_tensors(images, 32)

    def prepare_targets(self, targets):
        new_targets = []
        for t in targets:
            h, w = t.image_size
            if not t.has("gt_masks"):
                gt_masks = torch.empty(0, h, w)
            else:
                gt_masks = t.gt_masks
                if self.mask_format == "polygon":
                    raise NotImplementedError("SparseInst: INPUT.MASK_FORMAT bitmask (Base-SparseInst.yaml:34)")
                gt_masks = gt_masks.tensor if hasattr(gt_masks, "tensor") else gt_masks
            new_targets.append({"labels": t.gt_classes.to(self.device), "masks": gt_masks.to(self.device)})
        return new_targets

    # ---- the training step split at the host / device line (graph_step.GraphedTrainStep captures the device half)
    target_capacity = 96      # instances per image the packed targets hold AT LEAST (a multiple of 32; COCO: <= 93 after crowd
                              # removal); an image with more instances raises the batch's capacity to the next multiple of 32
                              # (the reference's criterion takes any count, sparseinst_loss.py:320-346) - a shape of its own

    def _capacity(self, batched_inputs):
        mx = max((len(x["instances"]) for x in batched_inputs if "instances" in x), default=0)
        return max(self.target_capacity, (mx + 31) // 32 * 32)

    def grad_cut_modules(self):
        """GraphedTrainStep's backward stages under data parallel: encoder + decoder + criterion, then res5, res4, res3"""
        return self.backbone.stage_modules() if hasattr(self.backbone, "stage_modules") else []

    def batch_key(self, batched_inputs):
        r = 32                          # ImageList.from_tensors(images, 32) of preprocess_inputs (sparseinst.py:95-98)
        up = lambda v: (v + r - 1) // r * r
        return (len(batched_inputs), up(max(int(x["image"].shape[-2]) for x in batched_inputs)),
                up(max(int(x["image"].shape[-1]) for x in batched_inputs)), self._capacity(batched_inputs))

    def prepare_batch(self, batched_inputs, static=None):
        """everything of the training forward that touches the host: normalise + zero-pad the images into one tensor
        (ImageList.from_tensors(images, 32), sparseinst.py:95-98) and move the ground truth over as PackedMaskTargets (masks
        padded to the batch shape and resized to the prediction size HERE, eagerly).  With `static` - an earlier result for
        the same batch_key - everything is refilled IN PLACE."""
        B, Hp, Wp, cap = self.batch_key(batched_inputs)
        dev = self.device
        if static is None:
            st = self.mask_stride
            static = dict(images=torch.zeros(B, 3, Hp, Wp, device=dev), key=(B, Hp, Wp, cap),
                          targets=PackedMaskTargets(B, cap, (Hp // st, Wp // st), dev))
        assert static["key"] == (B, Hp, Wp, cap), (static["key"], (B, Hp, Wp, cap))
        img = static["images"]
        if feed_batch_enabled():
            # normalise + zero-pad the whole batch in one launch (the reference: two torch calls + a slice copy per image)
            normalize_pad_batch([x["image"] for x in batched_inputs], img, self.pixel_mean_host, self.pixel_std_host)
        else:
            img.zero_()
            for b, x in enumerate(batched_inputs):
                t = self.normalizer(x["image"].to(dev).float())
                img[b, :, : t.shape[-2], : t.shape[-1]].copy_(t)
        gt = self.prepare_targets([x["instances"].to(dev) for x in batched_inputs])
        static["targets"].fill(gt, (Hp, Wp))
        return static

    def forward_prepared(self, static):
        """the device half: backbone, encoder, decoder, criterion - no host value of the batch enters a launch"""
        output = self.decoder(self.encoder(self.backbone(static["images"])))
        return self.criterion(output, static["targets"])

    def forward(self, batched_inputs):
        if self.device.type != "cuda":
            raise L.MI355Error(f"SparseInst on MODEL.DEVICE={self.device}: the MI355X path needs a HIP device (no CPU fallback)")
        images = self.preprocess_inputs(batched_inputs)
        max_shape = images.tensor.shape[2:]
        features = self.backbone(images.tensor)
        output = self.decoder(self.encoder(features))
        if self.training:
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
            # (the same fixed-capacity packing as prepare_batch: the eager and the captured step run identical launches)
            pk = _pack_targets(self.prepare_targets(gt_instances), max_shape, output["_masks_nhwc"].shape[1:3], self.device,
                               cap=self._capacity(batched_inputs))
            return self.criterion(output, pk)
        results = self.inference(output, batched_inputs, max_shape, images.image_sizes)
        return [{"instances": r} for r in results]

    @torch.no_grad()
    def inference(self, output, batched_inputs, max_shape, image_sizes):
        """sparseinst.py:169-234: sqrt(cls * objectness) scores, threshold, maskness rescoring, masks resized to the
        padded input, cropped to the image, resized to the requested output size, thresholded"""
        results = []
        pred_scores = torch.sqrt(output["pred_logits"].sigmoid() * output["pred_scores"].sigmoid())
        pred_masks = output["pred_masks"].float().sigmoid()
        for scores_per_image, mask_pred, inp, img_shape in zip(pred_scores, pred_masks, batched_inputs, image_sizes):
            ori_shape = (inp.get("height", img_shape[0]), inp.get("width", img_shape[1]))
            result = Instances(ori_shape)
            scores, labels = scores_per_image.max(dim=-1)
            keep = scores > self.cls_threshold
            scores, labels, mask_pred = scores[keep], labels[keep], mask_pred[keep]
            if scores.size(0) == 0:
                result.scores, result.pred_classes = scores, labels
                results.append(result)
                continue
            h, w = img_shape
            scores = rescoring_mask(scores, mask_pred > self.mask_threshold, mask_pred)
            mask_pred = F.interpolate(mask_pred.unsqueeze(1), size=tuple(max_shape), mode="bilinear", align_corners=False)[:, :, :h, :w]
            mask_pred = F.interpolate(mask_pred, size=ori_shape, mode="bilinear", align_corners=False).squeeze(1)
            result.pred_masks = mask_pred > self.mask_threshold
            result.scores, result.pred_classes = scores, labels
            results.append(result)
        return results
