"""DETR transformer on the MI355X kernels — drop-in for yolov7/modeling/backbone/detr_backbone.py:25-278
(`Transformer`, `TransformerEncoder`, `TransformerDecoder`, `TransformerEncoderLayer`, `TransformerDecoderLayer`) and the
`nn.MultiheadAttention` / `nn.Linear` / `nn.LayerNorm` they are built from (config 4 of BASELINE.json).

Same constructor, attribute names and state_dict keys as the reference (self_attn / multihead_attn .in_proj_weight /
in_proj_bias / out_proj.{weight,bias}, linear1, linear2, norm1..3, encoder.layers.N.*, decoder.layers.N.*, *.norm),
same forward signatures.  Every op runs in libmi355det:
  * nn.Linear       -> the implicit-GEMM conv kernel as a 1x1 convolution over the token rows (fwd, dgrad, wgrad)
  * attention core  -> mi_mha_fwd / mi_mha_bwd (fused MFMA attention, key-padding mask)
  * nn.LayerNorm    -> mi_layernorm_fwd / bwd
  * residual / ReLU -> mi_ew_bf16
  * dropout (p = 0.1 in the reference's configs, yolov7/config.py:228) -> mi_dropout_bf16 for the residual / FFN
    dropouts and a mask inside the fused attention kernels for nn.MultiheadAttention's attention-weight dropout.  Masks
    are counter-based (a pure function of a per-call seed drawn from torch's default generator, so torch.manual_seed
    makes a run reproducible) and recomputed in the backward pass; torch's own Philox stream is not reproduced - no
    parity target exists for it - the tests check the statistics and the exact gradient for the mask in use.
Tokens are bf16 [L, B, E] (sequence first, as the reference passes them).
"""
import ctypes as C
import math

import torch
from torch import nn

from .. import _lib as L
from ..ops import WgradBatch, pack_images, wgrad_bias_fused, wgrad_flush_point
from .attention import mha_core, mha_core_qk


def _rup(a, b):
    return (a + b - 1) // b * b


def _factor(T):
    for w in (64, 32, 16, 8, 4, 2, 1):
        if T % w == 0:
            return T // w, w


def _conv1x1(x, w_img, y, T, K, Cout, CoutPad, bias=None, relu=False):
    H, W = _factor(T)
    d = L.mi_conv_desc()
    d.x, d.w, d.y = x.data_ptr(), w_img.data_ptr(), y.data_ptr()
    d.bias = L.ptr(bias)
    d.flags = L.MI_CONV_RELU if relu else 0
    d.ldx, d.ldy = x.shape[-1], y.shape[-1]
    d.N, d.H, d.W, d.outH, d.outW, d.gridH, d.gridW = 1, H, W, H, W, H, W
    d.in_stride = d.out_stride = 1
    d.K8, d.Cout, d.CoutPad, d.ntaps = K // 8, Cout, CoutPad, 1
    if T <= 1024 and K >= 1024 and W >= 16 and CoutPad % 64 == 0:
        # a long reduction over few rows (the decoder's FFN: 400 x 2048 -> 256): 16-row tiles put 4x as many blocks on the
        # chip as the launcher's 64-row default, each with the same K loop (tools/linear_sweep.py: 25.3 -> 16.2 us)
        d.TH, d.TW, d.KC, d.BN = 1, 16, 128, 64
    L.check(L.lib().mi_conv2d(C.byref(d), L.stream_ptr()), "mi_conv2d (linear)")


def _linear_fwd(x, w32, bias, relu=False):
    """y [T, rup(Cout,32)] = x W^T + b (relu: max(., 0) in the epilogue) and the data-gradient image of W.  x bf16 [T, Cin]
    contiguous, w32 fp32 [Cout, Cin] contiguous (a row block of a larger parameter is), bias fp32 [Cout] or None"""
    T, Cin = x.shape
    Cout = w32.shape[0]
    assert Cin % 32 == 0, "linear: input channels must be a multiple of 32"
    CoutP = _rup(Cout, 32)
    dev = x.device
    wf, wd = pack_images(w32, Cout, Cin, 1, 1, Cin, CoutP, CoutP, Cin)
    y = torch.empty(T, CoutP, dtype=torch.bfloat16, device=dev)
    b32 = None
    if bias is not None:
        if CoutP == Cout and bias.dtype == torch.float32 and bias.is_contiguous():
            b32 = bias                   # (no padded copy: two launches per Linear call otherwise)
        else:
            b32 = torch.zeros(CoutP, dtype=torch.float32, device=dev)
            b32[:Cout] = bias.float()
    _conv1x1(x, wf, y, T, Cin, CoutP, CoutP, b32, relu)
    return y, wd


def _can_defer(*params):
    """the weight gradient may be written later (ops.WgradBatch) only into a tensor autograd takes over untouched: every
    parameter is a leaf whose .grad is None (else AccumulateGrad adds the returned tensor to .grad right away, and a
    non-leaf weight's producer reads it right away)"""
    return WgradBatch.enabled() and all(p is None or (p.is_leaf and p.grad is None) for p in params)


def _linear_bwd(x, wd, dy, Cout, need_dx=True, gw=None, gb=None, defer_ok=False, owner=None):
    """(dx or None); the weight gradient is WRITTEN to gw (fp32 [Cout, Cin] contiguous: a row block of a larger gradient
    is) and the bias gradient to gb (fp32 [Cout]) when they are given"""
    T, Cin = x.shape
    CoutP = _rup(Cout, 32)
    dev = x.device
    if CoutP != Cout:   # zero-padded out-gradient: K of the data-gradient GEMM is the padded channel count
        dyp = torch.zeros(T, CoutP, dtype=torch.bfloat16, device=dev)
        dyp[:, :Cout] = dy
        dy = dyp
    dy = dy.contiguous()
    dx = None
    if need_dx:
        dx = torch.empty(T, Cin, dtype=torch.bfloat16, device=dev)
        _conv1x1(dy, wd, dx, T, CoutP, Cin, Cin)
    defer = defer_ok and gw is not None and (gb is None or CoutP == Cout)
    fused_gb = gb is not None and gw is not None and CoutP == Cout and (defer or wgrad_bias_fused(T))
    if gw is not None:
        H, W = _factor(T)
        d = L.mi_wgrad_desc()
        d.x, d.dy, d.gw = x.data_ptr(), dy.data_ptr(), gw.data_ptr()
        if fused_gb:
            d.gbias = gb.data_ptr()          # the bias gradient from the same two launches (mi_wgrad_desc.gbias)
        d.ldx, d.ldy, d.N, d.H, d.W, d.outH, d.outW, d.stride = Cin, CoutP, 1, H, W, H, W, 1
        d.Cin, d.Cout, d.CinPad, d.CoutPad, d.ntaps = Cin, Cout, Cin, CoutP, 1
        if defer:
            # one grouped launch per transformer layer (ops.WgradBatch): only the job is registered here; x and dy stay
            # alive until the layer's flush point, gw / gb are written by the group's reduce grid
            WgradBatch.add(d, (x, dy), owner)
            return dx
        need = L.lib().mi_conv2d_wgrad_plan(C.byref(d))
        L.check(need, "mi_conv2d_wgrad_plan")
        ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=dev)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
        L.check(L.lib().mi_conv2d_wgrad(C.byref(d), L.stream_ptr()), "mi_conv2d_wgrad (linear)")
    if gb is not None and not fused_gb:
        gbp = gb if CoutP == Cout else torch.empty(CoutP, dtype=torch.float32, device=dev)
        cws = torch.empty(128 * CoutP, dtype=torch.float32, device=dev)
        L.check(L.lib().mi_colsum_bf16_wide(dy.data_ptr(), CoutP, T, CoutP, gbp.data_ptr(), 0, cws.data_ptr(), L.stream_ptr()),
                "mi_colsum_bf16_wide")
        if gbp is not gb:
            gb.copy_(gbp[:Cout])
    return dx


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b over token rows; x bf16 [T, Cin], W fp32 [Cout, Cin] (nn.Linear layout), b fp32 [Cout].
    Cin must be a multiple of 32; any Cout: the output channels are zero-padded to a multiple of 32 inside (the packed
    weight image and the bias carry zero rows), the caller sees [T, Cout]."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        """relu=True: y = relu(x W^T + b) with the ReLU in the convolution's epilogue (the FFN's linear1 + activation,
        detr_backbone.py:166 `self.linear2(self.dropout(self.activation(self.linear1(src))))`: one launch instead of two)"""
        Cout = weight.shape[0]
        y, wd = _linear_fwd(x, weight.detach().float().contiguous(), None if bias is None else bias.detach(), relu)
        out = y if y.shape[1] == Cout else y[:, :Cout]
        ctx.save_for_backward(x, wd, out if relu else None)
        ctx.dims = (Cout, bias is not None)
        ctx.params = (weight, bias)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, wd, y = ctx.saved_tensors
        Cout, has_bias = ctx.dims
        if y is not None:
            dy = _ew(dy.contiguous(), y.contiguous(), 2)       # dy * (y > 0)
        gw = torch.empty(Cout, x.shape[1], dtype=torch.float32, device=x.device)
        gb = torch.empty(Cout, dtype=torch.float32, device=x.device) if has_bias else None
        dx = _linear_bwd(x, wd, dy, Cout, True, gw, gb, defer_ok=_can_defer(*ctx.params), owner=ctx.params[0])
        return dx, gw, gb, None


class _InProjFn(torch.autograd.Function):
    """nn.MultiheadAttention's in-projection (detr_backbone.py:140,155-157,222-230 reach F.multi_head_attention_forward's
    _in_projection_packed): q, k, v = query W[:E]^T + b[:E], key W[E:2E]^T + b[E:2E], value W[2E:]^T + b[2E:], as ONE
    autograd node over the WHOLE in_proj_weight [3E, E] / in_proj_bias [3E].  Slicing the parameters through autograd
    instead costs, per attention module and step, 6 zero fills + 6 slice copies + 4 full-size additions (SliceBackward of
    three weight and three bias blocks, then their accumulation): 288 of the ~1900 launches of a DETR-R50 step.  Here the
    three weight-gradient launches write their row blocks of one [3E, E] tensor directly."""

    @staticmethod
    def forward(ctx, xq, xk, xv, w, b):
        E = w.shape[1]
        w32, b32 = w.detach().float().contiguous(), b.detach().float().contiguous()
        outs, wds = [], []
        for i, x in enumerate((xq, xk, xv)):
            y, wd = _linear_fwd(x, w32[i * E:(i + 1) * E], b32[i * E:(i + 1) * E])
            outs.append(y)
            wds.append(wd)
        ctx.save_for_backward(xq, xk, xv, *wds)
        ctx.E = E
        ctx.params = (w, b)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        xq, xk, xv, wq, wk, wv = ctx.saved_tensors
        E = ctx.E
        gw = torch.empty(3 * E, E, dtype=torch.float32, device=xq.device)
        gb = torch.empty(3 * E, dtype=torch.float32, device=xq.device)
        dxs = []
        ok = _can_defer(*ctx.params)
        for i, (x, wd, dy) in enumerate(((xq, wq, dq), (xk, wk, dk), (xv, wv, dv))):
            dxs.append(_linear_bwd(x, wd, dy, E, ctx.needs_input_grad[i], gw[i * E:(i + 1) * E], gb[i * E:(i + 1) * E],
                                   defer_ok=ok, owner=ctx.params[0]))
        return dxs[0], dxs[1], dxs[2], gw, gb


class _InProjQKFn(torch.autograd.Function):
    """_InProjFn for a self-attention whose query and key are the SAME tensor: q | k = x W[:2E]^T + b[:2E] as ONE [T, 2E]
    projection (one forward, one data-gradient, one weight-gradient launch where _InProjFn runs two of each, and no
    accumulation of the two data gradients by autograd), v as before.  Bit-identical values: the packed image of W[:2E] is the
    two row blocks' images side by side, every output element is the same dot product."""

    @staticmethod
    def forward(ctx, xqk, xv, w, b):
        E = w.shape[1]
        w32, b32 = w.detach().float().contiguous(), b.detach().float().contiguous()
        yqk, wdqk = _linear_fwd(xqk, w32[: 2 * E], b32[: 2 * E])
        yv, wdv = _linear_fwd(xv, w32[2 * E:], b32[2 * E:])
        ctx.save_for_backward(xqk, xv, wdqk, wdv)
        ctx.E = E
        ctx.params = (w, b)
        return yqk, yv

    @staticmethod
    def backward(ctx, dqk, dv):
        xqk, xv, wdqk, wdv = ctx.saved_tensors
        E = ctx.E
        gw = torch.empty(3 * E, E, dtype=torch.float32, device=xqk.device)
        gb = torch.empty(3 * E, dtype=torch.float32, device=xqk.device)
        ok = _can_defer(*ctx.params)
        dxqk = _linear_bwd(xqk, wdqk, dqk, 2 * E, ctx.needs_input_grad[0], gw[: 2 * E], gb[: 2 * E], defer_ok=ok, owner=ctx.params[0])
        dxv = _linear_bwd(xv, wdv, dv, E, ctx.needs_input_grad[1], gw[2 * E:], gb[2 * E:], defer_ok=ok, owner=ctx.params[0])
        return dxqk, dxv, gw, gb


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        T, E = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(T, dtype=torch.float32, device=x.device)
        rstd = torch.empty(T, dtype=torch.float32, device=x.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        L.check(L.lib().mi_layernorm_fwd(x.data_ptr(), g32.data_ptr(), b32.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), T, E, eps, L.stream_ptr()), "mi_layernorm_fwd")
        ctx.save_for_backward(x, g32, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32, mean, rstd = ctx.saved_tensors
        T, E = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(E, dtype=torch.float32, device=x.device)
        db = torch.empty(E, dtype=torch.float32, device=x.device)
        ws = torch.empty((T + 15) // 16 * E * 2, dtype=torch.float32, device=x.device)
        L.check(L.lib().mi_layernorm_bwd(x.data_ptr(), dy.data_ptr(), g32.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                         dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), T, E,
                                         L.stream_ptr()), "mi_layernorm_bwd")
        return dx, dg, db, None


class _AddDroppedLayerNormFn(torch.autograd.Function):
    """LayerNorm(res + dropout(x)) - the post-norm residual of every transformer sub-layer (detr_backbone.py:163-168,
    235-243) - as one node: forward = ONE launch (mi_dropout_add_layernorm_fwd: the sum is formed, rounded to bf16 and
    written while the row's statistics are taken), backward = mi_layernorm_bwd_dropout, which writes the residual's
    gradient dx and the dropped branch's dropout(dx) in the same pass.  Bit-identical to _AddDroppedFn (or _AddFn for
    p = 0) followed by _LayerNormFn: two launches forward and two backward less per norm."""

    @staticmethod
    def forward(ctx, res, x, gamma, beta, eps, p, seed):
        T, E = x.shape
        res, x = res.contiguous(), x.contiguous()
        y, sm = torch.empty_like(x), torch.empty_like(x)
        mean = torch.empty(T, dtype=torch.float32, device=x.device)
        rstd = torch.empty(T, dtype=torch.float32, device=x.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        L.check(L.lib().mi_dropout_add_layernorm_fwd(x.data_ptr(), res.data_ptr(), sm.data_ptr(), g32.data_ptr(), b32.data_ptr(),
                                                     y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), T, E, eps, float(p), int(seed),
                                                     L.stream_ptr()), "mi_dropout_add_layernorm_fwd")
        ctx.save_for_backward(sm, g32, mean, rstd)
        ctx.ps = (float(p), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        sm, g32, mean, rstd = ctx.saved_tensors
        T, E = sm.shape
        dy = dy.contiguous()
        dx = torch.empty_like(sm)
        dxd = torch.empty_like(sm) if ctx.ps[0] > 0 else None
        dg = torch.empty(E, dtype=torch.float32, device=sm.device)
        db = torch.empty(E, dtype=torch.float32, device=sm.device)
        ws = torch.empty((T + 15) // 16 * E * 2, dtype=torch.float32, device=sm.device)
        L.check(L.lib().mi_layernorm_bwd_dropout(sm.data_ptr(), dy.data_ptr(), g32.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                 dx.data_ptr(), L.ptr(dxd), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), T, E,
                                                 ctx.ps[0], ctx.ps[1], L.stream_ptr()), "mi_layernorm_bwd_dropout")
        return dx, (dx if dxd is None else dxd), dg, db, None, None, None


def _ew(a, b, op):
    out = torch.empty_like(a)
    L.check(L.lib().mi_ew_bf16(a.data_ptr(), L.ptr(b), out.data_ptr(), a.numel(), op, L.stream_ptr()), "mi_ew_bf16")
    return out


def _next_seed():
    """a fresh 62-bit seed from torch's default (CPU) generator"""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.contiguous()
        out = torch.empty_like(x)
        L.check(L.lib().mi_dropout_bf16(x.data_ptr(), out.data_ptr(), x.numel(), float(p), int(seed), L.stream_ptr()),
                "mi_dropout_bf16")
        ctx.ps = (float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty_like(g)
        L.check(L.lib().mi_dropout_bf16(g.data_ptr(), out.data_ptr(), g.numel(), ctx.ps[0], ctx.ps[1], L.stream_ptr()),
                "mi_dropout_bf16 (backward)")
        return out, None, None


def _dropout(x, p, training):
    if not training or p <= 0:
        return x
    return _DropoutFn.apply(x, p, _next_seed())


class _AddDroppedFn(torch.autograd.Function):
    """res + dropout(x) in one pass (mi_dropout_add_bf16); backward: (g, dropout(g)) - the mask is a pure function of the seed"""

    @staticmethod
    def forward(ctx, res, x, p, seed):
        res, x = res.contiguous(), x.contiguous()
        out = torch.empty_like(x)
        L.check(L.lib().mi_dropout_add_bf16(x.data_ptr(), res.data_ptr(), out.data_ptr(), x.numel(), float(p), int(seed),
                                            L.stream_ptr()), "mi_dropout_add_bf16")
        ctx.ps = (float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gx = torch.empty_like(g)
        L.check(L.lib().mi_dropout_bf16(g.data_ptr(), gx.data_ptr(), g.numel(), ctx.ps[0], ctx.ps[1], L.stream_ptr()),
                "mi_dropout_bf16 (backward)")
        return g, gx, None, None


def _add_dropped(res, x, p, training):
    """res + F.dropout(x, p, training)"""
    if not training or p <= 0:
        return _AddFn.apply(res, x)
    return _AddDroppedFn.apply(res, x, p, _next_seed())


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return _ew(a.contiguous(), b.contiguous(), 0)

    @staticmethod
    def backward(ctx, g):
        return g, g


class _ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        y = _ew(a.contiguous(), None, 1)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return _ew(g.contiguous(), y, 2)


def _tok(x):
    if not x.is_cuda:
        raise L.MI355Error("transformer: the MI355X path needs device tensors (no CPU fallback)")
    return x.to(torch.bfloat16).contiguous()


def _qk_packed():
    """MI_MHA_QK_PACKED=0: q and k of a self-attention as two projections (round 4's form; A/B, tests)"""
    import os
    return os.environ.get("MI_MHA_QK_PACKED", "1") != "0"


class MultiheadAttention(nn.Module):
    """nn.MultiheadAttention(embed_dim, num_heads) with the reference's usage: separate query/key/value, key_padding_mask,
    no attn_mask, batch_first=False; parameters named as torch names them."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim == num_heads * 32, "the MFMA attention kernel is built for head_dim 32 (DETR: 256 = 8 x 32)"
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None):
        if attn_mask is not None:
            raise NotImplementedError("attn_mask (the reference always passes None)")
        E = self.embed_dim
        Lq, B, _ = query.shape
        Lk = key.shape[0]
        drop = self.dropout if self.training else 0.0
        if query is key and _qk_packed():
            # self-attention with q = k = x + pos (detr_backbone.py:155-157,222-224): one [T, 2E] projection feeds both
            qk, v = _InProjQKFn.apply(_tok(query).view(Lq * B, E), _tok(value).view(Lk * B, E), self.in_proj_weight, self.in_proj_bias)
            o = mha_core_qk(qk.view(Lq, B, 2 * E), v.view(Lk, B, E), key_padding_mask, self.num_heads, drop,
                            _next_seed() if drop > 0 else 0)
            out = _LinearFn.apply(o.reshape(Lq * B, E), self.out_proj.weight, self.out_proj.bias).view(Lq, B, E)
            return out, None
        q, k, v = _InProjFn.apply(_tok(query).view(Lq * B, E), _tok(key).view(Lk * B, E), _tok(value).view(Lk * B, E),
                                  self.in_proj_weight, self.in_proj_bias)
        q, k, v = q.view(Lq, B, E), k.view(Lk, B, E), v.view(Lk, B, E)
        drop = self.dropout if self.training else 0.0
        o = mha_core(q, k, v, key_padding_mask, self.num_heads, drop, _next_seed() if drop > 0 else 0)
        out = _LinearFn.apply(o.reshape(Lq * B, E), self.out_proj.weight, self.out_proj.bias).view(Lq, B, E)
        return out, None


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError(activation)
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout_p = dropout
        self.normalize_before = normalize_before

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else _AddFn.apply(_tok(tensor), _tok(pos))

    def _ln(self, norm, x):
        Lx, B, E = x.shape
        return _LayerNormFn.apply(x.reshape(Lx * B, E), norm.weight, norm.bias, norm.eps).view(Lx, B, E)

    def _ln_res(self, norm, res, x):
        """norm(res + dropout(x)): one node, one launch forward (see _AddDroppedLayerNormFn)"""
        Lx, B, E = x.shape
        p = self.dropout_p if self.training else 0.0
        return _AddDroppedLayerNormFn.apply(res.reshape(Lx * B, E), x.reshape(Lx * B, E), norm.weight, norm.bias, norm.eps, p,
                                            _next_seed() if p > 0 else 0).view(Lx, B, E)

    def _ffn(self, x):
        Lx, B, E = x.shape
        h = _LinearFn.apply(x.reshape(Lx * B, E), self.linear1.weight, self.linear1.bias, True)
        h = _dropout(h, self.dropout_p, self.training)
        return _LinearFn.apply(h, self.linear2.weight, self.linear2.bias).view(Lx, B, E)

    def forward(self, src, src_mask=None, src_key_padding_mask=None, pos=None):
        src = wgrad_flush_point(_tok(src))      # this layer's weight gradients leave as one group when backward gets here
        if self.normalize_before:   # forward_pre (detr_backbone.py:170-182)
            src2 = self._ln(self.norm1, src)
            q = k = self.with_pos_embed(src2, pos)
            src2 = self.self_attn(q, k, value=src2, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)[0]
            src = _add_dropped(src, src2, self.dropout_p, self.training)
            src2 = self._ffn(self._ln(self.norm2, src))
            return _add_dropped(src, src2, self.dropout_p, self.training)
        # forward_post (detr_backbone.py:156-168)
        q = k = self.with_pos_embed(src, pos)
        src2 = self.self_attn(q, k, value=src, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)[0]
        src = self._ln_res(self.norm1, src, src2)
        src2 = self._ffn(src)
        return self._ln_res(self.norm2, src, src2)


class TransformerDecoderLayer(nn.Module):
    """detr_backbone.py:197-278: self-attention over the queries, cross-attention into the encoder memory, FFN"""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError(activation)
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout_p = dropout
        self.normalize_before = normalize_before

    with_pos_embed = staticmethod(TransformerEncoderLayer.with_pos_embed)
    _ln = TransformerEncoderLayer._ln
    _ln_res = TransformerEncoderLayer._ln_res
    _ffn = TransformerEncoderLayer._ffn

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None, mem_k=None):
        """mem_k: memory + pos when the caller has formed it already (the decoder stack does, once for its six layers:
        detr_backbone.py:226 adds the same two tensors in every layer)"""
        tgt, memory = wgrad_flush_point(_tok(tgt)), _tok(memory)
        if mem_k is None:
            mem_k = self.with_pos_embed(memory, pos)
        if self.normalize_before:   # forward_pre (detr_backbone.py:245-264)
            tgt2 = self._ln(self.norm1, tgt)
            q = k = self.with_pos_embed(tgt2, query_pos)
            tgt2 = self.self_attn(q, k, value=tgt2, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
            tgt = _add_dropped(tgt, tgt2, self.dropout_p, self.training)
            tgt2 = self._ln(self.norm2, tgt)
            tgt2 = self.multihead_attn(self.with_pos_embed(tgt2, query_pos), mem_k, value=memory, attn_mask=memory_mask,
                                       key_padding_mask=memory_key_padding_mask)[0]
            tgt = _add_dropped(tgt, tgt2, self.dropout_p, self.training)
            tgt2 = self._ffn(self._ln(self.norm3, tgt))
            return _add_dropped(tgt, tgt2, self.dropout_p, self.training)
        # forward_post (detr_backbone.py:222-243)
        q = k = self.with_pos_embed(tgt, query_pos)
        tgt2 = self.self_attn(q, k, value=tgt, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
        tgt = self._ln_res(self.norm1, tgt, tgt2)
        tgt2 = self.multihead_attn(self.with_pos_embed(tgt, query_pos), mem_k, value=memory, attn_mask=memory_mask,
                                   key_padding_mask=memory_key_padding_mask)[0]
        tgt = self._ln_res(self.norm2, tgt, tgt2)
        tgt2 = self._ffn(tgt)
        return self._ln_res(self.norm3, tgt, tgt2)


def _norm_tokens(norm, x):
    Lx, B, E = x.shape
    return _LayerNormFn.apply(x.reshape(Lx * B, E), norm.weight, norm.bias, norm.eps).view(Lx, B, E)


def _clones(module, n):
    import copy
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class TransformerEncoder(nn.Module):
    """detr_backbone.py:68-90"""

    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, mask=None, src_key_padding_mask=None, pos=None):
        output = _tok(src)
        pos = None if pos is None else _tok(pos)
        for layer in self.layers:
            output = layer(output, src_mask=mask, src_key_padding_mask=src_key_padding_mask, pos=pos)
        if self.norm is not None:
            output = _norm_tokens(self.norm, output)
        return output


class TransformerDecoder(nn.Module):
    """detr_backbone.py:93-132 (return_intermediate: the final norm applied to every layer's output, stacked)"""

    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        output = _tok(tgt)
        memory = _tok(memory)
        pos = None if pos is None else _tok(pos)
        query_pos = None if query_pos is None else _tok(query_pos)
        intermediate = []
        mem_k = TransformerDecoderLayer.with_pos_embed(memory, pos)      # the same sum in every layer: formed once
        for layer in self.layers:
            output = layer(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                           tgt_key_padding_mask=tgt_key_padding_mask, memory_key_padding_mask=memory_key_padding_mask,
                           pos=pos, query_pos=query_pos, mem_k=mem_k)
            if self.return_intermediate:
                intermediate.append(_norm_tokens(self.norm, output))
        if self.norm is not None:
            output = intermediate[-1] if self.return_intermediate else _norm_tokens(self.norm, output)
        if self.return_intermediate:
            return torch.stack(intermediate)
        return output.unsqueeze(0)


class Transformer(nn.Module):
    """detr_backbone.py:25-65: src [B, C, H, W] feature map, mask [B, H, W] (True = padding), query_embed [Q, C],
    pos_embed [B, C, H, W]  ->  (hs [layers, B, Q, C], memory [B, C, H, W]), both bf16"""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        enc = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.encoder = TransformerEncoder(enc, num_encoder_layers, nn.LayerNorm(d_model) if normalize_before else None)
        dec = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.decoder = TransformerDecoder(dec, num_decoder_layers, nn.LayerNorm(d_model),
                                          return_intermediate=return_intermediate_dec)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model, self.nhead = d_model, nhead

    def forward(self, src, mask, query_embed, pos_embed):
        bs, c, h, w = src.shape
        src = src.flatten(2).permute(2, 0, 1)
        pos_embed = pos_embed.flatten(2).permute(2, 0, 1)
        if pos_embed.is_cuda and not pos_embed.requires_grad:
            pos_embed = _tok(pos_embed)      # (a constant of the batch: its bf16 token layout once, not in every layer's _tok(pos))
        query_embed = query_embed.unsqueeze(1).repeat(1, bs, 1)
        mask = mask.flatten(1)
        if mask.is_cuda:
            mask = mask.to(torch.uint8).contiguous()     # (what the attention kernels read: once, not in each of the 18 attention calls)
        tgt = torch.zeros_like(query_embed, dtype=torch.bfloat16)
        memory = self.encoder(src, src_key_padding_mask=mask, pos=pos_embed)
        hs = self.decoder(tgt, memory, memory_key_padding_mask=mask, pos=pos_embed, query_pos=query_embed)
        return hs.transpose(1, 2), memory.permute(1, 2, 0).reshape(bs, c, h, w)
