"""Attention core of nn.MultiheadAttention for DETR's transformer (detr_backbone.py:140,155-157,200-202,222-230),
as a torch.autograd.Function over libmi355det's fused MFMA kernels (mi_mha_fwd / mi_mha_bwd).

q, k, v: bf16 [L, B, E] (sequence first, E = num_heads * 32) — the tensors after the in-projection;
key_padding_mask: bool/uint8 [B, Lk] (True = padded key).  Returns bf16 [Lq, B, E] (before the out-projection).
"""
import math

import torch

from .. import _lib as L


def _O32():
    import os
    return os.environ.get("MI_MHA_O32", "1") != "0"


class _MhaCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, num_heads, drop_p=0.0, seed=0):
        if not q.is_cuda:
            raise L.MI355Error("mha_core: the MI355X path needs device tensors (no CPU fallback)")
        Lq, B, E = q.shape
        Lk = k.shape[0]
        q, k, v = (t.to(torch.bfloat16).contiguous() for t in (q, k, v))
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        o = torch.empty_like(q)
        # fp32 copy of the output for the backward's delta = rowsum(dO o O) (see mi_mha_fwd_dropout_o32): only when a
        # backward will follow; MI_MHA_O32=0 keeps the bf16-only form (A/B, tests)
        need32 = _O32() and any(ctx.needs_input_grad[:3])
        o32 = torch.empty(Lq, B, E, dtype=torch.float32, device=q.device) if need32 else None
        lse = torch.empty(B, num_heads, Lq, dtype=torch.float32, device=q.device)
        scale = 1.0 / math.sqrt(E // num_heads)
        L.check(L.lib().mi_mha_fwd_dropout_o32(q.data_ptr(), k.data_ptr(), v.data_ptr(), L.ptr(m), o.data_ptr(), L.ptr(o32),
                                               lse.data_ptr(), B, num_heads, Lq, Lk, E, scale, float(drop_p), int(seed),
                                               L.stream_ptr()), "mi_mha_fwd")
        ctx.save_for_backward(q, k, v, m, o, lse, o32)
        ctx.num_heads, ctx.scale, ctx.drop = num_heads, scale, (float(drop_p), int(seed))
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, m, o, lse, o32 = ctx.saved_tensors
        Lq, B, E = q.shape
        Lk = k.shape[0]
        do = do.to(torch.bfloat16).contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        L.check(L.lib().mi_mha_bwd_dropout_o32(q.data_ptr(), k.data_ptr(), v.data_ptr(), L.ptr(m), o.data_ptr(), L.ptr(o32),
                                               lse.data_ptr(), do.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                               dv.data_ptr(), B, ctx.num_heads, Lq, Lk, E, ctx.scale, ctx.drop[0], ctx.drop[1],
                                               L.stream_ptr()), "mi_mha_bwd")
        _dump(ctx, q, k, v, m, o, lse, do, dq, dk, dv)
        return dq, dk, dv, None, None, None, None


class _MhaCoreQK(torch.autograd.Function):
    """the attention core of a SELF-attention whose query and key are one tensor (q = k = x + pos: every encoder layer and
    the decoder's self-attention, detr_backbone.py:155-157,222-224): qk bf16 [L, B, 2E] holds q in columns [0, E) and k in
    [E, 2E) - ONE in-projection launch produced both - and the backward hands dq | dk back in the same layout, so that the
    projection's data gradient and weight gradient are one launch each (mi_mha_fwd / bwd_dropout_ld, row strides 2E)"""

    @staticmethod
    def forward(ctx, qk, v, mask, num_heads, drop_p=0.0, seed=0):
        if not qk.is_cuda:
            raise L.MI355Error("mha_core: the MI355X path needs device tensors (no CPU fallback)")
        Lq, B, E2 = qk.shape
        E = E2 // 2
        qk, v = qk.to(torch.bfloat16).contiguous(), v.to(torch.bfloat16).contiguous()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        o = torch.empty_like(v)
        need32 = _O32() and any(ctx.needs_input_grad[:2])
        o32 = torch.empty(Lq, B, E, dtype=torch.float32, device=qk.device) if need32 else None
        lse = torch.empty(B, num_heads, Lq, dtype=torch.float32, device=qk.device)
        scale = 1.0 / math.sqrt(E // num_heads)
        L.check(L.lib().mi_mha_fwd_dropout_ld(qk.data_ptr(), E2, qk.data_ptr() + 2 * E, E2, v.data_ptr(), L.ptr(m), o.data_ptr(),
                                              L.ptr(o32), lse.data_ptr(), B, num_heads, Lq, Lq, E, scale, float(drop_p), int(seed),
                                              L.stream_ptr()), "mi_mha_fwd (packed q | k)")
        ctx.save_for_backward(qk, v, m, o, lse, o32)
        ctx.num_heads, ctx.scale, ctx.drop = num_heads, scale, (float(drop_p), int(seed))
        return o

    @staticmethod
    def backward(ctx, do):
        qk, v, m, o, lse, o32 = ctx.saved_tensors
        Lq, B, E2 = qk.shape
        E = E2 // 2
        do = do.to(torch.bfloat16).contiguous()
        dqk, dv = torch.empty_like(qk), torch.empty_like(v)
        delta = torch.empty_like(lse)
        L.check(L.lib().mi_mha_bwd_dropout_ld(qk.data_ptr(), E2, qk.data_ptr() + 2 * E, E2, v.data_ptr(), L.ptr(m), o.data_ptr(),
                                              L.ptr(o32), lse.data_ptr(), do.data_ptr(), delta.data_ptr(), dqk.data_ptr(),
                                              dqk.data_ptr() + 2 * E, dv.data_ptr(), B, ctx.num_heads, Lq, Lq, E, ctx.scale,
                                              ctx.drop[0], ctx.drop[1], L.stream_ptr()), "mi_mha_bwd (packed q | k)")
        return dqk, dv, None, None, None, None


def mha_core_qk(qk, v, key_padding_mask=None, num_heads=8, dropout_p=0.0, seed=0):
    return _MhaCoreQK.apply(qk, v, key_padding_mask, num_heads, dropout_p, seed)


_DUMPS = [0]


def _dump(ctx, q, k, v, m, o, lse, do, dq, dk, dv):
    """debugging aid (MI_MHA_DUMP=<dir>, MI_MHA_DUMP_SEL=<comma-separated backward-call indices>): the operands and
    results of selected backward calls as .pt files, for the offline error analysis of tools/attn_bwd_error.py"""
    import os
    d = os.environ.get("MI_MHA_DUMP")
    if not d:
        return
    i = _DUMPS[0]
    _DUMPS[0] += 1
    sel = os.environ.get("MI_MHA_DUMP_SEL", "")
    if sel and str(i) not in sel.split(","):
        return
    os.makedirs(d, exist_ok=True)
    torch.save(dict(q=q.cpu(), k=k.cpu(), v=v.cpu(), mask=None if m is None else m.cpu(), o=o.cpu(), lse=lse.cpu(), do=do.cpu(),
                    dq=dq.cpu(), dk=dk.cpu(), dv=dv.cpu(), heads=ctx.num_heads, scale=ctx.scale, drop=ctx.drop), os.path.join(d, f"mha_bwd_{i}.pt"))


def mha_core(q, k, v, key_padding_mask=None, num_heads=8, dropout_p=0.0, seed=0):
    """dropout_p > 0: attention-weight dropout (training mode of nn.MultiheadAttention(dropout=p)); the mask is a pure
    function of `seed`, recomputed by the backward kernels"""
    return _MhaCore.apply(q, k, v, key_padding_mask, num_heads, dropout_p, seed)
