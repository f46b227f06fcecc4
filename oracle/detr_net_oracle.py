"""TEST INFRASTRUCTURE (never imported by the product): fp32 torch restatement of the DETR network around the backbone -
`Transformer` / `TransformerEncoder(Layer)` / `TransformerDecoder(Layer)` (yolov7/modeling/backbone/detr_backbone.py:25-278,
post- and pre-norm), `PositionEmbeddingSine` (:309-375), `DETR.forward`'s input projection, class / box heads and `MLP`
(yolov7/modeling/meta_arch/detr.py:406-472) - as FUNCTIONS over a state_dict with the reference's keys, with the two hooks
the forward-pinned parity tests need (oracle/resnet_oracle.py has the same pair for the backbone):

  quant   storage-rounding emulation applied where the product stores a tensor (bf16 layer outputs)
  force   {site: tensor}: the VALUE of that intermediate becomes the given one (the implementation under test's), the
          gradient path stays this restatement's - every layer's local Jacobian is then evaluated at the same point on
          both sides, and a whole-network gradient comparison measures the backward, not the forward's rounding lottery
          sites: "enc.<i>" (encoder layer output), "memory", "dec.<i>" (decoder layer output, before the shared final norm)

Pinned: tests/test_oracle_golden.py::test_detr_net_oracle_against_reference_golden holds transformer() to the golden the
REFERENCE'S OWN Transformer class produced (tests/golden/transformer.npz: hs, memory, d src, d query, parameter gradients)."""
import math

import torch
import torch.nn.functional as F


def _force(y, force, site, q):
    if force is not None and site in force:
        y = y + (force[site].to(y.dtype) - y).detach()
    return q(y)


def mha(sd, p, query, key, value, nhead, key_padding_mask=None):
    """nn.MultiheadAttention.forward (the module detr_backbone.py:140,200-202 instantiates; dropout 0): packed in-projection,
    scaled dot product per head with the key-padding mask as -inf, out-projection.  [L, B, E] tensors"""
    E = query.shape[-1]
    W, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    qp = F.linear(query, W[:E], b[:E])
    kp = F.linear(key, W[E:2 * E], b[E:2 * E])
    vp = F.linear(value, W[2 * E:], b[2 * E:])
    Lq, B, _ = qp.shape
    Lk, d = kp.shape[0], E // nhead
    qh = qp.reshape(Lq, B, nhead, d).permute(1, 2, 0, 3) * (d ** -0.5)
    kh = kp.reshape(Lk, B, nhead, d).permute(1, 2, 0, 3)
    vh = vp.reshape(Lk, B, nhead, d).permute(1, 2, 0, 3)
    s = qh @ kh.transpose(-1, -2)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, -1) @ vh).permute(2, 0, 1, 3).reshape(Lq, B, E)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _ffn(sd, p, x):
    return F.linear(F.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), sd[p + ".linear2.weight"],
                    sd[p + ".linear2.bias"])


def encoder_layer(sd, p, src, mask, pos, nhead, pre):
    """TransformerEncoderLayer.forward_post / forward_pre (detr_backbone.py:153-183)"""
    if not pre:
        qk = src + pos
        src = _ln(sd, p + ".norm1", src + mha(sd, p + ".self_attn", qk, qk, src, nhead, mask))
        return _ln(sd, p + ".norm2", src + _ffn(sd, p, src))
    s2 = _ln(sd, p + ".norm1", src)
    qk = s2 + pos
    src = src + mha(sd, p + ".self_attn", qk, qk, s2, nhead, mask)
    return src + _ffn(sd, p, _ln(sd, p + ".norm2", src))


def decoder_layer(sd, p, tgt, memory, mask, pos, query_pos, nhead, pre):
    """TransformerDecoderLayer.forward_post / forward_pre (detr_backbone.py:210-262).  query_pos: one tensor, or the triple
    (for the self-attention query, the self-attention key, the cross-attention query) of equal values that the parity tests
    use to see the three contributions to d / d query_pos separately"""
    qs, ks, qc = query_pos if isinstance(query_pos, (tuple, list)) else (query_pos, query_pos, query_pos)
    if not pre:
        tgt = _ln(sd, p + ".norm1", tgt + mha(sd, p + ".self_attn", tgt + qs, tgt + ks, tgt, nhead))
        tgt = _ln(sd, p + ".norm2", tgt + mha(sd, p + ".multihead_attn", tgt + qc, memory + pos, memory, nhead, mask))
        return _ln(sd, p + ".norm3", tgt + _ffn(sd, p, tgt))
    t2 = _ln(sd, p + ".norm1", tgt)
    tgt = tgt + mha(sd, p + ".self_attn", t2 + qs, t2 + ks, t2, nhead)
    t2 = _ln(sd, p + ".norm2", tgt)
    tgt = tgt + mha(sd, p + ".multihead_attn", t2 + qc, memory + pos, memory, nhead, mask)
    return tgt + _ffn(sd, p, _ln(sd, p + ".norm3", tgt))


def transformer(sd, src, mask, query_embed, pos_embed, nhead=8, pre=False, prefix="", quant=None, force=None,
                query_embed_layers=None):
    """Transformer.forward (detr_backbone.py:52-65) with return_intermediate_dec: src / pos [B, C, H, W], mask bool [B, H, W],
    query_embed [Q, C] -> (hs [layers, B, Q, C], memory [B, C, H, W]).
    query_embed_layers: optional list of per-decoder-layer copies of query_embed (same values; a tensor or a (self q, self k,
    cross q) triple per layer): their gradients are the TERMS whose sum is d / d query_embed - the parity tests bound an
    error against the size of what is summed"""
    q = quant if quant is not None else (lambda t: t)
    B, Cc, H, W = src.shape
    x = src.flatten(2).permute(2, 0, 1)
    pos = pos_embed.flatten(2).permute(2, 0, 1)
    qe = query_embed[:, None, :].repeat(1, B, 1)
    m = mask.flatten(1)
    ne = 1 + max(int(k[len(prefix + "encoder.layers."):].split(".")[0]) for k in sd if k.startswith(prefix + "encoder.layers."))
    nd = 1 + max(int(k[len(prefix + "decoder.layers."):].split(".")[0]) for k in sd if k.startswith(prefix + "decoder.layers."))
    for i in range(ne):
        x = _force(encoder_layer(sd, f"{prefix}encoder.layers.{i}", x, m, pos, nhead, pre), force, f"enc.{i}", q)
    if pre:
        x = _ln(sd, prefix + "encoder.norm", x)
    memory = _force(x, force, "memory", q)
    out, inter = torch.zeros_like(qe), []
    for i in range(nd):
        if query_embed_layers is None:
            qi = qe
        elif isinstance(query_embed_layers[i], (tuple, list)):
            qi = tuple(t[:, None, :].repeat(1, B, 1) for t in query_embed_layers[i])
        else:
            qi = query_embed_layers[i][:, None, :].repeat(1, B, 1)
        out = _force(decoder_layer(sd, f"{prefix}decoder.layers.{i}", out, memory, m, pos, qi, nhead, pre), force, f"dec.{i}", q)
        inter.append(q(_ln(sd, prefix + "decoder.norm", out)))
    return torch.stack(inter).transpose(1, 2), memory.permute(1, 2, 0).reshape(B, Cc, H, W)


def position_embedding_sine(mask, num_pos_feats=128, temperature=10000, normalize=True, scale=2 * math.pi):
    """PositionEmbeddingSine.forward (detr_backbone.py:325-348): mask bool [B, H, W] (True = padding) -> [B, 2 N, H, W]"""
    nm = ~mask
    y = nm.cumsum(1, dtype=torch.float32)
    x = nm.cumsum(2, dtype=torch.float32)
    if normalize:
        eps = 1e-6
        y = y / (y[:, -1:, :] + eps) * scale
        x = x / (x[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x[:, :, :, None] / dim_t, y[:, :, :, None] / dim_t
    px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def detr_heads(sd, hs, prefix=""):
    """DETR.forward after the transformer (detr.py:449-452): class_embed, bbox_embed (MLP 256-256-256-4, ReLU between) +
    sigmoid.  hs [layers, B, Q, C] -> (logits [layers, B, Q, classes + 1], boxes [layers, B, Q, 4])"""
    logits = F.linear(hs, sd[prefix + "class_embed.weight"], sd[prefix + "class_embed.bias"])
    x, n = hs, 1 + max(int(k[len(prefix + "bbox_embed.layers."):].split(".")[0]) for k in sd if k.startswith(prefix + "bbox_embed.layers."))
    for i in range(n):
        x = F.linear(x, sd[f"{prefix}bbox_embed.layers.{i}.weight"], sd[f"{prefix}bbox_embed.layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return logits, x.sigmoid()


def detr_after_backbone(sd, feat, mask, pos, nhead=8, pre=False, prefix="", quant=None, force=None, query_embed_layers=None):
    """input_proj (1x1 conv) -> transformer -> heads: everything of DETR.forward (detr.py:444-452) behind the backbone.
    feat [B, 2048, H, W] (the backbone's last map), mask bool [B, H, W], pos [B, 256, H, W]"""
    q = quant if quant is not None else (lambda t: t)
    src = F.conv2d(q(feat), sd[prefix + "input_proj.weight"], sd[prefix + "input_proj.bias"])
    src = _force(src, force, "src", q)
    hs, memory = transformer(sd, src, mask, sd[prefix + "query_embed.weight"], pos, nhead, pre, prefix + "transformer.", q, force,
                             query_embed_layers)
    logits, boxes = detr_heads(sd, hs, prefix)
    return dict(src=src, hs=hs, memory=memory, logits=logits, boxes=boxes)
