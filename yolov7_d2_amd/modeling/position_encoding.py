"""PositionEmbeddingSine — drop-in for yolov7/modeling/backbone/detr_backbone.py:309-375 (DETR, config 4).

Same constructor (num_pos_feats, temperature, normalize, scale, centered) and forward contract: takes a NestedTensor-like
object (`.tensors` [B,C,H,W], `.mask` [B,H,W] bool, True = padding) - or (tensors, mask) - and returns the fp32
[B, 2*num_pos_feats, H, W] encoding, computed by one launch of mi_pos_embed_sine (the reference builds it from two
cumsums, a pow, 4 strided slices, 2 stacks and a cat).
"""
import math

import torch
from torch import nn

from .. import _lib as L


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None, centered=False):
        super().__init__()
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.scale = 2 * math.pi if scale is None else scale
        self.centered = centered

    @torch.no_grad()
    def forward(self, tensor_list, mask=None):
        if mask is None:
            mask = tensor_list.mask
        assert mask is not None
        if not mask.is_cuda:
            raise L.MI355Error("PositionEmbeddingSine: the MI355X path needs device tensors (no CPU fallback)")
        B, H, W = mask.shape
        m8 = mask.to(torch.uint8).contiguous()
        out = torch.empty(B, 2 * self.num_pos_feats, H, W, dtype=torch.float32, device=mask.device)
        L.check(L.lib().mi_pos_embed_sine(m8.data_ptr(), B, H, W, self.num_pos_feats, float(self.temperature),
                                          int(bool(self.normalize)), float(self.scale), int(bool(self.centered)),
                                          out.data_ptr(), L.stream_ptr()), "mi_pos_embed_sine")
        return out
