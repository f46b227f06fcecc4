"""DETR detection module on the MI355X kernels — drop-in for `DETR` and `MLP` of yolov7/modeling/meta_arch/detr.py:
282-294, 406-472 (config 4), with the reference's attribute names / state_dict keys (transformer.*, class_embed,
bbox_embed.layers.N, query_embed, input_proj, backbone.*).

forward(samples) follows detr.py:426-462: features, pos = backbone(samples); src, mask = features[-1].decompose();
hs = transformer(input_proj(src), mask, query_embed.weight, pos[-1])[0]; class_embed / bbox_embed(+sigmoid) on every
decoder level; {'pred_logits', 'pred_boxes', 'aux_outputs'}.  input_proj (1x1 conv with bias), class_embed, the MLP and
the transformer all run on the implicit-GEMM / attention / LayerNorm kernels; logits and boxes are returned in fp32 as
SetCriterion consumes them.  The backbone is whatever module the caller passes (the reference wraps detectron2's
ResNet-50, an un-vendored dependency: SURVEY a36, not rebuilt here).
"""
import torch
from torch import nn

from .. import _lib as L
from .transformer import _LinearFn, _tok


class _SigmoidF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float().contiguous()
        y = torch.empty_like(x)
        L.check(L.lib().mi_sigmoid_f32(x.data_ptr(), None, y.data_ptr(), None, x.numel(), L.stream_ptr()), "mi_sigmoid_f32")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.float().contiguous()
        dx = torch.empty_like(y)
        L.check(L.lib().mi_sigmoid_f32(None, dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), L.stream_ptr()),
                "mi_sigmoid_f32")
        return dx


def _linear(x, lin, relu=False):
    """nn.Linear applied to the last dimension of a bf16 device tensor through the conv kernels (relu: in the epilogue)"""
    shp = x.shape
    y = _LinearFn.apply(_tok(x).reshape(-1, shp[-1]), lin.weight, lin.bias, relu)
    return y.reshape(*shp[:-1], lin.out_features)


class MLP(nn.Module):
    """detr.py:282-294: num_layers nn.Linear with ReLU in between"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = _linear(x, layer, relu=i < self.num_layers - 1)
        return x


class DETR(nn.Module):
    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False):
        super().__init__()
        self.num_queries = num_queries
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.class_embed = nn.Linear(hidden_dim, num_classes + 1)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.input_proj = nn.Conv2d(backbone.num_channels, hidden_dim, kernel_size=1)
        self.backbone = backbone
        self.aux_loss = aux_loss

    def forward(self, samples):
        features, pos = self.backbone(samples)
        src, mask = features[-1].decompose()
        assert mask is not None
        if not src.is_cuda:
            raise L.MI355Error("DETR: the MI355X path needs device tensors (no CPU fallback)")
        B, Cc, H, W = src.shape
        # input_proj: 1x1 conv with bias == a linear over the pixel rows
        # (the backbone hands over NCHW-shaped channels_last memory: its NHWC view IS the token matrix - forming the NCHW copy
        #  first and the NHWC one from it were two 18 MB passes per step)
        tok = _tok(src.permute(0, 2, 3, 1)).reshape(B * H * W, Cc)
        w = self.input_proj.weight.view(self.input_proj.out_channels, Cc)
        proj = _LinearFn.apply(tok, w, self.input_proj.bias)
        proj = proj.reshape(B, H, W, -1).permute(0, 3, 1, 2)
        hs = self.transformer(proj, mask, self.query_embed.weight, pos[-1])[0]
        outputs_class = _linear(hs, self.class_embed).float()
        outputs_coord = _SigmoidF32.apply(self.bbox_embed(hs))
        out = {"pred_logits": outputs_class[-1], "pred_boxes": outputs_coord[-1]}
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b} for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]
        return out
