"""ORACLE support: seeded input generators shared by gen_golden.py (reference side, needs /root/reference) and the
tests (which must not import gen_golden.py's reference loader on the GPU box)."""
import torch


def synth_box_pairs(n, seed):
    g = torch.Generator().manual_seed(seed)
    c = 50 + 400 * torch.rand(n, 2, generator=g)
    wh = 8 + 120 * torch.rand(n, 2, generator=g)
    pred = torch.cat([c, wh], 1)
    tgt = torch.cat([c + 30 * (torch.rand(n, 2, generator=g) - 0.5), wh * (0.6 + 0.8 * torch.rand(n, 2, generator=g))], 1)
    tgt[: n // 8] = torch.cat([c[: n // 8] + 300, wh[: n // 8]], 1)     # some disjoint pairs
    return pred, tgt


def synth_encoder_case(L=70, B=2, E=256, seed=61):
    """seeded inputs of one DETR encoder layer call: tokens, positional embedding, key-padding mask, output gradient"""
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(torch.bfloat16).float()
    src = bf(torch.randn(L, B, E, generator=g))
    pos = bf(0.5 * torch.randn(L, B, E, generator=g))
    go = bf(torch.randn(L, B, E, generator=g))
    mask = torch.zeros(B, L, dtype=torch.bool)
    mask[1, L - L // 4:] = True
    return src, pos, mask, go


def encoder_state_dict(E=256, ff=2048, seed=62):
    """seeded parameters with the reference's key names (TransformerEncoderLayer state_dict)"""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=0.05: sc * torch.randn(*s, generator=g)
    return {"self_attn.in_proj_weight": r(3 * E, E), "self_attn.in_proj_bias": r(3 * E, sc=0.02),
            "self_attn.out_proj.weight": r(E, E), "self_attn.out_proj.bias": r(E, sc=0.02),
            "linear1.weight": r(ff, E), "linear1.bias": r(ff, sc=0.02), "linear2.weight": r(E, ff, sc=0.02),
            "linear2.bias": r(E, sc=0.02), "norm1.weight": 1 + r(E, sc=0.1), "norm1.bias": r(E, sc=0.1),
            "norm2.weight": 1 + r(E, sc=0.1), "norm2.bias": r(E, sc=0.1)}


def synth_transformer_case(B=2, E=256, H=6, W=10, Q=40, seed=71):
    """seeded inputs of one DETR Transformer call (detr_backbone.py:56): feature map, padding mask, query embedding,
    positional embedding, and the gradient fed back into hs"""
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(torch.bfloat16).float()
    src = bf(torch.randn(B, E, H, W, generator=g))
    pos = bf(0.5 * torch.randn(B, E, H, W, generator=g))
    qe = bf(torch.randn(Q, E, generator=g))
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    mask[1, :, W - 3:] = True
    mask[1, H - 1:, :] = True
    return src, mask, qe, pos


def seeded_state_dict(module, seed=72):
    """seeded parameters for any module, keyed and shaped by the module's own state_dict (sorted key order):
    LayerNorm weights around 1, matrices N(0, 0.05) (N(0, 0.02) when the fan-in is > 1024), biases N(0, 0.02)"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(module.state_dict().keys()):
        shp = module.state_dict()[k].shape
        if "norm" in k and k.endswith("weight"):
            out[k] = 1 + 0.1 * torch.randn(*shp, generator=g)
        elif len(shp) == 2:
            out[k] = (0.02 if shp[1] > 1024 else 0.05) * torch.randn(*shp, generator=g)
        else:
            out[k] = 0.02 * torch.randn(*shp, generator=g)
    return out


def synth_detr_case(B=2, C=64, H=6, W=10, seed=101):
    """seeded backbone feature map + padding mask of one DETR forward, and the seeds of the output gradients"""
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(torch.bfloat16).float()
    feat = bf(torch.randn(B, C, H, W, generator=g))
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    mask[1, :, W - 3:] = True
    mask[1, H - 1:, :] = True
    return feat, mask


class StubBackbone(torch.nn.Module):
    """stands in for the (un-vendored) detectron2 ResNet-50 wrapper: hands a fixed feature map and its positional
    encoding to DETR in the (features, pos) structure detr.py:441-443 expects"""

    def __init__(self, feat, mask, pos, nested_cls):
        super().__init__()
        self.num_channels = feat.shape[1]
        self.feat, self.mask, self.pos, self.nested_cls = feat, mask, pos, nested_cls

    def forward(self, samples):
        return [self.nested_cls(self.feat, self.mask)], [self.pos]


class SimpleNested:
    def __init__(self, tensors, mask):
        self.tensors, self.mask = tensors, mask

    def decompose(self):
        return self.tensors, self.mask


def seeded_tensor_dict(shapes, seed=72):
    """seeded values for a {key: shape} table (sorted key order), the rule of seeded_state_dict plus detectron2 ResNet
    entries: conv weights N(0, sqrt(2 / fan_out)), frozen-norm weight in [0.5, 1.5], running_var in [0.5, 1.5]"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        if k.endswith("empty_weight"):      # SetCriterion's class-weight buffer (ones, eos_coef last): a constant, not a weight
            continue
        if k.endswith("running_var") or (".norm.weight" in k and ".backbone." in k):
            out[k] = 0.5 + torch.rand(*shp, generator=g)
        elif len(shp) == 4 and ".backbone." in k:
            out[k] = torch.randn(*shp, generator=g) * (2.0 / (shp[0] * shp[2] * shp[3])) ** 0.5
        elif "norm" in k and k.endswith("weight"):
            out[k] = 1 + 0.1 * torch.randn(*shp, generator=g)
        elif len(shp) == 2:
            out[k] = (0.02 if shp[1] > 1024 else 0.05) * torch.randn(*shp, generator=g)
        elif len(shp) == 0:
            out[k] = torch.zeros(shp, dtype=torch.long)
        else:
            out[k] = 0.02 * torch.randn(*shp, generator=g)
    return out


def synth_detr_batch(seed=201, sizes=((160, 200), (128, 224)), ncls=80):
    """batched_inputs of the Detr meta-arch: float images 0..255 of DIFFERENT sizes (padding masks), 2-4 ground truths
    each as (XYXY absolute boxes, classes)"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for (h, w) in sizes:
        img = torch.randint(0, 256, (3, h, w), generator=g).float()
        n = int(torch.randint(2, 5, (1,), generator=g))
        wh = 20 + torch.rand(n, 2, generator=g) * torch.tensor([w * 0.5, h * 0.5])
        xy = torch.rand(n, 2, generator=g) * (torch.tensor([w, h]) - wh)
        boxes = torch.cat([xy, xy + wh], 1)
        cls = torch.randint(0, ncls, (n,), generator=g)
        out.append(dict(image=img, boxes=boxes, classes=cls, size=(h, w)))
    return out


def synth_sparseinst_case(seed=301, B=2, H=128, W=160):
    """seeded ResNet features (res3 / res4 / res5 of a padded B x 3 x H x W batch) and bitmask targets (random rectangles
    and ellipses on images SMALLER than the padded batch, so the zero padding of nested_masks_from_list is exercised)"""
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(torch.bfloat16).float()
    feats = {n: bf(torch.relu(torch.randn(B, c, H // s, W // s, generator=g))) for n, c, s in
             (("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    targets = []
    for b in range(B):
        h, w = (H, W) if b == 0 else (H - 24, W - 32)
        n = 3 if b == 0 else 2
        masks = torch.zeros(n, h, w)
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        for k in range(n):
            cy, cx = float(torch.rand(1, generator=g)) * h, float(torch.rand(1, generator=g)) * w
            ry, rx = 10 + float(torch.rand(1, generator=g)) * h * 0.3, 10 + float(torch.rand(1, generator=g)) * w * 0.3
            if k % 2 == 0:
                masks[k] = ((yy - cy).abs() < ry) & ((xx - cx).abs() < rx)
            else:
                masks[k] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) < 1
        targets.append(dict(labels=torch.randint(0, 80, (n,), generator=g), masks=masks, size=(h, w)))
    return feats, targets, (H, W)


def sparseinst_spread(sd):
    """de-generate a seeded SparseInst state dict: with small random weights all 100 instance activation maps (and hence
    all predictions) are nearly identical and the Hungarian matching is decided by rounding noise.  Larger IAM / kernel /
    class weights make the instances differ, as a trained model's do."""
    sd = dict(sd)
    for k, f in (("decoder.inst_branch.iam_conv.weight", 40.0), ("decoder.inst_branch.mask_kernel.weight", 6.0),
                 ("decoder.inst_branch.cls_score.weight", 0.3), ("decoder.inst_branch.fc.weight", 0.5)):
        sd[k] = sd[k] * f
    sd["decoder.inst_branch.cls_score.bias"] = sd["decoder.inst_branch.cls_score.bias"] - 2.0
    return sd


def sparseinst_onnx_weights(shapes, seed=411):
    """seeded weights of the WHOLE SparseInst (backbone.* + encoder.* + decoder.*) for the export-mode golden: a randomly
    initialised ResNet shrinks its activations to ~0.04, after which all 100 instances predict the same thing and the
    top-k order is rounding noise - the lateral convolutions are scaled so that the encoder works at unit scale, the IAM /
    class / objectness weights so that the instances differ (scores 0.6 .. 0.7, half a dozen classes)"""
    sd = sparseinst_spread(seeded_tensor_dict(shapes, seed=seed))
    for k in list(sd):
        if k.startswith("encoder.fpn_laterals") and k.endswith("weight"):
            sd[k] = sd[k] * 25.0
    for k, f in (("decoder.inst_branch.iam_conv.weight", 20.0), ("decoder.inst_branch.cls_score.weight", 30.0),
                 ("decoder.inst_branch.objectness.weight", 30.0)):
        sd[k] = sd[k] * f
    return sd


def synth_sparseinst_images(B, H, W, seed):
    """float NHWC images in [0, 255] with structure (plane waves + a little noise): spatially varying features"""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    out = []
    for b in range(B):
        ph = torch.rand(6, generator=g) * 6.28
        base = 127 + 90 * torch.sin(xx / (5 + 3 * b) + ph[0]) * torch.cos(yy / 7.0 + ph[1])
        img = torch.stack([base, 127 + 80 * torch.sin(xx / 11 + yy / 5 + ph[2]), 127 + 70 * torch.cos(xx / 4 - yy / 9 + ph[3])], -1)
        out.append((img + torch.randn(H, W, 3, generator=g) * 8).clamp(0, 255))
    return torch.stack(out)


def detr_onnx_weights(shapes, seed=431):
    """seeded weights of the whole reference Detr for the export golden: input_proj scaled down (tokens of norm O(10^2), not
    10^4), the class head widened so that the arg-max class of a query is not a rounding-noise tie"""
    sd = seeded_tensor_dict(shapes, seed=seed)
    sd["detr.input_proj.weight"] = sd["detr.input_proj.weight"] * 1e-2
    sd["detr.class_embed.weight"] = sd["detr.class_embed.weight"] * 8.0
    # sharper attention and wider query embeddings: the queries differ a little (a random-init encoder still maps its six
    # tokens to nearly the same vector, so the 100 rows stay close to each other - every layer's arithmetic is in them all the
    # same, which is what the comparison needs)
    for k in list(sd):
        if k.endswith("in_proj_weight"):
            sd[k] = sd[k] * 4.0
    sd["detr.query_embed.weight"] = sd["detr.query_embed.weight"] * 30.0
    return sd


def synth_nms_case(n, ncls, seed, spread=14.0):
    """overlapping boxes in clusters (xyxy), scores, class ids (as float, the way the meta-archs pass them)"""
    g = torch.Generator().manual_seed(seed)
    nc = max(2, n // 6)
    centers = 40 + 500 * torch.rand(nc, 2, generator=g)
    which = torch.randint(0, nc, (n,), generator=g)
    c = centers[which] + spread * torch.randn(n, 2, generator=g)
    wh = 30 + 60 * torch.rand(n, 2, generator=g)
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    scores = 0.05 + 0.9 * torch.rand(n, generator=g)
    scores[: n // 10] = 0.0001 + 0.0008 * torch.rand(n // 10, generator=g)     # already below the score threshold (distinct:
    #                                                                          the order of tied scores is unspecified)
    idxs = torch.randint(0, ncls, (n,), generator=g).float()
    return boxes, scores, idxs


def synth_mask_case(n, H, W, ncls, seed):
    """n binary masks (overlapping blobs), descending scores, labels: the inputs of matrix_nms"""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    nc = max(2, n // 5)
    centers = torch.rand(nc, 2, generator=g) * torch.tensor([H, W]).float()
    which = torch.randint(0, nc, (n,), generator=g)
    c = centers[which] + 3 * torch.randn(n, 2, generator=g)
    r = 4 + 8 * torch.rand(n, generator=g)
    masks = (((yy[None] - c[:, 0, None, None]) ** 2 + (xx[None] - c[:, 1, None, None]) ** 2) < r[:, None, None] ** 2)
    masks[0, 0, 0] = True                # no empty mask (the reference would divide 0 / 0)
    masks = masks | (torch.arange(n)[:, None, None] == 10 ** 9)
    for k in range(n):
        if not masks[k].any():
            masks[k, int(c[k, 0].clamp(0, H - 1)), int(c[k, 1].clamp(0, W - 1))] = True
    scores = torch.sort(0.1 + 0.85 * torch.rand(n, generator=g), descending=True)[0]
    labels = torch.randint(0, ncls, (n,), generator=g)
    return labels, masks, masks.flatten(1).sum(1).float(), scores


def synth_yolov6_case(B=3, H=160, W=192, seed=91, nc=80):
    """raw head outputs per level [B, 1, h, w, 5 + nc] and NORMALISED targets [B, 30, 5] for ComputeLoss"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import yolox_oracle as O
    _, labels = O.synth_batch(B, H, W, seed=seed, max_gt=10, min_gt=4)
    labels[1] = 0.0
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    raw, anchors = O.synth_raw(B, hw, seed + 1, labels=labels)
    outs, a0 = [], 0
    for (h, w) in hw:
        outs.append(raw[:, a0:a0 + h * w].reshape(B, 1, h, w, 5 + nc).clone())
        a0 += h * w
    t = labels.clone()
    t[..., 1:5] = t[..., 1:5] / torch.tensor([W, H, W, H]).float()
    return outs, t, labels, raw, anchors


BIFPN_CASES = {"dense": dict(out_channels=64, num_bifpn=2, separable_conv=False),
               "separable": dict(out_channels=96, num_bifpn=1, separable_conv=True)}


def synth_bifpn_case(B=2, chans=(32, 64, 128), size=256, out_channels=64, seed=131):
    """seeded C3..C5 feature maps of a `size` x `size` image (strides 8 / 16 / 32) and the gradients fed into p3..p7"""
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(torch.bfloat16).float()
    feats = {f"res{i + 3}": bf(torch.randn(B, c, size // s, size // s, generator=g)) for i, (c, s) in enumerate(zip(chans, (8, 16, 32)))}
    gos = {f"p{l}": bf(torch.randn(B, out_channels, size >> l, size >> l, generator=g)) for l in range(3, 8)}
    return feats, gos


def bifpn_state_dict(module, seed=132):
    """seeded parameters keyed / shaped by the module's own state_dict (sorted key order): GroupNorm weights around 1,
    edge weights spread around 1 with one negative entry per 7 (the relu of fastattn), conv weights N(0, 1/sqrt(fan_in))"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for i, k in enumerate(sorted(module.state_dict().keys())):
        shp = module.state_dict()[k].shape
        if k.endswith("edge_weights"):
            w = 1.0 + 0.5 * torch.randn(*shp, generator=g)
            if i % 7 == 0:
                w[0] = -0.3
            out[k] = w
        elif ".bn." in k:
            out[k] = (1 + 0.1 * torch.randn(*shp, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(*shp, generator=g)
        elif len(shp) == 4:
            out[k] = torch.randn(*shp, generator=g) / (shp[1] * shp[2] * shp[3]) ** 0.5
        else:
            out[k] = 0.02 * torch.randn(*shp, generator=g)
    return out


def grad_signature(named_grads, nproj=8):
    """compact fingerprint of a set of gradients: per tensor its L2 norm and `nproj` projections on seeded +-1 vectors
    (seed = crc32 of the parameter name).  For Rademacher r: E[(r . d)^2] = |d|^2, so the mean squared difference of two
    tensors' projections estimates |g1 - g2|^2 - enough to bound a relative gradient error without shipping 160 MB"""
    import zlib
    import numpy as np
    sig = {}
    for name, g in named_grads:
        g = g.detach().float().reshape(-1).cpu()
        gen = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        proj = []
        for _ in range(nproj):
            r = torch.randint(0, 2, (g.numel(),), generator=gen, dtype=torch.int8).float() * 2 - 1
            proj.append(float((r.double() * g.double()).sum()))
        sig[name] = np.array([float(g.double().norm())] + proj, dtype=np.float64)
    return sig
