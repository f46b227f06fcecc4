"""TEST INFRASTRUCTURE (never imported by the product): fp32 torch restatement of SparseInst's network behind the backbone -
`InstanceContextEncoder` + `PyramidPoolingModule` + `MyAdaptiveAvgPool2d` (yolov7/modeling/transcoders/encoder_sparseinst.py:
18-127) and `GroupIAMDecoder` = `BaseIAMDecoder` with `GroupInstanceBranch` + `MaskBranch`
(transcoders/decoder_sparseinst.py:18-255) - as FUNCTIONS over a state_dict with the reference's keys ("encoder.*",
"decoder.*"), with the two hooks of the forward-pinned parity tests (see oracle/detr_net_oracle.py):

  quant   storage-rounding emulation where the product stores a tensor (bf16 conv outputs)
  force   {site: tensor}: pins the VALUE of that intermediate.  Sites: every convolution output, named by its module path
          ("encoder.fpn_laterals.0", "encoder.ppm.stages.2.1", "decoder.inst_branch.inst_convs.4", ...; PRE-activation where
          the product's hook sees the pre-activation value, else the stored post-ReLU value - relu of either is the same).

Pinned: tests/test_oracle_golden.py::test_sparseinst_net_oracle_against_reference_golden holds it to the golden the
REFERENCE'S OWN encoder + decoder produced by path (tests/golden/sparseinst.npz)."""
import math

import torch
import torch.nn.functional as F


def _q(quant):
    return quant if quant is not None else (lambda t: t)


def _conv(sd, p, x, pad, q, force, groups=1):
    y = F.conv2d(q(x), sd[p + ".weight"], sd[p + ".bias"], 1, pad, 1, groups)
    if force is not None and p in force:
        y = y + (force[p].to(y.dtype) - y).detach()
    return q(y)


def _my_adaptive_avg_pool(x, sz):
    """MyAdaptiveAvgPool2d (encoder_sparseinst.py:18-40): avg_pool2d with kernel ceil(size / sz), NOT nn.AdaptiveAvgPool2d"""
    kh, kw = math.ceil(x.shape[2] / sz), math.ceil(x.shape[3] / sz)
    return F.avg_pool2d(x, kernel_size=(kh, kw), ceil_mode=False)


def encoder(sd, feats, in_features=("res3", "res4", "res5"), prefix="encoder.", quant=None, force=None, sizes=(1, 2, 3, 6)):
    """InstanceContextEncoder.forward (encoder_sparseinst.py:107-127)"""
    q = _q(quant)
    fs = [feats[f] for f in in_features][::-1]
    x = _conv(sd, prefix + "fpn_laterals.0", fs[0], 0, q, force)
    # PyramidPoolingModule.forward (:55-68)
    h, w = x.shape[2:]
    priors = [q(F.interpolate(F.relu(_conv(sd, f"{prefix}ppm.stages.{i}.1", _my_adaptive_avg_pool(x, s), 0, q, force)), size=(h, w),
                              mode="bilinear", align_corners=False)) for i, s in enumerate(sizes)] + [x]
    prev = F.relu(_conv(sd, prefix + "ppm.bottleneck", torch.cat(priors, 1), 0, q, force))
    outs = [_conv(sd, prefix + "fpn_outputs.0", prev, 1, q, force)]
    for k, f in enumerate(fs[1:], start=1):
        lat = _conv(sd, f"{prefix}fpn_laterals.{k}", f, 0, q, force)
        prev = q(lat + F.interpolate(prev, scale_factor=2.0, mode="nearest"))
        outs.insert(0, _conv(sd, f"{prefix}fpn_outputs.{k}", prev, 1, q, force))
    size = outs[0].shape[2:]
    fused = [outs[0]] + [q(F.interpolate(o, size, mode="bilinear", align_corners=False)) for o in outs[1:]]
    return _conv(sd, prefix + "fusion", torch.cat(fused, 1), 0, q, force)


def _stack(sd, p, x, n, q, force):
    for i in range(n):
        x = F.relu(_conv(sd, f"{p}.{2 * i}", x, 1, q, force))
    return x


def decoder(sd, features, groups=4, scale_factor=2.0, prefix="decoder.", quant=None, force=None):
    """GroupIAMDecoder.forward = BaseIAMDecoder.forward (decoder_sparseinst.py:118-161) with GroupInstanceBranch.forward
    (:203-233) and MaskBranch.forward (:99-101)"""
    q = _q(quant)
    B, _, H, W = features.shape
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    coords = torch.stack([xs, ys])[None].expand(B, 2, H, W).to(features)
    f = torch.cat([coords, features], 1)
    ib = prefix + "inst_branch."
    n_inst = 1 + max(int(k[len(ib + "inst_convs."):].split(".")[0]) for k in sd if k.startswith(ib + "inst_convs.")) // 2
    x = _stack(sd, ib + "inst_convs", f, n_inst, q, force)
    iam = _conv(sd, ib + "iam_conv", x, 1, q, force, groups=groups)
    prob = q(iam.sigmoid())
    N, Cc = prob.shape[1], x.shape[1]
    prob = prob.view(B, N, -1)
    inst = torch.bmm(prob, x.view(B, Cc, -1).permute(0, 2, 1))
    inst = inst / prob.sum(-1).clamp(min=1e-6, max=1e5)[:, :, None]
    d4 = N // 4
    inst = inst.reshape(B, 4, d4, -1).transpose(1, 2).reshape(B, d4, -1)
    lin = lambda name, t: F.linear(t, sd[ib + name + ".weight"], sd[ib + name + ".bias"])
    inst = F.relu(lin("fc", inst))
    logits, kernel, scores = lin("cls_score", inst), lin("mask_kernel", inst), lin("objectness", inst)
    mb = prefix + "mask_branch."
    n_mask = 1 + max(int(k[len(mb + "mask_convs."):].split(".")[0]) for k in sd if k.startswith(mb + "mask_convs.")) // 2
    mf = _conv(sd, mb + "projection", _stack(sd, mb + "mask_convs", f, n_mask, q, force), 0, q, force)
    masks = torch.bmm(kernel, mf.view(B, mf.shape[1], H * W)).view(B, d4, H, W)
    masks = F.interpolate(q(masks), scale_factor=scale_factor, mode="bilinear", align_corners=False)
    return dict(pred_logits=logits, pred_masks=masks, pred_scores=scores, iam=iam)
