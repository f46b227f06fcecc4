"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the reference's YOLOX path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (yolov7_d2_amd) never does.  Everything here runs on the torch CPU backend in fp32 and
is a functional re-statement (over a plain state_dict with the reference's key names) of:

  BaseConv / Bottleneck / CSPLayer / SPPBottleneck / Focus   yolov7/modeling/backbone/layers/wrappers.py:60-220
  CSPDarknet                                                  yolov7/modeling/backbone/darknetx.py:103-177
  YOLOPAFPN                                                   yolov7/modeling/neck/yolo_pafpn.py:79-114
  YOLOXHead forward / decode / get_losses / SimOTA            yolov7/modeling/head/yolox_head.py:151-669
  bboxes_iou, IOUloss, postprocess                            yolov7/utils/boxes.py:57-81,125-210
  YOLOX.preprocess_image label packing                        yolov7/modeling/meta_arch/yolox.py:139-157
  torchvision.ops.nms / batched_nms (un-vendored dependency; semantics restated from torchvision 0.12,
  the version paired with the torch 1.11 the reference's readme.md:174 recommends)

Pinning: the reference ships no tests or golden vectors.  This restatement is pinned
against the reference's own source executed here by path (oracle/ref_loader.py, oracle/gen_golden.py)
and the resulting vectors are committed under tests/golden/.  The NMS arithmetic has no in-repo
reference implementation at all: *parity unpinned* for NMS beyond torchvision's documented semantics.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3      # yolox.py:85-90
BN_MOM = 0.03


# ----------------------------------------------------------------------------- network
class Net:
    """functional YOLOX over a state_dict `sd` (reference key names). training=True -> batch-stat BN
    (running stats in `sd` are updated in place, like nn.BatchNorm2d)."""

    def __init__(self, sd, depth=0.33, width=0.5, num_classes=80, training=True, quant=None, force=None, depthwise=False):
        self.sd, self.depth, self.width, self.nc, self.training = sd, depth, width, num_classes, training
        # MODEL.DARKNET.DEPTH_WISE (darknetx.py:113,210): the 3x3 convs of the BACKBONE become DWConv (the meta-arch
        # builds neck and head with depthwise=False, yolox.py:60-83)
        self.depthwise = depthwise
        # quant: optional callable emulating the product's storage rounding (e.g. bf16) after each op
        self.q = quant if quant is not None else (lambda t: t)
        self.taps = {}
        # force (teacher forcing, whole-step parity tests): {"<layer>.y": tensor} - the conv output of that layer as
        # ANOTHER implementation stored it.  The oracle's own conv result is compared with it (force_err, relative L2)
        # and then replaced by it in value (the gradient still flows through the oracle's conv), so that every layer
        # is checked on identical inputs and rounding differences cannot compound through the depth of the network:
        # a bf16-storage forward pass decorrelates from any other one to ~1 % at the head output, which the train-mode
        # BatchNorm backward amplifies to 10-60 % in the weight gradients (tests/test_oracle_golden.py::
        # test_bf16_storage_noise_floor) - no end-to-end comparison can be tighter than that without forcing.
        self.force = force
        self.force_err = {}

    def dw_conv(self, p, x, k, s, res=None):
        """DWConv (wrappers.py:86-102): depthwise BaseConv(k, s, groups=C) then pointwise BaseConv(1, 1)"""
        return self.base_conv(p + ".pconv", self.base_conv(p + ".dconv", x, k, s, groups=x.shape[1]), 1, 1, res=res)

    def conv3(self, p, x, s, res=None):
        """the 3x3 convs darknetx.py / wrappers.Bottleneck switch to DWConv under depthwise"""
        return self.dw_conv(p, x, 3, s, res=res) if self.depthwise else self.base_conv(p, x, 3, s, res=res)

    def base_conv(self, p, x, k, s, res=None, groups=1):
        sd = self.sd
        y = F.conv2d(self.q(x), self.q(sd[p + ".conv.weight"]), None, stride=s, padding=(k - 1) // 2, groups=groups)
        self.taps[p + ".y"] = y
        y = self.q(y)
        if self.force is not None and p + ".y" in self.force:
            f = self.force[p + ".y"]
            if f.dim() == 1:       # flat NHWC image of the tensor (the product's layout)
                f = f[: y.numel()].view(y.shape[0], y.shape[2], y.shape[3], y.shape[1]).permute(0, 3, 1, 2)
            f = f.to(y.dtype)
            self.force_err[p] = float((y.detach() - f).norm() / (f.norm() + 1e-30))
            y = y + (f - y).detach()
        if self.training:
            y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                             sd[p + ".bn.bias"], True, BN_MOM, BN_EPS)
            if p + ".bn.num_batches_tracked" in sd:
                sd[p + ".bn.num_batches_tracked"] += 1
        else:
            y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                             sd[p + ".bn.bias"], False, BN_MOM, BN_EPS)
        y = F.silu(y)
        if res is not None:
            y = y + res
        y = self.q(y)
        self.taps[p + ".out"] = y
        return y

    def bottleneck(self, p, x, shortcut):
        h = self.base_conv(p + ".conv1", x, 1, 1)
        if p.startswith("backbone"):
            return self.conv3(p + ".conv2", h, 1, res=x if shortcut else None)
        return self.base_conv(p + ".conv2", h, 3, 1, res=x if shortcut else None)

    def csp(self, p, x, n, shortcut):
        x1 = self.base_conv(p + ".conv1", x, 1, 1)
        x2 = self.base_conv(p + ".conv2", x, 1, 1)
        for i in range(n):
            x1 = self.bottleneck(f"{p}.m.{i}", x1, shortcut)
        return self.base_conv(p + ".conv3", torch.cat((x1, x2), 1), 1, 1)

    def spp(self, p, x):
        x = self.base_conv(p + ".conv1", x, 1, 1)
        x = torch.cat([x] + [F.max_pool2d(x, ks, 1, ks // 2) for ks in (5, 9, 13)], 1)
        return self.base_conv(p + ".conv2", x, 1, 1)

    def focus(self, p, x):
        tl, tr = x[..., ::2, ::2], x[..., ::2, 1::2]
        bl, br = x[..., 1::2, ::2], x[..., 1::2, 1::2]
        return self.base_conv(p + ".conv", torch.cat((tl, bl, tr, br), 1), 3, 1)

    def backbone(self, x, p="backbone"):
        bd = max(round(self.depth * 3), 1)
        x = self.focus(p + ".stem", x)
        x = self.csp(p + ".dark2.1", self.conv3(p + ".dark2.0", x, 2), bd, True)
        d3 = self.csp(p + ".dark3.1", self.conv3(p + ".dark3.0", x, 2), bd * 3, True)
        d4 = self.csp(p + ".dark4.1", self.conv3(p + ".dark4.0", d3, 2), bd * 3, True)
        x = self.spp(p + ".dark5.1", self.conv3(p + ".dark5.0", d4, 2))
        d5 = self.csp(p + ".dark5.2", x, bd, False)
        return {"dark3": d3, "dark4": d4, "dark5": d5}

    def neck(self, feats, p="neck"):
        n = round(3 * self.depth)
        x2, x1, x0 = feats["dark3"], feats["dark4"], feats["dark5"]
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
        fpn_out0 = self.base_conv(p + ".lateral_conv0", x0, 1, 1)
        f_out0 = self.csp(p + ".C3_p4", torch.cat([up(fpn_out0), x1], 1), n, False)
        fpn_out1 = self.base_conv(p + ".reduce_conv1", f_out0, 1, 1)
        pan_out2 = self.csp(p + ".C3_p3", torch.cat([up(fpn_out1), x2], 1), n, False)
        p_out1 = self.base_conv(p + ".bu_conv2", pan_out2, 3, 2)
        pan_out1 = self.csp(p + ".C3_n3", torch.cat([p_out1, fpn_out1], 1), n, False)
        p_out0 = self.base_conv(p + ".bu_conv1", pan_out1, 3, 2)
        pan_out0 = self.csp(p + ".C3_n4", torch.cat([p_out0, fpn_out0], 1), n, False)
        return (pan_out2, pan_out1, pan_out0)

    def head_raw(self, fpn, p="head"):
        """per level raw maps [B, 5+nc, h, w] = cat(reg, obj, cls) (yolox_head.py:160-175)"""
        sd, outs = self.sd, []
        for k, x in enumerate(fpn):
            t = self.base_conv(f"{p}.stems.{k}", x, 1, 1)
            c = self.base_conv(f"{p}.cls_convs.{k}.1", self.base_conv(f"{p}.cls_convs.{k}.0", t, 3, 1), 3, 1)
            r = self.base_conv(f"{p}.reg_convs.{k}.1", self.base_conv(f"{p}.reg_convs.{k}.0", t, 3, 1), 3, 1)
            pc = lambda name, z: F.conv2d(self.q(z), self.q(sd[f"{p}.{name}.{k}.weight"]), sd[f"{p}.{name}.{k}.bias"])
            outs.append(torch.cat([pc("reg_preds", r), pc("obj_preds", r), pc("cls_preds", c)], 1))
        return outs

    def forward_raw(self, images):
        """-> raw [B, A, 5+nc] (undecoded), hw list"""
        raw = self.head_raw(self.neck(self.backbone(images)))
        hw = [tuple(o.shape[-2:]) for o in raw]
        flat = torch.cat([o.flatten(2).permute(0, 2, 1) for o in raw], 1)
        return flat, hw


def make_anchors(hw, strides=(8, 16, 32)):
    """(grid_x, grid_y, stride) per anchor, anchor = y*w + x per level (yolox_head.py:233-241)"""
    g = []
    for (h, w), s in zip(hw, strides):
        yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        g.append(torch.stack((xv.reshape(-1).float(), yv.reshape(-1).float(), torch.full((h * w,), float(s))), 1))
    return torch.cat(g, 0)


def decode(raw, anchors):
    """yolox_head.py:243-244 — returns decoded [B,A,5+nc] (xy, wh decoded; logits untouched)"""
    out = raw.clone()
    g, s = anchors[:, :2], anchors[:, 2:3]
    out[..., :2] = (raw[..., :2] + g) * s
    out[..., 2:4] = torch.exp(raw[..., 2:4]) * s
    return out


def decode_eval(raw, anchors):
    """eval head (yolox_head.py:197-224,247-272): sigmoid(obj, cls) then decode"""
    out = raw.clone()
    out[..., 4:] = torch.sigmoid(raw[..., 4:])
    g, s = anchors[:, :2], anchors[:, 2:3]
    out[..., :2] = (raw[..., :2] + g) * s
    out[..., 2:4] = torch.exp(raw[..., 2:4]) * s
    return out


# ----------------------------------------------------------------------------- boxes
def pairwise_iou_cxcywh(a, b):
    """bboxes_iou(a, b, xyxy=False), boxes.py:66-81 (no eps)"""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[:, :2] - b[:, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[:, :2] + b[:, 2:] / 2)
    area_a, area_b = torch.prod(a[:, 2:], 1), torch.prod(b[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=2)
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)


def iou_loss(pred, target):
    """IOUloss(reduction='none', loss_type='iou'), boxes.py:131-151"""
    tl = torch.max(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    br = torch.min(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    area_p, area_g = torch.prod(pred[:, 2:], 1), torch.prod(target[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=1)
    area_i = torch.prod(br - tl, 1) * en
    iou = area_i / (area_p + area_g - area_i + 1e-16)
    return 1 - iou ** 2


# ----------------------------------------------------------------------------- SimOTA
@torch.no_grad()
def simota_image(gt, gcls, bbox, obj_logit, cls_logit, anchors, num_classes, center_radius=2.5, cls_weight=1.0,
                 iou_weight=3.0):
    """one image. gt [G,4] cxcywh, gcls [G]; bbox [A,4] decoded; returns dict with the reference's
    intermediate and final assignment tensors (yolox_head.py:450-669).  The keyword arguments are the constants the
    YOLOv6 head's copy of this code makes configurable (yolov6_head.py:320-337, 597-754)."""
    G, A = gt.shape[0], bbox.shape[0]
    s = anchors[:, 2]
    xc = (anchors[:, 0] * s + 0.5 * s)[None].repeat(G, 1)
    yc = (anchors[:, 1] * s + 0.5 * s)[None].repeat(G, 1)
    l_, r_ = (gt[:, 0] - 0.5 * gt[:, 2])[:, None], (gt[:, 0] + 0.5 * gt[:, 2])[:, None]
    t_, b_ = (gt[:, 1] - 0.5 * gt[:, 3])[:, None], (gt[:, 1] + 0.5 * gt[:, 3])[:, None]
    in_box = torch.stack([xc - l_, yc - t_, r_ - xc, b_ - yc], 2).min(-1).values > 0.0
    rad = center_radius * s[None]
    cl, cr = gt[:, 0:1] - rad, gt[:, 0:1] + rad
    ct, cb = gt[:, 1:2] - rad, gt[:, 1:2] + rad
    in_ctr = torch.stack([xc - cl, yc - ct, cr - xc, cb - yc], 2).min(-1).values > 0.0
    cand = (in_box.sum(0) > 0) | (in_ctr.sum(0) > 0)
    both = in_box[:, cand] & in_ctr[:, cand]
    iou = pairwise_iou_cxcywh(gt, bbox[cand])
    iou_cost = -torch.log(iou + 1e-8)
    p = (cls_logit[cand].float().sigmoid() * obj_logit[cand].float().sigmoid()[:, None]).sqrt()   # [A', nc]
    onehot = F.one_hot(gcls.to(torch.int64), num_classes).float()                                   # [G, nc]
    cls_cost = F.binary_cross_entropy(p[None].repeat(G, 1, 1), onehot[:, None].repeat(1, p.shape[0], 1),
                                      reduction="none").sum(-1)
    cost = cls_weight * cls_cost + iou_weight * iou_cost + 100000.0 * (~both)
    # dynamic-k
    M = torch.zeros_like(cost)
    nk = min(10, iou.size(1))
    topk, _ = torch.topk(iou, nk, dim=1)
    ks = torch.clamp(topk.sum(1).int(), min=1)
    for g in range(G):
        _, pos = torch.topk(cost[g], k=int(ks[g]), largest=False)
        M[g][pos] = 1.0
    multi = M.sum(0) > 1
    if multi.sum() > 0:
        arg = torch.min(cost[:, multi], dim=0)[1]
        M[:, multi] *= 0.0
        M[arg, multi] = 1.0
    fg_in = M.sum(0) > 0.0
    fg = cand.clone()
    fg[cand] = fg_in
    matched = M[:, fg_in].argmax(0)
    return dict(cand=cand, cost=cost, iou=iou, ks=ks, fg=fg, matched_gt=matched, matched_cls=gcls[matched],
                matched_iou=(M * iou).sum(0)[fg_in], num_fg=int(fg_in.sum()))


def l1_target(gt, stride, x_shifts, y_shifts, eps=1e-8):
    """get_l1_target (yolox_head.py:443-448): gt [n,4] cxcywh in pixels, the matched anchors' stride / grid position"""
    return torch.stack([gt[:, 0] / stride - x_shifts, gt[:, 1] / stride - y_shifts,
                        torch.log(gt[:, 2] / stride + eps), torch.log(gt[:, 3] / stride + eps)], 1)


def iou_loss_v6(pred, target, iou_type="ciou", eps=1e-7):
    """IOUlossV6 (utils/boxes.py:666-752), box_format "xywh", reduction "none": pred / target [n,4] (cx,cy,w,h)"""
    px1, px2 = pred[:, 0] - pred[:, 2] / 2, pred[:, 0] + pred[:, 2] / 2
    py1, py2 = pred[:, 1] - pred[:, 3] / 2, pred[:, 1] + pred[:, 3] / 2
    tx1, tx2 = target[:, 0] - target[:, 2] / 2, target[:, 0] + target[:, 2] / 2
    ty1, ty2 = target[:, 1] - target[:, 3] / 2, target[:, 1] + target[:, 3] / 2
    inter = (torch.min(px2, tx2) - torch.max(px1, tx1)).clamp(0) * (torch.min(py2, ty2) - torch.max(py1, ty1)).clamp(0)
    w1, h1 = px2 - px1, py2 - py1 + eps
    w2, h2 = tx2 - tx1, ty2 - ty1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(px2, tx2) - torch.min(px1, tx1)
    ch = torch.max(py2, ty2) - torch.min(py1, ty1)
    if iou_type == "giou":
        ca = cw * ch + eps
        iou = iou - (ca - union) / ca
    elif iou_type in ("diou", "ciou"):
        c2 = cw ** 2 + ch ** 2 + eps
        rho2 = ((tx1 + tx2 - px1 - px2) ** 2 + (ty1 + ty2 - py1 - py2) ** 2) / 4
        if iou_type == "diou":
            iou = iou - rho2 / c2
        else:
            v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
            with torch.no_grad():
                alpha = v / (v - iou + (1 + eps))
            iou = iou - (rho2 / c2 + v * alpha)
    elif iou_type == "siou":
        scw, sch = (tx1 + tx2 - px1 - px2) * 0.5, (ty1 + ty2 - py1 - py2) * 0.5
        sigma = (scw ** 2 + sch ** 2) ** 0.5
        s1, s2 = scw.abs() / sigma, sch.abs() / sigma
        sa = torch.where(s1 > 2 ** 0.5 / 2, s2, s1)
        angle = torch.cos(torch.arcsin(sa) * 2 - math.pi / 2)
        gamma = angle - 2
        dist = 2 - torch.exp(gamma * (scw / cw) ** 2) - torch.exp(gamma * (sch / ch) ** 2)
        ow, oh = (w1 - w2).abs() / torch.max(w1, w2), (h1 - h2).abs() / torch.max(h1, h2)
        shape = (1 - torch.exp(-ow)) ** 4 + (1 - torch.exp(-oh)) ** 4
        iou = iou - 0.5 * (dist + shape)
    elif iou_type != "iou":
        raise ValueError(iou_type)
    return 1.0 - iou


def yolox_losses(raw, labels, anchors, num_classes=80, return_assign=False, use_l1=False, center_radius=2.5,
                 cls_weight=1.0, iou_weight=3.0, reg_weight=5.0, iou_type=None):
    """get_losses (yolox_head.py:274-441) on raw head output [B,A,5+nc] and labels [B,L,5].
    returns (total, 5*iou, obj, cls, l1, num_fg/num_gt) (+ per-image assignments); l1 (head.use_l1, :389-427) is
    nn.L1Loss(reduction="none") between the RAW regression outputs of the foreground anchors and get_l1_target,
    summed / num_fg, and 0.0 when the switch is off.
    The YOLOv6 head's ComputeLoss (yolov6_head.py:315-531) is the same computation with configurable constants
    (center_radius, cls_weight, iou_weight, reg_weight), an IOUlossV6 box loss (iou_type "giou" / "diou" / "ciou" /
    "siou"; None = the YOLOX head's 1 - iou^2) and the l1 term always on."""
    out = decode(raw, anchors)
    bbox, obj, cls = out[..., :4], out[..., 4:5], out[..., 5:]
    nlabel = (labels.sum(dim=2) > 0).sum(dim=1)
    B, A = raw.shape[:2]
    cls_t, reg_t, obj_t, fgs, assigns, l1_t = [], [], [], [], [], []
    num_fg, num_gts = 0.0, 0.0
    for b in range(B):
        G = int(nlabel[b])
        num_gts += G
        if G == 0:
            cls_t.append(raw.new_zeros((0, num_classes)))
            reg_t.append(raw.new_zeros((0, 4)))
            obj_t.append(raw.new_zeros((A, 1)))
            fgs.append(torch.zeros(A, dtype=torch.bool))
            assigns.append(None)
            l1_t.append(raw.new_zeros((0, 4)))
            continue
        gt, gcls = labels[b, :G, 1:5], labels[b, :G, 0]
        a = simota_image(gt, gcls, bbox[b].detach(), obj[b, :, 0].detach(), cls[b].detach(), anchors, num_classes,
                         center_radius, cls_weight, iou_weight)
        assigns.append(a)
        num_fg += a["num_fg"]
        cls_t.append(F.one_hot(a["matched_cls"].to(torch.int64), num_classes) * a["matched_iou"].unsqueeze(-1))
        obj_t.append(a["fg"].unsqueeze(-1).to(raw.dtype))
        reg_t.append(gt[a["matched_gt"]])
        fgs.append(a["fg"])
        if use_l1:
            l1_t.append(l1_target(gt[a["matched_gt"]], anchors[a["fg"], 2], anchors[a["fg"], 0], anchors[a["fg"], 1]))
    cls_t, reg_t, obj_t, fg = torch.cat(cls_t, 0), torch.cat(reg_t, 0), torch.cat(obj_t, 0), torch.cat(fgs, 0)
    num_fg = max(num_fg, 1)
    if iou_type is None:
        l_iou = iou_loss(bbox.reshape(-1, 4)[fg], reg_t).sum() / num_fg
    else:
        l_iou = iou_loss_v6(bbox.reshape(-1, 4)[fg], reg_t, iou_type).sum() / num_fg
    l_obj = F.binary_cross_entropy_with_logits(obj.reshape(-1, 1), obj_t, reduction="none").sum() / num_fg
    l_cls = F.binary_cross_entropy_with_logits(cls.reshape(-1, num_classes)[fg], cls_t, reduction="none").sum() / num_fg
    l_l1 = 0.0
    if use_l1:
        l_l1 = (raw[..., :4].reshape(-1, 4)[fg] - torch.cat(l1_t, 0)).abs().sum() / num_fg
    total = reg_weight * l_iou + l_obj + l_cls + l_l1
    res = (total, reg_weight * l_iou, l_obj, l_cls, l_l1, num_fg / max(num_gts, 1))
    return (res, assigns) if return_assign else res


# ----------------------------------------------------------------------------- NMS / postprocess
def nms(boxes, scores, thr):
    """torchvision.ops.nms (CPU kernel semantics): descending score, suppress iff IoU > thr, no +1"""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64)
    b = boxes.detach().cpu().float().numpy()
    order = torch.sort(scores.detach().cpu().float(), descending=True, stable=True)[1].numpy()
    import numpy as np
    areas = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(np.float32)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(b[i, 0], b[rest, 0]); yy1 = np.maximum(b[i, 1], b[rest, 1])
        xx2 = np.minimum(b[i, 2], b[rest, 2]); yy2 = np.minimum(b[i, 3], b[rest, 3])
        w = np.maximum(np.float32(0), xx2 - xx1); h = np.maximum(np.float32(0), yy2 - yy1)
        inter = (w * h).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > np.float32(thr)]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def batched_nms(boxes, scores, idxs, thr):
    """torchvision.ops.batched_nms (0.12): coordinate trick when boxes.numel() <= 4000 else per-class loop"""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    if boxes.numel() > 4000:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for cid in torch.unique(idxs):
            ci = torch.where(idxs == cid)[0]
            keep_mask[ci[nms(boxes[ci], scores[ci], thr)]] = True
        ki = torch.where(keep_mask)[0]
        return ki[torch.sort(scores[ki], descending=True, stable=True)[1]]
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, thr)


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45):
    """boxes.py:171-210 (prediction: decoded eval output [B,A,5+nc]); does not mutate its input"""
    pred = prediction.clone()
    c = prediction
    pred[:, :, 0] = c[:, :, 0] - c[:, :, 2] / 2
    pred[:, :, 1] = c[:, :, 1] - c[:, :, 3] / 2
    pred[:, :, 2] = c[:, :, 0] + c[:, :, 2] / 2
    pred[:, :, 3] = c[:, :, 1] + c[:, :, 3] / 2
    output = [None] * len(pred)
    for i, ip in enumerate(pred):
        class_conf, class_pred = torch.max(ip[:, 5:5 + num_classes], 1, keepdim=True)
        mask = (ip[:, 4] * class_conf.squeeze() >= conf_thre).squeeze()
        det = torch.cat((ip[:, :5], class_conf, class_pred.float()), 1)[mask]
        if not det.size(0):
            continue
        keep = batched_nms(det[:, :4], det[:, 4] * det[:, 5], det[:, 6], nms_thre)
        output[i] = det[keep]
    return output


# ----------------------------------------------------------------------------- synthetic inputs
def synth_batch(B, H, W, seed=1234, max_labels=100, num_classes=80, max_gt=20, min_gt=1):
    """COCO-shaped synthetic batch: images U{0..255} float [B,3,H,W]; labels [B,max_labels,5] (cls,cx,cy,w,h)"""
    g = torch.Generator().manual_seed(seed)
    images = torch.randint(0, 256, (B, 3, H, W), generator=g).float()
    labels = torch.zeros(B, max_labels, 5)
    for b in range(B):
        n = int(torch.randint(min_gt, max_gt + 1, (1,), generator=g))
        wh = 16 + torch.rand(n, 2, generator=g) * (min(272, min(H, W) - 2) - 16)
        cx = wh[:, 0] / 2 + torch.rand(n, generator=g) * (W - wh[:, 0])
        cy = wh[:, 1] / 2 + torch.rand(n, generator=g) * (H - wh[:, 1])
        labels[b, :n, 0] = torch.randint(0, num_classes, (n,), generator=g).float()
        labels[b, :n, 1], labels[b, :n, 2] = cx, cy
        labels[b, :n, 3:5] = wh
    return images, labels


def synth_decoded(B, n, seed, num_classes=80):
    """decoded eval predictions [B,n,5+nc] clustered around 20 centres (a random-init net yields no
    detections above threshold) — input of the postprocess / NMS parity tests"""
    g = torch.Generator().manual_seed(seed)
    ctr = torch.rand(B, 20, 2, generator=g) * 560 + 40
    which = torch.randint(0, 20, (B, n), generator=g)
    c = torch.gather(ctr, 1, which[..., None].expand(B, n, 2)) + torch.randn(B, n, 2, generator=g) * 6
    wh = 30 + torch.rand(B, n, 2, generator=g) * 90
    obj = torch.rand(B, n, 1, generator=g)
    cls = torch.rand(B, n, num_classes, generator=g) * 0.3
    hot = torch.randint(0, num_classes, (B, n), generator=g)
    cls.scatter_(2, hot[..., None], torch.rand(B, n, 1, generator=g) * 0.6 + 0.4)
    return torch.cat([c, wh, obj, cls], 2)


def synth_raw(B, hw, seed, num_classes=80, labels=None):
    """raw head outputs with a realistic spread.  With `labels`, ~60% of the anchors predict a jittered
    copy of their nearest ground truth (a half-trained head): SimOTA then sees high IoUs, dynamic k > 1,
    and anchors claimed by several ground truths."""
    g = torch.Generator().manual_seed(seed)
    anchors = make_anchors(hw)
    A = anchors.shape[0]
    raw = torch.zeros(B, A, 5 + num_classes)
    raw[..., :2] = torch.rand(B, A, 2, generator=g) * 1.4 - 0.2
    raw[..., 2:4] = torch.randn(B, A, 2, generator=g) * 0.7 + 1.0
    raw[..., 4] = torch.randn(B, A, generator=g) * 2.0 - 1.0
    raw[..., 5:] = torch.randn(B, A, num_classes, generator=g) * 1.5 - 2.0
    if labels is not None:
        s = anchors[:, 2]
        ctr = (anchors[:, :2] + 0.5) * s[:, None]
        for b in range(B):
            G = int((labels[b].sum(1) > 0).sum())
            if G == 0:
                continue
            gt = labels[b, :G, 1:5]
            near = torch.cdist(ctr, gt[:, :2]).argmin(1)
            use = torch.rand(A, generator=g) < 0.6
            tgt = gt[near]
            cxy = tgt[:, :2] + torch.randn(A, 2, generator=g) * 0.08 * tgt[:, 2:]
            wh = tgt[:, 2:] * torch.exp(torch.randn(A, 2, generator=g) * 0.15)
            raw[b, use, 0:2] = (cxy / s[:, None] - anchors[:, :2])[use]
            raw[b, use, 2:4] = torch.log(wh / s[:, None])[use]
            cl = labels[b, :G, 0].long()[near]
            rows = use.nonzero().squeeze(1)
            raw[b, rows, 5 + cl[rows]] += 3.0
    return raw, anchors


def init_state_dict(depth=0.33, width=0.5, num_classes=80, seed=0, depthwise=False):
    """random-init weights of the architecture with the reference's key names / shapes (default PyTorch
    init, BN defaults, initialize_biases(0.01)). Built from a key/shape table, not from reference code."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(p, cin, cout, k, groups=1):
        if depthwise and k == 3 and groups == 1 and p.startswith("backbone") and "stem" not in p:
            conv(p + ".dconv", cin, cin, 3, groups=cin)      # DWConv (wrappers.py:86-102)
            conv(p + ".pconv", cin, cout, 1)
            return
        fan_in = cin // groups * k * k
        bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        sd[p + ".conv.weight"] = (torch.rand(cout, cin // groups, k, k, generator=g) * 2 - 1) * bound
        sd[p + ".bn.weight"] = torch.ones(cout)
        sd[p + ".bn.bias"] = torch.zeros(cout)
        sd[p + ".bn.running_mean"] = torch.zeros(cout)
        sd[p + ".bn.running_var"] = torch.ones(cout)
        sd[p + ".bn.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def csp(p, cin, cout, n):
        h = int(cout * 0.5)
        conv(p + ".conv1", cin, h, 1); conv(p + ".conv2", cin, h, 1); conv(p + ".conv3", 2 * h, cout, 1)
        for i in range(n):
            conv(f"{p}.m.{i}.conv1", h, h, 1); conv(f"{p}.m.{i}.conv2", h, h, 3)

    bc, bd = int(width * 64), max(round(depth * 3), 1)
    conv("backbone.stem.conv", 12, bc, 3)
    conv("backbone.dark2.0", bc, bc * 2, 3); csp("backbone.dark2.1", bc * 2, bc * 2, bd)
    conv("backbone.dark3.0", bc * 2, bc * 4, 3); csp("backbone.dark3.1", bc * 4, bc * 4, bd * 3)
    conv("backbone.dark4.0", bc * 4, bc * 8, 3); csp("backbone.dark4.1", bc * 8, bc * 8, bd * 3)
    conv("backbone.dark5.0", bc * 8, bc * 16, 3)
    conv("backbone.dark5.1.conv1", bc * 16, bc * 8, 1); conv("backbone.dark5.1.conv2", bc * 32, bc * 16, 1)
    csp("backbone.dark5.2", bc * 16, bc * 16, bd)
    c0, c1, c2 = int(256 * width), int(512 * width), int(1024 * width)
    n = round(3 * depth)
    conv("neck.lateral_conv0", c2, c1, 1); csp("neck.C3_p4", 2 * c1, c1, n)
    conv("neck.reduce_conv1", c1, c0, 1); csp("neck.C3_p3", 2 * c0, c0, n)
    conv("neck.bu_conv2", c0, c0, 3); csp("neck.C3_n3", 2 * c0, c1, n)
    conv("neck.bu_conv1", c1, c1, 3); csp("neck.C3_n4", 2 * c1, c2, n)
    hid = int(256 * width)
    prior = -math.log((1 - 0.01) / 0.01)
    for k, cin in enumerate((c0, c1, c2)):
        conv(f"head.stems.{k}", cin, hid, 1)
        for br in ("cls_convs", "reg_convs"):
            conv(f"head.{br}.{k}.0", hid, hid, 3); conv(f"head.{br}.{k}.1", hid, hid, 3)
        for name, co in (("cls_preds", num_classes), ("reg_preds", 4), ("obj_preds", 1)):
            bound = 1.0 / math.sqrt(hid)
            sd[f"head.{name}.{k}.weight"] = (torch.rand(co, hid, 1, 1, generator=g) * 2 - 1) * bound
            sd[f"head.{name}.{k}.bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound
        sd[f"head.cls_preds.{k}.bias"].fill_(prior)
        sd[f"head.obj_preds.{k}.bias"].fill_(prior)
    return sd


def train_step_losses(sd, images, labels, depth=0.33, width=0.5, num_classes=80, quant=None, return_all=False,
                      use_l1=False, depthwise=False):
    """forward + loss on CPU fp32; sd tensors may require grad"""
    net = Net(sd, depth, width, num_classes, training=True, quant=quant, depthwise=depthwise)
    raw, hw = net.forward_raw(images)
    anchors = make_anchors(hw)
    if return_all:
        res, assigns = yolox_losses(raw, labels, anchors, num_classes, return_assign=True, use_l1=use_l1)
        return res, dict(raw=raw, hw=hw, anchors=anchors, assigns=assigns, taps=net.taps)
    return yolox_losses(raw, labels, anchors, num_classes, use_l1=use_l1)
