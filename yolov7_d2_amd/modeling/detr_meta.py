"""`Detr` meta-architecture - the drop-in for yolov7/modeling/meta_arch/detr.py:33-279 (BASELINE.json config 4,
configs/coco/detr/detr_256_6_6_torchvision.yaml) with its helpers: `NestedTensor` / `nested_tensor_from_tensor_list`
(utils/misc.py:52-108), `Joiner` (backbone/detr_backbone.py:496-512), `MaskedBackbone(TraceFriendly)`
(detr.py:296-403) and `PostProcess` (detr.py:650-678).

Same registry name, constructor `(cfg)`, config keys (MODEL.DETR.*, MODEL.RESNETS.*, MODEL.BACKBONE.FREEZE_AT,
MODEL.PIXEL_MEAN / STD), `forward(batched_inputs)` contract (train: dict of weighted losses attached to autograd; eval:
[{"instances": Instances(pred_boxes, scores, pred_classes)}]) and state_dict keys (detr.backbone.0.backbone.<resnet>,
detr.transformer.*, detr.class_embed, detr.bbox_embed.layers.N, detr.query_embed, detr.input_proj, criterion.empty_weight).
Compute: ResNet-50 (modeling/resnet.py), the transformer with dropout (modeling/transformer.py), the GPU Hungarian
matcher + set criterion (detr_matcher.py / detr_criterion.py), all in libmi355det.  Not built: the mask head
(MODEL.MASK_ON: DETRsegm) and the ONNX-export branches.
"""
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from .. import _lib as L
from ..d2shim import META_ARCH_REGISTRY, Boxes, ImageList, Instances, build_backbone, detector_postprocess
from ..ops import HostRing, feed_batch_enabled, normalize_pad_batch
from .box_ops import box_cxcywh_to_xyxy, box_xyxy_to_cxcywh
from .detr import DETR
from .detr_criterion import SetCriterion
from .detr_matcher import HungarianMatcher
from .position_encoding import PositionEmbeddingSine
from .transformer import Transformer


class NestedTensor(object):
    """utils/misc.py:52-74: a padded batch + its padding mask (True = padding)"""

    def __init__(self, tensors, mask: Optional[torch.Tensor]):
        self.tensors = tensors
        self.mask = mask
        self.image_sizes = [i.shape[1:] for i in self.tensors]

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[torch.Tensor]):
    """utils/misc.py:86-108: zero-pad [C, h, w] images to the batch maximum, mask = True on the padding"""
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    max_size = [max(s) for s in zip(*[list(img.shape) for img in tensor_list])]
    b, (c, h, w) = len(tensor_list), max_size
    tensor = torch.zeros((b, c, h, w), dtype=tensor_list[0].dtype, device=tensor_list[0].device)
    mask = torch.ones((b, h, w), dtype=torch.bool, device=tensor_list[0].device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(tensor, mask)


class MaskedBackbone(nn.Module):
    """detr.py:296-403 (MaskedBackbone == MaskedBackboneTraceFriendly without the ONNX branch): detectron2 backbone +
    per-level padding masks from the true image sizes, ceil(size / stride)"""

    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        shapes = self.backbone.output_shape()
        self.feature_strides = [shapes[f].stride for f in shapes.keys()]
        self.num_channels = shapes[list(shapes.keys())[-1]].channels

    def forward(self, images):
        tensor = images.tensor if isinstance(images, ImageList) else images.tensors
        features = self.backbone(tensor)
        if isinstance(images, ImageList) and getattr(images, "sizes_dev", None) is not None:
            masks = self.mask_out_padding_dev([f.shape for f in features.values()], images.sizes_dev)
        elif isinstance(images, ImageList):
            masks = self.mask_out_padding([f.shape for f in features.values()], images.image_sizes, tensor.device)
        else:   # a NestedTensor carries its own pixel mask: nearest down-sampling, as the reference's export branch does
            m = images.mask
            masks = [torch.nn.functional.interpolate(m[None].float(), size=f.shape[-2:]).to(torch.bool)[0]
                     for f in features.values()]
        assert len(features) == len(masks)
        return {k: NestedTensor(features[k], masks[i]) for i, k in enumerate(features.keys())}

    def mask_out_padding(self, feature_shapes, image_sizes, device):
        masks = []
        assert len(feature_shapes) == len(self.feature_strides)
        for idx, shape in enumerate(feature_shapes):
            N, _, H, W = shape
            m = torch.ones((N, H, W), dtype=torch.bool, device=device)
            for img_idx, (h, w) in enumerate(image_sizes):
                m[img_idx, : int(np.ceil(float(h) / self.feature_strides[idx])),
                  : int(np.ceil(float(w) / self.feature_strides[idx]))] = 0
            masks.append(m)
        return masks


    def mask_out_padding_dev(self, feature_shapes, sizes_dev):
        """the same masks from a DEVICE tensor of (h, w) rows: no host value enters a launch argument, so a step captured
        as a hipGraph serves every batch of the same padded shape (Detr.prepare_batch)"""
        masks = []
        dev = sizes_dev.device
        nlev = len(feature_shapes)
        if feed_batch_enabled() and dev.type == "cuda" and nlev <= 8 and sizes_dev.dtype == torch.int64 and sizes_dev.is_contiguous():
            # all levels in one launch (mi_padding_masks) instead of nine torch calls per level
            import ctypes as C
            masks = [torch.empty(s[0], s[2], s[3], dtype=torch.bool, device=dev) for s in feature_shapes]
            ptrs = (C.c_void_p * nlev)(*[m.data_ptr() for m in masks])
            Hs = (C.c_int * nlev)(*[int(s[2]) for s in feature_shapes])
            Ws = (C.c_int * nlev)(*[int(s[3]) for s in feature_shapes])
            sts = (C.c_int * nlev)(*[int(v) for v in self.feature_strides])
            L.check(L.lib().mi_padding_masks(sizes_dev.data_ptr(), int(feature_shapes[0][0]), nlev, ptrs, Hs, Ws, sts, L.stream_ptr()),
                    "mi_padding_masks")
            return masks
        for idx, shape in enumerate(feature_shapes):
            N, _, H, W = shape
            st = self.feature_strides[idx]
            hv = torch.div(sizes_dev[:, 0] + (st - 1), st, rounding_mode="floor")      # ceil(h / stride)
            wv = torch.div(sizes_dev[:, 1] + (st - 1), st, rounding_mode="floor")
            ys = torch.arange(H, device=dev)[None, :, None]
            xs = torch.arange(W, device=dev)[None, None, :]
            masks.append((ys >= hv[:, None, None]) | (xs >= wv[:, None, None]))
        return masks


MaskedBackboneTraceFriendly = MaskedBackbone


class Joiner(nn.Sequential):
    """detr_backbone.py:496-512: (backbone, position embedding) -> ([NestedTensor per level], [pos per level])"""

    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, tensor_list):
        xs = self[0](tensor_list)
        out, pos = list(xs.values()), []
        # DETR.forward reads features[-1] and pos[-1] only (detr.py:441-445); the reference encodes every level the
        # backbone returns (res2's 200 x 334 map included: 0.7 ms per step here) - the unused entries stay None
        for i, x in enumerate(out):
            pos.append(self[1](x).to(x.tensors.dtype) if (i == len(out) - 1 or self.all_levels) else None)
        return out, pos

    all_levels = False


class PostProcess(nn.Module):
    """detr.py:650-678: model output -> [{"scores", "labels", "boxes" (absolute xyxy)}] for the COCO API"""

    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(out_logits) == len(target_sizes)
        assert target_sizes.shape[1] == 2
        prob = torch.softmax(out_logits.float(), -1)
        scores, labels = prob[..., :-1].max(-1)
        boxes = box_cxcywh_to_xyxy(out_bbox.float())
        img_h, img_w = target_sizes.unbind(1)
        scale_fct = torch.stack([img_w, img_h, img_w, img_h], dim=1)
        boxes = boxes * scale_fct[:, None, :]
        return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, boxes)]


@META_ARCH_REGISTRY.register()
class Detr(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.ignore_thresh = cfg.MODEL.YOLO.CONF_THRESHOLD
        self.num_classes = cfg.MODEL.DETR.NUM_CLASSES
        self.mask_on = cfg.MODEL.MASK_ON
        if self.mask_on:
            raise NotImplementedError("Detr: MODEL.MASK_ON (DETRsegm mask head) is not on this path")
        d = cfg.MODEL.DETR
        hidden_dim, num_queries = d.HIDDEN_DIM, d.NUM_OBJECT_QUERIES
        deep_supervision = d.DEEP_SUPERVISION
        d2_backbone = MaskedBackbone(cfg)
        backbone = Joiner(d2_backbone, PositionEmbeddingSine(hidden_dim // 2, normalize=True))
        backbone.num_channels = d2_backbone.num_channels
        transformer = Transformer(d_model=hidden_dim, dropout=d.DROPOUT, nhead=d.NHEADS, dim_feedforward=d.DIM_FEEDFORWARD,
                                  num_encoder_layers=d.ENC_LAYERS, num_decoder_layers=d.DEC_LAYERS,
                                  normalize_before=d.PRE_NORM, return_intermediate_dec=deep_supervision)
        self.detr = DETR(backbone, transformer, num_classes=self.num_classes, num_queries=num_queries,
                         aux_loss=deep_supervision)
        matcher = HungarianMatcher(cost_class=1, cost_bbox=d.L1_WEIGHT, cost_giou=d.GIOU_WEIGHT)
        weight_dict = {"loss_ce": 1, "loss_bbox": d.L1_WEIGHT, "loss_giou": d.GIOU_WEIGHT}
        if deep_supervision:
            aux = {}
            for i in range(d.DEC_LAYERS - 1):
                aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
            weight_dict.update(aux)
        self.criterion = SetCriterion(self.num_classes, matcher=matcher, weight_dict=weight_dict,
                                      eos_coef=d.NO_OBJECT_WEIGHT, losses=["labels", "boxes", "cardinality"])
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1), persistent=False)
        self.pixel_mean_host, self.pixel_std_host = [float(v) for v in cfg.MODEL.PIXEL_MEAN], [float(v) for v in cfg.MODEL.PIXEL_STD]
        self.normalizer = lambda x: (x - self.pixel_mean) / self.pixel_std
        self.iter = 0
        self.to(self.device)

    def update_iter(self, i):
        self.iter = i

    def preprocess_image(self, batched_inputs):
        """detr.py:270-276: normalise, zero-pad to the batch maximum (ImageList.from_tensors, size_divisibility 0)"""
        images = [self.normalizer(x["image"].to(self.device).float()) for x in batched_inputs]
        return ImageList.from_tensors(images)

    def prepare_targets(self, targets):
        """detr.py:199-213: absolute XYXY -> normalised cxcywh"""
        new_targets = []
        for t in targets:
            h, w = t.image_size
            image_size_xyxy = torch.as_tensor([w, h, w, h], dtype=torch.float, device=self.device)
            gt_boxes = box_xyxy_to_cxcywh(t.gt_boxes.tensor.to(self.device) / image_size_xyxy)
            new_targets.append({"labels": t.gt_classes.to(self.device), "boxes": gt_boxes})
        return new_targets

    # ---- the training step split at the host / device line (graph_step.GraphedTrainStep captures the device half)
    target_capacity = 100     # ground-truth boxes per image the packed layout holds (COCO: <= 93 after crowd removal)

    # the captured step's padded shape is rounded UP to a multiple of this (0 / 1: the exact batch maximum, as
    # ImageList.from_tensors pads in forward()).  DETR's multi-scale input (MIN_SIZE_TRAIN 480..832, max 1333, random crops)
    # makes the exact (max h, max w) nearly unique per batch - a capture each; in 64-pixel buckets a few dozen shapes serve
    # the whole schedule (GraphedTrainStep keeps the most recent max_graphs of them in ONE memory pool).  Arithmetic
    # consequence, stated: the extra rows / columns are ordinary padding - zeros in the image, True in the mask, so no
    # attention weight and no position-embedding count reaches them - and every image keeps its own mask; what changes is
    # that the batch's LARGEST image now has padding pixels inside the tensor at its right / bottom border like every
    # smaller image of a batch always has (the backbone's FrozenBN shift makes padding non-zero after the first layer,
    # where the tensor border would have been an implicit zero): its last feature column / row sees the same values a
    # smaller image's does.  The eager forward() is unchanged (exact padding).
    shape_bucket = 64

    def grad_cut_modules(self):
        """where GraphedTrainStep may cut the backward under data parallel: the ResNet stages (transformer + heads are
        stage 0; res5 / res4 / res3 follow, each stage's gradients on the wire while the next one computes)"""
        bb = self.detr.backbone[0].backbone
        return bb.stage_modules() if hasattr(bb, "stage_modules") else []

    def batch_key(self, batched_inputs):
        r = max(int(self.shape_bucket), 1)
        up = lambda v: (v + r - 1) // r * r
        return (len(batched_inputs), up(max(int(x["image"].shape[-2]) for x in batched_inputs)),
                up(max(int(x["image"].shape[-1]) for x in batched_inputs)))

    def prepare_batch(self, batched_inputs, static=None):
        """Everything of forward() that touches the host: normalise + zero-pad the images into one tensor, record the
        image sizes ON THE DEVICE, and (training) move the ground truth over packed in the criterion's device layout
        (detr_matcher.PackedTargets).  With `static` - an earlier result for the same batch_key - the tensors are
        refilled IN PLACE, so a captured step that reads them sees the new batch."""
        from .detr_matcher import PackedTargets
        B, Hp, Wp = self.batch_key(batched_inputs)
        dev = self.device
        if static is None:
            tensor = torch.zeros(B, 3, Hp, Wp, device=dev)
            images = ImageList(tensor, [(0, 0)] * B)
            images.sizes_dev = torch.zeros(B, 2, dtype=torch.int64, device=dev)
            cap = self.target_capacity
            levels = len(self.detr.transformer.decoder.layers) if self.detr.aux_loss else 1
            targets = PackedTargets([None] * B, cap, torch.zeros(B + 1, dtype=torch.int32, device=dev),
                                    torch.zeros(B * cap, dtype=torch.int64, device=dev),
                                    torch.zeros(B * cap, 4, dtype=torch.float32, device=dev), torch.ones(1, device=dev),
                                    levels=levels)
            static = dict(images=images, targets=targets, key=(B, Hp, Wp))
        assert static["key"] == (B, Hp, Wp), (static["key"], (B, Hp, Wp))
        images, targets = static["images"], static["targets"]
        sizes = [(int(x["image"].shape[-2]), int(x["image"].shape[-1])) for x in batched_inputs]
        if feed_batch_enabled():
            # normalise + zero-pad the whole batch in one launch (the reference: two torch calls + a slice copy per image)
            normalize_pad_batch([x["image"] for x in batched_inputs], images.tensor, self.pixel_mean_host, self.pixel_std_host)
        else:
            images.tensor.zero_()
            for b, x in enumerate(batched_inputs):
                img = x["image"].to(dev).float()
                h, w = sizes[b]
                images.tensor[b, :, :h, :w].copy_(self.normalizer(img))
        images.image_sizes = sizes
        # (host values through the page-locked ring: a blocking copy here waits for the previous step's graph, ops.HostRing)
        HostRing.upload(images.sizes_dev, sizes)
        if self.training:
            cap = targets.cap
            labels, boxes, off = [], [], [0]
            for b, x in enumerate(batched_inputs):
                t = x["instances"]
                n = len(t)
                if n > cap:
                    raise ValueError(f"Detr.prepare_batch: {n} ground-truth boxes in one image, target_capacity is {cap}")
                h, w = t.image_size
                xy = t.gt_boxes.tensor.float().cpu() / torch.tensor([w, h, w, h], dtype=torch.float)
                x0, y0, x1, y1 = xy.unbind(-1)          # (box_xyxy_to_cxcywh of utils/boxes.py:28-32, on the host)
                bx = torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0), (y1 - y0)], dim=-1)
                labels.append(t.gt_classes.cpu().to(torch.int64))
                boxes.append(bx)
                off.append(off[-1] + n)
                targets[b] = None      # (the per-image dicts are views of the packed tensors, rebuilt below)
            ntot = off[-1]
            if ntot:
                HostRing.upload(targets.tgt_labels[:ntot], torch.cat(labels))
                HostRing.upload(targets.tgt_boxes[:ntot], torch.cat(boxes))
            HostRing.upload(targets.tgt_off, off)
            targets.fill_levels(off, torch.cat(labels) if ntot else None, torch.cat(boxes) if ntot else None)
            nb = HostRing.upload(torch.empty(1, device=dev), [float(ntot)])
            world = 1
            if torch.distributed.is_available() and torch.distributed.is_initialized():      # detr.py:616-619
                torch.distributed.all_reduce(nb)
                world = torch.distributed.get_world_size()
            targets.inv_num_boxes.copy_(1.0 / torch.clamp(nb / world, min=1.0))
            for b in range(B):
                targets[b] = {"labels": targets.tgt_labels[off[b]:off[b + 1]], "boxes": targets.tgt_boxes[off[b]:off[b + 1]]}
        return static

    def forward_prepared(self, static):
        """the device half of the training forward: no host value of the batch enters a launch (capturable)"""
        output = self.detr(static["images"])
        if hasattr(self.criterion, "weighted_packed"):
            return self.criterion.weighted_packed(output, static["targets"])      # (+ "total": see GraphedTrainStep)
        loss_dict = self.criterion(output, static["targets"])
        weight_dict = self.criterion.weight_dict
        return {k: (v * weight_dict[k] if k in weight_dict else v) for k, v in loss_dict.items()}

    def forward(self, batched_inputs):
        if self.device.type != "cuda":
            raise L.MI355Error(f"Detr on MODEL.DEVICE={self.device}: the MI355X path needs a HIP device (no CPU fallback)")
        images = self.preprocess_image(batched_inputs)
        output = self.detr(images)
        if self.training:
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
            targets = self.prepare_targets(gt_instances)
            loss_dict = self.criterion(output, targets)
            weight_dict = self.criterion.weight_dict
            for k in loss_dict.keys():
                if k in weight_dict:
                    loss_dict[k] = loss_dict[k] * weight_dict[k]
            return loss_dict
        results = self.inference(output["pred_logits"], output["pred_boxes"], images.image_sizes)
        processed = []
        for r, inp, image_size in zip(results, batched_inputs, images.image_sizes):
            height, width = inp.get("height", image_size[0]), inp.get("width", image_size[1])
            processed.append({"instances": detector_postprocess(r, height, width)})
        return processed

    def inference(self, box_cls, box_pred, image_sizes):
        """detr.py:215-262: best non-background class per query, score threshold, boxes scaled to the image size"""
        assert len(box_cls) == len(image_sizes)
        results = []
        scores, labels = torch.softmax(box_cls.float(), dim=-1)[:, :, :-1].max(-1)
        for s, l, b, image_size in zip(scores, labels, box_pred.float(), image_sizes):
            keep = s > self.ignore_thresh
            result = Instances(image_size)
            result.pred_boxes = Boxes(box_cxcywh_to_xyxy(b[keep]))
            result.pred_boxes.scale(image_size[1], image_size[0])
            result.scores = s[keep]
            result.pred_classes = l[keep]
            results.append(result)
        return results
