"""YOLOX meta-architecture — the drop-in for yolov7/modeling/meta_arch/yolox.py:35-252.

Same registry name (`YOLOX`), constructor `(cfg)`, config keys, `forward(batched_inputs)` contract,
loss-dict keys and state_dict keys as the reference; the compute is one MI355X step plan
(yolov7_d2_amd/plan.py) per input shape, executed by libmi355det.
"""
import torch
import torch.nn as nn

from .. import _lib as L
from ..d2shim import META_ARCH_REGISTRY, Boxes, ImageList, Instances, build_backbone, detector_postprocess
from ..params import ParamArena
from ..plan import PlanBuilder
from .blocks import EmitCtx
from .postprocess import postprocess
from .yolox_net import YOLOPAFPN, YOLOXHead


def xyxy_to_cxcywh(b):
    """BoxModeMy.convert(XYXY_ABS -> 'XYWH_ABS') of yolov7/utils/boxes.py:547-551, which yields (cx, cy, w, h)"""
    out = b.clone().float()
    out[:, 2] = b[:, 2] - b[:, 0]
    out[:, 3] = b[:, 3] - b[:, 1]
    out[:, 0] = b[:, 0] + out[:, 2] * 0.5
    out[:, 1] = b[:, 1] + out[:, 3] * 0.5
    return out


class _PlanState:
    """one compiled step for a fixed (B, H, W): plan + persistent I/O tensors"""

    def __init__(self, model, B, H, W, training, materialize=True, input_u8=False, use_l1=False):
        dev = model.device
        self.use_l1 = bool(use_l1)
        self.B, self.H, self.W, self.training = B, H, W, training
        # input_u8: the plan reads the uint8 image of the data loader directly (the .type(torch.float) of
        # yolox.py:96-99 is fused into the Focus packer); float32 is the reference-shaped default
        self.image = torch.zeros(B, 3, H, W, dtype=torch.uint8 if input_u8 else torch.float32, device=dev)
        # labels in a flat allocation padded to whole 16-byte pieces: a host-fed step copies them from its staging buffer
        # with an in-graph strided-copy command that moves 16 bytes per lane (engine.NativeTrainer.feed)
        nlab = B * model.max_boxes_num * 5
        self.labels_flat = torch.zeros((nlab + 3) // 4 * 4, dtype=torch.float32, device=dev)
        self.labels = self.labels_flat[:nlab].view(B, model.max_boxes_num, 5)
        hw = [(H // s, W // s) for s in model.head.strides]
        self.A = sum(h * w for h, w in hw)
        self.anchors = YOLOXHead.anchors_for(hw, model.head.strides).to(dev)
        b = PlanBuilder(dev, training=training)
        ctx = EmitCtx(b, model.params)
        outs = model.neck.alloc(ctx, B, H // 8, W // 8)
        feats = model.backbone.emit(ctx, self.image, B, H, W, outs=outs)
        for k, sl in outs.items():  # a backbone that did not write in place: copy (generic fallback)
            if feats[k].buf is not sl.buf:
                raise NotImplementedError("backbone output not written into the neck concat slice")
        fpn = model.neck.emit(ctx, feats)
        nch = 5 + model.num_classes
        self.preds_buf = b.small("preds", B * self.A * nch * 4)
        model.head.emit(ctx, fpn, self.preds_buf, self.A)
        if training:
            self.loss = b.yolox_loss(self.preds_buf, self.labels, self.anchors, B, self.A, model.num_classes,
                                     model.max_boxes_num, model.max_boxes_num, use_l1=use_l1)
        else:
            b.emit("DECODE", i=[B, self.A, model.num_classes], p=[self.preds_buf, self.anchors], tag="decode")
        self.builder = b
        self.plan = b.finalize(materialize)
        self.nch = nch

    def preds(self):
        return self.plan.buf_view(self.preds_buf, torch.float32, self.B * self.A * self.nch).view(self.B, self.A, self.nch)

    def loss_out(self):
        return self.plan.buf_view(self.loss["out"], torch.float32, 8)

    def gw(self):
        """upstream gradients of (total, iou, conf, cls[, l1]) read by the loss backward"""
        return self.plan.buf_view(self.loss["gw"], torch.float32, 5 if self.use_l1 else 4)


class _YoloxTrainFn(torch.autograd.Function):
    """autograd bridge: forward = forward command list, backward = backward command list"""

    @staticmethod
    def forward(ctx, ps, model, images, labels, *params):
        ps.image.copy_(images)
        ps.labels.copy_(labels)
        ps.plan.run("fwd")
        ctx.ps, ctx.model = ps, model
        return ps.loss_out()[:6].clone()

    @staticmethod
    def backward(ctx, g):
        ps, model = ctx.ps, ctx.model
        ps.gw().copy_(g[:5 if ps.use_l1 else 4].to(torch.float32))
        # The kernels write every parameter gradient into the flat arena.  Zero-copy hand-over: a parameter whose .grad
        # is None (optimizer.zero_grad(), set_to_none=True - the default, and what detectron2's trainer does each
        # iteration) gets the arena view bound as its .grad and autograd receives None for it - no 240-tensor clone
        # (36 MB per step).  A parameter whose .grad IS that view already (zero_grad(set_to_none=False), gradient
        # accumulation over micro-batches, a second backward of another loss term) keeps autograd's semantics: its
        # current contents are saved before the kernels overwrite the arena and added back afterwards (zeros after a
        # zero_grad, the earlier gradient otherwise).  A parameter with some other .grad tensor receives the view and
        # autograd adds it.
        named = list(model.named_parameters())
        views = [model.params.grad_of(p) for _, p in named]
        bound = [k for k, ((_, p), v) in enumerate(zip(named, views)) if p.grad is not None and p.grad.data_ptr() == v.data_ptr()]
        saved = None
        if bound:
            saved = model.params.grad.clone() if len(bound) == len(named) else [views[k].clone() for k in bound]
        ps.plan.run("bwd")
        if bound:
            if len(bound) == len(named):
                model.params.grad.add_(saved)
            else:
                for k, o in zip(bound, saved):
                    views[k].add_(o)
        if getattr(model, "grad_accumulate", False):   # (kept for callers of round 2: plain clones for autograd)
            return (None, None, None, None, *[v.clone() for v in views])
        grads = []
        for (_, p), v in zip(named, views):
            if p.grad is None or p.grad.data_ptr() == v.data_ptr():
                p.grad = v
                grads.append(None)
            else:
                grads.append(v)
        return (None, None, None, None, *grads)


@META_ARCH_REGISTRY.register()
class YOLOX(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.conf_threshold = cfg.MODEL.YOLO.CONF_THRESHOLD
        self.nms_threshold = cfg.MODEL.YOLO.NMS_THRESHOLD
        self.nms_type = cfg.MODEL.NMS_TYPE
        self.loss_type = cfg.MODEL.YOLO.LOSS_TYPE
        # l1 loss on the raw regression outputs "at last 15 epochs" (yolox.py:47-55): switched on in forward() once
        # self.iter (update_iter) passes INPUT.MOSAIC_AND_MIXUP.DISABLE_AT_ITER; nothing in the reference tree calls
        # update_iter, a trainer hook that does gets the same behaviour here
        self.use_l1 = False
        self.depth_mul = cfg.MODEL.YOLO.DEPTH_MUL
        self.width_mul = cfg.MODEL.YOLO.WIDTH_MUL
        self.iter = 0
        self.max_iter = cfg.SOLVER.MAX_ITER
        self.enable_l1_loss_at = cfg.INPUT.MOSAIC_AND_MIXUP.DISABLE_AT_ITER
        self.num_classes = cfg.MODEL.YOLO.CLASSES
        self.max_boxes_num = cfg.MODEL.YOLO.MAX_BOXES_NUM
        self.in_features = cfg.MODEL.YOLO.IN_FEATURES
        self.backbone = build_backbone(cfg)
        self.size_divisibility = 32 if self.backbone.size_divisibility == 0 else self.backbone.size_divisibility
        self.neck = YOLOPAFPN(depth=self.depth_mul, width=self.width_mul, in_features=self.in_features)
        self.head = YOLOXHead(self.num_classes, width=self.width_mul)
        self.padded_value = cfg.MODEL.PADDED_VALUE
        self.onnx_export = False   # yolox.py:79-80: forward(tensor [B,H,W,3]) -> decoded predictions in the export layout
        self.onnx_vis = False      # ... or, with onnx_vis, the post-processed detections
        self.apply(self._init_model)
        self.head.initialize_biases(1e-2)
        self.params = None
        self._plans = {}
        self.to(self.device)

    @staticmethod
    def _init_model(M):
        for m in M.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eps = 1e-3
                m.momentum = 0.03

    def update_iter(self, i):
        self.iter = i

    def _apply(self, fn, *a, **k):
        # moving the module invalidates every cached pointer
        self.params = None
        self._plans = {}
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------ plans
    def ensure_params(self):
        if self.params is None:
            if self.device.type != "cuda":
                raise L.MI355Error(f"YOLOX on MODEL.DEVICE={self.device}: the MI355X path needs a HIP device; "
                                   "the CPU reference lives in oracle/ (test infrastructure only)")
            L.require_device()
            self.params = ParamArena(self, self.device)
        return self.params

    def plan_for(self, B, H, W, training, input_u8=False, variant=""):
        """variant: a second plan of the same shape built under other build-time switches (engine.NativeTrainer builds the
        data-parallel step once with the all-reduce overlapped and once exposed, and keeps the faster)"""
        self.ensure_params()
        l1 = bool(training and self.use_l1)
        key = (B, H, W, bool(training)) + (("u8",) if input_u8 else ()) + (("l1",) if l1 else ()) + ((variant,) if variant else ())
        ps = self._plans.get(key)
        if ps is None:
            assert H % 32 == 0 and W % 32 == 0, (H, W)
            ps = _PlanState(self, B, H, W, training, input_u8=input_u8, use_l1=l1)
            self._plans[key] = ps
        return ps

    # ------------------------------------------------------------------ reference-shaped preprocessing
    def preprocess_image(self, batched_inputs, training):
        """yolox.py:95-162: float (no mean/std), pad to /32 with PADDED_VALUE, labels [B, max_boxes, 5]"""
        images = [x["image"].to(self.device).type(torch.float) for x in batched_inputs]
        bs = len(images)
        images = ImageList.from_tensors(images, size_divisibility=self.size_divisibility, pad_value=self.padded_value)
        labels = None
        if training:
            key = "instances" if "instances" in batched_inputs[0] else "targets"
            labels = torch.zeros((bs, self.max_boxes_num, 5))
            for i, x in enumerate(batched_inputs):
                inst = x[key]
                boxes = inst.gt_boxes.tensor.detach().cpu()
                t = torch.cat([inst.gt_classes.detach().cpu().float().unsqueeze(-1), xyxy_to_cxcywh(boxes)], dim=-1)
                t = t[: self.max_boxes_num]
                labels[i, : t.shape[0]] = t
        return images, labels, images.image_sizes

    def preprocess_input(self, x):
        """yolox.py:164-170 (export path): NHWC image tensor -> NCHW, no normalisation"""
        return x.permute(0, 3, 1, 2)

    def _forward_export(self, batched_inputs):
        """yolox.py:172-178, 211-224: the graph the reference hands to torch.onnx.export, run on the HIP path"""
        assert isinstance(batched_inputs, (torch.Tensor, list)), "onnx export, batched_inputs only needs image tensor"
        x = self.preprocess_input(batched_inputs).to(self.device).float().contiguous()
        B, _, H, W = x.shape
        ps = self.plan_for(B, H, W, False)
        ps.image.copy_(x)
        ps.plan.run("fwd")
        dec = ps.preds()
        if self.onnx_vis:
            return postprocess(dec.clone(), self.num_classes, self.conf_threshold, self.nms_threshold)
        out = torch.empty(B, ps.A, 6 + self.num_classes, dtype=torch.float32, device=x.device)
        L.check(L.lib().mi_yolox_onnx_layout(dec.data_ptr(), out.data_ptr(), B, ps.A, self.num_classes, L.stream_ptr()),
                "mi_yolox_onnx_layout")
        return out

    def forward(self, batched_inputs):
        self.ensure_params()   # raises MI355Error when there is no HIP device: no CPU fallback
        if self.onnx_export:
            if self.training:
                raise RuntimeError("YOLOX.onnx_export is an inference mode (the reference's export.py calls model.eval())")
            return self._forward_export(batched_inputs)
        images, labels, image_ori_sizes = self.preprocess_image(batched_inputs, self.training)
        x = images.tensor
        B, _, H, W = x.shape
        if self.training and self.iter > self.enable_l1_loss_at and not self.use_l1:
            # yolox.py:105-121.  The reference broadcasts rank 0's flag; every rank evaluates the same condition on the
            # same iteration counter, so the flag is identical without the collective.
            self.use_l1 = True
            self.head.use_l1 = True
        if self.training:
            ps = self.plan_for(B, H, W, True)
            out = _YoloxTrainFn.apply(ps, self, x, labels.to(x.device), *[p for _, p in self.named_parameters()])
            losses = {"total_loss": out[0], "iou_loss": out[1], "conf_loss": out[2], "cls_loss": out[3]}
            if self.use_l1:
                losses["l1_loss"] = out[4]
            return losses
        ps = self.plan_for(B, H, W, False)
        ps.image.copy_(x)
        ps.plan.run("fwd")
        outputs = ps.preds().clone()
        detections = postprocess(outputs, self.num_classes, self.conf_threshold, self.nms_threshold)
        results = []
        for idx, out in enumerate(detections):
            if out is None:
                out = x.new_zeros((0, 7))
            result = Instances(image_ori_sizes[idx])
            result.pred_boxes = Boxes(out[:, :4])
            result.scores = out[:, 5] * out[:, 4]
            result.pred_classes = out[:, -1]
            results.append(result)
        processed = []
        for r, inp, image_size in zip(results, batched_inputs, images.image_sizes):
            height = inp.get("height", image_size[0])
            width = inp.get("width", image_size[1])
            processed.append({"instances": detector_postprocess(r, height, width)})
        return processed


def run_backbone_standalone(backbone, x):
    """Backbone.forward(x) -> dict[name, Tensor] (darknetx.py:165-177) as a forward-only plan.
    BatchNorm uses batch statistics iff backbone.training; no autograd through a bare backbone."""
    if not x.is_cuda:
        raise L.MI355Error("CSPDarknet.forward needs a HIP tensor (no CPU fallback)")
    L.require_device()
    B, _, H, W = x.shape
    cache = backbone.__dict__.setdefault("_mi_plans", {})
    key = (B, H, W, backbone.training)
    st = cache.get(key)
    if st is None:
        from ..params import ParamArena  # parameters stay where they are: a grad-less arena is not needed

        class _NoGrad:
            def grad_of(self, p):
                return None

        b = PlanBuilder(x.device, training=False, bn_train=backbone.training)
        image = torch.zeros(B, 3, H, W, dtype=torch.float32, device=x.device)
        feats = backbone.emit(EmitCtx(b, _NoGrad()), image, B, H, W)
        st = (b.finalize(), image, feats)
        cache[key] = st
    plan, image, feats = st
    image.copy_(x)
    plan.run("fwd")
    return {k: plan.view(t).float() for k, t in feats.items()}
