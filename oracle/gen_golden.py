"""ORACLE support: generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN SOURCE (by path, see
ref_loader.py) on seeded inputs.  Run in the build container (where /root/reference exists):

    python oracle/gen_golden.py

The vectors pin oracle/yolox_oracle.py (tests/test_oracle_golden.py, CPU) and are the fixed points the
GPU parity tests are checked against.  Weights and inputs are regenerated from seeds through
yolox_oracle.init_state_dict / synth_batch (same torch build on the GPU box), so only reference
OUTPUTS are stored.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
import yolox_oracle as O  # noqa: E402

OUT = os.environ.get("MI_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def gold_step():
    """full training step of the reference on YOLOX-s (depth .33, width .5), B=2, 64x96"""
    depth, width, nc = 0.33, 0.5, 80
    ref, r = ref_loader.build_reference_yolox(depth, width, nc, seed=0)
    sd0 = O.init_state_dict(depth, width, nc, seed=0)
    ref.load_state_dict(sd0)
    ref.train()
    imgs, labels = O.synth_batch(2, 64, 96, seed=11, max_gt=4)
    out = ref(imgs, labels)
    loss_dict_sum = out[0] + out[1] + out[2] + out[3]   # detectron2 sums every value of the returned dict (Q1)
    loss_dict_sum.backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    res = dict(losses=np.array([float(x) for x in out], dtype=np.float64))
    for k in ("head.cls_preds.0.weight", "head.obj_preds.2.bias", "backbone.stem.conv.conv.weight",
              "backbone.stem.conv.bn.weight", "backbone.dark3.1.m.1.conv2.conv.weight", "neck.C3_p4.conv3.bn.bias",
              "backbone.dark2.0.conv.weight"):
        res["grad:" + k] = grads[k].numpy()
    names = sorted(grads)
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array([float(grads[k].norm()) for k in names], dtype=np.float64)
    st = ref.state_dict()
    res["rm:backbone.stem.conv.bn.running_mean"] = st["backbone.stem.conv.bn.running_mean"].numpy()
    res["rv:head.stems.1.bn.running_var"] = st["head.stems.1.bn.running_var"].numpy()
    # eval forward with the (updated) running stats
    ref.eval()
    with torch.no_grad():
        res["eval_out"] = ref(imgs).numpy()
    np.savez_compressed(os.path.join(OUT, "yolox_s_step_64x96.npz"), **res)
    print("step:", res["losses"])


def gold_onnx_layout():
    """eval forward of the reference with head.onnx_export = True (decode_outputs' export layout, yolox_head.py:263-269):
    [B, A, 4 + 1 + 1 + nc] = (xy, wh, conf, argmax class as float, class probabilities)"""
    depth, width, nc = 0.33, 0.5, 80
    ref, r = ref_loader.build_reference_yolox(depth, width, nc, seed=0)
    ref.load_state_dict(O.init_state_dict(depth, width, nc, seed=0))
    ref.eval()
    ref.head.onnx_export = True
    imgs, _ = O.synth_batch(2, 64, 96, seed=11, max_gt=4)
    with torch.no_grad():
        out = ref(imgs)
    np.savez_compressed(os.path.join(OUT, "yolox_s_onnx_layout_64x96.npz"), out=out.numpy())
    print("onnx layout:", tuple(out.shape), float(out[..., 5].mean()))


def gold_dw_step():
    """full training step of the reference with MODEL.DARKNET.DEPTH_WISE True (DWConv in the backbone's 3x3 convs,
    darknetx.py:113; neck and head dense, as yolox.py:60-83 builds them), YOLOX-s widths, B=2, 64x96"""
    depth, width, nc = 0.33, 0.5, 80
    ref, r = ref_loader.build_reference_yolox(depth, width, nc, seed=0, depthwise=True)
    ref.load_state_dict(O.init_state_dict(depth, width, nc, seed=3, depthwise=True))
    ref.train()
    imgs, labels = O.synth_batch(2, 64, 96, seed=12, max_gt=4)
    out = ref(imgs, labels)
    (out[0] + out[1] + out[2] + out[3]).backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    res = dict(losses=np.array([float(x) for x in out], dtype=np.float64))
    for k in ("backbone.dark2.0.dconv.conv.weight", "backbone.dark2.0.pconv.conv.weight", "backbone.dark4.1.m.1.conv2.dconv.conv.weight",
              "backbone.dark5.0.dconv.bn.weight", "backbone.dark3.1.m.0.conv2.dconv.bn.bias", "backbone.stem.conv.conv.weight"):
        res["grad:" + k] = grads[k].numpy()
    names = sorted(grads)
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array([float(grads[k].norm()) for k in names], dtype=np.float64)
    st = ref.state_dict()
    res["rm:backbone.dark3.0.dconv.bn.running_mean"] = st["backbone.dark3.0.dconv.bn.running_mean"].numpy()
    np.savez_compressed(os.path.join(OUT, "yolox_s_dw_step_64x96.npz"), **res)
    print("dw step:", res["losses"], len(names))


def gold_tiny_step():
    """config 0 of BASELINE.json: YOLOX-tiny (depth .33, width .375), 416x416, bs=2, fp32 on the CPU device - the
    reference's own CPU-runnable case; losses, gradient norms of every parameter and the eval output"""
    depth, width, nc = 0.33, 0.375, 80
    ref, r = ref_loader.build_reference_yolox(depth, width, nc, seed=0)
    sd0 = O.init_state_dict(depth, width, nc, seed=0)
    ref.load_state_dict(sd0)
    ref.train()
    imgs, labels = O.synth_batch(2, 416, 416, seed=31, max_gt=6)
    out = ref(imgs, labels)
    (out[0] + out[1] + out[2] + out[3]).backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    names = sorted(grads)
    res = dict(losses=np.array([float(x) for x in out], dtype=np.float64), grad_names=np.array(names),
               grad_norms=np.array([float(grads[k].norm()) for k in names], dtype=np.float64))
    res["grad:head.cls_preds.1.weight"] = grads["head.cls_preds.1.weight"].numpy()
    ref.eval()
    with torch.no_grad():
        ev = ref(imgs)
    res["eval_out_stride"] = ev[:, ::7].numpy()          # every 7th anchor keeps the fixture small
    np.savez_compressed(os.path.join(OUT, "yolox_tiny_step_416.npz"), **res)
    print("tiny step:", res["losses"])


def gold_simota():
    """head loss + SimOTA of the reference on synthetic raw predictions, B=3, 160x160 (A=525)"""
    ref, r = ref_loader.build_reference_yolox(0.33, 0.5, 80, seed=0)
    head = ref.head
    head.train()
    B, H, W = 3, 160, 160
    _, labels = O.synth_batch(B, H, W, seed=21, max_gt=12, min_gt=6)
    labels[1] = 0.0
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    raw, anchors = O.synth_raw(B, hw, 22, labels=labels)
    raw.requires_grad_(True)
    # feed the head's loss exactly what YOLOXHead.forward builds (yolox_head.py:175-224)
    outs, xs, ys, es = [], [], [], []
    a0 = 0
    for (h, w), s in zip(hw, (8, 16, 32)):
        o = (raw * 1.0)[:, a0:a0 + h * w].permute(0, 2, 1).reshape(B, 85, h, w)
        o, grid = head.get_output_and_grid(o, len(outs), s, torch.zeros(1).type())
        xs.append(grid[:, :, 0]); ys.append(grid[:, :, 1])
        es.append(torch.zeros(1, grid.shape[1]).fill_(s))
        outs.append(o)
        a0 += h * w
    res6 = head.get_losses(None, xs, ys, es, labels, torch.cat(outs, 1), [], dtype=torch.float32)
    (res6[0] + res6[1] + res6[2] + res6[3]).backward()
    out = dict(losses=np.array([float(x) for x in res6], dtype=np.float64), draw=raw.grad.numpy())
    # per-image assignment through the reference's get_assignments
    dec = O.decode(raw.detach(), anchors)
    bbox, obj, cls = dec[..., :4], dec[..., 4:5], dec[..., 5:]
    nl = (labels.sum(2) > 0).sum(1)
    for b in range(B):
        G = int(nl[b])
        if G == 0:
            continue
        gm, fg, pi, mg, nfg = head.get_assignments(b, G, anchors.shape[0], labels[b, :G, 1:5], labels[b, :G, 0],
                                                   bbox[b], torch.cat(es, 1), torch.cat(xs, 1), torch.cat(ys, 1),
                                                   cls, bbox, obj, labels, None)
        out[f"fg{b}"] = fg.numpy()
        out[f"matched_gt{b}"] = mg.numpy()
        out[f"matched_iou{b}"] = pi.numpy()
        out[f"matched_cls{b}"] = gm.numpy()
    np.savez_compressed(os.path.join(OUT, "simota_160.npz"), **out)
    print("simota:", out["losses"])


def gold_simota_l1():
    """the same head loss with head.use_l1 = True (yolox_head.py:186-196, 389-427): origin_preds are the raw regression
    outputs per level; upstream gradient 1 on total, iou, conf, cls and l1"""
    ref, r = ref_loader.build_reference_yolox(0.33, 0.5, 80, seed=0)
    head = ref.head
    head.train()
    head.use_l1 = True
    B, H, W = 3, 160, 160
    _, labels = O.synth_batch(B, H, W, seed=21, max_gt=12, min_gt=6)
    labels[1] = 0.0
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    raw, anchors = O.synth_raw(B, hw, 22, labels=labels)
    raw.requires_grad_(True)
    outs, xs, ys, es, origin = [], [], [], [], []
    a0 = 0
    for (h, w), s in zip(hw, (8, 16, 32)):
        o = (raw * 1.0)[:, a0:a0 + h * w].permute(0, 2, 1).reshape(B, 85, h, w)
        reg = o[:, :4]                                      # reg_output of this level, [B,4,h,w]
        origin.append(reg.view(B, 1, 4, h, w).permute(0, 1, 3, 4, 2).reshape(B, -1, 4).clone())
        o, grid = head.get_output_and_grid(o, len(outs), s, torch.zeros(1).type())
        xs.append(grid[:, :, 0]); ys.append(grid[:, :, 1])
        es.append(torch.zeros(1, grid.shape[1]).fill_(s))
        outs.append(o)
        a0 += h * w
    res6 = head.get_losses(None, xs, ys, es, labels, torch.cat(outs, 1), origin, dtype=torch.float32)
    (res6[0] + res6[1] + res6[2] + res6[3] + res6[4]).backward()
    out = dict(losses=np.array([float(x) for x in res6], dtype=np.float64), draw=raw.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "simota_160_l1.npz"), **out)
    print("simota l1:", out["losses"])


def gold_postprocess():
    """postprocess (utils/boxes.py:171-210) on synthetic decoded predictions; NMS = torchvision semantics
    as restated in yolox_oracle (torchvision itself is not installed: parity unpinned for NMS)."""
    ref, r = ref_loader.build_reference_yolox(0.33, 0.5, 80, seed=0)
    res = {}
    for name, n, seed in (("small", 300, 31), ("large", 2500, 32)):
        pred = O.synth_decoded(2, n, seed)
        out = r.boxes.postprocess(pred.clone(), 80, 0.3, 0.65)
        for b, o in enumerate(out):
            res[f"{name}_out{b}"] = o.numpy() if o is not None else np.zeros((0, 7), np.float32)
    np.savez_compressed(os.path.join(OUT, "postprocess.npz"), **res)
    print("postprocess:", {k: v.shape for k, v in res.items() if "out" in k})


def gold_hungarian():
    """the reference's own HungarianMatcher (utils/detr_utils.py:12-91, scipy LSAP) on seeded DETR-shaped inputs"""
    import detr_oracle as D
    r = ref_loader.load()
    res = {}
    for name, (bs, nq, seed, sizes) in dict(a=(3, 100, 41, None), b=(2, 100, 42, [100, 1]), c=(2, 16, 43, [30, 7])).items():
        logits, boxes, targets = D.synth_detr(bs, nq, 91, seed, sizes=sizes)
        m = r.detr_utils.HungarianMatcher(cost_class=1.0, cost_bbox=5.0, cost_giou=2.0)   # DETR weights (config.py:216-218)
        idx = m({"pred_logits": logits, "pred_boxes": boxes}, targets)
        for b, (i, j) in enumerate(idx):
            res[f"{name}_i{b}"], res[f"{name}_j{b}"] = i.numpy(), j.numpy()
    np.savez_compressed(os.path.join(OUT, "hungarian.npz"), **res)
    print("hungarian:", {k: v.shape for k, v in res.items()})


from gen_golden_inputs import synth_box_pairs  # noqa: E402


def gold_iou_v6():
    """the reference's own IOUlossV6 (utils/boxes.py:666-752) + autograd on seeded box pairs"""
    r = ref_loader.load()
    res = {}
    pred, tgt = synth_box_pairs(257, 51)
    for t in ("giou", "diou", "ciou", "siou"):
        p = pred.clone().requires_grad_(True)
        loss = r.boxes.IOUlossV6(box_format="xywh", iou_type=t, reduction="none")(p.T, tgt)
        loss.sum().backward()
        res[t + "_loss"], res[t + "_grad"] = loss.detach().numpy(), p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "iou_v6.npz"), **res)
    print("iou_v6:", {k: float(np.abs(v).mean()) for k, v in res.items()})


def gold_yolox_iou():
    """the reference's IOUloss module (utils/boxes.py:125-168, "iou" and "giou") with autograd, bboxes_iou (:57-81) and
    pairwise_bbox_iou (:755-779) on seeded boxes (some pairs disjoint, some identical: ties of max / min)"""
    r = ref_loader.load()
    res = {}
    pred, tgt = synth_box_pairs(257, 52)
    tgt[200:210] = pred[200:210]                   # identical boxes: every max / min is a tie
    tgt[210:220, :2] = pred[210:220, :2]           # same centre, different size
    for t in ("iou", "giou"):
        p = pred.clone().requires_grad_(True)
        loss = r.boxes.IOUloss(reduction="none", loss_type=t)(p, tgt)
        loss.sum().backward()
        res[t + "_loss"], res[t + "_grad"] = loss.detach().numpy(), p.grad.numpy()
    a, b = synth_box_pairs(37, 53)[0], synth_box_pairs(61, 54)[1]
    res["pair_xywh"] = r.boxes.pairwise_bbox_iou(a, b, "xywh").numpy()
    ax = torch.cat([a[:, :2] - a[:, 2:] / 2, a[:, :2] + a[:, 2:] / 2], 1)
    bx = torch.cat([b[:, :2] - b[:, 2:] / 2, b[:, :2] + b[:, 2:] / 2], 1)
    res["pair_xyxy"] = r.boxes.pairwise_bbox_iou(ax, bx, "xyxy").numpy()
    res["bboxes_iou_xyxy"] = r.boxes.bboxes_iou(ax, bx, True).numpy()
    res["bboxes_iou_xywh"] = r.boxes.bboxes_iou(a, b, False).numpy()
    np.savez_compressed(os.path.join(OUT, "yolox_iou.npz"), **res)
    print("yolox_iou:", {k: float(np.abs(v).mean()) for k, v in res.items()})


def gold_box_ops():
    """the reference's DETR box utilities (utils/boxes.py:28-37 box_cxcywh_to_xyxy / box_xyxy_to_cxcywh, :85-122 box_iou /
    generalized_box_iou) on seeded boxes incl. identical and disjoint pairs"""
    r = ref_loader.load()
    g = torch.Generator().manual_seed(77)
    c = torch.rand(3, 37, 4, generator=g) * torch.tensor([1.0, 1.0, 0.4, 0.4]) + torch.tensor([0.0, 0.0, 0.01, 0.01])
    xyxy = r.boxes.box_cxcywh_to_xyxy(c)
    a, b = xyxy[0], torch.cat([xyxy[1][:23], xyxy[0][:2]])
    iou, uni = r.boxes.box_iou(a, b)
    res = dict(cxcywh=c.numpy(), xyxy=xyxy.numpy(), back=r.boxes.box_xyxy_to_cxcywh(xyxy).numpy(), a=a.numpy(), b=b.numpy(),
               iou=iou.numpy(), union=uni.numpy(), giou=r.boxes.generalized_box_iou(a, b).numpy())
    np.savez_compressed(os.path.join(OUT, "box_ops.npz"), **res)
    print("box_ops:", {k: v.shape for k, v in res.items()})


def gold_nms_family():
    """the reference's own softnms (linear / gaussian), cluster NMS and matrix NMS (meta_arch/utils.py:33-113,
    utils/solov2_utils.py:160-206) on seeded candidates"""
    from gen_golden_inputs import synth_nms_case, synth_mask_case
    r = ref_loader.load_nms_family()
    res = {}
    # case d: tight clusters and a high score threshold, so that Soft-NMS retires boxes
    for name, (n, ncls, seed, spread, thr) in dict(a=(300, 5, 71, 14.0, 0.001), b=(1500, 80, 72, 14.0, 0.001),
                                                   c=(7, 1, 73, 14.0, 0.001), d=(400, 2, 74, 4.0, 0.05)).items():
        boxes, scores, idxs = synth_nms_case(n, ncls, seed, spread)
        for t in ("softnms-linear", "softnms-gaussian", "cluster"):
            sc = scores.clone()
            keep = r.nms_utils.generalized_batched_nms(boxes.clone(), sc, idxs, 0.5, score_threshold=thr, nms_type=t)
            res[f"{name}_{t}_keep"] = keep.numpy()
            if t != "cluster":
                res[f"{name}_{t}_scores"] = sc.numpy()
    for name, (n, H, W, ncls, seed) in dict(m1=(60, 40, 48, 3, 81), m2=(500, 64, 64, 10, 82), m3=(1, 8, 8, 1, 83)).items():
        labels, masks, sums, scores = synth_mask_case(n, H, W, ncls, seed)
        for kern in ("gaussian", "linear"):
            out = r.solov2_utils.matrix_nms(labels, masks, sums, scores, sigma=2.0, kernel=kern)
            res[f"{name}_{kern}"] = out.numpy()
        if n <= 60:      # the reference's greedy mask_nms is an O(n^2) Python loop over full masks
            res[f"{name}_masknms"] = np.asarray(r.solov2_utils.mask_nms(labels, masks, sums, scores, nms_thr=0.3)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "nms_family.npz"), **res)
    print("nms family:", {k: v.shape for k, v in res.items() if "keep" in k or "m2" in k})


def gold_yolov6_loss():
    """the reference's ComputeLoss (head/yolov6_head.py:315-754) on seeded head outputs: total, the four components and
    the gradient with respect to the raw outputs, for two parameter sets"""
    import contextlib, io
    from gen_golden_inputs import synth_yolov6_case
    m = ref_loader.load_yolov6_loss()
    res = {}
    for name, kw in dict(ciou=dict(iou_type="ciou"),
                         siou=dict(iou_type="siou", center_radius=1.5, iou_weight=2.0, cls_weight=0.5, reg_weight=2.5)).items():
        outs, t, _, _, _ = synth_yolov6_case()
        outs = [o.requires_grad_(True) for o in outs]
        cl = m.ComputeLoss(**kw)
        with contextlib.redirect_stdout(io.StringIO()):        # the reference prints the targets
            total, parts = cl([o * 1.0 for o in outs], t)      # (it decodes in place: hand it non-leaf tensors)
        total.sum().backward()
        res[name + "_total"] = total.detach().numpy()
        res[name + "_parts"] = parts.numpy()
        res[name + "_grad"] = torch.cat([o.grad.reshape(o.shape[0], -1, o.shape[-1]) for o in outs], 1).numpy()
        res[name + "_targets_after"] = t.numpy()
    np.savez_compressed(os.path.join(OUT, "yolov6_loss.npz"), **res)
    print("yolov6 loss:", {k: v for k, v in res.items() if "parts" in k or "total" in k})


def gold_random_perspective():
    """the reference's own random_perspective (data/transforms/data_augment.py:31-101) with its random.uniform draws fixed:
    the matrix it hands to cv2.warpAffine (captured by a spy in place of cv2), the output size and the filtered labels"""
    m = ref_loader.load_data_augment()
    draws = [(3.7, 0.8, 1.2, -0.7, 0.45, 0.55), (-9.5, 1.45, -2.0, 2.0, 0.6, 0.4), (0.0, 1.0, 0.0, 0.0, 0.5, 0.5), (10.0, 0.5, 0.3, 0.1, 0.41, 0.59)]
    res = {}
    orig_uniform, orig_warp = m.random.uniform, sys.modules["cv2"].warpAffine
    try:
        for k, d in enumerate(draws):
            r = np.random.RandomState(100 + k)
            hw, n = (1200, 1400), 12
            r.randint(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)           # (the test's image draw: keeps the streams aligned)
            x1 = r.uniform(0, hw[1] - 50, n); y1 = r.uniform(0, hw[0] - 50, n)
            t = np.stack([x1, y1, x1 + r.uniform(3, 400, n), y1 + r.uniform(3, 400, n), r.randint(0, 80, n).astype(np.float64)], 1)
            seq = list(d)
            m.random.uniform = lambda a, b, _s=seq: _s.pop(0)
            seen = {}

            def spy(img, M, dsize, borderValue=(0, 0, 0)):
                seen["M"], seen["dsize"] = np.array(M), dsize
                return np.zeros((dsize[1], dsize[0], 3), np.uint8)
            sys.modules["cv2"].warpAffine = spy
            img = np.zeros((hw[0], hw[1], 3), np.uint8)
            _, lab = m.random_perspective(img, t.copy(), degrees=10, translate=0.1, scale=(0.5, 1.5), shear=2.0,
                                          border=[-hw[0] // 4, -hw[1] // 4])
            M3 = np.eye(3)
            M3[:2] = seen["M"]
            res[f"M{k}"], res[f"labels{k}"], res[f"wh{k}"] = M3, lab, np.array(seen["dsize"])
    finally:
        m.random.uniform, sys.modules["cv2"].warpAffine = orig_uniform, orig_warp
    np.savez_compressed(os.path.join(OUT, "random_perspective.npz"), **res)
    print("random_perspective:", [len(res[f"labels{k}"]) for k in range(4)], "labels kept")


def gold_distortion():
    """the reference's own YOLOFDistortTransform.apply_image (data/transforms/transform.py:250-308) on seeded uint8 images
    with numpy's global stream seeded: the distorted image (float32 in the reference; stored as uint8 - same integers) and
    the state of the stream afterwards (one more uniform draw: pins HOW MANY variates the transform consumed).  cv2.cvtColor
    is oracle/augment_oracle.py's restatement (no cv2 here): this pins the reference's numpy arithmetic, dtype rules and
    draws, not OpenCV's colour conversion."""
    m = ref_loader.load_transforms()
    res = {}
    for k, (hw, seed) in enumerate((((33, 47), 3), ((64, 40), 4), ((21, 90), 5), ((50, 50), 6))):
        img = np.random.RandomState(200 + k).randint(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
        if k == 3:
            img[:, :25] = img[:, :25, :1]                            # grey pixels (S = 0) and a flat region
            img[30:, :] = 255
        np.random.seed(seed)
        out = m.YOLOFDistortTransform(hue=0.1, saturation=1.5, exposure=1.5).apply_image(img)
        assert out.dtype == np.float32 and np.array_equal(out, np.floor(out))
        res[f"out{k}"], res[f"next{k}"] = out.astype(np.uint8), np.float64(np.random.uniform())
    np.savez_compressed(os.path.join(OUT, "distortion.npz"), **res)
    print("distortion:", {k: (v.shape if v.ndim else float(v)) for k, v in res.items()})


def gold_pil_resize():
    """Pillow's own Image.resize(BILINEAR) - what detectron2's ResizeTransform.apply_image runs for a uint8 image inside
    T.ResizeShortestEdge (the first entry of build_normal_augmentation, data/detection_utils.py:37-86) - on seeded images:
    up- and down-scaling, one axis unchanged, both unchanged, a real-size COCO shape.  Pillow is a third-party dependency of
    the reference's detectron2 (un-vendored); the version installed in this image made these (`pil_version`)."""
    import PIL
    from PIL import Image
    r = np.random.RandomState(77)
    res = {"pil_version": np.array(PIL.__version__)}
    cases = [(37, 53, 20, 29), (37, 53, 74, 106), (40, 60, 40, 31), (40, 60, 17, 60), (33, 47, 33, 47), (50, 80, 31, 50),
             (23, 31, 64, 80), (48, 64, 13, 17), (30, 30, 7, 91), (120, 160, 152, 203)]
    for k, (h, w, nh, nw) in enumerate(cases):
        img = r.randint(0, 256, (h, w, 3), dtype=np.uint8)
        if k % 3 == 2:
            img = (np.linspace(0, 255, h * w * 3).reshape(h, w, 3)).astype(np.uint8)        # a smooth ramp: rounding ties
        res[f"src{k}"] = img
        res[f"size{k}"] = np.array([nh, nw])
        res[f"out{k}"] = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    np.savez_compressed(os.path.join(OUT, "pil_resize.npz"), **res)


def gold_jpeg():
    """JPEG files written AND decoded by the Pillow of this image (libjpeg-turbo underneath): what detectron2's
    utils.read_image hands the reference's mapper (data/dataset_mapper.py:646-648) - `rgb` = Image.open(f).convert("RGB"),
    `bgr` = the same after the EXIF transpose, channels reversed (read_image(..., format="BGR")).  Qualities, chroma
    sub-samplings, a grey file, optimised Huffman tables, restart intervals, EXIF orientations 3 / 6 / 8, two progressive files."""
    import io
    import PIL
    from PIL import Image, ImageOps
    r = np.random.RandomState(123)

    def smooth(h, w):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 100 * np.sin(xx / 9.0 + yy / 17.0), 127 + 100 * np.cos(xx / 13.0), 127 + 100 * np.sin(yy / 7.0)], -1)
        return np.clip(base + r.randint(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)
    specs = [((48, 64), dict(quality=75, subsampling=2)), ((33, 47), dict(quality=90, subsampling=1)), ((40, 40), dict(quality=50, subsampling=0)),
             ((17, 23), dict(quality=95, subsampling=2)), ((64, 96), dict(quality=85, subsampling=2, optimize=True)),
             ((64, 96), dict(quality=85, subsampling=2, restart_marker_blocks=3)), ((50, 3), dict(quality=80, subsampling=2)),
             ((40, 60), "grey"), ((40, 56), 3), ((40, 56), 6), ((40, 56), 8), ((96, 128), dict(quality=92, subsampling=2)),
             ((64, 96), dict(quality=85, subsampling=2, progressive=True)), ((33, 47), dict(quality=60, subsampling=0, progressive=True))]
    res = {"pil_version": np.array(PIL.__version__)}
    for k, ((h, w), kw) in enumerate(specs):
        buf = io.BytesIO()
        if kw == "grey":
            Image.fromarray(smooth(h, w)[..., 0]).save(buf, format="JPEG", quality=80)
        elif isinstance(kw, int):
            ex = Image.Exif()
            ex[0x0112] = kw
            Image.fromarray(smooth(h, w)).save(buf, format="JPEG", quality=90, exif=ex.tobytes())
        else:
            Image.fromarray(smooth(h, w)).save(buf, format="JPEG", **kw)
        data = buf.getvalue()
        res[f"file{k}"] = np.frombuffer(data, np.uint8)
        res[f"rgb{k}"] = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        res[f"bgr{k}"] = np.ascontiguousarray(np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(data))).convert("RGB"))[:, :, ::-1])
    np.savez_compressed(os.path.join(OUT, "jpeg_decode.npz"), **res)


def gold_bifpn():
    """the reference's BiFPN (neck/bifpn.py:307-395) over seeded C3..C5 maps, fp32: p3..p7, the gradients with respect to
    the inputs, every edge weight / GroupNorm parameter, and two convolution weights; dense and separable variants"""
    from gen_golden_inputs import BIFPN_CASES, synth_bifpn_case, bifpn_state_dict
    m = ref_loader.load_bifpn()

    class Feats(m.Backbone):
        def __init__(self, chans):
            super().__init__()
            self.chans = chans

        def output_shape(self):
            return {f"res{i + 3}": types.SimpleNamespace(channels=c, stride=8 << i) for i, c in enumerate(self.chans)}

        def forward(self, x):
            return x

    res = {}
    for name, kw in BIFPN_CASES.items():
        feats, gos = synth_bifpn_case(out_channels=kw["out_channels"])
        net = m.BiFPN(cfg=None, bottom_up=Feats([v.shape[1] for v in feats.values()]), in_features=list(feats.keys()),
                      norm="GN", num_levels=5, **kw)
        net.load_state_dict(bifpn_state_dict(net))
        xs = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
        out = net(xs)
        sum((out[k] * gos[k]).sum() for k in out).backward()
        res[name + "_keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items()])
        for k, v in out.items():
            res[f"{name}_out_{k}"] = v.detach().numpy()
        for k, v in xs.items():
            res[f"{name}_dx_{k}"] = v.grad.numpy()
        big = [k for k, p in net.named_parameters() if p.dim() == 4]
        keep = {big[0], big[-1]}
        for k, p in net.named_parameters():
            if p.dim() == 1 or k in keep:
                res[f"{name}_grad_{k}"] = p.grad.numpy()
        print("bifpn", name, {k: float(v.abs().mean()) for k, v in out.items()}, len(list(net.parameters())), "parameters")
    np.savez_compressed(os.path.join(OUT, "bifpn.npz"), **res)


def gold_encoder_layer():
    """the reference's own TransformerEncoderLayer (backbone/detr_backbone.py:135-194), eval mode (dropout off), fp32"""
    import importlib
    from gen_golden_inputs import synth_encoder_case, encoder_state_dict
    ref_loader.load()
    m = importlib.import_module("yolov7.modeling.backbone.detr_backbone")
    res = {}
    for name, pre in (("post", False), ("pre", True)):
        layer = m.TransformerEncoderLayer(256, 8, 2048, dropout=0.1, normalize_before=pre)
        layer.load_state_dict(encoder_state_dict())
        layer.eval()
        src, pos, mask, go = synth_encoder_case()
        x = src.clone().requires_grad_(True)
        out = layer(x, src_key_padding_mask=mask, pos=pos)
        out.backward(go)
        res[name + "_out"] = out.detach().numpy()
        res[name + "_dsrc"] = x.grad.numpy()
        for k, p in layer.named_parameters():   # large matrices: every 16th row keeps the fixture small
            res[f"{name}_g:{k}"] = (p.grad[::16] if p.dim() == 2 else p.grad).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "encoder_layer.npz"), **res)
    print("encoder_layer:", {k: float(np.abs(v).mean()) for k, v in res.items() if "out" in k or "dsrc" in k})


def gold_set_criterion():
    """the reference's own SetCriterion (meta_arch/detr.py:475-647) with its own HungarianMatcher, DETR weights
    (config.py:214-219: giou 2, l1 5, eos 0.1, deep supervision = aux outputs) + autograd of the weighted total"""
    import detr_oracle as D
    r = ref_loader.load()
    det = ref_loader.load_detr()
    res = {}
    cases = dict(a=(3, 100, 91, 81, None), b=(2, 100, 80, 82, [100, 0]), c=(2, 16, 20, 83, [30, 7]))
    for name, (bs, nq, ncls, seed, sizes) in cases.items():
        outs, targets = D.synth_detr_levels(bs, nq, ncls, seed, levels=3, sizes=sizes)
        leaves = [(l.clone().requires_grad_(True), b.clone().requires_grad_(True)) for l, b in outs]
        outputs = {"pred_logits": leaves[-1][0], "pred_boxes": leaves[-1][1],
                   "aux_outputs": [{"pred_logits": l, "pred_boxes": b} for l, b in leaves[:-1]]}
        wd = {"loss_ce": 1.0, "loss_bbox": 5.0, "loss_giou": 2.0}
        wd.update({k + f"_{i}": v for i in range(2) for k, v in list(wd.items())[:3]})
        crit = det.SetCriterion(ncls, r.detr_utils.HungarianMatcher(1.0, 5.0, 2.0), wd, 0.1, ["labels", "boxes", "cardinality"])
        ld = crit(outputs, targets)
        total = sum(ld[k] * wd[k] for k in ld if k in wd)
        total.backward()
        for k, v in ld.items():
            res[f"{name}:{k}"] = np.float32(v.detach().item())
        res[f"{name}:total"] = np.float32(total.item())
        for i, (l, b) in enumerate(leaves):
            res[f"{name}:dlogits{i}"], res[f"{name}:dboxes{i}"] = l.grad.numpy(), b.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "set_criterion.npz"), **res)
    print("set_criterion:", {k: float(v) for k, v in res.items() if k.startswith("a:") and v.ndim == 0})


def gold_pos_embed():
    """the reference's own PositionEmbeddingSine (backbone/detr_backbone.py:309-375) on seeded padding masks"""
    import importlib
    import types
    ref_loader.load()
    m = importlib.import_module("yolov7.modeling.backbone.detr_backbone")
    g = torch.Generator().manual_seed(91)
    B, H, W = 3, 19, 25
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    mask[1, :, 18:] = True
    mask[1, 15:, :] = True
    mask[2, :, 24:] = True
    res = {"mask": mask.numpy()}
    for name, kw in dict(detr=dict(num_pos_feats=128, normalize=True), raw=dict(num_pos_feats=64),
                         centered=dict(num_pos_feats=32, normalize=True, centered=True, temperature=20)).items():
        pe = m.PositionEmbeddingSine(**kw)
        out = pe(types.SimpleNamespace(tensors=torch.zeros(B, 1, H, W), mask=mask))
        res[name] = out.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "pos_embed.npz"), **res)
    print("pos_embed:", {k: v.shape for k, v in res.items()})


def gold_detr():
    """the reference's own DETR module (meta_arch/detr.py:406-472) around its own Transformer / MLP /
    PositionEmbeddingSine, with a stub backbone handing over a seeded feature map: logits and boxes of every decoder
    level, and the gradients of a seeded linear functional of them"""
    import importlib
    from gen_golden_inputs import synth_detr_case, seeded_state_dict, StubBackbone, SimpleNested
    ref_loader.load()
    det = ref_loader.load_detr()
    tb = importlib.import_module("yolov7.modeling.backbone.detr_backbone")
    feat, mask = synth_detr_case()
    x = feat.clone().requires_grad_(True)
    pos = tb.PositionEmbeddingSine(128, normalize=True)(SimpleNested(x, mask))
    tr = tb.Transformer(256, 8, 2, 2, 512, 0.1, normalize_before=False, return_intermediate_dec=True)
    net = det.DETR(StubBackbone(x, mask, pos, SimpleNested), tr, num_classes=20, num_queries=40, aux_loss=True)
    net.load_state_dict(seeded_state_dict(net))
    net.eval()
    out = net(SimpleNested(x, mask))
    logits = torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]])
    boxes = torch.stack([a["pred_boxes"] for a in out["aux_outputs"]] + [out["pred_boxes"]])
    g = torch.Generator().manual_seed(102)
    gl = torch.randn(logits.shape, generator=g)
    gb = torch.randn(boxes.shape, generator=g)
    ((logits * gl).sum() + (boxes * gb).sum()).backward()
    res = {"logits": logits.detach().numpy(), "boxes": boxes.detach().numpy(), "dfeat": x.grad.numpy()}
    for k in ("query_embed.weight", "class_embed.weight", "class_embed.bias", "bbox_embed.layers.2.weight",
              "bbox_embed.layers.2.bias", "bbox_embed.layers.0.weight", "input_proj.weight", "input_proj.bias"):
        res["g:" + k] = dict(net.named_parameters())[k].grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "detr_module.npz"), **res)
    print("detr_module:", {k: v.shape for k, v in res.items()})


def gold_detr_meta():
    """the reference's own `Detr` META ARCH (meta_arch/detr.py:33-279) executed by path: its preprocess_image /
    MaskedBackboneTraceFriendly / Joiner / DETR / Transformer / SetCriterion / HungarianMatcher / inference, around the
    CPU restatement of detectron2's ResNet-50 (resnet_oracle.R50Module standing in for the un-vendored
    detectron2.modeling.build_backbone; ImageList / Instances / Boxes / detector_postprocess from the product's d2 shim,
    which restates the same un-vendored classes).  2 encoder + 2 decoder layers, 30 queries, dropout 0 (no parity target
    exists for torch's dropout stream).  Stores the training loss dict and the eval-mode logits / boxes / detections."""
    import importlib
    import resnet_oracle as R
    from gen_golden_inputs import seeded_tensor_dict, synth_detr_batch
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from yolov7_d2_amd import d2shim, detr_r50_cfg
    ref_loader.load()
    det = ref_loader.load_detr()
    det.build_backbone = lambda cfg: R.R50Module(50, cfg.MODEL.RESNETS.OUT_FEATURES, cfg.MODEL.RESNETS.STRIDE_IN_1X1)
    det.ImageList, det.Instances, det.Boxes = d2shim.ImageList, d2shim.Instances, d2shim.Boxes
    det.detector_postprocess = d2shim.detector_postprocess
    cfg = detr_r50_cfg(device="cpu")
    cfg.MODEL.DETR.ENC_LAYERS, cfg.MODEL.DETR.DEC_LAYERS, cfg.MODEL.DETR.NUM_OBJECT_QUERIES = 2, 2, 30
    cfg.MODEL.DETR.DROPOUT = 0.0
    cfg.MODEL.YOLO.CONF_THRESHOLD = 0.02
    torch.manual_seed(0)
    model = det.Detr(cfg)
    sd = model.state_dict()
    model.load_state_dict(seeded_tensor_dict({k: v.shape for k, v in sd.items()}, seed=203), strict=False)
    batch = synth_detr_batch()
    inputs = [dict(image=b["image"], instances=d2shim.Instances(b["size"], gt_boxes=d2shim.Boxes(b["boxes"]),
                                                                gt_classes=b["classes"])) for b in batch]
    model.train()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):       # (the reference prints the loss dict every step)
        losses = model(inputs)
    res = {"loss:" + k: np.float32(v.detach()) for k, v in losses.items()}
    res["loss_keys"] = np.array(sorted(losses.keys()))
    model.eval()
    with torch.no_grad():
        images = model.preprocess_image(inputs)
        out = model.detr(images)
        dets = model(inputs)
    res["eval_logits"] = out["pred_logits"].numpy()
    res["eval_boxes"] = out["pred_boxes"].numpy()
    for i, d_ in enumerate(dets):
        inst = d_["instances"]
        res[f"det{i}_boxes"] = inst.pred_boxes.tensor.numpy()
        res[f"det{i}_scores"] = inst.scores.numpy()
        res[f"det{i}_classes"] = inst.pred_classes.numpy()
    res["state_keys"] = np.array(sorted(sd.keys()))
    np.savez_compressed(os.path.join(OUT, "detr_meta.npz"), **res)
    print("detr_meta:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in res.items() if not k.startswith("state")},
          "ndet", [len(d_["instances"]) for d_ in dets])


def gold_detr_real():
    """the reference's own `Detr` meta-arch (meta_arch/detr.py:33-279, by path) at the REAL configuration of BASELINE
    configs[3]: 6 + 6 layers, 100 queries, deep supervision, 800 x 1333 and 768 x 1205 images in one padded batch; dropout 0
    (torch's dropout stream has no parity target).  Stores the 25-entry loss dict of the training forward, the eval logits /
    boxes and grad_signature() of every trainable parameter's gradient of the weighted loss sum."""
    import contextlib, io
    import resnet_oracle as R
    from gen_golden_inputs import seeded_tensor_dict, synth_detr_batch, grad_signature
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from yolov7_d2_amd import d2shim, detr_r50_cfg
    ref_loader.load()
    det = ref_loader.load_detr()
    det.build_backbone = lambda cfg: R.R50Module(50, cfg.MODEL.RESNETS.OUT_FEATURES, cfg.MODEL.RESNETS.STRIDE_IN_1X1)
    det.ImageList, det.Instances, det.Boxes = d2shim.ImageList, d2shim.Instances, d2shim.Boxes
    det.detector_postprocess = d2shim.detector_postprocess
    cfg = detr_r50_cfg(device="cpu")
    cfg.MODEL.DETR.DROPOUT = 0.0
    torch.manual_seed(0)
    model = det.Detr(cfg)
    sd = model.state_dict()
    model.load_state_dict(seeded_tensor_dict({k: v.shape for k, v in sd.items()}, seed=207), strict=False)
    # conditioning: the seeded ResNet's res5 activations are ~1e2 per channel, input_proj would hand the encoder tokens of
    # norm 1.4e4 - attention then is one-hot and the gradient a lottery of arg-max flips (measured: two boundary tokens
    # carry 99 % of d src).  A trained network's tokens have norm O(10): scale input_proj accordingly
    with torch.no_grad():
        model.detr.input_proj.weight.mul_(1e-3)
    batch = synth_detr_batch(seed=211, sizes=((800, 1333), (768, 1205)))
    inputs = [dict(image=b["image"], instances=d2shim.Instances(b["size"], gt_boxes=d2shim.Boxes(b["boxes"]),
                                                                gt_classes=b["classes"])) for b in batch]
    model.train()
    feats = {}
    def keep(m, i, o):
        for k, v in o.items():
            v.retain_grad()
            feats[k] = v
    hk = model.detr.backbone[0].backbone.register_forward_hook(keep)

    def keep_src(m, args):
        args[0].retain_grad()
        feats["src"] = args[0]
    hk2 = model.detr.transformer.register_forward_pre_hook(keep_src)
    # the matcher's answers in call order (last decoder level, then the aux levels 0..4): near ties at random
    # initialisation make the assignment - and with it which two queries per image carry the box gradients - a coin toss
    # under bf16 noise, so the gradient comparison is made with these assignments forced
    matches = []
    orig_match = model.criterion.matcher.forward

    def rec(outputs, targets):
        r = orig_match(outputs, targets)
        matches.append([(i.clone(), j.clone()) for i, j in r])
        return r
    model.criterion.matcher.forward = rec
    with contextlib.redirect_stdout(io.StringIO()):
        losses = model(inputs)
    model.criterion.matcher.forward = orig_match
    hk.remove()
    hk2.remove()
    res = {"loss:" + k: np.float32(v.detach()) for k, v in losses.items()}
    res["loss_keys"] = np.array(sorted(losses.keys()))
    res["n_match_calls"] = np.int64(len(matches))
    for c, m_ in enumerate(matches):
        for b_, (i, j) in enumerate(m_):
            res[f"match:{c}:{b_}:q"], res[f"match:{c}:{b_}:t"] = i.numpy().astype(np.int64), j.numpy().astype(np.int64)
    total = sum(v for k, v in losses.items() if k in model.criterion.weight_dict)
    total.backward()
    named = [(n, p.grad) for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
    for n, v in grad_signature(named).items():
        res["gsig:" + n] = v
    res["dfsig:src"] = grad_signature([("dfeat:src", feats.pop("src").grad)])["dfeat:src"]   # gradient of the transformer's input
    for n, v in grad_signature([("feat:" + k, v) for k, v in feats.items()]).items():     # the backbone's output maps (NCHW)
        res["fsig:" + n[5:]] = v
    res["dfsig:res5"] = grad_signature([("dfeat:res5", feats["res5"].grad)])["dfeat:res5"]  # and the gradient that enters it
    model.eval()
    with torch.no_grad():
        out = model.detr(model.preprocess_image(inputs))
    res["eval_logits"] = out["pred_logits"].numpy()
    res["eval_boxes"] = out["pred_boxes"].numpy()
    np.savez_compressed(os.path.join(OUT, "detr_real.npz"), **res)
    print("detr_real: losses", {k: round(float(v.detach()), 4) for k, v in losses.items() if "_" not in k[-2:]}, "params", len(named))


def gold_sparseinst():
    """the reference's own InstanceContextEncoder + GroupIAMDecoder + SparseInstCriterion / SparseInstMatcher (loaded by
    path) on seeded ResNet features and bitmask targets: encoder output, decoder outputs, matcher indices, the four
    weighted losses, and the gradients of their sum with respect to the input features and every parameter (norms + two
    full tensors)"""
    import types
    from gen_golden_inputs import seeded_tensor_dict, synth_sparseinst_case
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from yolov7_d2_amd import sparse_inst_r50_giam_cfg
    si = ref_loader.load_sparseinst()
    cfg = sparse_inst_r50_giam_cfg(device="cpu")
    shapes = {n: types.SimpleNamespace(channels=c, stride=s) for n, c, s in (("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    torch.manual_seed(0)
    enc = si.encoder.InstanceContextEncoder(cfg, shapes)
    dec = si.decoder.GroupIAMDecoder(cfg)
    crit = si.loss.SparseInstCriterion(cfg, si.loss.SparseInstMatcher(cfg))
    net = torch.nn.ModuleDict(dict(encoder=enc, decoder=dec))
    from gen_golden_inputs import sparseinst_spread
    net.load_state_dict(sparseinst_spread(seeded_tensor_dict({k: v.shape for k, v in net.state_dict().items()}, seed=303)))
    feats, targets, input_shape = synth_sparseinst_case()
    fin = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    e = enc(fin)
    out = dec(e)
    class _BM:                       # BitMasks stand-in: .tensor + len()
        def __init__(self, t): self.tensor = t
        def __len__(self): return self.tensor.shape[0]
    tg = [dict(labels=t["labels"], masks=_BM(t["masks"])) for t in targets]
    indices = crit.matcher(out, tg, input_shape)
    with torch.no_grad():      # how decisive is the matching?  (margin of the chosen entries over the column medians)
        pm = out["pred_masks"][0].flatten(1); tm = torch.nn.functional.interpolate(
            si.loss.nested_masks_from_list([t["masks"].tensor for t in tg], input_shape).tensors[:, None], size=out["pred_masks"].shape[-2:],
            mode="bilinear", align_corners=False).squeeze(1).flatten(1)[:3]
        sc = si.loss.dice_score(pm, tm)
        print("dice score image 0: col max", sc.max(0).values.tolist(), "col 2nd", sc.topk(2, 0).values[1].tolist())
    losses = crit(out, tg, input_shape)
    total = sum(losses.values())
    total.backward()
    res = {"enc_out": e.detach().numpy()[:, ::8], "pred_logits": out["pred_logits"].detach().numpy(),
           "pred_scores": out["pred_scores"].detach().numpy(), "pred_masks": out["pred_masks"].detach().numpy()[:, ::5, ::2, ::2]}
    for b, (i, j) in enumerate(indices):
        res[f"match_i{b}"], res[f"match_j{b}"] = i.numpy(), j.numpy()
    for k, v in losses.items():
        res["loss:" + k] = np.float32(v.detach())
    for k, v in fin.items():
        res["dfeat_norm:" + k] = np.float32(v.grad.norm())
    names = sorted(dict(net.named_parameters()).keys())
    res["param_names"] = np.array(names)
    res["param_grad_norms"] = np.array([float(dict(net.named_parameters())[n].grad.norm()) for n in names], dtype=np.float32)
    for n in ("decoder.inst_branch.mask_kernel.weight", "encoder.fusion.weight"):
        res["g:" + n] = dict(net.named_parameters())[n].grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "sparseinst.npz"), **res)
    print("sparseinst:", {k: float(v) for k, v in losses.items()}, [(i.tolist(), j.tolist()) for i, j in indices])


def gold_sparseinst_real():
    """gold_sparseinst at the size configs[4] runs: the reference's own InstanceContextEncoder + GroupIAMDecoder (100
    instance queries) + SparseInstCriterion / Matcher on the res3 / res4 / res5 maps of a 640 x 640 batch (80 x 80 x 512,
    40 x 40 x 1024, 20 x 20 x 2048; second image 616 x 608 inside the padded batch) and bitmask targets.  Stores the class
    logits / objectness in full, the matcher's indices, the four weighted losses, and grad_signature() fingerprints of the
    encoder output, the mask logits, d feature maps and every parameter gradient."""
    import types
    from gen_golden_inputs import seeded_tensor_dict, synth_sparseinst_case, sparseinst_spread, grad_signature
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from yolov7_d2_amd import sparse_inst_r50_giam_cfg
    si = ref_loader.load_sparseinst()
    cfg = sparse_inst_r50_giam_cfg(device="cpu")
    shapes = {n: types.SimpleNamespace(channels=c, stride=s) for n, c, s in (("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    torch.manual_seed(0)
    enc = si.encoder.InstanceContextEncoder(cfg, shapes)
    dec = si.decoder.GroupIAMDecoder(cfg)
    crit = si.loss.SparseInstCriterion(cfg, si.loss.SparseInstMatcher(cfg))
    net = torch.nn.ModuleDict(dict(encoder=enc, decoder=dec))
    net.load_state_dict(sparseinst_spread(seeded_tensor_dict({k: v.shape for k, v in net.state_dict().items()}, seed=307)))
    feats, targets, input_shape = synth_sparseinst_case(seed=311, B=2, H=640, W=640)
    fin = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    e = enc(fin)
    out = dec(e)
    class _BM:
        def __init__(self, t): self.tensor = t
        def __len__(self): return self.tensor.shape[0]
    tg = [dict(labels=t["labels"], masks=_BM(t["masks"])) for t in targets]
    indices = crit.matcher(out, tg, input_shape)
    losses = crit(out, tg, input_shape)
    sum(losses.values()).backward()
    res = {"pred_logits": out["pred_logits"].detach().numpy(), "pred_scores": out["pred_scores"].detach().numpy()}
    for n, v in grad_signature([("enc_out", e.detach()), ("pred_masks", out["pred_masks"].detach())] +
                               [("dfeat:" + k, v.grad) for k, v in fin.items()]).items():
        res["sig:" + n] = v
    for b, (i, j) in enumerate(indices):
        res[f"match_i{b}"], res[f"match_j{b}"] = i.numpy(), j.numpy()
    for k, v in losses.items():
        res["loss:" + k] = np.float32(v.detach())
    for n, v in grad_signature([(n, p.grad) for n, p in net.named_parameters()]).items():
        res["gsig:" + n] = v
    np.savez_compressed(os.path.join(OUT, "sparseinst_real.npz"), **res)
    print("sparseinst_real:", {k: float(v) for k, v in losses.items()}, [(i.tolist(), j.tolist()) for i, j in indices],
          "|enc_out|", float(e.norm()), "|pred_masks|", float(out["pred_masks"].norm()))


def gold_sparseinst_inference():
    """the reference's own SparseInst.inference (meta_arch/sparseinst.py:173-234, with its jit-scripted rescoring_mask
    :24-27) on seeded decoder outputs (class logits, objectness, smooth mask-logit fields whose thresholded masks are
    blobs; some queries below the class threshold): per image the kept queries' rescored scores, classes and the
    thresholded masks at the requested output size - one image asks for an up-scaled, one for a down-scaled output, the
    second is smaller than the padded batch"""
    import types
    meta = ref_loader.load_sparseinst_meta()
    g = torch.Generator().manual_seed(909)
    B, Q, NC, mh, mw = 2, 40, 80, 16, 20
    max_shape = (128, 160)
    coarse = torch.randn(B, Q, 4, 5, generator=g)
    out = dict(pred_logits=torch.randn(B, Q, NC, generator=g) * 1.5 - 2.0, pred_scores=torch.randn(B, Q, 1, generator=g),
               pred_masks=torch.nn.functional.interpolate(coarse, size=(mh, mw), mode="bicubic", align_corners=False) * 3.0)
    image_sizes = [(128, 160), (104, 128)]
    batched_inputs = [dict(height=200, width=250), dict(height=80, width=100)]
    stub = types.SimpleNamespace(cls_threshold=0.3, mask_threshold=0.45)
    with torch.no_grad():
        results = meta.SparseInst.inference(stub, out, batched_inputs, max_shape, image_sizes)
    res = dict(pred_logits=out["pred_logits"].numpy(), pred_scores=out["pred_scores"].numpy(), pred_masks=out["pred_masks"].numpy(),
               max_shape=np.array(max_shape), image_sizes=np.array(image_sizes), out_sizes=np.array([[200, 250], [80, 100]]),
               cls_threshold=np.float32(stub.cls_threshold), mask_threshold=np.float32(stub.mask_threshold))
    for b, r in enumerate(results):
        res[f"scores{b}"], res[f"classes{b}"] = r.scores.numpy(), r.pred_classes.numpy()
        res[f"masks{b}"] = np.packbits(r.pred_masks.numpy().astype(np.uint8), axis=-1)
        res[f"mask_area{b}"] = r.pred_masks.flatten(1).sum(1).numpy()
    np.savez_compressed(os.path.join(OUT, "sparseinst_inference.npz"), **res)
    print("sparseinst_inference:", [(len(r.scores), float(r.scores.min()), float(r.scores.max()), float(r.pred_masks.float().mean())) for r in results])


def gold_sparseinst_onnx():
    """what the reference's SparseInst computes in EXPORT mode (meta_arch/sparseinst.py:127-162 under
    torch.onnx.is_in_onnx_export(): preprocess_inputs_onnx -> backbone -> encoder -> decoder -> inference_onnx), from its own
    InstanceContextEncoder / GroupIAMDecoder modules and its own `inference_onnx` (:236-345; rescoring_mask_batch :30-44), all
    loaded by path.  detectron2's ResNet is un-vendored: the backbone is oracle/resnet_oracle.py (PARITY UNPINNED for that
    part, as everywhere).  Batch of 1 (what export.py traces with) and batch of 2 (where inference_onnx's flattened top-k
    indexing selects from the first image only - the exported graph must reproduce that, not fix it).  64 x 96 input."""
    import types
    from gen_golden_inputs import sparseinst_onnx_weights, synth_sparseinst_images
    import resnet_oracle as RO
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from yolov7_d2_amd import sparse_inst_r50_giam_cfg
    si = ref_loader.load_sparseinst()
    meta = ref_loader.load_sparseinst_meta()
    cfg = sparse_inst_r50_giam_cfg(device="cpu")
    shapes = {n: types.SimpleNamespace(channels=c, stride=s) for n, c, s in (("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    torch.manual_seed(0)
    net = torch.nn.ModuleDict(dict(backbone=RO.R50Module(50, ("res3", "res4", "res5")), encoder=si.encoder.InstanceContextEncoder(cfg, shapes),
                                   decoder=si.decoder.GroupIAMDecoder(cfg)))
    sd = sparseinst_onnx_weights({k: v.shape for k, v in net.state_dict().items()})
    net.load_state_dict(sd)
    net.eval()
    H, W = 64, 96
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(cfg.MODEL.PIXEL_STD).view(1, 3, 1, 1)
    stub = types.SimpleNamespace(max_detections=cfg.MODEL.SPARSE_INST.MAX_DETECTIONS, mask_threshold=cfg.MODEL.SPARSE_INST.MASK_THRESHOLD)
    res = dict(hw=np.array([H, W]))
    import contextlib, io
    for B in (1, 2):
        seed = 500 + 100 * B
        while True:       # a case whose 50th and 51st scores are clearly apart (the top-k SET must not hinge on rounding; the
                          # order inside it may: the test matches rows, it does not compare them position by position)
            img = synth_sparseinst_images(B, H, W, seed)
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                x = (img.permute(0, 3, 1, 2) - mean) / std                       # preprocess_inputs_onnx (:122-125)
                out = net["decoder"](net["encoder"](net["backbone"](x)))
                ps = torch.sqrt(out["pred_logits"].sigmoid() * out["pred_scores"].sigmoid() + 1e-3).max(-1)[0]
                v = ps.sort(dim=1, descending=True)[0]
                if float((v[:, 49] - v[:, 50]).min()) > 5e-4:
                    masks, scores, labels = meta.SparseInst.inference_onnx(stub, out, img, (H, W))
                    break
            seed += 1
        res[f"seed{B}"] = np.int64(seed)
        res[f"masks{B}"] = np.packbits(masks.numpy().astype(np.uint8), axis=-1)
        res[f"mask_shape{B}"] = np.array(masks.shape)
        res[f"scores{B}"], res[f"labels{B}"] = scores.numpy(), labels.numpy()
    np.savez_compressed(os.path.join(OUT, "sparseinst_onnx.npz"), **res)
    print("sparseinst_onnx:", res["mask_shape1"], int(res["seed1"]), int(res["seed2"]), float(res["scores1"].max()), float(res["scores2"].min()), res["labels1"][0, :6])


def gold_detr_onnx():
    """the reference's own `Detr` (meta_arch/detr.py, by path) in EXPORT mode - `model.onnx_export = True; model(x)` with x
    [B, 3, H, W] as export.py:283-300 runs it: preprocess_input (:126-134), nested_tensor_from_tensor_list, the
    trace-friendly MaskedBackbone branch (:362-375), DETR.forward's (logits, boxes) of the last decoder layer (:458-459) and
    the [x0, y0, x1, y1, score, label] rows of :178-185.  6 + 6 layers, 100 queries, 64 x 96 input; backbone = the ResNet
    restatement (d2 un-vendored)."""
    import contextlib, io
    import resnet_oracle as R
    from gen_golden_inputs import detr_onnx_weights, synth_sparseinst_images
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from yolov7_d2_amd import d2shim, detr_r50_cfg
    ref_loader.load()
    det = ref_loader.load_detr()
    det.build_backbone = lambda cfg: R.R50Module(50, cfg.MODEL.RESNETS.OUT_FEATURES, cfg.MODEL.RESNETS.STRIDE_IN_1X1)
    det.ImageList, det.Instances, det.Boxes = d2shim.ImageList, d2shim.Instances, d2shim.Boxes
    det.detector_postprocess = d2shim.detector_postprocess
    cfg = detr_r50_cfg(device="cpu")
    torch.manual_seed(0)
    model = det.Detr(cfg)
    model.load_state_dict(detr_onnx_weights({k: v.shape for k, v in model.state_dict().items()}), strict=False)
    model.eval()
    model.onnx_export = True
    H, W = 64, 96
    res = dict(hw=np.array([H, W]))
    for B in (1, 2):
        x = synth_sparseinst_images(B, H, W, 900 + B).permute(0, 3, 1, 2).contiguous()
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            out = model(x)
        res[f"seed{B}"] = np.int64(900 + B)
        res[f"outs{B}"] = out.numpy()
    np.savez_compressed(os.path.join(OUT, "detr_onnx.npz"), **res)
    print("detr_onnx:", res["outs2"].shape, "scores", float(res["outs1"][..., 4].min()), float(res["outs1"][..., 4].max()),
          "labels", sorted(set(res["outs1"][0, :, 5].astype(int).tolist()))[:8])


def gold_transformer():
    """the reference's own Transformer (backbone/detr_backbone.py:25-65): 2 encoder + 2 decoder layers, d_model 256,
    8 heads, ffn 512, return_intermediate_dec, eval mode, fp32; post- and pre-norm"""
    import importlib
    from gen_golden_inputs import synth_transformer_case, seeded_state_dict
    ref_loader.load()
    m = importlib.import_module("yolov7.modeling.backbone.detr_backbone")
    res = {}
    for name, pre in (("post", False), ("pre", True)):
        net = m.Transformer(256, 8, 2, 2, 512, 0.1, normalize_before=pre, return_intermediate_dec=True)
        net.load_state_dict(seeded_state_dict(net))
        net.eval()
        src, mask, qe, pos = synth_transformer_case()
        x = src.clone().requires_grad_(True)
        q = qe.clone().requires_grad_(True)
        hs, mem = net(x, mask, q, pos)
        gh = torch.randn(hs.shape, generator=torch.Generator().manual_seed(73)).to(torch.bfloat16).float()
        (hs * gh).sum().backward()
        res[name + "_hs"] = hs.detach().numpy()
        res[name + "_mem"] = mem.detach().numpy()
        res[name + "_dsrc"] = x.grad.numpy()
        res[name + "_dquery"] = q.grad.numpy()
        for k, p in net.named_parameters():   # matrices: every 32nd row keeps the fixture small
            res[f"{name}_g:{k}"] = (p.grad[::32] if p.dim() == 2 else p.grad).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "transformer.npz"), **res)
    print("transformer:", {k: float(np.abs(v).mean()) for k, v in res.items() if "_g:" not in k})
    # the configuration DETR-R50 runs (6 + 6 layers, ffn 2048, 100 queries) on an 800 x 1333 batch's 25 x 42 map: fingerprints
    from gen_golden_inputs import grad_signature
    net = m.Transformer(256, 8, 6, 6, 2048, 0.1, normalize_before=False, return_intermediate_dec=True)
    net.load_state_dict(seeded_state_dict(net, seed=76))
    net.eval()
    src, mask, qe, pos = synth_transformer_case(B=2, H=25, W=42, Q=100, seed=75)
    x = src.clone().requires_grad_(True)
    q = qe.clone().requires_grad_(True)
    hs, mem = net(x, mask, q, pos)
    gh = torch.randn(hs.shape, generator=torch.Generator().manual_seed(77)).to(torch.bfloat16).float()
    (hs * gh).sum().backward()
    real = {"hs_last": hs[-1].detach().numpy(), "dquery": q.grad.numpy()}
    valid = ~mask.flatten(1)
    real["sig:mem_valid"] = grad_signature([("mem_valid", mem.detach().flatten(2).transpose(1, 2)[valid])])["mem_valid"]
    real["sig:hs"] = grad_signature([("hs", hs.detach())])["hs"]
    real["sig:dsrc"] = grad_signature([("dsrc", x.grad)])["dsrc"]
    real["sig:dsrc_valid"] = grad_signature([("dsrc_valid", x.grad.flatten(2).transpose(1, 2)[valid])])["dsrc_valid"]
    for k, v in grad_signature([(k, p.grad) for k, p in net.named_parameters()]).items():
        real["gsig:" + k] = v
    np.savez_compressed(os.path.join(OUT, "transformer_real.npz"), **real)
    print("transformer_real: |dsrc|", float(x.grad.norm()), "|dsrc valid|", real["sig:dsrc_valid"][0], "|hs|", float(hs.norm()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    gold_step()
    gold_onnx_layout()
    gold_dw_step()
    gold_tiny_step()
    gold_simota()
    gold_simota_l1()
    gold_postprocess()
    gold_hungarian()
    gold_iou_v6()
    gold_yolox_iou()
    gold_box_ops()
    gold_nms_family()
    gold_yolov6_loss()
    gold_bifpn()
    gold_random_perspective()
    gold_distortion()
    gold_pil_resize()
    gold_jpeg()
    gold_encoder_layer()
    gold_transformer()
    gold_pos_embed()
    gold_detr()
    gold_detr_meta()
    gold_detr_real()
    gold_sparseinst()
    gold_sparseinst_real()
    gold_sparseinst_inference()
    gold_sparseinst_onnx()
    gold_detr_onnx()
    gold_set_criterion()
