"""Training input pipeline of the YOLOX path on the GPU: a batch-level drop-in for what the reference
does per image on CPU workers with cv2 -

    MyDatasetMapper2.__call__, mosaic branch          yolov7/data/dataset_mapper.py:477-612
    random_perspective, box_candidates                yolov7/data/transforms/data_augment.py:15-101
    YOLOX.preprocess_image (pad with 114, label rows) yolov7/modeling/meta_arch/yolox.py:95-162

Decoded images live in HBM (`MosaicPool`, uint8 HWC); per batch the host draws the reference's random numbers in the
reference's order, does the reference's float64 label arithmetic (a few dozen boxes: numpy, as the reference), and the GPU
does every pixel: four resizes + pastes per sample in ONE launch, the affine warp + NCHW transpose + pad-to-batch in a
second - `GpuMosaicMapper.make_batch` returns exactly what `NativeTrainer.load_batch` / `feed` take (uint8 [B, 3, H, W],
float32 [B, 100, 5] rows (cls, cx, cy, w, h)).  At 2 600 images / s / GPU this replaces ~40 cv2 CPU workers per GPU.

Mixup (`ENABLE_MIXUP`, dataset_mapper.py:686-768) is a third launch that blends a resized / jittered / mirrored pool image
into the warped sample in place.  Further down in this file: the detectron2 `T.*` augmentations ahead of the mosaic
(`GpuFrontAugment`), the whole mapper call (`GpuDatasetMapper`), JPEG decoding (`GpuJpegDecoder`), DETR's mapper
(`GpuDetrMapper`).  No CPU path for the pixels: device tensors only.
"""
import ctypes as C
import math
import random

import numpy as np
import torch

from . import _lib as L

MOSAIC_DEFAULTS = dict(NUM_IMAGES=4, DEGREES=10.0, TRANSLATE=0.1, SCALE=[0.5, 1.5], MSCALE=[0.5, 1.5], SHEAR=2.0, PERSPECTIVE=0.0,
                       MOSAIC_WIDTH_RANGE=(512, 800), MOSAIC_HEIGHT_RANGE=(512, 800))        # yolov7/config.py:258-272


def _get(cfg, k):
    return cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)


def box_candidates(box1, box2, wh_thr=2, ar_thr=20, area_thr=0.2):
    """data_augment.py:15-28"""
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + 1e-16), h2 / (w2 + 1e-16))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + 1e-16) > area_thr) & (ar < ar_thr)


class MosaicPool:
    """the reference's `mosaic_pool` deque (dataset_mapper.py:404-405) with the decoded images resident on the device"""

    def __init__(self, device="cuda", capacity=1000):
        self.device, self.capacity = torch.device(device), capacity
        self.images, self.labels = [], []

    def append(self, image_hwc_u8, labels_xyxy_cls):
        img = torch.as_tensor(image_hwc_u8)
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("MosaicPool.append: uint8 [H, W, 3] image")
        self.images.append(img.to(self.device).contiguous())
        self.labels.append(np.asarray(labels_xyxy_cls, np.float64).reshape(-1, 5))
        if len(self.images) > self.capacity:
            self.images.pop(0)
            self.labels.pop(0)

    def __len__(self):
        return len(self.images)


class GpuMosaicMapper:
    def __init__(self, mosaic_cfg=None, device="cuda", max_boxes=100, pad_value=114, size_divisibility=32):
        cfg = dict(MOSAIC_DEFAULTS)
        if mosaic_cfg is not None:
            for k in MOSAIC_DEFAULTS:
                try:
                    cfg[k] = _get(mosaic_cfg, k)
                except (KeyError, AttributeError):
                    pass
        if cfg["NUM_IMAGES"] != 4 or cfg["PERSPECTIVE"] != 0.0:
            raise NotImplementedError("GpuMosaicMapper: 4-image mosaic with an affine warp (the reference's defaults)")
        self.cfg, self.device = cfg, torch.device(device)
        self.max_boxes, self.pad, self.divis = max_boxes, pad_value, size_divisibility

    # ---- the reference's random draws, in its order ----------------------------------------------------------------
    def draw(self, rng_np=np.random, rng_py=random):
        """dataset_mapper.py:505-520 (np.random.randint width, height; random.uniform yc, xc) then data_augment.py:45-62
        (random.uniform angle, scale, shear x, shear y, translate x, translate y)"""
        c = self.cfg
        w = int(rng_np.randint(c["MOSAIC_WIDTH_RANGE"][0], c["MOSAIC_WIDTH_RANGE"][1] + 1))
        h = int(rng_np.randint(c["MOSAIC_HEIGHT_RANGE"][0], c["MOSAIC_HEIGHT_RANGE"][1] + 1))
        if max(w / h, h / w) > 1.2:
            h = min(h, w)
            w = int(1.2 * h)
        dim = (h, w)
        yc = int(rng_py.uniform(0.5 * dim[0], 1.5 * dim[0]))
        xc = int(rng_py.uniform(0.5 * dim[1], 1.5 * dim[1]))
        a = rng_py.uniform(-c["DEGREES"], c["DEGREES"])
        s = rng_py.uniform(c["SCALE"][0], c["SCALE"][1])
        shx = rng_py.uniform(-c["SHEAR"], c["SHEAR"])
        shy = rng_py.uniform(-c["SHEAR"], c["SHEAR"])
        tx = rng_py.uniform(0.5 - c["TRANSLATE"], 0.5 + c["TRANSLATE"])
        ty = rng_py.uniform(0.5 - c["TRANSLATE"], 0.5 + c["TRANSLATE"])
        return dict(input_dim=dim, yc=yc, xc=xc, draws=(a, s, shx, shy, tx, ty))

    def draw_mixup(self, pool, input_dim, target_hw, rng_np=np.random, rng_py=random):
        """the draws of MyDatasetMapper2.mixup in its order (dataset_mapper.py:687-743): random.uniform(*MSCALE),
        random.uniform(0, 1) > 0.5, np.random.choice over the pool, then random.randint for the y and the x offset where the
        jittered image exceeds the target (height, width) of the warped sample"""
        jit = rng_py.uniform(*self.cfg["MSCALE"])
        flip = rng_py.uniform(0, 1) > 0.5
        idx = int(rng_np.choice(len(pool), 1)[0])
        oh, ow = int(input_dim[0] * jit), int(input_dim[1] * jit)
        ph, pw = max(oh, target_hw[0]), max(ow, target_hw[1])
        y_off = rng_py.randint(0, ph - target_hw[0] - 1) if ph > target_hw[0] else 0
        x_off = rng_py.randint(0, pw - target_hw[1] - 1) if pw > target_hw[1] else 0
        return dict(idx=idx, jit=jit, flip=bool(flip), x_off=x_off, y_off=y_off)

    @staticmethod
    def _mixup_labels(origin_labels, cp_labels, r, jit, flip, x_off, y_off, origin_hw, target_hw):
        """dataset_mapper.py:745-766: returns (labels, blended?)"""
        cp = np.array(cp_labels, np.float64).reshape(-1, 5).copy()
        oh, ow = origin_hw
        bo = cp[:, :4]
        ratio = r * jit
        bo[:, 0::2] = np.clip(bo[:, 0::2] * ratio + 0, 0, ow)            # adjust_box_anns (utils/boxes.py:381-384)
        bo[:, 1::2] = np.clip(bo[:, 1::2] * ratio + 0, 0, oh)
        if flip:
            bo[:, 0::2] = ow - bo[:, 0::2][:, ::-1]
        bt = bo.copy()
        bt[:, 0::2] = np.clip(bt[:, 0::2] - x_off, 0, target_hw[1])
        bt[:, 1::2] = np.clip(bt[:, 1::2] - y_off, 0, target_hw[0])
        keep = box_candidates(bo.T, bt.T, 5)
        if keep.sum() >= 1.0:
            return np.vstack((origin_labels.reshape(-1, 5), np.hstack((bt[keep], cp[keep, 4:5])))), True
        return origin_labels, False

    # ---- host geometry / labels (float64, the reference's operation order) ------------------------------------------
    @staticmethod
    def _placement(i, w, h, xc, yc, dim):
        H2, W2 = dim[0] * 2, dim[1] * 2
        if i == 0:
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b = w - (x2a - x1a), h - (y2a - y1a)
        elif i == 1:
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, W2), yc
            x1b, y1b = 0, h - (y2a - y1a)
        elif i == 2:
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(H2, yc + h)
            x1b, y1b = w - (x2a - x1a), 0
        else:
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, W2), min(H2, yc + h)
            x1b, y1b = 0, 0
        return (x1a, y1a, x2a, y2a), (x1b, y1b)

    @staticmethod
    def _matrix(canvas_hw, draws, border):
        a, s, shx, shy, tx, ty = draws
        height, width = canvas_hw[0] + border[0] * 2, canvas_hw[1] + border[1] * 2
        Cm = np.eye(3)
        Cm[0, 2] = -canvas_hw[1] / 2
        Cm[1, 2] = -canvas_hw[0] / 2
        R = np.eye(3)
        r = a * math.pi / 180.0                                   # cv2.getRotationMatrix2D(center=(0, 0))
        al, be = s * math.cos(r), s * math.sin(r)
        R[:2] = [[al, be, 0.0], [-be, al, 0.0]]
        S = np.eye(3)
        S[0, 1] = math.tan(shx * math.pi / 180)
        S[1, 0] = math.tan(shy * math.pi / 180)
        T = np.eye(3)
        T[0, 2] = tx * width
        T[1, 2] = ty * height
        return T @ S @ R @ Cm, width, height

    @staticmethod
    def _warp_labels(targets, M, s, width, height):
        n = len(targets)
        if not n:
            return targets
        xy = np.ones((n * 4, 3))
        xy[:, :2] = targets[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
        xy = xy @ M.T
        xy = xy[:, :2].reshape(n, 8)
        x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
        xy = np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1))).reshape(4, n).T
        xy[:, [0, 2]] = xy[:, [0, 2]].clip(0, width)
        xy[:, [1, 3]] = xy[:, [1, 3]].clip(0, height)
        i = box_candidates(box1=targets[:, :4].T * s, box2=xy.T)
        targets = targets[i]
        targets[:, :4] = xy[i]
        return targets

    @staticmethod
    def _invert(M):
        """the inversion cv2.warpAffine applies to a forward matrix (imgwarp.cpp)"""
        D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
        D = 1.0 / D if D != 0 else 0.0
        A11, A22 = M[1, 1] * D, M[0, 0] * D
        A12, A21 = -M[0, 1] * D, -M[1, 0] * D
        return [A11, A12, -A11 * M[0, 2] - A12 * M[1, 2], A21, A22, -A21 * M[0, 2] - A22 * M[1, 2]]

    # ---- one batch --------------------------------------------------------------------------------------------------
    def make_batch(self, pool, groups, params, mixups=None, float_src=False):
        """groups: B tuples of four pool indices (the current image first, then the three sampled ones); params: B dicts
        from draw(); mixups: None or B entries (None or a dict from draw_mixup()); float_src: the pool images are float32
        in the reference (its front ended with YOLOFRandomDistortion): cv2.resize's float path + truncation.  Returns (uint8 [B, 3, H, W] on the device, float32 [B, max_boxes, 5] (cls, cx, cy, w, h) on the
        device, per-sample (h, w))"""
        if self.device.type != "cuda":
            raise L.MI355Error("GpuMosaicMapper: the MI355X path needs a device (no CPU pixel path)")
        B = len(groups)
        lib = L.lib()
        dims = [p["input_dim"] for p in params]
        Hp = (max(d[0] for d in dims) + self.divis - 1) // self.divis * self.divis
        Wp = (max(d[1] for d in dims) + self.divis - 1) // self.divis * self.divis
        out = torch.full((B, 3, Hp, Wp), self.pad, dtype=torch.uint8, device=self.device)
        csz = [4 * d[0] * d[1] * 3 for d in dims]
        coff = np.concatenate([[0], np.cumsum(csz)])
        canvas = torch.full((int(coff[-1]),), self.pad, dtype=torch.uint8, device=self.device)
        paste = (L.mi_mosaic_paste_job * (4 * B))()
        warp = (L.mi_warp_job * B)()
        mix = (L.mi_mixup_job * B)()
        nmix = 0
        rows = np.zeros((B, self.max_boxes, 5), np.float32)
        for b, (grp, p) in enumerate(zip(groups, params)):
            dim, yc, xc = p["input_dim"], p["yc"], p["xc"]
            cbase = canvas.data_ptr() + int(coff[b])
            labels4 = []
            for i, idx in enumerate(grp):
                img, lab = pool.images[idx], pool.labels[idx]
                h0, w0 = img.shape[:2]
                scale = min(1. * dim[0] / h0, 1. * dim[1] / w0)
                w, h = int(w0 * scale), int(h0 * scale)
                (x1a, y1a, x2a, y2a), (x1b, y1b) = self._placement(i, w, h, xc, yc, dim)
                j = paste[4 * b + i]
                j.src, j.canvas = img.data_ptr(), cbase
                j.h0, j.w0, j.rh, j.rw, j.cw = h0, w0, h, w, dim[1] * 2
                j.x1a, j.y1a, j.x2a, j.y2a, j.x1b, j.y1b = x1a, y1a, max(x2a, x1a), max(y2a, y1a), x1b, y1b
                j.fsrc = int(bool(float_src))
                if lab.size > 0:
                    t = lab.copy()
                    padw, padh = x1a - x1b, y1a - y1b
                    t[:, 0] = scale * lab[:, 0] + padw
                    t[:, 1] = scale * lab[:, 1] + padh
                    t[:, 2] = scale * lab[:, 2] + padw
                    t[:, 3] = scale * lab[:, 3] + padh
                    labels4.append(t)
            if labels4:
                labels4 = np.concatenate(labels4, 0)
                np.clip(labels4[:, 0], 0, 2 * dim[1], out=labels4[:, 0])
                np.clip(labels4[:, 1], 0, 2 * dim[0], out=labels4[:, 1])
                np.clip(labels4[:, 2], 0, 2 * dim[1], out=labels4[:, 2])
                np.clip(labels4[:, 3], 0, 2 * dim[0], out=labels4[:, 3])
            else:
                labels4 = np.zeros((0, 5))
            M, width, height = self._matrix((dim[0] * 2, dim[1] * 2), p["draws"], [-dim[0] // 2, -dim[1] // 2])
            t = self._warp_labels(labels4, M, p["draws"][1], width, height)
            mx = mixups[b] if mixups is not None else None
            if mx is not None and len(t):                           # dataset_mapper.py:602: only with surviving labels
                im = pool.images[mx["idx"]]
                h0, w0 = im.shape[:2]
                r = min(dim[0] / h0, dim[1] / w0)
                oh, ow = int(dim[0] * mx["jit"]), int(dim[1] * mx["jit"])
                t, blended = self._mixup_labels(t, pool.labels[mx["idx"]], r, mx["jit"], mx["flip"], mx["x_off"], mx["y_off"],
                                                (oh, ow), (height, width))
                if blended:
                    mj = mix[nmix]
                    nmix += 1
                    mj.src, mj.out = im.data_ptr(), out.data_ptr() + b * 3 * Hp * Wp
                    mj.h0, mj.w0, mj.rh1, mj.rw1, mj.dh, mj.dw = h0, w0, int(h0 * r), int(w0 * r), dim[0], dim[1]
                    mj.oh, mj.ow, mj.flip, mj.x_off, mj.y_off = oh, ow, int(mx["flip"]), mx["x_off"], mx["y_off"]
                    mj.th, mj.tw, mj.Hp, mj.Wp = height, width, Hp, Wp
                    mj.fsrc = int(bool(float_src))
            t = t[: self.max_boxes]
            wj = warp[b]
            wj.canvas, wj.out = cbase, out.data_ptr() + b * 3 * Hp * Wp
            for q, v in enumerate(self._invert(M)):
                wj.minv[q] = v
            wj.ch, wj.cw, wj.h, wj.w, wj.Hp, wj.Wp, wj.border = dim[0] * 2, dim[1] * 2, height, width, Hp, Wp, self.pad
            if len(t):                                             # meta_arch/yolox.py:131-160: XYXY -> (cls, cx, cy, w, h)
                box = t[:, :4].astype(np.float32)
                rows[b, : len(t), 0] = t[:, 4]
                rows[b, : len(t), 1] = (box[:, 0] + box[:, 2]) / 2
                rows[b, : len(t), 2] = (box[:, 1] + box[:, 3]) / 2
                rows[b, : len(t), 3] = box[:, 2] - box[:, 0]
                rows[b, : len(t), 4] = box[:, 3] - box[:, 1]
        L.check(lib.mi_mosaic_jobs_layout(paste, 4 * B, warp, B), "mi_mosaic_jobs_layout")
        pb = paste[4 * B - 1].blk0 + ((paste[4 * B - 1].x2a - paste[4 * B - 1].x1a) * (paste[4 * B - 1].y2a - paste[4 * B - 1].y1a) + 255) // 256
        wb = warp[B - 1].blk0 + (warp[B - 1].w * warp[B - 1].h + 255) // 256
        mb = L.check(lib.mi_mixup_jobs_layout(mix, nmix), "mi_mixup_jobs_layout") if nmix else 0
        tab = torch.frombuffer(bytearray(bytes(paste) + bytes(warp) + bytes(mix)), dtype=torch.uint8).to(self.device)
        st = L.stream_ptr()
        L.check(lib.mi_mosaic_paste(tab.data_ptr(), 4 * B, pb, st), "mi_mosaic_paste")
        L.check(lib.mi_warp_affine_u8(tab.data_ptr() + C.sizeof(paste), B, wb, st), "mi_warp_affine_u8")
        if nmix:
            L.check(lib.mi_mixup_blend(tab.data_ptr() + C.sizeof(paste) + C.sizeof(warp), nmix, mb, st), "mi_mixup_blend")
        self._keep = (canvas, tab)                                 # alive until the stream has run the two launches
        return out, torch.from_numpy(rows).to(self.device, non_blocking=True), [(d[0], d[1]) for d in dims]


# ------------------------------------------------------------------------------------------------ the T.* front
FRONT_DEFAULTS = dict(MIN_SIZE_TRAIN=(416, 512, 608, 768), MAX_SIZE_TRAIN=800, MIN_SIZE_TRAIN_SAMPLING="choice",
                      HFLIP=True, HFLIP_PROB=0.5, VFLIP=True, VFLIP_PROB=0.5, SATURATION=False, BRIGHTNESS=False,
                      DISTORTION=False, DISTORTION_HUE=0.1, DISTORTION_SATURATION=1.5, DISTORTION_EXPOSURE=1.5,
                      SHIFT=True, SHIFT_PIXELS=32)
# (configs/coco/yolox_s.yaml:35-50 + yolov7/config.py:276-299: INPUT.RANDOM_FLIP_HORIZONTAL / _VERTICAL / SHIFT / DISTORTION
#  defaults; yolox_s.yaml:46-50 switches DISTORTION and both COLOR_JITTER entries on)


def front_cfg_from_cfg(cfg):
    """FRONT_DEFAULTS overridden by the INPUT.* keys `build_normal_augmentation` reads (yolov7/data/detection_utils.py:37-86;
    defaults of absent keys: yolov7/config.py:276-299) - so that `configs/coco/yolox_s.yaml` (DISTORTION + both COLOR_JITTER
    entries on) configures this front as it configures the reference's"""
    import ast
    inp = cfg.INPUT

    def get(node, path, default):
        for k in path.split("."):
            if not (hasattr(node, "get") and k in node):
                return default
            node = node[k]
        return ast.literal_eval(node) if isinstance(node, str) and node[:1] in "([" else node
    c = dict(FRONT_DEFAULTS)
    c.update(MIN_SIZE_TRAIN=tuple(get(inp, "MIN_SIZE_TRAIN", c["MIN_SIZE_TRAIN"])), MAX_SIZE_TRAIN=get(inp, "MAX_SIZE_TRAIN", c["MAX_SIZE_TRAIN"]),
             MIN_SIZE_TRAIN_SAMPLING=get(inp, "MIN_SIZE_TRAIN_SAMPLING", "choice"),
             HFLIP=bool(get(inp, "RANDOM_FLIP_HORIZONTAL.ENABLED", True)), HFLIP_PROB=get(inp, "RANDOM_FLIP_HORIZONTAL.PROB", 0.5),
             VFLIP=bool(get(inp, "RANDOM_FLIP_VERTICAL.ENABLED", True)), VFLIP_PROB=get(inp, "RANDOM_FLIP_VERTICAL.PROB", 0.5),
             SATURATION=bool(get(inp, "COLOR_JITTER.SATURATION", False)), BRIGHTNESS=bool(get(inp, "COLOR_JITTER.BRIGHTNESS", False)),
             DISTORTION=bool(get(inp, "DISTORTION.ENABLED", False)), DISTORTION_HUE=get(inp, "DISTORTION.HUE", 0.1),
             DISTORTION_SATURATION=get(inp, "DISTORTION.SATURATION", 1.5), DISTORTION_EXPOSURE=get(inp, "DISTORTION.EXPOSURE", 1.5),
             SHIFT=bool(get(inp, "SHIFT.ENABLED", True)), SHIFT_PIXELS=get(inp, "SHIFT.SHIFT_PIXELS", 32))
    return c


class GpuFrontAugment:
    """The detectron2 augmentation list every loaded image goes through before anything else
    (`MyDatasetMapper2._load_image_with_annos`, yolov7/data/dataset_mapper.py:642-683; `build_normal_augmentation`,
    yolov7/data/detection_utils.py:37-86): T.ResizeShortestEdge -> T.RandomFlip(horizontal) -> T.RandomFlip(vertical) ->
    YOLOFRandomShift, with `transform_instance_annotations` (detection_utils.py:158-190) for the boxes - and, with the mosaic
    off (after INPUT.MOSAIC_AND_MIXUP.DISABLE_AT_ITER, or mosaic_flag 0), the WHOLE mapper: dataset_mapper.py:615-640 then
    only builds Instances and drops empty boxes.  The host draws the reference's random numbers in the reference's order and
    does its float64 box arithmetic; the pixels (Pillow's 8-bit bilinear resampling, the flips, the shift) are two launches
    for any number of images (mi_pil_resize_h / _v), RandomSaturation / RandomBrightness (INPUT.COLOR_JITTER; detectron2's
    BlendTransform in numpy's fp64 / fp32 arithmetic) and YOLOFRandomDistortion (DISTORTION: cv2's 8-bit RGB <-> HSV
    conversions around three float32 scalings, data/transforms/transform.py:250-308; OpenCV's published integer / float
    algorithms restated, csrc/pil_resize_core.h pil_distort) included, per pixel between the flips and the shift.  After the
    distortion the reference's image is float32 (same integers): `float_images` tells the mosaic / mixup launches to take
    cv2.resize's float path for these sources (GpuMosaicMapper.make_batch(float_src=...))."""

    def __init__(self, cfg=None, device="cuda", max_boxes=100, pad_value=114, size_divisibility=32):
        c = dict(FRONT_DEFAULTS)
        c.update(cfg or {})
        self.cfg, self.device = c, torch.device(device)
        self.max_boxes, self.pad, self.divis = max_boxes, pad_value, size_divisibility

    # ---- ResizeShortestEdge.get_output_shape (d2 upstream)
    @staticmethod
    def output_shape(oldh, oldw, short_edge_length, max_size):
        size = short_edge_length * 1.0
        scale = size / min(oldh, oldw)
        newh, neww = (size, scale * oldw) if oldh < oldw else (scale * oldh, size)
        if max(newh, neww) > max_size:
            scale = max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return int(newh + 0.5), int(neww + 0.5)

    def draw(self, hw, rng_np=np.random):
        """one image's random numbers, in AugmentationList order (each get_transform sees the previous result)"""
        c = self.cfg
        h, w = hw
        sizes = c["MIN_SIZE_TRAIN"]
        if c["MIN_SIZE_TRAIN_SAMPLING"] == "range":
            size = int(rng_np.randint(sizes[0], sizes[1] + 1))
        else:
            size = int(rng_np.choice(sizes))
        nh, nw = (h, w) if size == 0 else self.output_shape(h, w, size, c["MAX_SIZE_TRAIN"])
        d = dict(nh=nh, nw=nw, hflip=False, vflip=False, sx=0, sy=0)
        if c["HFLIP"]:
            d["hflip"] = bool(rng_np.uniform(0, 1.0) < c["HFLIP_PROB"])
        if c["VFLIP"]:
            d["vflip"] = bool(rng_np.uniform(0, 1.0) < c["VFLIP_PROB"])
        if c["SATURATION"]:                                        # INPUT.COLOR_JITTER.SATURATION: RandomSaturation(0.8, 1.2)
            d["sat"] = float(rng_np.uniform(0.8, 1.2))
        if c["BRIGHTNESS"]:                                        # INPUT.COLOR_JITTER.BRIGHTNESS: RandomBrightness(0.8, 1.2)
            d["bri"] = float(rng_np.uniform(0.8, 1.2))
        if c["DISTORTION"]:
            # YOLOFDistortTransform.apply_image draws while the image is transformed (transform.py:268-270, _rand_scale
            # :293-308), i.e. after RandomBrightness's draw and before YOLOFRandomShift's: dhue, then (scale, coin) twice
            def rand_scale(upper):
                scale = rng_np.uniform(low=1, high=upper)
                return scale if rng_np.rand() > 0.5 else 1 / scale
            dhue = rng_np.uniform(low=-c["DISTORTION_HUE"], high=c["DISTORTION_HUE"])
            d["dis"] = (float(dhue), float(rand_scale(c["DISTORTION_SATURATION"])), float(rand_scale(c["DISTORTION_EXPOSURE"])))
        if c["SHIFT"] and c["SHIFT_PIXELS"] > 0:
            if rng_np.uniform(0, 1.0) < 0.5:                       # YOLOFRandomShift(prob=0.5 default, max_shifts)
                d["sx"] = int(rng_np.randint(low=-c["SHIFT_PIXELS"], high=c["SHIFT_PIXELS"]))
                d["sy"] = int(rng_np.randint(low=-c["SHIFT_PIXELS"], high=c["SHIFT_PIXELS"]))
        return d

    @staticmethod
    def _hull(b, fn):
        """fvcore Transform.apply_box: corners through the coordinate map, then the axis-aligned hull"""
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = fn(b.reshape(-1, 4)[:, idxs].reshape(-1, 2)).reshape((-1, 4, 2))
        return np.concatenate((coords.min(axis=1), coords.max(axis=1)), axis=1)

    @classmethod
    def boxes(cls, labels, hw, d):
        """labels float64 [n, 5] (x1, y1, x2, y2, cls) of the source image -> the same rows on the augmented image
        (clipped to it; empty boxes are NOT dropped here: the mosaic branch keeps them, the plain branch filters)"""
        lab = np.asarray(labels, np.float64).reshape(-1, 5).copy()
        if len(lab) == 0:
            return lab
        h, w = hw
        nh, nw = d["nh"], d["nw"]
        b = lab[:, :4]

        def scale(c):
            c[:, 0] = c[:, 0] * (nw * 1.0 / w)
            c[:, 1] = c[:, 1] * (nh * 1.0 / h)
            return c

        def hf(c):
            c[:, 0] = nw - c[:, 0]
            return c

        def vf(c):
            c[:, 1] = nh - c[:, 1]
            return c

        def sh(c):
            c[:, 0] += d["sx"]
            c[:, 1] += d["sy"]
            return c
        if (nh, nw) != (h, w):
            b = cls._hull(b, scale)
        if d["hflip"]:
            b = cls._hull(b, hf)
        if d["vflip"]:
            b = cls._hull(b, vf)
        if d["sx"] or d["sy"]:
            b = cls._hull(b, sh)
        lab[:, :4] = np.minimum(b.clip(min=0), np.array([nw, nh, nw, nh], np.float64))
        return lab

    def _launch(self, jobs, keep):
        n = len(jobs)
        bh, bv = C.c_int32(0), C.c_int32(0)
        lib = L.lib()
        L.check(lib.mi_pil_resize_jobs_layout(jobs, n, C.byref(bh), C.byref(bv)), "mi_pil_resize_jobs_layout")
        tab = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(self.device)
        st = L.stream_ptr()
        L.check(lib.mi_pil_resize_h(tab.data_ptr(), n, bh.value, st), "mi_pil_resize_h")
        L.check(lib.mi_pil_resize_v(tab.data_ptr(), n, bv.value, st), "mi_pil_resize_v")
        self._keep = (tab, keep)                                   # alive until the stream has run the two launches

    def _jobs(self, images, draws, dsts):
        """the job table: dsts = (address, channel / row / column stride in bytes) of each image's destination"""
        jobs = (L.mi_pil_resize_job * len(images))()
        tmps = []
        for j, img, d, (ptr, dsc, dsy, dsx) in zip(jobs, images, draws, dsts):
            if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous():
                raise ValueError("GpuFrontAugment: contiguous uint8 [H, W, 3] images")
            h0, w0 = img.shape[:2]
            j.src, j.src_ld, j.h0, j.w0, j.nh, j.nw = img.data_ptr(), 3 * w0, h0, w0, d["nh"], d["nw"]
            j.hflip, j.vflip, j.shift_x, j.shift_y = int(d["hflip"]), int(d["vflip"]), d["sx"], d["sy"]
            if d.get("sat") is not None:                           # BlendTransform(grey, 1 - w, w): numpy's dtypes (see pil_color)
                j.color |= 1
                j.sat_src, j.sat_dst = 1 - d["sat"], float(np.float32(d["sat"]))
            if d.get("bri") is not None:
                j.color |= 2
                j.bri_dst = float(np.float32(d["bri"]))
            if d.get("dis") is not None:                           # numpy: float32 image (op) weak Python scalar -> float32
                dhue, dsat, dexp = d["dis"]
                j.color |= 4
                j.dis_hue, j.dis_sat, j.dis_exp = (float(np.float32(dhue * 179 / 255.)), float(np.float32(dsat)),
                                                    float(np.float32(dexp)))
                j.dis_pos = int(dhue > 0)
            if d["nw"] != w0:
                t = torch.empty(h0, d["nw"], 3, dtype=torch.uint8, device=img.device)
                tmps.append(t)
                j.tmp = t.data_ptr()
            j.dst, j.dsc, j.dsy, j.dsx = ptr, dsc, dsy, dsx
        return jobs, tmps

    def _check_device(self, images):
        if self.device.type != "cuda" or not all(i.is_cuda for i in images):
            raise L.MI355Error("GpuFrontAugment: the MI355X path needs device tensors (no CPU pixel path)")

    def apply(self, images, draws):
        """images: device uint8 HWC tensors (decoded, as MosaicPool holds them); draws: one dict from draw() each.
        Returns the augmented HWC device images (what the mosaic branch then resizes and pastes)."""
        self._check_device(images)
        outs = [torch.empty(d["nh"], d["nw"], 3, dtype=torch.uint8, device=self.device) for d in draws]
        jobs, tmps = self._jobs(images, draws, [(o.data_ptr(), 1, 3 * d["nw"], 3) for o, d in zip(outs, draws)])
        self._launch(jobs, (tmps, list(images)))
        return outs

    def batch_shape(self, draws):
        Hp = (max(d["nh"] for d in draws) + self.divis - 1) // self.divis * self.divis
        Wp = (max(d["nw"] for d in draws) + self.divis - 1) // self.divis * self.divis
        return Hp, Wp

    def make_batch(self, images, labels, draws):
        """the mapper with the mosaic OFF (dataset_mapper.py:615-640) + YOLOX.preprocess_image (meta_arch/yolox.py:95-162) for
        a batch: images as in apply(), labels float64 [n_i, 5] (x1, y1, x2, y2, cls) per image.  Returns what
        GpuMosaicMapper.make_batch returns: (uint8 [B, 3, Hp, Wp] padded with 114, float32 [B, max_boxes, 5] rows
        (cls, cx, cy, w, h), per-sample (h, w))."""
        self._check_device(images)
        B = len(images)
        Hp, Wp = self.batch_shape(draws)
        out = torch.full((B, 3, Hp, Wp), self.pad, dtype=torch.uint8, device=self.device)
        rows = self.label_rows(images, labels, draws)
        jobs, tmps = self._jobs(images, draws, [(out.data_ptr() + b * 3 * Hp * Wp, Hp * Wp, Wp, 1) for b in range(B)])
        self._launch(jobs, (tmps, list(images)))
        return out, torch.from_numpy(rows).to(self.device, non_blocking=True), [(d["nh"], d["nw"]) for d in draws]

    def label_rows(self, images, labels, draws):
        """host half of make_batch: transform_instance_annotations -> annotations_to_instances (float32 Boxes) ->
        filter_empty_instances (width, height > 1e-5) -> preprocess_image's (cls, cx, cy, w, h) rows"""
        rows = np.zeros((len(images), self.max_boxes, 5), np.float32)
        for b, (img, lab, d) in enumerate(zip(images, labels, draws)):
            t = self.boxes(lab, tuple(img.shape[:2]), d)
            box = t[:, :4].astype(np.float32)
            keep = ((box[:, 2] - box[:, 0]) > 1e-5) & ((box[:, 3] - box[:, 1]) > 1e-5)
            box, cls_ = box[keep][: self.max_boxes], t[keep, 4][: self.max_boxes]
            n = len(box)
            rows[b, :n, 0] = cls_
            rows[b, :n, 1] = (box[:, 0] + box[:, 2]) / 2
            rows[b, :n, 2] = (box[:, 1] + box[:, 3]) / 2
            rows[b, :n, 3] = box[:, 2] - box[:, 0]
            rows[b, :n, 4] = box[:, 3] - box[:, 1]
        return rows


# ------------------------------------------------------------------------------------------------ the whole mapper
class _PoolView:
    """what GpuMosaicMapper.make_batch reads of a pool: `images` (device uint8 HWC) and `labels` (float64 [n, 5])"""

    def __init__(self):
        self.images, self.labels = [], []

    def add(self, image, labels):
        self.images.append(image)
        self.labels.append(labels)
        return len(self.images) - 1

    def __len__(self):
        return len(self.images)


class GpuDatasetMapper:
    """`MyDatasetMapper2.__call__` (yolov7/data/dataset_mapper.py:477-640) for a batch of decoded images, one call per
    training batch instead of one per image on a CPU worker:

      per sample, in the reference's order - mosaic_flag = np.random.randint(2) once the pool holds more than NUM_IMAGES
      entries, the three partners by np.random.choice over the pool, the sample appended to the pool; the T.* front on the
      current image; with the flag set: the mosaic size / centre draws, the front on each partner as it is loaded, the four
      pastes, random_perspective, and (ENABLE_MIXUP, only when labels survived) mixup with one more pool image, itself
      loaded through the front; without the flag: Instances of the fronted image, empty boxes dropped;
      then YOLOX.preprocess_image's pad-to-batch and label rows over the MIXED batch (both kinds of sample, as the
      reference's loader collates them).

    The host draws every random number from the two streams the reference uses (numpy's and Python's `random`) in the
    reference's order and does its float64 label arithmetic; the pixels are the launches of GpuFrontAugment (one pair for all
    loads of the batch) and GpuMosaicMapper (paste, warp, mixup), the mixed batch is assembled on the device.  `enable_aug`
    False = `MyDatasetMapper2.disable_aug()` (after DISABLE_AT_ITER): the front only.  The images come decoded (device uint8
    HWC: `GpuJpegDecoder`).  With the front's DISTORTION on (configs/coco/yolox_s.yaml:46-47) every loaded image is float32 in
    the reference from there on: the mosaic / mixup resizes then take cv2's float path (see GpuFrontAugment)."""

    def __init__(self, mosaic_cfg=None, front_cfg=None, device="cuda", enable_mosaic=True, enable_mixup=False, pool_capacity=1000,
                 max_boxes=100, pad_value=114, size_divisibility=32):
        self.device = torch.device(device)
        self.front = GpuFrontAugment(front_cfg, device, max_boxes, pad_value, size_divisibility)
        self.mosaic = GpuMosaicMapper(mosaic_cfg, device, max_boxes, pad_value, size_divisibility)
        self.pool = MosaicPool(device, pool_capacity)
        self.enable_mosaic, self.enable_mixup, self.enable_aug = enable_mosaic, enable_mixup, True
        self.max_boxes, self.pad, self.divis = max_boxes, pad_value, size_divisibility

    def disable_aug(self):
        self.enable_aug = False

    @staticmethod
    def _mosaic_labels(shapes, labels, p):
        """dataset_mapper.py:537-590 for the labels: the four images' rows scaled by their resize factor, shifted to their
        quadrant and clipped to the 2x canvas (shapes = the (h, w) of the four LOADED images)"""
        dim, yc, xc = p["input_dim"], p["yc"], p["xc"]
        labels4 = []
        for i, ((h0, w0), lab) in enumerate(zip(shapes, labels)):
            scale = min(1. * dim[0] / h0, 1. * dim[1] / w0)
            w, h = int(w0 * scale), int(h0 * scale)
            (x1a, y1a, _x2a, _y2a), (x1b, y1b) = GpuMosaicMapper._placement(i, w, h, xc, yc, dim)
            if lab.size > 0:
                t = lab.copy()
                padw, padh = x1a - x1b, y1a - y1b
                t[:, 0] = scale * lab[:, 0] + padw
                t[:, 1] = scale * lab[:, 1] + padh
                t[:, 2] = scale * lab[:, 2] + padw
                t[:, 3] = scale * lab[:, 3] + padh
                labels4.append(t)
        if not labels4:
            return np.zeros((0, 5))
        labels4 = np.concatenate(labels4, 0)
        np.clip(labels4[:, 0], 0, 2 * dim[1], out=labels4[:, 0])
        np.clip(labels4[:, 1], 0, 2 * dim[0], out=labels4[:, 1])
        np.clip(labels4[:, 2], 0, 2 * dim[1], out=labels4[:, 2])
        np.clip(labels4[:, 3], 0, 2 * dim[0], out=labels4[:, 3])
        return labels4

    def plan(self, image, labels, rng_np=np.random, rng_py=random):
        """the host half of one `__call__`: appends (image, labels) to the pool and returns the sample's plan -
        loads = [(pool image, its labels, its front draw)] (the current image first), and for a mosaic sample the mosaic
        draws and the mixup draw (or None)"""
        pool = self.pool
        flag, partners = 0, None
        if self.enable_mosaic and self.enable_aug and len(pool) > self.mosaic.cfg["NUM_IMAGES"]:
            flag = int(rng_np.randint(2))
            if flag == 1:
                partners = [int(i) for i in rng_np.choice(len(pool), self.mosaic.cfg["NUM_IMAGES"] - 1)]
                partners = [(pool.images[i], pool.labels[i]) for i in partners]      # (the entries, not their positions)
        if self.enable_mosaic and self.enable_aug:
            pool.append(image, labels)
            cur = (pool.images[-1], pool.labels[-1])
        else:
            img = torch.as_tensor(image)
            cur = (img.to(self.device).contiguous(), np.asarray(labels, np.float64).reshape(-1, 5))

        def load(entry):
            img, lab = entry
            return (img, lab, self.front.draw(tuple(img.shape[:2]), rng_np))
        plan = dict(mosaic=False, loads=[load(cur)], params=None, mixup=None)
        if not (flag == 1 and partners is not None):
            return plan
        plan["mosaic"] = True
        p = plan["params"] = self.mosaic.draw(rng_np, rng_py)           # (w, h from numpy's stream BEFORE the partners load)
        plan["loads"] += [load(e) for e in partners]
        if self.enable_mixup:
            shapes = [(d["nh"], d["nw"]) for (_, _, d) in plan["loads"]]
            labs = [self.front.boxes(lab, tuple(img.shape[:2]), d) for (img, lab, d) in plan["loads"]]
            dim = p["input_dim"]
            M, width, height = self.mosaic._matrix((dim[0] * 2, dim[1] * 2), p["draws"], [-dim[0] // 2, -dim[1] // 2])
            t = self.mosaic._warp_labels(self._mosaic_labels(shapes, labs, p), M, p["draws"][1], width, height)
            if len(t):                                                  # dataset_mapper.py:602
                jit = rng_py.uniform(*self.mosaic.cfg["MSCALE"])
                flip = rng_py.uniform(0, 1) > 0.5
                idx = int(rng_np.choice(len(pool), 1)[0])
                plan["loads"].append(load((pool.images[idx], pool.labels[idx])))
                oh, ow = int(dim[0] * jit), int(dim[1] * jit)
                ph, pw = max(oh, height), max(ow, width)
                y_off = rng_py.randint(0, ph - height - 1) if ph > height else 0
                x_off = rng_py.randint(0, pw - width - 1) if pw > width else 0
                plan["mixup"] = dict(jit=jit, flip=bool(flip), x_off=x_off, y_off=y_off)
        return plan

    def make_batch(self, samples, rng_np=np.random, rng_py=random):
        """samples: [(uint8 HWC image (numpy / tensor, host or device), labels float64 [n, 5] (x1, y1, x2, y2, cls))].
        Returns (uint8 [B, 3, Hp, Wp], float32 [B, max_boxes, 5] rows (cls, cx, cy, w, h), per-sample (h, w)) on the device -
        what NativeTrainer.load_batch / feed take."""
        if self.device.type != "cuda":
            raise L.MI355Error("GpuDatasetMapper: the MI355X path needs a device (no CPU pixel path)")
        plans = [self.plan(img, lab, rng_np, rng_py) for img, lab in samples]
        return self.run(plans)

    def run(self, plans):
        B = len(plans)
        plain = [b for b, p in enumerate(plans) if not p["mosaic"]]
        mos = [b for b, p in enumerate(plans) if p["mosaic"]]
        parts = {}
        if plain:
            imgs = [plans[b]["loads"][0][0] for b in plain]
            out, rows, sizes = self.front.make_batch(imgs, [plans[b]["loads"][0][1] for b in plain], [plans[b]["loads"][0][2] for b in plain])
            for k, b in enumerate(plain):
                parts[b] = (out[k], rows[k], sizes[k])
        if mos:
            loads = [ld for b in mos for ld in plans[b]["loads"]]
            fronted = self.front.apply([ld[0] for ld in loads], [ld[2] for ld in loads])
            view, groups, params, mixups, k = _PoolView(), [], [], [], 0
            for b in mos:
                p = plans[b]
                ids = []
                for (img, lab, d) in p["loads"]:
                    ids.append(view.add(fronted[k], self.front.boxes(lab, tuple(img.shape[:2]), d)))
                    k += 1
                groups.append(tuple(ids[:4]))
                params.append(p["params"])
                mixups.append(dict(p["mixup"], idx=ids[4]) if p["mixup"] is not None else None)
            out, rows, sizes = self.mosaic.make_batch(view, groups, params, mixups if any(m is not None for m in mixups) else None,
                                                      float_src=bool(self.front.cfg["DISTORTION"]))
            for k, b in enumerate(mos):
                parts[b] = (out[k], rows[k], sizes[k])
        if not plain or not mos:                                   # one kind of sample only: that launch's batch is the batch
            return out, rows, [parts[b][2] for b in range(B)]
        Hp = max(parts[b][0].shape[1] for b in range(B))
        Wp = max(parts[b][0].shape[2] for b in range(B))
        batch = torch.full((B, 3, Hp, Wp), self.pad, dtype=torch.uint8, device=self.device)
        rows_all = torch.zeros(B, self.max_boxes, 5, dtype=torch.float32, device=self.device)
        for b in range(B):
            img, r, _ = parts[b]
            batch[b, :, : img.shape[1], : img.shape[2]] = img      # (each part is padded with 114 beyond its own size already)
            rows_all[b] = r
        return batch, rows_all, [parts[b][2] for b in range(B)]


# ------------------------------------------------------------------------------------------------ image decoding
class GpuJpegDecoder:
    """detectron2 `utils.read_image(file_name, format)` - the first thing `MyDatasetMapper2._load_image_with_annos` does
    (yolov7/data/dataset_mapper.py:646-648; d2 un-vendored: PIL.Image.open -> EXIF orientation -> convert("RGB") -> channel
    order) - for a BATCH of JPEG files (sequential and progressive): the Huffman decoding of every scan runs on host threads inside the library
    (`mi_jpeg_parse`, `mi_jpeg_huffman`; ctypes drops the GIL), the coefficient blocks go to the device in one pinned copy,
    and two launches do the rest for all images (de-quantisation + libjpeg's ISLOW IDCT per block; fancy chroma up-sampling,
    YCbCr -> RGB, EXIF transpose and channel order per pixel).  Bit-identical to Pillow's decode.  Arithmetic-
    coded, 12-bit and CMYK files raise MI355Error (there is no CPU decoder to fall back to)."""

    # Pillow refuses an image above 2 * Image.MAX_IMAGE_PIXELS (DecompressionBombError; MAX_IMAGE_PIXELS = 89 478 485) - the
    # PIL path this replaces never allocates for a 65535 x 65535 header; neither does this one (a SOF of that size would ask
    # for ~13 GB of pinned coefficients before a single scan byte is checked)
    MAX_IMAGE_PIXELS = 2 * 89478485

    def __init__(self, device="cuda", format="BGR", apply_orientation=True, workers=8, max_image_pixels=None):      # noqa: A002 (d2's argument name)
        if format not in ("BGR", "RGB"):
            raise ValueError("GpuJpegDecoder: format BGR (the reference's INPUT.FORMAT default) or RGB")
        self.device, self.bgr, self.orient, self.workers = torch.device(device), format == "BGR", bool(apply_orientation), workers
        self.max_image_pixels = self.MAX_IMAGE_PIXELS if max_image_pixels is None else int(max_image_pixels)

    def _host_half(self, files, alloc):
        """parse every file, decode the entropy-coded data on host threads into ONE int16 buffer.  alloc(count) ->
        (owner, address) of that buffer.  Returns (infos, offsets of each file's coefficients in the buffer, owner)"""
        lib = L.lib()
        n = len(files)
        bufs = [(C.c_uint8 * len(f)).from_buffer_copy(f) for f in files]
        infos = [L.mi_jpeg_info() for _ in range(n)]
        for k in range(n):
            L.check(lib.mi_jpeg_parse(bufs[k], len(files[k]), C.byref(infos[k])), f"mi_jpeg_parse (file {k})")
            if int(infos[k].width) * int(infos[k].height) > self.max_image_pixels:      # before ANY allocation sized by the header
                raise L.MI355Error(f"GpuJpegDecoder: file {k} declares {infos[k].width} x {infos[k].height} pixels, above the "
                                   f"limit of {self.max_image_pixels} (decompression-bomb guard, as PIL.Image.MAX_IMAGE_PIXELS)")
        offs = np.concatenate([[0], np.cumsum([i.coef_count for i in infos])]).astype(np.int64)
        host, base = alloc(int(offs[-1]))

        # one call: the library spreads the files over its own host threads (no Python thread / GIL hand-off per file)
        datas = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        lens = (C.c_int64 * n)(*[len(f) for f in files])
        iarr = (L.mi_jpeg_info * n)(*infos)
        coefs = (C.c_void_p * n)(*[base + 2 * int(offs[k]) for k in range(n)])
        rcs = (C.c_int32 * n)()
        rc = lib.mi_jpeg_huffman_batch(datas, lens, iarr, coefs, n, max(1, int(self.workers)), rcs)
        if rc != 0:
            k = next(i for i in range(n) if rcs[i] != 0)
            L.check(rcs[k], f"mi_jpeg_huffman (file {k})")
        return infos, offs, host

    def out_shape(self, info):
        tr = self.orient and info.orientation >= 5
        return (info.width, info.height) if tr else (info.height, info.width)

    def _jobs(self, infos, offs, coef_ptr, planes_ptr, out_ptrs):
        """the job table of a batch (addresses of the coefficient buffer, the plane scratch and each output) + block counts"""
        lib = L.lib()
        n = len(infos)
        jobs = (L.mi_jpeg_job * n)()
        for k, info in enumerate(infos):
            L.check(lib.mi_jpeg_job_fill(C.byref(info), coef_ptr + 2 * int(offs[k]), planes_ptr + int(offs[k]), out_ptrs[k],
                                         int(self.bgr), int(self.orient), C.byref(jobs[k])), "mi_jpeg_job_fill")
        bi, bp = C.c_int32(0), C.c_int32(0)
        L.check(lib.mi_jpeg_jobs_layout(jobs, n, C.byref(bi), C.byref(bp)), "mi_jpeg_jobs_layout")
        return jobs, bi.value, bp.value

    def _check_device(self):
        if self.device.type != "cuda":
            raise L.MI355Error("GpuJpegDecoder: the MI355X path needs a device (no CPU decode)")

    def _alloc_host(self, count):
        t = torch.empty(count, dtype=torch.int16, pin_memory=True)
        return t, t.data_ptr()

    def _launch(self, jobs, n, blocks_idct, blocks_pix):
        lib = L.lib()
        tab = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(self.device)
        st = L.stream_ptr()
        L.check(lib.mi_jpeg_idct(tab.data_ptr(), n, blocks_idct, st), "mi_jpeg_idct")
        L.check(lib.mi_jpeg_color(tab.data_ptr(), n, blocks_pix, st), "mi_jpeg_color")
        return tab

    def decode(self, files):
        """files: bytes-like JPEG files.  Returns device uint8 [H, W, 3] tensors (H, W after the EXIF transpose)."""
        self._check_device()
        infos, offs, host = self._host_half(files, self._alloc_host)
        total = int(offs[-1])
        coef = host.to(self.device, non_blocking=True)
        planes = torch.empty(total, dtype=torch.uint8, device=self.device)
        outs = [torch.empty(*self.out_shape(i), 3, dtype=torch.uint8, device=self.device) for i in infos]
        jobs, bi, bp = self._jobs(infos, offs, coef.data_ptr(), planes.data_ptr(), [o.data_ptr() for o in outs])
        tab = self._launch(jobs, len(files), bi, bp)
        self._keep = (host, coef, planes, tab)                      # alive until the stream has run the copy and the launches
        return outs


# ------------------------------------------------------------------------------------------------ DETR's mapper
class GpuDetrMapper:
    """`DetrDatasetMapper.__call__` (yolov7/data/dataset_mapper.py:804-900, training) for a batch of decoded images - the input
    pipeline of BASELINE configs[3]: T.RandomFlip, then with INPUT.CROP on and np.random.rand() <= 0.5 a
    T.ResizeShortestEdge([400, 500, 600]) + T.RandomCrop("absolute_range", SIZE), then T.ResizeShortestEdge(MIN_SIZE_TRAIN,
    MAX_SIZE_TRAIN) (`build_transform_gen` :777-800); boxes through `transform_instance_annotations`, Instances, empty boxes
    dropped.  The host draws the reference's random numbers in its order and does the float64 box arithmetic; the pixels are
    the Pillow-exact resampling launches of the front (`mi_pil_resize_h / _v`): the flip is applied to the SOURCE of the first
    resize (`src_hflip`: d2 flips before it resizes), the crop is an offset pointer + row stride into the first resize's
    output, and the last resize writes each sample as the [3, h, w] uint8 tensor the DETR meta-architecture takes."""

    def __init__(self, device="cuda", min_sizes=(480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832), max_size=1333,
                 sample_style="choice", crop=(384, 600), crop_sizes=(400, 500, 600)):
        self.device = torch.device(device)
        self.min_sizes, self.max_size, self.sample_style = tuple(min_sizes), max_size, sample_style
        self.crop, self.crop_sizes = (tuple(crop) if crop else None), tuple(crop_sizes)
        self._front = GpuFrontAugment(device=device)

    def plan(self, hw, rng_np=np.random):
        """one image's draws, in the reference's order: rand (branch), uniform (flip), [choice, 4 x randint], choice"""
        import sys
        h, w = hw
        p = dict(src=(h, w), crop=None)
        take_crop = self.crop is not None and not (rng_np.rand() > 0.5)
        p["flip"] = bool(rng_np.uniform(0, 1.0) < 0.5)
        if take_crop:
            h, w = GpuFrontAugment.output_shape(h, w, int(rng_np.choice(self.crop_sizes)), sys.maxsize)
            ch = int(rng_np.randint(min(h, self.crop[0]), min(h, self.crop[1]) + 1))
            cw = int(rng_np.randint(min(w, self.crop[0]), min(w, self.crop[1]) + 1))
            y0 = int(rng_np.randint(h - ch + 1))
            x0 = int(rng_np.randint(w - cw + 1))
            p["crop"] = (h, w, x0, y0, cw, ch)
            h, w = ch, cw
        if self.sample_style == "range":
            size = int(rng_np.randint(self.min_sizes[0], self.min_sizes[1] + 1))
        else:
            size = int(rng_np.choice(self.min_sizes))
        p["size"] = GpuFrontAugment.output_shape(h, w, size, self.max_size)
        return p

    @staticmethod
    def boxes(labels, p):
        """float64 [n, 5] (x1, y1, x2, y2, cls) -> (float32 boxes [m, 4], classes [m]) on the output image"""
        lab = np.asarray(labels, np.float64).reshape(-1, 5)
        b = lab[:, :4].copy()
        h, w = p["src"]
        hull = GpuFrontAugment._hull
        if len(b):
            if p["flip"]:
                def hf(c, w=w):
                    c[:, 0] = w - c[:, 0]
                    return c
                b = hull(b, hf)

            def scale(h, w, nh, nw):
                def f(c):
                    c[:, 0] = c[:, 0] * (nw * 1.0 / w)
                    c[:, 1] = c[:, 1] * (nh * 1.0 / h)
                    return c
                return f
            if p["crop"] is not None:
                h1, w1, x0, y0, cw, ch = p["crop"]
                b = hull(b, scale(h, w, h1, w1))

                def cr(c):
                    c[:, 0] -= x0
                    c[:, 1] -= y0
                    return c
                b = hull(b, cr)
                h, w = ch, cw
            H, W = p["size"]
            b = hull(b, scale(h, w, H, W))
            b = np.minimum(b.clip(min=0), np.array([W, H, W, H], np.float64))
        box = b.astype(np.float32)
        keep = ((box[:, 2] - box[:, 0]) > 1e-5) & ((box[:, 3] - box[:, 1]) > 1e-5)
        return box[keep], lab[keep, 4]

    def _stage_jobs(self, images, plans, mids, outs):
        """(first-resize jobs of the crop samples, final-resize jobs of all samples); mids[k]: the crop samples' intermediate
        HWC tensors, outs[k]: every sample's [3, H, W] output"""
        J1, J2 = [], []
        for k, (img, p) in enumerate(zip(images, plans)):
            h0, w0 = p["src"]
            H, W = p["size"]
            j2 = dict(dst=outs[k].data_ptr(), dsc=H * W, dsy=W, dsx=1, nh=H, nw=W)
            if p["crop"] is not None:
                h1, w1, x0, y0, cw, ch = p["crop"]
                J1.append(dict(src=img.data_ptr(), src_ld=3 * w0, h0=h0, w0=w0, nh=h1, nw=w1, flip=int(p["flip"]), dst=mids[k].data_ptr(),
                               dsc=1, dsy=3 * w1, dsx=3))
                j2.update(src=mids[k].data_ptr() + (y0 * w1 + x0) * 3, src_ld=3 * w1, h0=ch, w0=cw, flip=0)
            else:
                j2.update(src=img.data_ptr(), src_ld=3 * w0, h0=h0, w0=w0, flip=int(p["flip"]))
            J2.append(j2)
        return J1, J2

    @staticmethod
    def _table(specs, alloc):
        """job dicts -> the C table; alloc(h, w) -> (owner, address) of a [h][w][3] scratch of the horizontal pass"""
        jobs = (L.mi_pil_resize_job * len(specs))()
        owners = []
        for j, s in zip(jobs, specs):
            j.src, j.src_ld, j.h0, j.w0, j.nh, j.nw, j.src_hflip = s["src"], s["src_ld"], s["h0"], s["w0"], s["nh"], s["nw"], s["flip"]
            j.dst, j.dsc, j.dsy, j.dsx = s["dst"], s["dsc"], s["dsy"], s["dsx"]
            if s["nw"] != s["w0"]:
                o, a = alloc(s["h0"], s["nw"])
                owners.append(o)
                j.tmp = a
        return jobs, owners

    def _check_device(self, images):
        if self.device.type != "cuda" or not all(i.is_cuda for i in images):
            raise L.MI355Error("GpuDetrMapper: the MI355X path needs device tensors (no CPU pixel path)")

    def make_batch(self, images, labels, rng_np=np.random):
        """images: device uint8 HWC tensors (e.g. GpuJpegDecoder(format="RGB") output), labels: float64 [n_i, 5] each.
        Returns [(uint8 [3, h, w] device tensor, float32 boxes [m, 4], classes [m])] - the "image" / gt_boxes / gt_classes of
        the reference's dataset dicts."""
        self._check_device(images)
        plans = [self.plan(tuple(i.shape[:2]), rng_np) for i in images]
        mids = [torch.empty(p["crop"][0], p["crop"][1], 3, dtype=torch.uint8, device=self.device) if p["crop"] is not None else None
                for p in plans]
        outs = [torch.empty(3, p["size"][0], p["size"][1], dtype=torch.uint8, device=self.device) for p in plans]
        J1, J2 = self._stage_jobs(images, plans, mids, outs)

        def alloc(h, w):
            t = torch.empty(h, w, 3, dtype=torch.uint8, device=self.device)
            return t, t.data_ptr()
        keep = [images, mids]
        for specs in (J1, J2):
            if specs:
                jobs, owners = self._table(specs, alloc)
                self._front._launch(jobs, owners)
                keep.append((owners, self._front._keep))
        self._keep = keep
        res = []
        for k, p in enumerate(plans):
            box, cls_ = self.boxes(labels[k], p)
            res.append((outs[k], box, cls_))
        return res
