"""CPU restatement (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it) of
the YOLOX training input pipeline of the reference, SURVEY 8(f) rank 2:

    MyDatasetMapper2.__call__, mosaic branch           yolov7/data/dataset_mapper.py:477-612
    random_perspective, box_candidates                 yolov7/data/transforms/data_augment.py:15-101
    YOLOX.preprocess_image (pad to /32 with 114,
    XYXY -> (cls, cx, cy, w, h) rows)                  yolov7/modeling/meta_arch/yolox.py:95-162

Label / matrix arithmetic is numpy float64 in the reference's operation order and is PINNED: tests run the reference's own
`random_perspective` by path (oracle/ref_loader.py::load_data_augment, cv2 stubbed by the two functions below) on the same
draws.

Pixel arithmetic is OpenCV's (cv2.resize INTER_LINEAR, cv2.warpAffine INTER_LINEAR / BORDER_CONSTANT 114), a third-party
dependency that is neither under /root/reference nor installed here (the reference pins no version; readme.md asks for
"opencv-python").  `resize_linear_u8` and `warp_affine_u8` restate OpenCV 4.x's published fixed-point algorithms
(modules/imgproc/src/resize.cpp: 11-bit coefficients, two-stage rounding of the 8-bit vertical pass;
modules/imgproc/src/imgwarp.cpp: source coordinates in 1/1024 px rounded to 1/32 px, 15-bit bilinear weight table):
PARITY UNPINNED for these two functions - no cv2 to check them against.  The HIP kernels are held bit-exact to THIS file.
"""
import math

import numpy as np

# ------------------------------------------------------------------------------------------------ OpenCV pixel arithmetic
INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15
INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS


def _sat_short(v):
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def resize_coeffs(src, dst):
    """per destination index: (source index of the left / upper tap, the two 11-bit coefficients).  resize.cpp: fx =
    (dx + 0.5) * scale - 0.5, sx = floor(fx); sx < 0 -> (0, fx = 0); sx >= src - 1 -> (src - 1, fx = 0) (both taps then
    read the edge pixel); coefficients saturate_cast<short>((1 - fx) * 2048), saturate_cast<short>(fx * 2048), fx in fp32"""
    scale = np.float64(src) / np.float64(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0
    s[hi] = src - 1
    a0 = _sat_short((np.float32(1.0) - f) * np.float32(INTER_RESIZE_COEF_SCALE))
    a1 = _sat_short(f * np.float32(INTER_RESIZE_COEF_SCALE))
    return s.astype(np.int32), a0, a1


def resize_linear_u8(img, dsize):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HWC.  Horizontal pass in int32 (pixel x 11-bit
    coefficient), vertical pass as OpenCV's 8-bit vector path: ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16), + 2, >> 2"""
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    if (w, h) == (W, H):
        return img.copy()
    sx, ax0, ax1 = resize_coeffs(W, w)
    sy, by0, by1 = resize_coeffs(H, h)
    sx1 = np.minimum(sx + 1, W - 1)
    sy1 = np.minimum(sy + 1, H - 1)
    src = img.astype(np.int32)
    rows = src[:, sx] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None]      # [H, w, C]
    r0, r1 = rows[sy], rows[sy1]                                                     # [h, w, C]
    v = ((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16)
    return np.clip((v + 2) >> 2, 0, 255).astype(np.uint8)


def bilinear_tab():
    """imgwarp.cpp initInterTab2D(INTER_LINEAR, fixpt): 32 x 32 x (2 x 2) weights saturate_cast<short>(w * 32768), each cell
    then made to sum to 32768 (a deficit goes to the largest weight, an excess is taken from the smallest).  The cell of an
    integer source position therefore carries the full 32768 (kept in a wider integer here: with 32767 or 32768 an 8-bit
    pixel comes out unchanged either way, (p * 32767 + 16384) >> 15 == p)"""
    tab = np.zeros((INTER_TAB_SIZE, INTER_TAB_SIZE, 4), np.int32)
    t1 = np.zeros((INTER_TAB_SIZE, 2), np.float32)
    for i in range(INTER_TAB_SIZE):
        x = np.float32(i) * np.float32(1.0 / INTER_TAB_SIZE)
        t1[i] = (np.float32(1.0) - x, x)
    for i in range(INTER_TAB_SIZE):
        for j in range(INTER_TAB_SIZE):
            w = np.array([t1[i, 0] * t1[j, 0], t1[i, 0] * t1[j, 1], t1[i, 1] * t1[j, 0], t1[i, 1] * t1[j, 1]], np.float32)
            it = _sat_short(w * np.float32(INTER_REMAP_COEF_SCALE))
            diff = int(it.sum()) - INTER_REMAP_COEF_SCALE
            if diff < 0:
                it[int(np.argmax(it))] -= diff
            elif diff > 0:
                it[int(np.argmin(it))] -= diff
            tab[i, j] = it
    return tab          # [fy][fx][(y0x0, y0x1, y1x0, y1x1)]


_TAB = None


def invert_affine(M):
    """cv2.invertAffineTransform's arithmetic as warpAffine applies it (imgwarp.cpp: D = M00 M11 - M01 M10, ...), float64"""
    M = np.asarray(M, np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    A12, A21 = -M[0, 1] * D, -M[1, 0] * D
    b1 = -A11 * M[0, 2] - A12 * M[1, 2]
    b2 = -A21 * M[0, 2] - A22 * M[1, 2]
    return np.array([[A11, A12, b1], [A21, A22, b2]], np.float64)


def warp_fixed_coords(Minv, w, h):
    """the per-pixel source coordinates of warpAffine in 1/32 px: adelta[x] = rint(Minv00 * x * 1024), bdelta[y] =
    rint((Minv01 * y + Minv02) * 1024) + 16; X = (adelta + bdelta) >> 5 (likewise Y)"""
    x = np.arange(w, dtype=np.float64)
    y = np.arange(h, dtype=np.float64)
    rd = AB_SCALE // INTER_TAB_SIZE // 2
    sat = lambda v: np.clip(np.rint(v), -2147483648, 2147483647).astype(np.int64)
    ax, bx = sat(Minv[0, 0] * x * AB_SCALE), sat(Minv[1, 0] * x * AB_SCALE)
    ay, by = sat((Minv[0, 1] * y + Minv[0, 2]) * AB_SCALE) + rd, sat((Minv[1, 1] * y + Minv[1, 2]) * AB_SCALE) + rd
    X = (ax[None, :] + ay[:, None]) >> (AB_BITS - INTER_BITS)
    Y = (bx[None, :] + by[:, None]) >> (AB_BITS - INTER_BITS)
    return X, Y


def warp_affine_u8(img, M, dsize, border=114):
    """cv2.warpAffine(img, M, (w, h), borderValue=(border,) * 3) (INTER_LINEAR, BORDER_CONSTANT, M maps source -> destination)"""
    global _TAB
    if _TAB is None:
        _TAB = bilinear_tab()
    w, h = int(dsize[0]), int(dsize[1])
    H, W, Cc = img.shape
    X, Y = warp_fixed_coords(invert_affine(M), w, h)
    sx, sy = (X >> INTER_BITS), (Y >> INTER_BITS)
    # remap clamps the integer coordinates to short before sampling
    sx, sy = np.clip(sx, -32768, 32767), np.clip(sy, -32768, 32767)
    wt = _TAB[(Y & (INTER_TAB_SIZE - 1)), (X & (INTER_TAB_SIZE - 1))]               # [h, w, 4]
    src = img.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(ok[..., None], v, border)
    acc = (tap(sy, sx) * wt[..., 0:1] + tap(sy, sx + 1) * wt[..., 1:2] + tap(sy + 1, sx) * wt[..., 2:3]
           + tap(sy + 1, sx + 1) * wt[..., 3:4])
    return np.clip((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS, 0, 255).astype(np.uint8)


def rotation_matrix_2d(angle, scale):
    """cv2.getRotationMatrix2D(center=(0, 0), angle (degrees), scale): [[a, b, 0], [-b, a, 0]], a = s cos, b = s sin"""
    r = angle * math.pi / 180.0
    a, b = scale * math.cos(r), scale * math.sin(r)
    return np.array([[a, b, 0.0], [-b, a, 0.0]], np.float64)


# ------------------------------------------------------------------------------------------------ label / geometry logic
def box_candidates(box1, box2, wh_thr=2, ar_thr=20, area_thr=0.2):
    """data_augment.py:15-28"""
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + 1e-16), h2 / (w2 + 1e-16))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + 1e-16) > area_thr) & (ar < ar_thr)


def perspective_matrix(img_hw, draws, border):
    """data_augment.py:34-66 with the five random.uniform draws given in the reference's call order:
    draws = (angle, scale, shear_x_deg, shear_y_deg, tx_frac, ty_frac)"""
    a, s, shx, shy, tx, ty = draws
    height, width = img_hw[0] + border[0] * 2, img_hw[1] + border[1] * 2
    Cm = np.eye(3)
    Cm[0, 2] = -img_hw[1] / 2
    Cm[1, 2] = -img_hw[0] / 2
    R = np.eye(3)
    R[:2] = rotation_matrix_2d(a, s)
    S = np.eye(3)
    S[0, 1] = math.tan(shx * math.pi / 180)
    S[1, 0] = math.tan(shy * math.pi / 180)
    T = np.eye(3)
    T[0, 2] = tx * width
    T[1, 2] = ty * height
    return T @ S @ R @ Cm, width, height


def perspective_labels(targets, M, s, width, height):
    """data_augment.py:77-101 (affine branch): corners through M, new boxes, clip, box_candidates filter"""
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


def random_perspective(img, targets, draws, border):
    """data_augment.py:31-101 (perspective 0.0): image through warp_affine_u8, labels through perspective_labels"""
    M, width, height = perspective_matrix(img.shape[:2], draws, border)
    if border[0] != 0 or border[1] != 0 or (M != np.eye(3)).any():
        img = warp_affine_u8(img, M[:2], (width, height), 114)
    return img, perspective_labels(np.array(targets, np.float64).reshape(-1, 5).copy(), M, draws[1], width, height)


def mosaic_placement(i, w, h, xc, yc, input_dim):
    """dataset_mapper.py:540-563: (x1a, y1a, x2a, y2a) in the 2h x 2w canvas and (x1b, y1b, x2b, y2b) in the resized image"""
    H2, W2 = input_dim[0] * 2, input_dim[1] * 2
    if i == 0:
        x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
        x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
    elif i == 1:
        x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, W2), yc
        x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
    elif i == 2:
        x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(H2, yc + h)
        x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, w, min(y2a - y1a, h)
    else:
        x1a, y1a, x2a, y2a = xc, yc, min(xc + w, W2), min(H2, yc + h)
        x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
    return (x1a, y1a, x2a, y2a), (x1b, y1b, x2b, y2b)


def mosaic4(imgs, labels, input_dim, yc, xc):
    """dataset_mapper.py:523-590: four images resized by min(h / h0, w / w0), pasted around (xc, yc) on a 114 canvas;
    labels rows (x1, y1, x2, y2, cls) scaled, shifted and clipped to the canvas"""
    img4 = np.full((input_dim[0] * 2, input_dim[1] * 2, 3), 114, dtype=np.uint8)
    labels4 = []
    for i in range(4):
        img, _labels = imgs[i], np.asarray(labels[i], np.float64).reshape(-1, 5)
        h0, w0 = img.shape[:2]
        scale = min(1. * input_dim[0] / h0, 1. * input_dim[1] / w0)
        img = resize_linear_u8(img, (int(w0 * scale), int(h0 * scale)))
        h, w = img.shape[:2]
        (x1a, y1a, x2a, y2a), (x1b, y1b, x2b, y2b) = mosaic_placement(i, w, h, xc, yc, input_dim)
        img4[y1a:y2a, x1a:x2a] = img[y1b:y2b, x1b:x2b]
        padw, padh = x1a - x1b, y1a - y1b
        lab = _labels.copy()
        if _labels.size > 0:
            lab[:, 0] = scale * _labels[:, 0] + padw
            lab[:, 1] = scale * _labels[:, 1] + padh
            lab[:, 2] = scale * _labels[:, 2] + padw
            lab[:, 3] = scale * _labels[:, 3] + padh
            labels4.append(lab)
    if len(labels4):
        labels4 = np.concatenate(labels4, 0)
        np.clip(labels4[:, 0], 0, 2 * input_dim[1], out=labels4[:, 0])
        np.clip(labels4[:, 1], 0, 2 * input_dim[0], out=labels4[:, 1])
        np.clip(labels4[:, 2], 0, 2 * input_dim[1], out=labels4[:, 2])
        np.clip(labels4[:, 3], 0, 2 * input_dim[0], out=labels4[:, 3])
    else:
        labels4 = np.zeros((0, 5))
    return img4, labels4


def mosaic_sample(imgs, labels, input_dim, yc, xc, draws):
    """one training sample of the mosaic branch (mixup off, as configs/coco/yolox_s.yaml:57-62): HWC uint8 image of size
    input_dim and its (x1, y1, x2, y2, cls) rows"""
    img4, labels4 = mosaic4(imgs, labels, input_dim, yc, xc)
    return random_perspective(img4, labels4, draws, border=[-input_dim[0] // 2, -input_dim[1] // 2])


def preprocess_batch(samples, max_boxes=100, pad=114, divis=32):
    """YOLOX.preprocess_image (meta_arch/yolox.py:95-162): images padded at the bottom / right to the batch maximum rounded
    up to a multiple of 32 with 114, as uint8 [B, 3, H, W]; labels [B, max_boxes, 5] float32 rows (cls, cx, cy, w, h)"""
    Hm = max(s[0].shape[0] for s in samples)
    Wm = max(s[0].shape[1] for s in samples)
    Hm, Wm = (Hm + divis - 1) // divis * divis, (Wm + divis - 1) // divis * divis
    B = len(samples)
    out = np.full((B, 3, Hm, Wm), pad, np.uint8)
    lab = np.zeros((B, max_boxes, 5), np.float32)
    for b, (img, t) in enumerate(samples):
        out[b, :, : img.shape[0], : img.shape[1]] = img.transpose(2, 0, 1)
        t = np.asarray(t, np.float64).reshape(-1, 5)[:max_boxes]
        if len(t):
            box = t[:, :4].astype(np.float32)
            lab[b, : len(t), 0] = t[:, 4]
            lab[b, : len(t), 1] = (box[:, 0] + box[:, 2]) / 2
            lab[b, : len(t), 2] = (box[:, 1] + box[:, 3]) / 2
            lab[b, : len(t), 3] = box[:, 2] - box[:, 0]
            lab[b, : len(t), 4] = box[:, 3] - box[:, 1]
    return out, lab


def draw_mosaic_params(rng_np, rng_py, cfg):
    """the random draws of one mosaic sample in the reference's order (dataset_mapper.py:505-520 + data_augment.py:45-62):
    np.random.randint width, height; random.uniform yc, xc; random.uniform angle, scale, shear x, shear y, tx, ty"""
    w = int(rng_np.randint(cfg["MOSAIC_WIDTH_RANGE"][0], cfg["MOSAIC_WIDTH_RANGE"][1] + 1))
    h = int(rng_np.randint(cfg["MOSAIC_HEIGHT_RANGE"][0], cfg["MOSAIC_HEIGHT_RANGE"][1] + 1))
    if max(w / h, h / w) > 1.2:
        h = min(h, w)
        w = int(1.2 * h)
    input_dim = (h, w)
    yc = int(rng_py.uniform(0.5 * input_dim[0], 1.5 * input_dim[0]))
    xc = int(rng_py.uniform(0.5 * input_dim[1], 1.5 * input_dim[1]))
    a = rng_py.uniform(-cfg["DEGREES"], cfg["DEGREES"])
    s = rng_py.uniform(cfg["SCALE"][0], cfg["SCALE"][1])
    shx = rng_py.uniform(-cfg["SHEAR"], cfg["SHEAR"])
    shy = rng_py.uniform(-cfg["SHEAR"], cfg["SHEAR"])
    tx = rng_py.uniform(0.5 - cfg["TRANSLATE"], 0.5 + cfg["TRANSLATE"])
    ty = rng_py.uniform(0.5 - cfg["TRANSLATE"], 0.5 + cfg["TRANSLATE"])
    return input_dim, yc, xc, (a, s, shx, shy, tx, ty)


MOSAIC_DEFAULTS = dict(DEGREES=10.0, TRANSLATE=0.1, SCALE=[0.5, 1.5], SHEAR=2.0, MOSAIC_WIDTH_RANGE=(512, 800),
                       MOSAIC_HEIGHT_RANGE=(512, 800))        # yolov7/config.py:258-272


# ------------------------------------------------------------------------------------------------ mixup
def resize_linear_f64(img, dsize):
    """cv2.resize(img, (w, h)) of a float64 image (default INTER_LINEAR): resize.cpp's generic path - the same source
    index / fraction rule as the 8-bit one, coefficients kept as float32 (1 - fx, fx), rows combined in double:
    horizontal D = S[sx] * a0 + S[sx + 1] * a1, then vertical dst = D0 * b0 + D1 * b1"""
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]

    def coef(src, dst):
        scale = np.float64(src) / np.float64(dst)
        d = np.arange(dst, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        f[lo] = 0.0
        s[lo] = 0
        hi = s >= src - 1
        f[hi] = 0.0
        s[hi] = src - 1
        return s, (np.float32(1.0) - f).astype(np.float64), f.astype(np.float64)
    sx, ax0, ax1 = coef(W, w)
    sy, by0, by1 = coef(H, h)
    sx1, sy1 = np.minimum(sx + 1, W - 1), np.minimum(sy + 1, H - 1)
    src = img.astype(np.float64)
    rows = src[:, sx] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None]
    return rows[sy] * by0[:, None, None] + rows[sy1] * by1[:, None, None]


def adjust_box_anns(bbox, scale_ratio, padw, padh, w_max, h_max):
    """utils/boxes.py:381-384 (in place, as the reference)"""
    bbox[:, 0::2] = np.clip(bbox[:, 0::2] * scale_ratio + padw, 0, w_max)
    bbox[:, 1::2] = np.clip(bbox[:, 1::2] * scale_ratio + padh, 0, h_max)
    return bbox


def mixup_geometry(img_hw, input_dim, jit):
    """sizes of dataset_mapper.py:703-727: first resize (rw1, rh1), the 114 canvas (input_dim), second resize (ow, oh)"""
    r = min(input_dim[0] / img_hw[0], input_dim[1] / img_hw[1])
    rw1, rh1 = int(img_hw[1] * r), int(img_hw[0] * r)
    ow, oh = int(input_dim[1] * jit), int(input_dim[0] * jit)
    return r, (rw1, rh1), (ow, oh)


def mixup(origin_img, origin_labels, img, cp_labels, input_dim, jit, flip, offsets):
    """MyDatasetMapper2.mixup (dataset_mapper.py:686-768) with its draws given: jit = random.uniform(*MSCALE), flip =
    random.uniform(0, 1) > 0.5, the pool sample (img, cp_labels) and offsets = (x_offset, y_offset) - the two
    random.randint draws, taken only when the padded image exceeds the target (mixup_offsets_range gives their ranges)"""
    cp_labels = np.array(cp_labels, np.float64).reshape(-1, 5).copy()
    origin_labels = np.asarray(origin_labels, np.float64).reshape(-1, 5)
    cp_img = np.ones((input_dim[0], input_dim[1], 3)) * 114.0
    r, (rw1, rh1), (ow, oh) = mixup_geometry(img.shape[:2], input_dim, jit)
    cp_img[:rh1, :rw1] = resize_linear_u8(img, (rw1, rh1)).astype(np.float32)
    cp_img = resize_linear_f64(cp_img, (ow, oh))
    cp_scale_ratio = r * jit
    if flip:
        cp_img = cp_img[:, ::-1, :]
    origin_h, origin_w = cp_img.shape[:2]
    target_h, target_w = origin_img.shape[:2]
    padded = np.zeros((max(origin_h, target_h), max(origin_w, target_w), 3)).astype(np.uint8)
    padded[:origin_h, :origin_w] = cp_img                      # float64 -> uint8: truncation
    x_offset, y_offset = offsets
    crop = padded[y_offset: y_offset + target_h, x_offset: x_offset + target_w]
    bo = adjust_box_anns(cp_labels[:, :4], cp_scale_ratio, 0, 0, origin_w, origin_h)
    if flip:
        bo[:, 0::2] = origin_w - bo[:, 0::2][:, ::-1]
    bt = bo.copy()
    bt[:, 0::2] = np.clip(bt[:, 0::2] - x_offset, 0, target_w)
    bt[:, 1::2] = np.clip(bt[:, 1::2] - y_offset, 0, target_h)
    keep = box_candidates(bo.T, bt.T, 5)
    if keep.sum() >= 1.0:
        labels = np.hstack((bt[keep], cp_labels[keep, 4:5]))
        origin_labels = np.vstack((origin_labels, labels))
        out = origin_img.astype(np.float32)
        out = 0.5 * out + 0.5 * crop.astype(np.float32)
        return out.astype(np.uint8), origin_labels
    return origin_img.astype(np.uint8), origin_labels


def mixup_offsets_range(img_hw, input_dim, jit, target_hw):
    """the (exclusive-free) upper bounds of the two random.randint(0, n) draws of dataset_mapper.py:738-743, or None where
    the reference does not draw: (x_max, y_max)"""
    _, _, (ow, oh) = mixup_geometry(img_hw, input_dim, jit)
    ph, pw = max(oh, target_hw[0]), max(ow, target_hw[1])
    return (pw - target_hw[1] - 1 if pw > target_hw[1] else None, ph - target_hw[0] - 1 if ph > target_hw[0] else None)
