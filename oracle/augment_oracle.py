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


def resize_linear_f32(img, dsize):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) of a FLOAT32 image (what the mosaic / mixup branches resize
    once YOLOFRandomDistortion has turned the loaded image into float32): resize.cpp's generic path with WT = float - the
    source index / fraction rule of the 8-bit path, coefficients (1 - fx, fx) in float32, horizontal D = S[sx] * a0 +
    S[sx + 1] * a1, vertical dst = D0 * b0 + D1 * b1, every product and sum rounded to float32 (no fused multiply-add).
    PARITY UNPINNED (no cv2), and inherently build-dependent where it matters: on a flat region S * a0 + S * a1 lands one
    ulp under S as often as not, the uint8 assignment that follows truncates, and whether a given OpenCV build contracts the
    expression into an FMA decides the result - the restatement documents one consistent choice."""
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    src = np.asarray(img, np.float32)
    if (w, h) == (W, H):
        return src.copy()

    def coef(n_src, n_dst):
        scale = np.float64(n_src) / np.float64(n_dst)
        d = np.arange(n_dst, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s_ = np.floor(f).astype(np.int64)
        f = (f - s_.astype(np.float32)).astype(np.float32)
        lo = s_ < 0
        f[lo] = 0.0
        s_[lo] = 0
        hi = s_ >= n_src - 1
        f[hi] = 0.0
        s_[hi] = n_src - 1
        return s_, (np.float32(1.0) - f).astype(np.float32), f
    sx, ax0, ax1 = coef(W, w)
    sy, by0, by1 = coef(H, h)
    sx1, sy1 = np.minimum(sx + 1, W - 1), np.minimum(sy + 1, H - 1)
    rows = (src[:, sx] * ax0[None, :, None]).astype(np.float32) + (src[:, sx1] * ax1[None, :, None]).astype(np.float32)
    rows = rows.astype(np.float32)
    out = (rows[sy] * by0[:, None, None]).astype(np.float32) + (rows[sy1] * by1[:, None, None]).astype(np.float32)
    return out.astype(np.float32)


# ---- YOLOFRandomDistortion (data/transforms/augmentation_impl.py:115-133, transform.py:250-308): cv2.cvtColor's 8-bit
# RGB <-> HSV both ways around three float32 scalings.  OpenCV's published 8-bit algorithms (modules/imgproc/src/
# color_hsv.simd.hpp): RGB2HSV_b = integer arithmetic over two 12-bit reciprocal tables (H in [0, 180)); HSV2RGB_b = the
# float formula on (h, s / 255, v / 255), the result x 255 rounded to nearest-even and saturated.  PARITY UNPINNED (no cv2).
# cv2.COLOR_RGB2HSV takes channel 0 as R whatever the image's real order is (the YAML's FORMAT is BGR: the reference
# distorts "the wrong way round" and so does this restatement).
HSV_SHIFT = 12


def _hsv_div_tables():
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    sdiv[1:] = np.rint((255 << HSV_SHIFT) / (1.0 * i)).astype(np.int64)        # saturate_cast<int>(double) = cvRound
    hdiv[1:] = np.rint((180 << HSV_SHIFT) / (6.0 * i)).astype(np.int64)
    return sdiv, hdiv


def rgb2hsv_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_RGB2HSV), uint8 HWC: RGB2HSV_b with hrange 180"""
    sdiv, hdiv = _hsv_div_tables()
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s_ = (diff * sdiv[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * hdiv[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s_ & 255, v], axis=-1).astype(np.uint8)


def hsv2rgb_u8(hsv):
    """cv2.cvtColor(hsv, cv2.COLOR_HSV2RGB), uint8 HWC: HSV2RGB_b = HSV2RGB_native on (h, s * (1 / 255), v * (1 / 255)) with
    hscale = 6 / 180 in float32, then saturate_cast<uchar>(x * 255) (round to nearest even)"""
    f32 = np.float32
    h = hsv[..., 0].astype(f32)
    s_ = (hsv[..., 1].astype(f32) * f32(1.0 / 255.0)).astype(f32)
    v = (hsv[..., 2].astype(f32) * f32(1.0 / 255.0)).astype(f32)
    h = (h * f32(6.0 / 180.0)).astype(f32)
    h = np.where(h >= 6, (h - f32(6)).astype(f32), h)          # (h8 <= 255: at most one turn)
    sector = np.floor(h).astype(np.int64)
    h = (h - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    h = np.where(bad, f32(0), h)
    one = f32(1)
    t0 = v
    t1 = (v * (one - s_).astype(f32)).astype(f32)
    t2 = (v * (one - (s_ * h).astype(f32)).astype(f32)).astype(f32)
    t3 = (v * (one - (s_ * (one - h).astype(f32)).astype(f32)).astype(f32)).astype(f32)
    tab = np.stack([t0, t1, t2, t3], axis=-1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])      # (b, g, r) table rows per sector
    pick = lambda k: np.take_along_axis(tab, sd[sector][..., k][..., None], axis=-1)[..., 0]
    b, g, r = pick(0), pick(1), pick(2)
    grey = s_ == 0
    b, g, r = (np.where(grey, v, c) for c in (b, g, r))
    q = lambda c: np.clip(np.rint((c * f32(255)).astype(f32)), 0, 255).astype(np.uint8)
    return np.stack([q(r), q(g), q(b)], axis=-1)


def draw_distortion(rng_np, hue, saturation, exposure):
    """YOLOFDistortTransform.apply_image's draws (transform.py:268-270, _rand_scale :293-308): (dhue, dsat, dexp)"""
    def rand_scale(upper):
        scale = rng_np.uniform(low=1, high=upper)
        return scale if rng_np.rand() > 0.5 else 1 / scale
    dhue = rng_np.uniform(low=-hue, high=hue)
    dsat = rand_scale(saturation)
    dexp = rand_scale(exposure)
    return float(dhue), float(dsat), float(dexp)


def distort_image(img, dhue, dsat, dexp):
    """YOLOFDistortTransform.apply_image (transform.py:272-288) on a uint8 HWC image, numpy's float32 arithmetic as the
    reference runs it; returns uint8 (the reference returns the same integers as float32)"""
    x = np.asarray(rgb2hsv_u8(img), dtype=np.float32) / 255.
    x[:, :, 1] *= dsat
    x[:, :, 2] *= dexp
    H = x[:, :, 0] + dhue * 179 / 255.
    if dhue > 0:
        H[H > 1.0] -= 1.0
    else:
        H[H < 0.0] += 1.0
    x[:, :, 0] = H
    x = (x * 255).clip(0, 255).astype(np.uint8)
    return hsv2rgb_u8(x)


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


def mosaic4(imgs, labels, input_dim, yc, xc, float_src=False):
    """dataset_mapper.py:523-590: four images resized by min(h / h0, w / w0), pasted around (xc, yc) on a 114 canvas;
    labels rows (x1, y1, x2, y2, cls) scaled, shifted and clipped to the canvas.  float_src: the loaded images are float32
    (YOLOFRandomDistortion ran: transform.py:286) - cv2.resize then takes its float path and the assignment into the uint8
    canvas truncates"""
    img4 = np.full((input_dim[0] * 2, input_dim[1] * 2, 3), 114, dtype=np.uint8)
    labels4 = []
    for i in range(4):
        img, _labels = imgs[i], np.asarray(labels[i], np.float64).reshape(-1, 5)
        h0, w0 = img.shape[:2]
        scale = min(1. * input_dim[0] / h0, 1. * input_dim[1] / w0)
        if float_src:
            img = resize_linear_f32(img, (int(w0 * scale), int(h0 * scale)))       # (float32; img4[...] = img truncates)
        else:
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


def mosaic_sample(imgs, labels, input_dim, yc, xc, draws, float_src=False):
    """one training sample of the mosaic branch (mixup off, as configs/coco/yolox_s.yaml:57-62): HWC uint8 image of size
    input_dim and its (x1, y1, x2, y2, cls) rows"""
    img4, labels4 = mosaic4(imgs, labels, input_dim, yc, xc, float_src)
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


def mixup(origin_img, origin_labels, img, cp_labels, input_dim, jit, flip, offsets, float_src=False):
    """MyDatasetMapper2.mixup (dataset_mapper.py:686-768) with its draws given: jit = random.uniform(*MSCALE), flip =
    random.uniform(0, 1) > 0.5, the pool sample (img, cp_labels) and offsets = (x_offset, y_offset) - the two
    random.randint draws, taken only when the padded image exceeds the target (mixup_offsets_range gives their ranges)"""
    cp_labels = np.array(cp_labels, np.float64).reshape(-1, 5).copy()
    origin_labels = np.asarray(origin_labels, np.float64).reshape(-1, 5)
    cp_img = np.ones((input_dim[0], input_dim[1], 3)) * 114.0
    r, (rw1, rh1), (ow, oh) = mixup_geometry(img.shape[:2], input_dim, jit)
    # (float_src: the pool image went through YOLOFRandomDistortion and is float32 - cv2.resize's float path, no rounding)
    cp_img[:rh1, :rw1] = (resize_linear_f32(img, (rw1, rh1)) if float_src else resize_linear_u8(img, (rw1, rh1))).astype(np.float32)
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


# ------------------------------------------------------------------------------------------------ detectron2 T.* front
# The augmentations `MyDatasetMapper2._load_image_with_annos` (dataset_mapper.py:642-683) applies to EVERY loaded image -
# the current one and the three mosaic samples - before the mosaic branch, and all there is in the non-mosaic branch
# (dataset_mapper.py:615-640): `build_normal_augmentation` (data/detection_utils.py:37-86) = T.ResizeShortestEdge,
# T.RandomFlip (horizontal), T.RandomFlip (vertical), RandomSaturation, RandomBrightness, YOLOFRandomDistortion, YOLOFRandomShift
# (data/transforms/augmentation_impl.py:168-191, transform.py:341-410).  detectron2 is un-vendored (readme.md:178 "latest"):
# ResizeShortestEdge.get_output_shape / ResizeTransform / HFlipTransform / Transform.apply_box are restated from its
# published source (detectron2/data/transforms/{augmentation_impl,transform}.py, fvcore/transforms/transform.py) - PARITY
# UNPINNED for those few lines of arithmetic; the pixel work of ResizeTransform is Pillow's Image.resize(BILINEAR), which IS
# installed here and on the GPU box (12.2.0): `pil_resize_bilinear_u8` restates libImaging/Resample.c's 8-bit path and is
# PINNED against the real library (tests/test_augment_oracle.py, golden `pil_resize.npz` made by oracle/gen_golden.py).
PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1.0) over the full
    axis: (bounds int32 [out, 2] = (first input index, tap count), coefficients int32 [out, ksize], ksize)"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = np.zeros(ksize, np.float64)
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = 0.0
        for x in range(xmax):       # (the C loop's own left-to-right summation order)
            ww += w[x]
        if ww != 0.0:
            w[:xmax] = w[:xmax] / ww
        for x in range(ksize):
            v = w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pil_pass(a, bounds, kk, axis):
    """one 8-bit resampling pass along `axis` (0 rows / 1 columns) of an HWC uint8 array"""
    a = np.moveaxis(a, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + a.shape[1:], np.uint8)
    for i in range(bounds.shape[0]):
        x0, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = np.full(a.shape[1:], 1 << (PIL_PRECISION_BITS - 1), np.int64)
        for t in range(n):
            acc += a[x0 + t] * int(kk[i, t])
        out[i] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear_u8(img, new_h, new_w):
    """PIL.Image.fromarray(img).resize((new_w, new_h), Image.BILINEAR) for an HWC uint8 image: horizontal pass first (skipped
    when the width is unchanged), then the vertical pass on its 8-bit result (skipped when the height is unchanged)"""
    h, w = img.shape[:2]
    out = img
    if new_w != w:
        b, k, _ = pil_bilinear_coeffs(w, new_w)
        out = _pil_pass(out, b, k, 1)
    if new_h != h:
        b, k, _ = pil_bilinear_coeffs(h, new_h)
        out = _pil_pass(out, b, k, 0)
    return np.ascontiguousarray(out)


def resize_shortest_edge_shape(oldh, oldw, short_edge_length, max_size):
    """detectron2 ResizeShortestEdge.get_output_shape (d2 upstream)"""
    h, w = oldh, oldw
    size = short_edge_length * 1.0
    scale = size / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh = newh * scale
        neww = neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def draw_front(rng_np, hw, min_sizes=(416, 512, 608, 768), max_size=800, sample_style="choice", hflip_prob=0.5, vflip_prob=0.5,
               shift_prob=0.5, max_shifts=32, hflip=True, vflip=True, shift=True, saturation=False, brightness=False,
               distortion=None):
    """the random numbers of the chain in the reference's order (AugmentationList: each get_transform sees the image the
    previous transforms produced): ResizeShortestEdge (np.random.choice / randint), RandomFlip x2 (np.random.uniform),
    YOLOFRandomShift (uniform, then randint x, randint y when it fires).  NOTE on the order of the stream with
    distortion=(hue, saturation, exposure): AugmentationList calls get_transform AND applies the transform to the image
    before the next augmentation draws, and YOLOFDistortTransform draws inside apply_image - so its three draws sit between
    RandomBrightness's and YOLOFRandomShift's, exactly where this function takes them"""
    h, w = hw
    if sample_style == "range":
        size = int(rng_np.randint(min_sizes[0], min_sizes[1] + 1))
    else:
        size = int(rng_np.choice(min_sizes))
    nh, nw = (h, w) if size == 0 else resize_shortest_edge_shape(h, w, size, max_size)
    d = dict(nh=nh, nw=nw, hflip=False, vflip=False, sx=0, sy=0)
    if hflip:
        d["hflip"] = bool(rng_np.uniform(0, 1.0) < hflip_prob)
    if vflip:
        d["vflip"] = bool(rng_np.uniform(0, 1.0) < vflip_prob)
    if saturation:                  # detection_utils.py:70-73: RandomSaturation(0.8, 1.2), RandomBrightness(0.8, 1.2) (d2 upstream)
        d["sat"] = float(rng_np.uniform(0.8, 1.2))
    if brightness:
        d["bri"] = float(rng_np.uniform(0.8, 1.2))
    if distortion is not None:      # detection_utils.py:75-80: YOLOFRandomDistortion(hue, saturation, exposure) - its draws
        hue, sat_hi, exp_hi = distortion                           # happen in YOLOFDistortTransform.apply_image (transform.py:268-270)
        d["dis"] = draw_distortion(rng_np, hue, sat_hi, exp_hi)
    if shift and max_shifts > 0:
        if rng_np.uniform(0, 1.0) < shift_prob:
            d["sx"] = int(rng_np.randint(low=-max_shifts, high=max_shifts))
            d["sy"] = int(rng_np.randint(low=-max_shifts, high=max_shifts))
    return d


def front_image(img, d):
    """ResizeTransform.apply_image (PIL bilinear) -> HFlipTransform -> VFlipTransform -> YOLOFShiftTransform.apply_image
    (zeros where the shifted image does not reach)"""
    out = pil_resize_bilinear_u8(img, d["nh"], d["nw"])
    if d["hflip"]:
        out = np.flip(out, axis=1)
    if d["vflip"]:
        out = np.flip(out, axis=0)
    if d.get("sat") is not None:    # detectron2 RandomSaturation.get_transform + BlendTransform.apply_image, verbatim numpy
        w = d["sat"]
        grayscale = out.dot([0.299, 0.587, 0.114])[:, :, np.newaxis]
        x = out.astype(np.float32)
        x = (1 - w) * grayscale + w * x
        out = np.clip(x, 0, 255).astype(np.uint8)
    if d.get("bri") is not None:    # RandomBrightness: BlendTransform(src_image=0, src_weight=1 - w, dst_weight=w)
        w = d["bri"]
        x = out.astype(np.float32)
        x = (1 - w) * 0 + w * x
        out = np.clip(x, 0, 255).astype(np.uint8)
    if d.get("dis") is not None:    # YOLOFDistortTransform.apply_image; the reference returns this image as float32
        out = distort_image(out, *d["dis"])
    sx, sy = d["sx"], d["sy"]
    if sx or sy:
        new = np.zeros_like(out)
        new_x, orig_x = (0, -sx) if sx < 0 else (sx, 0)
        new_y, orig_y = (0, -sy) if sy < 0 else (sy, 0)
        hh, ww = out.shape[0] - abs(sy), out.shape[1] - abs(sx)
        new[new_y:new_y + hh, new_x:new_x + ww] = out[orig_y:orig_y + hh, orig_x:orig_x + ww]
        out = new
    return np.ascontiguousarray(out)


def _apply_box(boxes, fn):
    """fvcore Transform.apply_box: the four corners through apply_coords, then the axis-aligned hull"""
    idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
    coords = np.asarray(boxes, np.float64).reshape(-1, 4)[:, idxs].reshape(-1, 2)
    coords = fn(coords).reshape((-1, 4, 2))
    return np.concatenate((coords.min(axis=1), coords.max(axis=1)), axis=1)


def front_boxes(boxes_xyxy, hw, d):
    """transform_instance_annotations (data/detection_utils.py:158-190) for every box: TransformList.apply_box = each
    transform's apply_box in turn, then clip(min=0) and the minimum with (w, h, w, h) of the FINAL image"""
    h, w = hw
    nh, nw = d["nh"], d["nw"]
    b = np.asarray(boxes_xyxy, np.float64).reshape(-1, 4)
    if len(b) == 0:
        return b

    def resize(c):
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
        b = _apply_box(b, resize)
    if d["hflip"]:
        b = _apply_box(b, hf)
    if d["vflip"]:
        b = _apply_box(b, vf)
    if d["sx"] or d["sy"]:
        b = _apply_box(b, sh)
    b = b.clip(min=0)
    return np.minimum(b, np.array([nw, nh, nw, nh], np.float64))


def filter_empty(boxes_xyxy, classes, threshold=1e-5):
    """detectron2 annotations_to_instances (Boxes = float32) + filter_empty_instances(by_box): keep width, height > 1e-5"""
    b = np.asarray(boxes_xyxy, np.float64).reshape(-1, 4).astype(np.float32)
    keep = ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)
    return b[keep], np.asarray(classes)[keep]


# ------------------------------------------------------------------------------------------------ the whole mapper call
def mapper_call(pool, cur, rng_np, rng_py, mcfg=None, front_kw=None, enable_mosaic=True, enable_aug=True, enable_mixup=False,
                mscale=(0.5, 1.5), num_images=4, capacity=1000):
    """MyDatasetMapper2.__call__ (dataset_mapper.py:477-640) for one dataset_dict: `pool` = the mosaic_pool deque as a list of
    (image, labels) (mutated: the current entry is appended), cur = (HWC uint8 image, labels float64 [n, 5] xyxy + cls).
    Returns (image, labels, mosaic?) - labels of a plain sample after filter_empty_instances, of a mosaic sample as
    random_perspective / mixup leave them.  Every image is loaded through the T.* front (`_load_image_with_annos`)."""
    mcfg = dict(MOSAIC_DEFAULTS, **(mcfg or {}))
    front_kw = front_kw or {}
    fsrc = front_kw.get("distortion") is not None     # every loaded image is float32 then (YOLOFDistortTransform's return)
    flag, partners = 0, None
    if enable_mosaic and enable_aug:
        if len(pool) > num_images:
            flag = int(rng_np.randint(2))
            if flag == 1:
                partners = [pool[int(i)] for i in rng_np.choice(len(pool), num_images - 1)]
        pool.append(cur)
        if len(pool) > capacity:
            pool.pop(0)

    def load(entry):
        img, lab = entry
        lab = np.asarray(lab, np.float64).reshape(-1, 5)
        d = draw_front(rng_np, img.shape[:2], **front_kw)
        return front_image(img, d), np.concatenate([front_boxes(lab[:, :4], img.shape[:2], d), lab[:, 4:5]], 1)
    img, lab = load(cur)
    if flag == 1 and partners is not None:
        input_dim, yc, xc, draws = draw_mosaic_params(rng_np, rng_py, mcfg)
        loaded = [(img, lab)] + [load(e) for e in partners]
        out, t = mosaic_sample([x[0] for x in loaded], [x[1] for x in loaded], input_dim, yc, xc, draws, float_src=fsrc)
        if enable_mixup and len(t):
            jit = rng_py.uniform(*mscale)
            flip = rng_py.uniform(0, 1) > 0.5
            cp_img, cp_lab = load(pool[int(rng_np.choice(len(pool), 1)[0])])
            xm, ym = mixup_offsets_range(cp_img.shape[:2], input_dim, jit, out.shape[:2])
            y_off = rng_py.randint(0, ym) if ym is not None else 0
            x_off = rng_py.randint(0, xm) if xm is not None else 0
            out, t = mixup(out, t, cp_img, cp_lab, input_dim, jit, flip, (x_off, y_off), float_src=fsrc)
        return out, np.asarray(t, np.float64).reshape(-1, 5), True
    box, cls_ = filter_empty(lab[:, :4], lab[:, 4])
    return img, np.concatenate([box.astype(np.float64), np.asarray(cls_, np.float64)[:, None]], 1), False


# ------------------------------------------------------------------------------------------------ DetrDatasetMapper
def detr_mapper_call(img, labels, rng_np, min_sizes=(480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832), max_size=1333,
                     sample_style="choice", crop=(384, 600), crop_sizes=(400, 500, 600)):
    """DetrDatasetMapper.__call__ (yolov7/data/dataset_mapper.py:836-900, training; configs/coco/detr/*.yaml INPUT) after
    read_image: with INPUT.CROP on, np.random.rand() > 0.5 picks the plain list [T.RandomFlip(), T.ResizeShortestEdge(min_size,
    max_size)], otherwise RandomFlip, T.ResizeShortestEdge([400, 500, 600]), T.RandomCrop("absolute_range", (384, 600)), then the
    final resize (build_transform_gen :777-800).  The transforms are detectron2's (un-vendored; restated: RandomFlip /
    ResizeShortestEdge as above, RandomCrop.get_crop_size / get_transform, CropTransform.apply_coords); the pixels are Pillow's
    bilinear resampling of the FLIPPED (and cropped) image.  labels float64 [n, 5] (x1, y1, x2, y2, cls).
    Returns (HWC uint8 image, float32 boxes [m, 4], classes [m], record of the draws)."""
    import sys
    lab = np.asarray(labels, np.float64).reshape(-1, 5)
    h, w = img.shape[:2]
    take_crop = crop is not None and not (rng_np.rand() > 0.5)
    b = lab[:, :4].copy()
    rec = dict(crop=None)
    rec["flip"] = bool(rng_np.uniform(0, 1.0) < 0.5)
    if rec["flip"]:
        img = np.flip(img, axis=1)
        ww = w

        def hf(c):
            c[:, 0] = ww - c[:, 0]
            return c
        b = _apply_box(b, hf) if len(b) else b

    def resize(img, b, size, mx):
        h, w = img.shape[:2]
        nh, nw = resize_shortest_edge_shape(h, w, size, mx)
        out = pil_resize_bilinear_u8(np.ascontiguousarray(img), nh, nw)

        def sc(c):
            c[:, 0] = c[:, 0] * (nw * 1.0 / w)
            c[:, 1] = c[:, 1] * (nh * 1.0 / h)
            return c
        return out, (_apply_box(b, sc) if len(b) else b)
    if take_crop:
        img, b = resize(img, b, int(rng_np.choice(crop_sizes)), sys.maxsize)
        h, w = img.shape[:2]
        ch = int(rng_np.randint(min(h, crop[0]), min(h, crop[1]) + 1))
        cw = int(rng_np.randint(min(w, crop[0]), min(w, crop[1]) + 1))
        y0 = int(rng_np.randint(h - ch + 1))
        x0 = int(rng_np.randint(w - cw + 1))
        rec["crop"] = (h, w, x0, y0, cw, ch)
        img = img[y0: y0 + ch, x0: x0 + cw]

        def cr(c):
            c[:, 0] -= x0
            c[:, 1] -= y0
            return c
        b = _apply_box(b, cr) if len(b) else b
    if sample_style == "range":
        size = int(rng_np.randint(min_sizes[0], min_sizes[1] + 1))
    else:
        size = int(rng_np.choice(min_sizes))
    img, b = resize(img, b, size, max_size)
    H, W = img.shape[:2]
    rec["size"] = (H, W)
    if len(b):
        b = np.minimum(b.clip(min=0), np.array([W, H, W, H], np.float64))
    box, cls_ = filter_empty(b, lab[:, 4])
    return np.ascontiguousarray(img), box, cls_, rec
