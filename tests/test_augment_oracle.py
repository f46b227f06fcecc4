"""CPU: the input-pipeline oracle (oracle/augment_oracle.py).  Label / matrix logic against the reference's own
random_perspective run by path (when /root/reference is present) and against the committed golden; the restated OpenCV
pixel arithmetic through properties OpenCV's definitions imply (identity, integer shifts, constant images, border)."""
import os
import random

import numpy as np
import pytest

import augment_oracle as A
import ref_loader


def _case(seed, n=12, hw=(1200, 1400)):
    r = np.random.RandomState(seed)
    img = r.randint(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    x1 = r.uniform(0, hw[1] - 50, n); y1 = r.uniform(0, hw[0] - 50, n)
    t = np.stack([x1, y1, x1 + r.uniform(3, 400, n), y1 + r.uniform(3, 400, n), r.randint(0, 80, n).astype(np.float64)], 1)
    return img, t


DRAWS = [(3.7, 0.8, 1.2, -0.7, 0.45, 0.55), (-9.5, 1.45, -2.0, 2.0, 0.6, 0.4), (0.0, 1.0, 0.0, 0.0, 0.5, 0.5), (10.0, 0.5, 0.3, 0.1, 0.41, 0.59)]


def test_random_perspective_labels_against_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "random_perspective.npz"))
    for k, d in enumerate(DRAWS):
        img, t = _case(100 + k)
        border = [-img.shape[0] // 4, -img.shape[1] // 4]
        M, w, h = A.perspective_matrix(img.shape[:2], d, border)
        out = A.perspective_labels(t.copy(), M, d[1], w, h)
        assert np.array_equal(M, g[f"M{k}"])                      # float64, same operation order: identical bits
        assert np.array_equal(out, g[f"labels{k}"])
        assert (w, h) == tuple(g[f"wh{k}"])


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")
def test_random_perspective_against_reference_by_path(monkeypatch):
    m = ref_loader.load_data_augment()
    for k, d in enumerate(DRAWS):
        img, t = _case(100 + k, hw=(240, 320))
        t[:, :4] *= 0.2
        border = [-img.shape[0] // 4, -img.shape[1] // 4]
        seq = list(d)
        monkeypatch.setattr(m.random, "uniform", lambda a, b, _s=seq: _s.pop(0))
        ref_img, ref_t = m.random_perspective(img.copy(), t.copy(), degrees=10, translate=0.1, scale=(0.5, 1.5), shear=2.0,
                                              border=border)
        got_img, got_t = A.random_perspective(img.copy(), t.copy(), d, border)
        assert np.array_equal(ref_t, got_t)
        assert np.array_equal(ref_img, got_img)                   # (same warp restatement on both sides: pins the call, not cv2)


def test_resize_properties():
    r = np.random.RandomState(1)
    img = r.randint(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(A.resize_linear_u8(img, (53, 37)), img)
    c = np.full((20, 30, 3), 77, np.uint8)
    assert np.array_equal(A.resize_linear_u8(c, (61, 47)), np.full((47, 61, 3), 77, np.uint8))      # weights sum to 1
    up = A.resize_linear_u8(img, (106, 74))                       # x2: dst 2k / 2k+1 are 3:1 / 1:3 blends of src k-1, k, k+1
    k = np.arange(1, 52)
    ref = (3 * img[:, k].astype(np.int32) + img[:, k + 1].astype(np.int32))
    row = A.resize_linear_u8(img, (106, 37))[:, 2 * k + 1]
    assert np.abs(row.astype(np.int32) * 4 - ref).max() <= 2      # exact 0.75 / 0.25 coefficients, one rounding
    assert up.shape == (74, 106, 3)
    s, a0, a1 = A.resize_coeffs(53, 20)
    assert ((a0 + a1) == 2048).all() and s.min() >= 0 and s.max() <= 52


def test_warp_affine_properties():
    r = np.random.RandomState(2)
    img = r.randint(0, 256, (40, 56, 3), dtype=np.uint8)
    I = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(A.warp_affine_u8(img, I, (56, 40)), img)
    T = np.array([[1.0, 0, 5], [0, 1.0, -3]])                     # integer shift: exact copy + constant border
    out = A.warp_affine_u8(img, T, (56, 40))
    assert np.array_equal(out[:37, 5:], img[3:, :51])
    assert (out[:, :4] == 114).all() and (out[38:] == 114).all()
    tab = A.bilinear_tab()
    assert (tab.sum(-1) == 32768).all() and tab[0, 0, 0] == 32768 and tab[16, 16].tolist() == [8192] * 4
    H = np.array([[1.0, 0, 0.5], [0, 1.0, 0]])                    # half-pixel shift: mean of horizontal neighbours
    out = A.warp_affine_u8(img, H, (56, 40)).astype(np.int32)
    ref = (img[:, :-1].astype(np.int32) + img[:, 1:].astype(np.int32) + 1) >> 1
    assert np.abs(out[:, 1:] - ref).max() <= 0


def test_mosaic_geometry_and_batch():
    rs = np.random.RandomState(3)
    imgs = [rs.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((480, 640), (375, 500), (640, 427), (333, 500))]
    labs = []
    for im in imgs:
        n = rs.randint(0, 6)
        x1 = rs.uniform(0, im.shape[1] - 40, n); y1 = rs.uniform(0, im.shape[0] - 40, n)
        labs.append(np.stack([x1, y1, x1 + rs.uniform(8, 200, n), y1 + rs.uniform(8, 200, n), rs.randint(0, 80, n).astype(float)], 1))
    py = random.Random(5)
    dim, yc, xc, draws = A.draw_mosaic_params(np.random.RandomState(4), py, A.MOSAIC_DEFAULTS)
    assert 512 <= dim[0] <= 800 and max(dim[1] / dim[0], dim[0] / dim[1]) <= 1.2 + 1e-9
    img4, l4 = A.mosaic4(imgs, labs, dim, yc, xc)
    assert img4.shape == (2 * dim[0], 2 * dim[1], 3)
    # every pasted quadrant is the matching crop of the resized image; the rest of the canvas is 114
    mask = np.zeros(img4.shape[:2], bool)
    for i, im in enumerate(imgs):
        sc = min(dim[0] / im.shape[0], dim[1] / im.shape[1])
        rz = A.resize_linear_u8(im, (int(im.shape[1] * sc), int(im.shape[0] * sc)))
        (x1a, y1a, x2a, y2a), (x1b, y1b, x2b, y2b) = A.mosaic_placement(i, rz.shape[1], rz.shape[0], xc, yc, dim)
        assert (x2a - x1a, y2a - y1a) == (x2b - x1b, y2b - y1b)
        assert np.array_equal(img4[y1a:y2a, x1a:x2a], rz[y1b:y2b, x1b:x2b])
        mask[y1a:y2a, x1a:x2a] = True
    assert (img4[~mask] == 114).all()
    out, lab = A.mosaic_sample(imgs, labs, dim, yc, xc, draws)
    assert out.shape == (dim[0], dim[1], 3) and lab.shape[1] == 5
    assert (lab[:, 0] >= 0).all() and (lab[:, 2] <= dim[1]).all() and (lab[:, 3] <= dim[0]).all()
    batch, rows = A.preprocess_batch([(out, lab), (out[:500, :480], lab[:1])])
    assert batch.shape[2] % 32 == 0 and batch.shape[3] % 32 == 0 and batch.dtype == np.uint8
    assert (batch[1, :, 500:, :] == 114).all() and (batch[1, :, :, 480:] == 114).all()
    assert np.array_equal(batch[0, :, : dim[0], : dim[1]], out.transpose(2, 0, 1))
    n = len(lab)
    assert (rows[0, n:] == 0).all() and np.allclose(rows[0, :n, 3], lab[:, 2] - lab[:, 0], rtol=1e-6)


def test_mixup_properties():
    rs = np.random.RandomState(9)
    origin = rs.randint(0, 256, (300, 360, 3), dtype=np.uint8)
    ol = np.array([[10.0, 20, 100, 200, 3]])
    img = rs.randint(0, 256, (240, 320, 3), dtype=np.uint8)
    cl = np.array([[30.0, 40, 200, 220, 7], [0.0, 0, 3, 3, 9]])
    dim = (300, 360)
    # jit 1, no flip, no offset: the blend partner is the 114 canvas with the resized image in its top-left corner
    out, lab = A.mixup(origin, ol, img, cl, dim, 1.0, False, (0, 0))
    r, (rw1, rh1), (ow, oh) = A.mixup_geometry(img.shape[:2], dim, 1.0)
    assert (ow, oh) == (360, 300) and np.array_equal(A.resize_linear_f64(np.ones((5, 7, 3)) * 114.0, (9, 4)), np.ones((4, 9, 3)) * 114.0)
    part = np.full((300, 360, 3), 114, np.uint8)
    part[:rh1, :rw1] = A.resize_linear_u8(img, (rw1, rh1))
    ref = (0.5 * origin.astype(np.float32) + 0.5 * part.astype(np.float32)).astype(np.uint8)
    assert np.array_equal(out, ref)
    assert len(lab) == 2 and lab[1, 4] == 7 and np.allclose(lab[1, :4], cl[0, :4] * r)       # the 3 x 3 box is filtered
    # flip mirrors the partner and the boxes; a larger jit crops at the drawn offset
    out2, lab2 = A.mixup(origin, ol, img, cl, dim, 1.4, True, (17, 5))
    xm, ym = A.mixup_offsets_range(img.shape[:2], dim, 1.4, origin.shape[:2])
    assert xm == int(360 * 1.4) - 360 - 1 and ym == int(300 * 1.4) - 300 - 1 and out2.shape == origin.shape
    assert len(lab2) >= 1 and (lab2[:, 0] >= 0).all() and (lab2[:, 2] <= 360).all()
    # no surviving box: the image comes back unblended
    out3, lab3 = A.mixup(origin, ol, img, cl[1:], dim, 1.0, False, (0, 0))
    assert np.array_equal(out3, origin) and len(lab3) == 1
    # a small jit leaves the partner smaller than the target: zeros (not 114) outside it, as the reference pads
    out4, _ = A.mixup(origin, ol, img, cl, dim, 0.5, False, (0, 0))
    assert np.array_equal(out4[200:, 250:], (0.5 * origin[200:, 250:].astype(np.float32)).astype(np.uint8))
    assert A.mixup_offsets_range(img.shape[:2], dim, 0.5, origin.shape[:2]) == (None, None)


def test_host_mirror_label_logic_matches_oracle():
    """yolov7_d2_amd.data_pipeline's host side (the random draws in the reference's order, mosaic placement, the combined
    matrix, label warp / filter, the matrix inversion handed to the warp kernel, mixup labels) against the oracle, which is
    pinned to the reference's own random_perspective - no device needed"""
    from yolov7_d2_amd.data_pipeline import GpuMosaicMapper
    mp = GpuMosaicMapper.__new__(GpuMosaicMapper)
    mp.cfg = dict(A.MOSAIC_DEFAULTS, MSCALE=[0.5, 1.5], NUM_IMAGES=4, PERSPECTIVE=0.0)
    for seed in range(6):
        p = mp.draw(np.random.RandomState(seed), random.Random(100 + seed))
        dim, yc, xc, draws = A.draw_mosaic_params(np.random.RandomState(seed), random.Random(100 + seed), A.MOSAIC_DEFAULTS)
        assert (p["input_dim"], p["yc"], p["xc"], p["draws"]) == (dim, yc, xc, draws)
        for i in range(4):
            w, h = 300 + 37 * i + seed, 280 + 11 * i
            (a, _b) = A.mosaic_placement(i, w, h, xc, yc, dim)
            got_a, got_b = mp._placement(i, w, h, xc, yc, dim)
            assert got_a == a and got_b == _b[:2]
        border = [-dim[0] // 2, -dim[1] // 2]
        M, width, height = mp._matrix((2 * dim[0], 2 * dim[1]), draws, border)
        Mr, wr, hr = A.perspective_matrix((2 * dim[0], 2 * dim[1]), draws, border)
        assert np.array_equal(M, Mr) and (width, height) == (wr, hr)
        assert np.array_equal(np.array(mp._invert(M)).reshape(2, 3), A.invert_affine(M[:2]))
        _, t = _case(200 + seed, n=15, hw=(2 * dim[0], 2 * dim[1]))
        assert np.array_equal(mp._warp_labels(t.copy(), M, draws[1], width, height), A.perspective_labels(t.copy(), Mr, draws[1], wr, hr))
    # mixup labels: same boxes, same filter, same stacking as the oracle's mixup on a dummy image pair
    rs = np.random.RandomState(3)
    origin = np.zeros((300, 360, 3), np.uint8)
    img = rs.randint(0, 256, (240, 320, 3), dtype=np.uint8)
    ol = np.array([[10.0, 20, 100, 200, 3]])
    cl = np.array([[30.0, 40, 200, 220, 7], [0.0, 0, 3, 3, 9], [100.0, 100, 310, 230, 1]])
    for jit, flip, off in ((1.0, False, (0, 0)), (1.4, True, (17, 5)), (0.6, True, (0, 0))):
        _, ref = A.mixup(origin, ol, img, cl, (300, 360), jit, flip, off)
        r, _, (ow, oh) = A.mixup_geometry(img.shape[:2], (300, 360), jit)
        got, blended = mp._mixup_labels(ol.copy(), cl, r, jit, flip, off[0], off[1], (oh, ow), (300, 360))
        assert np.array_equal(got, ref) and blended == (len(ref) > 1)
