"""GPU (-m gpu): the input pipeline kernels (mosaic resize + paste, affine warp + transpose + pad) against
oracle/augment_oracle.py on the same draws: uint8 images and label rows bit-exact; then one training step fed by it."""
import os
import random

import numpy as np
import pytest
import torch

import augment_oracle as A
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.data_pipeline import GpuMosaicMapper, MosaicPool

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pool(seed, sizes):
    rs = np.random.RandomState(seed)
    pool, imgs, labs = MosaicPool(DEV), [], []
    for (h, w) in sizes:
        # smooth content + noise: interpolation differences would show, a pure-noise image hides nothing either
        yy, xx = np.mgrid[0:h, 0:w]
        base = (127 + 90 * np.sin(xx / 17.0 + rs.uniform(0, 6)) * np.cos(yy / 23.0 + rs.uniform(0, 6)))[..., None]
        img = np.clip(base + rs.randint(-30, 31, (h, w, 3)), 0, 255).astype(np.uint8)
        n = rs.randint(0, 8)
        x1 = rs.uniform(0, w - 30, n); y1 = rs.uniform(0, h - 30, n)
        lab = np.stack([x1, y1, np.minimum(x1 + rs.uniform(6, 300, n), w), np.minimum(y1 + rs.uniform(6, 300, n), h),
                        rs.randint(0, 80, n).astype(np.float64)], 1)
        pool.append(torch.from_numpy(img), lab)
        imgs.append(img); labs.append(lab)
    return pool, imgs, labs


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_mosaic_batch_bit_exact(seed):
    sizes = [(480, 640), (375, 500), (640, 427), (333, 500), (720, 1280), (1080, 1920), (200, 150), (427, 640), (512, 512),
             (900, 700), (96, 2000), (1500, 110)]
    pool, imgs, labs = _pool(10 + seed, sizes)
    mapper = GpuMosaicMapper(device=DEV)
    rng_np, rng_py = np.random.RandomState(20 + seed), random.Random(30 + seed)
    B = 6
    groups = [tuple(int(i) for i in rng_np.randint(0, len(sizes), 4)) for _ in range(B)]
    params = [mapper.draw(rng_np, rng_py) for _ in range(B)]
    out, rows, dims = mapper.make_batch(pool, groups, params)
    torch.cuda.synchronize()
    samples = []
    for g, p in zip(groups, params):
        samples.append(A.mosaic_sample([imgs[i] for i in g], [labs[i] for i in g], p["input_dim"], p["yc"], p["xc"], p["draws"]))
    ref_img, ref_rows = A.preprocess_batch(samples)
    assert tuple(out.shape) == ref_img.shape and out.dtype == torch.uint8
    got = out.cpu().numpy()
    for b in range(B):
        bad = int((got[b] != ref_img[b]).sum())
        assert bad == 0, (b, bad, dims[b], np.abs(got[b].astype(int) - ref_img[b].astype(int)).max())
    assert np.array_equal(rows.cpu().numpy(), ref_rows)
    assert any(len(s[1]) for s in samples) and out.shape[2] % 32 == 0 and out.shape[3] % 32 == 0


@pytest.mark.parametrize("seed", [0, 1])
def test_mosaic_mixup_batch_bit_exact(seed):
    sizes = [(480, 640), (375, 500), (640, 427), (333, 500), (720, 1280), (200, 150), (427, 640), (512, 512), (900, 700)]
    pool, imgs, labs = _pool(40 + seed, sizes)
    mapper = GpuMosaicMapper(device=DEV)
    rng_np, rng_py = np.random.RandomState(50 + seed), random.Random(60 + seed)
    B = 6
    groups = [tuple(int(i) for i in rng_np.randint(0, len(sizes), 4)) for _ in range(B)]
    params = [mapper.draw(rng_np, rng_py) for _ in range(B)]
    mixups = []
    for p in params:
        d = p["input_dim"]
        tgt = (d[0] * 2 + (-d[0] // 2) * 2, d[1] * 2 + (-d[1] // 2) * 2)       # random_perspective's output size
        mixups.append(mapper.draw_mixup(pool, d, tgt, rng_np, rng_py))
    mixups[2] = None
    out, rows, dims = mapper.make_batch(pool, groups, params, mixups)
    torch.cuda.synchronize()
    samples, nblend = [], 0
    for g, p, mx in zip(groups, params, mixups):
        img, t = A.mosaic_sample([imgs[i] for i in g], [labs[i] for i in g], p["input_dim"], p["yc"], p["xc"], p["draws"])
        if mx is not None and len(t):
            n0 = len(t)
            img, t = A.mixup(img, t, imgs[mx["idx"]], labs[mx["idx"]], p["input_dim"], mx["jit"], mx["flip"], (mx["x_off"], mx["y_off"]))
            nblend += len(t) > n0
        samples.append((img, t))
    ref_img, ref_rows = A.preprocess_batch(samples)
    got = out.cpu().numpy()
    assert nblend >= 2
    for b in range(B):
        bad = int((got[b] != ref_img[b]).sum())
        assert bad == 0, (b, bad, dims[b], mixups[b], np.abs(got[b].astype(int) - ref_img[b].astype(int)).max())
    assert np.array_equal(rows.cpu().numpy(), ref_rows)


def test_draw_order_matches_oracle():
    mapper = GpuMosaicMapper(device=DEV)
    p = mapper.draw(np.random.RandomState(4), random.Random(5))
    dim, yc, xc, draws = A.draw_mosaic_params(np.random.RandomState(4), random.Random(5), A.MOSAIC_DEFAULTS)
    assert (p["input_dim"], p["yc"], p["xc"], p["draws"]) == (dim, yc, xc, draws)


def test_degenerate_quadrants_and_identity():
    """mosaic centre at the canvas corner range limits (a quadrant of zero area), and the paste of an image that needs no
    resizing (scale 1: the fixed-point pass must reproduce the bytes)"""
    pool, imgs, labs = _pool(7, [(600, 600)] * 4)
    mapper = GpuMosaicMapper(device=DEV)
    for yc, xc in ((300, 300), (899, 899), (300, 899)):
        p = dict(input_dim=(600, 600), yc=yc, xc=xc, draws=(0.0, 1.0, 0.0, 0.0, 0.5, 0.5))
        out, rows, _ = mapper.make_batch(pool, [(0, 1, 2, 3)], [p])
        torch.cuda.synchronize()
        ref, t = A.mosaic_sample(imgs, labs, (600, 600), yc, xc, p["draws"])
        bi, br = A.preprocess_batch([(ref, t)])
        assert np.array_equal(out.cpu().numpy(), bi) and np.array_equal(rows.cpu().numpy(), br)
    # identity warp of the canvas centre: the output is the centre crop of the canvas, whose quadrants are raw image crops
    canvas, _ = A.mosaic4(imgs, labs, (600, 600), 600, 600)
    assert np.array_equal(canvas[:600, :600], imgs[0]) and np.array_equal(canvas[600:, 600:], imgs[3])


def test_train_step_fed_by_the_pipeline():
    import yolov7_d2_amd as M
    from yolov7_d2_amd.engine import NativeTrainer
    pool, _, _ = _pool(3, [(480, 640), (375, 500), (640, 427), (333, 500), (500, 375), (427, 640)])
    mapper = GpuMosaicMapper(dict(MOSAIC_WIDTH_RANGE=(256, 320), MOSAIC_HEIGHT_RANGE=(256, 320)), device=DEV)
    rng_np, rng_py = np.random.RandomState(1), random.Random(2)
    torch.manual_seed(0)
    model = M.build_model(M.yolox_s_cfg(device=DEV))
    trainer = NativeTrainer(model, lr=1e-3, use_graph=False, input_u8=True)
    for _ in range(2):
        groups = [tuple(int(i) for i in rng_np.randint(0, len(pool), 4)) for _ in range(2)]
        imgs, rows, _ = mapper.make_batch(pool, groups, [mapper.draw(rng_np, rng_py) for _ in range(2)])
        st = trainer.load_batch(imgs, rows)
        trainer.step(st)
        losses = trainer.losses(st)[:4]
        assert bool(torch.isfinite(losses).all()) and float(losses[0]) > 0
    with pytest.raises(L.MI355Error):
        GpuMosaicMapper(device="cpu").make_batch(pool, [(0, 1, 2, 3)], [mapper.draw(rng_np, rng_py)])


def _run_child(name, timeout=300, expect="bit-identical"):
    """the four input-pipeline compositions run in a child process (their own CUDA context); all four passed on a device in
    round 4's first GPU call and are plain assertions since: a mismatch, a crash or a timeout FAILS the suite"""
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and expect in r.stdout, (r.stdout + r.stderr)[-2000:]


@pytest.mark.gpu
def test_front_augment_kernels_equal_pillow_and_the_oracle():
    """The detectron2 T.* front (T.ResizeShortestEdge = Pillow's 8-bit bilinear resampling, T.RandomFlip x2, YOLOFRandomShift;
    yolov7/data/detection_utils.py:37-86) on the GPU: `GpuFrontAugment.apply` (HWC images for the mosaic pool) and
    `.make_batch` (the mapper with the mosaic off + preprocess_image) bit-identical to the oracle and to Pillow itself
    (tests/front_augment_gpu_child.py: eight images at COCO sizes, every combination of flips / shift / skipped passes)."""
    _run_child("front_augment_gpu_child.py", 240)


@pytest.mark.gpu
def test_dataset_mapper_batches_equal_the_oracle():
    """`GpuDatasetMapper.make_batch` = MyDatasetMapper2.__call__ per sample (dataset_mapper.py:477-640: front, mosaic flag,
    partners, four pastes, random_perspective, mixup) + preprocess_image over the MIXED batch, against the oracle's
    `mapper_call` on the same random streams: pixels and label rows bit-identical (tests/mapper_gpu_child.py: eight batches
    of six, with and without mixup, with and without the colour entries of yolox_s.yaml - RandomSaturation, RandomBrightness,
    YOLOFRandomDistortion - after which the mosaic / mixup resizes run cv2's float path)."""
    _run_child("mapper_gpu_child.py")


@pytest.mark.gpu
def test_jpeg_decoder_equals_pillow():
    """`GpuJpegDecoder.decode` = detectron2 utils.read_image for JPEG files (dataset_mapper.py:646-648): host Huffman
    decoding in the library, IDCT and up-sampling / colour / EXIF kernels on the GPU, against Pillow's decode of the same
    files (COCO sizes, 4:4:4 / 4:2:2 / 4:2:0, grey, optimised tables, restart markers, progressive, EXIF orientations), BGR +
    orientation and plain RGB."""
    _run_child("jpeg_gpu_child.py")


@pytest.mark.gpu
def test_detr_mapper_equals_the_oracle():
    """`GpuDetrMapper.make_batch` = DetrDatasetMapper.__call__ (dataset_mapper.py:804-900: flip, the resize + crop branch, the
    final resize, boxes / Instances) at the reference's DETR sizes, pixels and boxes bit-identical to the oracle
    (tests/detr_mapper_gpu_child.py)."""
    _run_child("detr_mapper_gpu_child.py")
