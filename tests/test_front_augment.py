"""The detectron2 T.* front of the input pipeline (yolov7/data/detection_utils.py:37-86,158-190,
dataset_mapper.py:615-683) without a GPU:

* the oracle's restatement of Pillow's 8-bit bilinear resampling against the committed golden (made by the real library,
  oracle/gen_golden.py::gold_pil_resize) and, when Pillow is importable, against the library itself on fresh shapes;
* the per-pixel functions the HIP kernels call (yolov7_d2_amd/csrc/pil_resize_core.h) compiled for the HOST
  (tests/native/pil_resize_host.cpp, g++ -ffp-contract=off) against the same - resize + flips + shift, HWC and padded-NCHW
  destination strides;
* the product's host half (random draws in the reference's order, float64 box arithmetic, instance filtering, label rows)
  against the oracle."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import augment_oracle as A  # noqa: E402


def _cases(seed, n):
    r = np.random.RandomState(seed)
    out = [(480, 640, 608, 811), (427, 640, 416, 623), (96, 64, 96, 40), (64, 96, 31, 96), (50, 70, 50, 70)]
    for _ in range(n):
        h, w = r.randint(8, 90, 2)
        nh, nw = r.randint(4, 140, 2)
        out.append((int(h), int(w), int(nh), int(nw)))
    return out


def test_pil_oracle_against_the_golden_made_by_pillow(golden_dir):
    g = np.load(os.path.join(golden_dir, "pil_resize.npz"))
    k = 0
    while f"src{k}" in g.files:
        nh, nw = (int(v) for v in g[f"size{k}"])
        assert np.array_equal(A.pil_resize_bilinear_u8(g[f"src{k}"], nh, nw), g[f"out{k}"]), k
        k += 1
    assert k == 10


def test_pil_oracle_against_the_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    r = np.random.RandomState(5)
    for (h, w, nh, nw) in _cases(11, 25):
        img = r.randint(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(A.pil_resize_bilinear_u8(img, nh, nw), ref), (h, w, nh, nw)


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    so = str(tmp_path_factory.mktemp("pil") / "pil_host.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "native", "pil_resize_host.cpp")],
                   check=True)
    return C.CDLL(so)


def _run_host(lib, img, d, nchw_pad=None):
    h, w = img.shape[:2]
    nh, nw = d["nh"], d["nw"]
    tmp = np.zeros((h, nw, 3), np.uint8)
    if nchw_pad is None:
        dst = np.full((nh, nw, 3), 77, np.uint8)
        strides = (1, 3 * nw, 3)
    else:
        Hp, Wp = nchw_pad
        dst = np.full((3, Hp, Wp), 114, np.uint8)
        strides = (Hp * Wp, Wp, 1)
    lib.pil_front_host(img.ctypes.data_as(C.c_void_p), h, w, nh, nw, int(d["hflip"]), int(d["vflip"]), d["sx"], d["sy"],
                       tmp.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), C.c_longlong(strides[0]),
                       C.c_longlong(strides[1]), C.c_longlong(strides[2]))
    return dst


def test_kernel_pixel_functions_on_the_host_equal_pillow_and_the_oracle(host_lib, golden_dir):
    g = np.load(os.path.join(golden_dir, "pil_resize.npz"))
    for k in range(10):                                         # the golden: plain resizes
        nh, nw = (int(v) for v in g[f"size{k}"])
        d = dict(nh=nh, nw=nw, hflip=False, vflip=False, sx=0, sy=0)
        assert np.array_equal(_run_host(host_lib, np.ascontiguousarray(g[f"src{k}"]), d), g[f"out{k}"]), k
    r = np.random.RandomState(9)
    for i, (h, w, nh, nw) in enumerate(_cases(3, 20)):          # the whole front against the oracle (itself pinned above)
        img = r.randint(0, 256, (h, w, 3), dtype=np.uint8)
        lim = max(1, min(nh, nw, 32) - 1)
        d = dict(nh=nh, nw=nw, hflip=bool(i & 1), vflip=bool(i & 2), sx=int(r.randint(-lim, lim)) if i % 3 else 0,
                 sy=int(r.randint(-lim, lim)) if i % 3 else 0)
        ref = A.front_image(img, d)
        assert np.array_equal(_run_host(host_lib, img, d), ref), (h, w, d)
        Hp, Wp = (nh + 31) // 32 * 32, (nw + 31) // 32 * 32 + 32
        out = _run_host(host_lib, img, d, nchw_pad=(Hp, Wp))
        assert np.array_equal(out[:, :nh, :nw], ref.transpose(2, 0, 1)) and (out[:, nh:] == 114).all() and (out[:, :, nw:] == 114).all()


def test_front_host_logic_equals_the_oracle():
    """draws (same numpy stream -> same values, same number of variates consumed), boxes through the four transforms +
    clipping, filter_empty_instances and preprocess_image's rows"""
    from yolov7_d2_amd.data_pipeline import GpuFrontAugment
    fa = GpuFrontAugment(device="cpu")
    assert fa.output_shape(480, 640, 608, 800) == A.resize_shortest_edge_shape(480, 640, 608, 800) == (600, 800)
    assert fa.output_shape(640, 427, 768, 800) == A.resize_shortest_edge_shape(640, 427, 768, 800)
    r1, r2 = np.random.RandomState(21), np.random.RandomState(21)
    fired = dict(h=0, v=0, s=0)
    images, labels, draws = [], [], []
    for k in range(40):
        h, w = int(r1.randint(200, 700)), int(r1.randint(200, 700))
        assert (int(r2.randint(200, 700)), int(r2.randint(200, 700))) == (h, w)
        d1, d2 = fa.draw((h, w), r1), A.draw_front(r2, (h, w))
        assert d1 == d2
        fired["h"] += d1["hflip"]; fired["v"] += d1["vflip"]; fired["s"] += bool(d1["sx"] or d1["sy"])
        n = int(r1.randint(0, 9)); r2.randint(0, 9)
        x1 = r1.uniform(-5, w - 10, n); y1 = r1.uniform(-5, h - 10, n)
        lab = np.stack([x1, y1, x1 + r1.uniform(0, 300, n), y1 + r1.uniform(0, 300, n), r1.randint(0, 80, n).astype(np.float64)], 1)
        r2.uniform(-5, w - 10, n); r2.uniform(-5, h - 10, n); r2.uniform(0, 300, n); r2.uniform(0, 300, n); r2.randint(0, 80, n)
        if n > 2:
            lab[0, 2:4] = lab[0, :2]                              # an empty box: dropped by filter_empty_instances
        got = fa.boxes(lab, (h, w), d1)
        ref = A.front_boxes(lab[:, :4], (h, w), d2)
        assert np.array_equal(got[:, :4], ref) and np.array_equal(got[:, 4], lab[:, 4])
        images.append(np.zeros((h, w, 3), np.uint8)); labels.append(lab); draws.append(d1)
    assert min(fired.values()) > 5
    rows = fa.label_rows(images, labels, draws)
    for b in range(40):
        box, cls_ = A.filter_empty(A.front_boxes(labels[b][:, :4], images[b].shape[:2], draws[b]), labels[b][:, 4])
        _, ref = A.preprocess_batch([(np.zeros((draws[b]["nh"], draws[b]["nw"], 3), np.uint8), np.concatenate([box.astype(np.float64), cls_[:, None]], 1))])
        assert np.array_equal(rows[b], ref[0]), b


def test_front_refuses_a_cpu_pixel_path():
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.data_pipeline import GpuFrontAugment
    import torch
    fa = GpuFrontAugment(device="cpu")
    with pytest.raises(L.MI355Error):
        fa.apply([torch.zeros(8, 8, 3, dtype=torch.uint8)], [dict(nh=4, nw=4, hflip=False, vflip=False, sx=0, sy=0)])


def test_job_table_layout_and_both_launches_emulated_on_the_host(host_lib):
    """everything of `GpuFrontAugment.make_batch` / `.apply` except hipLaunchKernelGGL: the product's job table (host
    tensors here), mi_pil_resize_jobs_layout from the real library (host code), then the kernels' thread bodies walked over
    the same grid by the host build - against the oracle"""
    import torch
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.data_pipeline import GpuFrontAugment
    fa = GpuFrontAugment(device="cpu")
    r = np.random.RandomState(31)
    shapes = [(120, 160), (107, 160), (160, 120), (96, 64), (50, 70), (64, 64)]
    imgs = [r.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    draws = [dict(nh=104, nw=139, hflip=False, vflip=True, sx=31, sy=0), dict(nh=134, nw=200, hflip=True, vflip=True, sx=-6, sy=10),
             dict(nh=200, nw=150, hflip=True, vflip=False, sx=0, sy=0), dict(nh=96, nw=40, hflip=True, vflip=False, sx=0, sy=-7),
             dict(nh=50, nw=70, hflip=False, vflip=True, sx=5, sy=0), dict(nh=32, nw=64, hflip=False, vflip=False, sx=0, sy=0)]
    timgs = [torch.from_numpy(i) for i in imgs]
    lib = L.lib()

    def run(jobs):
        bh, bv = C.c_int32(0), C.c_int32(0)
        L.check(lib.mi_pil_resize_jobs_layout(jobs, len(jobs), C.byref(bh), C.byref(bv)), "layout")
        assert bv.value == sum((d["nh"] * d["nw"] + 255) // 256 for d in draws)
        assert bh.value == sum((i.shape[0] * d["nw"] + 255) // 256 for i, d in zip(imgs, draws) if d["nw"] != i.shape[1])
        host_lib.pil_emulate_launches(C.cast(jobs, C.c_void_p), len(jobs), bh.value, bv.value)

    # apply(): HWC outputs
    outs = [torch.full((d["nh"], d["nw"], 3), 7, dtype=torch.uint8) for d in draws]
    jobs, tmps = fa._jobs(timgs, draws, [(o.data_ptr(), 1, 3 * d["nw"], 3) for o, d in zip(outs, draws)])
    run(jobs)
    for o, i, d in zip(outs, imgs, draws):
        assert np.array_equal(o.numpy(), A.front_image(i, d)), d
    # make_batch(): one padded NCHW tensor
    Hp, Wp = fa.batch_shape(draws)
    assert (Hp, Wp) == (224, 224)
    out = torch.full((len(imgs), 3, Hp, Wp), 114, dtype=torch.uint8)
    jobs, tmps = fa._jobs(timgs, draws, [(out.data_ptr() + b * 3 * Hp * Wp, Hp * Wp, Wp, 1) for b in range(len(imgs))])
    run(jobs)
    ref, _ = A.preprocess_batch([(A.front_image(i, d), np.zeros((0, 5))) for i, d in zip(imgs, draws)])
    assert np.array_equal(out.numpy(), ref)
    # the layout refuses what the kernels do not serve
    bad = (L.mi_pil_resize_job * 1)()
    C.memmove(bad, jobs, C.sizeof(L.mi_pil_resize_job))
    bad[0].nw, bad[0].nh = 8, 8                                  # 120 x 160 -> 8 x 8: shrinks by more than 8
    bh, bv = C.c_int32(0), C.c_int32(0)
    assert lib.mi_pil_resize_jobs_layout(bad, 1, C.byref(bh), C.byref(bv)) != 0


def _mapper_data(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        h, w = int(rs.randint(60, 140)), int(rs.randint(60, 140))
        img = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
        m = int(rs.randint(0, 6))
        x1 = rs.uniform(0, w - 20, m); y1 = rs.uniform(0, h - 20, m)
        lab = np.stack([x1, y1, np.minimum(x1 + rs.uniform(6, 80, m), w), np.minimum(y1 + rs.uniform(6, 80, m), h),
                        rs.randint(0, 80, m).astype(np.float64)], 1)
        out.append((img, lab))
    return out


MAPPER_FRONT = dict(MIN_SIZE_TRAIN=(64, 96, 128), MAX_SIZE_TRAIN=160, SHIFT_PIXELS=8)
MAPPER_MOSAIC = dict(MOSAIC_WIDTH_RANGE=(96, 160), MOSAIC_HEIGHT_RANGE=(96, 160))
ORACLE_FRONT = dict(min_sizes=(64, 96, 128), max_size=160, max_shifts=8)


@pytest.mark.parametrize("mixup", [False, True])
def test_mapper_call_host_half_equals_the_oracle(mixup):
    """`GpuDatasetMapper.plan` (one MyDatasetMapper2.__call__, dataset_mapper.py:477-640) against the oracle's `mapper_call`
    on the same two random streams: the same decisions (mosaic or not, partners, mixup only when labels survive), the same
    number of variates consumed from numpy's and from Python's generator after every sample, and the same final labels
    (plain: fronted, clipped, empty boxes dropped; mosaic: placed, warped, filtered, mixed)"""
    import random
    from yolov7_d2_amd.data_pipeline import GpuDatasetMapper
    mp = GpuDatasetMapper(device="cpu", enable_mixup=mixup, front_cfg=MAPPER_FRONT, mosaic_cfg=MAPPER_MOSAIC)
    r1n, r1p, r2n, r2p = np.random.RandomState(7), random.Random(8), np.random.RandomState(7), random.Random(8)
    pool, kinds = [], dict(plain=0, mosaic=0, mixed=0)
    for img, lab in _mapper_data(3, 18):
        plan = mp.plan(img, lab, r1n, r1p)
        _, olab, omos = A.mapper_call(pool, (img, lab), r2n, r2p, mcfg=MAPPER_MOSAIC, front_kw=ORACLE_FRONT, enable_mixup=mixup)
        assert plan["mosaic"] == omos and len(mp.pool) == len(pool)
        assert r1n.randint(1 << 30) == r2n.randint(1 << 30) and r1p.random() == r2p.random()      # streams still aligned
        fr = mp.front
        if not omos:
            (im, lb, d), = plan["loads"]
            rows = fr.label_rows([im], [lb], [d])[0]
            _, ref = A.preprocess_batch([(np.zeros((d["nh"], d["nw"], 3), np.uint8), olab)])
            assert np.array_equal(rows, ref[0])
            kinds["plain"] += 1
            continue
        labs = [fr.boxes(lb, tuple(im.shape[:2]), d) for (im, lb, d) in plan["loads"]]
        shapes = [(d["nh"], d["nw"]) for (_, _, d) in plan["loads"]]
        p = plan["params"]
        dim = p["input_dim"]
        M, width, height = mp.mosaic._matrix((dim[0] * 2, dim[1] * 2), p["draws"], [-dim[0] // 2, -dim[1] // 2])
        t = mp.mosaic._warp_labels(mp._mosaic_labels(shapes[:4], labs[:4], p), M, p["draws"][1], width, height)
        if plan["mixup"] is not None:
            mx = plan["mixup"]
            r = min(dim[0] / shapes[4][0], dim[1] / shapes[4][1])
            t, blended = mp.mosaic._mixup_labels(t, labs[4], r, mx["jit"], mx["flip"], mx["x_off"], mx["y_off"],
                                                 (int(dim[0] * mx["jit"]), int(dim[1] * mx["jit"])), (height, width))
            kinds["mixed"] += bool(blended)
        assert np.array_equal(np.asarray(t, np.float64).reshape(-1, 5), olab)
        kinds["mosaic"] += 1
    assert kinds["plain"] >= 5 and kinds["mosaic"] >= 3 and (not mixup or kinds["mixed"] >= 1), kinds


def test_mapper_without_augmentation_is_the_front_only():
    """MyDatasetMapper2.disable_aug() (after DISABLE_AT_ITER): no flag draw, no pool growth, every sample plain"""
    import random
    from yolov7_d2_amd.data_pipeline import GpuDatasetMapper
    mp = GpuDatasetMapper(device="cpu", front_cfg=MAPPER_FRONT, mosaic_cfg=MAPPER_MOSAIC)
    mp.disable_aug()
    r1n, r2n = np.random.RandomState(1), np.random.RandomState(1)
    pool = []
    for img, lab in _mapper_data(5, 8):
        plan = mp.plan(img, lab, r1n, random.Random(0))
        _, _, omos = A.mapper_call(pool, (img, lab), r2n, random.Random(0), mcfg=MAPPER_MOSAIC, front_kw=ORACLE_FRONT, enable_aug=False)
        assert not plan["mosaic"] and not omos and len(mp.pool) == 0 and len(pool) == 0
        assert r1n.randint(1 << 30) == r2n.randint(1 << 30)


def _detr_data(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        h, w = int(rs.randint(90, 220)), int(rs.randint(90, 220))
        img = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
        m = int(rs.randint(0, 6))
        x1 = rs.uniform(0, w - 20, m); y1 = rs.uniform(0, h - 20, m)
        lab = np.stack([x1, y1, np.minimum(x1 + rs.uniform(6, 120, m), w), np.minimum(y1 + rs.uniform(6, 120, m), h),
                        rs.randint(0, 80, m).astype(np.float64)], 1)
        out.append((img, lab))
    return out


DETR_KW = dict(min_sizes=(96, 112, 128, 144), max_size=200, crop=(60, 100), crop_sizes=(80, 100, 120))


def test_detr_mapper_oracle_pixels_are_pillows():
    """the oracle's DetrDatasetMapper chain with the REAL Pillow doing every resize of the flipped / cropped arrays"""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(2)
    seen = dict(crop=0, plain=0, flip=0)
    for img, lab in _detr_data(1, 16):
        out, box, cls_, rec = A.detr_mapper_call(img, lab, rng, **DETR_KW)
        ref = np.flip(img, 1) if rec["flip"] else img
        if rec["crop"] is not None:
            h1, w1, x0, y0, cw, ch = rec["crop"]
            ref = np.asarray(Image.fromarray(np.ascontiguousarray(ref)).resize((w1, h1), Image.BILINEAR))[y0: y0 + ch, x0: x0 + cw]
        H, W = rec["size"]
        ref = np.asarray(Image.fromarray(np.ascontiguousarray(ref)).resize((W, H), Image.BILINEAR))
        assert np.array_equal(out, ref)
        assert len(box) == len(cls_) and (len(box) == 0 or (box[:, 2] <= W).all() and (box[:, 3] <= H).all() and (box >= 0).all())
        seen["crop"] += rec["crop"] is not None; seen["plain"] += rec["crop"] is None; seen["flip"] += rec["flip"]
    assert min(seen.values()) >= 3, seen


def test_detr_mapper_host_half_and_emulated_launches_equal_the_oracle(host_lib):
    """`GpuDetrMapper`: draws (stream alignment after every sample), boxes, and the two stages of jobs - first resize of the
    crop samples from the (mirrored) source, final resize from the crop window / the source into [3, h, w] - walked by the
    host build of the kernels' thread bodies, against the oracle"""
    import torch
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.data_pipeline import GpuDetrMapper
    mp = GpuDetrMapper(device="cpu", **DETR_KW)
    r1, r2 = np.random.RandomState(4), np.random.RandomState(4)
    data = _detr_data(6, 12)
    plans, refs = [], []
    for img, lab in data:
        p = mp.plan(img.shape[:2], r1)
        out, box, cls_, rec = A.detr_mapper_call(img, lab, r2, **DETR_KW)
        assert r1.randint(1 << 30) == r2.randint(1 << 30)
        assert p["flip"] == rec["flip"] and p["crop"] == rec["crop"] and p["size"] == rec["size"]
        b2, c2 = mp.boxes(lab, p)
        assert np.array_equal(b2, box) and np.array_equal(c2, cls_)
        plans.append(p); refs.append(out)
    assert sum(p["crop"] is not None for p in plans) >= 3 and sum(p["crop"] is None for p in plans) >= 3
    timgs = [torch.from_numpy(i) for i, _ in data]
    mids = [torch.zeros(p["crop"][0], p["crop"][1], 3, dtype=torch.uint8) if p["crop"] is not None else None for p in plans]
    outs = [torch.full((3,) + tuple(p["size"]), 9, dtype=torch.uint8) for p in plans]
    J1, J2 = mp._stage_jobs(timgs, plans, mids, outs)

    def alloc(h, w):
        t = torch.empty(h, w, 3, dtype=torch.uint8)
        return t, t.data_ptr()
    lib = L.lib()
    for specs in (J1, J2):
        jobs, owners = mp._table(specs, alloc)
        bh, bv = C.c_int32(0), C.c_int32(0)
        L.check(lib.mi_pil_resize_jobs_layout(jobs, len(jobs), C.byref(bh), C.byref(bv)), "layout")
        host_lib.pil_emulate_launches(C.cast(jobs, C.c_void_p), len(jobs), bh.value, bv.value)
    for o, ref in zip(outs, refs):
        assert np.array_equal(o.numpy(), ref.transpose(2, 0, 1))
    with pytest.raises(L.MI355Error):
        mp.make_batch(timgs, [l for _, l in data])


@pytest.mark.parametrize("mixup,distort", [(False, False), (True, False), (False, True), (True, True)])
def test_mapper_run_plumbing_on_the_host(host_lib, monkeypatch, mixup, distort):
    """`GpuDatasetMapper.run` - which loads go through the front, the pool view handed to the mosaic mapper (fronted images,
    fronted labels, the mixup partner's index), the assembly of the MIXED batch - executed on host tensors: the front's
    launches are walked by the host build of its thread bodies, the mosaic mapper's three launches are stood in for by the
    oracle's `mosaic_sample` / `mixup` on the same view (they have their own bit-exact GPU tests); the batch must equal
    `preprocess_batch` of the oracle's `mapper_call` outputs, pixels and rows"""
    import random
    import torch
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.data_pipeline import GpuDatasetMapper, GpuFrontAugment, GpuMosaicMapper
    lib = L.lib()

    def launch(self, jobs, keep):
        bh, bv = C.c_int32(0), C.c_int32(0)
        L.check(lib.mi_pil_resize_jobs_layout(jobs, len(jobs), C.byref(bh), C.byref(bv)), "layout")
        host_lib.pil_emulate_launches(C.cast(jobs, C.c_void_p), len(jobs), bh.value, bv.value)
        self._keep = keep

    def fake_mosaic(self, pool, groups, params, mixups=None, float_src=False):
        assert float_src == distort              # (the front's DISTORTION makes every loaded image float32 in the reference)
        samples = []
        for b, (grp, p) in enumerate(zip(groups, params)):
            img, t = A.mosaic_sample([pool.images[i].numpy() for i in grp], [pool.labels[i] for i in grp], p["input_dim"], p["yc"], p["xc"], p["draws"],
                                     float_src=float_src)
            mx = mixups[b] if mixups is not None else None
            if mx is not None and len(t):
                img, t = A.mixup(img, t, pool.images[mx["idx"]].numpy(), pool.labels[mx["idx"]], p["input_dim"], mx["jit"], mx["flip"], (mx["x_off"], mx["y_off"]),
                                 float_src=float_src)
            samples.append((img, t))
        out, rows = A.preprocess_batch(samples)
        return torch.from_numpy(out), torch.from_numpy(rows), [tuple(p["input_dim"]) for p in params]
    monkeypatch.setattr(GpuFrontAugment, "_launch", launch)
    monkeypatch.setattr(GpuFrontAugment, "_check_device", lambda self, images: None)
    monkeypatch.setattr(GpuMosaicMapper, "make_batch", fake_mosaic)
    fcfg = dict(MAPPER_FRONT, SATURATION=distort, BRIGHTNESS=distort, DISTORTION=distort)     # (yolox_s.yaml switches the three on together)
    okw = dict(ORACLE_FRONT, saturation=distort, brightness=distort, distortion=(0.1, 1.5, 1.5) if distort else None)
    mp = GpuDatasetMapper(device="cpu", enable_mixup=mixup, front_cfg=fcfg, mosaic_cfg=MAPPER_MOSAIC)
    r1n, r1p, r2n, r2p = np.random.RandomState(17), random.Random(18), np.random.RandomState(17), random.Random(18)
    pool, kinds = [], []
    data = _mapper_data(40 + mixup, 24)
    for k in range(0, 24, 6):
        chunk = data[k: k + 6]
        plans = [mp.plan(torch.from_numpy(i), l, r1n, r1p) for i, l in chunk]
        out, rows, sizes = mp.run(plans)
        ref = [A.mapper_call(pool, (i, l), r2n, r2p, mcfg=MAPPER_MOSAIC, front_kw=okw, enable_mixup=mixup) for i, l in chunk]
        ref_img, ref_rows = A.preprocess_batch([(r[0], r[1]) for r in ref])
        kinds += [r[2] for r in ref]
        assert tuple(out.shape) == ref_img.shape
        assert np.array_equal(out.numpy(), ref_img), k
        assert np.array_equal(rows.numpy(), ref_rows), k
        assert all(tuple(s) == r[0].shape[:2] for s, r in zip(sizes, ref) if not r[2])
    assert kinds.count(True) >= 4 and kinds.count(False) >= 8


def test_detr_mapper_make_batch_on_the_host(host_lib, monkeypatch):
    """`GpuDetrMapper.make_batch` end to end on host tensors (its allocations, the two stages of launches in order, the
    boxes): the launches walked by the host build of the thread bodies, against the oracle"""
    import torch
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.data_pipeline import GpuDetrMapper, GpuFrontAugment
    lib = L.lib()

    def launch(self, jobs, keep):
        bh, bv = C.c_int32(0), C.c_int32(0)
        L.check(lib.mi_pil_resize_jobs_layout(jobs, len(jobs), C.byref(bh), C.byref(bv)), "layout")
        host_lib.pil_emulate_launches(C.cast(jobs, C.c_void_p), len(jobs), bh.value, bv.value)
        self._keep = keep
    monkeypatch.setattr(GpuFrontAugment, "_launch", launch)
    monkeypatch.setattr(GpuDetrMapper, "_check_device", lambda self, images: None)
    mp = GpuDetrMapper(device="cpu", **DETR_KW)
    r1, r2 = np.random.RandomState(14), np.random.RandomState(14)
    data = _detr_data(16, 10)
    res = mp.make_batch([torch.from_numpy(i) for i, _ in data], [l for _, l in data], r1)
    crops = 0
    for (img, lab), (o, box, cls_) in zip(data, res):
        ref, rbox, rcls, rec = A.detr_mapper_call(img, lab, r2, **DETR_KW)
        crops += rec["crop"] is not None
        assert np.array_equal(o.numpy(), ref.transpose(2, 0, 1))
        assert np.array_equal(box, rbox) and np.array_equal(cls_, rcls)
    assert 0 < crops < len(data)


def test_front_apply_and_make_batch_bodies_on_the_host(host_lib, monkeypatch):
    """the bodies of `GpuFrontAugment.apply` / `.make_batch` (allocations, destination strides, the launch call) on host tensors"""
    import torch
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.data_pipeline import GpuFrontAugment
    lib = L.lib()

    def launch(self, jobs, keep):
        bh, bv = C.c_int32(0), C.c_int32(0)
        L.check(lib.mi_pil_resize_jobs_layout(jobs, len(jobs), C.byref(bh), C.byref(bv)), "layout")
        host_lib.pil_emulate_launches(C.cast(jobs, C.c_void_p), len(jobs), bh.value, bv.value)
        self._keep = keep
    monkeypatch.setattr(GpuFrontAugment, "_launch", launch)
    monkeypatch.setattr(GpuFrontAugment, "_check_device", lambda self, images: None)
    fa = GpuFrontAugment(MAPPER_FRONT, device="cpu")
    r = np.random.RandomState(8)
    data = _mapper_data(9, 7)
    draws = [fa.draw(i.shape[:2], r) for i, _ in data]
    timgs = [torch.from_numpy(i) for i, _ in data]
    for o, (i, _), d in zip(fa.apply(timgs, draws), data, draws):
        assert np.array_equal(o.numpy(), A.front_image(i, d))
    out, rows, sizes = fa.make_batch(timgs, [l for _, l in data], draws)
    samples = []
    for (i, l), d in zip(data, draws):
        box, cls_ = A.filter_empty(A.front_boxes(l[:, :4], i.shape[:2], d), l[:, 4])
        samples.append((A.front_image(i, d), np.concatenate([box.astype(np.float64), cls_[:, None]], 1)))
    ref, ref_rows = A.preprocess_batch(samples)
    assert np.array_equal(out.numpy(), ref) and np.array_equal(rows.numpy(), ref_rows) and sizes == [(d["nh"], d["nw"]) for d in draws]


def test_colour_blends_equal_numpys_arithmetic(host_lib, monkeypatch):
    """INPUT.COLOR_JITTER: detectron2's RandomSaturation / RandomBrightness (BlendTransform on the uint8 image) between the
    flips and the shift - the oracle runs the d2 lines verbatim in numpy (fp64 grey + fp32 image product for the saturation,
    fp32 for the brightness, truncation); the kernels' thread bodies (host build) must reproduce every pixel; the draws take
    their place in the stream (after the flips, before the shift)"""
    import torch
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.data_pipeline import GpuFrontAugment
    lib = L.lib()

    def launch(self, jobs, keep):
        bh, bv = C.c_int32(0), C.c_int32(0)
        L.check(lib.mi_pil_resize_jobs_layout(jobs, len(jobs), C.byref(bh), C.byref(bv)), "layout")
        host_lib.pil_emulate_launches(C.cast(jobs, C.c_void_p), len(jobs), bh.value, bv.value)
        self._keep = keep
    monkeypatch.setattr(GpuFrontAugment, "_launch", launch)
    monkeypatch.setattr(GpuFrontAugment, "_check_device", lambda self, images: None)
    for sat, bri in ((True, True), (True, False), (False, True)):
        fa = GpuFrontAugment(dict(MAPPER_FRONT, SATURATION=sat, BRIGHTNESS=bri), device="cpu")
        r1, r2 = np.random.RandomState(3), np.random.RandomState(3)
        data = _mapper_data(12, 8)
        draws = []
        for i, _ in data:
            d1 = fa.draw(i.shape[:2], r1)
            d2 = A.draw_front(r2, i.shape[:2], saturation=sat, brightness=bri, **ORACLE_FRONT)
            assert d1 == d2 and (("sat" in d1) == sat) and (("bri" in d1) == bri)
            draws.append(d1)
        assert r1.randint(1 << 30) == r2.randint(1 << 30)
        outs = fa.apply([torch.from_numpy(i) for i, _ in data], draws)
        for o, (i, _), d in zip(outs, data, draws):
            ref = A.front_image(i, d)
            assert np.array_equal(o.numpy(), ref), d
            assert not np.array_equal(ref, A.front_image(i, dict(d, sat=None, bri=None)))      # the blends did something


# ------------------------------------------------------------------------------------------------ YOLOFRandomDistortion
def test_distortion_oracle_against_the_reference_golden(golden_dir):
    """oracle distort_image / draw_distortion against the reference's own YOLOFDistortTransform.apply_image run by path
    (golden distortion.npz, oracle/gen_golden.py::gold_distortion): same pixels, same number of variates drawn.  cv2.cvtColor
    is the restatement on both sides: this pins the numpy arithmetic, its dtype rules and the random stream - not OpenCV"""
    g = np.load(os.path.join(golden_dir, "distortion.npz"))
    for k, (hw, seed) in enumerate((((33, 47), 3), ((64, 40), 4), ((21, 90), 5), ((50, 50), 6))):
        img = np.random.RandomState(200 + k).randint(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
        if k == 3:
            img[:, :25] = img[:, :25, :1]
            img[30:, :] = 255
        r = np.random.RandomState(seed)
        d = A.draw_distortion(r, 0.1, 1.5, 1.5)
        assert np.array_equal(A.distort_image(img, *d), g[f"out{k}"]), k
        assert float(r.uniform()) == float(g[f"next{k}"]), k         # five variates consumed, as the reference consumed


def test_hsv_restatement_sanity():
    """OpenCV's 8-bit RGB <-> HSV as restated (no cv2 to pin it): against colorsys within the quantisation (H in 2-degree
    steps: one unit of 180), grey pixels have S = 0 and survive the round trip exactly, pure colours land on their hue"""
    import colorsys
    r = np.random.RandomState(1)
    img = r.randint(0, 256, (40, 50, 3), dtype=np.uint8)
    hsv = A.rgb2hsv_u8(img).astype(np.float64)
    ref = np.array([[colorsys.rgb_to_hsv(*(img[y, x] / 255.0)) for x in range(50)] for y in range(40)])
    dh = np.abs(hsv[..., 0] - ref[..., 0] * 180)
    dh = np.minimum(dh, 180 - dh)
    assert dh.max() <= 1.0 and np.abs(hsv[..., 1] - ref[..., 1] * 255).max() <= 1.0 and np.array_equal(hsv[..., 2], img.max(-1))
    grey = np.repeat(r.randint(0, 256, (8, 8, 1), dtype=np.uint8), 3, axis=2)
    assert (A.rgb2hsv_u8(grey)[..., 1] == 0).all() and np.array_equal(A.hsv2rgb_u8(A.rgb2hsv_u8(grey)), grey)
    pure = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0]]], np.uint8)
    assert A.rgb2hsv_u8(pure)[0, :, 0].tolist() == [0, 60, 120, 30]
    assert np.array_equal(A.hsv2rgb_u8(A.rgb2hsv_u8(pure)), pure)
    assert np.abs(A.hsv2rgb_u8(A.rgb2hsv_u8(img)).astype(int) - img.astype(int)).max() <= 6     # (the hue quantisation)


def test_distortion_pixel_function_on_the_host_equals_the_oracle(host_lib):
    """the per-pixel function the HIP kernel calls (csrc/pil_resize_core.h pil_distort, host build) through the product's own
    job table: resize + flips + saturation + brightness + DISTORTION + shift against oracle.front_image - bit for bit"""
    import torch
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.data_pipeline import GpuFrontAugment
    lib = L.lib()
    fa = GpuFrontAugment(dict(MIN_SIZE_TRAIN=(48, 64, 96), MAX_SIZE_TRAIN=128, SHIFT_PIXELS=6, SATURATION=True, BRIGHTNESS=True,
                              DISTORTION=True), device="cpu")
    r1, r2 = np.random.RandomState(31), np.random.RandomState(31)
    imgs, draws = [], []
    for k in range(10):
        h, w = int(r1.randint(40, 120)), int(r1.randint(40, 120)); r2.randint(40, 120); r2.randint(40, 120)
        img = r1.randint(0, 256, (h, w, 3), dtype=np.uint8); r2.randint(0, 256, (h, w, 3), dtype=np.uint8)
        if k % 3 == 0:
            img[: h // 2] = img[: h // 2, :, 1:2]                   # grey half
        d1 = fa.draw((h, w), r1)
        d2 = A.draw_front(r2, (h, w), min_sizes=(48, 64, 96), max_size=128, max_shifts=6, saturation=True, brightness=True,
                          distortion=(0.1, 1.5, 1.5))
        assert d1 == d2 and d1["dis"] is not None
        imgs.append(img); draws.append(d1)
    assert any(d["dis"][0] > 0 for d in draws) and any(d["dis"][0] <= 0 for d in draws)
    timgs = [torch.from_numpy(i) for i in imgs]
    outs = [torch.full((d["nh"], d["nw"], 3), 7, dtype=torch.uint8) for d in draws]
    jobs, tmps = fa._jobs(timgs, draws, [(o.data_ptr(), 1, 3 * d["nw"], 3) for o, d in zip(outs, draws)])
    assert all(j.color == 7 for j in jobs)
    bh, bv = C.c_int32(0), C.c_int32(0)
    L.check(lib.mi_pil_resize_jobs_layout(jobs, len(jobs), C.byref(bh), C.byref(bv)), "layout")
    host_lib.pil_emulate_launches(C.cast(jobs, C.c_void_p), len(jobs), bh.value, bv.value)
    for o, i, d in zip(outs, imgs, draws):
        assert np.array_equal(o.numpy(), A.front_image(i, d)), d


def test_float_source_resize_oracle():
    """cv2.resize's float path as restated (what the mosaic / mixup branches run once the distortion has made the image
    float32): equals the float64 restatement to float32 rounding, the identity size copies, a constant image stays within one
    ulp of its value (and therefore truncates to c or c - 1: the build-dependent case the docstring names)"""
    r = np.random.RandomState(2)
    img = r.randint(0, 256, (37, 53, 3)).astype(np.float32)
    for dsize in ((80, 60), (20, 17), (53, 90)):
        a, b = A.resize_linear_f32(img, dsize), A.resize_linear_f64(img.astype(np.float64), dsize)
        assert a.dtype == np.float32 and np.abs(a - b).max() < 1e-3
    assert np.array_equal(A.resize_linear_f32(img, (53, 37)), img)
    flat = np.full((16, 16, 3), 200, np.float32)
    out = A.resize_linear_f32(flat, (37, 29))
    assert np.abs(out - 200).max() < 1e-4 and set(np.unique(out.astype(np.uint8))) <= {199, 200}
