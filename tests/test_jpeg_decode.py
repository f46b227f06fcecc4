"""Baseline JPEG decoding of the input pipeline (what detectron2's utils.read_image does for the
reference's mapper, yolov7/data/dataset_mapper.py:646-648) without a GPU:

* oracle/jpeg_oracle.py (numpy restatement of libjpeg-turbo's default path) against the Pillow installed here, on files
  Pillow writes (qualities, 4:4:4 / 4:2:2 / 4:2:0, grey, optimised Huffman tables, restart intervals, EXIF orientations,
  progressive scan scripts) and
  against the committed golden `jpeg_decode.npz` (files + Pillow's own decode);
* the product: libmi355det.so's HOST half (mi_jpeg_parse, mi_jpeg_huffman: coefficient blocks equal to the oracle's) and
  its DEVICE half's thread bodies compiled for the host and walked over the job table the library itself laid out
  (tests/native/jpeg_host_test.cpp) - bit-identical to Pillow, RGB and d2's BGR + orientation."""
import ctypes as C
import io
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpeg_oracle as J  # noqa: E402
from yolov7_d2_amd import _lib as L  # noqa: E402


def _smooth(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 9.0 + yy / 17.0), 127 + 100 * np.cos(xx / 13.0), 127 + 100 * np.sin(yy / 7.0)], -1)
    return np.clip(base + rng.randint(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)


def _files():
    """(tag, jpeg bytes) written by Pillow"""
    from PIL import Image
    rng = np.random.RandomState(3)
    out = []
    for (h, w) in [(16, 16), (17, 23), (64, 96), (33, 70), (8, 8), (1, 1), (50, 3), (3, 50)]:
        for sub in (0, 1, 2):
            q = (30, 75, 95)[(h + w + sub) % 3]
            img = _smooth(rng, h, w) if (h + sub) % 2 else rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="JPEG", quality=q, subsampling=sub)
            out.append((f"{h}x{w} sub{sub} q{q}", buf.getvalue()))
    buf = io.BytesIO(); Image.fromarray(_smooth(rng, 40, 60)[..., 0]).save(buf, format="JPEG", quality=80); out.append(("grey", buf.getvalue()))
    buf = io.BytesIO(); Image.fromarray(_smooth(rng, 70, 45)).save(buf, format="JPEG", quality=85, subsampling=2, optimize=True); out.append(("optimised", buf.getvalue()))
    for kw in (dict(restart_marker_blocks=3), dict(restart_marker_rows=1)):
        buf = io.BytesIO()
        try:
            Image.fromarray(_smooth(rng, 64, 96)).save(buf, format="JPEG", quality=85, subsampling=2, **kw)
            out.append(("restart " + str(kw), buf.getvalue()))
        except TypeError:
            pass
    for o in range(1, 9):
        ex = Image.Exif(); ex[0x0112] = o
        buf = io.BytesIO(); Image.fromarray(_smooth(rng, 40, 56)).save(buf, format="JPEG", quality=90, exif=ex.tobytes())
        out.append((f"exif {o}", buf.getvalue()))
    buf = io.BytesIO(); Image.fromarray(_smooth(rng, 240, 320)).save(buf, format="JPEG", quality=90, subsampling=2); out.append(("240x320", buf.getvalue()))
    for (h, w, sub, q) in [(64, 96, 2, 85), (33, 70, 1, 75), (17, 23, 0, 95), (50, 3, 2, 60), (120, 90, 2, 30)]:      # progressive (SOF2)
        buf = io.BytesIO(); Image.fromarray(_smooth(rng, h, w)).save(buf, format="JPEG", quality=q, subsampling=sub, progressive=True)
        out.append((f"progressive {h}x{w} sub{sub} q{q}", buf.getvalue()))
    buf = io.BytesIO(); Image.fromarray(_smooth(rng, 40, 60)[..., 0]).save(buf, format="JPEG", quality=80, progressive=True); out.append(("progressive grey", buf.getvalue()))
    from jpeg_craft import craft_jpeg                              # samplings Pillow cannot write, random coefficients
    for (W, H, samp) in [(40, 56, [(1, 2), (1, 1), (1, 1)]), (33, 47, [(1, 2), (1, 1), (1, 1)]), (17, 70, [(1, 2), (1, 1), (1, 1)]),
                         (64, 64, [(2, 2), (1, 1), (1, 1)]), (50, 30, [(2, 1), (1, 1), (1, 1)]), (9, 9, [(2, 2), (1, 1), (1, 1)])]:
        out.append((f"crafted {W}x{H} {samp[0]}", craft_jpeg(W, H, samp, rng)))
    from jpeg_craft import craft_noninterleaved                  # one scan per component, restart intervals in blocks
    for (W, H, samp, dri) in [(40, 56, [(2, 2), (1, 1), (1, 1)], 0), (50, 30, [(1, 2), (1, 1), (1, 1)], 3), (70, 9, [(2, 2), (1, 1), (1, 1)], 5)]:
        out.append((f"crafted non-interleaved {W}x{H} {samp[0]} dri {dri}", craft_noninterleaved(W, H, samp, rng, dri=dri)))
    try:
        buf = io.BytesIO(); Image.fromarray(_smooth(rng, 64, 96)).save(buf, format="JPEG", quality=85, progressive=True, restart_marker_blocks=4)
        out.append(("progressive + restart", buf.getvalue()))
    except TypeError:
        pass
    return out


def _pillow(data, orient):
    from PIL import Image, ImageOps
    im = Image.open(io.BytesIO(data))
    if orient:
        im = ImageOps.exif_transpose(im)
    return np.asarray(im.convert("RGB"))


def test_oracle_against_the_golden_made_by_pillow(golden_dir):
    g = np.load(os.path.join(golden_dir, "jpeg_decode.npz"))
    k = 0
    while f"file{k}" in g.files:
        data = g[f"file{k}"].tobytes()
        assert np.array_equal(J.decode_rgb(data), g[f"rgb{k}"]), k
        assert np.array_equal(J.read_image_bgr(data), g[f"bgr{k}"]), k
        k += 1
    assert k >= 10


def test_oracle_against_the_installed_pillow():
    pytest.importorskip("PIL.Image")
    files = _files()
    assert len(files) >= 51 and sum(t.startswith("progressive") for t, _ in files) >= 6 and sum(t.startswith("crafted") for t, _ in files) == 9
    for tag, data in files:
        assert np.array_equal(J.decode_rgb(data), _pillow(data, False)), tag
        assert np.array_equal(J.decode_rgb(data, orient=True), _pillow(data, True)), tag
    from PIL import Image
    buf = io.BytesIO(); Image.fromarray(np.zeros((32, 32, 4), np.uint8), mode="CMYK").save(buf, format="JPEG")
    with pytest.raises(J.JpegUnsupported):
        J.decode_rgb(buf.getvalue())


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    so = str(tmp_path_factory.mktemp("jpeg") / "jpeg_host_test.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "native", "jpeg_host_test.cpp")], check=True)
    lib = C.CDLL(so)
    assert lib.jpeg_job_size() == C.sizeof(L.mi_jpeg_job)
    return lib


def _host_decode(emu, files, bgr, orient):
    """the product's host half + the emulated launches over a BATCH of files; returns the HWC outputs"""
    lib = L.lib()
    n = len(files)
    jobs = (L.mi_jpeg_job * n)()
    keep, outs = [], []
    for j, data in zip(jobs, files):
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        info = L.mi_jpeg_info()
        L.check(lib.mi_jpeg_parse(buf, len(data), C.byref(info)), "mi_jpeg_parse")
        coef = np.empty(info.coef_count, np.int16)
        L.check(lib.mi_jpeg_huffman(buf, len(data), C.byref(info), coef.ctypes.data_as(C.c_void_p)), "mi_jpeg_huffman")
        planes = np.zeros(info.coef_count, np.uint8)
        h, w = (info.width, info.height) if (orient and info.orientation >= 5) else (info.height, info.width)
        out = np.full((h, w, 3), 99, np.uint8)
        L.check(lib.mi_jpeg_job_fill(C.byref(info), coef.ctypes.data_as(C.c_void_p), planes.ctypes.data_as(C.c_void_p),
                                     out.ctypes.data_as(C.c_void_p), int(bgr), int(orient), C.byref(j)), "mi_jpeg_job_fill")
        keep.append((buf, coef, planes, info))
        outs.append(out)
    bi, bp = C.c_int32(0), C.c_int32(0)
    L.check(lib.mi_jpeg_jobs_layout(jobs, n, C.byref(bi), C.byref(bp)), "mi_jpeg_jobs_layout")
    emu.jpeg_emulate_launches(C.cast(jobs, C.c_void_p), n, bi.value, bp.value)
    return outs, keep


def test_host_half_and_emulated_launches_equal_pillow(emu):
    pytest.importorskip("PIL.Image")
    files = _files()
    datas = [d for _, d in files]
    outs, keep = _host_decode(emu, datas, bgr=False, orient=False)
    for (tag, data), out, (_, coef, _, info) in zip(files, outs, keep):
        oc = J.huffman(data, J.parse(data))                                   # the coefficient blocks, component by component
        assert np.array_equal(coef, np.concatenate([c.reshape(-1) for c in oc])), tag
        assert np.array_equal(out, _pillow(data, False)), tag
    outs, _ = _host_decode(emu, datas, bgr=True, orient=True)                 # detectron2 read_image(format="BGR")
    for (tag, data), out in zip(files, outs):
        assert np.array_equal(out, _pillow(data, True)[:, :, ::-1]), tag


def test_unsupported_files_are_refused_not_decoded():
    from PIL import Image
    lib = L.lib()
    info = L.mi_jpeg_info()
    assert lib.mi_jpeg_parse((C.c_uint8 * 4)(1, 2, 3, 4), 4, C.byref(info)) != 0
    buf = io.BytesIO(); Image.fromarray(np.zeros((32, 32, 4), np.uint8), mode="CMYK").save(buf, format="JPEG")
    data = buf.getvalue()
    assert lib.mi_jpeg_parse((C.c_uint8 * len(data)).from_buffer_copy(data), len(data), C.byref(info)) != 0
    ok = io.BytesIO(); Image.fromarray(_smooth(np.random.RandomState(0), 64, 96)).save(ok, format="JPEG", quality=90)
    data = ok.getvalue()[:-200]                                               # truncated entropy data: decodes (zeros fed), no crash
    assert lib.mi_jpeg_parse((C.c_uint8 * len(data)).from_buffer_copy(data), len(data), C.byref(info)) == 0
    coef = np.empty(info.coef_count, np.int16)
    assert lib.mi_jpeg_huffman((C.c_uint8 * len(data)).from_buffer_copy(data), len(data), C.byref(info), coef.ctypes.data_as(C.c_void_p)) == 0


@pytest.mark.parametrize("fmt,orient", [("BGR", True), ("RGB", False)])
def test_decoder_host_mirror_through_the_emulated_launches(emu, fmt, orient):
    """`GpuJpegDecoder` without its device allocations and launches: the threaded host half into one buffer, the job table of
    the batch (host addresses here), then the emulated launches - against Pillow"""
    pytest.importorskip("PIL.Image")
    from yolov7_d2_amd.data_pipeline import GpuJpegDecoder
    dec = GpuJpegDecoder(device="cpu", format=fmt, apply_orientation=orient, workers=4)
    files = _files()
    datas = [d for _, d in files]

    def alloc(count):
        a = np.empty(count, np.int16)
        return a, a.ctypes.data
    infos, offs, coef = dec._host_half(datas, alloc)
    planes = np.zeros(int(offs[-1]), np.uint8)
    outs = [np.full(dec.out_shape(i) + (3,), 99, np.uint8) for i in infos]
    jobs, bi, bp = dec._jobs(infos, offs, coef.ctypes.data, planes.ctypes.data, [o.ctypes.data for o in outs])
    emu.jpeg_emulate_launches(C.cast(jobs, C.c_void_p), len(datas), bi, bp)
    for (tag, data), out in zip(files, outs):
        ref = _pillow(data, orient)
        assert np.array_equal(out, ref[:, :, ::-1] if fmt == "BGR" else ref), tag
    with pytest.raises(L.MI355Error):
        dec.decode(datas[:1])
    # the body of decode() itself on host tensors: allocations, the H2D stand-in, the job table, the launches (emulated)
    import torch

    def alloc(count):
        t = torch.empty(count, dtype=torch.int16)
        return t, t.data_ptr()

    def launch(jobs, n, bi_, bp_):
        emu.jpeg_emulate_launches(C.cast(jobs, C.c_void_p), n, bi_, bp_)
        return None
    dec._check_device = lambda: None
    dec._alloc_host = alloc
    dec._launch = launch
    for (tag, data), o in zip(files, dec.decode(datas)):
        ref = _pillow(data, orient)
        assert np.array_equal(o.numpy(), ref[:, :, ::-1] if fmt == "BGR" else ref), tag


def test_host_half_survives_corrupt_files():
    """600 mutated files through the library's parser and Huffman decoder in a child process: decoded or refused, never a crash"""
    pytest.importorskip("PIL.Image")
    child = os.path.join(ROOT, "tests", "jpeg_fuzz_child.py")
    r = subprocess.run([sys.executable, child, "600"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])
    assert "fuzz done" in r.stdout


def test_decompression_bomb_is_refused_before_any_allocation():
    """a header that declares a huge frame (a 300-byte file can claim 65535 x 65535) must be refused the way
    PIL.Image.open refuses it (DecompressionBombError above 2 * MAX_IMAGE_PIXELS) - before the coefficient buffer sized by
    the header is allocated"""
    from PIL import Image
    from yolov7_d2_amd.data_pipeline import GpuJpegDecoder
    buf = io.BytesIO(); Image.fromarray(_smooth(np.random.RandomState(1), 64, 96)).save(buf, format="JPEG", quality=85)
    data = bytearray(buf.getvalue())
    k = data.index(b"\xff\xc0")                               # SOF0: marker, length(2), precision(1), height(2), width(2)
    data[k + 5:k + 9] = bytes([0xFF, 0xFF, 0xFF, 0xFF])       # 65535 x 65535
    dec = GpuJpegDecoder(device="cpu")
    called = []

    def alloc(count):
        called.append(count)
        a = np.empty(count, np.int16)
        return a, a.ctypes.data
    with pytest.raises(L.MI355Error, match="decompression-bomb"):
        dec._host_half([bytes(data)], alloc)
    assert not called
    small = GpuJpegDecoder(device="cpu", max_image_pixels=1000)       # configurable, like Image.MAX_IMAGE_PIXELS
    with pytest.raises(L.MI355Error, match="decompression-bomb"):
        small._host_half([buf.getvalue()], alloc)
    assert GpuJpegDecoder.MAX_IMAGE_PIXELS == 2 * Image.MAX_IMAGE_PIXELS
