"""child process of tests/test_gpu_augment.py::test_jpeg_decoder_equals_pillow: `GpuJpegDecoder.decode` (host Huffman in the
library, IDCT and colour kernels on the GPU) against Pillow's own decode of the same files.  Exit code 0 = bit-identical."""
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image, ImageOps  # noqa: E402
from yolov7_d2_amd.data_pipeline import GpuJpegDecoder  # noqa: E402

rng = np.random.RandomState(11)


def smooth(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 9.0 + yy / 17.0), 127 + 100 * np.cos(xx / 13.0), 127 + 100 * np.sin(yy / 7.0)], -1)
    return np.clip(base + rng.randint(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)


files = []
for (h, w, kw) in [(480, 640, dict(quality=90, subsampling=2)), (427, 640, dict(quality=75, subsampling=2)), (375, 500, dict(quality=95, subsampling=1)),
                   (333, 500, dict(quality=85, subsampling=0)), (17, 23, dict(quality=80, subsampling=2)), (50, 3, dict(quality=80, subsampling=2)),
                   (640, 480, dict(quality=85, subsampling=2, optimize=True)), (200, 300, dict(quality=85, subsampling=2, restart_marker_blocks=5))]:
    buf = io.BytesIO()
    Image.fromarray(smooth(h, w)).save(buf, format="JPEG", **kw)
    files.append(buf.getvalue())
buf = io.BytesIO(); Image.fromarray(smooth(120, 90)[..., 0]).save(buf, format="JPEG", quality=80); files.append(buf.getvalue())
buf = io.BytesIO(); Image.fromarray(smooth(300, 400)).save(buf, format="JPEG", quality=85, subsampling=2, progressive=True); files.append(buf.getvalue())
for o in (3, 6, 8, 5):
    ex = Image.Exif(); ex[0x0112] = o
    buf = io.BytesIO(); Image.fromarray(smooth(96, 140)).save(buf, format="JPEG", quality=90, exif=ex.tobytes()); files.append(buf.getvalue())
bad = []
for fmt, orient in (("BGR", True), ("RGB", False)):
    outs = GpuJpegDecoder(format=fmt, apply_orientation=orient, workers=4).decode(files)
    torch.cuda.synchronize()
    for k, (o, f) in enumerate(zip(outs, files)):
        im = Image.open(io.BytesIO(f))
        if orient:
            im = ImageOps.exif_transpose(im)
        ref = np.asarray(im.convert("RGB"))
        ref = ref[:, :, ::-1] if fmt == "BGR" else ref
        got = o.cpu().numpy()
        if got.shape != ref.shape or not np.array_equal(got, ref):
            bad.append((fmt, k, got.shape, ref.shape))
print("jpeg decoder on the GPU:", "bit-identical to Pillow (%d files x 2 formats)" % len(files) if not bad else bad[:8])
sys.exit(1 if bad else 0)
