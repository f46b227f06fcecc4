#!/usr/bin/env python
"""host half of the JPEG decoder: files/s of mi_jpeg_parse + mi_jpeg_huffman per thread (no GPU needed), beside Pillow's full
decode of the same file on one thread.  usage: jpeg_host_bench.py [threads]"""
import ctypes as C, io, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from yolov7_d2_amd import _lib as L

lib = L.lib()
rng = np.random.RandomState(0)
yy, xx = np.mgrid[0:480, 0:640]
img = np.clip(np.stack([127 + 100 * np.sin(xx / 9.0 + yy / 17.0), 127 + 100 * np.cos(xx / 13.0), 127 + 100 * np.sin(yy / 7.0)], -1)
              + rng.randint(-25, 26, (480, 640, 3)), 0, 255).astype(np.uint8)
for tag, kw in (("baseline q90 4:2:0", dict(quality=90, subsampling=2)), ("progressive q90 4:2:0", dict(quality=90, subsampling=2, progressive=True))):
    buf = io.BytesIO(); Image.fromarray(img).save(buf, format="JPEG", **kw)
    data = buf.getvalue()
    cb = (C.c_uint8 * len(data)).from_buffer_copy(data)

    info0 = L.mi_jpeg_info()
    lib.mi_jpeg_parse(cb, len(data), C.byref(info0))
    import threading
    tls = threading.local()

    def one(_):
        # (one coefficient buffer per thread, as GpuJpegDecoder decodes a batch into ONE pinned buffer: a fresh 0.9 MB
        #  allocation per file would measure the kernel's page-fault path instead)
        if not hasattr(tls, "coef"):
            tls.coef = np.empty(info0.coef_count, np.int16)
        info = L.mi_jpeg_info()
        lib.mi_jpeg_parse(cb, len(data), C.byref(info))
        lib.mi_jpeg_huffman(cb, len(data), C.byref(info), tls.coef.ctypes.data_as(C.c_void_p))
    for threads in ([1] + ([int(sys.argv[1])] if len(sys.argv) > 1 else [8])):
        n = 200 * threads
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(one, range(threads)))
            t0 = time.perf_counter(); list(ex.map(one, range(n))); dt = time.perf_counter() - t0
        print(f"{tag}: {len(data) / 1e3:.0f} kB file, {threads} thread(s): {n / dt:7.0f} files/s ({dt / n * 1e3 * threads:.2f} ms per file per thread)")
    for threads in ([1] + ([int(sys.argv[1])] if len(sys.argv) > 1 else [8])):     # the library's own thread pool, one call
        n = 400 * threads
        bufs = [np.empty(info0.coef_count, np.int16) for _ in range(threads * 2)]
        datas = (C.c_void_p * n)(*[C.addressof(cb)] * n)
        lens = (C.c_int64 * n)(*[len(data)] * n)
        iarr = (L.mi_jpeg_info * n)(*[info0] * n)
        coefs = (C.c_void_p * n)(*[bufs[k % len(bufs)].ctypes.data for k in range(n)])     # (buffers shared round-robin: a throughput test)
        rcs = (C.c_int32 * n)()
        t0 = time.perf_counter(); rc = lib.mi_jpeg_huffman_batch(datas, lens, iarr, coefs, n, threads, rcs); dt = time.perf_counter() - t0
        print(f"{tag}: mi_jpeg_huffman_batch, {threads} thread(s): {n / dt:7.0f} files/s (rc {rc})")
    t0 = time.perf_counter()
    for _ in range(100):
        np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    print(f"{tag}: Pillow full decode, 1 thread: {100 / (time.perf_counter() - t0):7.0f} files/s")
