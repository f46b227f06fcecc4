"""TEST INFRASTRUCTURE: a minimal baseline-JPEG WRITER for the decoder tests - arbitrary sampling factors (e.g. 4:4:0, which no
encoder in this image produces) and RANDOM quantised coefficients, entropy coded with the standard tables borrowed from a file
Pillow wrote.  The files are valid JPEGs: Pillow decodes them, and its decode is the reference the tests compare with."""
import io
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import jpeg_oracle as J  # noqa: E402

def _std_tables():
    """the Huffman / quantisation tables of a file Pillow writes without optimisation (the standard Annex K tables)"""
    buf = io.BytesIO(); Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(buf, format="JPEG", quality=75)
    return J.parse(buf.getvalue())

def _codes(bits, vals):
    code, k, out = 0, 0, {}
    for l in range(1, 17):
        for _ in range(bits[l]):
            out[vals[k]] = (code, l); code += 1; k += 1
        code <<= 1
    return out

class _W:
    def __init__(self): self.acc = 0; self.n = 0; self.out = bytearray()
    def put(self, v, l):
        self.acc = (self.acc << l) | (v & ((1 << l) - 1)); self.n += l
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 255; self.out.append(b)
            if b == 255: self.out.append(0)
            self.n -= 8
    def flush(self):
        if self.n: self.put((1 << (8 - self.n)) - 1, 8 - self.n)

def _cat(v):
    a = abs(v); s = 0
    while a: a >>= 1; s += 1
    return s

def craft_jpeg(width, height, samp, rng, density=0.15):
    """a baseline JPEG with the given (h, v) factors per component (Y, Cb, Cr) and RANDOM quantised coefficients, entropy
    coded with the standard tables: a valid file no encoder in this image can produce (e.g. 4:4:0)"""
    t = _std_tables()
    hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
    mw, mh = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    dc = [_codes(*t["dc"][0]), _codes(*t["dc"][1])]; ac = [_codes(*t["ac"][0]), _codes(*t["ac"][1])]
    w = _W(); pred = [0, 0, 0]
    for my in range(mh):
        for mx in range(mw):
            for ci, (h, v) in enumerate(samp):
                tb = 0 if ci == 0 else 1
                for _ in range(h * v):
                    blk = np.zeros(64, np.int64)
                    blk[0] = rng.randint(-60, 61)
                    nz = rng.rand(63) < density
                    blk[1:][nz] = rng.randint(-12, 13, nz.sum())
                    diff = int(blk[0]) - pred[ci]; pred[ci] = int(blk[0])
                    s = _cat(diff); c, l = dc[tb][s]; w.put(c, l)
                    if s: w.put(diff if diff > 0 else diff + (1 << s) - 1, s)
                    run = 0
                    last = max([k for k in range(1, 64) if blk[k]] + [0])
                    for k in range(1, last + 1):
                        v_ = int(blk[k])
                        if v_ == 0: run += 1; continue
                        while run > 15:
                            c, l = ac[tb][0xF0]; w.put(c, l); run -= 16
                        s = _cat(v_); c, l = ac[tb][(run << 4) | s]; w.put(c, l)
                        w.put(v_ if v_ > 0 else v_ + (1 << s) - 1, s); run = 0
                    if last < 63:
                        c, l = ac[tb][0]; w.put(c, l)
    w.flush()
    def seg(m, payload): return bytes([0xFF, m]) + (len(payload) + 2).to_bytes(2, "big") + payload
    out = bytearray(b"\xff\xd8")
    zz = J.ZIGZAG
    for tq in (0, 1):
        q = t["qt"][tq]; out += seg(0xDB, bytes([tq]) + bytes(int(q[zz[i]]) for i in range(64)))
    out += seg(0xC0, bytes([8]) + height.to_bytes(2, "big") + width.to_bytes(2, "big") + bytes([3]) +
               b"".join(bytes([ci + 1, (h << 4) | v, 0 if ci == 0 else 1]) for ci, (h, v) in enumerate(samp)))
    for cls, tabs in ((0, t["dc"]), (1, t["ac"])):
        for th in (0, 1):
            bits, vals = tabs[th]; out += seg(0xC4, bytes([(cls << 4) | th]) + bytes(bits[1:17]) + bytes(vals))
    out += seg(0xDA, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0]))
    out += w.out + b"\xff\xd9"
    return bytes(out)


def craft_noninterleaved(width, height, samp, rng, density=0.15, dri=0):
    """the same, as a NON-interleaved sequential file: one scan per component over the component's own block grid (not padded to
    the MCU grid), optional restart interval (in blocks) - the multi-scan script baseline JPEG allows and encoders rarely write"""
    t = _std_tables()
    hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
    dc = [_codes(*t["dc"][0]), _codes(*t["dc"][1])]
    ac = [_codes(*t["ac"][0]), _codes(*t["ac"][1])]

    def seg(m, payload):
        return bytes([0xFF, m]) + (len(payload) + 2).to_bytes(2, "big") + payload
    out = bytearray(b"\xff\xd8")
    zz = J.ZIGZAG
    for tq in (0, 1):
        q = t["qt"][tq]
        out += seg(0xDB, bytes([tq]) + bytes(int(q[zz[i]]) for i in range(64)))
    out += seg(0xC0, bytes([8]) + height.to_bytes(2, "big") + width.to_bytes(2, "big") + bytes([3]) +
               b"".join(bytes([ci + 1, (h << 4) | v, 0 if ci == 0 else 1]) for ci, (h, v) in enumerate(samp)))
    for cls, tabs in ((0, t["dc"]), (1, t["ac"])):
        for th in (0, 1):
            bits, vals = tabs[th]
            out += seg(0xC4, bytes([(cls << 4) | th]) + bytes(bits[1:17]) + bytes(vals))
    if dri:
        out += seg(0xDD, dri.to_bytes(2, "big"))
    for ci, (h, v) in enumerate(samp):
        tb = 0 if ci == 0 else 1
        dw, dh = -(-width * h // hmax), -(-height * v // vmax)
        bw, bh = -(-dw // 8), -(-dh // 8)
        w = _W()
        pred = cnt = rst = 0
        for _ in range(bw * bh):
            if dri and cnt == dri:
                w.flush()
                w.out += bytes([0xFF, 0xD0 + (rst & 7)])
                rst += 1
                pred = cnt = 0
            cnt += 1
            blk = np.zeros(64, np.int64)
            blk[0] = rng.randint(-60, 61)
            nz = rng.rand(63) < density
            blk[1:][nz] = rng.randint(-12, 13, nz.sum())
            diff = int(blk[0]) - pred
            pred = int(blk[0])
            s = _cat(diff)
            c, l = dc[tb][s]
            w.put(c, l)
            if s:
                w.put(diff if diff > 0 else diff + (1 << s) - 1, s)
            run = 0
            last = max([k for k in range(1, 64) if blk[k]] + [0])
            for k in range(1, last + 1):
                v_ = int(blk[k])
                if v_ == 0:
                    run += 1
                    continue
                while run > 15:
                    c, l = ac[tb][0xF0]
                    w.put(c, l)
                    run -= 16
                s = _cat(v_)
                c, l = ac[tb][(run << 4) | s]
                w.put(c, l)
                w.put(v_ if v_ > 0 else v_ + (1 << s) - 1, s)
                run = 0
            if last < 63:
                c, l = ac[tb][0]
                w.put(c, l)
        w.flush()
        out += seg(0xDA, bytes([1, ci + 1, (tb << 4) | tb, 0, 63, 0])) + w.out
    out += b"\xff\xd9"
    return bytes(out)
