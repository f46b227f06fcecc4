"""CPU restatement (TEST INFRASTRUCTURE: only tests/ may import it) of baseline JPEG decoding as the reference's mapper gets it:

    MyDatasetMapper2._load_image_with_annos -> detectron2 utils.read_image(file_name, format="BGR")
    (yolov7/data/dataset_mapper.py:646-648; d2 data/detection_utils.py read_image / convert_PIL_to_numpy, un-vendored)
    = PIL.Image.open(f) [-> EXIF orientation] -> .convert("RGB") -> np.asarray -> [:, :, ::-1]

Pillow decodes through libjpeg(-turbo) with its defaults: JDCT_ISLOW, fancy up-sampling, no colour quantisation.  This file
restates that pipeline from the library's published sources (jdhuff.c, jidctint.c, jdsample.c, jdcolor.c, jdmainct.c) for
sequential AND progressive Huffman JPEGs (SOF0 / SOF1 / SOF2, 8 bit, any scan script; 4:4:4, 4:2:2, 4:2:0, 4:4:0, grey;
restart intervals) and is PINNED against the Pillow installed in this image: tests/test_jpeg_decode.py decodes files
written by Pillow at several qualities / sub-samplings / sizes and compares bit for bit with Pillow's own decode, and the
golden `jpeg_decode.npz` (made by oracle/gen_golden.py::gold_jpeg) carries such files with Pillow's output.
Arithmetic-coded / lossless / 12-bit / CMYK files are refused (JpegUnsupported)."""
import numpy as np


class JpegUnsupported(ValueError):
    pass


ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                   54, 47, 55, 62, 63])       # zigzag index -> natural (row-major) index


def parse(data):
    """markers up to the scan: dict(width, height, comps=[(id, h, v, tq)], qt={tq: [64] natural order}, dc / ac Huffman
    tables {id: (bits[17], vals)}, restart interval, scan=[(component index, td, ta)], scan_start = offset of the entropy-coded
    data, orientation = EXIF tag 0x0112 or 1)"""
    b = memoryview(data)
    if b[0] != 0xFF or b[1] != 0xD8:
        raise JpegUnsupported("not a JPEG (no SOI)")
    p = 2
    info = dict(qt={}, dc={}, ac={}, dri=0, orientation=1, adobe_transform=None)
    while True:
        while b[p] != 0xFF:
            p += 1
        while b[p] == 0xFF:
            p += 1
        m = b[p]
        p += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        L = (b[p] << 8) | b[p + 1]
        seg = bytes(b[p + 2: p + L])
        if m == 0xDB:
            q = 0
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                if pq:
                    raise JpegUnsupported("16-bit quantisation table")
                t = np.zeros(64, np.int32)
                t[ZIGZAG] = np.frombuffer(seg[q + 1: q + 65], np.uint8)
                info["qt"][tq] = t
                q += 65
        elif m in (0xC0, 0xC1, 0xC2):
            info["progressive"] = m == 0xC2
            if seg[0] != 8:
                raise JpegUnsupported("sample precision %d" % seg[0])
            info["height"], info["width"] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            n = seg[5]
            info["comps"] = [(seg[6 + 3 * i], seg[7 + 3 * i] >> 4, seg[7 + 3 * i] & 15, seg[8 + 3 * i]) for i in range(n)]
        elif m in (0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegUnsupported("SOF marker 0x%02X (lossless / hierarchical / arithmetic)" % m)
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                bits = [0] + list(seg[q + 1: q + 17])
                nv = sum(bits)
                vals = list(seg[q + 17: q + 17 + nv])
                info["ac" if tc else "dc"][th] = (bits, vals)
                q += 17 + nv
        elif m == 0xDD:
            info["dri"] = (seg[0] << 8) | seg[1]
        elif m == 0xE1 and seg[:6] == b"Exif\x00\x00":
            info["orientation"] = _exif_orientation(seg[6:]) or 1
        elif m == 0xEE and seg[:5] == b"Adobe":
            info["adobe_transform"] = seg[11]
        elif m == 0xDA:
            info["sos_pos"] = p                       # (the length field of the first SOS: the scans are walked from here)
            return info
        elif m == 0xD9:
            raise JpegUnsupported("EOI before a scan")
        p += L


def _exif_orientation(t):
    if len(t) < 8:
        return None
    le = t[:2] == b"II"
    u16 = (lambda o: t[o] | (t[o + 1] << 8)) if le else (lambda o: (t[o] << 8) | t[o + 1])
    u32 = (lambda o: u16(o) | (u16(o + 2) << 16)) if le else (lambda o: (u16(o) << 16) | u16(o + 2))
    ifd = u32(4)
    if ifd + 2 > len(t):
        return None
    for k in range(u16(ifd)):
        e = ifd + 2 + 12 * k
        if e + 12 > len(t):
            break
        if u16(e) == 0x0112:
            return u16(e + 8)
    return None


def _huff_table(bits, vals):
    """jdhuff.c jpeg_make_d_derived_tbl: code -> (length, value) as maxcode / valptr / mincode arrays"""
    huffsize = []
    for l in range(1, 17):
        huffsize += [l] * bits[l]
    code, si, huffcode = 0, huffsize[0] if huffsize else 0, []
    k = 0
    while k < len(huffsize):
        while k < len(huffsize) and huffsize[k] == si:
            huffcode.append(code)
            code += 1
            k += 1
        code <<= 1
        si += 1
    maxcode, valptr, mincode = [-1] * 18, [0] * 17, [0] * 17
    k = 0
    for l in range(1, 17):
        if bits[l]:
            valptr[l] = k
            mincode[l] = huffcode[k]
            k += bits[l]
            maxcode[l] = huffcode[k - 1]
    maxcode[17] = 0xFFFFF
    return maxcode, valptr, mincode, vals


class _Bits:
    def __init__(self, data, pos):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0

    def _fill(self):
        d = self.d
        while self.n <= 24:
            c = d[self.p] if self.p < len(d) else 0
            if c == 0xFF:
                nx = d[self.p + 1] if self.p + 1 < len(d) else 0xD9
                if nx == 0:
                    self.p += 2
                else:                  # a marker: feed zeros (jdhuff.c does the same once it has hit one)
                    c = 0
                    self.acc = (self.acc << 8) | c
                    self.n += 8
                    continue
            else:
                self.p += 1
            self.acc = (self.acc << 8) | c
            self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        if self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def decode(self, tab):
        maxcode, valptr, mincode, vals = tab
        code, l = self.get(1), 1
        while code > maxcode[l]:
            code = (code << 1) | self.get(1)
            l += 1
        if l > 16:
            return 0
        return vals[valptr[l] + code - mincode[l]]

    def restart(self):
        """byte-align, skip the RSTn marker"""
        self.acc, self.n = 0, 0
        d = self.d
        while self.p < len(d) - 1 and not (d[self.p] == 0xFF and 0xD0 <= d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def _extend(v, s):
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1


def huffman(data, info):
    """every scan of the file -> per component int16 coefficient blocks [blocks_h][blocks_w][64] (natural order, NOT
    de-quantised), covering whole MCUs.  Sequential scans (jdhuff.c) and progressive ones (jdphuff.c: DC / AC, first /
    refinement passes, end-of-band runs); tables (DHT) and the restart interval (DRI) may change between scans."""
    data = bytes(data)
    comps = info["comps"]
    if len(comps) == 1:
        comps = info["comps"] = [(comps[0][0], 1, 1, comps[0][3])]            # a single component is never interleaved
    ids = [c[0] for c in comps]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    W, H = info["width"], info["height"]
    mw, mh = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    coef = [np.zeros((mh * c[2], mw * c[1], 64), np.int16) for c in comps]
    dc, ac, dri = dict(info["dc"]), dict(info["ac"]), info["dri"]
    p = info["sos_pos"]
    while True:
        L = (data[p] << 8) | data[p + 1]
        seg = data[p + 2: p + L]
        n = seg[0]
        scan = [(ids.index(seg[1 + 2 * i]), seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15) for i in range(n)]
        Ss, Se, Ah, Al = seg[1 + 2 * n], seg[2 + 2 * n], seg[3 + 2 * n] >> 4, seg[3 + 2 * n] & 15
        if not info.get("progressive"):
            Ss, Se, Ah, Al = 0, 63, 0, 0
        br = _Bits(data, p + L)
        dct = {k: _huff_table(*v) for k, v in dc.items()}
        act = {k: _huff_table(*v) for k, v in ac.items()}
        if n > 1:
            units = [(my, mx) for my in range(mh) for mx in range(mw)]
        else:
            ci = scan[0][0]
            bw, bh = -(-(-(-W * comps[ci][1] // hmax)) // 8), -(-(-(-H * comps[ci][2] // vmax)) // 8)
            units = [(by, bx) for by in range(bh) for bx in range(bw)]
        pred = [0] * len(comps)
        eobrun, todo = 0, dri
        for (uy, ux) in units:
            if dri:
                if todo == 0:
                    br.restart()
                    pred, eobrun, todo = [0] * len(comps), 0, dri
                todo -= 1
            blocks = []
            for (ci, td, ta) in scan:
                h, v = (comps[ci][1], comps[ci][2]) if n > 1 else (1, 1)
                for by in range(v):
                    for bx in range(h):
                        blocks.append((ci, td, ta, coef[ci][uy * v + by, ux * h + bx]))
            for (ci, td, ta, blk) in blocks:
                if Ss == 0:
                    if Ah == 0:
                        s_ = br.decode(dct[td])
                        pred[ci] += _extend(br.get(s_), s_) if s_ else 0
                        blk[0] = pred[ci] << Al
                    elif br.get(1):
                        blk[0] |= (1 << Al)
                    if Se == 0:
                        continue
                k = max(Ss, 1)
                if Ah == 0:                                           # sequential, or an AC first pass
                    if eobrun > 0:
                        eobrun -= 1
                        continue
                    while k <= Se:
                        rs = br.decode(act[ta])
                        r, s_ = rs >> 4, rs & 15
                        if s_:
                            k += r
                            blk[ZIGZAG[k]] = _extend(br.get(s_), s_) << Al
                            k += 1
                        elif r == 15:
                            k += 16
                        else:
                            if info.get("progressive"):
                                eobrun = (1 << r) + (br.get(r) if r else 0) - 1
                            break
                    continue
                p1, m1 = 1 << Al, -1 << Al                            # AC refinement
                if eobrun == 0:
                    while k <= Se:
                        rs = br.decode(act[ta])
                        r, s_ = rs >> 4, rs & 15
                        if s_:
                            s_ = p1 if br.get(1) else m1
                        elif r != 15:
                            eobrun = (1 << r) + (br.get(r) if r else 0)
                            break
                        while k <= Se:
                            z = ZIGZAG[k]
                            if blk[z] != 0:
                                if br.get(1) and (int(blk[z]) & p1) == 0:
                                    blk[z] += p1 if blk[z] >= 0 else m1
                            else:
                                r -= 1
                                if r < 0:
                                    break
                            k += 1
                        if s_:
                            blk[ZIGZAG[k]] = s_
                        k += 1
                if eobrun > 0:
                    while k <= Se:
                        z = ZIGZAG[k]
                        if blk[z] != 0 and br.get(1) and (int(blk[z]) & p1) == 0:
                            blk[z] += p1 if blk[z] >= 0 else m1
                        k += 1
                    eobrun -= 1
        # the next marker segment(s): tables may be redefined between scans
        p = br.p
        while True:
            while not (data[p] == 0xFF and data[p + 1] != 0x00 and not (0xD0 <= data[p + 1] <= 0xD7) and data[p + 1] != 0xFF):
                p += 1
            m = data[p + 1]
            p += 2
            if m == 0xD9:
                return coef
            L2 = (data[p] << 8) | data[p + 1]
            seg2 = data[p + 2: p + L2]
            if m == 0xDA:
                break
            if m == 0xC4:
                q = 0
                while q < len(seg2):
                    tc, th = seg2[q] >> 4, seg2[q] & 15
                    bits = [0] + list(seg2[q + 1: q + 17])
                    nv = sum(bits)
                    (ac if tc else dc)[th] = (bits, list(seg2[q + 17: q + 17 + nv]))
                    q += 17 + nv
            elif m == 0xDD:
                dri = (seg2[0] << 8) | seg2[1]
            p += L2


# ---- jidctint.c (JDCT_ISLOW), 8x8
CONST_BITS, PASS1_BITS = 13, 2
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865 = 2446, 3196, 4433, 6270
F_0_899976223, F_1_175875602, F_1_501321110, F_1_847759065 = 7373, 9633, 12299, 15137
F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(d, shift, pre_shift_dc):
    """one pass over the LAST axis of int64 [..., 8]"""
    z2, z3 = d[..., 2], d[..., 6]
    z1 = (z2 + z3) * F_0_541196100
    tmp2 = z1 + z3 * (-F_1_847759065)
    tmp3 = z1 + z2 * F_0_765366865
    z2, z3 = d[..., 0], d[..., 4]
    tmp0 = (z2 + z3) << CONST_BITS
    tmp1 = (z2 - z3) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = d[..., 7], d[..., 5], d[..., 3], d[..., 1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F_1_175875602
    t0 = t0 * F_0_298631336
    t1 = t1 * F_2_053119869
    t2 = t2 * F_3_072711026
    t3 = t3 * F_1_501321110
    z1 = z1 * (-F_0_899976223)
    z2 = z2 * (-F_2_562915447)
    z3 = z3 * (-F_1_961570560) + z5
    z4 = z4 * (-F_0_390180644) + z5
    t0 = t0 + z1 + z3
    t1 = t1 + z2 + z4
    t2 = t2 + z2 + z3
    t3 = t3 + z1 + z4
    out = np.stack([tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3], -1)
    return _descale(out, shift)


def idct_islow(coef, qt):
    """int16 [..., 64] natural order -> uint8 [..., 8, 8] samples (range-limited, +128)"""
    d = coef.astype(np.int64) * qt.astype(np.int64)
    d = d.reshape(coef.shape[:-1] + (8, 8))
    ws = _idct_1d(np.swapaxes(d, -1, -2), CONST_BITS - PASS1_BITS, True)        # pass 1: columns
    ws = np.swapaxes(ws, -1, -2)
    out = _idct_1d(ws, CONST_BITS + PASS1_BITS + 3, False)                      # pass 2: rows
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def _planes(coef, info):
    """IDCT of every block -> per component uint8 plane [blocks_h * 8, blocks_w * 8]"""
    out = []
    for c, (cid, h, v, tq) in zip(coef, info["comps"]):
        s = idct_islow(c, info["qt"][tq])
        bh, bw = s.shape[:2]
        out.append(s.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8))
    return out


# ---- jdsample.c
def _h2_fancy_rows(p, width):
    """h2v1_fancy_upsample of every row of p[:, :width] (int) -> [rows, 2 * width]"""
    p = p[:, :width].astype(np.int64)
    out = np.empty((p.shape[0], 2 * width), np.int64)
    out[:, 0] = p[:, 0]
    out[:, 1] = (p[:, 0] * 3 + p[:, 1] + 2) >> 2
    out[:, 2:-2:2] = (p[:, 1:-1] * 3 + p[:, :-2] + 1) >> 2
    out[:, 3:-2:2] = (p[:, 1:-1] * 3 + p[:, 2:] + 2) >> 2
    out[:, -2] = (p[:, -1] * 3 + p[:, -2] + 1) >> 2
    out[:, -1] = p[:, -1]
    return out


def _upsample(plane, h, v, hmax, vmax, width, height, fancy=True):
    """one component's plane -> full resolution [>= height, >= width] (uint8)"""
    hx, vx = hmax // h, vmax // v
    dw, dh = -(-width * h // hmax), -(-height * v // vmax)          # downsampled_width / _height
    p = plane[:dh, :dw].astype(np.int64)
    if hx == 1 and vx == 1:
        return p.astype(np.uint8)
    if hx == 2 and vx == 1:
        if fancy and dw > 2:
            return _h2_fancy_rows(p, dw).astype(np.uint8)
        return np.repeat(p, 2, 1).astype(np.uint8)
    if hx == 2 and vx == 2:
        if fancy and dw > 2:
            up = np.concatenate([p[:1], p[:-1]], 0)                 # the row above (replicated at the top)
            dn = np.concatenate([p[1:], p[-1:]], 0)                 # the row below (replicated at the bottom)
            rows = np.empty((2 * dh, dw), np.int64)
            rows[0::2] = p * 3 + up
            rows[1::2] = p * 3 + dn
            out = np.empty((2 * dh, 2 * dw), np.int64)
            out[:, 0] = (rows[:, 0] * 4 + 8) >> 4
            out[:, 1] = (rows[:, 0] * 3 + rows[:, 1] + 7) >> 4
            out[:, 2:-2:2] = (rows[:, 1:-1] * 3 + rows[:, :-2] + 8) >> 4
            out[:, 3:-2:2] = (rows[:, 1:-1] * 3 + rows[:, 2:] + 7) >> 4
            out[:, -2] = (rows[:, -1] * 3 + rows[:, -2] + 8) >> 4
            out[:, -1] = (rows[:, -1] * 4 + 7) >> 4
            return out.astype(np.uint8)
        return np.repeat(np.repeat(p, 2, 0), 2, 1).astype(np.uint8)
    if hx == 1 and vx == 2:
        if fancy:                                                   # h1v2_fancy_upsample (libjpeg-turbo >= 1.5)
            up = np.concatenate([p[:1], p[:-1]], 0)
            dn = np.concatenate([p[1:], p[-1:]], 0)
            out = np.empty((2 * dh, dw), np.int64)
            out[0::2] = (p * 3 + up + 1) >> 2
            out[1::2] = (p * 3 + dn + 2) >> 2
            return out.astype(np.uint8)
        return np.repeat(p, 2, 0).astype(np.uint8)
    raise JpegUnsupported("sampling factors %dx%d of %dx%d" % (h, v, hmax, vmax))


# ---- jdcolor.c
def _ycc_tables():
    x = np.arange(256, dtype=np.int64) - 128
    fix = lambda f: int(f * 65536 + 0.5)
    return ((fix(1.40200) * x + 32768) >> 16, (fix(1.77200) * x + 32768) >> 16, -fix(0.71414) * x, -fix(0.34414) * x + 32768)


def ycc_to_rgb(y, cb, cr):
    cr_r, cb_b, cr_g, cb_g = _ycc_tables()
    y = y.astype(np.int64)
    r = np.clip(y + cr_r[cr], 0, 255)
    g = np.clip(y + ((cb_g[cb] + cr_g[cr]) >> 16), 0, 255)
    b = np.clip(y + cb_b[cb], 0, 255)
    return np.stack([r, g, b], -1).astype(np.uint8)


def apply_orientation(img, o):
    """PIL.ImageOps.exif_transpose / detectron2 _apply_exif_orientation"""
    if o == 2:
        return img[:, ::-1]
    if o == 3:
        return img[::-1, ::-1]
    if o == 4:
        return img[::-1]
    if o == 5:
        return img.transpose(1, 0, 2)
    if o == 6:
        return img.transpose(1, 0, 2)[:, ::-1]
    if o == 7:
        return img.transpose(1, 0, 2)[::-1, ::-1]
    if o == 8:
        return img.transpose(1, 0, 2)[::-1]
    return img


def decode_rgb(data, orient=False):
    """Image.open(...).convert("RGB") as an HWC uint8 array (orient: apply the EXIF orientation like d2's read_image)"""
    info = parse(data)
    coef = huffman(data, info)
    planes = _planes(coef, info)
    W, H = info["width"], info["height"]
    comps = info["comps"]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    full = [_upsample(p, c[1], c[2], hmax, vmax, W, H)[:H, :W] for p, c in zip(planes, comps)]
    if len(comps) == 1:
        img = np.repeat(full[0][..., None], 3, -1)
    elif len(comps) == 3:
        if info["adobe_transform"] == 0:
            img = np.stack(full, -1)
        else:
            img = ycc_to_rgb(*full)
    else:
        raise JpegUnsupported("%d components" % len(comps))
    img = np.ascontiguousarray(img)
    return np.ascontiguousarray(apply_orientation(img, info["orientation"])) if orient else img


def read_image_bgr(data):
    """detectron2 utils.read_image(file, format="BGR") on the file's bytes"""
    return np.ascontiguousarray(decode_rgb(data, orient=True)[:, :, ::-1])
