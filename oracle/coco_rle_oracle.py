"""ORACLE (test infrastructure, not product): numpy restatement of COCO's mask run-length encoding — cocoapi
common/maskApi.c rleEncode / rleToString / rleFrString / rleDecode, the code behind pycocotools.mask.encode that
yolov7/evaluation/coco_evaluation.py:38-50 calls.  pycocotools is an un-vendored dependency of the reference (absent from
/root/reference and from this image): *parity unpinned* - the format is restated from the published algorithm."""
import numpy as np


def rle_counts(mask):
    """mask [H, W] (non-zero = foreground) -> run lengths of the column-major scan, starting with a run of zeros"""
    m = (np.asarray(mask) != 0).astype(np.uint8).flatten(order="F")
    cnts, c, p = [], 0, 0
    for v in m:
        if v != p:
            cnts.append(c)
            c, p = 0, v
        c += 1
    cnts.append(c)
    return np.array(cnts, dtype=np.uint32)


def rle_counts_fast(mask):
    """the same by vector operations (for the large cases)"""
    m = (np.asarray(mask) != 0).astype(np.int8).flatten(order="F")
    d = np.flatnonzero(np.diff(np.concatenate([[0], m])) != 0)
    starts = np.concatenate([[0], d])
    return np.diff(np.concatenate([starts, [m.size]])).astype(np.uint32)


def rle_to_string(cnts):
    out = []
    for i, x in enumerate(int(v) for v in cnts):
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return "".join(out)


def rle_from_string(s):
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return np.array(cnts, dtype=np.int64)


def rle_decode(cnts, H, W):
    m = np.zeros(H * W, dtype=np.uint8)
    p, v = 0, 0
    for c in cnts:
        m[p:p + int(c)] = v
        p += int(c)
        v = 1 - v
    return m.reshape((H, W), order="F")
