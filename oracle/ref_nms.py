"""TEST INFRASTRUCTURE (oracle/): ctypes view of oracle/_ref/libref_nms.so - the reference's own greedy NMS
(deploy/trt_cc/demo_yolox.cc:53-135) compiled from /root/reference by oracle/Makefile over a stand-in for cv::Rect_
(oracle/ref_nms_wrap.cc).  Only tests/ may import this; the product never does."""
import ctypes as C
import os

import numpy as np

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_nms.so")


def available():
    return os.path.exists(LIB)


_lib = None


def nms_xyxy(boxes, scores, thr):
    """boxes [n, 4] (x1, y1, x2, y2) float32, scores [n] -> kept input indices in the reference's pick order"""
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.ref_nms_sorted_bboxes.restype = C.c_int
        _lib.ref_nms_sorted_bboxes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    b = np.ascontiguousarray(np.asarray(boxes, dtype=np.float32))
    s = np.ascontiguousarray(np.asarray(scores, dtype=np.float32))
    n = b.shape[0]
    xywh = np.ascontiguousarray(np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32)) if n else b
    kept = np.zeros(max(n, 1), dtype=np.int32)
    k = _lib.ref_nms_sorted_bboxes(xywh.ctypes.data, s.ctypes.data, n, float(thr), kept.ctypes.data)
    return kept[:k].astype(np.int64)


def batched_nms_xyxy(boxes, scores, idxs, thr):
    """class-aware NMS the way the reference's eval path applies it (utils/boxes.py:199: boxes of different classes never
    suppress each other): the native greedy NMS class by class, kept indices merged in descending score order"""
    boxes, scores, idxs = np.asarray(boxes, np.float32), np.asarray(scores, np.float32), np.asarray(idxs)
    keep = []
    for c in np.unique(idxs):
        ci = np.nonzero(idxs == c)[0]
        keep.append(ci[nms_xyxy(boxes[ci], scores[ci], thr)])
    keep = np.concatenate(keep) if keep else np.zeros(0, np.int64)
    return keep[np.argsort(-scores[keep], kind="stable")]
