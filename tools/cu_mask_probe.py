"""What a hipExtStreamCreateWithCUMask mask selects on this device: 1024 spinning blocks per launch on a masked stream,
each reports its XCC id and HW_ID (libmi355dbg.so: mi_debug_cu_census) -> distinct (XCC, SE/SH/CU) slots per XCC.
usage: python tools/cu_mask_probe.py"""
import ctypes as C
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_d2_amd import _lib as L  # noqa: E402
from yolov7_d2_amd.engine import NativeTrainer  # noqa: E402


def census(stream, blocks=1024, spin_us=300):
    out = torch.zeros(2 * blocks, dtype=torch.int32, device="cuda")
    L.check(L.dbg().mi_debug_cu_census(out.data_ptr(), blocks, spin_us, stream), "census")
    torch.cuda.synchronize()
    o = out.cpu().view(-1, 2)
    slots = Counter()
    for xcc, hw in o.tolist():
        slots[(xcc, (hw >> 8) & 0xff)] += 1
    per = Counter(x for (x, _) in slots)
    return len(slots), dict(sorted(per.items()))


def masked(words):
    arr = (C.c_uint32 * len(words))(*words)
    out = C.c_void_p()
    L.check(L.lib().mi_stream_create_cu_mask(arr, len(words), C.byref(out)), "create")
    return out.value


if __name__ == "__main__":
    torch.cuda.set_device(0)
    print("unmasked:", census(torch.cuda.Stream().cuda_stream))
    for n in (32, 64, 96, 128):
        for whole in (False, True):
            w = NativeTrainer.cu_mask_words(n, whole)
            s = masked(w)
            print(f"mask n={n} whole_xcds={whole} words={[hex(x) for x in w]}:", census(s))
            L.lib().mi_stream_destroy(s)
    w = NativeTrainer.cu_mask_words(64, False, invert=True)
    s = masked(w)
    print("complement of n=64:", census(s))
