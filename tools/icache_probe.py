#!/usr/bin/env python
"""GPU box: is a kernel's first run after OTHER code slow because its instructions are cold?

In the step's in-graph trace the same kernel on the same shape takes 21 .. 57 us (w3_kernel, 40x40 K=128), reproducibly per
instance: the first of a series is slow, the third fast.  This probe runs ONE convolution in captured sequences that differ
only in what runs between two launches of it:
  warm      conv x 12 on the same buffers
  fresh     conv on 3 different (x, w, y, stats) sets in turn            -> data / TLB warmth
  code<KB>  conv, then a kernel that is KB KiB of straight-line no-ops    -> instruction-cache warmth only (no data traffic)
  flush     conv, then a 512 MiB copy                                      -> L2 / MALL warmth of code + data (small copy kernel)
  both      conv, copy, code64
Per-dispatch durations come from rocprofv3 --kernel-trace of this process (tools/icache_probe.sh); the script prints the
launch order so the trace can be cut.  usage: icache_probe.py [w3|c1s|bn]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from yolov7_d2_amd import _lib as L

DEV = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "w3"
sp = L.stream_ptr
N, H, W, K = 16, 40, 40, 128
g = torch.Generator().manual_seed(0)


def conv_set(taps):
    w = (torch.randn(K, K, 3 if taps == 9 else 1, 3 if taps == 9 else 1, generator=g) / (taps * K) ** 0.5).to(DEV)
    img = torch.empty(taps * K * K, dtype=torch.bfloat16, device=DEV)
    kk = 3 if taps == 9 else 1
    L.check(L.lib().mi_pack_conv_weight(w.data_ptr(), K, K, kk, kk, img.data_ptr(), K, K, None, 0, 0, sp()), "pack")
    x = torch.randn(N, H, W, K, generator=g).to(DEV, torch.bfloat16)
    y = torch.empty(N, H, W, K, dtype=torch.bfloat16, device=DEV)
    st = torch.zeros(16, K, 2, dtype=torch.float64, device=DEV)
    d = L.mi_conv_desc()
    d.x, d.w, d.y = x.data_ptr(), img.data_ptr(), y.data_ptr()
    d.ldx = d.ldy = K
    d.N, d.H, d.W, d.outH, d.outW, d.gridH, d.gridW = N, H, W, H, W, H, W
    d.in_stride = d.out_stride = 1
    d.K8, d.Cout, d.CoutPad, d.ntaps = K // 8, K, K, taps
    t = 0
    for r in range(kk):
        for s in range(kk):
            d.tap_dy[t], d.tap_dx[t], d.tap_w[t] = r - kk // 2, s - kk // 2, r * kk + s
            t += 1
    d.stats_acc, d.stats_slots = st.data_ptr(), 16
    return dict(d=d, keep=(w, img, x, y, st))


sets = [conv_set(9 if which == "w3" else 1) for _ in range(3)]
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
big_b = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)


def conv(i):
    L.check(L.lib().mi_conv2d(C.byref(sets[i]["d"]), sp()), "conv2d")


def code(kb):
    L.check(L.dbg().mi_debug_code_polluter(kb, 512, sp()), "polluter")


def flush():
    big_b.copy_(big_a)


SEQS = {
    "warm": lambda: [conv(0) for _ in range(12)],
    "fresh": lambda: [conv(i % 3) for i in range(12)],
    "code8": lambda: [(conv(0), code(8)) for _ in range(8)],
    "code16": lambda: [(conv(0), code(16)) for _ in range(8)],
    "code32": lambda: [(conv(0), code(32)) for _ in range(8)],
    "code64": lambda: [(conv(0), code(64)) for _ in range(8)],
    "code128": lambda: [(conv(0), code(128)) for _ in range(8)],
    "flush": lambda: [(conv(0), flush()) for _ in range(8)],
    "both": lambda: [(conv(0), flush(), code(64)) for _ in range(8)],
}
for name, fn in SEQS.items():
    fn(); torch.cuda.synchronize()            # eager warm-up (lazy init)
s = torch.cuda.Stream()
graphs = {}
with torch.cuda.stream(s):
    for name, fn in SEQS.items():
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        graphs[name] = gr
torch.cuda.synchronize()
for name, gr in graphs.items():
    for _ in range(4):
        L.check(L.dbg().mi_debug_code_polluter(8, 1, sp()), "marker")     # ONE-block launch = the marker between replays
        gr.replay()
    torch.cuda.synchronize()
print("ORDER", " ".join(SEQS))
