#!/usr/bin/env python
"""in-graph marginal cost of each command class: replay the fwd/bwd graphs with one op class NOP'ed out
(results are garbage; only the timing matters).  Runs on the GPU box."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from bench import synth_batch_device

B, S = 16, 640
torch.manual_seed(0)
model = M.build_model(M.yolox_s_cfg(device="cuda"))
model.train()
ps = model.plan_for(B, S, S, True)
imgs, labels = synth_batch_device(B, S, S, 1234, "cuda")
ps.image.copy_(imgs); ps.labels.copy_(labels); ps.gw().fill_(1.0)
plan = ps.plan
lib = L.lib()
stream = torch.cuda.Stream()
sp = L.stream_ptr(stream)

def timed(skip=()):
    hs = []
    saved = []
    for which in ("fwd", "bwd"):
        arr, n = plan.fwd_cmds if which == "fwd" else plan.bwd_cmds
        for k in range(n):
            if L.OPS[arr[k].op] in skip:
                saved.append((arr, k, arr[k].op)); arr[k].op = 0
    with torch.cuda.stream(stream):
        for which in ("fwd", "bwd"):
            arr, n = plan.fwd_cmds if which == "fwd" else plan.bwd_cmds
            hs.append(L.check(lib.mi_graph_capture(arr, n, sp), "cap"))
        for _ in range(3):
            for h in hs: lib.mi_graph_launch(h, sp)
        stream.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            for h in hs: lib.mi_graph_launch(h, sp)
        stream.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
    for h in hs: lib.mi_graph_destroy(h)
    for arr, k, op in saved: arr[k].op = op
    return dt

plan.run("fwd"); plan.run("bwd"); torch.cuda.synchronize()
base = timed()
print(f"baseline fwd+bwd graph: {base:.3f} ms")
classes = [("CONV",), ("WGRAD",), ("BN_ACT_FWD",), ("BN_BWD_REDUCE",), 
           ("BN_BWD_APPLY",), ("PACK_W",), ("COLSUM",), ("SPLIT_DPREDS",), ("LOSS_FWD", "LOSS_BWD"), ("SPP_FWD", "SPP_BWD"),
           ("WGRAD_GROUP",), ("MEMSET",)]
for cl in classes:
    t = timed(cl)
    print(f"without {'+'.join(cl):40s}: {t:.3f} ms  (marginal {base - t:.3f})")
