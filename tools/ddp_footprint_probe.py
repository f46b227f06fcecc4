#!/usr/bin/env python
"""What does a resident collective kernel cost the YOLOX step, and does the one-launch BatchNorm backward (grid barrier:
every block must be resident) survive it?  RCCL cannot run on this pool's 1-GPU boxes, so its FOOTPRINT is emulated:
tools/micro/occupy.hip keeps G blocks of 256 threads resident on a second stream while the captured forward + backward
graphs replay - sleeping (CU slots only) or streaming a 256 MB buffer (the HBM share of a ring all-reduce).  Run once per
BatchNorm-backward form (MI_BN_FUSED=1 / 0); prints ms per step for every G.  Runs on the GPU box."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from bench import synth_batch_device

occ = C.CDLL(os.path.join(ROOT, "tools", "micro", "libocc.so"))
occ.occupy.argtypes = [C.c_int, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
B, S = 16, 640
if os.environ.get("FUSED_CAP"):      # blocks the grid-barrier kernel may use (default: 2 per CU); before the plan is built
    print("fused capacity ->", L.lib().mi_bn_fused_set_capacity(int(os.environ["FUSED_CAP"])))
torch.manual_seed(0)
model = M.build_model(M.yolox_s_cfg(device="cuda"))
model.train()
ps = model.plan_for(B, S, S, True)
imgs, labels = synth_batch_device(B, S, S, 1234, "cuda")
ps.image.copy_(imgs); ps.labels.copy_(labels); ps.gw().fill_(1.0)
plan = ps.plan
lib = L.lib()
stream, side = torch.cuda.Stream(), torch.cuda.Stream()
sp, sidep = L.stream_ptr(stream), L.stream_ptr(side)
buf = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
plan.run("fwd"); plan.run("bwd"); torch.cuda.synchronize()
hs = []
with torch.cuda.stream(stream):
    for which in ("fwd", "bwd"):
        arr, n = plan.fwd_cmds if which == "fwd" else plan.bwd_cmds
        hs.append(L.check(lib.mi_graph_capture(arr, n, sp), "cap"))


def steps(K):
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        for _ in range(K):
            for h in hs:
                lib.mi_graph_launch(h, sp)
        stream.synchronize()
        return (time.perf_counter() - t0) / K * 1e3


steps(3)
base = steps(20)
print(f"MI_BN_FUSED={os.environ.get('MI_BN_FUSED', 'auto')} bn_fused={plan.bn_fused}: no collective footprint {base:.3f} ms/step", flush=True)
for mem in ((0,) if os.environ.get("QUICK") else (0, 1)):
    for G in ((64,) if os.environ.get("QUICK") else (16, 32, 64, 128, 256)):
        torch.cuda.synchronize()
        # a short trial first: a barrier kernel that cannot become resident waits ~1 s per launch before it gives up
        occ.occupy(G, int(3 * base * 1.5 * 1e3) + 2000, buf.data_ptr(), buf.numel(), mem, sidep)
        t3 = steps(3)
        torch.cuda.synchronize()
        if t3 > 10 * base:
            print(f"  footprint {G:4d} blocks {'streaming' if mem else 'sleeping '}: {t3:.3f} ms/step over 3 steps - STALLED, sweep stopped", flush=True)
            break
        occ.occupy(G, int(20 * t3 * 1.3 * 1e3) + 2000, buf.data_ptr(), buf.numel(), mem, sidep)
        t = steps(20)
        torch.cuda.synchronize()
        print(f"  footprint {G:4d} blocks {'streaming' if mem else 'sleeping '}: {t:.3f} ms/step ({(t / base - 1) * 100:+.1f} %)", flush=True)
