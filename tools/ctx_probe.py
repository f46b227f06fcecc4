#!/usr/bin/env python
"""GPU box: why does the SAME kernel on the SAME shape take 21 .. 57 us inside the step (w3_kernel, 40x40 K = 128; per
instance reproducible)?  The instruction-cache / L2 probe (tools/icache_probe.py) says neither code nor cache warmth.  This
probe replays single commands of the REAL step plan (real activations, real arena addresses) out of their context:
  A  every 40x40 / 80x80 bottleneck 3x3 conv alone, 8 times back to back            -> data / address effect if still slow
  B  one bottleneck (conv1, bn, conv2, bn) 4 times in a row                           -> local context
  C  the dark4 stage in forward order, then the three bottlenecks in REVERSE order  -> position in the sequence
Durations from rocprofv3 --kernel-trace (tools/ctx_probe.sh); a one-block polluter launch separates the experiments."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from bench import synth_batch_device

torch.manual_seed(0)
model = M.build_model(M.yolox_s_cfg(device="cuda")); model.train()
ps = model.plan_for(16, 640, 640, True)
imgs, labels = synth_batch_device(16, 640, 640, 1234, "cuda")
ps.image.copy_(imgs); ps.labels.copy_(labels); ps.gw().fill_(1.0)
plan = ps.plan
for _ in range(3):
    plan.run("fwd"); plan.run("bwd")
torch.cuda.synchronize()
arr, n = plan.fwd_cmds
tags = plan.fwd_tags
idx = {tags[k]: k for k in range(n)}
sp = L.stream_ptr()
CMD = C.sizeof(L.mi_cmd)
base = C.addressof(arr)


def run(k, cnt=1):
    L.check(L.lib().mi_cmdlist_run(C.cast(base + k * CMD, C.POINTER(L.mi_cmd)), cnt, sp), f"cmd {k}")


def mark():
    L.check(L.dbg().mi_debug_code_polluter(8, 1, sp), "marker")


order = []
def exp(name, fn):
    mark(); fn(); order.append(name)

kz = idx["bn_acc_zero.fwd"]          # (the statistics accumulators are re-zeroed before every repetition)
convs = [t for t in tags if t.endswith("conv2.conv") and ".m." in t]
for t in convs:                                               # A
    exp("A " + t, lambda t=t: [run(idx[t]) for _ in range(8)])
for blk in ("backbone.dark4.1.m.0", "backbone.dark4.1.m.2", "backbone.dark3.1.m.0"):     # B
    k0 = idx[blk + ".conv1.conv"]
    exp("B " + blk + " x4: " + " | ".join(tags[k0:k0 + 4]), lambda k0=k0: [(run(kz), run(k0, 4)) for _ in range(4)])
k_first = idx["backbone.dark4.0.conv"]; k_last = idx["backbone.dark4.1.conv3.bnact"]
exp("C dark4 forward order: " + " | ".join(tags[k_first:k_last + 1]), lambda: [(run(kz), run(k_first, k_last - k_first + 1)) for _ in range(2)])
def rev():
    for _ in range(2):
        run(kz)
        for m in (2, 1, 0):
            run(idx[f"backbone.dark4.1.m.{m}.conv1.conv"], 4)
exp("C dark4 bottlenecks in reverse order m2 m1 m0 (x2)", rev)
mark()
torch.cuda.synchronize()
with open(os.path.join(ROOT, "gpurun_out", "ctx_order.txt"), "w") as f:
    f.write("\n".join(order) + "\n")
print("experiments", len(order))
