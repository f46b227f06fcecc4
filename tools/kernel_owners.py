#!/usr/bin/env python
"""which aten ops / autograd nodes own the launches of a kernel family in one eager DETR / SparseInst step?
torch.profiler: for every device activity whose name contains PATTERN, the chain of CPU ops above it (op < parent < ...)
and the input shapes.  usage: kernel_owners.py [detr|sparseinst] PATTERN [PATTERN ...]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "detr"
pats = sys.argv[2:] or ["elementwise_kernel_manual_unroll<128, 8"]
sys.argv = [sys.argv[0], which]
exec(open(os.path.join(ROOT, "tools", "host_step_probe.py")).read().split("for _ in range(3):")[0].replace("gs = GraphedTrainStep(model, opt)", "gs = None"))
static = model.prepare_batch(inputs)


def step():
    losses = model.forward_prepared(static)
    wd = getattr(getattr(model, "criterion", None), "weight_dict", None) if which == "detr" else None
    total = losses["total"] if "total" in losses else sum(v for k, v in losses.items() if wd is None or k in wd)
    opt.zero_grad(set_to_none=True)
    total.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with torch.autograd.set_multithreading_enabled(False):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    ks = [k for k in (e.kernels or []) if any(p in k.name for p in pats)]
    if not ks:
        continue
    chain, p = [e.name], e.cpu_parent
    while p is not None and len(chain) < 5:
        chain.append(p.name)
        p = p.cpu_parent
    a = agg[(" < ".join(chain), str(e.input_shapes)[:70])]
    a[0] += len(ks)
    a[1] += sum(k.duration for k in ks)
print(f"{which}: launches of {pats} by owning op chain")
for (c, shp), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} {us:8.1f} us  {c[:150]:150s} {shp}")
