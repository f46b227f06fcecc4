#!/usr/bin/env python
"""every call of one DETR / SparseInst step (prepare_batch + forward_prepared + backward + optimizer) that BLOCKS the host on
the device (torch.cuda.set_sync_debug_mode("warn")): a blocking call in prepare_batch serialises the host half of step N + 1
behind the graph of step N.  usage: sync_sites.py [detr|sparseinst]"""
import os, sys, warnings, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "sparseinst"
sys.argv = [sys.argv[0], which]
exec(open(os.path.join(ROOT, "tools", "host_step_probe.py")).read().split("# instrument")[0].replace("gs = GraphedTrainStep(model, opt)", "gs = None").split("for _ in range(3):")[0])
static = model.prepare_batch(inputs)


def step():
    model.prepare_batch(inputs, static=static)
    losses = model.forward_prepared(static)
    wd = getattr(getattr(model, "criterion", None), "weight_dict", None) if which == "detr" else None
    total = losses["total"] if "total" in losses else sum(v for k, v in losses.items() if wd is None or k in wd)
    opt.zero_grad(set_to_none=True)
    total.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
sites = collections.Counter()
orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/yolov7_d2_amd/" in f.filename]
    sites[" < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-4:])) + "   [" + str(message)[:60] + "]"] += 1


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
with torch.autograd.set_multithreading_enabled(False):
    step()
torch.cuda.set_sync_debug_mode("default")
print(f"{which}: host-blocking calls in one step")
for s, c in sites.most_common():
    print(f"{c:4d}  {s}")
