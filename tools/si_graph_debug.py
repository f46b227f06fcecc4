"""GPU box: SparseInst bench configuration, graphed vs eager loss trajectories (debug of a divergence seen in bench.py)"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd.d2shim import Instances
from yolov7_d2_amd.graph_step import GraphedTrainStep
from yolov7_d2_amd.optim import MultiTensorAdamW
B, S = int(os.environ.get("B", 8)), int(os.environ.get("S", 640))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = M.sparse_inst_r50_giam_cfg(device="cuda:0")
eager = M.build_model(cfg); eager.train()
graphed = copy.deepcopy(eager); graphed.train()
g = torch.Generator().manual_seed(5)
inputs = []
yy, xx = torch.meshgrid(torch.arange(S).float(), torch.arange(S).float(), indexing="ij")
for b in range(B):
    n = int(torch.randint(1, 11, (1,), generator=g))
    masks = torch.zeros(n, S, S)
    for k in range(n):
        cy, cx = float(torch.rand(1, generator=g)) * S, float(torch.rand(1, generator=g)) * S
        ry, rx = 16 + float(torch.rand(1, generator=g)) * S * 0.25, 16 + float(torch.rand(1, generator=g)) * S * 0.25
        masks[k] = (((yy - cy).abs() < ry) & ((xx - cx).abs() < rx)) if k % 2 == 0 else ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) < 1)
    inst = Instances((S, S), gt_classes=torch.randint(0, 80, (n,), generator=g).to(dev), gt_masks=masks.to(dev))
    inputs.append(dict(image=torch.randint(0, 256, (3, S, S), generator=g).float().to(dev), instances=inst, height=S, width=S))
mk = lambda m: MultiTensorAdamW([p for p in m.parameters() if p.requires_grad], lr=5e-5, weight_decay=0.05)
oe, og = mk(eager), mk(graphed)
step = GraphedTrainStep(graphed, og)
for it in range(int(os.environ.get("N", 12))):
    losses = eager(inputs)
    te = sum(losses.values())
    oe.zero_grad(set_to_none=True); te.backward(); oe.step()
    out = step(inputs)
    torch.cuda.synchronize()
    dmax = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(eager.parameters(), graphed.parameters()))
    print(it, "eager", {k: round(float(v), 4) for k, v in losses.items()}, "| graphed", {k: round(float(out[k]), 4) for k in losses},
          "| max param diff %.3e" % dmax, "step_count", int(oe.step_count), int(og.step_count), flush=True)
