"""GPU box: DETR-R50 at the bench configuration (800 x 1333, bs 4, dropout 0): the captured step against the eager step over
N optimizer steps - losses and parameters must stay identical (the small-size test holds that bit for bit; this checks the
REAL size, where torch's own reductions take their multi-block form whose hipMemsetAsync nodes misbehaved in the SparseInst
graph)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd.d2shim import Boxes, Instances
from yolov7_d2_amd.graph_step import GraphedTrainStep
from yolov7_d2_amd.optim import MultiTensorAdamW
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = M.detr_r50_cfg(device="cuda:0"); cfg.MODEL.DETR.DROPOUT = 0.0
eager = M.build_model(cfg); eager.train()
graphed = copy.deepcopy(eager); graphed.train()
B, H_, W_ = 4, 800, 1333
g = torch.Generator().manual_seed(1234)
inputs = []
for b in range(B):
    h, w = (H_, W_) if b == 0 else (H_ - 32 * (b % 3), W_ - 64 * (b % 4))
    n = int(torch.randint(1, 21, (1,), generator=g))
    wh = 16 + torch.rand(n, 2, generator=g) * 256
    xy = torch.rand(n, 2, generator=g) * (torch.tensor([w, h]) - wh).clamp(min=1)
    inst = Instances((h, w), gt_boxes=Boxes(torch.cat([xy, xy + wh], 1)), gt_classes=torch.randint(0, 80, (n,), generator=g))
    inputs.append(dict(image=torch.randint(0, 256, (3, h, w), generator=g).float().to(dev), instances=inst))
eager.shape_bucket = graphed.shape_bucket        # (eager forward pads exactly; the prepared path pads to the bucket)
mk = lambda m: MultiTensorAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4, clip_norm=0.1)
oe, og = mk(eager), mk(graphed)
step = GraphedTrainStep(graphed, og)
st = None
for it in range(int(os.environ.get("N", 6))):
    # eager reference through the SAME prepared path (same padded shape), uncaptured
    st = eager.prepare_batch(inputs, static=st)
    losses = eager.forward_prepared(st)
    te = sum(v for k, v in losses.items() if k in eager.criterion.weight_dict)
    oe.zero_grad(set_to_none=True); te.backward(); oe.step()
    out = step(inputs)
    torch.cuda.synchronize()
    dmax = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(eager.parameters(), graphed.parameters()))
    worst = max(abs(float(out[k]) - float(v)) / (abs(float(v)) + 1e-6) for k, v in losses.items())
    print(it, "total eager %.6f graphed %.6f" % (float(te), float(out["total"])), "worst loss rel diff %.3e" % worst, "max param diff %.3e" % dmax,
          "clip", [round(x, 5) for x in oe.clip_out.tolist()], [round(x, 5) for x in og.clip_out.tolist()], flush=True)
