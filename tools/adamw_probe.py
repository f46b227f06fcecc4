import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_detr_graph import _model, _batch, _opt
a, b = _model(0.0), _model(0.0)
oa, ob = _opt(a), _opt(b, kind="multi")
batch = _batch(1, ((256, 320), (224, 288)), (3, 2))
for it in range(2):
    for m, o in ((a, oa), (b, ob)):
        losses = m(batch)
        total = sum(v for k, v in losses.items() if k in m.criterion.weight_dict)
        o.zero_grad(set_to_none=True)
        total.backward()
    gd = [(n, float((p.grad - q.grad).abs().max())) for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()) if p.requires_grad]
    print("step", it, "max grad diff", max(gd, key=lambda t: t[1]))
    before = {n: p.detach().clone() for n, p in a.named_parameters() if p.requires_grad}
    oa.step(); ob.step()
    torch.cuda.synchronize()
    rows = []
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if p.requires_grad:
            d = float((p.detach() - q.detach()).abs().max())
            mv = float((p.detach() - before[n]).abs().max())
            rows.append((d, mv, n, tuple(p.shape)))
    rows.sort(reverse=True)
    print("step", it, "worst param diffs (diff, torch's own move):", rows[:6])
