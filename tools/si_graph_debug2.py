"""GPU box: bisect the graphed-SparseInst divergence: which captured piece goes wrong on the second replay"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import yolov7_d2_amd as M
from test_gpu_sparseinst import _si_inputs
from yolov7_d2_amd.optim import MultiTensorAdamW
B, S = int(os.environ.get("B", 8)), int(os.environ.get("S", 640))
torch.manual_seed(0)
cfg = M.sparse_inst_r50_giam_cfg(device="cuda:0")
model = M.build_model(cfg); model.train()
inputs = [dict(x, image=x["image"].cuda()) for x in _si_inputs(5, [(S, S)] * B)]
params = [p for p in model.parameters() if p.requires_grad]
st = model.prepare_batch(inputs)
def fwd_stages():
    feats = model.backbone(st["images"])
    enc = model.encoder(feats)
    out = model.decoder(enc)
    losses = model.criterion(out, st["targets"])
    return feats, enc, out, losses
def summary(feats, enc, out, losses):
    d = {"res5": float(feats["res5"].float().abs().mean()), "enc": float(enc.float().abs().mean()),
         "logits": float(out["pred_logits"].float().abs().mean()), "masks": float(out["_masks_nhwc"].float().abs().mean())}
    d.update({k: float(v) for k, v in losses.items()})
    return {k: round(v, 4) for k, v in d.items()}
# warm-up + capture of the forward alone
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s), torch.no_grad():
    for _ in range(2): fwd_stages()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.no_grad():
    with torch.cuda.graph(g):
        cap = fwd_stages()
for rep in range(3):
    g.replay(); torch.cuda.synchronize()
    with torch.no_grad():
        ref = fwd_stages()
    print("fwd-only replay", rep, "graph", summary(*cap), "| eager", summary(*ref), flush=True)
    with torch.no_grad():
        for p in params: p.add_(torch.randn_like(p) * 1e-4 * p.abs().mean())
# forward + backward captured, parameters perturbed between replays
g2 = torch.cuda.CUDAGraph()
def fb():
    f = fwd_stages()
    for p in params: p.grad = None
    sum(f[3].values()).backward()
    return f
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): fb()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g2):
    cap2 = fb()
for rep in range(3):
    g2.replay(); torch.cuda.synchronize()
    gn = float(sum(p.grad.float().pow(2).sum() for p in params).sqrt())
    ref = fb(); torch.cuda.synchronize()
    gr = float(sum(p.grad.float().pow(2).sum() for p in params).sqrt())
    print("fwd+bwd replay", rep, "graph", summary(*cap2), "gradnorm %.4e" % gn, "| eager", summary(*ref), "gradnorm %.4e" % gr, flush=True)
    with torch.no_grad():
        for p in params: p.add_(torch.randn_like(p) * 1e-4 * p.abs().mean())
