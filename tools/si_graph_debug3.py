"""GPU box: which decoder stage of SparseInst goes wrong on the SECOND replay of a captured forward"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd.modeling import sparseinst as S
B, H, W = 8, 80, 80
torch.manual_seed(0)
cfg = M.sparse_inst_r50_giam_cfg(device="cuda:0")
dec = S.GroupIAMDecoder(cfg).cuda()
params = list(dec.parameters())
enc = (torch.randn(B, 256, H, W, device="cuda") * 30).to(torch.bfloat16)
def stages():
    out = {}
    f = torch.cat([dec.compute_coordinates(enc), enc], dim=1); out["cat"] = f
    ib = dec.inst_branch
    x = S._run_stack(ib.inst_convs, f); out["inst_convs"] = x
    G = ib.num_groups
    cin, cout = x.shape[1] // G, ib.iam_conv.out_channels // G
    iam = torch.cat([torch.ops.mi355.conv2d(x[:, g * cin:(g + 1) * cin], ib.iam_conv.weight[g * cout:(g + 1) * cout],
                                            ib.iam_conv.bias[g * cout:(g + 1) * cout], 1, 1) for g in range(G)], 1); out["iam"] = iam
    inst, norm = S._aggregate(iam, x); out["aggregate"] = inst; out["norm"] = norm
    inst = inst / norm.clamp(min=1e-6, max=1e5)[:, :, None]
    Bn, N = inst.shape[:2]; d4 = N // 4
    inst = inst.reshape(Bn, 4, d4, -1).transpose(1, 2).reshape(Bn, d4, -1)
    fc = S._Ew1.apply(S._linear(inst, ib.fc), "relu"); out["fc"] = fc
    out["cls"] = S._linear(fc, ib.cls_score); out["kernel"] = S._linear(fc, ib.mask_kernel)
    mf = dec.mask_branch(f); out["mask_feat"] = mf
    return out
def summ(o): return {k: "%.5g" % float(v.float().abs().mean()) for k, v in o.items()}
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s), torch.no_grad():
    for _ in range(2): stages()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.no_grad():
    with torch.cuda.graph(g):
        cap = stages()
for rep in range(3):
    g.replay(); torch.cuda.synchronize()
    with torch.no_grad(): ref = stages()
    print("replay", rep, "\n   graph", summ(cap), "\n   eager", summ(ref), flush=True)
    if os.environ.get("PERTURB", "1") == "1":
        with torch.no_grad():
            for p in params: p.add_(torch.randn_like(p) * 1e-4 * p.abs().mean())
# ---- the suspected torch reduction alone, captured and replayed
x = torch.rand(8, 80, 80, 400, device="cuda").to(torch.bfloat16)
ref = x.sum((1, 2), dtype=torch.float32)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): x.sum((1, 2), dtype=torch.float32)
torch.cuda.current_stream().wait_stream(s)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    y = x.sum((1, 2), dtype=torch.float32)
for rep in range(3):
    g3.replay(); torch.cuda.synchronize()
    print("torch sum((1,2)) of bf16 [8,80,80,400] replay", rep, "max rel err vs eager %.3e" % float(((y - ref).abs() / ref).max()), flush=True)
