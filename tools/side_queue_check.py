"""the side-queue form of the step (MI_WGRAD_SIDE) computes the step: after 3 steps on one batch the parameters equal the
default form's to the tolerance of the fp64 BatchNorm accumulation order; prints the max relative difference"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import yolox_oracle as O
    import yolov7_d2_amd as M
    from yolov7_d2_amd.engine import NativeTrainer
    model = M.build_model(M.yolox_s_cfg(device="cuda"))
    model.load_state_dict(O.init_state_dict(0.33, 0.5, 80, seed=0))
    imgs, labels = O.synth_batch(4, 256, 256, seed=17, max_gt=5)
    tr = NativeTrainer(model, lr=0.002, use_graph=sys.argv[2] == "graph")
    st = tr.load_batch(imgs.cuda(), labels.cuda())
    for _ in range(4):
        tr.step(st)
    print("losses", tr.losses(st)[:4].tolist())
    torch.save(tr.params.data.cpu(), sys.argv[3])
    sys.exit(0)

outs = {}
for name, env in (("default", {}), ("side2", {"MI_WGRAD_SIDE": "2"}), ("side3_mask64", {"MI_WGRAD_SIDE": "3", "MI_WGRAD_CUMASK": "64", "MI_MAIN_CUMASK": "1"})):
    for mode in ("graph", "eager"):
        f = f"/tmp/sq_{name}_{mode}.pt"
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, __file__, "child", mode, f], env=e, capture_output=True, text=True)
        print(name, mode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
        outs[(name, mode)] = torch.load(f)
ref = outs[("default", "graph")]
for k, v in outs.items():
    d = (v - ref).abs().max() / ref.abs().max()
    print(k, "max |dp| / max |p| vs default graph:", float(d))
    assert d < 2e-3, k
print("OK")
