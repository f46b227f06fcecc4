#!/usr/bin/env python
"""sweep split-K / pixel-tile choices of mi_conv2d_wgrad over the distinct layer shapes of the YOLOX-s step"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from bench import synth_batch_device

B, S = 16, 640
torch.manual_seed(0)
model = M.build_model(M.yolox_s_cfg(device="cuda"))
model.train()
ps = model.plan_for(B, S, S, True)
imgs, labels = synth_batch_device(B, S, S, 1234, "cuda")
ps.image.copy_(imgs); ps.labels.copy_(labels); ps.gw().fill_(1.0)
plan = ps.plan
plan.run("fwd"); plan.run("bwd"); torch.cuda.synchronize()
lib = L.lib()
ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
arr, n = plan.bwd_cmds
seen = {}
for k in range(n):
    if L.OPS[arr[k].op] != "WGRAD":
        continue
    d = plan.cmd_descs["bwd"][k]
    key = (d.H, d.W, d.CinPad, d.CoutPad, d.ntaps, d.stride)
    seen.setdefault(key, []).append(k)
tot_auto = tot_best = 0.0
for key, ks in sorted(seen.items(), key=lambda kv: -len(kv[1])):
    d0 = plan.cmd_descs["bwd"][ks[0]]
    res = []
    for tp, ns in ((0, 0), (0, 2), (0, 3), (0, 4), (64, 2), (64, 3), (64, 4), (128, 2), (128, 3), (128, 4)):
        for sk in (0, 32, 64, 128, 256, 512):
            d = L.mi_wgrad_desc.from_buffer_copy(d0)
            d.splitk, d.cfg_tp, d.cfg_ns = sk, tp, ns
            need = lib.mi_conv2d_wgrad_plan(C.byref(d))
            if need < 0 or need > ws.numel():
                continue
            d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
            cmd = (L.mi_cmd * 1)()
            cmd[0].op = L.OP["WGRAD"]
            cmd[0].p[0] = C.cast(C.pointer(d), C.c_void_p).value
            per = (C.c_float * 1)(); tot = C.c_float(0)
            rc = lib.mi_cmdlist_time(cmd, 1, 5, C.byref(tot), per, L.stream_ptr())
            if rc < 0:
                continue
            res.append((per[0] * 1e3, tp, sk, need >> 20, ns))
    auto = [r for r in res if r[1] == 0 and r[2] == 0 and r[4] == 0][0]
    best = min(res)
    tot_auto += auto[0] * len(ks); tot_best += best[0] * len(ks)
    print(f"{key} x{len(ks)}: auto {auto[0]:.1f}us ws{auto[3]}MB | best {best[0]:.1f}us tp{best[1]} split{best[2]} ws{best[3]}MB | " +
          " ".join(f"tp{r[1]}/ns{r[4]}/s{r[2]}:{r[0]:.0f}" for r in sorted(res, key=lambda r: r[0])[:6]))
print(f"total auto {tot_auto/1e3:.3f} ms  best {tot_best/1e3:.3f} ms")
