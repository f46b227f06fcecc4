"""Shared helpers of the whole-step GPU parity tests: one HIP step with every stored tensor the comparison needs, and the
oracle's forward / backward with the product's bf16 storage emulation, optionally with the forward state pinned to the
HIP values (teacher forcing, see tests/test_gpu_parity_bench.py for why)."""
import os

import torch

os.environ.setdefault("MI_LOSS_DPREDS", "1")   # the fused loss backward keeps the fp32 gradient tensor only on request

import yolox_oracle as O
import yolov7_d2_amd as M

DEV = "cuda"


class QuantBf16(torch.autograd.Function):
    """the product's storage rounding: bf16 activations forward, bf16 gradients backward"""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def build_model(depth, width, sd, conf=None, depthwise=False):
    cfg = M.yolox_s_cfg(device=DEV)
    cfg.MODEL.YOLO.DEPTH_MUL, cfg.MODEL.YOLO.WIDTH_MUL = depth, width
    cfg.MODEL.DARKNET.DEPTH_WISE = bool(depthwise)
    if conf is not None:
        cfg.MODEL.YOLO.CONF_THRESHOLD = conf
    model = M.build_model(cfg)
    model.load_state_dict(sd)
    return model


def hip_step(sd, imgs, labels, depth=0.33, width=0.5, want_y=False, depthwise=False):
    """forward + loss + backward of the HIP plan on (imgs, labels); returns host copies"""
    B, _, H, W = imgs.shape
    model = build_model(depth, width, sd, depthwise=depthwise)
    model.train()
    ps = model.plan_for(B, H, W, True)
    ps.image.copy_(imgs.to(DEV)); ps.labels.copy_(labels.to(DEV))
    ps.gw().fill_(1.0)
    ps.plan.run("fwd"); ps.plan.run("bwd")
    torch.cuda.synchronize()
    nch = ps.nch
    out = dict(
        raw=ps.preds().float().cpu().clone(), anchors=ps.anchors.float().cpu().clone(),
        losses=ps.loss_out()[:8].cpu().clone(),
        dpreds=ps.plan.buf_view(ps.loss["dpreds"], torch.float32, B * ps.A * nch).view(B, ps.A, nch).cpu().clone(),
        fg=ps.plan.buf_view(ps.loss["fg"], torch.uint8, B * ps.A).view(B, ps.A).cpu().clone(),
        mgt=ps.plan.buf_view(ps.loss["matched_gt"], torch.int32, B * ps.A).view(B, ps.A).cpu().clone(),
        grads={n: model.params.grad_of(p).detach().float().cpu().clone() for n, p in model.named_parameters()},
        rm={k: v.detach().float().cpu().clone() for k, v in model.state_dict().items() if "running_mean" in k})
    if want_y:   # every BaseConv's stored conv output (bf16 NHWC, buffer "<layer>.y"), flat
        out["y"] = {b.name: ps.plan.buf_view(b, torch.bfloat16).cpu().clone() for b in ps.builder.bufs
                    if b.name.endswith(".y")}
    del model
    return out


def oracle_backward(sd, imgs, dpreds, force, depth=0.33, width=0.5, depthwise=False):
    """oracle forward (bf16 storage emulation; conv outputs forced to `force` when given) and autograd backward from
    the given d(loss)/d(raw)"""
    osd = {k: v.clone() for k, v in sd.items()}
    for k, v in osd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    net = O.Net(osd, depth, width, 80, training=True, quant=QuantBf16.apply, force=force, depthwise=depthwise)
    raw_ref, hw = net.forward_raw(imgs)
    raw_ref.backward(dpreds)
    return dict(raw=raw_ref.detach(), grads={k: v.grad.detach().clone() for k, v in osd.items() if v.requires_grad},
                osd=osd, force_err=dict(net.force_err))


def grad_table(hip_grads, ref_grads):
    """[(name, cosine, relative L2, |hip|, |ref|)]"""
    rows = []
    for n, g in hip_grads.items():
        r = ref_grads[n]
        gn, rn = float(g.norm()), float(r.norm())
        if gn == 0.0 and rn == 0.0:      # e.g. the class branch of a level without a foreground anchor: exactly zero on both sides
            rows.append((n, 1.0, 0.0, 0.0, 0.0))
            continue
        cos = float((g * r).sum() / (gn * rn + 1e-30))
        rel = float((g - r).norm() / (rn + 1e-30))
        rows.append((n, cos, rel, gn, rn))
    return rows
