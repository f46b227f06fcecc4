"""GPU (-m gpu): every HIP kernel of libmi355det against a CPU fp32 reference of the same op computed from the
same bf16-rounded operands (tolerances: bf16 outputs rtol 2e-2 / atol 2e-2; fp32 reductions 1e-3)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import yolox_oracle as O
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.plan import PlanBuilder, TRef

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(t):
    return t.to(torch.bfloat16).float()


def relerr(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def sp():
    return L.stream_ptr()


def test_library_and_device():
    assert L.lib().mi_version() >= 100
    assert L.lib().mi_device_count() >= 1


def test_mfma_lane_layouts():
    g = torch.Generator().manual_seed(0)
    for name, (m, k, n) in (("mi_probe_mfma32", (32, 16, 32)), ("mi_probe_mfma16", (16, 32, 16))):
        a = bf(torch.randn(m, k, generator=g))
        b = bf(torch.randn(k, n, generator=g))   # asymmetric: catches transposed C/D maps
        d = torch.zeros(m, n, device=DEV)
        ad, bd = a.to(DEV, torch.bfloat16), b.to(DEV, torch.bfloat16)   # keep alive across the launch
        L.check(getattr(L.lib(), name)(ad.data_ptr(), bd.data_ptr(), d.data_ptr(), sp()), name)
        torch.cuda.synchronize()
        np.testing.assert_allclose(d.cpu().numpy(), (a @ b).numpy(), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------ BaseConv via a 1-layer plan
class _Case:
    """Conv(k,s) -> BN(train) -> SiLU (+res) as a one-layer plan, with a given upstream gradient."""

    def __init__(self, N, H, W, Cin, Cout, k, s, res=False, xC=None, seed=0, in_slice=False):
        g = torch.Generator().manual_seed(seed)
        self.N, self.H, self.W, self.Cin, self.Cout, self.k, self.s = N, H, W, Cin, Cout, k, s
        self.weight = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).to(DEV)
        self.wgrad = torch.zeros_like(self.weight)
        self.gamma = (1 + 0.2 * torch.randn(Cout, generator=g)).to(DEV)
        self.beta = (0.2 * torch.randn(Cout, generator=g)).to(DEV)
        self.rm, self.rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
        self.nbt = torch.zeros((), dtype=torch.long, device=DEV)
        self.ggamma, self.gbeta = torch.zeros(Cout, device=DEV), torch.zeros(Cout, device=DEV)
        b = PlanBuilder(DEV, training=True)
        xC = Cin if xC is None else xC
        if in_slice:   # x is a channel slice of a wider concat buffer (exercises ld != C)
            big = b.new_act(N, H, W, 2 * xC, "xbig")
            self.x = big.slice(xC, 2 * xC)
        else:
            self.x = b.new_act(N, H, W, xC, "x")
        pad = (k - 1) // 2
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        self.Ho, self.Wo = Ho, Wo
        self.res = b.new_act(N, Ho, Wo, Cout, "res") if res else None
        bn = dict(gamma=self.gamma, beta=self.beta, rm=self.rm, rv=self.rv, nbt=self.nbt, eps=1e-3, momentum=0.03,
                  ggamma=self.ggamma, gbeta=self.gbeta)
        self.out = b.base_conv("c", self.x, self.weight, bn, k, s, self.wgrad, res=self.res)
        assert b.grad_mode(self.out) == 0          # upstream gradient is written by the test
        self.plan = b.finalize()
        self.b = b
        self.xin = bf(torch.randn(N, Cin, H, W, generator=g))
        self.rin = bf(torch.randn(N, Cout, Ho, Wo, generator=g)) if res else None
        self.gout = bf(torch.randn(N, Cout, Ho, Wo, generator=g))
        xv = self.plan.view(self.x)
        xv.zero_()
        xv[:, :Cin] = self.xin.to(DEV, torch.bfloat16)
        if res:
            self.plan.view(self.res).copy_(self.rin.to(DEV))
        self.plan.view(self.out.grad).copy_(self.gout.to(DEV))
        if self.res is not None:
            self.plan.view(self.res.grad).fill_(0.25)   # must be overwritten (first writer) by the kernel

    def run(self):
        self.plan.run("fwd")
        self.plan.run("bwd")
        torch.cuda.synchronize()

    def reference(self):
        x = self.xin.clone().requires_grad_(True)
        w = bf(self.weight.cpu()).requires_grad_(True)
        gamma, beta = self.gamma.cpu().clone().requires_grad_(True), self.beta.cpu().clone().requires_grad_(True)
        y = F.conv2d(x, w, None, self.s, (self.k - 1) // 2)
        yq = y + (bf(y) - y).detach()                      # stored as bf16
        z = F.batch_norm(yq, None, None, gamma, beta, True, 0.03, 1e-3)
        a = F.silu(z)
        r = None
        if self.rin is not None:
            r = self.rin.clone().requires_grad_(True)
            a = a + r
        a.backward(self.gout)
        return dict(y=y.detach(), out=a.detach(), dx=x.grad, dw=w.grad, dgamma=gamma.grad, dbeta=beta.grad,
                    dres=None if r is None else r.grad, mean=yq.detach().mean((0, 2, 3)),
                    var=yq.detach().var((0, 2, 3), unbiased=True))


CASES = [
    # N, H, W, Cin, Cout, k, s, res
    (2, 20, 20, 64, 64, 3, 1, True),      # KC=64 BN=64, W<TW tiling (20x20 -> 6x20 tiles)
    (1, 40, 40, 32, 32, 3, 1, False),     # KC=32 BN=32
    (2, 16, 24, 12, 32, 3, 1, False),     # stem: Cin 12 padded to 16, KC=16
    (2, 32, 32, 32, 64, 3, 2, False),     # stride 2: fwd halo stride, dgrad parity classes
    (2, 22, 14, 64, 128, 3, 2, False),    # stride 2, odd output tiles, BN=128
    (2, 20, 12, 128, 128, 1, 1, False),   # 1x1, direct wgrad into the OIHW gradient
    (1, 9, 7, 64, 32, 3, 1, True),        # odd sizes, partial tiles
    (2, 8, 8, 256, 512, 1, 1, False),     # deep 1x1: 4 k-chunks, 4 cout tiles
    (1, 80, 80, 128, 128, 3, 1, False),   # head-sized 3x3 (8x16 tiles, many blocks)
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[3]}to{c[4]}_k{c[5]}s{c[6]}_{c[1]}x{c[2]}" for c in CASES])
def test_base_conv_fwd_bwd(case):
    N, H, W, Cin, Cout, k, s, res = case
    c = _Case(N, H, W, Cin, Cout, k, s, res=res, xC=16 if Cin == 12 else None)
    c.run()
    ref = c.reference()
    # raw conv output (bf16) and activated output
    ynm = [t for t in c.b.bufs if t.name == "c.y"][0]
    y = c.plan.view(TRef(ynm, N, c.Ho, c.Wo, Cout, Cout)).float().cpu()
    np.testing.assert_allclose(y.numpy(), ref["y"].numpy(), rtol=2e-2, atol=2e-2)
    out = c.plan.view(c.out).float().cpu()
    np.testing.assert_allclose(out.numpy(), ref["out"].numpy(), rtol=2e-2, atol=3e-2)
    # running statistics (fp32 path; momentum 0.03)
    np.testing.assert_allclose(c.rm.cpu().numpy(), 0.03 * ref["mean"].numpy(), rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(c.rv.cpu().numpy(), 0.97 + 0.03 * ref["var"].numpy(), rtol=2e-3, atol=1e-4)
    assert int(c.nbt) == 1
    # gradients: norm-relative error (bf16 dy operand)
    assert relerr(c.wgrad.cpu(), ref["dw"]) < 2e-2
    assert relerr(c.ggamma.cpu(), ref["dgamma"]) < 2e-2
    assert relerr(c.gbeta.cpu(), ref["dbeta"]) < 2e-2
    dx = c.plan.view(c.x.grad).float().cpu()[:, :Cin]
    assert relerr(dx, ref["dx"]) < 2e-2
    if res:
        dres = c.plan.view(c.res.grad).float().cpu()
        np.testing.assert_allclose(dres.numpy(), ref["dres"].numpy(), rtol=1e-2, atol=1e-2)


def test_conv_input_slice():
    """x read from / dx written to a channel slice of a wider buffer (ld != C)"""
    c = _Case(2, 12, 12, 64, 64, 3, 1, in_slice=True, seed=3)
    c.run()
    ref = c.reference()
    out = c.plan.view(c.out).float().cpu()
    np.testing.assert_allclose(out.numpy(), ref["out"].numpy(), rtol=2e-2, atol=3e-2)
    assert relerr(c.plan.view(c.x.grad).float().cpu(), ref["dx"]) < 2e-2
    assert relerr(c.wgrad.cpu(), ref["dw"]) < 2e-2


FORCED = [
    # N, H, W, Cin, Cout, k, s
    (1, 40, 40, 128, 128, 3, 1),
    (2, 32, 32, 64, 128, 3, 2),
    (2, 20, 20, 256, 64, 1, 1),
]


@pytest.mark.parametrize("case", FORCED, ids=[f"{c[3]}to{c[4]}_k{c[5]}s{c[6]}" for c in FORCED])
def test_conv_forced_configs_agree(case):
    """every (k-chunk, taps-per-step, cout tile, pixel tile) configuration the launcher may pick computes the same
    convolution (forward and every data-gradient launch): outputs equal up to the fp32 accumulation order"""
    import ctypes as C
    N, H, W, Cin, Cout, k, s = case
    c = _Case(N, H, W, Cin, Cout, k, s)
    c.run()
    lib = L.lib()
    ran = 0
    for which in ("fwd", "bwd"):
        arr, n = c.plan.fwd_cmds if which == "fwd" else c.plan.bwd_cmds
        for i in range(n):
            if L.OPS[arr[i].op] != "CONV":
                continue
            d0 = c.plan.cmd_descs[which][i]
            if d0.flags & L.MI_CONV_ACCUM:
                continue
            elems = d0.N * d0.outH * d0.outW * d0.ldy
            base = torch.empty(elems, dtype=torch.bfloat16, device=DEV)
            got = torch.empty(elems, dtype=torch.bfloat16, device=DEV)

            def run(d, dst):
                dst.fill_(0)
                d.y = dst.data_ptr()
                d.stats_acc = None
                L.check(lib.mi_conv2d(C.byref(d), L.stream_ptr()), "mi_conv2d")
                torch.cuda.synchronize()

            d = L.mi_conv_desc.from_buffer_copy(d0)
            run(d, base)
            K = d0.K8 * 8
            for kc in (16, 32, 64, 128):
                if K % kc:
                    continue
                for tps in [t for t in (1, 2, 3, 4, 9) if d0.ntaps % t == 0]:
                    for bn, th, tw in ((32, 8, 16), (64, 8, 8), (128, 4, 32), (64, 3, 20)):
                        if d0.CoutPad % bn:
                            continue
                        d = L.mi_conv_desc.from_buffer_copy(d0)
                        d.KC, d.TPS, d.BN, d.TH, d.TW = kc, tps, bn, th, tw
                        probe = L.mi_conv_desc.from_buffer_copy(d)
                        if lib.mi_conv2d_plan(C.byref(probe)) < 0:   # does not fit in LDS: the launcher refuses
                            continue
                        run(d, got)
                        np.testing.assert_allclose(got.float().cpu().numpy(), base.float().cpu().numpy(), rtol=2e-2,
                                                   atol=2e-2, err_msg=f"{which}[{i}] KC{kc} TPS{tps} BN{bn} {th}x{tw}")
                        ran += 1
    assert ran > 20


@pytest.mark.parametrize("strides,nfused", [((1, 2), 1), ((2, 1), 4)])
def test_bn_backward_sums_fused_into_dgrad(monkeypatch, strides, nfused):
    """conv -> BN -> SiLU -> two consumer convs (3x3 stride 1, and stride 2 = four parity-class launches): with
    MI_FUSE_BN_BWD=1 the first layer's BatchNorm-backward sums come from the data-gradient epilogue(s) of the consumer
    that writes its output gradient LAST (the first consumer in forward order; MI_CONV_BNBWD); every gradient must
    equal the unfused plan's (same math, different summation order)"""
    outs = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("MI_FUSE_BN_BWD", fuse)
        g = torch.Generator().manual_seed(9)
        N, H, W, C0, C1, C2 = 2, 24, 24, 32, 64, 64
        b = PlanBuilder(DEV, training=True)
        x = b.new_act(N, H, W, C0, "x")
        mk = lambda co, ci, k: (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).to(DEV)
        ws = [mk(C1, C0, 3), mk(C2, C1, 3), mk(C2, C1, 3)]
        wg = [torch.zeros_like(w) for w in ws]
        bns = []
        for co in (C1, C2, C2):
            bns.append(dict(gamma=(1 + 0.1 * torch.randn(co, generator=g)).to(DEV), beta=(0.1 * torch.randn(co, generator=g)).to(DEV),
                            rm=torch.zeros(co, device=DEV), rv=torch.ones(co, device=DEV),
                            nbt=torch.zeros((), dtype=torch.long, device=DEV), eps=1e-3, momentum=0.03,
                            ggamma=torch.zeros(co, device=DEV), gbeta=torch.zeros(co, device=DEV)))
        h = b.base_conv("c0", x, ws[0], bns[0], 3, 1, wg[0])
        o1 = b.base_conv("c1", h, ws[1], bns[1], 3, strides[0], wg[1])
        o2 = b.base_conv("c2", h, ws[2], bns[2], 3, strides[1], wg[2])
        b.keep += bns
        for o in (o1, o2):
            assert b.grad_mode(o) == 0
        plan = b.finalize()
        tags = [c.tag for c in b.bwd]
        assert ("c0.bnred" in tags) == (fuse == "0")
        assert sum(t.endswith("+bnred") for t in tags) == (nfused if fuse == "1" else 0)
        plan.view(x).copy_(bf(torch.randn(N, C0, H, W, generator=g)).to(DEV))
        plan.view(o1.grad).copy_(bf(torch.randn(N, C2, H // strides[0], W // strides[0], generator=g)).to(DEV))
        plan.view(o2.grad).copy_(bf(torch.randn(N, C2, H // strides[1], W // strides[1], generator=g)).to(DEV))
        plan.run("fwd"); plan.run("bwd"); torch.cuda.synchronize()
        outs[fuse] = dict(dx=plan.view(x.grad).float().cpu(), dw0=wg[0].cpu(), dg0=bns[0]["ggamma"].cpu(),
                          db0=bns[0]["gbeta"].cpu(), dw1=wg[1].cpu())
    for k in outs["0"]:
        assert relerr(outs["1"][k], outs["0"][k]) < 2e-3, k


def test_dgrad_accumulates():
    """two consumers of one tensor: the second data gradient must add to the first (MI_CONV_ACCUM)"""
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 1, 16, 16, 64
    b = PlanBuilder(DEV, training=True)
    x = b.new_act(N, H, W, C, "x")
    ws = [(torch.randn(C, C, 1, 1, generator=g) / 8).to(DEV) for _ in range(2)]
    wg = [torch.zeros_like(w) for w in ws]
    outs = []
    for i in range(2):
        bn = dict(gamma=torch.ones(C, device=DEV), beta=torch.zeros(C, device=DEV), rm=torch.zeros(C, device=DEV),
                  rv=torch.ones(C, device=DEV), nbt=torch.zeros((), dtype=torch.long, device=DEV), eps=1e-3,
                  momentum=0.03, ggamma=torch.zeros(C, device=DEV), gbeta=torch.zeros(C, device=DEV))
        outs.append(b.base_conv(f"c{i}", x, ws[i], bn, 1, 1, wg[i]))
        b.keep.append(bn)
    for o in outs:
        b.grad_mode(o)
    plan = b.finalize()
    flags = [c.desc.flags for c in b.bwd if c.tag.endswith(".dgrad")]
    assert sorted(flags) == [0, L.MI_CONV_ACCUM]
    xin = bf(torch.randn(N, C, H, W, generator=g))
    gos = [bf(torch.randn(N, C, H, W, generator=g)) for _ in range(2)]
    plan.view(x).copy_(xin.to(DEV))
    for o, go in zip(outs, gos):
        plan.view(o.grad).copy_(go.to(DEV))
    plan.view(x.grad).fill_(7.0)  # garbage that the first (overwriting) dgrad must erase
    plan.run("fwd"); plan.run("bwd"); torch.cuda.synchronize()
    xr = xin.clone().requires_grad_(True)
    tot = 0
    for w, go in zip(ws, gos):
        y = F.conv2d(xr, bf(w.cpu()))
        yq = y + (bf(y) - y).detach()
        tot = tot + (F.silu(F.batch_norm(yq, None, None, None, None, True, 0.03, 1e-3)) * go).sum()
    tot.backward()
    assert relerr(plan.view(x.grad).float().cpu(), xr.grad) < 2.5e-2


# ------------------------------------------------------------------------------------------ data-movement ops
def test_focus_upsample_spp():
    g = torch.Generator().manual_seed(1)
    N, H, W = 2, 16, 24
    img = torch.randint(0, 256, (N, 3, H, W), generator=g).float()
    b = PlanBuilder(DEV, training=True)
    image = img.to(DEV)
    f = b.focus(image, N, H, W)
    x = b.new_act(N, 6, 5, 32, "x")
    cat = b.new_act(N, 12, 10, 64, "cat")
    b.upsample_into("up", x, cat.slice(32, 64))
    sx = b.new_act(N, 7, 9, 32, "sx")
    scat = b.new_act(N, 7, 9, 128, "scat")
    b.spp_into("spp", sx, scat.slice(32, 64), scat.slice(64, 96), scat.slice(96, 128))
    b.grad_mode(cat); b.grad_mode(scat)
    plan = b.finalize()
    xin = bf(torch.randn(N, 32, 6, 5, generator=g))
    # bf16 ties inside a 13x13 window are likely with few mantissa bits: make the values distinct
    sxin = bf(torch.randperm(N * 32 * 7 * 9, generator=g).float().view(N, 32, 7, 9) / 64.0)
    gcat = bf(torch.randn(N, 64, 12, 10, generator=g))
    gscat = bf(torch.randn(N, 128, 7, 9, generator=g))
    plan.view(x).copy_(xin.to(DEV)); plan.view(sx).copy_(sxin.to(DEV))
    plan.view(cat.grad).copy_(gcat.to(DEV)); plan.view(scat.grad).copy_(gscat.to(DEV))
    plan.run("fwd"); plan.run("bwd"); torch.cuda.synchronize()
    # focus: channel order TL, BL, TR, BR (wrappers.py:212-219), pad channels zero
    tl, tr, bl, br = img[..., ::2, ::2], img[..., ::2, 1::2], img[..., 1::2, ::2], img[..., 1::2, 1::2]
    fo = plan.view(f).float().cpu()
    assert torch.equal(fo[:, :12], torch.cat((tl, bl, tr, br), 1)) and float(fo[:, 12:].abs().max()) == 0
    up = plan.view(cat.slice(32, 64)).float().cpu()
    assert torch.equal(up, F.interpolate(xin, scale_factor=2, mode="nearest"))
    xr = xin.clone().requires_grad_(True)
    (F.interpolate(xr, scale_factor=2, mode="nearest") * gcat[:, 32:]).sum().backward()
    np.testing.assert_allclose(plan.view(x.grad).float().cpu().numpy(), bf(xr.grad).numpy(), rtol=1e-2, atol=1e-2)
    sr = sxin.clone().requires_grad_(True)
    tot = 0
    for i, ks in enumerate((5, 9, 13)):
        p = F.max_pool2d(sr, ks, 1, ks // 2)
        assert torch.equal(plan.view(scat.slice(32 * (i + 1), 32 * (i + 2))).float().cpu(), p.detach())
        tot = tot + (p * gscat[:, 32 * (i + 1): 32 * (i + 2)]).sum()
    tot.backward()
    np.testing.assert_allclose(plan.view(sx.grad).float().cpu().numpy(), sr.grad.numpy(), rtol=2e-2, atol=2e-2)


def test_sgd_momentum_matches_torch():
    g = torch.Generator().manual_seed(2)
    n = 70001
    p0, g0 = torch.randn(n, generator=g), torch.randn(n, generator=g)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pd, gd, md = p0.to(DEV).clone(), g0.to(DEV), torch.zeros(n, device=DEV)
    segs = []
    k = 0
    while k < n:
        c = min(16384, n - k); segs.append((k, c)); k += c
    arr = (L.mi_sgd_seg * len(segs))()
    for i, (o, c) in enumerate(segs):
        arr[i].offset, arr[i].count, arr[i].weight_decay, arr[i].lr = o, c, 1e-4, 0.01
    sd = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    for step in range(3):
        p.grad = g0.clone()
        opt.step()
        L.check(L.lib().mi_sgd_momentum_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), sd.data_ptr(), len(segs),
                                             0.9, 1.0, 0, sp()), "sgd")
    torch.cuda.synchronize()
    np.testing.assert_allclose(pd.cpu().numpy(), p.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_argument_errors_are_reported_not_launched():
    d = L.mi_conv_desc()
    rc = L.lib().mi_conv2d(C.byref(d), sp())
    assert rc == -1 and b"null" in L.lib().mi_last_error()


def test_batched_pack_and_split_match_the_per_layer_launches():
    """the one-launch weight packing (flat grid over all layers) and the one-launch split of the loss gradient write the
    same bytes as the per-layer kernels - including padded channel counts and 1x1 / 3x3 / 4x4-tap weights"""
    lib = L.lib()
    g = torch.Generator().manual_seed(5)
    shapes = [(32, 16, 3, 16, 32, 32, 16), (24, 48, 1, 64, 32, 32, 64), (85, 128, 1, 128, 96, 96, 128),
              (64, 12, 4, 16, 64, 64, 16), (256, 256, 3, 256, 256, 256, 256), (40, 24, 3, 32, 64, 64, 32)]
    jobs = (L.mi_pack_job * len(shapes))()
    keep, single = [], []
    for j, (Cout, Cin, k, CinPad, CoutPad, CoutPadK, CinPadN) in zip(jobs, shapes):
        w = torch.randn(Cout, Cin, k, k, generator=g).to(DEV)
        KK = k * k
        wf = torch.full((KK * CinPad * CoutPad,), 7.0, dtype=torch.bfloat16, device=DEV)
        wd = torch.full((KK * CoutPadK * CinPadN,), 7.0, dtype=torch.bfloat16, device=DEV) if Cout != 85 else None
        wf1, wd1 = torch.zeros_like(wf), (torch.zeros_like(wd) if wd is not None else None)
        L.check(lib.mi_pack_conv_weight(w.data_ptr(), Cout, Cin, k, k, wf1.data_ptr(), CinPad, CoutPad,
                                        wd1.data_ptr() if wd1 is not None else None, CoutPadK, CinPadN, sp()), "pack")
        j.w, j.wf, j.wd = w.data_ptr(), wf.data_ptr(), (wd.data_ptr() if wd is not None else None)
        j.Cout, j.Cin, j.KK, j.CinPad, j.CoutPad, j.CoutPadK, j.CinPadN = Cout, Cin, KK, CinPad, CoutPad, CoutPadK, CinPadN
        keep.append((w, wf, wd))
        single.append((wf1, wd1))
    nblk = L.check(lib.mi_pack_jobs_layout(jobs, len(shapes)), "layout")
    assert nblk == sum(-(-max(s[4], s[5] if s[0] != 85 else 0) // 32) * -(-max(s[3], s[6] if s[0] != 85 else 0) // 64) for s in shapes)
    tab = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(DEV)
    L.check(lib.mi_pack_conv_weights_batch(tab.data_ptr(), len(shapes), nblk, 16, sp()), "pack_batch")
    torch.cuda.synchronize()
    for (w, wf, wd), (wf1, wd1) in zip(keep, single):
        assert torch.equal(wf.view(torch.int16), wf1.view(torch.int16))
        if wd is not None:
            assert torch.equal(wd.view(torch.int16), wd1.view(torch.int16))

    B, nch = 3, 85
    levels = [(0, 20 * 12), (240, 10 * 6), (300, 5 * 3)]
    A = 315
    dp = torch.randn(B, A, nch, generator=g).to(DEV)
    sj = (L.mi_split_job * 9)()
    outs = []
    for li, (a0, HW) in enumerate(levels):
        for bi, (c0, nc) in enumerate([(0, 4), (4, 1), (5, 80)]):
            ld = (nc + 31) // 32 * 32
            d1 = torch.full((B * HW * ld,), 3.0, dtype=torch.bfloat16, device=DEV)
            d2 = torch.full((B * HW * ld,), 5.0, dtype=torch.bfloat16, device=DEV)
            L.check(lib.mi_yolox_split_dpreds(dp.data_ptr(), B, A, nch, a0, HW, c0, nc, d1.data_ptr(), ld, sp()), "split")
            j = sj[li * 3 + bi]
            j.dst, j.a0, j.HW, j.c0, j.nc, j.ld = d2.data_ptr(), a0, HW, c0, nc, ld
            outs.append((d1, d2))
    L.check(lib.mi_yolox_split_dpreds_batch(dp.data_ptr(), B, A, nch, sj, 9, sp()), "split_batch")
    torch.cuda.synchronize()
    for d1, d2 in outs:
        assert torch.equal(d1.view(torch.int16), d2.view(torch.int16))
    sj[0].ld = 12
    assert lib.mi_yolox_split_dpreds_batch(dp.data_ptr(), B, A, nch, sj, 9, sp()) < 0
    lib.mi_last_error()


@pytest.mark.parametrize("C,npix,act,res", [(64, 16 * 40 * 40, 1, 0), (24, 5000, 1, 1), (256, 3 * 20 * 20, 0, 2),
                                           (32, 16 * 320 * 320, 1, 0), (128, 16 * 80 * 80, 1, 2), (80, 777, 1, 0)])
def test_bn_backward_fused_matches_two_pass(C, npix, act, res):
    """the one-launch BatchNorm backward (registers across a grid-wide barrier) against reduce + apply on the same inputs:
    the channel sums (dgamma, dbeta) agree to fp32 summation-order noise and dy / dres to one bf16 ulp of that; every
    block reached the barrier (the give-up flag stays 0); the 16x320x320x32 case exceeds the register capacity and streams
    its tail; C = 24 / 80 are not powers of two; res = 1 overwrites, 2 accumulates the residual gradient"""
    lib = L.lib()
    g = torch.Generator().manual_seed(C + npix)
    ld = (C + 31) // 32 * 32
    da = torch.randn(npix, ld, generator=g).to(DEV).to(torch.bfloat16)
    y = (torch.randn(npix, ld, generator=g) * 1.5 + 0.3).to(DEV).to(torch.bfloat16)
    yf = y[:, :C].float()
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    invstd = (var + 1e-3).rsqrt()
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * gamma * invstd).contiguous()
    CA = ld
    outs = []
    for fused in (0, 1):
        dacc = torch.zeros(L.MI_BN_SLOTS * CA * 2, dtype=torch.float64, device=DEV)
        dy = torch.zeros(npix, ld, dtype=torch.bfloat16, device=DEV)
        dres = (torch.ones(npix, ld, dtype=torch.bfloat16, device=DEV) * 0.25) if res else None
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        bar = torch.zeros(L.MI_BN_BAR_WORDS, dtype=torch.int32, device=DEV)
        common = (da.data_ptr(), ld, y.data_ptr(), ld, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr())
        tail = (dg.data_ptr(), db.data_ptr(), dy.data_ptr(), ld, dres.data_ptr() if res else None, ld if res else 0,
                int(res == 2), npix, C, act)
        if fused:
            for _ in range(3):      # repeated launches reuse the barrier words (generation counter)
                dacc.zero_()
                if res == 2:
                    dres.fill_(0.25)
                L.check(lib.mi_bn_act_bwd_fused(*common, gamma.data_ptr(), dacc.data_ptr(), L.MI_BN_SLOTS, npix, *tail,
                                                bar.data_ptr(), sp()), "fused")
        else:
            nblk = max(1, min(1024, math.ceil(npix / (256 // (C // 8)) / 4)))
            L.check(lib.mi_bn_act_bwd_reduce(*common, dacc.data_ptr(), L.MI_BN_SLOTS, nblk, npix, C, act, sp()), "reduce")
            L.check(lib.mi_bn_act_bwd_apply(*common, gamma.data_ptr(), dacc.data_ptr(), L.MI_BN_SLOTS, npix, *tail, sp()), "apply")
        torch.cuda.synchronize()
        outs.append((dy.float(), dres.float() if res else None, dg.clone(), db.clone(), bar.cpu()))
    (dy0, dr0, dg0, db0, _), (dy1, dr1, dg1, db1, bar1) = outs
    assert int(bar1[2]) == 0 and int(bar1[0]) == 0 and int(bar1[64]) == 3     # flag, top arrivals, generation of group 0
    scale_g = dg0.abs().max().clamp_min(1.0)
    assert float((dg0 - dg1).abs().max() / scale_g) < 2e-5 and float((db0 - db1).abs().max() / db0.abs().max().clamp_min(1.0)) < 2e-5
    assert torch.all(dy1[:, C:] == 0)
    diff = (dy0 - dy1).abs()
    assert float(diff.max()) <= float(dy0.abs().max()) * 2 ** -7         # at most an ulp of the largest magnitude
    assert float((diff > 0).float().mean()) < 0.02                       # and almost every element identical
    if res:
        assert torch.equal(dr0, dr1)


@pytest.mark.parametrize("N,H,W", [(2, 16, 24), (1, 6, 10), (3, 64, 96), (2, 8, 12), (2, 6, 32), (1, 70, 64), (2, 640, 640)])
def test_focus_pack_float_and_uint8_bit_exact(N, H, W):
    """Focus (wrappers.py:202-220) from the float image and from the uint8 image, the 4-pixels-per-thread kernels (W % 8
    == 0), the scalar ones and the row-staged uint8 kernel (W % 16 == 0; an odd number of output rows leaves half a row
    pair): TL, BL, TR, BR x (c0, c1, c2) into 16-channel pixels, the 4 pad channels zero - exact"""
    lib = L.lib()
    g = torch.Generator().manual_seed(H * W)
    u8 = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8)
    f32 = u8.float()
    ref = torch.cat([f32[..., ::2, ::2], f32[..., 1::2, ::2], f32[..., ::2, 1::2], f32[..., 1::2, 1::2]], 1)   # [N,12,H/2,W/2]
    ref = ref.permute(0, 2, 3, 1)
    import os
    prev = os.environ.get("MI_FOCUS_ROWS")
    try:
        # uint8: the lane-pair-store kernel (round 6, the default), the row-staged kernel, the per-pixel kernel - forced
        for src, fn, mode in ((f32.to(DEV), lib.mi_focus_pack, None), (u8.to(DEV), lib.mi_focus_pack_u8, "2"),
                              (u8.to(DEV), lib.mi_focus_pack_u8, "1"), (u8.to(DEV), lib.mi_focus_pack_u8, "0")):
            if mode is not None:
                os.environ["MI_FOCUS_ROWS"] = mode
            for ld in (16, 24):            # a contiguous output and a channel slice of a wider buffer
                out = torch.full((N, H // 2, W // 2, ld), 5.0, dtype=torch.bfloat16, device=DEV)
                L.check(fn(src.data_ptr(), N, H, W, out.data_ptr(), ld, sp()), "focus")
                torch.cuda.synchronize()
                o = out.float().cpu()
                assert torch.equal(o[..., :12], ref) and torch.all(o[..., 12:16] == 0), (mode, ld)
                assert torch.all(o[..., 16:] == 5.0)
    finally:
        if prev is None:
            os.environ.pop("MI_FOCUS_ROWS", None)
        else:
            os.environ["MI_FOCUS_ROWS"] = prev
