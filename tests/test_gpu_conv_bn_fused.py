"""GPU (-m gpu): convolution + train-mode BatchNorm + SiLU (+ residual) as ONE launch (mi_conv2d_bn_fwd: the BatchNorm pass
is the second phase of the persistent streaming 1x1 / weight-stationary 3x3 kernels behind a grid barrier; the whole
BaseConv.forward of backbone/layers/wrappers.py:76-83 and the Bottleneck shortcut :119-123) against the two-launch form
(mi_conv2d + mi_bn_act_fwd, themselves held to the oracle by test_gpu_kernels / test_gpu_parity_bench): the activation,
scale / shift / mean / invstd and the running statistics must be BIT-IDENTICAL - same sums, same finalisation, same
expression per element - whenever both forms cut the tensor into the same blocks (the per-block fp32 partial sums are then
the same numbers); the 32- and 64-channel 3x3 kernels run fewer blocks per CU in the one-launch form (every block must be
resident for the grid barrier), so there the statistics agree to fp32 summation error and the activation to one bf16 ulp."""
import ctypes as C

import pytest
import torch

from yolov7_d2_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda"
NSL = 8


def sp():
    return L.stream_ptr()


def _pack(w):
    Cout, K, kh, kw = w.shape
    img = torch.empty(kh * kw * K * Cout, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib().mi_pack_conv_weight(w.data_ptr(), Cout, K, kh, kw, img.data_ptr(), K, Cout, None, 0, 0, sp()), "pack")
    return img


def _conv_desc(x, ldx, xoff, N, H, W, K, k, wimg, y, Cout, stats):
    d = L.mi_conv_desc()
    d.x, d.w, d.y = x.data_ptr() + xoff * 2, wimg.data_ptr(), y.data_ptr()
    d.ldx, d.ldy = ldx, Cout
    d.N, d.H, d.W, d.outH, d.outW, d.gridH, d.gridW = N, H, W, H, W, H, W
    d.in_stride = d.out_stride = 1
    d.K8, d.Cout, d.CoutPad, d.ntaps = K // 8, Cout, Cout, k * k
    t = 0
    for r in range(k):
        for s in range(k):
            d.tap_dy[t], d.tap_dx[t], d.tap_w[t] = r - k // 2, s - k // 2, r * k + s
            t += 1
    d.stats_acc, d.stats_slots = stats.data_ptr(), NSL
    return d


class _Layer:
    """one conv + BatchNorm layer: tensors of one run (fresh statistics / running buffers per run)"""

    def __init__(self, N, H, W, K, k, Cout, x, ldx, xoff, w, act, res, a_extra, seed):
        g = torch.Generator().manual_seed(seed)
        self.N, self.H, self.W, self.Cout = N, H, W, Cout
        npix = N * H * W
        self.y = torch.zeros(npix, Cout, dtype=torch.bfloat16, device=DEV)
        self.stats = torch.zeros(NSL, Cout, 2, dtype=torch.float64, device=DEV)
        self.gamma = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
        self.beta = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
        self.rm = torch.randn(Cout, generator=g).to(DEV)
        self.rv = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
        self.nbt = torch.full((1,), 3, dtype=torch.int64, device=DEV)
        self.scale, self.shift, self.mean, self.invstd = (torch.zeros(Cout, device=DEV) for _ in range(4))
        self.lda = Cout + a_extra
        self.a = torch.full((npix, self.lda), 7.0, dtype=torch.bfloat16, device=DEV)
        self.res = res
        self.act = act
        self.desc = _conv_desc(x, ldx, xoff, N, H, W, K, k, w, self.y, Cout, self.stats)

    def job(self):
        j = L.mi_bn_job()
        j.y, j.a, j.acc = self.y.data_ptr(), self.a.data_ptr() + (self.lda - self.Cout) * 2, self.stats.data_ptr()
        j.res = self.res.data_ptr() if self.res is not None else None
        j.gamma, j.beta, j.rmean, j.rvar, j.nbt = (t.data_ptr() for t in (self.gamma, self.beta, self.rm, self.rv, self.nbt))
        j.scale, j.shift, j.mean, j.invstd = (t.data_ptr() for t in (self.scale, self.shift, self.mean, self.invstd))
        j.npix = j.count = self.N * self.H * self.W
        j.ldy, j.lda, j.ldres, j.C, j.nslots, j.act = self.Cout, self.lda, (self.res.shape[1] if self.res is not None else 0), self.Cout, NSL, self.act
        j.eps, j.momentum = 1e-3, 0.03
        return j

    def bn_launch(self):
        j = self.job()
        L.check(L.lib().mi_bn_act_fwd(j.y, j.ldy, j.acc, j.nslots, j.count, j.gamma, j.beta, j.eps, j.momentum, j.rmean, j.rvar, j.nbt,
                                      j.scale, j.shift, j.mean, j.invstd, j.res, j.ldres, j.a, j.lda, j.npix, j.C, j.act, sp()), "bn_act_fwd")

    def tensors(self):
        return dict(y=self.y, a=self.a, scale=self.scale, shift=self.shift, mean=self.mean, invstd=self.invstd, rm=self.rm,
                    rv=self.rv, nbt=self.nbt)


def _fused(layers):
    n = len(layers)
    descs = (L.mi_conv_desc * n)()
    jobs = (L.mi_bn_job * n)()
    for i, l in enumerate(layers):
        C.memmove(C.byref(descs[i]), C.byref(l.desc), C.sizeof(L.mi_conv_desc))
        j = l.job()
        C.memmove(C.byref(jobs[i]), C.byref(j), C.sizeof(L.mi_bn_job))
    L.check(L.lib().mi_conv2d_bn_fwd(descs, jobs, n, sp()), "conv2d_bn_fwd")


def _compare(build, conv_entry, exact=True):
    """build() -> list of layers over the same inputs; run both forms on fresh layer state and compare everything.
    conv_entry: the explicit entry of the convolution kernel (all layers in one launch, as the fused form cuts them)"""
    ref, got = build(), build()
    descs = (L.mi_conv_desc * len(ref))()
    for i, l in enumerate(ref):
        C.memmove(C.byref(descs[i]), C.byref(l.desc), C.sizeof(L.mi_conv_desc))
    L.check(conv_entry(descs, len(ref), sp()), "conv")
    for l in ref:
        l.bn_launch()
    for it in range(3):         # the barrier words persist across launches: a second and third launch must work too
        fresh = build()
        for l, f in zip(got, fresh):
            l.stats.zero_()
            l.rm.copy_(f.rm); l.rv.copy_(f.rv); l.nbt.copy_(f.nbt)     # (what the previous fused launch advanced)
        _fused(got)
    torch.cuda.synchronize()
    fl = C.c_uint32(9)
    L.check(L.lib().mi_conv_bn_barrier_status(C.byref(fl)), "status")
    assert fl.value == 0
    for l, r in zip(got, ref):
        for k, t in l.tensors().items():
            u = r.tensors()[k]
            if exact or k in ("y", "nbt"):
                assert torch.equal(t, u), k
            elif k == "a":
                d = (t.float() - u.float()).abs()
                ulp = torch.maximum(t.float().abs(), u.float().abs()) * 2.0 ** -7 + 1e-6
                assert float((d / ulp).max()) <= 1.01 and float((d > 0).float().mean()) < 0.01, k
            else:
                torch.testing.assert_close(t, u, rtol=2e-5, atol=1e-6, msg=k)
        assert bool((l.a[:, : l.lda - l.Cout] == 7.0).all()), "channels outside the activation view were touched"
        assert not bool(torch.isnan(l.a.float()).any())


STREAM_CASES = [
    # N, H, W, K, [Cout...], act, residual, a extra channels
    (16, 80, 80, 128, [128], 1, False, 0),
    (2, 40, 40, 128, [128], 1, True, 64),        # 64-pixel tiles; output is a channel slice of a concat buffer
    (16, 160, 160, 64, [64], 1, False, 0),       # blocks walk 6+ tiles
    (4, 160, 160, 32, [32], 0, False, 32),       # no activation
    (16, 80, 80, 128, [64, 64], 1, False, 0),    # CSP conv1 + conv2 of one input: two BatchNorms in one launch
    (16, 40, 40, 256, [128, 128], 1, False, 128),
    (16, 20, 20, 512, [256], 1, False, 0),       # two cout tiles per pixel tile
    (8, 80, 80, 64, [256], 1, True, 0),
]


@pytest.mark.parametrize("case", STREAM_CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}_K{c[3]}_Co{'+'.join(map(str, c[4]))}" for c in STREAM_CASES])
def test_stream_conv_bn_one_launch_equals_two(case):
    N, H, W, K, couts, act, with_res, a_extra = case
    g = torch.Generator().manual_seed(K + sum(couts) + H)
    npix = N * H * W
    x = torch.randn(npix, K, generator=g).to(DEV, torch.bfloat16)
    ws = [_pack((torch.randn(c, K, 1, 1, generator=g) / K ** 0.5).to(DEV)) for c in couts]
    ress = [torch.randn(npix, c, generator=g).to(DEV, torch.bfloat16) if with_res else None for c in couts]

    def build():
        return [_Layer(N, H, W, K, 1, c, x, K, 0, w, act, r, a_extra, 11 + i) for i, (c, w, r) in enumerate(zip(couts, ws, ress))]
    _compare(build, L.lib().mi_conv1x1_stream)


WS_CASES = [
    # K, [(N, H, W) per job], act, residual, a extra
    (128, [(2, 80, 80)], 1, False, 0),
    (128, [(16, 40, 40)], 1, True, 128),                          # Bottleneck conv2 + shortcut into a concat slice
    (128, [(2, 20, 20)], 1, False, 0),                            # ragged tiles
    (128, [(4, 80, 80), (4, 40, 40), (4, 20, 20)] * 2, 1, False, 0),   # the head's six jobs: six BatchNorms in one launch
    (64, [(4, 80, 80)], 1, True, 0),
    (64, [(2, 9, 33)], 0, False, 64),
    (32, [(2, 160, 160)], 1, True, 0),
    (32, [(1, 30, 50)], 1, False, 32),
]


@pytest.mark.parametrize("case", WS_CASES, ids=[f"K{c[0]}_{'+'.join('%dx%dx%d' % j for j in c[1][:3])}{'x2' if len(c[1]) > 3 else ''}" for c in WS_CASES])
def test_ws_conv_bn_one_launch_equals_two(case):
    K, jobs, act, with_res, a_extra = case
    g = torch.Generator().manual_seed(K + len(jobs))
    xs = [torch.randn(N * H * W, K, generator=g).to(DEV, torch.bfloat16) for (N, H, W) in jobs]
    ws = [_pack((torch.randn(K, K, 3, 3, generator=g) / (3 * K ** 0.5)).to(DEV)) for _ in jobs]
    ress = [torch.randn(N * H * W, K, generator=g).to(DEV, torch.bfloat16) if with_res else None for (N, H, W) in jobs]

    def build():
        return [_Layer(N, H, W, K, 3, K, x, K, 0, w, act, r, a_extra, 31 + i) for i, ((N, H, W), x, w, r) in enumerate(zip(jobs, xs, ws, ress))]
    _compare(build, L.lib().mi_conv3x3_ws, exact=(K == 128))


def test_conv_bn_plan_declines_what_needs_two_launches():
    """tile-kernel shapes (stride 2, 256-channel 3x3), a BatchNorm job of another tensor, three convs of one input: _plan
    returns 0 (the caller keeps two launches), _fwd fails instead of falling back"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2 * 16 * 16, 256, generator=g).to(DEV, torch.bfloat16)
    w = _pack((torch.randn(256, 256, 3, 3, generator=g) / 48).to(DEV))
    l = _Layer(2, 16, 16, 256, 3, 256, x, 256, 0, w, 1, None, 0, 3)
    meta = L.mi_conv_group()
    j = l.job()
    assert L.lib().mi_conv2d_bn_plan(C.byref(l.desc), C.byref(j), 1, C.byref(meta)) == 0
    assert L.lib().mi_conv2d_bn_fwd(C.byref(l.desc), C.byref(j), 1, sp()) < 0 and b"conv_bn_fwd" in L.lib().mi_last_error()
    x1 = torch.randn(2 * 16 * 16, 64, generator=g).to(DEV, torch.bfloat16)
    w1 = _pack((torch.randn(64, 64, 1, 1, generator=g) / 8).to(DEV))
    l1 = _Layer(2, 16, 16, 64, 1, 64, x1, 64, 0, w1, 1, None, 0, 4)
    j1 = l1.job()
    assert L.lib().mi_conv2d_bn_plan(C.byref(l1.desc), C.byref(j1), 1, C.byref(meta)) == 1
    j1.y = l1.a.data_ptr()       # not the convolution's output
    assert L.lib().mi_conv2d_bn_plan(C.byref(l1.desc), C.byref(j1), 1, C.byref(meta)) == 0


def test_plan_fuses_on_request_and_keeps_the_forward(monkeypatch):
    """MI_CONV_BN_FUSE=1: the YOLOX-s plan merges every CONV / CONV_GROUP that runs on the persistent kernels with the
    BatchNorm pass behind it (41 commands at 16 x 640 x 640, 24 here); the raw head outputs stay what the two-launch plan computes (the
    3x3 32/64-channel layers sum their statistics in another block order: fp32 rounding, amplified by bf16 storage)"""
    import yolov7_d2_amd as M
    from bench import synth_batch_device
    B, S = 2, 320
    outs = {}
    for v in ("0", "1"):
        monkeypatch.setenv("MI_CONV_BN_FUSE", v)
        torch.manual_seed(0)
        model = M.build_model(M.yolox_s_cfg(device="cuda"))
        model.train()
        ps = model.plan_for(B, S, S, True)
        imgs, labels = synth_batch_device(B, S, S, 1234, "cuda")
        ps.image.copy_(imgs)
        ps.labels.copy_(labels)
        ps.plan.run("fwd")
        torch.cuda.synchronize()
        ps.plan.check_bn_barriers()
        nf = sum(bool(getattr(c, "fused_bn", False)) for c in ps.plan.fwd_list)
        outs[v] = (nf, ps.plan.buf_view(ps.preds_buf, torch.float32).clone())
    assert outs["0"][0] == 0 and outs["1"][0] >= 20, (outs["0"][0], outs["1"][0])
    a, b = outs["0"][1], outs["1"][1]
    assert float((a - b).norm() / a.norm()) < 2e-2
