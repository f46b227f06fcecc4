"""GPU (-m gpu): the weight-stationary 3x3 convolution (csrc/conv3x3_ws.h; Bottleneck conv2 and the YOLOXHead towers,
layers/wrappers.py:105-123, head/yolox_head.py:73-102) through the C-ABI against (1) an fp32 conv2d of the same bf16
operands (bf16 output tolerance) and (2) the tile kernel (MI_CONV_WS=0): the two sum the K * 9 products in a different
order, so they agree to fp32 rounding - at most one bf16 ulp on a small fraction of the outputs.  Forward and data-gradient
tap orders, BatchNorm accumulators vs fp64 sums of the stored values, the accumulate mode, ragged maps (partial tiles
are not stored and stay out of the statistics), channel-slice views, the six-job head launch."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from yolov7_d2_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda"


def sp():
    return L.stream_ptr()


def _pack(w, dgrad):
    """OIHW fp32 [K, K, 3, 3] -> packed image: forward [tap][Cin/8][Cout][8] or data-gradient [tap][Cout/8][Cin][8]"""
    K = w.shape[0]
    img = torch.empty(9 * K * K, dtype=torch.bfloat16, device=DEV)
    if dgrad:
        L.check(L.lib().mi_pack_conv_weight(w.data_ptr(), K, K, 3, 3, None, 0, 0, img.data_ptr(), K, K, sp()), "pack")
    else:
        L.check(L.lib().mi_pack_conv_weight(w.data_ptr(), K, K, 3, 3, img.data_ptr(), K, K, None, 0, 0, sp()), "pack")
    return img


def _desc(x, ldx, xoff, N, H, W, K, wimg, y, ldy, yoff, dgrad, stats=None, flags=0):
    d = L.mi_conv_desc()
    d.x, d.w, d.y = x.data_ptr() + xoff * 2, wimg.data_ptr(), y.data_ptr() + yoff * 2
    d.ldx, d.ldy = ldx, ldy
    d.N, d.H, d.W, d.outH, d.outW, d.gridH, d.gridW = N, H, W, H, W, H, W
    d.in_stride = d.out_stride = 1
    d.K8, d.Cout, d.CoutPad, d.ntaps = K // 8, K, K, 9
    t = 0
    for r in range(3):
        for s in range(3):
            # forward: input pixel = output + (r - 1, s - 1); data gradient: output-gradient pixel = input + (1 - r, 1 - s)
            d.tap_dy[t], d.tap_dx[t], d.tap_w[t] = ((1 - r, 1 - s) if dgrad else (r - 1, s - 1)) + (r * 3 + s,)
            t += 1
    d.flags = flags
    if stats is not None:
        d.stats_acc, d.stats_slots = stats.data_ptr(), 16
    return d


def _reference(x, w, dgrad):
    """x [N, H, W, K] bf16, w OIHW fp32 -> fp32 NHWC result of the same bf16 operands"""
    xn = x.float().permute(0, 3, 1, 2)
    wb = w.to(torch.bfloat16).float()
    if dgrad:
        wb = wb.transpose(0, 1).flip(2, 3)
    return F.conv2d(xn, wb, None, 1, 1).permute(0, 2, 3, 1).contiguous()


CASES = [
    # K, [(N, H, W) per job], x extra channels, y extra channels
    (128, [(2, 80, 80)], 0, 0),
    (128, [(16, 40, 40)], 128, 0),            # 240 tiles on 240 blocks; x is the upper half of a 256-channel buffer
    (128, [(2, 20, 20)], 0, 64),              # ragged: 24 x 32 tiles over a 20 x 20 map
    (128, [(1, 13, 17)], 0, 0),
    (128, [(4, 80, 80), (4, 40, 40), (4, 20, 20)] * 2, 0, 0),   # the head's six jobs in one launch
    (64, [(4, 80, 80)], 0, 0),
    (64, [(2, 9, 33)], 64, 0),
    (32, [(2, 160, 160)], 0, 0),
    (32, [(1, 30, 50)], 0, 32),
]


@pytest.mark.parametrize("mode", ["stats", "plain", "accum"])
@pytest.mark.parametrize("dgrad", [False, True], ids=["fwd", "dgrad"])
@pytest.mark.parametrize("case", CASES, ids=[f"K{c[0]}_{'+'.join('%dx%dx%d' % j for j in c[1][:3])}{'x2' if len(c[1]) > 3 else ''}" for c in CASES])
def test_ws_matches_reference_and_tile_kernel(case, dgrad, mode, monkeypatch):
    K, jobs, xextra, yextra = case
    if mode == "stats" and dgrad:
        pytest.skip("data gradients take no statistics")
    g = torch.Generator().manual_seed(K + len(jobs) + 7 * dgrad)
    ws = [(torch.randn(K, K, 3, 3, generator=g) / (3 * K ** 0.5)).to(DEV) for _ in jobs]
    imgs = [_pack(w, dgrad) for w in ws]
    xs = [torch.randn(N, H, W, K + xextra, generator=g).to(DEV, torch.bfloat16) for (N, H, W) in jobs]
    flags = L.MI_CONV_ACCUM if mode == "accum" else 0

    def run(use_ws):
        ys, sts = [], []
        descs = (L.mi_conv_desc * len(jobs))()
        for j, (N, H, W) in enumerate(jobs):
            gy = torch.Generator().manual_seed(100 + j)
            y = torch.randn(N, H, W, K + yextra, generator=gy).to(DEV, torch.bfloat16)
            st = torch.zeros(16, K, 2, dtype=torch.float64, device=DEV) if mode == "stats" else None
            d = _desc(xs[j], K + xextra, xextra, N, H, W, K, imgs[j], y, K + yextra, yextra, dgrad, st, flags)
            C.memmove(C.byref(descs[j]), C.byref(d), C.sizeof(d))
            ys.append(y)
            sts.append(st)
        if use_ws:
            L.check(L.lib().mi_conv3x3_ws(descs, len(jobs), sp()), "conv3x3_ws")
        else:
            monkeypatch.setenv("MI_CONV_WS", "0")
            for j in range(len(jobs)):
                L.check(L.lib().mi_conv2d(C.byref(descs[j]), sp()), "conv2d")
            monkeypatch.delenv("MI_CONV_WS")
        torch.cuda.synchronize()
        return ys, sts

    ys_w, st_w = run(True)
    ys_t, _ = run(False)
    for j, (N, H, W) in enumerate(jobs):
        a, b = ys_w[j], ys_t[j]
        assert torch.equal(a[..., :yextra], b[..., :yextra]), "channels outside the output view were touched"
        ref = _reference(xs[j][..., xextra:], ws[j], dgrad).cpu()
        if mode == "accum":
            gy = torch.Generator().manual_seed(100 + j)
            old = torch.randn(N, H, W, K + yextra, generator=gy).to(torch.bfloat16)[..., yextra:].float()
            ref = ref.to(torch.bfloat16).float() + old
        got = a[..., yextra:].float().cpu()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
        # against the tile kernel: same operands, another summation order -> bf16 neighbours at most, and only a few
        tile = b[..., yextra:].float().cpu()
        diff = (got - tile).abs()
        # one bf16 ulp of the larger value; outputs near zero (cancellation) differ by the fp32 summation error of the
        # K * 9 products instead: a small fraction of the output scale
        # (accumulate mode rounds twice - bf16(bf16(result) + old): the ulp that matters is that of |result| <= |sum| + |old|)
        mag = torch.maximum(got.abs(), tile.abs()) + (old.abs() if mode == "accum" else 0.0)
        ulp = mag * 2.0 ** -7 + 2e-3 * float(tile.abs().mean())
        assert float((diff / ulp).max()) <= (1.5 if mode == "accum" else 1.01), float((diff / ulp).max())
        assert float((diff > 0).float().mean()) < 0.02
        if mode == "stats":
            v = a[..., yextra:].double().reshape(-1, K)
            s = st_w[j].sum(0)
            np.testing.assert_allclose(s[:, 0].cpu().numpy(), v.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3 * v.shape[0] ** 0.5)
            np.testing.assert_allclose(s[:, 1].cpu().numpy(), (v * v).sum(0).cpu().numpy(), rtol=1e-5)


def test_ws_rejects_what_it_cannot_do():
    x = torch.zeros(2, 16, 16, 256, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(9 * 256 * 256, dtype=torch.bfloat16, device=DEV)
    y = torch.zeros(2, 16, 16, 256, dtype=torch.bfloat16, device=DEV)
    d = _desc(x, 256, 0, 2, 16, 16, 256, w, y, 256, 0, False)       # 256 channels: 576 VGPRs of weights per wave
    assert L.lib().mi_conv3x3_ws(C.byref(d), 1, sp()) < 0 and b"conv3x3_ws" in L.lib().mi_last_error()
    L.check(L.lib().mi_conv2d(C.byref(d), sp()), "conv2d")         # ... the tile kernel serves it
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------- stride 2 (the down-sampling convs)
def _desc_s2(x, N, H, W, K, wimg, y, Cout, stats=None):
    d = L.mi_conv_desc()
    d.x, d.w, d.y = x.data_ptr(), wimg.data_ptr(), y.data_ptr()
    d.ldx, d.ldy = K, Cout
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    d.N, d.H, d.W, d.outH, d.outW, d.gridH, d.gridW = N, H, W, Ho, Wo, Ho, Wo
    d.in_stride, d.out_stride = 2, 1
    d.K8, d.Cout, d.CoutPad, d.ntaps = K // 8, Cout, Cout, 9
    t = 0
    for r in range(3):
        for s_ in range(3):
            d.tap_dy[t], d.tap_dx[t], d.tap_w[t] = r - 1, s_ - 1, r * 3 + s_
            t += 1
    if stats is not None:
        d.stats_acc, d.stats_slots = stats.data_ptr(), 16
    return d


S2_CASES = [
    # K, Cout, N, H, W
    (32, 64, 16, 320, 320),      # dark2.0
    (64, 128, 16, 160, 160),     # dark3.0
    (128, 256, 16, 80, 80),      # dark4.0: two 128-channel blocks over one input
    (128, 128, 16, 80, 80),      # bu_conv2
    (128, 128, 2, 37, 45),       # odd input, ragged tiles
    (64, 128, 1, 9, 34),
    (32, 64, 3, 50, 31),
]


@pytest.mark.parametrize("with_stats", [True, False], ids=["stats", "plain"])
@pytest.mark.parametrize("case", S2_CASES, ids=["K%d_Co%d_%dx%dx%d" % c for c in S2_CASES])
def test_ws_stride2_matches_reference_and_tile_kernel(case, with_stats, monkeypatch):
    """the stride-2 form (BaseConv(k=3, s=2) of CSPDarknet's dark2-4 / PAFPN's bu_conv2: darknetx.py:113-160,
    yolo_pafpn.py:60-77): de-interleaved halo columns, K -> 2 K as two 128-channel jobs of one input"""
    K, Cout, N, H, W = case
    g = torch.Generator().manual_seed(K + Cout + H)
    w = (torch.randn(Cout, K, 3, 3, generator=g) / (3 * K ** 0.5)).to(DEV)
    img = torch.empty(9 * K * Cout, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib().mi_pack_conv_weight(w.data_ptr(), Cout, K, 3, 3, img.data_ptr(), K, Cout, None, 0, 0, sp()), "pack")
    x = torch.randn(N, H, W, K, generator=g).to(DEV, torch.bfloat16)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1

    def run(ws_on):
        y = torch.full((N, Ho, Wo, Cout), 5.0, dtype=torch.bfloat16, device=DEV)
        st = torch.zeros(16, Cout, 2, dtype=torch.float64, device=DEV) if with_stats else None
        d = _desc_s2(x, N, H, W, K, img, y, Cout, st)
        if ws_on:
            assert L.lib().mi_conv2d_route(C.byref(d)) == 2
            L.check(L.lib().mi_conv3x3_ws(C.byref(d), 1, sp()), "conv3x3_ws")
        else:
            monkeypatch.setenv("MI_CONV_WS", "0")
            L.check(L.lib().mi_conv2d(C.byref(d), sp()), "conv2d")
            monkeypatch.delenv("MI_CONV_WS")
        torch.cuda.synchronize()
        return y, st
    yw, sw = run(True)
    yt, stt = run(False)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), None, 2, 1).permute(0, 2, 3, 1).cpu()
    got, tile = yw.float().cpu(), yt.float().cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
    diff = (got - tile).abs()
    ulp = torch.maximum(got.abs(), tile.abs()) * 2.0 ** -7 + 2e-3 * float(tile.abs().mean())
    assert float((diff / ulp).max()) <= 1.01 and float((diff > 0).float().mean()) < 0.02
    if with_stats:
        v = yw.double().reshape(-1, Cout)
        s = sw.sum(0)
        np.testing.assert_allclose(s[:, 0].cpu().numpy(), v.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3 * v.shape[0] ** 0.5)
        np.testing.assert_allclose(s[:, 1].cpu().numpy(), (v * v).sum(0).cpu().numpy(), rtol=1e-5)
