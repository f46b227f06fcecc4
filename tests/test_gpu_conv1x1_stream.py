"""GPU (-m gpu): the streaming 1x1 convolution (csrc/conv1x1_stream.h; 1x1 nn.Conv2d forward / data gradient of BaseConv,
layers/wrappers.py:60-83) through the C-ABI.  Three checks per shape: (1) the raw output is BIT-IDENTICAL to the tile
kernel's (same bf16 operands, same MFMA order over K) and within bf16 tolerance of an fp32 matmul of the same operands;
(2) the BatchNorm (sum, sumsq) accumulators agree with fp64 sums of the stored values; (3) the accumulate mode equals
bf16(bf16(result) + old).  Shapes cover every (K, waves-over-cout, pixel-tile) instantiation, channel-slice views on
both sides, the two-convolutions-of-one-input launch, one-tile blocks and blocks that walk many tiles."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from yolov7_d2_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda"


def sp():
    return L.stream_ptr()


def _pack(w):
    """OIHW fp32 [Cout, K, 1, 1] -> packed forward image [K/8][Cout][8] bf16"""
    Cout, K = w.shape[:2]
    wf = torch.empty(K * Cout, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib().mi_pack_conv_weight(w.data_ptr(), Cout, K, 1, 1, wf.data_ptr(), K, Cout, None, 0, 0, sp()), "pack")
    return wf


def _desc(x, ldx, coff, N, H, W, K, wf, y, ldy, yoff, Cout, stats=None, nslots=0, flags=0):
    d = L.mi_conv_desc()
    d.x = x.data_ptr() + coff * 2
    d.w = wf.data_ptr()
    d.y = y.data_ptr() + yoff * 2
    d.ldx, d.ldy = ldx, ldy
    d.N, d.H, d.W, d.outH, d.outW, d.gridH, d.gridW = N, H, W, H, W, H, W
    d.in_stride = d.out_stride = 1
    d.K8, d.Cout, d.CoutPad, d.ntaps = K // 8, Cout, Cout, 1
    d.flags = flags
    if stats is not None:
        d.stats_acc = stats.data_ptr()
        d.stats_slots = nslots
    return d


CASES = [
    # N, H, W, K, [Cout...], x channel offset / extra, y extra
    (16, 80, 80, 128, [128], 0, 0),        # WM 4, 128-pixel tiles, 800 tiles: ~2 tiles per block
    (2, 40, 40, 128, [128], 128, 64),      # WM 4, 64-pixel tiles (small map), x and y are channel slices
    (16, 160, 160, 64, [64], 0, 0),        # WM 2, 3200 tiles: blocks walk 6+ tiles, ring of 4
    (4, 160, 160, 32, [32], 32, 0),        # WM 1, 256-pixel tiles
    (4, 80, 80, 32, [64], 0, 64),          # K 32 -> 64 (data gradient of a CSP conv1), WM 2
    (16, 40, 40, 256, [256], 0, 0),        # two cout tiles per pixel tile (XCD-ordered grid), ring of 2 or 3
    (16, 20, 20, 512, [128], 0, 0),        # K 512: 128 VGPRs of weights per wave, 64-pixel tiles
    (16, 80, 80, 128, [64, 64], 0, 0),     # CSP conv1 + conv2: two convolutions of one input in one launch
    (16, 40, 40, 256, [128, 128], 256, 0),
    (8, 80, 80, 64, [256], 0, 0),          # K 64 -> 256, two cout tiles
    (16, 20, 20, 256, [512], 0, 0),        # four cout tiles
    (4, 50, 84, 256, [256], 0, 0),         # round 6: 16 800 pixels - a ragged last tile (plain / accumulate only)
    (4, 50, 84, 128, [1024], 128, 0),      # ... and 1024 output channels: two launches of 512 over the same input
]


@pytest.mark.parametrize("mode", ["stats", "plain", "accum"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}_K{c[3]}_Co{'+'.join(map(str, c[4]))}" for c in CASES])
def test_stream_matches_tile_kernel(case, mode, monkeypatch):
    N, H, W, K, couts, xextra, yextra = case
    if mode == "stats" and ((N * H * W) % 64 or sum(couts) > 512):
        pytest.skip("statistics launches need whole pixel tiles and one slice table")
    g = torch.Generator().manual_seed(hash((N, H, K, sum(couts))) & 0xFFFF)
    npix = N * H * W
    ldx = K + xextra
    xbuf = torch.randn(npix, ldx, generator=g).to(DEV, torch.bfloat16)
    ws = [(torch.randn(c, K, 1, 1, generator=g) / K ** 0.5).to(DEV) for c in couts]
    wfs = [_pack(w) for w in ws]
    nslots = 8
    flags = L.MI_CONV_ACCUM if mode == "accum" else 0

    def run(stream):
        ys, sts, descs = [], [], (L.mi_conv_desc * len(couts))()
        for j, (c, wf) in enumerate(zip(couts, wfs)):
            ldy = c + yextra
            gy = torch.Generator().manual_seed(7 + j)
            y = torch.randn(npix, ldy, generator=gy).to(DEV, torch.bfloat16)   # old values (accum) / must be overwritten
            st = torch.zeros(nslots, c, 2, dtype=torch.float64, device=DEV) if mode == "stats" else None
            d = _desc(xbuf, ldx, xextra, N, H, W, K, wf, y, ldy, yextra, c, st, nslots, flags)
            C.memmove(C.byref(descs[j]), C.byref(d), C.sizeof(d))
            ys.append(y)
            sts.append(st)
        if stream:
            L.check(L.lib().mi_conv1x1_stream(descs, len(couts), sp()), "conv1x1_stream")
        else:
            monkeypatch.setenv("MI_CONV_STREAM", "0")
            for j in range(len(couts)):
                L.check(L.lib().mi_conv2d(C.byref(descs[j]), sp()), "conv2d")
            monkeypatch.delenv("MI_CONV_STREAM")
        torch.cuda.synchronize()
        return ys, sts

    ys_s, st_s = run(True)
    ys_t, st_t = run(False)
    xf = xbuf[:, xextra:].float()
    for j, c in enumerate(couts):
        a, b = ys_s[j], ys_t[j]
        assert torch.equal(a[:, :yextra], b[:, :yextra]), "channels outside the output view were touched"
        assert torch.equal(a[:, yextra:], b[:, yextra:]), f"conv {j}: stream output differs from the tile kernel's"
        ref = xf @ ws[j].view(c, K).to(torch.bfloat16).float().t()
        if mode == "accum":
            gy = torch.Generator().manual_seed(7 + j)
            old = torch.randn(npix, c + yextra, generator=gy).to(DEV, torch.bfloat16)[:, yextra:].float()
            ref = ref.to(torch.bfloat16).float() + old
        got = a[:, yextra:].float()
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 1e-2, (j, err)
        if mode == "stats":
            v = a[:, yextra:].double()
            s = st_s[j].sum(0)
            np.testing.assert_allclose(s[:, 0].cpu().numpy(), v.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3 * npix ** 0.5)
            np.testing.assert_allclose(s[:, 1].cpu().numpy(), (v * v).sum(0).cpu().numpy(), rtol=1e-5)
            t = st_t[j].sum(0)
            np.testing.assert_allclose(s.cpu().numpy(), t.cpu().numpy(), rtol=1e-5, atol=1e-3 * npix ** 0.5)


def test_stream_rejects_what_it_cannot_do():
    """the explicit entry fails (it never falls back): 3x3 taps, fp32 outputs, pixel counts that leave a partial tile"""
    x = torch.zeros(2 * 13 * 13, 64, dtype=torch.bfloat16, device=DEV)
    w = _pack(torch.zeros(64, 64, 1, 1, device=DEV))
    y = torch.zeros(2 * 13 * 13, 64, dtype=torch.bfloat16, device=DEV)
    d = _desc(x, 64, 0, 2, 13, 13, 64, w, y, 64, 0, 64)
    assert L.lib().mi_conv1x1_stream(C.byref(d), 1, sp()) < 0     # 338 pixels: a ragged map below MI_C1S_RAGGED_MINPIX
    assert b"conv1x1_stream" in L.lib().mi_last_error()
    st = torch.zeros(8, 64, 2, dtype=torch.float64, device=DEV)
    xr = torch.zeros(9797, 64, dtype=torch.bfloat16, device=DEV)
    yr = torch.zeros(9797, 64, dtype=torch.bfloat16, device=DEV)
    d = _desc(xr, 64, 0, 1, 97, 101, 64, w, yr, 64, 0, 64, st, 8)
    assert L.lib().mi_conv1x1_stream(C.byref(d), 1, sp()) < 0     # statistics over a ragged map: not served
    d = _desc(x, 64, 0, 1, 16, 16, 64, w, y, 64, 0, 64, flags=L.MI_CONV_OUT_F32)
    assert L.lib().mi_conv1x1_stream(C.byref(d), 1, sp()) < 0
    # ... and mi_conv2d itself still serves such shapes on the tile kernel
    d = _desc(x, 64, 0, 2, 13, 13, 64, w, y, 64, 0, 64)
    L.check(L.lib().mi_conv2d(C.byref(d), sp()), "conv2d")
    torch.cuda.synchronize()


EPI_CASES = [
    # N, H, W, K, Cout, x extra, y extra, aux extra
    (4, 40, 56, 64, 256, 0, 0, 0),        # ResNet res2 conv3 / shortcut (K 64 -> 256), two cout tiles
    (4, 40, 56, 256, 64, 0, 0, 0),        # res2 conv1 (256 -> 64)
    (2, 40, 64, 128, 512, 128, 64, 32),   # res3 conv3 on channel-slice views, four cout tiles
    (2, 40, 64, 512, 128, 0, 0, 0),       # res3 conv1 / data gradient of conv3 (K 512: 64-pixel tiles)
    (2, 32, 32, 32, 32, 0, 0, 0),         # WM 1
    (4, 50, 84, 256, 1024, 0, 0, 0),      # res4 conv3 at 800 x 1333: 16 800 pixels (ragged last tile), 1024 channels = two launches
    (4, 50, 84, 512, 256, 0, 0, 0),       # ragged, K 512
    (1, 97, 101, 64, 2048, 64, 32, 0),    # 9 797 pixels, 2048 channels = four launches, slice views
]


@pytest.mark.parametrize("mode", ["bias", "bias_relu", "relu", "addrelu", "addrelu_nobias", "relumask", "accum_relumask"])
@pytest.mark.parametrize("case", EPI_CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}_K{c[3]}_Co{c[4]}" for c in EPI_CASES])
def test_stream_epilogues_match_tile_kernel(case, mode, monkeypatch):
    """MODE 4 / 5 (round 6): fp32 bias, ReLU, residual-add + ReLU (MI_CONV_ADDRELU) and the ReLU mask of a data gradient
    (MI_CONV_RELUMASK) in the streaming kernel's epilogue - the detectron2 Conv2d + FrozenBatchNorm2d (+ ReLU / + shortcut)
    layers of the ResNet bottlenecks (modeling/resnet.py) with K <= 512, Cout <= 512 - bit-identical to the tile kernel's
    epilogues (same MFMA order over K, same roundings) and within bf16 tolerance of fp32 torch"""
    N, H, W, K, Cout, xe, ye, ae = case
    g = torch.Generator().manual_seed(hash((N, H, K, Cout)) & 0xFFFF)
    npix = N * H * W
    xbuf = torch.randn(npix, K + xe, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(Cout, K, 1, 1, generator=g) / K ** 0.5).to(DEV)
    wf = _pack(w)
    bias = torch.randn(Cout, generator=g).to(DEV) if "nobias" not in mode and mode not in ("relu", "relumask", "accum_relumask") else None
    aux = torch.randn(npix, Cout + ae, generator=g).to(DEV, torch.bfloat16) if mode in ("addrelu", "addrelu_nobias", "relumask", "accum_relumask") else None
    flags = {"bias": 0, "bias_relu": L.MI_CONV_RELU, "relu": L.MI_CONV_RELU, "addrelu": L.MI_CONV_ADDRELU,
             "addrelu_nobias": L.MI_CONV_ADDRELU, "relumask": L.MI_CONV_RELUMASK,
             "accum_relumask": L.MI_CONV_ACCUM | L.MI_CONV_RELUMASK}[mode]      # MODE 6: y = mask(bf16(bf16(conv) + y_old))

    def run(stream):
        y = torch.full((npix, Cout + ye), 3.0, dtype=torch.bfloat16, device=DEV)
        d = _desc(xbuf, K + xe, xe, N, H, W, K, wf, y, Cout + ye, ye, Cout, flags=flags)
        if bias is not None:
            d.bias = bias.data_ptr()
        if aux is not None:
            d.bn_y, d.bn_ldy = aux.data_ptr() + ae * 2, Cout + ae
        if stream:
            L.check(L.lib().mi_conv1x1_stream(C.byref(d), 1, sp()), "conv1x1_stream")
        else:
            monkeypatch.setenv("MI_CONV_STREAM", "0")
            L.check(L.lib().mi_conv2d(C.byref(d), sp()), "conv2d")
            monkeypatch.delenv("MI_CONV_STREAM")
        torch.cuda.synchronize()
        return y

    a, b = run(True), run(False)
    assert torch.equal(a[:, :ye], b[:, :ye]) and torch.all(a[:, :ye] == 3.0)
    assert torch.equal(a[:, ye:], b[:, ye:]), float((a[:, ye:].float() - b[:, ye:].float()).abs().max())
    ref = xbuf[:, xe:].float() @ w.view(Cout, K).to(torch.bfloat16).float().t()
    if bias is not None:
        ref = ref + bias
    if flags & L.MI_CONV_RELU:
        ref = ref.clamp(min=0)
    if flags & L.MI_CONV_ADDRELU:
        ref = (ref.to(torch.bfloat16).float() + aux[:, ae:].float()).clamp(min=0)
    if flags & L.MI_CONV_ACCUM:
        ref = ref.to(torch.bfloat16).float() + 3.0                     # (the old values: y starts at 3.0 everywhere)
    if flags & L.MI_CONV_RELUMASK:
        ref = ref * (aux[:, ae:].float() > 0)
    err = float((a[:, ye:].float() - ref).abs().max() / ref.abs().max())
    assert err < 1e-2, err
    # and mi_conv2d routes such a descriptor to the streaming kernel by itself (MI_CONV_STREAM_EPI=0 would keep the tile kernel)
    y2 = torch.full((npix, Cout + ye), 3.0, dtype=torch.bfloat16, device=DEV)
    d = _desc(xbuf, K + xe, xe, N, H, W, K, wf, y2, Cout + ye, ye, Cout, flags=flags)
    if bias is not None:
        d.bias = bias.data_ptr()
    if aux is not None:
        d.bn_y, d.bn_ldy = aux.data_ptr() + ae * 2, Cout + ae
    L.check(L.lib().mi_conv2d(C.byref(d), sp()), "conv2d (routed)")
    torch.cuda.synchronize()
    assert torch.equal(y2, a)
