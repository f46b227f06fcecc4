"""GPU (-m gpu): the bias gradient written by the weight-gradient launches themselves (mi_wgrad_desc.gbias; csrc/conv_wgrad.hip
WgBias) against the separate column-sum launch it replaces and against torch: nn.Linear rows of the transformer
(detr_backbone.py:160-170), biased nn.Conv2d 1x1 / 3x3 of the SparseInst encoder / decoder (transcoders/encoder_sparseinst.py:
60-90).  The weight gradient must not change at all; the bias gradient is a sum in another (fixed) order."""
import pytest
import torch

from yolov7_d2_amd.modeling.transformer import _LinearFn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lin(T, Cin, Cout, seed, relu):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, Cin, generator=g).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.1
    dy = torch.randn(T, Cout, generator=g).to(torch.bfloat16)
    return x, w, b, dy


@pytest.mark.parametrize("T,Cin,Cout,relu", [(4200, 256, 256, False), (4200, 256, 2048, True), (400, 2048, 256, False), (200, 256, 96, False),
                                            (1050, 256, 768, False), (37, 64, 32, False)])
def test_linear_bias_gradient_from_the_wgrad_launches(monkeypatch, T, Cin, Cout, relu):
    x, w, b, dy = _lin(T, Cin, Cout, T + Cout, relu)
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("MI_WGRAD_BIAS", mode)
        xd = x.to(DEV).requires_grad_(True)
        wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        y = _LinearFn.apply(xd, wd, bd, relu)
        y.backward(dy.to(DEV))
        torch.cuda.synchronize()
        res[mode] = (y.detach().float().cpu(), xd.grad.float().cpu(), wd.grad.cpu(), bd.grad.cpu())
    assert torch.equal(res["0"][0], res["2"][0]) and torch.equal(res["0"][1], res["2"][1])
    assert torch.equal(res["0"][2], res["2"][2])                       # the weight gradient is untouched
    torch.testing.assert_close(res["2"][3], res["0"][3], rtol=2e-5, atol=2e-5)
    # against torch on the same bf16-rounded operands
    xr, wr, br = x.float().requires_grad_(True), w.to(torch.bfloat16).float().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, wr, br)
    if relu:
        yr = yr.relu()
        mask = (res["2"][0] > 0).float()       # (the ReLU gate of the bf16 output decides, as in the product)
        yr.backward(dy.float() * 1.0)
    else:
        yr.backward(dy.float())
    if not relu:
        torch.testing.assert_close(res["2"][3], br.grad, rtol=2e-3, atol=2e-2)
        torch.testing.assert_close(res["2"][2], wr.grad, rtol=2e-2, atol=2e-2 * float(wr.grad.abs().max()))


@pytest.mark.parametrize("k,Cin,Cout,H,W,N", [(1, 256, 256, 40, 40, 4), (3, 256, 128, 20, 24, 2), (3, 64, 64, 33, 17, 3), (1, 2048, 256, 20, 20, 2)])
def test_conv_bias_gradient_from_the_wgrad_launches(monkeypatch, k, Cin, Cout, H, W, N):
    import yolov7_d2_amd.ops  # noqa: F401  (registers torch.ops.mi355)
    g = torch.Generator().manual_seed(k * 100 + Cin)
    x = torch.randn(N, Cin, H, W, generator=g).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.03
    b = torch.randn(Cout, generator=g) * 0.1
    dy = torch.randn(N, Cout, H, W, generator=g).to(torch.bfloat16)
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("MI_WGRAD_BIAS", mode)
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        y = torch.ops.mi355.conv2d(xd, wd, bd, 1, (k - 1) // 2)
        y.backward(dy.to(DEV).contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
        res[mode] = (xd.grad.float().cpu(), wd.grad.cpu(), bd.grad.cpu())
    assert torch.equal(res["0"][0], res["2"][0]) and torch.equal(res["0"][1], res["2"][1])
    torch.testing.assert_close(res["2"][2], res["0"][2], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(res["2"][2], dy.float().sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
