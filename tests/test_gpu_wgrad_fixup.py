"""GPU (-m gpu): the split-K reduction INSIDE the grouped weight-gradient launches (MI_WG_FIXUP=1; csrc/conv_wgrad.hip wg_fixup)
against the reduce grid it replaces, through the C-ABI (mi_conv2d_wgrad_group_plan / _run): conv wgrad of BaseConv
(layers/wrappers.py:60-83) for the layer shapes of the YOLOX-s step.  The fix-up adds the splits in the reduce kernel's own
order, so the two forms must agree BIT FOR BIT - also on a second and third run of the same table (the tile counters re-arm
themselves), with `accumulate` and with a folded per-Cout scale."""
import ctypes as C

import pytest
import torch

from yolov7_d2_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rup(a, b):
    return (a + b - 1) // b * b


def _layer(k, stride, Cin, Cout, N, H, W, seed, accumulate=0, scaled=False):
    g = torch.Generator().manual_seed(seed)
    CinPad = _rup(Cin, 16) if k > 1 else _rup(Cin, 32)
    CoutPad = _rup(Cout, 32)
    pad = k // 2
    oH, oW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.zeros(N, H, W, CinPad, dtype=torch.bfloat16)
    x[..., :Cin] = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16)
    dy = torch.zeros(N, oH, oW, CoutPad, dtype=torch.bfloat16)
    dy[..., :Cout] = (torch.randn(N, oH, oW, Cout, generator=g) * 0.1).to(torch.bfloat16)
    g0 = torch.randn(Cout, Cin, k, k, generator=g)
    rs = (0.5 + torch.rand(Cout, generator=g)) if scaled else None
    return dict(k=k, stride=stride, Cin=Cin, Cout=Cout, CinPad=CinPad, CoutPad=CoutPad, N=N, H=H, W=W, oH=oH, oW=oW,
                x=x.to(DEV), dy=dy.to(DEV), g0=g0, accumulate=accumulate, rs=None if rs is None else rs.to(DEV))


def _descs(layers, gws):
    descs = (L.mi_wgrad_desc * len(layers))()
    for d, l, gw in zip(descs, layers, gws):
        d.x, d.dy, d.gw = l["x"].data_ptr(), l["dy"].data_ptr(), gw.data_ptr()
        d.ldx, d.ldy = l["CinPad"], l["CoutPad"]
        d.N, d.H, d.W, d.outH, d.outW, d.stride = l["N"], l["H"], l["W"], l["oH"], l["oW"], l["stride"]
        d.Cin, d.Cout, d.CinPad, d.CoutPad = l["Cin"], l["Cout"], l["CinPad"], l["CoutPad"]
        k, pad = l["k"], l["k"] // 2
        d.ntaps = k * k
        for t, (r, s) in enumerate((r, s) for r in range(k) for s in range(k)):
            d.tap_dy[t], d.tap_dx[t] = r - pad, s - pad
        d.accumulate = l["accumulate"]
        d.row_scale = l["rs"].data_ptr() if l["rs"] is not None else None
    return descs


def _run_group(layers, fixup, monkeypatch, runs=1, multi="0"):
    """plan + upload + run the grouped launch `runs` times; returns the gradients after every run and the launch meta"""
    lib = L.lib()
    monkeypatch.setenv("MI_WG_FIXUP", "1" if fixup else "0")
    monkeypatch.setenv("MI_WG_MULTI", multi)  # (default "0": the fix-up form keeps one grid per tile configuration - like with like, bit for bit)
    gws = [l["g0"].clone().to(DEV) for l in layers]
    descs = _descs(layers, gws)
    meta = L.mi_wgrad_group()
    L.check(lib.mi_conv2d_wgrad_group_plan(descs, len(layers), None, None, 0, C.byref(meta)), "plan (sizes)")
    ws = torch.empty(int(meta.ws_bytes) + 256, dtype=torch.uint8, device=DEV)
    ws.fill_(0xFF)                                                   # stale partials must never be read as sums
    host = (C.c_char * int(meta.table_bytes))()
    L.check(lib.mi_conv2d_wgrad_group_plan(descs, len(layers), ws.data_ptr(), host, meta.table_bytes, C.byref(meta)), "plan")
    table = torch.frombuffer(bytearray(bytes(host)[:int(meta.table_bytes)]), dtype=torch.uint8).to(DEV)
    outs = []
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(runs):
        L.check(lib.mi_conv2d_wgrad_group_run(C.byref(meta), table.data_ptr(), st), "run")
        torch.cuda.synchronize()
        outs.append([g.cpu().clone() for g in gws])
    return outs, meta


YOLOX_LAYERS = [
    # k, stride, Cin, Cout, N, H, W                (CSPDarknet / PAFPN / head shapes of the 640 x 640 step, batch 4)
    (3, 1, 64, 64, 4, 80, 80),       # dark3 bottleneck conv2: the 64 x 64 x 9 tile configuration
    (3, 1, 32, 32, 4, 160, 160),     # dark2 bottleneck conv2
    (3, 2, 32, 64, 4, 160, 160),     # dark2.0 (stride 2)
    (3, 2, 128, 256, 4, 40, 40),     # dark4.0
    (3, 1, 16, 32, 4, 96, 96),       # stem (Focus output: 12 -> 16 channels)
    (1, 1, 128, 128, 4, 80, 80),     # CSP conv1
    (1, 1, 512, 256, 4, 20, 20),     # lateral conv
    (1, 1, 256, 512, 4, 20, 20),
    (1, 1, 64, 32, 4, 160, 160),
    (3, 1, 128, 128, 4, 40, 40),     # head tower
    (1, 1, 128, 85, 4, 40, 40),      # prediction convs (Cout 85 -> pad 96; reads a map with padded channels)
]


def test_fixup_equals_the_reduce_grid_bit_for_bit(monkeypatch):
    layers = [_layer(*s, seed=100 + i) for i, s in enumerate(YOLOX_LAYERS)]
    ref, meta0 = _run_group(layers, False, monkeypatch)
    got, meta1 = _run_group(layers, True, monkeypatch, runs=3)
    assert meta0.red_blocks + meta0.red9_blocks > 0 and all(meta0.g[i].fixup == 0 for i in range(meta0.ngroups))
    assert meta1.red_blocks == 0 and meta1.red9_blocks == 0 and all(meta1.g[i].fixup == 1 for i in range(meta1.ngroups))
    assert [meta1.g[i].nblocks for i in range(meta1.ngroups)] == [meta0.g[i].nblocks for i in range(meta0.ngroups)]
    for r, run in enumerate(got):                       # runs 2 and 3: the counters re-armed themselves
        for l, a, b in zip(YOLOX_LAYERS, ref[0], run):
            assert torch.isfinite(b).all(), (l, r)
            assert torch.equal(a, b), (l, r, float((a - b).abs().max()))
    # and against torch on the same bf16 operands (the reduce grid's own check, repeated for the new form)
    for l, gw in zip(layers, got[0]):
        xr = l["x"][..., :l["Cin"]].float().permute(0, 3, 1, 2).cpu()
        dyr = l["dy"][..., :l["Cout"]].float().permute(0, 3, 1, 2).cpu()
        w = torch.zeros(l["Cout"], l["Cin"], l["k"], l["k"], requires_grad=True)
        torch.nn.functional.conv2d(xr, w, stride=l["stride"], padding=l["k"] // 2).backward(dyr)
        err = float((gw - w.grad).norm() / w.grad.norm())
        assert err < 2e-3, (l["k"], l["Cin"], l["Cout"], err)


def test_fixup_accumulate_and_row_scale(monkeypatch):
    """`accumulate` (gw += dW, the shared-weight case) and `row_scale` (gradient of a weight whose packed image carried a
    folded per-Cout factor) go through the same epilogue in both forms; every run adds once more"""
    specs = [(3, 1, 64, 64, 2, 40, 40), (1, 1, 128, 256, 2, 40, 40), (3, 2, 64, 128, 2, 40, 40), (1, 1, 256, 85, 2, 20, 20)]
    layers = [_layer(*s, seed=300 + i, accumulate=i % 2, scaled=i >= 1) for i, s in enumerate(specs)]
    ref, _ = _run_group(layers, False, monkeypatch, runs=2)
    got, _ = _run_group(layers, True, monkeypatch, runs=2)
    for r in range(2):
        for s, a, b in zip(specs, ref[r], got[r]):
            assert torch.equal(a, b), (s, r)
    assert not torch.equal(ref[0][1], ref[1][1])       # (layer 1 accumulates: its second run differs from its first)


def test_fixup_single_split_and_tiny_layers(monkeypatch):
    """layers so small that a tile has ONE split (the block is its own last arriver) or fewer float4s than splits"""
    specs = [(1, 1, 32, 32, 1, 8, 8), (3, 1, 16, 32, 1, 6, 10), (1, 1, 64, 64, 2, 16, 16), (3, 1, 64, 64, 1, 12, 12)]
    layers = [_layer(*s, seed=500 + i) for i, s in enumerate(specs)]
    ref, _ = _run_group(layers, False, monkeypatch)
    got, _ = _run_group(layers, True, monkeypatch, runs=2)
    for r in range(2):
        for s, a, b in zip(specs, ref[0], got[r]):
            assert torch.equal(a, b), (s, r)


def test_mixed_grid_equals_one_grid_per_configuration(monkeypatch):
    """MI_WG_MULTI (round 6: the 1x1 layers of a plan in ONE grid whatever their tile widths, wgrad2_multi_kernel; 4: the 3x3
    layers' 256-thread configurations too) against one grid per tile configuration: fewer groups, the same gradients up to
    the split-K summation order (the pooled slot count changes the splits), and both equal to the fp64 reference"""
    layers = [_layer(*s, seed=300 + i) for i, s in enumerate(YOLOX_LAYERS)]
    ref, meta0 = _run_group(layers, False, monkeypatch, multi="0")
    for mode in ("3", "4"):                             # (3: forced for this 11-layer plan; the default needs >= 16 1x1 layers)
        got, meta1 = _run_group(layers, False, monkeypatch, multi=mode)
        assert meta1.ngroups < meta0.ngroups
        assert any(meta1.g[i].cfg[0] == 1 and meta1.g[i].cfg[1] == 0 for i in range(meta1.ngroups))
        for l, s, a, b in zip(layers, YOLOX_LAYERS, ref[0], got[0]):
            xr = l["x"][..., :l["Cin"]].float().permute(0, 3, 1, 2).cpu()
            dyr = l["dy"][..., :l["Cout"]].float().permute(0, 3, 1, 2).cpu()
            w = torch.zeros(l["Cout"], l["Cin"], l["k"], l["k"], requires_grad=True)
            torch.nn.functional.conv2d(xr, w, stride=l["stride"], padding=l["k"] // 2).backward(dyr)
            assert float((b - w.grad).norm() / w.grad.norm()) < 2e-3, (s, mode)
            assert float((a - b).abs().max()) <= 1e-5 * float(w.grad.abs().max()) + 1e-7, (s, mode)   # (fp32 sums in another order)


def test_fixup_in_the_captured_step(monkeypatch):
    """the whole YOLOX-s plan with MI_WG_FIXUP=1: backward replayed twice from a captured hipGraph gives the gradients of the
    eager command list both times (the counters live in the job table the graph's kernels point at)"""
    import yolox_oracle as O
    import yolov7_d2_amd as M
    monkeypatch.setenv("MI_WG_FIXUP", "1")
    model = M.build_model(M.yolox_s_cfg(device=DEV))
    model.load_state_dict(O.init_state_dict(0.33, 0.5, 80, seed=4))
    model.train()
    imgs, labels = O.synth_batch(2, 96, 128, seed=19, max_gt=4)
    ps = model.plan_for(2, 96, 128, True)
    assert ps.plan.bwd_tags[-1] == "wgrad_group"
    ps.image.copy_(imgs.to(DEV)); ps.labels.copy_(labels.to(DEV))
    ps.gw().fill_(1.0)
    ps.plan.run("fwd"); ps.plan.run("bwd"); torch.cuda.synchronize()
    g_eager = model.params.grad.detach().cpu().clone()
    assert torch.isfinite(g_eager).all() and float(g_eager.abs().max()) > 0
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ps.plan.capture("bwd", s)
        for _ in range(2):
            model.params.grad.zero_()
            ps.plan.launch("bwd", s)
            s.synchronize()
            g = model.params.grad.detach().cpu()
            assert torch.isfinite(g).all()
            assert float((g - g_eager).norm() / g_eager.norm()) < 1e-3     # (BatchNorm's fp64 atomics may reorder)
