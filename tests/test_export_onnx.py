"""ONNX export of YOLOX (export.py:237-303) in an image with neither `onnx` nor `onnxruntime`:

* the test reader / interpreter (tests/onnx_interp.py) is first checked against a file TORCH's own exporter writes for a small
  module with the same operator kinds (Conv, SiLU, MaxPool, nearest Resize, strided Slice, Concat) - that pins the protobuf
  field numbers the exporter uses and the interpreter's operator semantics to torch's;
* then the exported YOLOX-s graph (weights of the reference's golden, 64 x 96) is executed by that interpreter and compared
  with the REFERENCE's own export-mode output (`yolox_s_onnx_layout_64x96.npz`: head.onnx_export = True) in fp32."""
import io
import os
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import onnx_interp as OI  # noqa: E402


class _Small(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c = torch.nn.Conv2d(12, 8, 3, padding=1)
        self.p = torch.nn.MaxPool2d(5, 1, 2)
        self.u = torch.nn.Upsample(scale_factor=2, mode="nearest")

    def forward(self, x):
        x = x.permute(0, 3, 1, 2)
        x = torch.cat((x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]), 1)
        y = torch.nn.functional.silu(self.c(x))
        y = torch.cat([y, self.p(y)], 1)
        return self.u(y)


def _torch_export(m, x, **kw):
    """torch's TorchScript exporter without the `onnx` package: its C++ serialiser writes the ModelProto, only the
    onnxscript-function post-pass imports onnx - skipped (there are no such functions here)"""
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    keep = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    try:
        f = io.BytesIO()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(m, x, f, opset_version=11, dynamo=False, **kw)
        return f.getvalue()
    finally:
        onnx_proto_utils._add_onnxscript_fn = keep


def test_reader_and_interpreter_against_torchs_own_exporter():
    torch.manual_seed(0)
    m = _Small().eval()
    x = torch.randn(2, 16, 24, 3)
    try:
        data = _torch_export(m, x, input_names=["images"], output_names=["outs"], dynamic_axes={"images": {0: "batch"}})
    except Exception as e:       # noqa: BLE001  (a torch build whose exporter cannot run without onnx: nothing to pin against)
        pytest.skip("torch's exporter is unavailable here: %r" % (e,))
    model = OI.load(data)
    assert model["opset"] == 11 and model["inputs"] == ["images"] and model["outputs"] == ["outs"]
    assert {n[0] for n in model["nodes"]} >= {"Conv", "Sigmoid", "Mul", "MaxPool", "Resize", "Slice", "Concat", "Transpose"}
    x2 = torch.randn(3, 16, 24, 3)
    (got,) = OI.run(model, {"images": x2.numpy()})
    np.testing.assert_allclose(got, m(x2).detach().numpy(), rtol=1e-5, atol=1e-5)


def _model():
    import yolov7_d2_amd as M
    import yolox_oracle as O
    model = M.build_model(M.yolox_s_cfg(device="cpu"))
    model.load_state_dict(O.init_state_dict(0.33, 0.5, 80, seed=0))
    return model.eval(), O


def test_exported_yolox_graph_reproduces_the_references_export_mode_output(golden_dir):
    from yolov7_d2_amd.export_onnx import export_yolox_onnx
    model, O = _model()
    f = io.BytesIO()
    data = export_yolox_onnx(model, f, height=64, width=96)
    assert f.getvalue() == data and len(data) > 30_000_000        # 8.97 M fp32 parameters
    g = OI.load(data)
    assert g["opset"] == 11 and g["ir_version"] == 6 and g["inputs"] == ["images"] and g["outputs"] == ["outs"]
    ops = [n[0] for n in g["nodes"]]
    assert ops.count("Conv") == 83 and "BatchNormalization" not in ops          # 83 convs, every BN folded
    imgs, _ = O.synth_batch(2, 64, 96, seed=11, max_gt=4)
    (out,) = OI.run(g, {"images": imgs.permute(0, 2, 3, 1).contiguous().numpy()})
    ref = np.load(os.path.join(golden_dir, "yolox_s_onnx_layout_64x96.npz"))["out"]
    assert out.shape == ref.shape == (2, 126, 86)
    np.testing.assert_allclose(out[..., :5], ref[..., :5], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(out[..., 6:], ref[..., 6:], rtol=2e-4, atol=2e-5)
    assert np.array_equal(out[..., 5], out[..., 6:].argmax(-1).astype(np.float32))
    assert (out[..., 5] == ref[..., 5]).mean() > 0.98
    (one,) = OI.run(g, {"images": imgs.permute(0, 2, 3, 1).contiguous().numpy()[:1]})      # the batch axis is dynamic
    keep = [c for c in range(86) if c != 5]                     # (the class-index column flips on the random-init near-ties)
    np.testing.assert_allclose(one[..., keep], out[:1][..., keep], rtol=1e-4, atol=1e-5)
    model.train()
    with pytest.raises(RuntimeError):
        export_yolox_onnx(model, io.BytesIO(), 64, 96)


@pytest.mark.skipif(not os.path.isdir("/root/reference/yolov7"), reason="the reference tree only exists in the build container")
@pytest.mark.parametrize("depth,width,depthwise,nc", [(0.33, 0.375, False, 80), (0.33, 0.5, True, 20)], ids=["tiny", "depthwise_20cls"])
def test_exported_variants_against_the_reference_run_by_path(depth, width, depthwise, nc):
    """other widths (YOLOX-tiny, BASELINE configs[0]) and MODEL.DARKNET.DEPTH_WISE: the exported graph against the
    reference's own modules (loaded by path) in export mode, with BatchNorm running statistics that are not the identity"""
    import ref_loader
    import yolov7_d2_amd as M
    from yolov7_d2_amd.export_onnx import export_yolox_onnx
    ref, _ = ref_loader.build_reference_yolox(depth, width, nc, seed=3, depthwise=depthwise)
    g = torch.Generator().manual_seed(5)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.weight.data.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    ref.eval()
    ref.head.onnx_export = True
    cfg = M.yolox_s_cfg(device="cpu")
    cfg.MODEL.YOLO.DEPTH_MUL, cfg.MODEL.YOLO.WIDTH_MUL, cfg.MODEL.YOLO.CLASSES = depth, width, nc
    cfg.MODEL.DARKNET.DEPTH_WISE = depthwise
    model = M.build_model(cfg)
    missing = model.load_state_dict(ref.state_dict(), strict=False)
    assert not missing.missing_keys, missing.missing_keys[:5]
    data = export_yolox_onnx(model.eval(), io.BytesIO(), height=96, width=64)
    x = torch.rand(2, 96, 64, 3, generator=g) * 255
    (out,) = OI.run(OI.load(data), {"images": x.numpy()})
    with torch.no_grad():
        want = ref(x.permute(0, 3, 1, 2)).numpy()
    assert out.shape == want.shape == (2, 126, 6 + nc)
    keep = [c for c in range(6 + nc) if c != 5]
    np.testing.assert_allclose(out[..., keep], want[..., keep], rtol=5e-4, atol=5e-4)
    assert (out[..., 5] == want[..., 5]).mean() > 0.97


@pytest.mark.skipif(not os.path.isdir("/root/reference/yolov7"), reason="the reference tree only exists in the build container")
def test_written_graph_equals_the_file_torchs_exporter_traces_from_the_reference():
    """what export.py itself would write: the REFERENCE's modules (by path, head.onnx_export = True) traced by torch's
    TorchScript exporter at opset 11 with constant folding - against the graph this package writes from its module tree
    with the same weights: the same operator census where it matters (83 Conv, 80 Sigmoid, 76 Mul, 8 Add, 3 MaxPool, 2
    Resize, no BatchNormalization in either) and the same output when both files are executed"""
    import collections
    import contextlib
    import ref_loader
    import yolox_oracle as O
    from yolov7_d2_amd.export_onnx import export_yolox_onnx
    ref, _ = ref_loader.build_reference_yolox(0.33, 0.5, 80, seed=0)
    ref.load_state_dict(O.init_state_dict(0.33, 0.5, 80, seed=0))
    ref.eval()
    ref.head.onnx_export = True
    imgs, _ = O.synth_batch(2, 64, 96, seed=11, max_gt=4)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            traced = _torch_export(ref, imgs, do_constant_folding=True, input_names=["images"], output_names=["outs"])
    except Exception as e:       # noqa: BLE001
        pytest.skip("torch's exporter is unavailable here: %r" % (e,))
    model, _ = _model()
    mine = export_yolox_onnx(model, io.BytesIO(), height=64, width=96)
    gt, gm = OI.load(traced), OI.load(mine)
    ct, cm = collections.Counter(n[0] for n in gt["nodes"]), collections.Counter(n[0] for n in gm["nodes"])
    for op in ("Conv", "Sigmoid", "Mul", "Add", "MaxPool", "Resize", "Exp", "ArgMax", "Split", "BatchNormalization"):
        assert ct[op] == cm[op], (op, ct[op], cm[op])
    (a,) = OI.run(gt, {"images": imgs.numpy()})                                   # the traced file takes NCHW (the meta-arch's
    (b,) = OI.run(gm, {"images": imgs.permute(0, 2, 3, 1).contiguous().numpy()})  # preprocess_input permute sits outside `ref`)
    keep = [c for c in range(86) if c != 5]
    np.testing.assert_allclose(b[..., keep], a[..., keep], rtol=2e-4, atol=2e-4)
    assert (a[..., 5] == b[..., 5]).mean() > 0.98
