"""ONNX export of the YOLOX model (the reference's export.py:237-303: `model.onnx_export = True;
torch.onnx.export(model, inp, onnx_f, input_names, output_names, opset_version=11, dynamic_axes=...)`).

The reference traces its nn.Module forward.  This model's forward is a HIP plan, not a traceable graph - and a torch-op forward
kept around for tracing would be a CPU implementation of the model by another name - so the exporter WRITES the graph: it walks
the module tree (the same attribute paths the reference's checkpoints name), folds every eval-mode BatchNorm into its
convolution and emits the ONNX nodes of the reference's export-mode forward - `preprocess_input` (NHWC -> NCHW,
meta_arch/yolox.py:164-169), Focus, CSPDarknet, YOLOPAFPN, the head with `decode_outputs`' export layout (xy, wh, conf,
class index, class probabilities; head/yolox_head.py:247-269) - as an opset-11 ModelProto, serialised with a few lines of
protobuf wire format (no `onnx` package in this image; field numbers as torch's own exporter writes them, which the tests
cross-check).  No tensor arithmetic happens here beyond the BatchNorm fold of the weights.

`export_yolox_onnx(model, f, height, width)` -> bytes written to `f` (path or file object); input "images" [batch, H, W, 3]
float32 (batch dynamic), output "outs" [batch, A, 6 + num_classes] - the names of export.py:237-249's default branch."""
import struct

import numpy as np

FLOAT, INT64 = 1, 7            # TensorProto.DataType
A_FLOAT, A_INT, A_STRING, A_TENSOR, A_INTS = 1, 2, 3, 4, 7      # AttributeProto.AttributeType


# ---------------------------------------------------------------- protobuf wire format (varint / length-delimited / fixed32)
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _f_varint(field, v):
    return _varint(field << 3) + _varint(v)


def _f_bytes(field, b):
    if isinstance(b, str):
        b = b.encode()
    return _varint((field << 3) | 2) + _varint(len(b)) + bytes(b)


def _f_float(field, v):
    return _varint((field << 3) | 5) + struct.pack("<f", v)


def _tensor(name, arr):
    """TensorProto: dims = 1, data_type = 2, name = 8, raw_data = 9"""
    arr = np.ascontiguousarray(arr)
    dt = {np.dtype(np.float32): FLOAT, np.dtype(np.int64): INT64}[arr.dtype]
    out = b"".join(_f_varint(1, d) for d in arr.shape) + _f_varint(2, dt)
    if name:
        out += _f_bytes(8, name)
    return out + _f_bytes(9, arr.tobytes())


def _attr(name, v):
    """AttributeProto: name = 1, f = 2, i = 3, s = 4, t = 5, ints = 8, type = 20"""
    out = _f_bytes(1, name)
    if isinstance(v, bool) or isinstance(v, (int, np.integer)):
        return out + _f_varint(3, int(v)) + _f_varint(20, A_INT)
    if isinstance(v, float):
        return out + _f_float(2, v) + _f_varint(20, A_FLOAT)
    if isinstance(v, str):
        return out + _f_bytes(4, v) + _f_varint(20, A_STRING)
    if isinstance(v, np.ndarray):
        return out + _f_bytes(5, _tensor("", v)) + _f_varint(20, A_TENSOR)
    return out + b"".join(_f_varint(8, int(x)) for x in v) + _f_varint(20, A_INTS)


def _value_info(name, elem, dims):
    """ValueInfoProto name = 1, type = 2 { tensor_type = 1 { elem_type = 1, shape = 2 { dim = 1 { dim_value = 1 | dim_param = 2 } } } }"""
    shape = b"".join(_f_bytes(1, _f_bytes(2, d) if isinstance(d, str) else _f_varint(1, d)) for d in dims)
    return _f_bytes(1, name) + _f_bytes(2, _f_bytes(1, _f_varint(1, elem) + _f_bytes(2, shape)))


class OnnxGraph:
    """nodes in emission order (already topological), initializers, one input / output list"""

    def __init__(self, opset=11, producer="yolov7_d2_amd"):
        self.opset, self.producer = opset, producer
        self.nodes, self.inits, self._n = [], [], {}

    def _name(self, base):
        k = self._n.get(base, 0)
        self._n[base] = k + 1
        return base if k == 0 else f"{base}_{k}"

    def init(self, name, arr):
        self.inits.append(_f_bytes(5, _tensor(name, arr)))
        return name

    def node(self, op, inputs, scope, nout=1, outputs=None, **attrs):
        """NodeProto: input = 1, output = 2, name = 3, op_type = 4, attribute = 5.  Returns the output name(s)."""
        nm = self._name(f"{scope}/{op}")
        outs = list(outputs) if outputs is not None else [f"{nm}_output_{k}" for k in range(nout)]
        body = b"".join(_f_bytes(1, i) for i in inputs) + b"".join(_f_bytes(2, o) for o in outs) + _f_bytes(3, nm) + _f_bytes(4, op)
        body += b"".join(_f_bytes(5, _attr(k, v)) for k, v in sorted(attrs.items()))
        self.nodes.append(_f_bytes(1, body))
        return outs[0] if len(outs) == 1 else outs

    def serialize(self, inputs, outputs, name="main_graph"):
        """ModelProto: ir_version = 1, producer_name = 2, producer_version = 3, graph = 7, opset_import = 8 { version = 2 };
        GraphProto: node = 1, name = 2, initializer = 5, input = 11, output = 12"""
        g = b"".join(self.nodes) + _f_bytes(2, name) + b"".join(self.inits)
        g += b"".join(_f_bytes(11, _value_info(*i)) for i in inputs) + b"".join(_f_bytes(12, _value_info(*o)) for o in outputs)
        return _f_varint(1, 6) + _f_bytes(2, self.producer) + _f_bytes(3, "round3") + _f_bytes(7, g) + _f_bytes(8, _f_varint(2, self.opset))


# ---------------------------------------------------------------- the YOLOX graph
def _np(t):
    return t.detach().float().cpu().numpy()


class _YoloxEmitter:
    def __init__(self, g):
        self.g = g

    def base_conv(self, m, x, scope):
        """BaseConv in eval mode (layers/wrappers.py:60-83): conv -> BatchNorm (running statistics, folded) -> SiLU"""
        w = _np(m.conv.weight).astype(np.float64)
        bn = m.bn
        scale = _np(bn.weight).astype(np.float64) / np.sqrt(_np(bn.running_var).astype(np.float64) + bn.eps)
        bias = _np(bn.bias).astype(np.float64) - _np(bn.running_mean).astype(np.float64) * scale
        if m.conv.bias is not None:
            bias = bias + _np(m.conv.bias).astype(np.float64) * scale
        k, s = m.ksize, m.stride
        W = self.g.init(scope + ".conv.weight", (w * scale[:, None, None, None]).astype(np.float32))
        B = self.g.init(scope + ".conv.bias", bias.astype(np.float32))
        y = self.g.node("Conv", [x, W, B], scope, dilations=[1, 1], group=int(m.groups), kernel_shape=[k, k],
                        pads=[(k - 1) // 2] * 4, strides=[s, s])
        if not isinstance(m.act, type(None)) and m.act.__class__.__name__ == "SiLU":
            sg = self.g.node("Sigmoid", [y], scope)
            return self.g.node("Mul", [y, sg], scope)
        raise NotImplementedError("export: BaseConv activation %r (the reference's YOLOX configs use SiLU)" % m.act)

    def conv_like(self, m, x, scope):
        if m.__class__.__name__ == "DWConv":
            return self.base_conv(m.pconv, self.base_conv(m.dconv, x, scope + ".dconv"), scope + ".pconv")
        return self.base_conv(m, x, scope)

    def bottleneck(self, m, x, scope):
        y = self.conv_like(m.conv2, self.base_conv(m.conv1, x, scope + ".conv1"), scope + ".conv2")
        return self.g.node("Add", [y, x], scope) if m.use_add else y

    def csp(self, m, x, scope):
        x1 = self.base_conv(m.conv1, x, scope + ".conv1")
        x2 = self.base_conv(m.conv2, x, scope + ".conv2")
        for i, b in enumerate(m.m):
            x1 = self.bottleneck(b, x1, f"{scope}.m.{i}")
        return self.base_conv(m.conv3, self.g.node("Concat", [x1, x2], scope, axis=1), scope + ".conv3")

    def spp(self, m, x, scope):
        x = self.base_conv(m.conv1, x, scope + ".conv1")
        pools = [self.g.node("MaxPool", [x], f"{scope}.m.{i}", ceil_mode=0, dilations=[1, 1], kernel_shape=[p.kernel_size] * 2,
                             pads=[p.padding] * 4, strides=[1, 1]) for i, p in enumerate(m.m)]
        return self.base_conv(m.conv2, self.g.node("Concat", [x] + pools, scope, axis=1), scope + ".conv2")

    def focus(self, m, x, scope):
        """layers/wrappers.py:202-220: top-left, bottom-left, top-right, bottom-right, channel concat"""
        g = self.g
        big = np.iinfo(np.int64).max
        ends = g.init(scope + ".ends", np.array([big, big], np.int64))
        axes = g.init(scope + ".axes", np.array([2, 3], np.int64))
        steps = g.init(scope + ".steps", np.array([2, 2], np.int64))
        parts = []
        for tag, (oy, ox) in (("tl", (0, 0)), ("bl", (1, 0)), ("tr", (0, 1)), ("br", (1, 1))):
            st = g.init(f"{scope}.starts_{tag}", np.array([oy, ox], np.int64))
            parts.append(g.node("Slice", [x, st, ends, axes, steps], scope))
        return self.base_conv(m.conv, g.node("Concat", parts, scope, axis=1), scope + ".conv")

    def seq(self, mods, x, scope):
        for i, m in enumerate(mods):
            n = m.__class__.__name__
            x = (self.csp if n == "CSPLayer" else self.spp if n == "SPPBottleneck" else self.conv_like)(m, x, f"{scope}.{i}")
        return x

    def upsample(self, x, scope):
        g = self.g
        roi = g.init(scope + ".roi", np.zeros(0, np.float32))
        sc = g.init(scope + ".scales", np.array([1, 1, 2, 2], np.float32))
        return g.node("Resize", [x, roi, sc], scope, coordinate_transformation_mode="asymmetric", cubic_coeff_a=-0.75, mode="nearest",
                      nearest_mode="floor")


def build_yolox_graph(model, height, width):
    """-> (OnnxGraph, inputs, outputs) of the export-mode forward at a static H x W (divisible by 32), dynamic batch"""
    if height % 32 or width % 32:
        raise ValueError("export: height and width must be multiples of 32")
    bb, neck, head = model.backbone, model.neck, model.head
    g = OnnxGraph()
    e = _YoloxEmitter(g)
    x = g.node("Transpose", ["images"], "preprocess_input", perm=[0, 3, 1, 2])
    x = e.focus(bb.stem, x, "backbone.stem")
    x = e.seq(bb.dark2, x, "backbone.dark2")
    d3 = e.seq(bb.dark3, x, "backbone.dark3")
    d4 = e.seq(bb.dark4, d3, "backbone.dark4")
    d5 = e.seq(bb.dark5, d4, "backbone.dark5")
    # YOLOPAFPN.forward (neck/yolo_pafpn.py:75-114)
    fpn0 = e.base_conv(neck.lateral_conv0, d5, "neck.lateral_conv0")
    f0 = e.csp(neck.C3_p4, g.node("Concat", [e.upsample(fpn0, "neck.upsample0"), d4], "neck.p4", axis=1), "neck.C3_p4")
    fpn1 = e.base_conv(neck.reduce_conv1, f0, "neck.reduce_conv1")
    pan2 = e.csp(neck.C3_p3, g.node("Concat", [e.upsample(fpn1, "neck.upsample1"), d3], "neck.p3", axis=1), "neck.C3_p3")
    p1 = g.node("Concat", [e.conv_like(neck.bu_conv2, pan2, "neck.bu_conv2"), fpn1], "neck.n3", axis=1)
    pan1 = e.csp(neck.C3_n3, p1, "neck.C3_n3")
    p0 = g.node("Concat", [e.conv_like(neck.bu_conv1, pan1, "neck.bu_conv1"), fpn0], "neck.n4", axis=1)
    pan0 = e.csp(neck.C3_n4, p0, "neck.C3_n4")
    # YOLOXHead.forward, eval (head/yolox_head.py:140-224)
    nc = head.num_classes
    flat = g.init("head.flat_shape", np.array([0, 5 + nc, -1], np.int64))
    levels = []
    for k, feat in enumerate((pan2, pan1, pan0)):
        x = e.base_conv(head.stems[k], feat, f"head.stems.{k}")
        cf, rf = x, x
        for i, m in enumerate(head.cls_convs[k]):
            cf = e.conv_like(m, cf, f"head.cls_convs.{k}.{i}")
        for i, m in enumerate(head.reg_convs[k]):
            rf = e.conv_like(m, rf, f"head.reg_convs.{k}.{i}")

        def pred(m, t, scope):
            W = g.init(scope + ".weight", _np(m.weight))
            B = g.init(scope + ".bias", _np(m.bias))
            return g.node("Conv", [t, W, B], scope, dilations=[1, 1], group=1, kernel_shape=[1, 1], pads=[0, 0, 0, 0], strides=[1, 1])
        cls_o = g.node("Sigmoid", [pred(head.cls_preds[k], cf, f"head.cls_preds.{k}")], f"head.cls_preds.{k}")
        reg_o = pred(head.reg_preds[k], rf, f"head.reg_preds.{k}")
        obj_o = g.node("Sigmoid", [pred(head.obj_preds[k], rf, f"head.obj_preds.{k}")], f"head.obj_preds.{k}")
        out = g.node("Concat", [reg_o, obj_o, cls_o], f"head.level{k}", axis=1)
        levels.append(g.node("Reshape", [out, flat], f"head.level{k}"))
    o = g.node("Transpose", [g.node("Concat", levels, "head", axis=2)], "head", perm=[0, 2, 1])
    # decode_outputs, export layout (head/yolox_head.py:247-269)
    grids, strides = [], []
    for s in head.strides:
        hs, ws = height // s, width // s
        yv, xv = np.meshgrid(np.arange(hs), np.arange(ws), indexing="ij")
        grids.append(np.stack((xv, yv), 2).reshape(1, -1, 2))
        strides.append(np.full((1, hs * ws, 1), s))
    G = g.init("head.grids", np.concatenate(grids, 1).astype(np.float32))
    S = g.init("head.strides", np.concatenate(strides, 1).astype(np.float32))
    xy, wh, conf, prob = g.node("Split", [o], "head.decode", nout=4, axis=2, split=[2, 2, 1, nc])
    xy = g.node("Mul", [g.node("Add", [xy, G], "head.decode"), S], "head.decode")
    wh = g.node("Mul", [g.node("Exp", [wh], "head.decode"), S], "head.decode")
    idx = g.node("Cast", [g.node("ArgMax", [prob], "head.decode", axis=2, keepdims=1)], "head.decode", to=FLOAT)
    g.node("Concat", [xy, wh, conf, idx, prob], "head.decode", outputs=["outs"], axis=2)
    A = sum((height // s) * (width // s) for s in head.strides)
    return g, [("images", FLOAT, ["batch", height, width, 3])], [("outs", FLOAT, ["batch", A, 6 + nc])]


def export_yolox_onnx(model, f, height=640, width=640):
    """export.py:237-303 for the YOLOX meta-architecture: writes the opset-11 model and returns its bytes.  The model must be
    in eval mode (the reference's DefaultPredictor calls model.eval()); its parameters may live on any device."""
    if model.training:
        raise RuntimeError("export_yolox_onnx: an inference graph - call model.eval() first (BatchNorm folds its running statistics)")
    g, ins, outs = build_yolox_graph(model, height, width)
    data = g.serialize(ins, outs)
    if hasattr(f, "write"):
        f.write(data)
    else:
        with open(f, "wb") as fh:
            fh.write(data)
    return data
