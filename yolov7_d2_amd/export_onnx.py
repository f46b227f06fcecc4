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

FLOAT, INT64, BOOL = 1, 7, 9            # TensorProto.DataType
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


# ---------------------------------------------------------------- shared emitters of the detectron2-shaped networks
class _NetEmitter:
    """Conv2d (+ folded FrozenBatchNorm2d), ReLU, nn.Linear, bilinear / nearest resizes as opset-11 nodes"""

    def __init__(self, g):
        self.g = g

    def conv(self, m, x, scope, relu=False, weight=None, bias=None, stride=None, padding=None, groups=1):
        """nn.Conv2d, or the detectron2 Conv2d of modeling/resnet.py (attribute `norm`: FrozenBatchNorm2d folded here in
        fp64: W * scale, shift as the bias)"""
        w = _np(m.weight if weight is None else weight).astype(np.float64)
        b = None if (bias is None and getattr(m, "bias", None) is None) else _np(m.bias if bias is None else bias).astype(np.float64)
        norm = getattr(m, "norm", None)
        if norm is not None:
            sc = _np(norm.weight).astype(np.float64) / np.sqrt(_np(norm.running_var).astype(np.float64) + norm.eps)
            sh = _np(norm.bias).astype(np.float64) - _np(norm.running_mean).astype(np.float64) * sc
            w, b = w * sc[:, None, None, None], sh if b is None else b * sc + sh
        k = w.shape[2]
        st = stride if stride is not None else (m.stride if isinstance(m.stride, int) else m.stride[0])
        pd = padding if padding is not None else (m.padding if isinstance(m.padding, int) else m.padding[0])
        ins = [x, self.g.init(scope + ".weight", w.astype(np.float32))]
        if b is not None:
            ins.append(self.g.init(scope + ".bias", b.astype(np.float32)))
        y = self.g.node("Conv", ins, scope, dilations=[1, 1], group=int(groups), kernel_shape=[k, k], pads=[pd] * 4, strides=[st, st])
        return self.g.node("Relu", [y], scope) if relu else y

    def linear(self, lin, x, scope, relu=False):
        W = self.g.init(scope + ".weight_t", np.ascontiguousarray(_np(lin.weight).T))
        y = self.g.node("Add", [self.g.node("MatMul", [x, W], scope), self.g.init(scope + ".bias", _np(lin.bias))], scope)
        return self.g.node("Relu", [y], scope) if relu else y

    def resize_bilinear(self, x, scope, size=None, scale=None):
        """F.interpolate(mode="bilinear", align_corners=False) as torch's opset-11 exporter writes it: Resize, linear,
        coordinate_transformation_mode half_pixel (torch 2.x; 1.x wrote pytorch_half_pixel, which differs from eager
        PyTorch only for an output dimension of 1 - none here); `size` (H, W) -> the sizes input with N, C copied from the
        tensor's own shape (dynamic batch), `scale` -> the scales input"""
        g = self.g
        roi = g.init(scope + ".roi", np.zeros(0, np.float32))
        kw = dict(coordinate_transformation_mode="half_pixel", cubic_coeff_a=-0.75, mode="linear", nearest_mode="floor")
        if scale is not None:
            return g.node("Resize", [x, roi, g.init(scope + ".scales", np.array([1, 1, scale, scale], np.float32))], scope, **kw)
        nc = g.node("Slice", [g.node("Shape", [x], scope), g.init(scope + ".s0", np.array([0], np.int64)),
                              g.init(scope + ".s2", np.array([2], np.int64)), g.init(scope + ".ax0", np.array([0], np.int64))], scope)
        sizes = g.node("Concat", [nc, g.init(scope + ".hw", np.array(list(size), np.int64))], scope, axis=0)
        return g.node("Resize", [x, roi, g.init(scope + ".noscales", np.zeros(0, np.float32)), sizes], scope, **kw)

    def upsample_nearest2(self, x, scope):
        g = self.g
        return g.node("Resize", [x, g.init(scope + ".roi", np.zeros(0, np.float32)), g.init(scope + ".scales", np.array([1, 1, 2, 2], np.float32))],
                      scope, coordinate_transformation_mode="asymmetric", cubic_coeff_a=-0.75, mode="nearest", nearest_mode="floor")

    # ---- detectron2 ResNet (modeling/resnet.py; the reference builds it with d2's build_resnet_backbone)
    def resnet(self, bb, x, scope):
        g = self.g
        x = self.conv(bb.stem.conv1, x, scope + ".stem.conv1", relu=True)
        x = g.node("MaxPool", [x], scope + ".stem", ceil_mode=0, dilations=[1, 1], kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2])
        outs = {}
        if g.opset != 11:       # Softmax(axis), ReduceSum / Split attributes below are the opset-11 forms
            raise NotImplementedError(f"export_onnx: the SparseInst / Detr emitters write opset-11 nodes (graph opset {g.opset})")
        for name in bb.stage_names:
            for i, blk in enumerate(getattr(bb, name)):
                if not all(hasattr(blk, a) for a in ("conv1", "conv2", "conv3", "shortcut")):
                    # d2's BasicBlock (R18 / R34: two 3x3 convs) is not written here; the reference's Detr / SparseInst
                    # configs all use the bottleneck depths
                    raise NotImplementedError(f"export_onnx: {type(blk).__name__} in {name}: only bottleneck ResNet blocks "
                                              "(conv1 / conv2 / conv3 + shortcut) are emitted")
                sc = f"{scope}.{name}.{i}"
                y = self.conv(blk.conv1, x, sc + ".conv1", relu=True)
                y = self.conv(blk.conv2, y, sc + ".conv2", relu=True)
                y = self.conv(blk.conv3, y, sc + ".conv3")
                short = x if blk.shortcut is None else self.conv(blk.shortcut, x, sc + ".shortcut")
                x = g.node("Relu", [g.node("Add", [y, short], sc)], sc)
            outs[name] = x
        return outs


def _normalised_nchw(g, model, scope="preprocess"):
    """`preprocess_inputs_onnx` (sparseinst.py:122-125) / the DETR export branch: NHWC float image -> NCHW, (x - mean) / std"""
    x = g.node("Transpose", ["images"], scope, perm=[0, 3, 1, 2])
    mean = g.init(scope + ".pixel_mean", _np(model.pixel_mean).reshape(1, 3, 1, 1))
    std = g.init(scope + ".pixel_std", _np(model.pixel_std).reshape(1, 3, 1, 1))
    return g.node("Div", [g.node("Sub", [x, mean], scope), std], scope)


# ---------------------------------------------------------------- SparseInst
def build_sparseinst_graph(model, height, width):
    """The graph `torch.onnx.export` traces out of the reference's SparseInst in export mode (meta_arch/sparseinst.py:127-162
    with `torch.onnx.is_in_onnx_export()` true): preprocess_inputs_onnx, ResNet, InstanceContextEncoder
    (encoder_sparseinst.py:55-127), the IAM decoder with its export-mode bmm (decoder_sparseinst.py:133-161, 203-233) and
    `inference_onnx` (sparseinst.py:236-345) INCLUDING its two batch quirks, which a drop-in must reproduce: the top-k
    indices of every image index the FLATTENED [B * N] scores / masks (exact for the batch of 1 the reference exports
    with), and `[:, :h, :w]` slices dimensions 1 and 2 of the [B, K, H, W] masks (a no-op for square inputs).
    Static H x W (multiples of 32), dynamic batch.  Outputs masks (bool [B, K', H', W]), scores [B, K], labels [B, K]."""
    if height % 32 or width % 32:
        raise ValueError("export: height and width must be multiples of 32")
    g = OnnxGraph()
    e = _NetEmitter(g)
    I64 = lambda name, v: g.init(name, np.array(v, np.int64))
    x = _normalised_nchw(g, model)
    feats = e.resnet(model.backbone, x, "backbone")
    enc, dec = model.encoder, model.decoder
    # ---- InstanceContextEncoder.forward
    fs = [feats[f] for f in enc.in_features][::-1]
    strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
    hw = [(height // strides[f], width // strides[f]) for f in enc.in_features][::-1]
    t = e.conv(enc.fpn_laterals[0], fs[0], "encoder.fpn_laterals.0")
    h0, w0 = hw[0]
    priors = []
    for i, st in enumerate(enc.ppm.stages):
        sz = st[0].sz if isinstance(st[0].sz, (tuple, list)) else (st[0].sz, st[0].sz)
        kh, kw = -(-h0 // sz[0]), -(-w0 // sz[1])                     # MyAdaptiveAvgPool2d: ceil(size / sz) windows, floor mode
        p = g.node("AveragePool", [t], f"encoder.ppm.stages.{i}.0", ceil_mode=0, kernel_shape=[kh, kw], pads=[0, 0, 0, 0], strides=[kh, kw])
        p = e.conv(st[1], p, f"encoder.ppm.stages.{i}.1", relu=True)
        priors.append(e.resize_bilinear(p, f"encoder.ppm.stages.{i}.up", size=(h0, w0)))
    prev = e.conv(enc.ppm.bottleneck, g.node("Concat", priors + [t], "encoder.ppm", axis=1), "encoder.ppm.bottleneck", relu=True)
    outs = [e.conv(enc.fpn_outputs[0], prev, "encoder.fpn_outputs.0")]
    for k in range(1, len(fs)):
        lat = e.conv(enc.fpn_laterals[k], fs[k], f"encoder.fpn_laterals.{k}")
        prev = g.node("Add", [lat, e.upsample_nearest2(prev, f"encoder.up{k}")], f"encoder.top_down{k}")
        outs.insert(0, e.conv(enc.fpn_outputs[k], prev, f"encoder.fpn_outputs.{k}"))
    H, W = hw[-1]
    fused = [outs[0]] + [e.resize_bilinear(o, f"encoder.fuse{k}", size=(H, W)) for k, o in enumerate(outs[1:], 1)]
    feat = e.conv(enc.fusion, g.node("Concat", fused, "encoder.fuse", axis=1), "encoder.fusion")
    # ---- BaseIAMDecoder.forward: coordinates (x, y in [-1, 1]) ++ features
    ys, xs = np.meshgrid(np.linspace(-1, 1, H, dtype=np.float32), np.linspace(-1, 1, W, dtype=np.float32), indexing="ij")
    coords = g.init("decoder.coords", np.stack([xs, ys])[None].astype(np.float32))                     # [1, 2, H, W]
    bshape = g.node("Concat", [g.node("Slice", [g.node("Shape", [feat], "decoder"), I64("decoder.s0", [0]), I64("decoder.s1", [1]),
                                                 I64("decoder.ax0", [0])], "decoder"), I64("decoder.chw", [2, H, W])], "decoder", axis=0)
    f = g.node("Concat", [g.node("Expand", [coords, bshape], "decoder"), feat], "decoder", axis=1)

    def stack(seq, t, scope):
        mods = list(seq)
        for i in range(0, len(mods), 2):                              # (conv, ReLU) pairs of _make_stack_3x3_convs
            t = e.conv(mods[i], t, f"{scope}.{i}", relu=True)
        return t

    ib = dec.inst_branch
    xi = stack(ib.inst_convs, f, "decoder.inst_branch.inst_convs")
    G = getattr(ib, "num_groups", 1)
    iam = e.conv(ib.iam_conv, xi, "decoder.inst_branch.iam_conv", groups=G)
    N, Cc = ib.iam_conv.out_channels, ib.iam_conv.in_channels
    prob = g.node("Reshape", [g.node("Sigmoid", [iam], "decoder.inst_branch"), I64("decoder.prob_shape", [0, N, -1])], "decoder.inst_branch")
    xt = g.node("Transpose", [g.node("Reshape", [xi, I64("decoder.x_shape", [0, Cc, -1])], "decoder.inst_branch")], "decoder.inst_branch", perm=[0, 2, 1])
    inst = g.node("MatMul", [prob, xt], "decoder.inst_branch")                                          # [B, N, C]
    norm = g.node("ReduceSum", [prob], "decoder.inst_branch", axes=[2], keepdims=1)
    lo, hi = g.init("decoder.norm_min", np.array(1e-6, np.float32)), g.init("decoder.norm_max", np.array(1e5, np.float32))
    norm = g.node("Clip", [norm, lo, hi] if G > 1 else [norm, lo], "decoder.inst_branch")              # (:74 clamp(min) / :224 clamp(min, max))
    inst = g.node("Div", [inst, norm], "decoder.inst_branch")
    if G > 1:
        d4 = N // G
        inst = g.node("Reshape", [inst, I64("decoder.group_shape", [0, G, d4, -1])], "decoder.inst_branch")
        inst = g.node("Reshape", [g.node("Transpose", [inst], "decoder.inst_branch", perm=[0, 2, 1, 3]), I64("decoder.inst_shape", [0, d4, -1])],
                      "decoder.inst_branch")
        inst = e.linear(ib.fc, inst, "decoder.inst_branch.fc", relu=True)
        N = d4
    logits = e.linear(ib.cls_score, inst, "decoder.inst_branch.cls_score")
    kernel = e.linear(ib.mask_kernel, inst, "decoder.inst_branch.mask_kernel")
    objn = e.linear(ib.objectness, inst, "decoder.inst_branch.objectness")
    mb = dec.mask_branch
    mf = e.conv(mb.projection, stack(mb.mask_convs, f, "decoder.mask_branch.mask_convs"), "decoder.mask_branch.projection")
    Kd = mb.projection.out_channels
    masks = g.node("MatMul", [kernel, g.node("Reshape", [mf, I64("decoder.mf_shape", [0, Kd, -1])], "decoder")], "decoder")
    masks = g.node("Reshape", [masks, I64("decoder.mask_shape", [0, N, H, W])], "decoder")
    sf = float(dec.scale_factor)
    masks = e.resize_bilinear(masks, "decoder.upsample", scale=sf)
    Hm, Wm = int(H * sf), int(W * sf)
    # ---- inference_onnx
    sc = "inference_onnx"
    ps = g.node("Mul", [g.node("Sigmoid", [logits], sc), g.node("Sigmoid", [objn], sc)], sc)
    ps = g.node("Sqrt", [g.node("Add", [ps, g.init(sc + ".eps", np.array(1e-3, np.float32))], sc)], sc)
    pm = g.node("Sigmoid", [masks], sc)
    scores = g.node("ReduceMax", [ps], sc, axes=[2], keepdims=0)                                       # torch.max(pred_scores, dim=-1)
    labels = g.node("ArgMax", [ps], sc, axis=2, keepdims=0)
    K = min(50, int(model.max_detections))
    _, keep = g.node("TopK", [scores, I64(sc + ".k", [K])], sc, nout=2, axis=-1, largest=1, sorted=1)
    flat1 = I64(sc + ".flat", [-1])
    scores = g.node("Gather", [g.node("Reshape", [scores, flat1], sc), keep], sc, axis=0)              # scores.view(-1)[keep_flt]
    g.node("Gather", [g.node("Reshape", [labels, flat1], sc), keep], sc, outputs=["labels"], axis=0)
    mk = g.node("Gather", [g.node("Reshape", [pm, I64(sc + ".mflat", [-1, Hm, Wm])], sc), keep], sc, axis=0)      # [B, K, Hm, Wm]
    thr = g.init(sc + ".mask_threshold", np.array(float(model.mask_threshold), np.float32))
    hard = g.node("Cast", [g.node("Greater", [mk, thr], sc)], sc, to=FLOAT)
    num = g.node("ReduceSum", [g.node("Mul", [mk, hard], sc)], sc, axes=[2, 3], keepdims=0)
    den = g.node("Add", [g.node("ReduceSum", [hard], sc, axes=[2, 3], keepdims=0), g.init(sc + ".eps6", np.array(1e-6, np.float32))], sc)
    g.node("Mul", [scores, g.node("Div", [num, den], sc)], sc, outputs=["scores"])                   # rescoring_mask_batch
    up = e.resize_bilinear(mk, sc + ".upsample", size=(height, width))
    d1, d2 = min(K, height), min(height, width)                                                         # [:, :h, :w] on a 4-D tensor
    up = g.node("Slice", [up, I64(sc + ".sl0", [0, 0]), I64(sc + ".sl1", [d1, d2]), I64(sc + ".slax", [1, 2])], sc)
    g.node("Greater", [up, thr], sc, outputs=["masks"])
    return g, [("images", FLOAT, ["batch", height, width, 3])], \
        [("masks", BOOL, ["batch", d1, d2, width]), ("scores", FLOAT, ["batch", K]), ("labels", INT64, ["batch", K])]


def export_sparseinst_onnx(model, f, height=640, width=640):
    """export.py:237-303 for a `sparse_inst` config: input_names ["images"], output_names ["masks", "scores", "labels"],
    dynamic batch (get_model_infos, export.py:237-243).  The parameters may live on any device; eval mode required."""
    if model.training:
        raise RuntimeError("export_sparseinst_onnx: an inference graph - call model.eval() first")
    g, ins, outs = build_sparseinst_graph(model, height, width)
    return _write(g.serialize(ins, outs), f)


def _write(data, f):
    if hasattr(f, "write"):
        f.write(data)
    else:
        with open(f, "wb") as fh:
            fh.write(data)
    return data


# ---------------------------------------------------------------- DETR
def _sine_position_embedding(pe, H, W):
    """PositionEmbeddingSine.forward (backbone/detr_backbone.py:334-375) for a mask without padding - the export branch's
    mask (meta_arch/detr.py:362-375: the all-False pixel mask, nearest-resized) - as a constant [1, 2 * N, H, W]; fp32
    arithmetic in the reference's order"""
    f32 = np.float32
    y = np.broadcast_to(np.arange(1, H + 1, dtype=f32)[:, None], (H, W)).copy()
    x = np.broadcast_to(np.arange(1, W + 1, dtype=f32)[None, :], (H, W)).copy()
    if pe.normalize:
        eps = f32(1e-6)
        if pe.centered:
            y, x = (y - f32(0.5)) / (y[-1:, :] + eps) * f32(pe.scale), (x - f32(0.5)) / (x[:, -1:] + eps) * f32(pe.scale)
        else:
            y, x = y / (y[-1:, :] + eps) * f32(pe.scale), x / (x[:, -1:] + eps) * f32(pe.scale)
    N = pe.num_pos_feats
    dim_t = np.arange(N, dtype=f32)
    dim_t = (f32(pe.temperature) ** (f32(2) * np.floor(dim_t / 2) / f32(N))).astype(f32)
    out = []
    for e in (y, x):
        p = (e[:, :, None] / dim_t).astype(f32)                                              # [H, W, N]
        out.append(np.stack((np.sin(p[:, :, 0::2]), np.cos(p[:, :, 1::2])), axis=3).reshape(H, W, N))
    return np.concatenate(out, axis=2).transpose(2, 0, 1)[None].astype(f32)


class _TransformerEmitter(_NetEmitter):
    """nn.MultiheadAttention / LayerNorm / the DETR encoder and decoder layers on sequence-first [L, B, E] tensors"""

    def layer_norm(self, ln, x, scope):
        g = self.g
        mu = g.node("ReduceMean", [x], scope, axes=[-1], keepdims=1)
        d = g.node("Sub", [x, mu], scope)
        var = g.node("ReduceMean", [g.node("Mul", [d, d], scope)], scope, axes=[-1], keepdims=1)
        y = g.node("Div", [d, g.node("Sqrt", [g.node("Add", [var, g.init(scope + ".eps", np.array(ln.eps, np.float32))], scope)], scope)], scope)
        return g.node("Add", [g.node("Mul", [y, g.init(scope + ".weight", _np(ln.weight))], scope), g.init(scope + ".bias", _np(ln.bias))], scope)

    def mha(self, m, q, k, v, Lq, Lk, scope):
        """nn.MultiheadAttention.forward without masks (the export branch's key-padding mask is all False) and without
        dropout (eval): in-projection rows [0, E), [E, 2E), [2E, 3E) of in_proj_weight; q scaled by head_dim ** -0.5"""
        g = self.g
        E, h = m.embed_dim, m.num_heads
        hd = E // h
        Wm, bm = _np(m.in_proj_weight), _np(m.in_proj_bias)

        def proj(x, r, name):
            W = g.init(f"{scope}.{name}.weight_t", np.ascontiguousarray(Wm[r * E:(r + 1) * E].T))
            b = g.init(f"{scope}.{name}.bias", np.ascontiguousarray(bm[r * E:(r + 1) * E]))
            return g.node("Add", [g.node("MatMul", [x, W], scope), b], scope)

        def heads(x, Lx, name):      # [L, B, E] -> [B * h, L, hd]
            x = g.node("Reshape", [x, g.init(f"{scope}.{name}.split", np.array([Lx, -1, hd], np.int64))], scope)
            return g.node("Transpose", [x], scope, perm=[1, 0, 2])
        # (q * scaling after the projection, as F.multi_head_attention_forward does)
        qh = heads(g.node("Mul", [proj(q, 0, "q"), g.init(scope + ".scaling", np.array(float(hd) ** -0.5, np.float32))], scope), Lq, "q")
        kh = heads(proj(k, 1, "k"), Lk, "k")
        vh = heads(proj(v, 2, "v"), Lk, "v")
        att = g.node("Softmax", [g.node("MatMul", [qh, g.node("Transpose", [kh], scope, perm=[0, 2, 1])], scope)], scope, axis=2)
        o = g.node("Transpose", [g.node("MatMul", [att, vh], scope)], scope, perm=[1, 0, 2])          # [L, B * h, hd]
        o = g.node("Reshape", [o, g.init(scope + ".merge", np.array([Lq, -1, E], np.int64))], scope)
        return self.linear(m.out_proj, o, scope + ".out_proj")

    def ffn(self, layer, x, scope):
        return self.linear(layer.linear2, self.linear(layer.linear1, x, scope + ".linear1", relu=True), scope + ".linear2")

    def encoder_layer(self, layer, src, pos, Ls, scope):
        g = self.g
        add = lambda a, b: g.node("Add", [a, b], scope)
        if layer.normalize_before:       # forward_pre (detr_backbone.py:169-181)
            s2 = self.layer_norm(layer.norm1, src, scope + ".norm1")
            qk = add(s2, pos)
            src = add(src, self.mha(layer.self_attn, qk, qk, s2, Ls, Ls, scope + ".self_attn"))
            return add(src, self.ffn(layer, self.layer_norm(layer.norm2, src, scope + ".norm2"), scope))
        qk = add(src, pos)               # forward_post (:155-167)
        src = self.layer_norm(layer.norm1, add(src, self.mha(layer.self_attn, qk, qk, src, Ls, Ls, scope + ".self_attn")), scope + ".norm1")
        return self.layer_norm(layer.norm2, add(src, self.ffn(layer, src, scope)), scope + ".norm2")

    def decoder_layer(self, layer, tgt, memory, pos, qpos, Lq, Ls, scope):
        g = self.g
        add = lambda a, b: g.node("Add", [a, b], scope)
        mem_k = add(memory, pos)
        if layer.normalize_before:       # forward_pre (:245-266)
            t2 = self.layer_norm(layer.norm1, tgt, scope + ".norm1")
            qk = add(t2, qpos)
            tgt = add(tgt, self.mha(layer.self_attn, qk, qk, t2, Lq, Lq, scope + ".self_attn"))
            t2 = self.layer_norm(layer.norm2, tgt, scope + ".norm2")
            tgt = add(tgt, self.mha(layer.multihead_attn, add(t2, qpos), mem_k, memory, Lq, Ls, scope + ".multihead_attn"))
            return add(tgt, self.ffn(layer, self.layer_norm(layer.norm3, tgt, scope + ".norm3"), scope))
        qk = add(tgt, qpos)              # forward_post (:222-243)
        tgt = self.layer_norm(layer.norm1, add(tgt, self.mha(layer.self_attn, qk, qk, tgt, Lq, Lq, scope + ".self_attn")), scope + ".norm1")
        tgt = self.layer_norm(layer.norm2, add(tgt, self.mha(layer.multihead_attn, add(tgt, qpos), mem_k, memory, Lq, Ls,
                                                             scope + ".multihead_attn")), scope + ".norm2")
        return self.layer_norm(layer.norm3, add(tgt, self.ffn(layer, tgt, scope)), scope + ".norm3")


def build_detr_graph(model, height, width):
    """The reference's Detr in export mode (meta_arch/detr.py:151-185 with onnx_export = True): input [batch, 3, H, W]
    ("N, CHW already permuted", :126-134), per-image normalisation, ResNet, the 1 x 1 input projection, the sine position
    embedding of an all-valid mask (a constant of H x W), the transformer (backbone/detr_backbone.py:25-266) on
    [H*W/1024, batch, 256] sequences, class and box heads of the LAST decoder layer (:458-459), then
    softmax[..., :-1].max, cxcywh -> xyxy and the concatenation [x0, y0, x1, y1, score, label] per query -> [batch, Q, 6].
    Static H x W (multiples of 32), dynamic batch."""
    if height % 32 or width % 32:
        raise ValueError("export: height and width must be multiples of 32")
    if getattr(model, "mask_on", False):
        raise NotImplementedError("export: DETR with the segmentation head (MODEL.MASK_ON)")
    g = OnnxGraph()
    e = _TransformerEmitter(g)
    detr = model.detr
    I64 = lambda name, v: g.init(name, np.array(v, np.int64))
    mean = g.init("preprocess.pixel_mean", _np(model.pixel_mean).reshape(1, 3, 1, 1))
    std = g.init("preprocess.pixel_std", _np(model.pixel_std).reshape(1, 3, 1, 1))
    x = g.node("Div", [g.node("Sub", ["images", mean], "preprocess"), std], "preprocess")
    masked = detr.backbone[0]
    feats = e.resnet(masked.backbone, x, "detr.backbone.0.backbone")
    last = list(masked.backbone.output_shape().keys())[-1]
    stride = masked.feature_strides[-1]
    Hf, Wf = height // stride, width // stride
    Ls, E = Hf * Wf, detr.transformer.d_model
    src = e.conv(detr.input_proj, feats[last], "detr.input_proj")
    seq = I64("detr.seq_shape", [0, E, -1])
    src = g.node("Transpose", [g.node("Reshape", [src, seq], "detr.transformer")], "detr.transformer", perm=[2, 0, 1])      # flatten(2).permute(2, 0, 1)
    pos = g.init("detr.pos_embed", np.ascontiguousarray(_sine_position_embedding(detr.backbone[1], Hf, Wf).reshape(1, E, Ls).transpose(2, 0, 1)))
    tr = detr.transformer
    mem = src
    for i, layer in enumerate(tr.encoder.layers):
        mem = e.encoder_layer(layer, mem, pos, Ls, f"detr.transformer.encoder.layers.{i}")
    if tr.encoder.norm is not None:
        mem = e.layer_norm(tr.encoder.norm, mem, "detr.transformer.encoder.norm")
    Q = detr.num_queries
    qpos = g.init("detr.query_embed", _np(detr.query_embed.weight)[:, None, :])                      # [Q, 1, E], broadcast over the batch
    # tgt = zeros_like(query_embed repeated over the batch): built from the memory's own batch dimension
    bdim = g.node("Slice", [g.node("Shape", [mem], "detr.transformer"), I64("detr.s1", [1]), I64("detr.s2", [2]), I64("detr.ax0", [0])], "detr.transformer")
    tshape = g.node("Concat", [I64("detr.q", [Q]), bdim, I64("detr.e", [E])], "detr.transformer", axis=0)
    tgt = g.node("Expand", [g.init("detr.zero", np.zeros((1, 1, 1), np.float32)), tshape], "detr.transformer")
    for i, layer in enumerate(tr.decoder.layers):
        tgt = e.decoder_layer(layer, tgt, mem, pos, qpos, Q, Ls, f"detr.transformer.decoder.layers.{i}")
    hs = e.layer_norm(tr.decoder.norm, tgt, "detr.transformer.decoder.norm") if tr.decoder.norm is not None else tgt
    hs = g.node("Transpose", [hs], "detr", perm=[1, 0, 2])                                            # [B, Q, E]
    logits = e.linear(detr.class_embed, hs, "detr.class_embed")
    t = hs
    nl = len(detr.bbox_embed.layers)
    for i, lin in enumerate(detr.bbox_embed.layers):
        t = e.linear(lin, t, f"detr.bbox_embed.layers.{i}", relu=i < nl - 1)
    boxes = g.node("Sigmoid", [t], "detr.bbox_embed")
    # ---- meta_arch/detr.py:178-185
    sc = "export"
    nc1 = detr.class_embed.out_features
    prob = g.node("Softmax", [logits], sc, axis=2)
    prob = g.node("Slice", [prob, I64(sc + ".p0", [0]), I64(sc + ".p1", [nc1 - 1]), I64(sc + ".pax", [2])], sc)       # [:, :, :-1]
    scores = g.node("ReduceMax", [prob], sc, axes=[2], keepdims=1)
    labels = g.node("Cast", [g.node("ArgMax", [prob], sc, axis=2, keepdims=1)], sc, to=FLOAT)
    cx, cy, w, h = g.node("Split", [boxes], sc, nout=4, axis=2, split=[1, 1, 1, 1])                   # box_cxcywh_to_xyxy (utils/boxes.py:28-32)
    half = g.init(sc + ".half", np.array(0.5, np.float32))
    hw_, hh_ = g.node("Mul", [w, half], sc), g.node("Mul", [h, half], sc)
    xyxy = [g.node("Sub", [cx, hw_], sc), g.node("Sub", [cy, hh_], sc), g.node("Add", [cx, hw_], sc), g.node("Add", [cy, hh_], sc)]
    g.node("Concat", xyxy + [scores, labels], sc, outputs=["outs"], axis=2)
    return g, [("images", FLOAT, ["batch", 3, height, width])], [("outs", FLOAT, ["batch", Q, 6])]


def export_detr_onnx(model, f, height=800, width=1344):
    """export.py:237-303 for a `detr` config.  The reference's `get_model_infos` returns a bare 3-list for that branch where
    export.py unpacks (input_names, output_names, dynamic_axes), so its export cannot run as written and fixes no names:
    input "images" [batch, 3, H, W] (the layout meta_arch/detr.py:126-134 documents), ONE output "outs" [batch, Q, 6] - the
    tensor the export branch returns (boxes xyxy in [0, 1], score, label)."""
    if model.training:
        raise RuntimeError("export_detr_onnx: an inference graph - call model.eval() first")
    g, ins, outs = build_detr_graph(model, height, width)
    return _write(g.serialize(ins, outs), f)


def get_model_infos(config_file):
    """export.py:237-249 with the intent of its three branches: (input_names, output_names, dynamic_axes)"""
    if "sparse_inst" in config_file:
        return ["images"], ["masks", "scores", "labels"], {"images": {0: "batch"}}
    if "detr" in config_file:
        return ["images"], ["outs"], {"images": {0: "batch"}}
    return ["images"], ["outs"], {"images": {0: "batch"}}


def export_onnx(model, f, height, width):
    """the exporter of the model's meta-architecture (YOLOX / SparseInst / Detr)"""
    name = type(model).__name__
    fn = {"YOLOX": export_yolox_onnx, "SparseInst": export_sparseinst_onnx, "Detr": export_detr_onnx}.get(name)
    if fn is None:
        raise NotImplementedError(f"export_onnx: no exporter for meta-architecture {name}")
    return fn(model, f, height, width)
