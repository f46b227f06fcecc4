"""TEST INFRASTRUCTURE: a protobuf reader for ONNX ModelProto files and an interpreter for the handful of operators the YOLOX
export uses (torch ops on CPU), so that an exported file can be EXECUTED in an image that has neither `onnx` nor
`onnxruntime`.  The reader / interpreter pair is itself checked against files torch's own exporter writes
(tests/test_export_onnx.py)."""
import struct

import math

import numpy as np
import torch
import torch.nn.functional as F


def _rv(b, p):
    v = s = 0
    while True:
        c = b[p]
        p += 1
        v |= (c & 127) << s
        s += 7
        if c < 128:
            return v, p


def fields(b):
    p, out = 0, []
    while p < len(b):
        k, p = _rv(b, p)
        f, w = k >> 3, k & 7
        if w == 0:
            v, p = _rv(b, p)
        elif w == 2:
            n, p = _rv(b, p)
            v = bytes(b[p: p + n])
            p += n
        elif w == 5:
            v = struct.unpack("<f", b[p: p + 4])[0]
            p += 4
        elif w == 1:
            v = struct.unpack("<d", b[p: p + 8])[0]
            p += 8
        else:
            raise ValueError("wire type %d" % w)
        out.append((f, w, v))
    return out


def _sint(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def tensor(b):
    dims, dt, name, raw, f32, i64 = [], None, "", None, [], []
    for f, w, v in fields(b):
        if f == 1:
            dims += [_sint(x) for x in ([v] if w == 0 else [t[2] for t in _packed(v)])]
        elif f == 2:
            dt = v
        elif f == 8:
            name = v.decode()
        elif f == 9:
            raw = v
        elif f == 4:
            f32 += [v] if w == 5 else list(struct.unpack("<%df" % (len(v) // 4), v))
        elif f == 7:
            i64 += [_sint(v)] if w == 0 else [_sint(t[2]) for t in _packed(v)]
    np_dt = {1: np.float32, 7: np.int64, 6: np.int32, 9: np.bool_, 11: np.float64}[dt]
    if raw is not None:
        a = np.frombuffer(raw, np_dt).copy()
    else:
        a = np.array(f32 if dt == 1 else i64, np_dt)
    return name, a.reshape(dims)


def _packed(v):
    p, out = 0, []
    while p < len(v):
        x, p = _rv(v, p)
        out.append((0, 0, x))
    return out


def attribute(b):
    name, val, ints, floats = None, None, [], []
    for f, w, v in fields(b):
        if f == 1:
            name = v.decode()
        elif f == 2:
            val = v
        elif f == 3:
            val = _sint(v)
        elif f == 4:
            val = v.decode()
        elif f == 5:
            val = tensor(v)[1]
        elif f == 8:
            ints += [_sint(v)] if w == 0 else [_sint(t[2]) for t in _packed(v)]
        elif f == 7:
            floats += [v] if w == 5 else list(struct.unpack("<%df" % (len(v) // 4), v))
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def load(data):
    """-> dict(nodes=[(op, inputs, outputs, attrs)], inits={name: array}, inputs=[names], outputs=[names], opset, ir_version)"""
    m = dict(nodes=[], inits={}, inputs=[], outputs=[], opset=None, ir_version=None)
    for f, w, v in fields(data):
        if f == 1:
            m["ir_version"] = v
        elif f == 8:
            for f2, _, v2 in fields(v):
                if f2 == 2:
                    m["opset"] = v2
        elif f == 7:
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    ins, outs, op, attrs = [], [], None, {}
                    for f3, _, v3 in fields(v2):
                        if f3 == 1:
                            ins.append(v3.decode())
                        elif f3 == 2:
                            outs.append(v3.decode())
                        elif f3 == 4:
                            op = v3.decode()
                        elif f3 == 5:
                            k, a = attribute(v3)
                            attrs[k] = a
                    m["nodes"].append((op, ins, outs, attrs))
                elif f2 == 5:
                    n, a = tensor(v2)
                    m["inits"][n] = a
                elif f2 in (11, 12):
                    nm = [v3.decode() for f3, _, v3 in fields(v2) if f3 == 1][0]
                    m["inputs" if f2 == 11 else "outputs"].append(nm)
    m["inputs"] = [n for n in m["inputs"] if n not in m["inits"]]
    return m


def _t(v):
    v = np.asarray(v)
    return torch.from_numpy(v.copy() if v.ndim == 0 else np.ascontiguousarray(v))      # (ascontiguousarray makes a 0-d array 1-d)


def run(model, feeds, trace=None):
    env = {k: _t(v) for k, v in model["inits"].items()}
    env.update({k: torch.as_tensor(v) for k, v in feeds.items()})
    for op, ins, outs, a in model["nodes"]:
        x = [env[i] if i else None for i in ins]
        if trace is not None:
            trace.append((op, ins, outs, a, [tuple(t.shape) if t is not None else None for t in x]))
        if op == "Conv":
            p = a.get("pads", [0, 0, 0, 0])
            assert p[0] == p[2] and p[1] == p[3] and a.get("dilations", [1, 1]) == [1, 1]
            y = F.conv2d(x[0], x[1], x[2] if len(x) > 2 else None, stride=a.get("strides", [1, 1]), padding=(p[0], p[1]), groups=a.get("group", 1))
        elif op == "Sigmoid":
            y = torch.sigmoid(x[0])
        elif op == "Mul":
            y = x[0] * x[1]
        elif op == "Add":
            y = x[0] + x[1]
        elif op == "Exp":
            y = torch.exp(x[0])
        elif op == "Concat":
            y = torch.cat(x, a["axis"])
        elif op == "Transpose":
            y = x[0].permute(*a["perm"])
        elif op == "MaxPool":
            k, p = a["kernel_shape"], a["pads"]
            assert p[0] == p[2] and p[1] == p[3] and not a.get("ceil_mode", 0)
            y = F.max_pool2d(x[0], k, a.get("strides", [1, 1]), (p[0], p[1]))
        elif op == "Slice":
            y = x[0]
            starts, ends = x[1].tolist(), x[2].tolist()
            axes = x[3].tolist() if len(x) > 3 and x[3] is not None else list(range(len(starts)))
            steps = x[4].tolist() if len(x) > 4 and x[4] is not None else [1] * len(starts)
            for s, e_, ax, st in zip(starts, ends, axes, steps):
                n = y.shape[ax]
                s = max(0, min(n, s + n if s < 0 else s))
                e_ = max(0, min(n, e_ + n if e_ < 0 else e_))
                y = y.index_select(ax, torch.arange(s, e_, st))
        elif op == "Resize" and a["mode"] == "linear":
            # align_corners=False bilinear: "pytorch_half_pixel" (what torch's exporter writes) equals "half_pixel" unless an
            # output dimension is 1
            assert a["coordinate_transformation_mode"] in ("pytorch_half_pixel", "half_pixel")
            if len(x) > 3 and x[3] is not None and x[3].numel():
                size = [int(v) for v in x[3].tolist()]
                assert size[:2] == list(x[0].shape[:2])
                size = size[2:]
            else:
                sc = x[2].tolist()
                assert sc[:2] == [1.0, 1.0]
                size = [int(math.floor(x[0].shape[2] * sc[2])), int(math.floor(x[0].shape[3] * sc[3]))]
            assert a["coordinate_transformation_mode"] == "pytorch_half_pixel" or min(size) > 1
            y = F.interpolate(x[0], size=size, mode="bilinear", align_corners=False)
        elif op == "Resize":
            assert a["mode"] == "nearest" and a["coordinate_transformation_mode"] == "asymmetric" and a.get("nearest_mode", "round_prefer_floor") == "floor"
            if len(x) > 3 and x[3] is not None and x[3].numel():
                size = [int(v) for v in x[3].tolist()][2:]
                sc = [1.0, 1.0, size[0] / x[0].shape[2], size[1] / x[0].shape[3]]
            else:
                sc = x[2].tolist()
                assert sc[:2] == [1.0, 1.0]
                size = [int(math.floor(x[0].shape[2] * sc[2])), int(math.floor(x[0].shape[3] * sc[3]))]
            iy = torch.clamp(torch.floor(torch.arange(size[0]) / sc[2]).long(), max=x[0].shape[2] - 1)     # x_orig = x_out / scale, floor
            ix = torch.clamp(torch.floor(torch.arange(size[1]) / sc[3]).long(), max=x[0].shape[3] - 1)
            y = x[0].index_select(2, iy).index_select(3, ix)
        elif op == "Reshape":
            shp = [x[0].shape[i] if d == 0 else d for i, d in enumerate(x[1].tolist())]
            y = x[0].reshape(shp)
        elif op == "Split":
            y = list(torch.split(x[0], a["split"], a["axis"]))
        elif op == "ArgMax":
            y = torch.argmax(x[0], a["axis"], keepdim=bool(a.get("keepdims", 1)))
        elif op == "Cast":
            y = x[0].to({1: torch.float32, 7: torch.int64, 9: torch.bool, 6: torch.int32}[a["to"]])
        elif op == "Constant":
            y = _t(a["value"])
        elif op == "Identity":
            y = x[0]
        elif op == "Shape":
            y = torch.tensor(list(x[0].shape), dtype=torch.int64)
        elif op == "Unsqueeze":
            y = x[0]
            for ax in sorted(a["axes"]):
                y = y.unsqueeze(ax)
        elif op == "Expand":
            y = x[0] * torch.ones([int(v) for v in x[1].tolist()], dtype=x[0].dtype)
        elif op == "Gather":
            idx = x[1].reshape(-1)
            idx = torch.where(idx < 0, idx + x[0].shape[a.get("axis", 0)], idx)          # (negative indices count from the end)
            y = torch.index_select(x[0], a.get("axis", 0), idx).reshape(
                x[0].shape[: a.get("axis", 0)] + tuple(x[1].shape) + x[0].shape[a.get("axis", 0) + 1:])
        elif op == "Relu":
            y = torch.relu(x[0])
        elif op == "Sub":
            y = x[0] - x[1]
        elif op == "Div":
            y = x[0] / x[1]
        elif op == "Sqrt":
            y = torch.sqrt(x[0])
        elif op == "MatMul":
            y = torch.matmul(x[0], x[1])
        elif op == "Softmax":
            assert a.get("axis", 1) in (-1, x[0].dim() - 1)      # (opset 11 flattens to 2-D at `axis`: identical for the last axis)
            y = torch.softmax(x[0], -1)
        elif op == "Clip":
            y = torch.clamp(x[0], min=None if len(x) < 2 or x[1] is None else float(x[1]), max=None if len(x) < 3 or x[2] is None else float(x[2]))
        elif op == "AveragePool":
            k, p = a["kernel_shape"], a.get("pads", [0, 0, 0, 0])
            assert p == [0, 0, 0, 0] and not a.get("ceil_mode", 0)
            y = F.avg_pool2d(x[0], k, a.get("strides", [1, 1]))
        elif op in ("ReduceSum", "ReduceMax", "ReduceMean"):
            ax, kd = a["axes"], bool(a.get("keepdims", 1))
            y = x[0].sum(ax, keepdim=kd) if op == "ReduceSum" else x[0].mean(ax, keepdim=kd) if op == "ReduceMean" else x[0].amax(ax, keepdim=kd)
        elif op == "TopK":
            assert a.get("largest", 1) == 1 and a.get("sorted", 1) == 1
            v, i = torch.topk(x[0], int(x[1].reshape(-1)[0]), dim=a.get("axis", -1))
            y = [v, i]
        elif op == "Greater":
            y = x[0] > x[1]
        elif op == "CumSum":
            assert not a.get("exclusive", 0) and not a.get("reverse", 0)
            y = torch.cumsum(x[0], int(x[1].item()))
        elif op == "Sin":
            y = torch.sin(x[0])
        elif op == "Cos":
            y = torch.cos(x[0])
        elif op == "Tile":
            y = x[0].repeat(*[int(v) for v in x[1].tolist()])
        elif op == "Flatten":
            ax = a.get("axis", 1)
            y = x[0].reshape(int(np.prod(x[0].shape[:ax])) if ax else 1, -1)
        elif op == "Gemm":
            A = x[0].t() if a.get("transA", 0) else x[0]
            Bm = x[1].t() if a.get("transB", 0) else x[1]
            y = a.get("alpha", 1.0) * (A @ Bm)
            if len(x) > 2 and x[2] is not None:
                y = y + a.get("beta", 1.0) * x[2]
        elif op == "Less":
            y = x[0] < x[1]
        elif op == "Equal":
            y = x[0] == x[1]
        elif op == "Not":
            y = ~x[0]
        elif op == "And":
            y = x[0] & x[1]
        elif op == "Where":
            y = torch.where(x[0], x[1], x[2])
        elif op == "Neg":
            y = -x[0]
        elif op == "Floor":
            y = torch.floor(x[0])
        elif op == "Min":
            y = torch.minimum(x[0], x[1])
        elif op == "Max":
            y = torch.maximum(x[0], x[1])
        elif op == "Range":
            y = torch.arange(x[0].item(), x[1].item(), x[2].item(), dtype=x[0].dtype)
        elif op == "Pow":
            y = torch.pow(x[0], x[1])
        elif op == "Erf":
            y = torch.erf(x[0])
        elif op == "Tanh":
            y = torch.tanh(x[0])
        elif op == "Squeeze":
            y = x[0]
            for ax in sorted(a["axes"], reverse=True):
                y = y.squeeze(ax)
        elif op == "ConstantOfShape":
            v = a.get("value")
            y = torch.full([int(t) for t in x[0].tolist()], float(v.reshape(-1)[0]) if v is not None else 0.0,
                           dtype=torch.from_numpy(np.ascontiguousarray(v)).dtype if v is not None else torch.float32)
        else:
            raise NotImplementedError(op)
        for name, v in zip(outs, y if isinstance(y, list) else [y]):
            env[name] = v
            if trace is not None and v.is_floating_point() and bool(torch.isnan(v).any()) and not any(t[0] == "NaN" for t in trace):
                trace.append(("NaN", ins, outs, a, [op], dict(env)))
    return [env[o].numpy() for o in model["outputs"]]
