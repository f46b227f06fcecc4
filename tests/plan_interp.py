"""TEST INFRASTRUCTURE: a CPU interpreter of the symbolic step plan (yolov7_d2_amd.plan.PlanBuilder).

It executes the forward / backward command lists with torch CPU ops over byte buffers laid out exactly as
the plan lays them out (bf16 NHWC views, concat slices, gradient mirrors, packed weight images, tap
tables, accumulate flags).  It checks the HOST LOGIC of the product (graph wiring, buffer aliasing,
gradient fan-in, dgrad parity classes, weight packing) against the oracle without a GPU; the kernels
themselves are checked on the GPU by the `-m gpu` tests.  Lives under tests/: never imported by the
product.
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import yolox_oracle as O  # noqa: E402

from yolov7_d2_amd import _lib as L  # noqa: E402
from yolov7_d2_amd.plan import Buf, TRef  # noqa: E402

OPN = {v: k for k, v in L.OP.items()}


class Interp:
    def __init__(self, builder, dtype=torch.bfloat16):
        """dtype=torch.float32 stores activations / weight images in fp32 (same element strides): separates
        host-logic errors from bf16 storage noise."""
        self.b = builder
        self.dt = dtype
        self.mul = 2 if dtype == torch.float32 else 1
        self.store = {}
        self.spp_idx = {}

    # ---- storage
    def raw(self, buf):
        s = self.store.get(id(buf))
        if s is None or s.numel() < buf.nbytes * self.mul:
            s = torch.zeros(buf.nbytes * self.mul + 64, dtype=torch.uint8)
            self.store[id(buf)] = s
        return s

    def tv(self, t, C=None):
        """bf16 NHWC view [N,H,W,C] of a TRef"""
        C = t.C if C is None else C
        flat = self.raw(t.buf).view(self.dt)
        return flat.as_strided((t.N, t.H, t.W, C), (t.H * t.W * t.ld, t.W * t.ld, t.ld, 1), t.coff)

    def f32(self, p, n, off_bytes=0):
        o = p.obj
        if o is None:
            return None
        if isinstance(o, torch.Tensor):
            return o.view(-1)[:n] if p.off == 0 else o.view(-1)[p.off // 4: p.off // 4 + n]
        assert isinstance(o, Buf), type(o)
        return self.raw(o).view(torch.float32)[(p.off + off_bytes) // 4: (p.off + off_bytes) // 4 + n]

    def f64(self, p, n):
        assert isinstance(p.obj, Buf) and p.off % 8 == 0
        return self.raw(p.obj).view(torch.float64)[p.off // 8: p.off // 8 + n]

    def rawbytes(self, p, n):
        o = p.obj
        if isinstance(o, torch.Tensor):
            return o.view(torch.uint8).view(-1)[p.off: p.off + n]
        return self.raw(o)[p.off: p.off + n]

    # ---- ops
    def run(self, cmds):
        for c in cmds:
            getattr(self, "op_" + OPN[c.op])(c)

    def op_NOP(self, c):
        pass

    def op_MEMSET(self, c):
        self.rawbytes(c.p[0], c.l[0]).fill_(c.i[0])

    def op_PACK_W(self, c):
        Cout, Cin, KH, KW, CinPad, CoutPad, CoutPadK, CinPadN = c.i[:8]
        w = c.p[0].obj.detach().float().reshape(Cout, Cin, KH * KW)
        KK = KH * KW
        if c.p[1].obj is not None:  # wf[tap][ci/8][co][ci%8]
            img = torch.zeros(KK, CinPad // 8, CoutPad, 8)
            wp = torch.zeros(KK, CoutPad, CinPad)
            wp[:, :Cout, :Cin] = w.permute(2, 0, 1)
            img[:] = wp.view(KK, CoutPad, CinPad // 8, 8).permute(0, 2, 1, 3)
            o = c.p[1].off // 2
            self.raw(c.p[1].obj).view(self.dt)[o: o + img.numel()] = img.reshape(-1).to(self.dt)
        if c.p[2].obj is not None:  # wd[tap][co/8][ci][co%8]
            wp = torch.zeros(KK, CinPadN, CoutPadK)
            wp[:, :Cin, :Cout] = w.permute(2, 1, 0)
            img = wp.view(KK, CinPadN, CoutPadK // 8, 8).permute(0, 2, 1, 3).contiguous()
            o = c.p[2].off // 2      # (a CSP pair's two data-gradient images are row ranges of one buffer)
            self.raw(c.p[2].obj).view(self.dt)[o: o + img.numel()] = img.reshape(-1).to(self.dt)

    def _conv_core(self, s):
        x = s.x.obj
        K = s.K8 * 8
        xin = self.tv(x, K).float() if not isinstance(x, Buf) else None
        N, H, W = s.N, s.H, s.W
        nslab = max(t[2] for t in s.taps) + 1
        wo = s.w.off // 2
        wimg = self.raw(s.w.obj).view(self.dt)[wo: wo + nslab * s.K8 * s.CoutPad * 8].float().view(nslab, s.K8, s.CoutPad, 8)
        wt = wimg.permute(0, 2, 1, 3).reshape(nslab, s.CoutPad, K)
        P = 4
        xp = F.pad(xin, (0, 0, P, P + 2 * s.gridW, P, P + 2 * s.gridH))
        acc = torch.zeros(N, s.gridH, s.gridW, s.CoutPad)
        for (dy, dx, ws) in s.taps:
            sl = xp[:, P + dy: P + dy + s.gridH * s.in_stride: s.in_stride, P + dx: P + dx + s.gridW * s.in_stride: s.in_stride, :]
            acc += torch.einsum("nhwk,ck->nhwc", sl, wt[ws])
        return acc

    def op_CONV(self, c):
        s = c.desc
        acc = self._conv_core(s)
        Cout = s.Cout
        res = acc[..., :Cout].clone()
        bias = s.bias.obj
        if bias is not None:
            res += bias.detach().float().view(1, 1, 1, -1)
        ys = slice(s.out_oy, s.out_oy + s.gridH * s.out_stride, s.out_stride)
        xs = slice(s.out_ox, s.out_ox + s.gridW * s.out_stride, s.out_stride)
        if s.flags & L.MI_CONV_OUT_F32:
            nstr = s.y_nstride if s.y_nstride else s.outH * s.outW * s.ldy
            base = self.raw(s.y.obj).view(torch.float32)
            yv = base.as_strided((s.N, s.outH, s.outW, Cout), (nstr, s.outW * s.ldy, s.ldy, 1), s.y.off // 4)
        else:
            t = s.y.obj
            assert isinstance(t, TRef)
            yv = self.tv(t, Cout)
        if s.flags & L.MI_CONV_ACCUM:
            res = res + yv[:, ys, xs, :].float()
        yv[:, ys, xs, :] = res.to(yv.dtype)
        if s.flags & L.MI_CONV_BNBWD:   # BatchNorm-backward sums of the layer that produced this gradient's tensor
            b = s.bnb
            C = Cout
            yy = self.tv(b["y"].obj).float()[:, ys, xs, :]
            z = yy * self.f32(b["scale"], C) + self.f32(b["shift"], C)
            sg = torch.sigmoid(z)
            g = sg * (1 + z * (1 - sg)) if b["act"] else torch.ones_like(z)
            dz = yv[:, ys, xs, :].float() * g
            xh = (yy - self.f32(b["mean"], C)) * self.f32(b["invstd"], C)
            st = self.f64(s.stats, C * 2).view(C, 2)
            st[:, 0] += dz.double().sum((0, 1, 2))
            st[:, 1] += (dz * xh).double().sum((0, 1, 2))
        elif s.stats.obj is not None:   # fp64 accumulators [SLOTS][CoutPad][2]: the interpreter adds everything to slot 0
            st = self.f64(s.stats, s.CoutPad * 2).view(s.CoutPad, 2)
            stored = yv[:, ys, xs, :].float() if not (s.flags & L.MI_CONV_OUT_F32) else res
            st[:Cout, 0] += stored.double().sum((0, 1, 2))
            st[:Cout, 1] += (stored.double() ** 2).sum((0, 1, 2))

    def op_WGRAD(self, c):
        s = c.desc
        x, dyT = s.x.obj, s.dy.obj
        xin = self.tv(x, s.CinPad).float()
        dy = self.tv(dyT, s.CoutPad).float()
        P = 4
        xp = F.pad(xin, (0, 0, P, P + 2 * s.outW, P, P + 2 * s.outH))
        KK = len(s.taps)
        g = s.gw.obj   # fp32 OIHW gradient tensor (overwritten)
        kk = int(round(KK ** 0.5))
        res = torch.zeros(s.Cout, s.Cin, KK)
        for t, (ty, tx) in enumerate(s.taps):
            sl = xp[:, P + ty: P + ty + s.outH * s.stride: s.stride, P + tx: P + tx + s.outW * s.stride: s.stride, :]
            res[:, :, t] = torch.einsum("nhwo,nhwi->oi", dy, sl)[: s.Cout, : s.Cin]
        g.copy_(res.view(s.Cout, s.Cin, kk, kk))

    # ---- depthwise 3x3 (DWConv.dconv)
    def _dw_w(self, p, C):
        return p.obj.detach().float().to(self.dt).float().view(C, 1, 3, 3)   # rounded like every conv operand

    def op_DWCONV_FWD(self, c):
        ldx, ldy, N, H, W, C, stride, Ho, Wo, nsl = c.i[:10]
        x, y = c.p[0].obj, c.p[2].obj
        xin = self.tv(x, C).float().permute(0, 3, 1, 2)
        res = F.conv2d(xin, self._dw_w(c.p[1], C), None, stride=stride, padding=1, groups=C).permute(0, 2, 3, 1)
        assert tuple(res.shape) == (N, Ho, Wo, C)
        self.tv(y, C)[:] = res.to(self.dt)
        if c.p[3].obj is not None:
            CA = (C + 31) // 32 * 32
            st = self.f64(c.p[3], CA * 2).view(CA, 2)
            stored = self.tv(y, C).float()
            st[:C, 0] += stored.double().sum((0, 1, 2))
            st[:C, 1] += (stored.double() ** 2).sum((0, 1, 2))

    def op_DWCONV_DGRAD(self, c):
        lddy, lddx, N, H, W, C, stride, Ho, Wo, acc = c.i[:10]
        dy = self.tv(c.p[0].obj, C).float().permute(0, 3, 1, 2)
        dxv = self.tv(c.p[2].obj, C)
        xz = torch.zeros(N, C, H, W, requires_grad=True)
        F.conv2d(xz, self._dw_w(c.p[1], C), None, stride=stride, padding=1, groups=C).backward(dy)
        g = xz.grad.permute(0, 2, 3, 1)
        dxv[:] = (g + (dxv.float() if acc else 0)).to(self.dt)

    def op_DWCONV_WGRAD(self, c):
        ldx, lddy, N, H, W, C, stride, Ho, Wo = c.i[:9]
        x = self.tv(c.p[0].obj, C).float().permute(0, 3, 1, 2)
        dy = self.tv(c.p[1].obj, C).float().permute(0, 3, 1, 2)
        wz = torch.zeros(C, 1, 3, 3, requires_grad=True)
        F.conv2d(x, wz, None, stride=stride, padding=1, groups=C).backward(dy)
        c.p[3].obj.copy_(wz.grad)

    def op_BN_EVAL_AFFINE(self, c):
        C = c.i[0]
        g, b, rm, rv = (c.p[k].obj.detach().float() for k in range(4))
        inv = 1.0 / torch.sqrt(rv + c.f[0])
        self.f32(c.p[4], C).copy_(g * inv)
        self.f32(c.p[5], C).copy_(b - rm * g * inv)

    def op_BN_ACT_FWD(self, c):
        y, res, a = c.p[0].obj, c.p[11].obj, c.p[12].obj
        C, act = c.i[3], c.i[4]
        if c.p[1].obj is not None:   # train mode: finalize the statistics first
            count = c.l[0]
            nsl = c.i[5]
            CA = (C + 31) // 32 * 32   # accumulator layout [slot][C rounded up to 32][2] (bn_act.hip BN_ACC_C)
            part = self.f64(c.p[1], nsl * CA * 2).view(nsl, CA, 2).sum(0)[:C]
            mean = part[:, 0] / count
            var = (part[:, 1] / count - mean * mean).clamp(min=0)
            invstd = 1.0 / torch.sqrt(var + c.f[0])
            gamma, beta = c.p[2].obj.detach().double(), c.p[3].obj.detach().double()
            self.f32(c.p[7], C).copy_((gamma * invstd).float())
            self.f32(c.p[8], C).copy_((beta - mean * gamma * invstd).float())
            self.f32(c.p[9], C).copy_(mean.float())
            self.f32(c.p[10], C).copy_(invstd.float())
            m = c.f[1]
            rm, rv, nbt = c.p[4].obj, c.p[5].obj, c.p[6].obj
            if rm is not None:
                rm.mul_(1 - m).add_(m * mean.float())
                rv.mul_(1 - m).add_(m * (var * count / max(count - 1, 1)).float())
                nbt += 1
        z = self.tv(y).float() * self.f32(c.p[7], C) + self.f32(c.p[8], C)
        o = z * torch.sigmoid(z) if act else z
        if res is not None:
            o = o + self.tv(res).float()
        self.tv(a)[:] = o.to(self.dt)

    def _dz(self, c, da, y, C, act):
        yy = self.tv(y).float()
        z = yy * self.f32(c.p[2], C) + self.f32(c.p[3], C)
        s = torch.sigmoid(z)
        g = s * (1 + z * (1 - s)) if act else torch.ones_like(z)
        dz = self.tv(da).float() * g
        xh = (yy - self.f32(c.p[4], C)) * self.f32(c.p[5], C)
        return dz, xh

    def op_BN_BWD_REDUCE(self, c):
        C, act = c.i[3], c.i[4]
        dz, xh = self._dz(c, c.p[0].obj, c.p[1].obj, C, act)
        part = self.f64(c.p[6], ((C + 31) // 32 * 32) * 2).view(-1, 2)[:C]   # slot 0, [C rounded up to 32][2]
        part[:, 0] += dz.double().sum((0, 1, 2))
        part[:, 1] += (dz * xh).double().sum((0, 1, 2))

    def op_BN_BWD_APPLY(self, c):
        C, act = c.i[5], c.i[6]
        count = c.l[1]
        da = c.p[0].obj
        dz, xh = self._dz(c, da, c.p[1].obj, C, act)
        nsl = c.i[7]
        CA = (C + 31) // 32 * 32
        part = self.f64(c.p[7], nsl * CA * 2).view(nsl, CA, 2).sum(0)[:C]
        if c.p[8].obj is not None:
            c.p[8].obj.copy_(part[:, 1].float())
        if c.p[9].obj is not None:
            c.p[9].obj.copy_(part[:, 0].float())
        c1, c2 = (part[:, 0] / count).float(), (part[:, 1] / count).float()
        gamma = c.p[6].obj.detach().float()
        dy = gamma * self.f32(c.p[5], C) * (dz - c1 - xh * c2)
        self.tv(c.p[10].obj)[..., :C] = dy.to(self.dt)   # the out-gradient view carries zero pad channels up to 32
        dres = c.p[11].obj
        if dres is not None:
            v = self.tv(da).float()
            if c.i[4]:
                v = v + self.tv(dres).float()
            self.tv(dres)[:] = v.to(self.dt)

    def op_FOCUS(self, c):
        img = c.p[0].obj.float()
        out = c.p[1].obj
        tl, tr = img[..., ::2, ::2], img[..., ::2, 1::2]
        bl, br = img[..., 1::2, ::2], img[..., 1::2, 1::2]
        cat = torch.cat((tl, bl, tr, br), 1).permute(0, 2, 3, 1)
        v = self.tv(out)
        v.zero_()
        v[..., :12] = cat.to(self.dt)

    def op_UPSAMPLE_FWD(self, c):
        x, y = c.p[0].obj, c.p[1].obj
        self.tv(y)[:] = self.tv(x).repeat_interleave(2, 1).repeat_interleave(2, 2)

    def op_UPSAMPLE_BWD(self, c):
        dy, dx = c.p[0].obj, c.p[1].obj
        g = self.tv(dy).float()
        v = g[:, 0::2, 0::2] + g[:, 0::2, 1::2] + g[:, 1::2, 0::2] + g[:, 1::2, 1::2]
        if c.i[2]:
            v = v + self.tv(dx).float()
        self.tv(dx)[:] = v.to(self.dt)

    def op_SPP_FWD(self, c):
        x = c.p[0].obj
        xin = self.tv(x).float().permute(0, 3, 1, 2)
        idxs = []
        for k, o in zip((5, 9, 13), (c.p[1].obj, c.p[2].obj, c.p[3].obj)):
            v, idx = F.max_pool2d(xin, k, 1, k // 2, return_indices=True)
            self.tv(o)[:] = v.permute(0, 2, 3, 1).to(self.dt)
            idxs.append(idx)
        self.spp_idx[id(c.p[4].obj)] = idxs

    def op_SPP_BWD(self, c):
        idxs = self.spp_idx[id(c.p[3].obj)]
        dx = c.p[4].obj
        acc = None
        for k, d, idx in zip((5, 9, 13), (c.p[0].obj, c.p[1].obj, c.p[2].obj), idxs):
            g = self.tv(d).float().permute(0, 3, 1, 2)
            n_, c_, h_, w_ = g.shape  # stride-1 windows overlap: gradients to one argmax must ADD
            u = torch.zeros(n_, c_, h_ * w_).scatter_add_(2, idx.reshape(n_, c_, -1), g.reshape(n_, c_, -1))
            u = u.view(n_, c_, h_, w_)
            acc = u if acc is None else acc + u
        v = acc.permute(0, 2, 3, 1)
        if c.i[2]:
            v = v + self.tv(dx).float()
        self.tv(dx)[:] = v.to(self.dt)

    def op_COPY(self, c):
        raise NotImplementedError

    def op_BIAS_GRADS(self, c):
        B, A, nch, n = c.i[:4]
        dp = self.f32(c.p[1], B * A * nch).view(B, A, nch)
        for j in c.desc.jobs:
            j["out"].copy_(dp[:, j["a0"]: j["a0"] + j["HW"], j["c0"]: j["c0"] + j["nc"]].sum((0, 1)).view_as(j["out"]))

    def op_COLSUM(self, c):
        x, out = c.p[0].obj, c.p[1].obj
        C = c.i[1]
        v = self.tv(x, C).float().sum((0, 1, 2))
        if c.i[2]:
            out.add_(v.view_as(out))
        else:
            out.copy_(v.view_as(out))

    def _loss_inputs(self, s):
        nch = 5 + s.ncls
        preds = self.f32(s.preds, s.B * s.A * nch).view(s.B, s.A, nch)
        return preds, s.labels.obj, s.anchors.obj

    def op_LOSS_FWD(self, c):
        s = c.desc
        preds, labels, anchors = self._loss_inputs(s)
        res, assigns = O.yolox_losses(preds.clone(), labels, anchors, s.ncls, return_assign=True,
                                      use_l1=getattr(s, "use_l1", False))
        out = self.raw(s.ws["out"]).view(torch.float32)
        nfg = sum(a["num_fg"] for a in assigns if a is not None)
        ngt = int((labels.sum(2) > 0).sum())
        out[:8] = torch.tensor([float(res[0]), float(res[1]), float(res[2]), float(res[3]), float(res[4]), float(res[5]),
                                float(nfg), float(ngt)])

    def op_LOSS_BWD(self, c):
        s = c.desc
        preds, labels, anchors = self._loss_inputs(s)
        raw = preds.clone().requires_grad_(True)
        l1 = getattr(s, "use_l1", False)
        res = O.yolox_losses(raw, labels, anchors, s.ncls, use_l1=l1)
        gw = self.f32(c.p[1], 5 if l1 else 4)
        tot = gw[0] * res[0] + gw[1] * res[1] + gw[2] * res[2] + gw[3] * res[3]
        if l1:
            tot = tot + gw[4] * res[4]
        tot.backward()
        self.f32(c.p[2], raw.numel()).copy_(raw.grad.reshape(-1))

    def op_SPLIT_DPREDS(self, c):
        B, A, nch, a0, HW, c0, nc, ld = c.i[:8]
        dp = self.f32(c.p[0], B * A * nch).view(B, A, nch)
        dst = c.p[1].obj
        v = self.tv(dst)
        v.zero_()
        v.view(B, HW, ld)[..., :nc] = dp[:, a0:a0 + HW, c0:c0 + nc].to(self.dt)

    def op_DECODE(self, c):
        B, A, ncls = c.i[:3]
        p = self.f32(c.p[0], B * A * (5 + ncls)).view(B, A, 5 + ncls)
        p.copy_(O.decode_eval(p.clone(), c.p[1].obj))
