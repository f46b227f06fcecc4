"""GPU (-m gpu): DETR set matching (cost matrix + linear sum assignment) through the C-ABI, against the oracle and
the golden vectors produced by the reference's own HungarianMatcher (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment   # test infrastructure: the checker

import detr_oracle as D
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.modeling import HungarianMatcher

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _to_dev(targets):
    return [dict(labels=t["labels"].to(DEV), boxes=t["boxes"].to(DEV)) for t in targets]


def test_hungarian_matcher_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "hungarian.npz"))
    for name, (bs, nq, seed, sizes) in dict(a=(3, 100, 41, None), b=(2, 100, 42, [100, 1]), c=(2, 16, 43, [30, 7])).items():
        logits, boxes, targets = D.synth_detr(bs, nq, 91, seed, sizes=sizes)
        m = HungarianMatcher(cost_class=1.0, cost_bbox=5.0, cost_giou=2.0)
        idx = m({"pred_logits": logits.to(DEV), "pred_boxes": boxes.to(DEV)}, _to_dev(targets))
        # cost matrix: float parity with the fp32 CPU restatement
        Cref = D.matching_cost(logits, boxes, targets, 1.0, 5.0, 2.0)
        off = 0
        for b, t in enumerate(targets):
            G = len(t["boxes"])
            np.testing.assert_allclose(m.last_cost[b, :, :G].cpu().numpy(), Cref[b, :, off:off + G].numpy(), rtol=1e-5, atol=1e-6)
            off += G
        # indices: bit-exact (int64, rows sorted) against the reference's scipy result
        for b, (i, j) in enumerate(idx):
            assert i.dtype == torch.int64 and j.dtype == torch.int64
            assert np.array_equal(i.cpu().numpy(), g[f"{name}_i{b}"]), (name, b)
            assert np.array_equal(j.cpu().numpy(), g[f"{name}_j{b}"]), (name, b)


@pytest.mark.parametrize("Q,G", [(100, 1), (100, 7), (100, 20), (100, 100), (128, 128), (16, 30), (1, 1), (1, 5), (64, 63)])
def test_lsap_exact_on_given_cost(Q, G):
    """assignment kernel alone on oracle-provided cost matrices: identical pairs and optimal total cost (properties:
    one-to-one, min(Q,G) pairs, rows sorted)"""
    gen = torch.Generator().manual_seed(1000 + Q * 131 + G)
    B = 4
    C = torch.rand(B, Q, G, generator=gen)
    off = torch.arange(0, (B + 1) * G, G, dtype=torch.int32, device=DEV)
    mq = torch.full((B, G), -1, dtype=torch.int64, device=DEV)
    mt = torch.full((B, G), -1, dtype=torch.int64, device=DEV)
    nm = torch.zeros(B, dtype=torch.int32, device=DEV)
    Cd = C.to(DEV).contiguous()
    L.check(L.lib().mi_lsap(Cd.data_ptr(), off.data_ptr(), B, Q, G, mq.data_ptr(), mt.data_ptr(), nm.data_ptr(),
                            L.stream_ptr()), "mi_lsap")
    torch.cuda.synchronize()
    for b in range(B):
        ri, ci = linear_sum_assignment(C[b].numpy())
        n = int(nm[b])
        assert n == min(Q, G) == len(ri)
        got_q, got_t = mq[b, :n].cpu().numpy(), mt[b, :n].cpu().numpy()
        assert np.array_equal(got_q, ri) and np.array_equal(got_t, ci)
        assert len(set(got_q.tolist())) == n and len(set(got_t.tolist())) == n
        assert (np.diff(got_q) > 0).all() if n > 1 else True


def test_matcher_edge_cases():
    logits, boxes, targets = D.synth_detr(2, 100, 91, 7, sizes=[3, 3])
    targets[1] = dict(labels=torch.zeros(0, dtype=torch.int64), boxes=torch.zeros(0, 4))   # image without objects
    m = HungarianMatcher(1.0, 5.0, 2.0)
    idx = m({"pred_logits": logits.to(DEV), "pred_boxes": boxes.to(DEV)}, _to_dev(targets))
    assert len(idx[0][0]) == 3 and len(idx[1][0]) == 0 and len(idx[1][1]) == 0
    ref, _ = D.hungarian_match(logits[:1], boxes[:1], targets[:1], 1.0, 5.0, 2.0)
    assert torch.equal(idx[0][0].cpu(), ref[0][0]) and torch.equal(idx[0][1].cpu(), ref[0][1])


# ------------------------------------------------------------------------------------------ attention core
def _mha_ref(q, k, v, mask, H):
    """fp32 reference of nn.MultiheadAttention's core (after in-projection, before out-projection, no dropout)"""
    Lq, B, E = q.shape
    Lk, D = k.shape[0], E // H
    qh = q.reshape(Lq, B * H, D).transpose(0, 1)
    kh = k.reshape(Lk, B * H, D).transpose(0, 1)
    vh = v.reshape(Lk, B * H, D).transpose(0, 1)
    s = torch.bmm(qh, kh.transpose(1, 2)) / (D ** 0.5)
    if mask is not None:
        s = s.view(B, H, Lq, Lk).masked_fill(mask[:, None, None, :], float("-inf")).view(B * H, Lq, Lk)
    p = torch.softmax(s, dim=-1)
    return torch.bmm(p, vh).transpose(0, 1).reshape(Lq, B, E)


@pytest.mark.parametrize("Lq,Lk,B,masked", [(100, 100, 2, False), (100, 1050, 2, True), (1050, 1050, 1, True), (37, 65, 3, True),
                                            (64, 32, 1, False)])
def test_mha_core_fwd_bwd(Lq, Lk, B, masked):
    from yolov7_d2_amd.modeling import mha_core
    H, E = 8, 256
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    bf = lambda t: t.to(torch.bfloat16).float()
    q, k, v = (bf(torch.randn(L_, B, E, generator=g)) for L_ in (Lq, Lk, Lk))
    go = bf(torch.randn(Lq, B, E, generator=g))
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.bool)
        mask[0, Lk - Lk // 5:] = True            # image 0 is narrower: its last keys are padding
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = _mha_ref(qr, kr, vr, mask, H)
    ref.backward(go)
    qd, kd, vd = (t.to(DEV, torch.bfloat16).requires_grad_(True) for t in (q, k, v))
    out = mha_core(qd, kd, vd, None if mask is None else mask.to(DEV), H)
    out.backward(go.to(DEV, torch.bfloat16))
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.detach().float().cpu() - b).norm() / (b.norm() + 1e-12))
    assert rel(out, ref.detach()) < 1e-2
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), ref.detach().numpy(), rtol=3e-2, atol=3e-2)
    assert rel(qd.grad, qr.grad) < 2e-2 and rel(kd.grad, kr.grad) < 2e-2 and rel(vd.grad, vr.grad) < 2e-2


# ------------------------------------------------------------------------------------------ IOUlossV6 family
@pytest.mark.parametrize("iou_type", ["giou", "diou", "ciou", "siou"])
def test_iou_loss_v6_against_reference_golden(golden_dir, iou_type):
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from gen_golden_inputs import synth_box_pairs
    from yolov7_d2_amd.modeling import IOUlossV6
    g = np.load(os.path.join(golden_dir, "iou_v6.npz"))
    pred, tgt = synth_box_pairs(257, 51)
    p = pred.to(DEV).requires_grad_(True)
    loss = IOUlossV6(box_format="xywh", iou_type=iou_type, reduction="none")(p.T, tgt.to(DEV))
    loss.sum().backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g[iou_type + "_loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(p.grad.cpu().numpy(), g[iou_type + "_grad"], rtol=2e-3, atol=2e-6)


# ------------------------------------------------------------------------------------------ transformer encoder layer
@pytest.mark.parametrize("name,pre", [("post", False), ("pre", True)])
def test_transformer_encoder_layer_against_reference_golden(golden_dir, name, pre):
    """our TransformerEncoderLayer (HIP kernels: 1x1-conv linears, fused MFMA attention, LayerNorm) loaded with the
    reference's state_dict vs the reference's own layer (fp32, eval mode): outputs and all gradients, bf16 tolerance"""
    import sys
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from gen_golden_inputs import synth_encoder_case, encoder_state_dict
    from yolov7_d2_amd.modeling import TransformerEncoderLayer
    g = np.load(os.path.join(golden_dir, "encoder_layer.npz"))
    layer = TransformerEncoderLayer(256, 8, 2048, dropout=0.1, normalize_before=pre)
    layer.load_state_dict(encoder_state_dict())      # same keys as the reference module
    layer.to(DEV).eval()
    src, pos, mask, go = synth_encoder_case()
    x = src.to(DEV, torch.bfloat16).requires_grad_(True)
    out = layer(x, src_key_padding_mask=mask.to(DEV), pos=pos.to(DEV, torch.bfloat16))
    out.backward(go.to(DEV, torch.bfloat16))
    torch.cuda.synchronize()
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))
    assert rel(out.detach().float().cpu().numpy(), g[name + "_out"]) < 2e-2
    assert rel(x.grad.float().cpu().numpy(), g[name + "_dsrc"]) < 4e-2
    # parameter gradients (golden keeps every 16th row of the matrices): bf16 storage noise through LayerNorm / ReLU
    # masks / softmax gives 2-5 % norm-relative differences against the fp32 reference
    for k, p in layer.named_parameters():
        got = (p.grad[::16] if p.dim() == 2 else p.grad).float().cpu().numpy()
        assert rel(got, g[f"{name}_g:{k}"]) < 7e-2, k
