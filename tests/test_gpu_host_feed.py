"""The batch-building half of the DETR / SparseInst steps (csrc/host_feed.hip, ops.HostRing): one launch per batch against
the per-image torch calls of the reference (meta_arch/detr.py:273-278, meta_arch/sparseinst.py:95-98, utils/misc.py:148-170,
loss/sparseinst_loss.py:149-151) - the torch calls ARE the oracle here, run on the same device."""
import pytest
import torch
import torch.nn.functional as F

import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd import ops
from yolov7_d2_amd.d2shim import Boxes, Instances

pytestmark = pytest.mark.gpu
DEV = "cuda"
MEAN, STD = [123.675, 116.280, 103.530], [58.395, 57.120, 57.375]


@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8])
def test_normalize_pad_batch_equals_the_per_image_torch_calls_bit_for_bit(dtype):
    g = torch.Generator().manual_seed(0)
    sizes = [(96, 128), (61, 77), (1, 1), (96, 3), (40, 128)]
    imgs = [torch.randint(0, 256, (3, h, w), generator=g).to(dtype) for h, w in sizes]
    if dtype == torch.float32:
        imgs = [im + torch.rand(im.shape, generator=g) for im in imgs]           # (not only integers)
    imgs = [im.to(DEV) if i % 2 == 0 else im for i, im in enumerate(imgs)]       # (host images are moved over)
    mean, std = torch.tensor(MEAN, device=DEV).view(3, 1, 1), torch.tensor(STD, device=DEV).view(3, 1, 1)
    for Wp in (128, 130):                                                        # (130: DETR pads to the largest image, any width)
        dst = torch.full((len(sizes), 3, 96, Wp), 7.0, device=DEV)               # (stale values: every element is written)
        ops.normalize_pad_batch(imgs, dst, MEAN, STD)
        ref = torch.zeros_like(dst)
        for b, im in enumerate(imgs):
            t = (im.to(DEV).float() - mean) / std
            ref[b, :, :t.shape[1], :t.shape[2]] = t
        assert torch.equal(dst, ref), Wp


def test_normalize_pad_batch_beyond_one_launch_and_argument_checks():
    n = 70                                                                       # > MI_FEED_MAX_IMAGES (32): three launches
    g = torch.Generator().manual_seed(1)
    imgs = [torch.randint(0, 256, (3, 8 + b % 5, 4 + b % 7), generator=g).float().to(DEV) for b in range(n)]
    dst = torch.empty(n, 3, 12, 12, device=DEV)
    ops.normalize_pad_batch(imgs, dst, MEAN, STD)
    for b in (0, 31, 32, 63, 64, 69):
        h, w = imgs[b].shape[1:]
        ref = (imgs[b] - torch.tensor(MEAN, device=DEV).view(3, 1, 1)) / torch.tensor(STD, device=DEV).view(3, 1, 1)
        assert torch.equal(dst[b, :, :h, :w], ref) and float(dst[b, :, h:].abs().sum()) == 0 and float(dst[b, :, :, w:].abs().sum()) == 0
    with pytest.raises(L.MI355Error):
        ops.normalize_pad_batch([torch.zeros(3, 13, 4, device=DEV)], torch.empty(1, 3, 12, 12, device=DEV), MEAN, STD)


def _torch_targets(masks, labels, cap, in_shape, out_shape):
    B, P = len(masks), out_shape[0] * out_shape[1]
    tgt = torch.zeros(B * cap, P, device=DEV)
    lab = torch.zeros(B, cap, dtype=torch.int64, device=DEV)
    for b, m in enumerate(masks):
        Mb = m.shape[0]
        if Mb == 0:
            continue
        pad = torch.zeros(Mb, in_shape[0], in_shape[1], device=DEV)
        pad[:, :m.shape[1], :m.shape[2]] = m.to(DEV).float()
        r = F.interpolate(pad[:, None], size=out_shape, mode="bilinear", align_corners=False).squeeze(1)
        tgt[b * cap: b * cap + Mb] = r.flatten(1)
        lab[b, :Mb] = labels[b].to(DEV)
    return tgt, tgt.view(B, cap, P).transpose(1, 2).to(torch.bfloat16).contiguous(), lab


@pytest.mark.parametrize("case", ["binary_x4", "bool_ragged", "float_odd_ratio"])
def test_mask_targets_batch_equals_pad_interpolate_stack(case):
    g = torch.Generator().manual_seed(2)
    cap = 32
    if case == "binary_x4":          # the model's case: prediction size = padded input / 4 (every weight 0.5: exact)
        in_shape, out_shape, dt = (96, 128), (24, 32), torch.float32
        shapes = [(3, 96, 128), (0, 96, 128), (32, 80, 100), (1, 17, 128)]
    elif case == "bool_ragged":
        in_shape, out_shape, dt = (64, 96), (16, 24), torch.bool
        shapes = [(2, 64, 96), (5, 33, 50), (1, 1, 1)]
    else:                            # a ratio that is not a power of two: fractional weights
        in_shape, out_shape, dt = (50, 70), (17, 23), torch.float32
        shapes = [(4, 50, 70), (2, 31, 69)]
    masks, labels = [], []
    for (Mb, h, w) in shapes:
        m = (torch.rand(Mb, h, w, generator=g) > 0.5)
        if case == "float_odd_ratio":
            m = torch.rand(Mb, h, w, generator=g)
        masks.append(m.to(dt) if dt != torch.float32 else m.float())
        labels.append(torch.randint(0, 80, (Mb,), generator=g))
    masks = [m.to(DEV) if i % 2 == 0 else m for i, m in enumerate(masks)]
    B, P = len(shapes), out_shape[0] * out_shape[1]
    tgt = torch.full((B * cap, P), 3.0, device=DEV)
    tgtT = torch.full((B, P, cap), 3.0, dtype=torch.bfloat16, device=DEV)
    lab = torch.full((B, cap), 9, dtype=torch.int64, device=DEV)
    t2 = torch.full((B, cap), 5.0, device=DEV)
    ops.mask_targets_batch(masks, labels, cap, in_shape, out_shape, tgt, tgtT, lab, t2=t2)
    rt, rT, rl = _torch_targets(masks, labels, cap, in_shape, out_shape)
    r2 = (rt.double() * rt.double()).sum(-1).view(B, cap)
    assert float((t2.double() - r2).abs().max()) <= 1e-6 * float(r2.abs().max()) + 1e-12
    t2b = torch.empty_like(t2)
    ops.mask_targets_batch(masks, labels, cap, in_shape, out_shape, tgt, tgtT, lab, t2=t2b)
    assert torch.equal(t2, t2b)                                  # (a fixed summation order: run-to-run identical)
    if case == "float_odd_ratio":    # (torch's kernel may contract a * b + c * d into an fma: one ulp)
        assert float((tgt - rt).abs().max()) <= 2e-7
        assert float((tgtT.float() - rT.float()).abs().max()) <= 1e-2
    else:
        assert torch.equal(tgt, rt) and torch.equal(tgtT, rT)
    assert torch.equal(lab, rl)
    assert float(rt.abs().max()) > 0


@pytest.mark.parametrize("which", ["sparseinst", "detr"])
def test_prepare_batch_one_launch_forms_equal_the_per_image_forms(which, monkeypatch):
    """Detr.prepare_batch / SparseInst.prepare_batch with the batch kernels (default) against MI_FEED_BATCH=0: every static
    tensor of the step identical, on a ragged batch, refilled in place with a second batch"""
    g = torch.Generator().manual_seed(3)
    cfg = M.sparse_inst_r50_giam_cfg(device=DEV) if which == "sparseinst" else M.detr_r50_cfg(device=DEV)
    model = M.build_model(cfg).train()

    def batch(seed):
        gg = torch.Generator().manual_seed(seed)
        out = []
        for b, (h, w) in enumerate([(96, 128), (70, 90), (96, 64)]):
            n = (b + seed) % 3                      # (an image without instances among them)
            if which == "sparseinst":
                inst = Instances((h, w), gt_classes=torch.randint(0, 80, (n,), generator=gg).to(DEV),
                                 gt_masks=(torch.rand(n, h, w, generator=gg) > 0.6).float().to(DEV))
            else:
                xy = torch.rand(n, 2, generator=gg) * 40
                inst = Instances((h, w), gt_boxes=Boxes(torch.cat([xy, xy + 20], 1)), gt_classes=torch.randint(0, 80, (n,), generator=gg))
            out.append(dict(image=torch.randint(0, 256, (3, h, w), generator=gg).float().to(DEV), instances=inst, height=h, width=w))
        return out

    def tensors(static):
        out = {"images": (static["images"] if torch.is_tensor(static["images"]) else static["images"].tensor).clone()}
        t = static["targets"]
        for k, v in vars(t).items():
            if torch.is_tensor(v):
                out[k] = v.clone()
            elif isinstance(v, dict):
                out.update({f"{k}.{kk}": vv.clone() for kk, vv in v.items() if torch.is_tensor(vv)})
        return out

    res = []
    for mode in ("0", "1"):
        monkeypatch.setenv("MI_FEED_BATCH", mode)
        static = model.prepare_batch(batch(1))
        a = tensors(static)
        model.prepare_batch(batch(2), static=static)
        torch.cuda.synchronize()
        res.append((a, tensors(static)))
    for first, second in zip(res[0], res[1]):
        assert set(first) == set(second) and len(first) >= 4
        for k in first:
            assert torch.equal(first[k], second[k]), (which, k)


def test_padding_masks_of_all_levels_in_one_launch_equal_the_torch_spelling(monkeypatch):
    """MaskedBackbone.mask_out_padding (meta_arch/detr.py:385-403) from the device copy of the image sizes: mi_padding_masks
    against the host loop of the reference and against the per-level torch spelling (MI_FEED_BATCH=0)"""
    from yolov7_d2_amd.modeling.detr_meta import MaskedBackbone
    mb = MaskedBackbone.__new__(MaskedBackbone)
    torch.nn.Module.__init__(mb)
    mb.feature_strides = [4, 8, 16, 32]
    sizes = [(250, 310), (224, 288), (1, 1), (256, 320)]
    shapes = [(4, 8, 64, 80), (4, 8, 32, 40), (4, 8, 16, 20), (4, 8, 8, 10)]
    sd = torch.tensor(sizes, dtype=torch.int64, device=DEV)
    ref = mb.mask_out_padding(shapes, sizes, DEV)
    got = mb.mask_out_padding_dev(shapes, sd)
    monkeypatch.setenv("MI_FEED_BATCH", "0")
    spelled = mb.mask_out_padding_dev(shapes, sd)
    for r, g, s in zip(ref, got, spelled):
        assert g.dtype == torch.bool and torch.equal(r, g) and torch.equal(r, s)
    assert bool(ref[0].any()) and not bool(ref[0].all())
