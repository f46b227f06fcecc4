"""CPU (-m "not gpu"): host logic of the captured DETR / SparseInst steps that needs no device - the level-replicated
ground-truth layout one matching launch indexes, the vectorised loss bookkeeping of SetCriterion.weighted_packed against the
reference's per-key arithmetic (meta_arch/detr.py:262-267, 596-647), and which weights qualify for the one-launch packer."""
import torch

from yolov7_d2_amd.modeling import detr_criterion as dc
from yolov7_d2_amd.modeling.detr_matcher import PackedTargets


def test_level_replicated_targets_are_compact_copies_with_monotone_offsets():
    B, cap, levels = 3, 5, 4
    t = PackedTargets([None] * B, cap, torch.zeros(B + 1, dtype=torch.int32), torch.zeros(B * cap, dtype=torch.int64),
                      torch.zeros(B * cap, 4), torch.ones(1), levels=levels)
    off = [0, 2, 2, 5]                                  # image 1 has no box
    labels = torch.tensor([7, 8, 1, 2, 3])
    boxes = torch.arange(20, dtype=torch.float32).view(5, 4)
    t.fill_levels(off, labels, boxes)
    lo = t.lv["off"].tolist()
    assert len(lo) == levels * B + 1 and lo == sorted(lo) and lo[-1] == levels * 5
    for l in range(levels):
        for b in range(B):
            i = l * B + b
            assert lo[i + 1] - lo[i] == off[b + 1] - off[b]                      # image (l, b) sees image b's box count
            assert torch.equal(t.lv["labels"][lo[i]:lo[i + 1]], labels[off[b]:off[b + 1]])
            assert torch.equal(t.lv["boxes"][lo[i]:lo[i + 1]], boxes[off[b]:off[b + 1]])
    t.fill_levels([0, 0, 0, 0], None, None)             # a batch without boxes: all ranges empty, nothing is read
    assert t.lv["off"].tolist() == [0] * (levels * B + 1)
    single = PackedTargets([None] * B, cap, torch.zeros(B + 1, dtype=torch.int32), torch.zeros(B * cap, dtype=torch.int64),
                           torch.zeros(B * cap, 4), torch.ones(1))
    assert single.lv is None
    single.fill_levels(off, labels, boxes)              # no-op


def test_weighted_packed_bookkeeping_equals_the_per_key_arithmetic(monkeypatch):
    wd = {"loss_ce": 1.0, "loss_bbox": 5.0, "loss_giou": 2.0}
    wd.update({k + f"_{i}": v for i in range(5) for k, v in list(wd.items())[:3]})
    crit = dc.SetCriterion(80, None, wd, 0.1, ["labels", "boxes", "cardinality"])
    g = torch.Generator().manual_seed(0)
    vals = [torch.rand(5, generator=g).requires_grad_(True) for _ in range(6)]
    it = iter(vals)

    class Fake:
        @staticmethod
        def apply(*a):
            return next(it)

    class Logits:
        is_cuda, shape = True, (2, 100, 81)

    class Targets:
        inv_num_boxes = torch.tensor([0.37])
        lv = None

    monkeypatch.setattr(dc, "_SetLossFn", Fake)
    monkeypatch.setattr(crit, "_match", lambda lv, t: None)
    lvl = {"pred_logits": Logits(), "pred_boxes": None}
    r = crit.weighted_packed(dict(lvl, aux_outputs=[lvl] * 5), Targets())
    total = 0.0
    for l, v in enumerate(vals):
        s = "" if l == 0 else f"_{l - 1}"
        want = {"loss_ce" + s: v[0] * wd["loss_ce" + s], "loss_bbox" + s: v[3] * 0.37 * wd["loss_bbox" + s],
                "loss_giou" + s: v[4] * 0.37 * wd["loss_giou" + s], "cardinality_error" + s: v[2]}
        if l == 0:
            want["class_error"] = v[1]
        for k, e in want.items():
            assert abs(float(r[k]) - float(e)) < 1e-6, k
            if k.startswith("loss_"):
                total = total + float(e)
            else:
                assert not r[k].requires_grad
    assert set(r) == {k + s for s in [""] + [f"_{i}" for i in range(5)] for k in ("loss_ce", "loss_bbox", "loss_giou", "cardinality_error")} | {"class_error", "total"}
    assert abs(float(r["total"]) - total) < 1e-5
    r["total"].backward()
    torch.testing.assert_close(vals[3].grad, torch.tensor([1.0, 0.0, 0.0, 5 * 0.37, 2 * 0.37]))


def test_weight_images_only_takes_parameter_backed_weights():
    from yolov7_d2_amd.ops import WeightImages
    a, b = torch.nn.Parameter(torch.zeros(96, 64)), torch.nn.Parameter(torch.zeros(8, 3, 3, 3))
    reg = WeightImages([a, b, torch.nn.Parameter(torch.zeros(4, dtype=torch.float64))])
    ok = lambda t, n: reg._stable(t.data_ptr(), n * 4)
    assert ok(a, a.numel()) and ok(b, b.numel())
    assert ok(a[32:64], 32 * 64)                                        # a row block of a parameter (in_proj_weight[E:2E])
    assert not ok(a[64:], 64 * 64)                                      # would read past the parameter
    assert not ok(a.detach() * 2.0, a.numel())                          # a temporary
    calls = []
    assert reg.get(a.detach() * 2.0, 96, 64, 1, 64, 96, 96, 64, True, True, None, lambda wf, wd: calls.append(1)) is None and not calls


class _Toy(torch.nn.Module):
    """two 'stages' around a cut module; `share` re-uses the cut module's weight after the cut, `idle` adds an unused one"""

    def __init__(self, share=False, idle=False):
        super().__init__()
        self.device = torch.device("cpu")
        self.stage = torch.nn.Linear(4, 4)
        self.head = torch.nn.Linear(4, 2)
        self.share = share
        if idle:
            self.unused = torch.nn.Parameter(torch.zeros(3))

    def grad_cut_modules(self):
        return [self.stage]

    def forward_prepared(self, static):
        h = self.stage(static)
        if self.share:
            h = h @ self.stage.weight            # the SAME parameter on both sides of the cut
        return {"total": self.head(h).square().sum()}


def _staged(model):
    from yolov7_d2_amd.graph_step import GraphedTrainStep
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    gs = GraphedTrainStep(model, opt, batch_packs=False, backward_stages=True)
    try:
        fns = gs._stage_fns(torch.ones(3, 4))
        assert len(fns) == 2
        out = fns[0]()
        fns[1]()
        return gs, out
    finally:
        gs.close()


def test_staged_backward_assigns_every_parameter_to_one_stage():
    """the stage a parameter belongs to is found from which gradients appear - a parameter that gains gradient in
    two stages, or in none, must be refused (its first all-reduce would carry a partial sum / the one-launch AdamW would
    update a stale slot), not trained on silently"""
    import pytest
    from yolov7_d2_amd import _lib as L
    m = _Toy()
    gs, out = _staged(m)
    assert gs.stage_params == [[2, 3], [0, 1]]          # head in stage 0, the cut module in stage 1 (optimizer order)
    ref = _Toy()
    ref.load_state_dict(m.state_dict())
    ref.forward_prepared(torch.ones(3, 4))["total"].backward()
    for a, b in zip(m.parameters(), ref.parameters()):  # the cut changes nothing about the gradients
        torch.testing.assert_close(a.grad, b.grad)
    with pytest.raises(L.MI355Error, match="two backward stages"):
        _staged(_Toy(share=True))
    with pytest.raises(L.MI355Error, match="received no gradient"):
        _staged(_Toy(idle=True))
    frozen = _Toy(idle=True)
    frozen.unused.requires_grad_(False)                  # explicitly frozen: fine
    _staged(frozen)
