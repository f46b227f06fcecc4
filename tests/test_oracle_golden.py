"""CPU: the oracle restatement (oracle/yolox_oracle.py) reproduces the vectors obtained by executing the
reference's own source (oracle/gen_golden.py -> tests/golden/*.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

import yolox_oracle as O


def test_step_losses_grads_and_eval(golden_dir):
    g = np.load(os.path.join(golden_dir, "yolox_s_step_64x96.npz"))
    sd = O.init_state_dict(0.33, 0.5, 80, seed=0)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    imgs, labels = O.synth_batch(2, 64, 96, seed=11, max_gt=4)
    res = O.train_step_losses(sd, imgs, labels)
    got = np.array([float(x) for x in res])
    np.testing.assert_allclose(got, g["losses"], rtol=1e-6, atol=1e-6)
    (res[0] + res[1] + res[2] + res[3]).backward()
    for k in g.files:
        if k.startswith("grad:"):
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), g[k], rtol=1e-4, atol=1e-6)
    names = [str(n) for n in g["grad_names"]]
    norms = np.array([float(sd[n].grad.norm()) for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["backbone.stem.conv.bn.running_mean"].numpy(),
                               g["rm:backbone.stem.conv.bn.running_mean"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sd["head.stems.1.bn.running_var"].numpy(), g["rv:head.stems.1.bn.running_var"],
                               rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        net = O.Net({k: v.detach() for k, v in sd.items()}, 0.33, 0.5, 80, training=False)
        raw, hw = net.forward_raw(imgs)
        ev = O.decode_eval(raw, O.make_anchors(hw))
    np.testing.assert_allclose(ev.numpy(), g["eval_out"], rtol=1e-5, atol=1e-5)


def test_depthwise_backbone_step_oracle_against_reference(golden_dir):
    """MODEL.DARKNET.DEPTH_WISE True (DWConv = depthwise 3x3 BaseConv + pointwise BaseConv in the backbone,
    wrappers.py:86-102, darknetx.py:113): the oracle's restatement against the reference's own step"""
    g = np.load(os.path.join(golden_dir, "yolox_s_dw_step_64x96.npz"))
    sd = O.init_state_dict(0.33, 0.5, 80, seed=3, depthwise=True)
    assert sd["backbone.dark2.0.dconv.conv.weight"].shape == (32, 1, 3, 3)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    imgs, labels = O.synth_batch(2, 64, 96, seed=12, max_gt=4)
    res = O.train_step_losses(sd, imgs, labels, depthwise=True)
    np.testing.assert_allclose(np.array([float(x) for x in res]), g["losses"], rtol=1e-6, atol=1e-6)
    (res[0] + res[1] + res[2] + res[3]).backward()
    for k in g.files:
        if k.startswith("grad:"):
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), g[k], rtol=1e-4, atol=1e-6)
    names = [str(n) for n in g["grad_names"]]
    assert sorted(k for k, v in sd.items() if v.requires_grad) == names
    np.testing.assert_allclose(np.array([float(sd[n].grad.norm()) for n in names]), g["grad_norms"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["backbone.dark3.0.dconv.bn.running_mean"].numpy(),
                               g["rm:backbone.dark3.0.dconv.bn.running_mean"], rtol=1e-6, atol=1e-6)


def test_config0_yolox_tiny_416_cpu_step(golden_dir):
    """BASELINE.json configs[0] (YOLOX-tiny, width .375, 416x416, bs=2, CPU): the oracle against the reference's own
    modules run by path - the 4 losses, the gradient norm of every parameter, one full gradient and the eval output.
    (The same configuration on the HIP device: tests/test_gpu_widths.py::test_yolox_tiny_config0_against_reference_golden.)"""
    g = np.load(os.path.join(golden_dir, "yolox_tiny_step_416.npz"))
    sd = O.init_state_dict(0.33, 0.375, 80, seed=0)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    imgs, labels = O.synth_batch(2, 416, 416, seed=31, max_gt=6)
    res = O.train_step_losses(sd, imgs, labels, depth=0.33, width=0.375)
    np.testing.assert_allclose(np.array([float(x) for x in res]), g["losses"], rtol=1e-5, atol=1e-6)
    (res[0] + res[1] + res[2] + res[3]).backward()
    names = [str(n) for n in g["grad_names"]]
    norms = np.array([float(sd[n].grad.norm()) for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(sd["head.cls_preds.1.weight"].grad.numpy(), g["grad:head.cls_preds.1.weight"], rtol=2e-4,
                               atol=1e-6)
    with torch.no_grad():
        net = O.Net({k: v.detach() for k, v in sd.items()}, 0.33, 0.375, 80, training=False)
        raw, hw = net.forward_raw(imgs)
        ev = O.decode_eval(raw, O.make_anchors(hw))
    np.testing.assert_allclose(ev[:, ::7].numpy(), g["eval_out_stride"], rtol=1e-5, atol=1e-5)


def test_l1_loss_oracle_against_reference(golden_dir):
    """oracle restatement of head.use_l1 (yolox_head.py:389-448) against the reference head run with the switch on"""
    g = np.load(os.path.join(golden_dir, "simota_160_l1.npz"))
    B, H, W = 3, 160, 160
    _, labels = O.synth_batch(B, H, W, seed=21, max_gt=12, min_gt=6)
    labels[1] = 0.0
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    raw, anchors = O.synth_raw(B, hw, 22, labels=labels)
    raw.requires_grad_(True)
    res = O.yolox_losses(raw, labels, anchors, 80, use_l1=True)
    np.testing.assert_allclose(np.array([float(x) for x in res]), g["losses"], rtol=1e-6)
    (res[0] + res[1] + res[2] + res[3] + res[4]).backward()
    np.testing.assert_allclose(raw.grad.numpy(), g["draw"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name,kw", [("ciou", dict(iou_type="ciou")),
                                     ("siou", dict(iou_type="siou", center_radius=1.5, iou_weight=2.0, cls_weight=0.5, reg_weight=2.5))])
def test_yolov6_compute_loss_oracle_against_reference(golden_dir, name, kw):
    """the YOLOv6 head's ComputeLoss (yolov6_head.py:315-754) = the YOLOX loss with other constants, an IOUlossV6 box loss
    and the l1 term always on: the oracle's parameterised restatement against the reference class itself"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    from gen_golden_inputs import synth_yolov6_case
    g = np.load(os.path.join(golden_dir, "yolov6_loss.npz"))
    _, t, labels, raw, anchors = synth_yolov6_case()
    raw = raw.clone().requires_grad_(True)
    res = O.yolox_losses(raw, labels, anchors, 80, use_l1=True, **kw)
    np.testing.assert_allclose(float(res[0]), float(g[name + "_total"][0]), rtol=2e-6)
    np.testing.assert_allclose(np.array([float(res[1]), float(res[4]), float(res[2]), float(res[3])]), g[name + "_parts"], rtol=2e-6)
    res[0].backward()
    np.testing.assert_allclose(raw.grad.numpy(), g[name + "_grad"], rtol=1e-4, atol=1e-7)
    # the reference scaled its normalised targets to pixels in place: those are the labels the oracle was given
    np.testing.assert_allclose(g[name + "_targets_after"], labels.numpy(), rtol=1e-6, atol=1e-4)


def test_simota_assignment_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "simota_160.npz"))
    B, H, W = 3, 160, 160
    _, labels = O.synth_batch(B, H, W, seed=21, max_gt=12, min_gt=6)
    labels[1] = 0.0
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    raw, anchors = O.synth_raw(B, hw, 22, labels=labels)
    raw.requires_grad_(True)
    res, assigns = O.yolox_losses(raw, labels, anchors, 80, return_assign=True)
    np.testing.assert_allclose(np.array([float(x) for x in res]), g["losses"], rtol=1e-6)
    (res[0] + res[1] + res[2] + res[3]).backward()
    np.testing.assert_allclose(raw.grad.numpy(), g["draw"], rtol=1e-5, atol=1e-7)
    assert assigns[1] is None
    for b in (0, 2):
        a = assigns[b]
        assert np.array_equal(a["fg"].numpy(), g[f"fg{b}"])                 # integer / boolean: exact
        assert np.array_equal(a["matched_gt"].numpy(), g[f"matched_gt{b}"])
        assert np.array_equal(a["matched_cls"].numpy(), g[f"matched_cls{b}"])
        np.testing.assert_array_equal(a["matched_iou"].numpy(), g[f"matched_iou{b}"])
    assert len(set(assigns[2]["matched_gt"].tolist())) > 1
    assert sum(int(assigns[b]["fg"].sum()) for b in (0, 2)) > 20   # dynamic k > 1 is exercised


def test_postprocess(golden_dir):
    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    for name, n, seed in (("small", 300, 31), ("large", 2500, 32)):   # both batched_nms branches
        pred = O.synth_decoded(2, n, seed)
        out = O.postprocess(pred, 80, 0.3, 0.65)
        for b in range(2):
            np.testing.assert_array_equal(out[b].numpy(), g[f"{name}_out{b}"])
            cls = out[b][:, 6]
            assert (cls == cls.round()).all()


def test_nms_edge_cases():
    e = torch.empty(0, 4)
    assert O.batched_nms(e, torch.empty(0), torch.empty(0), 0.5).numel() == 0
    b = torch.tensor([[0., 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30], [0, 0, 10, 10.5]])
    s = torch.tensor([0.9, 0.8, 0.7, 0.6])
    assert O.nms(b, s, 0.5).tolist() == [0, 2]
    # same boxes, different classes: class-aware keeps all but the same-class duplicate
    assert O.batched_nms(b, s, torch.tensor([0., 1, 0, 0]), 0.5).tolist() == [0, 1, 2]
    # threshold is strict (IoU == thr is kept)
    b2 = torch.tensor([[0., 0, 2, 1], [1, 0, 3, 1]])  # IoU = 1/3
    assert O.nms(b2, torch.tensor([1., 0.5]), 1.0 / 3.0 + 1e-3).tolist() == [0, 1]


def test_detr_matcher_oracle_against_reference_golden(golden_dir):
    """oracle/detr_oracle.py (restated cost + scipy LSAP) == the reference's own HungarianMatcher run by path"""
    import detr_oracle as D
    g = np.load(os.path.join(golden_dir, "hungarian.npz"))
    for name, (bs, nq, seed, sizes) in dict(a=(3, 100, 41, None), b=(2, 100, 42, [100, 1]), c=(2, 16, 43, [30, 7])).items():
        logits, boxes, targets = D.synth_detr(bs, nq, 91, seed, sizes=sizes)
        idx, _ = D.hungarian_match(logits, boxes, targets, 1.0, 5.0, 2.0)
        for b, (i, j) in enumerate(idx):
            assert np.array_equal(i.numpy(), g[f"{name}_i{b}"]) and np.array_equal(j.numpy(), g[f"{name}_j{b}"])


SET_CRIT_CASES = dict(a=(3, 100, 91, 81, None), b=(2, 100, 80, 82, [100, 0]), c=(2, 16, 20, 83, [30, 7]))
SET_CRIT_W = {"loss_ce": 1.0, "loss_bbox": 5.0, "loss_giou": 2.0}


def set_crit_weights(naux=2):
    wd = dict(SET_CRIT_W)
    wd.update({k + f"_{i}": v for i in range(naux) for k, v in SET_CRIT_W.items()})
    return wd


def test_detr_set_criterion_oracle_against_reference_golden(golden_dir):
    """oracle/detr_oracle.py::set_criterion == the reference's own SetCriterion (+ its matcher) run by path: every
    loss of every decoder level, and the gradients of the weighted total wrt logits and boxes"""
    import detr_oracle as D
    g = np.load(os.path.join(golden_dir, "set_criterion.npz"))
    wd = set_crit_weights()
    for name, (bs, nq, ncls, seed, sizes) in SET_CRIT_CASES.items():
        outs, targets = D.synth_detr_levels(bs, nq, ncls, seed, levels=3, sizes=sizes)
        leaves = [(l.clone().requires_grad_(True), b.clone().requires_grad_(True)) for l, b in outs]
        outputs = {"pred_logits": leaves[-1][0], "pred_boxes": leaves[-1][1],
                   "aux_outputs": [{"pred_logits": l, "pred_boxes": b} for l, b in leaves[:-1]]}
        ld = D.set_criterion(outputs, targets, ncls, 0.1)
        keys = {k.split(":", 1)[1] for k in g.files if k.startswith(name + ":") and g[k].ndim == 0} - {"total"}
        assert set(ld.keys()) == keys
        for k in keys:
            np.testing.assert_allclose(float(ld[k].detach()), float(g[f"{name}:{k}"]), rtol=2e-6, atol=1e-6, err_msg=k)
        sum(ld[k] * wd[k] for k in ld if k in wd).backward()
        for i, (l, b) in enumerate(leaves):
            np.testing.assert_allclose(l.grad.numpy(), g[f"{name}:dlogits{i}"], rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(b.grad.numpy(), g[f"{name}:dboxes{i}"], rtol=1e-5, atol=1e-7)


def test_bf16_storage_noise_floor():
    """Why the whole-step GPU parity test (tests/test_gpu_parity_bench.py) pins the forward state: two bf16-STORAGE
    executions of the same network that differ only by a sub-ulp jitter before each rounding (i.e. two correct
    implementations with different accumulation orders) agree to ~1 % at the head output, but train-mode BatchNorm's
    backward amplifies that forward noise so much that their weight gradients agree only to cosine ~0.8 in the backbone.
    With the conv outputs of one run forced into the other (teacher forcing, O.Net(force=...)) the same two
    implementations agree to cosine > 0.999 on EVERY parameter - that is the bound the GPU test asserts."""
    torch.manual_seed(0)

    def mk(jitter):
        class Q(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                if jitter:
                    t = t * (1 + jitter * (torch.rand_like(t) - 0.5))
                return t.to(torch.bfloat16).float()

            @staticmethod
            def backward(ctx, g):
                if jitter:
                    g = g * (1 + jitter * (torch.rand_like(g) - 0.5))
                return g.to(torch.bfloat16).float()
        return Q.apply

    B, H, W = 2, 128, 160
    imgs, labels = O.synth_batch(B, H, W, seed=1234, max_gt=6)
    sd0 = O.init_state_dict(0.33, 0.5, 80, seed=0)

    def run(quant, dpreds=None, force=None):
        sd = {k: v.clone() for k, v in sd0.items()}
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        net = O.Net(sd, 0.33, 0.5, 80, training=True, quant=quant, force=force)
        raw, hw = net.forward_raw(imgs)
        if dpreds is None:
            r = raw.detach().clone().requires_grad_(True)
            res = O.yolox_losses(r, labels, O.make_anchors(hw), 80)
            (res[0] + res[1] + res[2] + res[3]).backward()
            dpreds = r.grad
        raw.backward(dpreds)
        ys = {k: v.detach().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous().flatten() for k, v in net.taps.items()
              if k.endswith(".y")}
        return raw.detach(), dpreds, {k: v.grad for k, v in sd.items() if v.requires_grad}, ys

    def cosines(a, b):
        return {k: float((a[k] * b[k]).sum() / (a[k].norm() * b[k].norm() + 1e-30)) for k in a if float(b[k].norm()) > 0}

    raw_a, dp, g_a, ys_a = run(mk(0.0))
    raw_b, _, g_b, _ = run(mk(1e-3), dp)
    raw_f, _, g_f, _ = run(mk(1e-3), dp, force=ys_a)
    free, forced = cosines(g_b, g_a), cosines(g_f, g_a)
    rel_raw = float((raw_b - raw_a).norm() / raw_a.norm())
    print("raw rel L2 between the two bf16 executions", rel_raw)
    print("un-forced: cos min %.3f median %.3f | forced: cos min %.5f" %
          (min(free.values()), float(np.median(list(free.values()))), min(forced.values())))
    assert rel_raw < 3e-2
    assert min(free.values()) < 0.95        # an end-to-end cos >= 0.99 bound fails between two correct implementations
    assert min(forced.values()) > 0.999     # ... and holds for every parameter once the forward state is pinned


@pytest.mark.skipif(not os.path.isdir("/root/reference/yolov7"), reason="the reference tree only exists in the build container")
def test_golden_recipe_regenerates_every_fixture_in_one_run(golden_dir, tmp_path):
    """oracle/gen_golden.py, ONE invocation, against /root/reference: every committed fixture comes out bit-identical
    (round 2's recipe only worked generator by generator: a stub installed by one loader shadowed the next loader's)"""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "gen_golden.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    committed = sorted(os.path.basename(f) for f in glob.glob(os.path.join(golden_dir, "*.npz")) if not os.path.basename(f).startswith("ref_"))
    made = sorted(os.path.basename(f) for f in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    assert set(made) <= set(committed) and len(made) >= 24, (sorted(set(made) - set(committed)), len(made))
    for name in made:
        a, b = np.load(os.path.join(str(tmp_path), name), allow_pickle=True), np.load(os.path.join(golden_dir, name), allow_pickle=True)
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (name, k)
            assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (name, k)


def test_detr_box_ops_oracle_against_reference_golden(golden_dir):
    """oracle/detr_oracle.py's box utilities (what the matcher / criterion oracles are built on) against the reference's
    own functions (utils/boxes.py:28-37,85-122; golden box_ops.npz)"""
    import detr_oracle as DO
    g = np.load(os.path.join(golden_dir, "box_ops.npz"))
    assert np.array_equal(DO.box_cxcywh_to_xyxy(torch.from_numpy(g["cxcywh"])).numpy(), g["xyxy"])
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    iou, uni = DO.box_iou(a, b)
    assert np.array_equal(iou.numpy(), g["iou"]) and np.array_equal(uni.numpy(), g["union"])
    assert np.array_equal(DO.generalized_box_iou(a, b).numpy(), g["giou"])


def test_resnet50_oracle_against_an_independent_implementation():
    """detectron2's ResNet-50 is un-vendored, so `oracle/resnet_oracle.py` (what the HIP ResNet is held to) restates it; here
    it is cross-checked against a THIRD implementation of the same published network that IS installed: transformers'
    ResNetModel in the torchvision-v1.5 / d2 `STRIDE_IN_1X1 False` topology (stride on the 3x3 conv), weights and BatchNorm
    statistics mapped key by key (stem.conv1 <-> embedder, res{s}.{b}.conv{k} <-> stages.{s-2}.layers.{b}.layer.{k-1}, shortcut
    <-> shortcut) - the four stage outputs in fp32.  Not detectron2 itself (parity with it stays unpinned), but an
    independent reading of the architecture: stem 7x7 s2 + max-pool 3/2/1, bottlenecks (3, 4, 6, 3), 1x1 stride-2 shortcuts."""
    import resnet_oracle as R
    # oracle/ref_loader.py (other tests of this suite) leaves spec-less stub modules in sys.modules (torchvision, detectron2,
    # cv2, ...): transformers probes those names with importlib.util.find_spec, which raises on them - hide them meanwhile
    stub_roots = ("alfred", "cv2", "detectron2", "fvcore", "loguru", "omegaconf", "pycocotools", "torchvision", "timm")
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules)
              if k.split(".")[0] in stub_roots and getattr(sys.modules[k], "__spec__", 1) is None}
    try:
        tr = pytest.importorskip("transformers")
        hf, hsd = _hf_resnet50(tr)
    finally:
        sys.modules.update(hidden)
    _check_resnet_oracle_against(hf, hsd, R)


def _hf_resnet50(tr):
    cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                          layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    torch.manual_seed(0)
    hf = tr.ResNetModel(cfg).eval()
    g = torch.Generator().manual_seed(1)
    for m in hf.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.weight.data.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    return hf, hf.state_dict()


def _check_resnet_oracle_against(hf, hsd, R):
    g = torch.Generator().manual_seed(2)
    sd = R.init_state_dict(50, seed=0)

    def put(ours, theirs):
        sd[ours + ".weight"] = hsd[theirs + ".convolution.weight"].clone()
        for k in ("weight", "bias", "running_mean", "running_var"):
            sd[f"{ours}.norm.{k}"] = hsd[f"{theirs}.normalization.{k}"].clone()
    put("stem.conv1", "embedder.embedder")
    for s, nb in enumerate((3, 4, 6, 3)):
        for b in range(nb):
            if b == 0:
                put(f"res{s + 2}.0.shortcut", f"encoder.stages.{s}.layers.0.shortcut")
            for k in (1, 2, 3):
                put(f"res{s + 2}.{b}.conv{k}", f"encoder.stages.{s}.layers.{b}.layer.{k - 1}")
    assert set(sd) == set(R.init_state_dict(50, seed=0))            # every oracle tensor was overwritten or kept by name
    x = torch.randn(2, 3, 75, 101, generator=g)
    with torch.no_grad():
        ref = hf(x, output_hidden_states=True).hidden_states
        out = R.forward(sd, x, 50, stride_in_1x1=False)
    for i, name in enumerate(("res2", "res3", "res4", "res5")):
        assert out[name].shape == ref[i + 1].shape
        rel = float((out[name] - ref[i + 1]).norm() / ref[i + 1].norm())
        assert rel < 2e-6, (name, rel)                               # (fp32 summation order only: the folded vs separate BatchNorm)


@pytest.mark.parametrize("name,pre", [("post", False), ("pre", True)])
def test_detr_net_oracle_against_reference_golden(golden_dir, name, pre):
    """oracle/detr_net_oracle.py::transformer (the functional fp32 restatement the forward-pinned DETR parity test runs on the
    GPU box's host cores) against the golden the REFERENCE'S OWN Transformer class produced by path (transformer.npz, 2 + 2
    layers, post- and pre-norm): hs, memory, d src, d query and every sampled parameter gradient to fp32 rounding."""
    import detr_net_oracle as DN
    from gen_golden_inputs import seeded_state_dict, synth_transformer_case
    from yolov7_d2_amd.modeling import Transformer
    g = np.load(os.path.join(golden_dir, "transformer.npz"))
    net = Transformer(256, 8, 2, 2, 512, 0.1, normalize_before=pre, return_intermediate_dec=True)     # (shapes / keys only)
    sd = {k: v.clone().requires_grad_(True) for k, v in seeded_state_dict(net).items()}
    src, mask, qe, pos = synth_transformer_case()
    x, q = src.clone().requires_grad_(True), qe.clone().requires_grad_(True)
    hs, mem = DN.transformer(sd, x, mask, q, pos, nhead=8, pre=pre)
    gh = torch.randn(hs.shape, generator=torch.Generator().manual_seed(73)).to(torch.bfloat16).float()
    (hs * gh).sum().backward()
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))
    valid = ~mask.flatten(1).numpy()
    mv = lambda m: m.reshape(2, 256, -1).transpose(0, 2, 1)[valid]
    assert rel(hs.detach().numpy(), g[name + "_hs"]) < 1e-5
    assert rel(mv(mem.detach().numpy()), mv(g[name + "_mem"])) < 1e-5
    assert rel(x.grad.numpy(), g[name + "_dsrc"]) < 1e-4 and rel(q.grad.numpy(), g[name + "_dquery"]) < 1e-4
    checked = 0
    for k, p in sd.items():
        ref = g[f"{name}_g:{k}"]
        got = (p.grad[::32] if p.dim() == 2 else p.grad).numpy()
        if k.startswith("decoder.layers.0.self_attn.in_proj"):
            # mathematically zero (the first decoder layer attends over tgt = 0, detr_backbone.py:61): both sides hold fp32
            # rounding residue, negligible against the same parameter one layer up
            scale = np.linalg.norm(g[f"{name}_g:{k.replace('layers.0', 'layers.1')}"])
            if np.linalg.norm(ref) < 1e-3 * scale:
                assert np.linalg.norm(got) < 1e-3 * scale, (k, np.linalg.norm(got), scale)
                continue
        assert rel(got, ref) < 1e-3, (k, rel(got, ref))
        checked += 1
    assert checked >= len(sd) - 4


def test_detr_net_oracle_position_embedding_against_reference_golden(golden_dir):
    """detr_net_oracle.position_embedding_sine against the reference's own PositionEmbeddingSine (pos_embed.npz: the DETR
    configuration - 128 features, normalised - and the un-normalised 64-feature form)"""
    import detr_net_oracle as DN
    g = np.load(os.path.join(golden_dir, "pos_embed.npz"))
    mask = torch.from_numpy(g["mask"])
    np.testing.assert_allclose(DN.position_embedding_sine(mask, 128, normalize=True).numpy(), g["detr"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(DN.position_embedding_sine(mask, 64, normalize=False).numpy(), g["raw"], rtol=1e-5, atol=1e-5)


def test_sparseinst_net_oracle_against_reference_golden(golden_dir):
    """oracle/sparseinst_net_oracle.py (functional fp32 restatement of InstanceContextEncoder + GroupIAMDecoder, the oracle of
    the forward-pinned SparseInst parity test) against the golden the REFERENCE'S OWN modules produced by path
    (sparseinst.npz): encoder output, class logits, objectness, mask logits to fp32 rounding"""
    import types
    import sparseinst_net_oracle as SN
    import yolov7_d2_amd as M
    from yolov7_d2_amd.modeling import sparseinst as S
    from gen_golden_inputs import seeded_tensor_dict, sparseinst_spread, synth_sparseinst_case
    g = np.load(os.path.join(golden_dir, "sparseinst.npz"))
    cfg = M.sparse_inst_r50_giam_cfg(device="cpu")
    shapes = {n: types.SimpleNamespace(channels=c, stride=s) for n, c, s in (("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    net = torch.nn.ModuleDict(dict(encoder=S.InstanceContextEncoder(cfg, shapes), decoder=S.GroupIAMDecoder(cfg)))     # keys / shapes
    assert sorted(dict(net.named_parameters()).keys()) == [str(n) for n in g["param_names"]]
    sd = sparseinst_spread(seeded_tensor_dict({k: v.shape for k, v in net.state_dict().items()}, seed=303))
    feats, _, _ = synth_sparseinst_case()
    e = SN.encoder(sd, feats)
    out = SN.decoder(sd, e, groups=cfg.MODEL.SPARSE_INST.DECODER.GROUPS, scale_factor=cfg.MODEL.SPARSE_INST.DECODER.SCALE_FACTOR)
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))
    assert rel(e.numpy()[:, ::8], g["enc_out"]) < 1e-5
    assert rel(out["pred_logits"].numpy(), g["pred_logits"]) < 1e-4
    assert rel(out["pred_scores"].numpy(), g["pred_scores"]) < 1e-4
    assert rel(out["pred_masks"].numpy()[:, ::5, ::2, ::2], g["pred_masks"]) < 1e-4


def test_focal_loss_restatement_against_an_independent_implementation():
    """fvcore.nn.sigmoid_focal_loss_jit (the SparseInst classification loss, loss/sparseinst_loss.py:218-228) is neither vendored
    nor installed; `oracle/ref_loader.py::fvcore_sigmoid_focal_loss` restates it and is what the reference's own criterion
    calls when the goldens are generated.  Cross-check it - and the form the product's criterion uses - against the one
    independent implementation of the same published formula that IS installed: transformers' `sigmoid_focal_loss`
    (returns loss.mean(1).sum() / num_boxes).  Not fvcore itself: parity with it stays unpinned."""
    import ref_loader as RL
    from yolov7_d2_amd.modeling.sparseinst import sigmoid_focal_loss as product_focal
    stub_roots = ("alfred", "cv2", "detectron2", "fvcore", "loguru", "omegaconf", "pycocotools", "torchvision", "timm")
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules)
              if k.split(".")[0] in stub_roots and getattr(sys.modules[k], "__spec__", 1) is None}
    try:
        pytest.importorskip("transformers")
        from transformers.loss.loss_for_object_detection import sigmoid_focal_loss as hf_focal
    finally:
        sys.modules.update(hidden)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 100, 80, generator=g, dtype=torch.float64) * 4
    t = (torch.rand(4, 100, 80, generator=g) < 0.02).to(torch.float64)
    for alpha, gamma in ((0.25, 2.0), (-1.0, 2.0), (0.5, 1.0)):
        mine = RL.fvcore_sigmoid_focal_loss(x, t, alpha=alpha, gamma=gamma, reduction="none")
        assert mine.shape == x.shape
        theirs = hf_focal(x, t, num_boxes=7.0, alpha=alpha, gamma=gamma)
        assert torch.allclose(mine.mean(1).sum() / 7.0, theirs, rtol=1e-12, atol=0)
        assert torch.allclose(RL.fvcore_sigmoid_focal_loss(x, t, alpha=alpha, gamma=gamma, reduction="sum"), mine.sum(), rtol=1e-12)
    # the criterion's form (alpha 0.25, gamma 2, reduction 'sum'), fp32 as the step runs it
    x32, t32 = x.float().flatten(0, 1), t.float().flatten(0, 1)
    ref = RL.fvcore_sigmoid_focal_loss(x32, t32, alpha=0.25, gamma=2.0, reduction="sum")
    assert torch.allclose(product_focal(x32, t32), ref, rtol=1e-5)
