"""GPU (-m gpu): the other YOLOX width / depth multipliers on the same kernels - channel counts that are not multiples of
32 and whose C/8 is not a power of two (24, 48, 96, 192, 384 / 48 ... 768 / 80 ... 1280): padded pixel strides, zero
weight rows, real-channel BatchNorm with partially used thread blocks.

  * YOLOX-tiny 416 x 416 bs 2 = BASELINE.json configs[0] (there on the CPU device; the product has no CPU path, so the
    configuration runs on the HIP device) against tests/golden/yolox_tiny_step_416.npz, i.e. against the reference's
    OWN modules run by path (oracle/gen_golden.py): losses, eval output, and every parameter gradient with the forward
    state pinned (see tests/test_gpu_parity_bench.py for why);
  * YOLOX-m, YOLOX-l (54 M parameters, up to 1024-channel layers, a 2048-channel SPP concat) and YOLOX-x: one step each
    against the oracle the same way.
Reference: yolov7/modeling/backbone/darknetx.py:103-162 (width / depth multipliers)."""
import os

import numpy as np
import pytest
import torch

import yolox_oracle as O
from parity_util import DEV, build_model, grad_table, hip_step, oracle_backward

pytestmark = pytest.mark.gpu


def _forced_check(sd, imgs, labels, depth, width, nparams):
    hip = hip_step(sd, imgs, labels, depth, width, want_y=True)
    ys = hip.pop("y")
    forced = oracle_backward(sd, imgs, hip["dpreds"], ys, depth, width)
    fe = forced["force_err"]
    assert len(fe) == len(ys)
    worst = max(fe.items(), key=lambda kv: kv[1])
    assert worst[1] < 1e-3, worst                           # every BaseConv output on identical inputs
    rel = float((hip["raw"] - forced["raw"]).norm() / forced["raw"].norm())
    assert rel < 1e-5, rel                                   # raw head output on forced features
    # SimOTA + losses on the raw output the HIP network produced: integers exact, floats 1e-4
    raw = hip["raw"].clone().requires_grad_(True)
    res, assigns = O.yolox_losses(raw, labels, hip["anchors"], 80, return_assign=True)
    np.testing.assert_allclose(hip["losses"][:4].numpy(), np.array([float(x.detach()) for x in res[:4]]), rtol=1e-4,
                               atol=1e-5)
    for b in range(imgs.shape[0]):
        if assigns[b] is not None:
            assert torch.equal(hip["fg"][b].bool(), assigns[b]["fg"])
    rows = grad_table(hip["grads"], forced["grads"])
    assert len(rows) == nparams
    rows.sort(key=lambda r: r[1])
    print("width %.3f: conv out rel max %.1e | grads cos min %.6f rel max %.4f" %
          (width, worst[1], rows[0][1], max(r[2] for r in rows)))
    bad = [r for r in rows if not (r[1] >= 0.999 and r[2] <= 0.05)]
    assert not bad, bad[:6]
    return hip, forced


def test_yolox_tiny_config0_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "yolox_tiny_step_416.npz"))
    depth, width = 0.33, 0.375
    sd = O.init_state_dict(depth, width, 80, seed=0)
    imgs, labels = O.synth_batch(2, 416, 416, seed=31, max_gt=6)      # the batch oracle/gen_golden.py used
    hip, forced = _forced_check(sd, imgs, labels, depth, width, 240)
    # the reference's own numbers (fp32): the whole bf16 step lands within the storage noise of them
    np.testing.assert_allclose(hip["losses"][0].numpy(), g["losses"][0], rtol=3e-2)
    np.testing.assert_allclose(hip["losses"][:4].numpy(), g["losses"][:4], rtol=1e-1, atol=5e-2)
    gn = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
    ratios = np.array([float(v.norm()) / gn[n] for n, v in hip["grads"].items() if gn[n] > 1e-6])
    assert 0.8 < np.median(ratios) < 1.25
    # eval forward (running statistics of the untouched init = the golden's model): decoded output vs the reference's
    model = build_model(depth, width, sd)
    model.eval()
    ps = model.plan_for(2, 416, 416, False)
    ps.image.copy_(imgs.to(DEV))
    ps.plan.run("fwd")
    torch.cuda.synchronize()
    ev = ps.preds().float().cpu().numpy()[:, ::7]
    ref = g["eval_out_stride"]
    np.testing.assert_allclose(ev[..., :4], ref[..., :4], rtol=5e-2, atol=1.0)      # boxes, px
    np.testing.assert_allclose(ev[..., 4:], ref[..., 4:], rtol=5e-2, atol=5e-3)     # probabilities


@pytest.mark.parametrize("name,depth,width", [("m", 0.67, 0.75), ("l", 1.0, 1.0), ("x", 1.33, 1.25)])
def test_yolox_m_l_x_training_step(name, depth, width):
    sd = O.init_state_dict(depth, width, 80, seed=2)
    n = sum(1 for k, v in sd.items() if v.is_floating_point() and "running" not in k)
    imgs, labels = O.synth_batch(2, 128, 160, seed=13, max_gt=5)
    _forced_check(sd, imgs, labels, depth, width, n)
