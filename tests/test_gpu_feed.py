"""GPU (-m gpu): the host-fed training step (NativeTrainer.feed, the .to(device) of meta_arch/yolox.py:96,183 made
asynchronous).  A step fed from pinned host memory - staged forward graphs whose Focus packer reads the staging buffer in
place - must compute what the step on a resident batch computes: the first step's losses bit for bit (same kernels, same
operands), later steps to the tolerance of the fp64 BatchNorm accumulation order."""
import os

import numpy as np
import pytest
import torch

import yolox_oracle as O
import yolov7_d2_amd as M

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(seed=0):
    model = M.build_model(M.yolox_s_cfg(device=DEV))
    model.load_state_dict(O.init_state_dict(0.33, 0.5, 80, seed=seed))
    return model


def _batches(n, B, H, W):
    out = []
    for k in range(n):
        imgs, labels = O.synth_batch(B, H, W, seed=40 + k, max_gt=5)
        out.append((imgs.to(torch.uint8), labels))
    return out


def _resident_losses(batches, use_graph):
    from yolov7_d2_amd.engine import NativeTrainer
    tr = NativeTrainer(_model(), lr=0.002, use_graph=use_graph, input_u8=True)
    hist = []
    for imgs, labels in batches:
        st = tr.load_batch(imgs.to(DEV), labels.to(DEV))
        tr.step(st)
        hist.append(tr.losses(st)[:4].numpy().copy())
    return np.stack(hist)


def _fed_losses(batches, use_graph, direct):
    from yolov7_d2_amd.engine import NativeTrainer
    prev = os.environ.get("MI_FEED_DIRECT")
    os.environ["MI_FEED_DIRECT"] = "1" if direct else "0"
    try:
        tr = NativeTrainer(_model(), lr=0.002, use_graph=use_graph, input_u8=True)
    finally:
        if prev is None:
            os.environ.pop("MI_FEED_DIRECT", None)
        else:
            os.environ["MI_FEED_DIRECT"] = prev
    assert tr.feed_direct == direct
    B, _, H, W = batches[0][0].shape
    st = tr._state(B, H, W)
    pinned = [(i.pin_memory(), l.pin_memory()) for i, l in batches]
    hist = []
    tr.feed(st, *pinned[0])
    for k in range(len(pinned)):
        tr.step(st)
        if k + 1 < len(pinned):
            tr.feed(st, *pinned[k + 1])        # the next batch travels while this step computes
        hist.append(tr.losses(st)[:4].numpy().copy())
    return np.stack(hist), tr, st


@pytest.mark.parametrize("use_graph", [False, True])
def test_fed_step_equals_resident_step(use_graph):
    batches = _batches(6, 4, 128, 128)
    ref = _resident_losses(batches, use_graph)
    for direct in (True, False):
        got, tr, st = _fed_losses(batches, use_graph, direct)
        assert np.all(np.isfinite(got))
        assert np.array_equal(got[0], ref[0]), (direct, got[0], ref[0])       # same kernels on the same operands
        np.testing.assert_allclose(got, ref, rtol=2e-3, err_msg=f"direct={direct}")
        if direct and use_graph:
            # one forward graph per staging buffer, both used; the plan's own image buffer was never written
            assert sorted(st["graphs"]["fwd_stage"]) == [0, 1]
            assert int(st["ps"].image.max()) == 0


def test_staged_forward_list_patches_exactly_the_focus_source():
    from yolov7_d2_amd import _lib as L
    from yolov7_d2_amd.engine import NativeTrainer
    tr = NativeTrainer(_model(), lr=0.0, use_graph=False, input_u8=True)
    st = tr._state(2, 64, 96)
    imgs, labels = O.synth_batch(2, 64, 96, seed=3, max_gt=4)
    tr.feed(st, imgs.to(torch.uint8).pin_memory(), labels.pin_memory())
    farr, fn = st["plan"].fwd_cmds
    for k in (0, 1):
        arr, n = tr._staged_fwd(st, k)
        assert n == fn + 1 and arr[0].op == L.OP["COPY"]
        assert arr[0].p[0] == st["stage"][k]["lab_flat"].data_ptr() and arr[0].p[1] == st["ps"].labels_flat.data_ptr()
        assert arr[0].l[0] * 16 == st["ps"].labels_flat.numel() * 4
        diff = [j for j in range(fn) if bytes(arr[j + 1]) != bytes(farr[j])]
        assert len(diff) == 1 and arr[diff[0] + 1].op == L.OP["FOCUS"]
        assert arr[diff[0] + 1].p[0] == st["stage"][k]["img"].data_ptr()
    with pytest.raises(ValueError):
        tr.feed(st, imgs[:1].to(torch.uint8).pin_memory(), labels[:1].pin_memory())
