"""GPU (-m gpu): the weight-gradient side queue (engine.NativeTrainer, MI_WGRAD_SIDE / MI_WGRAD_CUMASK - an opt-in
experiment, profiles/r06_wgrad_cumask_ab.txt) computes the step: chain pieces and weight-gradient groups as separate
hipGraphs on two streams (one of them CU-masked) land on the same parameters as the one-stream step, bit for bit."""
import ctypes as C
import os

import pytest
import torch

import yolox_oracle as O
import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L

pytestmark = pytest.mark.gpu


def _params_after(monkeypatch, env, use_graph, steps=3):
    from yolov7_d2_amd.engine import NativeTrainer
    for k in ("MI_WGRAD_SIDE", "MI_WGRAD_CUMASK", "MI_MAIN_CUMASK", "MI_WGRAD_ASYNC"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    model = M.build_model(M.yolox_s_cfg(device="cuda"))
    model.load_state_dict(O.init_state_dict(0.33, 0.5, 80, seed=0))
    imgs, labels = O.synth_batch(2, 128, 128, seed=17, max_gt=5)
    try:
        tr = NativeTrainer(model, lr=0.002, use_graph=use_graph)
        st = tr.load_batch(imgs.cuda(), labels.cuda())
        for _ in range(steps):
            tr.step(st)
        losses = tr.losses(st)[:4].clone()
        if env.get("MI_WGRAD_SIDE"):
            kinds = [k for (k, _, _) in tr._side_pieces(*st["plan"].bwd_cmds)]
            assert kinds.count("side") == int(env["MI_WGRAD_SIDE"]) and kinds[-1] == "join" and kinds[0] == "main"
        return tr.params.data.clone(), losses
    finally:
        L.lib().mi_aux_stream_set(L.MI_WGRAD_STREAM, None)
        assert "MI_WGRAD_ASYNC" not in os.environ          # the trainer sets it around its plan build only


@pytest.mark.parametrize("use_graph", [True, False])
def test_side_queue_step_equals_one_stream_step(monkeypatch, use_graph):
    ref, lref = _params_after(monkeypatch, {}, use_graph)
    for env in ({"MI_WGRAD_SIDE": "2"}, {"MI_WGRAD_SIDE": "3", "MI_WGRAD_CUMASK": "64", "MI_MAIN_CUMASK": "1"}):
        got, lgot = _params_after(monkeypatch, env, use_graph)
        assert torch.isfinite(got).all()
        torch.testing.assert_close(lgot, lref, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(got, ref, rtol=0, atol=2e-4 * float(ref.abs().max()))


def test_cu_mask_stream_confines_a_launch():
    """64 low mask bits = 8 CUs of every XCD: a 128-block launch of a conv-free kernel on the masked stream still runs
    (functional check of mi_stream_create_cu_mask / mi_stream_destroy; the census is tools/cu_mask_probe.py)"""
    from yolov7_d2_amd.engine import NativeTrainer
    words = NativeTrainer.cu_mask_words(64)
    assert sum(bin(w).count("1") for w in words) == 64 and len(words) == 8
    inv = NativeTrainer.cu_mask_words(64, invert=True)
    assert all((a ^ b) == 0xFFFFFFFF for a, b in zip(words, inv))
    arr = (C.c_uint32 * 8)(*words)
    out = C.c_void_p()
    L.check(L.lib().mi_stream_create_cu_mask(arr, 8, C.byref(out)), "create")
    s = torch.cuda.ExternalStream(out.value)
    x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    with torch.cuda.stream(s):
        y = (x * 2).sum()
    s.synchronize()
    assert float(y) == float((x * 2).sum())
    L.check(L.lib().mi_stream_destroy(out.value), "destroy")
    zero = (C.c_uint32 * 8)()
    assert L.lib().mi_stream_create_cu_mask(zero, 8, C.byref(out)) < 0
