"""CPU: the BiFPN drop-in builds from the reference's config keys, carries the reference's state_dict keys / shapes (the
golden records the reference's own), and refuses to run without the device path."""
import os

import numpy as np
import pytest
import torch

from gen_golden_inputs import BIFPN_CASES, synth_bifpn_case
from yolov7_d2_amd import _lib as L
from yolov7_d2_amd.config import add_yolo_config, get_cfg
from yolov7_d2_amd.d2shim import Backbone, ShapeSpec, build_backbone
from yolov7_d2_amd.modeling import bifpn as B


class _Feats(Backbone):
    def __init__(self, chans):
        super().__init__()
        self.chans = chans

    def output_shape(self):
        return {f"res{i + 3}": ShapeSpec(channels=c, stride=8 << i) for i, c in enumerate(self.chans)}

    def forward(self, x):
        return x


@pytest.mark.parametrize("name", list(BIFPN_CASES))
def test_state_dict_keys_are_the_references(golden_dir, name):
    gold = np.load(os.path.join(golden_dir, "bifpn.npz"))
    kw = BIFPN_CASES[name]
    feats, _ = synth_bifpn_case(out_channels=kw["out_channels"])
    net = B.BiFPN(cfg=None, bottom_up=_Feats([v.shape[1] for v in feats.values()]), in_features=list(feats.keys()), norm="GN",
                  num_levels=5, **kw)
    assert [f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items()] == list(gold[name + "_keys"])
    with pytest.raises(L.MI355Error):
        net(feats)


def test_config_keys_and_builder():
    cfg = add_yolo_config(get_cfg())
    b = cfg.MODEL.BIFPN                                   # yolov7/config.py:34-39
    assert (b.NUM_LEVELS, b.NUM_BIFPN, b.NORM, b.OUT_CHANNELS, b.SEPARABLE_CONV) == (5, 6, "GN", 160, False)
    cfg.MODEL.BACKBONE.NAME = "build_resnet_bifpn_backbone"
    cfg.MODEL.RESNETS.OUT_FEATURES = ["res3", "res4", "res5"]
    cfg.MODEL.FPN.IN_FEATURES = ["res3", "res4", "res5"]
    net = build_backbone(cfg)
    assert len(net.cell) == 6 and len(net.resample) == 2 and net.size_divisibility == 128
    assert sorted(net.output_shape()) == ["p3", "p4", "p5", "p6", "p7"]
    first = net.cell[0].fnode[0].combine
    assert first.inputs_offsets == [3, 4] and tuple(first.edge_weights.shape) == (2,)
    assert net.cell[0].fnode[3].combine.resample["0"].conv.conv.weight.shape == (160, 512, 1, 1)
    with pytest.raises(NotImplementedError):
        B.get_norm("BN", 64)


def test_symbols_exported():
    lib = L.lib()
    for s in ("mi_groupnorm_fwd", "mi_groupnorm_bwd", "mi_groupnorm_ws_bytes", "mi_maxpool2x2_fwd", "mi_maxpool2x2_bwd",
              "mi_fastattn_fwd", "mi_fastattn_bwd", "mi_fastattn_ws_bytes"):
        assert hasattr(lib, s)
