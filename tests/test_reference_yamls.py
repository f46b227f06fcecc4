"""CPU, only where /root/reference exists (the build container): the drop-in claim of BASELINE.json's north_star, asserted.
The reference's REAL config files - configs/coco/yolox_s.yaml, configs/coco/detr/detr_256_6_6_torchvision.yaml and all ten
configs/coco/sparseinst/*.yaml - merge through yolov7_d2_amd.config unchanged, build_model(cfg) constructs the registered
meta-architecture, and its state_dict has exactly the keys and shapes of the reference's own modules built from the SAME
cfg object by path (oracle/ref_loader.py).  detectron2's ResNet is un-vendored: on both sides the backbone entries come from
this repository's restatement (oracle/resnet_oracle.py for the reference's Detr), so only the non-backbone keys are the
reference's own."""
import glob
import os
import sys
import types

import pytest
import torch

import ref_loader
import yolov7_d2_amd as M

REF_CFG = "/root/reference/configs/coco"
pytestmark = pytest.mark.skipif(not (ref_loader.available() and os.path.isdir(REF_CFG)),
                                reason="the reference tree is not on this machine")


def _shapes(sd):
    return {k: tuple(v.shape) for k, v in sd.items()}


def test_yolox_s_yaml_builds_the_reference_state_dict():
    cfg = M.get_yolox_cfg(os.path.join(REF_CFG, "yolox_s.yaml"), ["MODEL.DEVICE", "cpu"])
    assert cfg.MODEL.META_ARCHITECTURE == "YOLOX" and cfg.MODEL.BACKBONE.NAME == "build_cspdarknetx_backbone"
    model = M.build_model(cfg)
    assert type(model).__name__ == "YOLOX"
    ref, _ = ref_loader.build_reference_yolox(cfg.MODEL.YOLO.DEPTH_MUL, cfg.MODEL.YOLO.WIDTH_MUL, cfg.MODEL.YOLO.CLASSES,
                                              depthwise=cfg.MODEL.DARKNET.DEPTH_WISE)
    ours, theirs = _shapes(model.state_dict()), _shapes(ref.state_dict())
    assert list(ours) == list(theirs)               # same keys in the same order (what a checkpoint loader walks)
    assert ours == theirs
    # the YAML's solver / input keys the trainer and the mapper read arrive too
    assert cfg.SOLVER.IMS_PER_BATCH == 112 and cfg.INPUT.MOSAIC_AND_MIXUP.ENABLED is True
    ref.load_state_dict(model.state_dict())         # and the reference accepts our checkpoint as is


def test_detr_yaml_builds_the_reference_state_dict():
    import resnet_oracle as R
    from yolov7_d2_amd import d2shim
    cfg = M.get_yolox_cfg(os.path.join(REF_CFG, "detr", "detr_256_6_6_torchvision.yaml"), ["MODEL.DEVICE", "cpu"])
    assert cfg.MODEL.META_ARCHITECTURE == "Detr" and cfg.MODEL.DETR.NUM_OBJECT_QUERIES == 100
    assert cfg.SOLVER.IMS_PER_BATCH == 56 and cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE == "full_model"
    model = M.build_model(cfg)
    assert type(model).__name__ == "Detr"
    ref_loader.load()
    det = ref_loader.load_detr()
    det.build_backbone = lambda c: R.R50Module(50, c.MODEL.RESNETS.OUT_FEATURES, c.MODEL.RESNETS.STRIDE_IN_1X1)
    det.ImageList, det.Instances, det.Boxes = d2shim.ImageList, d2shim.Instances, d2shim.Boxes
    det.detector_postprocess = d2shim.detector_postprocess
    ref = det.Detr(cfg)                              # the reference's own class reads the SAME merged cfg
    ours, theirs = _shapes(model.state_dict()), _shapes(ref.state_dict())
    assert sorted(ours) == sorted(theirs), (sorted(set(ours) ^ set(theirs))[:8])
    assert ours == theirs
    own = [k for k in theirs if ".backbone.0.backbone." not in k]
    assert len(own) > 150 and any(k.startswith("detr.transformer.encoder.layers.5.") for k in own)
    assert set(ref.criterion.weight_dict) == set(model.criterion.weight_dict)
    assert all(abs(ref.criterion.weight_dict[k] - model.criterion.weight_dict[k]) < 1e-12 for k in ref.criterion.weight_dict)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(REF_CFG, "sparseinst", "*.yaml"))))
def test_sparseinst_yamls_merge(path):
    cfg = M.get_yolox_cfg(path, ["MODEL.DEVICE", "cpu"])
    if os.path.basename(path).startswith("Base-"):
        assert cfg.MODEL.META_ARCHITECTURE == "SparseInst"
        return
    assert cfg.MODEL.META_ARCHITECTURE == "SparseInst" and cfg.MODEL.SPARSE_INST.DECODER.NAME


def test_sparseinst_r50_giam_yaml_builds_the_reference_state_dict():
    cfg = M.get_yolox_cfg(os.path.join(REF_CFG, "sparseinst", "sparse_inst_r50_giam.yaml"), ["MODEL.DEVICE", "cpu"])
    assert cfg.MODEL.SPARSE_INST.ENCODER.NAME == "InstanceContextEncoder" and cfg.MODEL.SPARSE_INST.DECODER.NAME == "GroupIAMDecoder"
    model = M.build_model(cfg)
    assert type(model).__name__ == "SparseInst"
    si = ref_loader.load_sparseinst()
    shapes = {n: types.SimpleNamespace(channels=c, stride=s) for n, c, s in (("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    enc = si.encoder.InstanceContextEncoder(cfg, shapes)        # the reference's classes on the SAME merged cfg
    dec = si.decoder.GroupIAMDecoder(cfg)
    theirs = _shapes(torch.nn.ModuleDict(dict(encoder=enc, decoder=dec)).state_dict())
    ours = {k: v for k, v in _shapes(model.state_dict()).items() if k.startswith(("encoder.", "decoder."))}
    assert sorted(ours) == sorted(theirs), (sorted(set(ours) ^ set(theirs))[:8])
    assert ours == theirs
    crit = si.loss.SparseInstCriterion(cfg, si.loss.SparseInstMatcher(cfg))
    assert dict(crit.weight_dict) == dict(model.criterion.weight_dict)
    # the backbone entries follow detectron2's ResNet naming (un-vendored; restated in modeling/resnet.py)
    assert "backbone.res5.2.conv3.weight" in model.state_dict() and "backbone.stem.conv1.norm.running_var" in model.state_dict()
