"""CPU: the C-ABI library loads and exports every symbol include/mi355_det.h declares; argument errors are
returned (not raised, not launched); the reference-shaped config / registry surface works; the product path fails
loudly without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

import yolov7_d2_amd as M
from yolov7_d2_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_GPU = torch.cuda.is_available()


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "mi355_det.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = C.CDLL(L.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(L.EXPORTS) == declared, set(L.EXPORTS) ^ declared


def test_opcode_table_matches_header():
    hdr = open(os.path.join(ROOT, "include", "mi355_det.h")).read()
    enum = re.search(r"enum \{(.*?)MI_OP_COUNT", hdr, re.S).group(1)
    vals = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"MI_OP_([A-Z_0-9]+) = (\d+)", enum))
    assert vals == L.OP


def test_struct_sizes_match_c():
    # sizes computed from the field lists in the header (no padding surprises across the FFI)
    raw = 5 * 8 + (3 + 3 + 2 + 2 + 4 + 4 + 3 * L.MI_MAX_TAPS + 5 + 2) * 4 + 5 * 8 + 2 * 4 + 8 + 2 * 4     # (.. bn_*, xf, xf_write, xf_C)
    assert C.sizeof(L.mi_conv_desc) == (raw + 7) // 8 * 8      # (pointer members: the struct is 8-byte aligned)
    assert C.sizeof(L.mi_cmd) % 8 == 0 and C.sizeof(L.mi_cmd) == 4 + 160 + 32 + 4 + 128 + 32  # op,i[40],f[8],pad,p[16],l[4]
    assert C.sizeof(L.mi_sgd_seg) == 24


def test_struct_layouts_match_the_loaded_library():
    """every public struct: ctypes.sizeof == the C compiler's sizeof inside the library that was loaded"""
    names = ["mi_conv_desc", "mi_wgrad_desc", "mi_wgrad_group", "mi_pack_job", "mi_bias_job", "mi_yolox_loss_desc",
             "mi_detr_loss_desc", "mi_sgd_seg", "mi_cmd", "mi_conv_group", "mi_bn_job", "mi_bn_group", "mi_pil_resize_job", "mi_jpeg_info", "mi_jpeg_job", "mi_bnx"]
    lib = L.lib()
    for i, n in enumerate(names):
        assert lib.mi_abi_sizeof(i) == C.sizeof(getattr(L, n)), n
    assert lib.mi_abi_sizeof(len(names)) == -1


def _conv_desc(H, W, K, Cout, taps, stride=1, N=16):
    d = L.mi_conv_desc()
    d.x = d.w = d.y = 4096          # non-null, aligned dummies: the planners never touch memory
    d.ldx, d.ldy, d.N, d.H, d.W = K, Cout, N, H, W
    d.outH, d.outW, d.gridH, d.gridW = H // stride, W // stride, H // stride, W // stride
    d.in_stride, d.out_stride, d.K8, d.Cout, d.CoutPad, d.ntaps = stride, 1, K // 8, Cout, Cout, taps
    offs = [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)] if taps == 9 else [(0, 0)]
    for i, (a, b) in enumerate(offs):
        d.tap_dy[i], d.tap_dx[i], d.tap_w[i] = a, b, i
    return d


def test_conv_launcher_configurations_are_valid():
    """host side of mi_conv2d (no GPU): for every conv shape class of YOLOX-s and a few odd ones the
    launcher's choice is a legal configuration: k-chunk divides K, taps-per-step divides the taps, the cout tile divides
    CoutPad, the pixel tile has <= 128 pixels"""
    lib = L.lib()
    shapes = [(320, 320, 16, 32, 9, 1), (320, 320, 32, 64, 9, 2), (160, 160, 64, 32, 1, 1), (160, 160, 32, 32, 9, 1),
              (160, 160, 64, 128, 9, 2), (80, 80, 128, 64, 1, 1), (80, 80, 64, 64, 9, 1), (80, 80, 128, 128, 9, 1),
              (80, 80, 128, 256, 9, 2), (40, 40, 256, 128, 1, 1), (40, 40, 128, 128, 9, 1), (40, 40, 256, 512, 9, 2),
              (20, 20, 512, 256, 1, 1), (20, 20, 1024, 512, 1, 1), (20, 20, 256, 256, 9, 1), (20, 20, 128, 96, 1, 1),
              (13, 17, 48, 160, 9, 1), (7, 5, 64, 32, 1, 1)]
    for (H, W, K, Co, taps, s) in shapes:
        d = _conv_desc(H, W, K, Co, taps, s)
        n = lib.mi_conv2d_plan(C.byref(d))
        assert n > 0, (H, W, K, Co, taps, s, lib.mi_last_error())
        assert d.KC in (16, 32, 64, 128) and K % d.KC == 0
        assert d.TPS >= 1 and taps % d.TPS == 0
        assert d.BN in (32, 64, 128) and d.CoutPad % d.BN == 0
        assert 1 <= d.TH * d.TW <= 128
        assert n == d.N * -(-d.gridH // d.TH) * -(-d.gridW // d.TW)
    bad = _conv_desc(20, 20, 64, 64, 1)
    bad.KC, bad.TPS = 48, 1                          # a forced k-chunk that is not a kernel configuration
    assert lib.mi_conv2d_plan(C.byref(bad)) < 0 and b"KC" in lib.mi_last_error()


def test_ws_and_stream_group_plans_host_side():
    """mi_conv2d_group_plan (no GPU) hands 3x3 K -> K jobs to the weight-stationary kernel (KC == -2: one persistent
    block per CU shared between the jobs in proportion to their tiles, 96 KB halo ring) and 1x1 jobs of ONE input to the
    streaming kernel (KC == -1); MI_CONV_WS=0 / MI_CONV_STREAM=0 and ineligible shapes fall back to the tile planner"""
    lib = L.lib()
    descs = (L.mi_conv_desc * 3)()
    tiles = 0
    for d, hw in zip(descs, (80, 40, 20)):
        C.memmove(C.byref(d), C.byref(_conv_desc(hw, hw, 128, 128, 9)), C.sizeof(L.mi_conv_desc))
        tiles += d.N * -(-hw // 8) * -(-hw // 16)
    meta = L.mi_conv_group()
    assert lib.mi_conv2d_group_plan(descs, 3, None, 0, C.byref(meta)) == 0, lib.mi_last_error()
    assert meta.KC == -2 and meta.njobs == 3 and meta.BN == 128 and meta.lds_bytes == 2 * 16 * 192 * 16
    assert meta.nblocks == min(256, tiles)
    descs[2].K8 = 32                                  # 256 channels: not a weight-stationary shape -> tile planner (and it
    assert lib.mi_conv2d_group_plan(descs, 3, None, 0, C.byref(meta)) == 0 and meta.KC > 0   # shares the 128-channel config)
    pair = (L.mi_conv_desc * 2)()
    for d in pair:
        C.memmove(C.byref(d), C.byref(_conv_desc(80, 80, 128, 64, 1)), C.sizeof(L.mi_conv_desc))
    assert lib.mi_conv2d_group_plan(pair, 2, None, 0, C.byref(meta)) == 0 and meta.KC == -1 and meta.BN == 128
    pair[1].x += 4096                                 # another input tensor: two launches' worth of work, tile planner
    assert lib.mi_conv2d_group_plan(pair, 2, None, 0, C.byref(meta)) == 0 and meta.KC > 0


def test_conv_group_planner_host_side(monkeypatch):
    """mi_conv2d_group_plan (no GPU), tile kernel: the head's three 3x3 level convs share one configuration, the smaller
    maps get tile shapes inside the leading job's LDS footprint, block ranges are contiguous; jobs that cannot share a
    configuration or mix accumulate / plain launches are refused"""
    monkeypatch.setenv("MI_CONV_WS", "0")
    lib = L.lib()
    descs = (L.mi_conv_desc * 3)()
    for d, hw in zip(descs, (80, 40, 20)):
        C.memmove(C.byref(d), C.byref(_conv_desc(hw, hw, 128, 128, 9)), C.sizeof(L.mi_conv_desc))
    meta = L.mi_conv_group()
    assert lib.mi_conv2d_group_plan(descs, 3, None, 0, C.byref(meta)) == 0, lib.mi_last_error()
    assert meta.njobs == 3 and meta.KC in (32, 64, 128) and meta.BN in (64, 128) and meta.lds_bytes <= 80 * 1024 + 4096
    host = (C.c_char * meta.table_bytes)()
    assert lib.mi_conv2d_group_plan(descs, 3, host, meta.table_bytes, C.byref(meta)) == 0
    starts = (C.c_int * 4).from_buffer_copy(bytes(host)[meta.starts_off: meta.starts_off + 16])
    assert starts[0] == 0 and starts[3] == meta.nblocks and starts[0] < starts[1] < starts[2] < starts[3]
    one = L.mi_conv_desc.from_buffer_copy(descs[0])      # the leading job alone needs the same blocks as in the group
    n0 = lib.mi_conv2d_plan(C.byref(one))
    assert starts[1] == n0 * (one.CoutPad // one.BN)
    assert lib.mi_conv2d_group_plan(descs, 3, host, 8, C.byref(meta)) < 0            # table too small
    descs[1].flags = L.MI_CONV_ACCUM
    assert lib.mi_conv2d_group_plan(descs, 3, None, 0, C.byref(meta)) < 0            # accumulate mixed with plain
    descs[1].flags = 0
    descs[2].K8 = 6                                                                  # K = 48: not a multiple of the k-chunk
    assert lib.mi_conv2d_group_plan(descs, 3, None, 0, C.byref(meta)) < 0


def test_bn_group_planner_host_side():
    lib = L.lib()
    jobs = (L.mi_bn_job * 2)()
    for j, (npix, Cc) in zip(jobs, ((16 * 1600, 128), (16 * 400, 256))):
        for f in ("y", "a", "scale", "shift", "acc", "gamma", "beta", "mean", "invstd"):
            setattr(j, f, 4096)
        j.npix = j.count = npix
        j.C, j.ldy, j.lda, j.act, j.nslots, j.eps, j.momentum = Cc, Cc, Cc, 1, 16, 1e-3, 0.03
    meta = L.mi_bn_group()
    assert lib.mi_bn_group_plan(0, jobs, 2, None, 0, C.byref(meta)) == 0, lib.mi_last_error()
    assert meta.kind == 0 and meta.njobs == 2 and meta.act == 1
    # (512 items per block: ew_blocks in csrc/bn_act.hip, round 4)
    assert meta.nblocks == (16 * 1600 * 16 + 511) // 512 + (16 * 400 * 32 + 511) // 512
    jobs[1].act = 0
    assert lib.mi_bn_group_plan(0, jobs, 2, None, 0, C.byref(meta)) < 0              # the jobs must agree on the activation
    jobs[1].act, jobs[1].C = 1, 100
    assert lib.mi_bn_group_plan(0, jobs, 2, None, 0, C.byref(meta)) < 0              # C must be a multiple of 8 dividing 2048


def test_argument_errors_without_launch():
    lib = L.lib()
    d = L.mi_conv_desc()
    assert lib.mi_conv2d(C.byref(d), None) == -1
    assert b"null" in lib.mi_last_error()
    w = L.mi_wgrad_desc()
    assert lib.mi_conv2d_wgrad(C.byref(w), None) == -1
    assert lib.mi_bn_act_fwd(None, 8, None, 0, 0, None, None, 1e-3, 0.03, None, None, None, None, None, None, None, None, 0, None, 8, 10, 8, 1,
                             None) == -1
    d.x = d.w = d.y = 256
    d.N, d.H, d.W, d.outH, d.outW, d.gridH, d.gridW = 1, 8, 8, 8, 8, 8, 8
    d.in_stride = d.out_stride = 1
    d.K8, d.Cout, d.CoutPad, d.ntaps, d.ldx, d.ldy = 3, 32, 32, 1, 24, 32   # odd K8
    assert lib.mi_conv2d_plan(C.byref(d)) == -1 and b"K8" in lib.mi_last_error()


@pytest.mark.skipif(HAVE_GPU, reason="CPU-only behaviour")
def test_fails_loudly_without_device():
    lib = L.lib()
    assert lib.mi_device_count() == 0
    cmds = (L.mi_cmd * 1)()
    assert lib.mi_cmdlist_run(cmds, 1, None) == -3        # MI_ENODEV
    model = M.build_model(M.yolox_s_cfg(device="cpu"))
    with pytest.raises(L.MI355Error):
        model([{"image": torch.zeros(3, 64, 64, dtype=torch.uint8)}])
    with pytest.raises(RuntimeError):
        model.backbone.stem.conv(torch.zeros(1, 12, 8, 8))   # no eager path for the blocks
    with pytest.raises(L.MI355Error):
        M.batched_nms(torch.zeros(1, 4), torch.zeros(1), torch.zeros(1), 0.5)


def test_registry_and_config_surface(tmp_path):
    assert "YOLOX" in M.META_ARCH_REGISTRY and "build_cspdarknetx_backbone" in M.BACKBONE_REGISTRY
    base = tmp_path / "Base-YOLOv7.yaml"
    base.write_text("MODEL:\n  META_ARCHITECTURE: \"YOLOV7\"\n  PADDED_VALUE: 114.0\nSOLVER:\n  BASE_LR: 0.02\nVERSION: 2\n")
    sub = tmp_path / "coco"
    sub.mkdir()
    y = sub / "yolox_s.yaml"
    # the reference's yolox_s.yaml spells its base "Base-YoloV7.yaml" (case differs from the file on disk)
    y.write_text("_BASE_: \"../Base-YoloV7.yaml\"\nMODEL:\n  META_ARCHITECTURE: \"YOLOX\"\n  BACKBONE:\n    NAME: \"build_cspdarknetx_backbone\"\n"
                 "  YOLO:\n    CLASSES: 80\n    WIDTH_MUL: 0.50\n    DEPTH_MUL: 0.33\n    CONF_THRESHOLD: 0.001\n    NMS_THRESHOLD: 0.65\n"
                 "  SOME_OTHER_FAMILY:\n    KEY: 1\nSOLVER:\n  AMP:\n    ENABLED: true\n")
    cfg = M.get_yolox_cfg(str(y), ["MODEL.DEVICE", "cpu"])
    assert cfg.MODEL.META_ARCHITECTURE == "YOLOX" and cfg.SOLVER.BASE_LR == 0.02 and cfg.MODEL.YOLO.MAX_BOXES_NUM == 100
    assert cfg.SOLVER.REFERENCE_WORLD_SIZE == 8 and cfg.MODEL.DEVICE == "cpu"
    model = M.build_model(cfg)
    keys = list(model.state_dict().keys())
    assert len(keys) == 462 and sum(p.numel() for p in model.parameters()) == 8968255
    assert keys[0] == "backbone.stem.conv.conv.weight" and "neck.C3_p4.m.0.conv1.bn.running_mean" in keys
    assert "head.cls_preds.0.bias" in keys
    bn = model.backbone.dark3[0].bn
    assert bn.eps == 1e-3 and bn.momentum == 0.03
    import math
    assert abs(float(model.head.obj_preds[1].bias[0].detach()) + math.log(99.0)) < 1e-6
    assert model.backbone.size_divisibility == 32 and model.backbone.output_shape()["dark5"].channels == 512


def test_preprocess_matches_reference_contract():
    from yolov7_d2_amd.d2shim import Boxes, Instances
    model = M.build_model(M.yolox_s_cfg(device="cpu"))
    a = {"image": torch.full((3, 50, 70), 7, dtype=torch.uint8),
         "instances": Instances((50, 70), gt_boxes=Boxes(torch.tensor([[10., 20, 30, 60]])), gt_classes=torch.tensor([4]))}
    b = {"image": torch.full((3, 64, 40), 9, dtype=torch.uint8),
         "instances": Instances((64, 40), gt_boxes=Boxes(torch.zeros(0, 4)), gt_classes=torch.zeros(0, dtype=torch.long))}
    images, labels, sizes = model.preprocess_image([a, b], True)
    assert images.tensor.shape == (2, 3, 64, 96) and sizes == [(50, 70), (64, 40)]
    assert float(images.tensor[0, 0, 0, 0]) == 7 and float(images.tensor[0, 0, 63, 95]) == 114.0   # pad value, no mean/std
    assert labels.shape == (2, 100, 5)
    assert labels[0, 0].tolist() == [4.0, 20.0, 40.0, 20.0, 40.0] and float(labels[1].abs().sum()) == 0
