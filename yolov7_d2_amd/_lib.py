"""ctypes binding of libmi355det.so (the C-ABI declared in include/mi355_det.h).

The library is the product: if it is missing, or no HIP device is present, every compute entry
fails loudly (`MI355Error`) — there is no CPU fallback on the product path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_LIB", os.path.join(_HERE, "libmi355det.so"))   # override: A/B testing of builds

MI_MAX_TAPS = 16
MI_CONV_ACCUM = 1
MI_CONV_RELU = 8
MI_CONV_RELUMASK = 16
MI_CONV_ADDRELU = 32
MI_CONV_BNBWD = 4
MI_CONV_OUT_F32 = 2
MI_BN_BAR_WORDS = 64 * (1 + 2 * 16)
MI_BN_SLOTS = 16
MI_MAX_AUX = 4
MI_WGRAD_STREAM = MI_MAX_AUX   # the last auxiliary stream is created with the lowest priority (background work)


class MI355Error(RuntimeError):
    pass


class mi_conv_desc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("bias", C.c_void_p),
        ("stats_acc", C.c_void_p),
        ("ldx", C.c_int32), ("ldy", C.c_int32), ("y_nstride", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("outH", C.c_int32), ("outW", C.c_int32), ("gridH", C.c_int32), ("gridW", C.c_int32),
        ("in_stride", C.c_int32), ("out_stride", C.c_int32), ("out_oy", C.c_int32), ("out_ox", C.c_int32),
        ("K8", C.c_int32), ("Cout", C.c_int32), ("CoutPad", C.c_int32), ("ntaps", C.c_int32),
        ("tap_dy", C.c_int32 * MI_MAX_TAPS), ("tap_dx", C.c_int32 * MI_MAX_TAPS),
        ("tap_w", C.c_int32 * MI_MAX_TAPS),
        ("flags", C.c_int32), ("TH", C.c_int32), ("TW", C.c_int32), ("KC", C.c_int32), ("BN", C.c_int32),
        ("stats_slots", C.c_int32), ("TPS", C.c_int32),
        ("bn_y", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("bn_mean", C.c_void_p),
        ("bn_invstd", C.c_void_p), ("bn_ldy", C.c_int32), ("bn_act", C.c_int32),
        ("xf", C.c_void_p), ("xf_write", C.c_int32), ("xf_C", C.c_int32),
    ]


class mi_bnx(C.Structure):
    """device record behind mi_conv_desc.xf: BatchNorm(train) + activation of a convolution's INPUT applied by the
    consuming launch (csrc/conv_bn.h BnXf)"""
    _fields_ = [
        ("res_unused", C.c_void_p), ("a", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("rmean", C.c_void_p), ("rvar", C.c_void_p), ("nbt", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("mean", C.c_void_p), ("invstd", C.c_void_p),
        ("ldres_unused", C.c_int32), ("lda", C.c_int32), ("act", C.c_int32), ("pad0_", C.c_int32),
        ("inv_count", C.c_double), ("unbias", C.c_double), ("eps", C.c_float), ("momentum", C.c_float),
        ("acc", C.c_void_p), ("sld", C.c_int32), ("nslots", C.c_int32), ("C", C.c_int32), ("pad1_", C.c_int32),
    ]


class mi_detr_loss_desc(C.Structure):
    _fields_ = [
        ("logits", C.c_void_p), ("boxes", C.c_void_p), ("tgt_labels", C.c_void_p), ("tgt_boxes", C.c_void_p),
        ("tgt_off", C.c_void_p), ("match_q", C.c_void_p), ("match_t", C.c_void_p), ("nmatch", C.c_void_p),
        ("B", C.c_int32), ("Q", C.c_int32), ("NC", C.c_int32), ("gmax", C.c_int32),
        ("eos_coef", C.c_float), ("num_boxes", C.c_float),
        ("losses", C.c_void_p), ("rowstate", C.c_void_p),
    ]


class mi_conv_group(C.Structure):
    _fields_ = [("njobs", C.c_int32), ("nblocks", C.c_int32), ("lds_bytes", C.c_int32), ("KC", C.c_int32),
                ("BN", C.c_int32), ("TPIX", C.c_int32), ("TPS", C.c_int32), ("EPI", C.c_int32),
                ("starts_off", C.c_int64), ("table_bytes", C.c_int64), ("priv", C.c_int64 * 288)]


class mi_adamw_tensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("count", C.c_int64),
                ("lr", C.c_float), ("weight_decay", C.c_float)]


class mi_adamw_chunk(C.Structure):
    _fields_ = [("tensor", C.c_int32), ("count", C.c_int32), ("offset", C.c_int64)]


class mi_bn_job(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("y", "res", "a", "da", "dy", "dres", "acc", "gamma", "beta", "rmean", "rvar",
                                          "nbt", "scale", "shift", "mean", "invstd", "dgamma", "dbeta")] + \
               [("npix", C.c_int64), ("count", C.c_int64)] + \
               [(n, C.c_int32) for n in ("ldy", "ldres", "lda", "ldda", "lddy", "lddres", "dres_accum", "C", "nslots",
                                         "nblk", "act", "pad_")] + [("eps", C.c_float), ("momentum", C.c_float), ("bar", C.c_void_p)]


class mi_bn_group(C.Structure):
    _fields_ = [("kind", C.c_int32), ("njobs", C.c_int32), ("nblocks", C.c_int32), ("act", C.c_int32),
                ("starts_off", C.c_int64), ("table_bytes", C.c_int64)]


class mi_wgrad_desc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("dy", C.c_void_p), ("gw", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("ldx", C.c_int32), ("ldy", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("outH", C.c_int32), ("outW", C.c_int32),
        ("stride", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("CinPad", C.c_int32), ("CoutPad", C.c_int32),
        ("ntaps", C.c_int32),
        ("tap_dy", C.c_int32 * MI_MAX_TAPS), ("tap_dx", C.c_int32 * MI_MAX_TAPS),
        ("accumulate", C.c_int32),
        ("TH", C.c_int32), ("TW", C.c_int32), ("splitk", C.c_int32), ("cfg_tp", C.c_int32), ("cfg_ns", C.c_int32),
        ("row_scale", C.c_void_p), ("gbias", C.c_void_p),
    ]


class mi_yolox_loss_desc(C.Structure):
    _fields_ = [
        ("preds", C.c_void_p), ("labels", C.c_void_p), ("anchors", C.c_void_p),
        ("B", C.c_int32), ("A", C.c_int32), ("ncls", C.c_int32), ("max_labels", C.c_int32),
        ("gmax", C.c_int32),
        ("cost", C.c_void_p), ("iou", C.c_void_p), ("match", C.c_void_p), ("ngt", C.c_void_p),
        ("fg", C.c_void_p), ("matched_gt", C.c_void_p), ("matched_iou", C.c_void_p),
        ("partial", C.c_void_p), ("out", C.c_void_p), ("use_l1", C.c_int32), ("rsv_", C.c_int32), ("partial_l1", C.c_void_p),
        ("center_radius", C.c_float), ("cls_weight", C.c_float), ("iou_weight", C.c_float), ("reg_weight", C.c_float),
        ("iou_type", C.c_int32), ("rsv2_", C.c_int32),
    ]


class mi_sgd_seg(C.Structure):
    _fields_ = [("offset", C.c_int64), ("count", C.c_int64), ("weight_decay", C.c_float), ("lr", C.c_float)]


MI_WGRAD_MAX_GROUPS = 32


class _mi_wgrad_group_g(C.Structure):
    _fields_ = [("cfg", C.c_int32 * 6), ("njobs", C.c_int32), ("nblocks", C.c_int32), ("lds_bytes", C.c_int32),
                ("fixup", C.c_int32), ("job_off", C.c_int64), ("starts_off", C.c_int64)]


class mi_wgrad_group(C.Structure):
    _fields_ = [("ngroups", C.c_int32), ("nred", C.c_int32), ("red_blocks", C.c_int32), ("pad_", C.c_int32),
                ("g", _mi_wgrad_group_g * MI_WGRAD_MAX_GROUPS),
                ("red_off", C.c_int64), ("red_starts_off", C.c_int64), ("table_bytes", C.c_int64),
                ("ws_bytes", C.c_int64), ("red9_off", C.c_int64), ("red9_starts_off", C.c_int64),
                ("nred9", C.c_int32), ("red9_blocks", C.c_int32)]


class mi_bias_job(C.Structure):
    _fields_ = [("out", C.c_void_p), ("a0", C.c_int32), ("HW", C.c_int32), ("c0", C.c_int32), ("nc", C.c_int32)]


class mi_split_job(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("a0", C.c_int32), ("HW", C.c_int32), ("c0", C.c_int32), ("nc", C.c_int32),
                ("ld", C.c_int32), ("rsv_", C.c_int32)]


class mi_pack_job(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wf", C.c_void_p), ("wd", C.c_void_p),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("KK", C.c_int32), ("CinPad", C.c_int32),
                ("CoutPad", C.c_int32), ("CoutPadK", C.c_int32), ("CinPadN", C.c_int32), ("blk0", C.c_int32),
                ("scale", C.c_void_p)]


class mi_sparseinst_loss_desc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("logits", "scores", "labels", "match_q", "match_t", "nmatch", "inv_num", "stats", "gup",
                                          "pairs", "valid", "row_cls", "row_pair", "kdev", "losses", "dlogits", "dscores", "coef")] + \
               [(n, C.c_int32) for n in ("B", "N", "C", "cap", "P", "use_labels", "use_masks", "pad_")] + \
               [(n, C.c_float) for n in ("alpha", "gamma", "w_ce", "w_mask", "w_dice", "w_obj")]


class mi_image_job(C.Structure):
    _fields_ = [("src", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32), ("dtype", C.c_int32), ("pad_", C.c_int32)]


class mi_mask_job(C.Structure):
    _fields_ = [("masks", C.c_void_p), ("labels", C.c_void_p), ("M", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("dtype", C.c_int32)]


class mi_mosaic_paste_job(C.Structure):
    _fields_ = [("src", C.c_void_p), ("canvas", C.c_void_p)] + [(n, C.c_int32) for n in
                ("h0", "w0", "rh", "rw", "cw", "x1a", "y1a", "x2a", "y2a", "x1b", "y1b", "blk0", "fsrc", "pad_")]


class mi_warp_job(C.Structure):
    _fields_ = [("canvas", C.c_void_p), ("out", C.c_void_p), ("minv", C.c_double * 6)] + [(n, C.c_int32) for n in
                ("ch", "cw", "h", "w", "Hp", "Wp", "border", "blk0")]


class mi_mixup_job(C.Structure):
    _fields_ = [("src", C.c_void_p), ("out", C.c_void_p)] + [(n, C.c_int32) for n in
                ("h0", "w0", "rh1", "rw1", "dh", "dw", "oh", "ow", "flip", "x_off", "y_off", "th", "tw", "Hp", "Wp", "blk0", "fsrc", "pad_")]


class mi_pil_resize_job(C.Structure):
    _fields_ = ([("src", C.c_void_p), ("tmp", C.c_void_p), ("dst", C.c_void_p), ("dsc", C.c_int64), ("dsy", C.c_int64),
                 ("dsx", C.c_int64), ("src_ld", C.c_int64), ("sat_src", C.c_double)] +
                [(n, C.c_int32) for n in ("h0", "w0", "nh", "nw", "hflip", "vflip", "shift_x", "shift_y", "src_hflip", "color")] +
                [("sat_dst", C.c_float), ("bri_dst", C.c_float), ("dis_hue", C.c_float), ("dis_sat", C.c_float), ("dis_exp", C.c_float),
                 ("dis_pos", C.c_int32), ("blk0h", C.c_int32), ("blk0v", C.c_int32)])


class mi_jpeg_info(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("width", "height", "ncomp", "restart_interval", "orientation", "adobe_transform",
                                         "progressive", "pad0_")] +
                [(n, C.c_int32 * 3) for n in ("comp_id", "hs", "vs", "tq")] +
                [(n, C.c_int32) for n in ("hmax", "vmax", "mcu_w", "mcu_h")] +
                [("blocks_w", C.c_int32 * 3), ("blocks_h", C.c_int32 * 3), ("coef_off", C.c_int64 * 3), ("coef_count", C.c_int64),
                 ("sos_pos", C.c_int64), ("qt", (C.c_uint16 * 64) * 4), ("dc_bits", (C.c_uint8 * 17) * 4),
                 ("dc_vals", (C.c_uint8 * 256) * 4), ("ac_bits", (C.c_uint8 * 17) * 4), ("ac_vals", (C.c_uint8 * 256) * 4),
                 ("have_qt", C.c_uint8 * 4), ("have_dc", C.c_uint8 * 4), ("have_ac", C.c_uint8 * 4), ("pad_", C.c_uint8 * 4)])


class mi_jpeg_job(C.Structure):
    _fields_ = ([("coef", C.c_void_p), ("planes", C.c_void_p), ("out", C.c_void_p), ("coef_off", C.c_int64 * 3),
                 ("plane_off", C.c_int64 * 3)] +
                [(n, C.c_int32) for n in ("width", "height", "ncomp", "orientation", "bgr", "ycc")] +
                [(n, C.c_int32 * 3) for n in ("hs", "vs", "blocks_w", "blocks_h")] +
                [(n, C.c_int32) for n in ("hmax", "vmax", "blk0_idct", "blk0_pix")] + [("qt", (C.c_uint16 * 64) * 3)])


class mi_cmd(C.Structure):
    _fields_ = [("op", C.c_int32), ("i", C.c_int32 * 40), ("f", C.c_float * 8), ("p", C.c_void_p * 16),
                ("l", C.c_int64 * 4)]


# opcode names must match the enum in include/mi355_det.h
OPS = ["NOP", "CONV", "WGRAD", "PACK_W", "RESERVED4", "RESERVED5", "BN_ACT_FWD", "BN_BWD_REDUCE",
       "RESERVED8", "BN_BWD_APPLY", "FOCUS", "UPSAMPLE_FWD", "UPSAMPLE_BWD", "SPP_FWD", "SPP_BWD", "COPY",
       "COLSUM", "LOSS_FWD", "LOSS_BWD", "SPLIT_DPREDS", "MEMSET", "SGD", "BN_EVAL_AFFINE", "DECODE", "PACK_W_BATCH", "WGRAD_GROUP", "STREAM", "FORK", "JOIN", "BIAS_GRADS",
       "CONV_GROUP", "BN_GROUP", "SPLIT_DPREDS_BATCH", "BN_BWD_FUSED", "DWCONV_FWD", "DWCONV_DGRAD", "DWCONV_WGRAD",
       "LOSS_BWD_FUSED"]
OP = {n: k for k, n in enumerate(OPS)}

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_PROTOS = {
    "mi_version": (C.c_int, []),
    "mi_device_count": (C.c_int, []),
    "mi_last_error": (C.c_char_p, []),
    "mi_conv2d": (C.c_int, [C.POINTER(mi_conv_desc), _vp]),
    "mi_conv2d_plan": (C.c_int, [C.POINTER(mi_conv_desc)]),
    "mi_conv2d_wgrad": (C.c_int, [C.POINTER(mi_wgrad_desc), _vp]),
    "mi_conv2d_wgrad_group_plan": (C.c_int, [C.POINTER(mi_wgrad_desc), _i, _vp, _vp, _i64, C.POINTER(mi_wgrad_group)]),
    "mi_conv2d_wgrad_group_run": (C.c_int, [C.POINTER(mi_wgrad_group), _vp, _vp]),
    "mi_pack_conv_weight": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    "mi_pack_conv_weight_scaled": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    "mi_scale_rows_f32": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp]),
    "mi_conv2d_wgrad_plan": (C.c_int64, [C.POINTER(mi_wgrad_desc)]),
    "mi_bn_eval_affine": (C.c_int, [_vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    "mi_bn_act_fwd": (C.c_int, [_vp, _i, _vp, _i, _i64, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i,
                                _i64, _i, _i, _vp]),
    "mi_bn_act_bwd_reduce": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i, _vp]),
    "mi_bn_act_bwd_apply": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, _i, _vp, _i, _i,
                                      _i64, _i, _i, _vp]),
    "mi_focus_pack": (C.c_int, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "mi_focus_pack_u8": (C.c_int, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "mi_maxpool3x3s2_fwd": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "mi_maxpool3x3s2_bwd": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mi_maxpool3x3s2_fwd_idx": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "mi_maxpool3x3s2_bwd_idx": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mi_upsample2x_fwd": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "mi_upsample2x_bwd": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mi_spp_pool_fwd": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "mi_spp_pool_bwd": (C.c_int, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mi_copy_bf16": (C.c_int, [_vp, _i, _vp, _i, _i, _i64, _i, _vp]),
    "mi_colsum_bf16": (C.c_int, [_vp, _i, _i64, _i, _vp, _i, _vp, _vp]),
    "mi_colsum_bf16_wide": (C.c_int, [_vp, _i, _i64, _i, _vp, _i, _vp, _vp]),
    "mi_colsumsq_bf16_wide": (C.c_int, [_vp, _i, _i64, _i, _vp, _i, _vp, _vp]),
    "mi_colsum_wide_ws_bytes": (C.c_int64, [_i]),
    "mi_pack_conv_weights_batch": (C.c_int, [_vp, _i, _i, _i, _vp]),
    "mi_pack_jobs_layout": (C.c_int, [C.POINTER(mi_pack_job), _i]),
    "mi_yolox_loss_fwd": (C.c_int, [C.POINTER(mi_yolox_loss_desc), _vp]),
    "mi_yolox_loss_bwd": (C.c_int, [C.POINTER(mi_yolox_loss_desc), _vp, _vp, _vp]),
    "mi_yolox_loss_bwd_fused": (C.c_int, [C.POINTER(mi_yolox_loss_desc), _vp, _vp, C.POINTER(mi_split_job), _i,
                                          C.POINTER(mi_bias_job), _i, _vp, _i64, _vp]),
    "mi_yolox_split_dpreds": (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mi_yolox_split_dpreds_batch": (C.c_int, [_vp, _i, _i, _i, C.POINTER(mi_split_job), _i, _vp]),
    "mi_yolox_bias_grads": (C.c_int, [_vp, _i, _i, _i, C.POINTER(mi_bias_job), _i, _vp, _vp]),
    "mi_yolox_decode": (C.c_int, [_vp, _vp, _i, _i, _i, _vp]),
    "mi_hungarian_match": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "mi_lsap": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mi_conv2d_group_plan": (C.c_int, [C.POINTER(mi_conv_desc), _i, _vp, _i64, C.POINTER(mi_conv_group)]),
    "mi_conv2d_group_run": (C.c_int, [C.POINTER(mi_conv_group), _vp, _vp]),
    "mi_conv2d_route": (C.c_int, [C.POINTER(mi_conv_desc)]),
    "mi_dropout_seed_offset": (C.c_int, [_vp]),
    "mi_adamw_step_multi": (C.c_int, [_vp, _vp, _i, _f, _f, _f, _vp, _f, _vp]),
    "mi_adamw_step_multi_clip": (C.c_int, [_vp, _vp, _i, _f, _f, _f, _vp, _f, _vp, _vp]),
    "mi_grad_norm_multi": (C.c_int, [_vp, _vp, _i, _vp, _f, _f, _vp, _vp]),
    "mi_grad_gather_multi": (C.c_int, [_vp, _vp, _i, _vp, _vp, _vp]),
    "mi_conv2d_bn_plan": (C.c_int, [C.POINTER(mi_conv_desc), C.POINTER(mi_bn_job), _i, C.POINTER(mi_conv_group)]),
    "mi_conv2d_bn_fwd": (C.c_int, [C.POINTER(mi_conv_desc), C.POINTER(mi_bn_job), _i, _vp]),
    "mi_conv_bn_barrier_status": (C.c_int, [C.POINTER(C.c_uint32)]),
    "mi_conv1x1_stream": (C.c_int, [C.POINTER(mi_conv_desc), _i, _vp]),
    "mi_conv3x3_ws": (C.c_int, [C.POINTER(mi_conv_desc), _i, _vp]),
    "mi_bn_act_bwd_fused": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, _i, _vp, _i,
                                      _i, _i64, _i, _i, _vp, _vp]),
    "mi_bn_fused_set_capacity": (C.c_int, [_i]),
    "mi_dwconv3x3_fwd": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mi_dwconv3x3_dgrad": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mi_dwconv3x3_wgrad_ws_bytes": (C.c_int64, [_i]),
    "mi_dwconv3x3_wgrad": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _vp]),
    "mi_batched_nms_ex": (C.c_int, [_vp, _vp, _vp, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_batched_softnms": (C.c_int, [_vp, _vp, _vp, _i, _f, _f, _i, _vp, _vp, _vp]),
    "mi_mask_nms": (C.c_int, [_vp, _vp, _vp, _i, _f, _vp, _vp]),
    "mi_matrix_nms": (C.c_int, [_vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp, _vp]),
    "mi_groupnorm_ws_bytes": (C.c_int64, [_i, _i]),
    "mi_groupnorm_fwd": (C.c_int, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp]),
    "mi_groupnorm_bwd": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "mi_maxpool2x2_fwd": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "mi_maxpool2x2_bwd": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "mi_fastattn_ws_bytes": (C.c_int64, []),
    "mi_fastattn_fwd": (C.c_int, [C.POINTER(C.c_void_p), _i, _vp, _vp, _i64, _vp]),
    "mi_fastattn_bwd": (C.c_int, [C.POINTER(C.c_void_p), _i, _vp, _vp, C.POINTER(C.c_void_p), _vp, _vp, _i64, _vp]),
    "mi_mosaic_jobs_layout": (C.c_int, [C.POINTER(mi_mosaic_paste_job), _i, C.POINTER(mi_warp_job), _i]),
    "mi_mosaic_paste": (C.c_int, [_vp, _i, _i, _vp]),
    "mi_warp_affine_u8": (C.c_int, [_vp, _i, _i, _vp]),
    "mi_mixup_jobs_layout": (C.c_int, [C.POINTER(mi_mixup_job), _i]),
    "mi_mixup_blend": (C.c_int, [_vp, _i, _i, _vp]),
    "mi_pil_resize_jobs_layout": (C.c_int, [C.POINTER(mi_pil_resize_job), _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mi_pil_resize_h": (C.c_int, [_vp, _i, _i, _vp]),
    "mi_pil_resize_v": (C.c_int, [_vp, _i, _i, _vp]),
    "mi_jpeg_parse": (C.c_int, [_vp, C.c_int64, C.POINTER(mi_jpeg_info)]),
    "mi_jpeg_huffman": (C.c_int, [_vp, C.c_int64, C.POINTER(mi_jpeg_info), _vp]),
    "mi_jpeg_huffman_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mi_jpeg_job_fill": (C.c_int, [C.POINTER(mi_jpeg_info), _vp, _vp, _vp, _i, _i, C.POINTER(mi_jpeg_job)]),
    "mi_jpeg_jobs_layout": (C.c_int, [C.POINTER(mi_jpeg_job), _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mi_jpeg_idct": (C.c_int, [_vp, _i, _i, _vp]),
    "mi_jpeg_color": (C.c_int, [_vp, _i, _i, _vp]),
    "mi_rle_encode": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mi_rle_to_string": (C.c_int, [_vp, _i, C.c_char_p, _i]),
    "mi_yolox_onnx_layout": (C.c_int, [_vp, _vp, _i, _i, _i, _vp]),
    "mi_yolox_iou_loss": (C.c_int, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "mi_pairwise_bbox_iou": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "mi_box_convert": (C.c_int, [_vp, _vp, _i64, _i, _vp]),
    "mi_box_iou_pairwise": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mi_bn_group_plan": (C.c_int, [_i, C.POINTER(mi_bn_job), _i, _vp, _i64, C.POINTER(mi_bn_group)]),
    "mi_bn_group_run": (C.c_int, [C.POINTER(mi_bn_group), _vp, _vp]),
    "mi_detr_set_loss_fwd": (C.c_int, [C.POINTER(mi_detr_loss_desc), _vp]),
    "mi_detr_set_loss_bwd": (C.c_int, [C.POINTER(mi_detr_loss_desc), _vp, _vp, _vp, _vp]),
    "mi_mha_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "mi_mha_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "mi_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "mi_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mi_ew_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _i, _vp]),
    "mi_dropout_bf16": (C.c_int, [_vp, _vp, _i64, _f, C.c_uint64, _vp]),
    "mi_dropout_add_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _f, C.c_uint64, _vp]),
    "mi_dropout_add_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, C.c_uint64, _vp]),
    "mi_layernorm_bwd_dropout": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, C.c_uint64, _vp]),
    "mi_bilinear_resize_bf16": (C.c_int, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "mi_bilinear_resize_bwd_bf16": (C.c_int, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "mi_sparseinst_mask_stats": (C.c_int, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "mi_sparseinst_mask_stats_ws_floats": (C.c_int64, [_i, _i]),
    "mi_sparseinst_mask_grad": (C.c_int, [_vp, _i, _i, _vp, _vp, _i, _vp, _f, _f, _vp, _vp]),
    "mi_sparseinst_mask_grad_dev": (C.c_int, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "mi_mha_fwd_dropout": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, C.c_uint64, _vp]),
    "mi_mha_bwd_dropout": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f,
                                     C.c_uint64, _vp]),
    "mi_mha_dropout_mask": (C.c_int, [_vp, _i, _i, _i, _i, _f, C.c_uint64, _vp]),
    "mi_mha_fwd_dropout_o32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, C.c_uint64, _vp]),
    "mi_mha_bwd_dropout_o32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f,
                                        C.c_uint64, _vp]),
    "mi_mha_fwd_dropout_ld": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, C.c_uint64, _vp]),
    "mi_mha_bwd_dropout_ld": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f,
                                       C.c_uint64, _vp]),
    "mi_iou_loss_v6": (C.c_int, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "mi_batched_nms": (C.c_int, [_vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_abi_sizeof": (C.c_int, [_i]),
    "mi_sigmoid_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "mi_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _i64, _f, _vp]),
    "mi_grad_clip_full_model": (C.c_int, [_vp, _i64, _f, _vp, _vp, _vp]),
    "mi_pos_embed_sine": (C.c_int, [_vp, _i, _i, _i, _i, _f, _i, _f, _i, _vp, _vp]),
    "mi_sgd_momentum_step": (C.c_int, [_vp, _vp, _vp, _vp, _i, _f, _f, _i, _vp]),
    "mi_cmdlist_run": (C.c_int, [C.POINTER(mi_cmd), _i, _vp]),
    "mi_stream_create_cu_mask": (C.c_int, [C.POINTER(C.c_uint32), _i, C.POINTER(C.c_void_p)]),
    "mi_stream_destroy": (C.c_int, [_vp]),
    "mi_upload_async": (C.c_int, [_vp, _vp, _i64, _vp]),
    "mi_pyramid_pool_fwd": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "mi_pyramid_pool_bwd": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "mi_sparseinst_match_cost": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _vp, _vp]),
    "mi_sparseinst_pairs": (C.c_int, [_vp, _vp]),
    "mi_sparseinst_head_loss": (C.c_int, [_vp, _vp]),
    "mi_sparseinst_head_loss_bwd": (C.c_int, [_vp, _vp]),
    "mi_padding_masks": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "mi_normalize_pad_batch": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "mi_mask_targets_batch": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_aux_stream_set": (C.c_int, [_i, _vp]),
    "mi_graph_capture": (C.c_int64, [C.POINTER(mi_cmd), _i, _vp]),
    "mi_graph_launch": (C.c_int, [_i64, _vp]),
    "mi_graph_destroy": (C.c_int, [_i64]),
    "mi_cmdlist_time": (C.c_int, [C.POINTER(mi_cmd), _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "mi_probe_mfma32": (C.c_int, [_vp, _vp, _vp, _vp]),
    "mi_probe_mfma16": (C.c_int, [_vp, _vp, _vp, _vp]),
}
EXPORTS = sorted(_PROTOS)

_lib = None


def lib():
    """Load the shared library (once). Raises MI355Error if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MI355Error(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_dbg = None


def dbg():
    """libmi355dbg.so (include/mi355_debug.h): diagnostic kernels for tools/ - not part of the product library"""
    global _dbg
    if _dbg is None:
        path = os.path.join(os.path.dirname(LIB_PATH), "libmi355dbg.so")
        if not os.path.exists(path):
            raise MI355Error(f"{path} not found (make -C yolov7_d2_amd/csrc)")
        D = C.CDLL(path)
        D.mi_debug_code_polluter.restype = C.c_int
        D.mi_debug_code_polluter.argtypes = [_i, _i, _vp]
        D.mi_debug_cu_census.restype = C.c_int
        D.mi_debug_cu_census.argtypes = [_vp, _i, _i, _vp]
        _dbg = D
    return _dbg


def check(rc, what=""):
    if rc is None or rc < 0:
        msg = lib().mi_last_error()
        raise MI355Error(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
    return rc


def require_device():
    n = lib().mi_device_count()
    if n <= 0:
        raise MI355Error("no HIP device visible: the MI355X path cannot run (no CPU fallback by design)")
    return n


def ptr(t):
    """device/host pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream
