/*
 * mi355_det.h — C-ABI of libmi355det.so: the MI355X (gfx950) hot path of
 * yolov7_d2's YOLOX data-parallel training step.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes and a
 * hipStream_t (passed as void*), never allocates, frees or synchronises, and
 * returns 0 on success or a negative MI_E* code for an argument error detected
 * BEFORE any launch.  The caller (PyTorch caching allocator, or any other
 * runtime) owns every buffer including workspaces.
 *
 * Each function names the reference interface it replaces (file:line relative
 * to lucasjinreal/yolov7_d2).  The reference has no native code of its own on
 * this path: these replace the ATen/cuDNN/torchvision calls the reference
 * reaches through torch.
 *
 * Layout conventions
 *   activations : bf16, NHWC ("channels_last"); a tensor is (ptr, ld) where ld is
 *                 the element stride between consecutive pixels, so a channel
 *                 slice of a concat buffer is just (ptr + c0, ld_of_buffer).
 *   conv weights: fp32 OIHW masters (the nn.Parameter layout of the reference);
 *                 mi_pack_conv_weight() produces the bf16 k8-major images the
 *                 conv kernels read: [tap][K/8][CoutPad][8].
 *   head preds  : fp32 [B][A][5+ncls] exactly as yolox_head.py:238-241 flattens.
 */
#ifndef MI355_DET_H
#define MI355_DET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument (shape / alignment / unsupported config) */
#define MI_ELAUNCH (-2)  /* hipGetLastError() != success after a launch          */
#define MI_ENODEV (-3)   /* no HIP device                                          */

typedef void* mi_stream_t; /* hipStream_t */

/* ---- library ---------------------------------------------------------- */
int mi_version(void);
/* number of visible HIP devices (0 => every compute entry returns MI_ENODEV) */
int mi_device_count(void);
const char* mi_last_error(void);

/* ---- convolution (implicit GEMM on MFMA, im2col-free, LDS halo tile) ---
 * replaces nn.Conv2d fwd/dgrad inside BaseConv (layers/wrappers.py:60-83) and
 * the biased 1x1 prediction convs (head/yolox_head.py:103-129).
 * One descriptor covers: 1x1, 3x3 s1, 3x3 s2 forward; their data gradients
 * (s2 dgrad = 4 parity-class launches with out_stride 2) via the tap table. */
#define MI_CONV_ACCUM 1    /* y += result (gradient fan-in)                  */
#define MI_CONV_OUT_F32 2  /* y is fp32 (prediction maps), else bf16         */
#define MI_CONV_RELU 8     /* y = max(result + bias, 0): the ReLU behind a Conv2d + FrozenBatchNorm2d of detectron2's ResNet
                              (forward, bf16 output, no MI_CONV_ACCUM / statistics); tile kernel only              */
#define MI_CONV_RELUMASK 16 /* y = result where aux > 0, else 0; aux = bn_y (bf16 NHWC, the shape of y, pixel stride bn_ldy): the
                              OUTPUT of the ReLU this data gradient flows back through - ReLU backward in the epilogue     */
#define MI_CONV_ADDRELU 32  /* y = max(bf16(result + bias) + aux, 0) rounded as mi_ew_bf16 op 7: conv3 + shortcut + ReLU of a
                              bottleneck block (detectron2 BottleneckBlock.forward) in the epilogue; aux = bn_y           */
#define MI_CONV_BNBWD 4    /* data-gradient launch that also reduces the BatchNorm backward sums of the layer that
                              PRODUCED its output tensor: stats_acc += (sum dz, sum dz*xhat) per channel with
                              dz = y_out * act'(bn_y*scale+shift), xhat = (bn_y-mean)*invstd - replaces the
                              mi_bn_act_bwd_reduce pass over (da, y) of that layer (bf16 staged outputs only) */
#define MI_MAX_TAPS 16 /* 3x3 = 9; 16 = the 7x7 stride-2 ResNet stem as a 4x4 conv over the 2x2 space-to-depth image */
#define MI_BN_SLOTS 16     /* max accumulator slots per channel (slot = pixel tile % nslots); callers pick
                              nslots per layer: more slots = less same-address atomic traffic in the conv
                              epilogue, fewer = a shorter statistics prologue in the BN kernels */

typedef struct mi_conv_desc {
  const void* x;   /* bf16 NHWC input view, >= K8*8 channels readable        */
  const void* w;   /* packed bf16 [n_wslabs][K8][CoutPad][8]                  */
  void* y;         /* output view                                             */
  const float* bias;    /* [Cout] or NULL                                     */
  double* stats_acc;    /* [stats_slots][CoutPad][2] fp64 (sum, sumsq) accumulators, atomically added
                           (caller zeroes them once per step), or NULL             */
  int32_t ldx, ldy;
  int32_t y_nstride;        /* elements between images of y; 0 => outH*outW*ldy */
  int32_t N, H, W;          /* input dims                                     */
  int32_t outH, outW;       /* real output tensor dims                        */
  int32_t gridH, gridW;     /* output sub-grid iterated (==outH,outW if out_stride==1) */
  int32_t in_stride;        /* input pixel  = grid*in_stride  + tap offset    */
  int32_t out_stride, out_oy, out_ox; /* output pixel = grid*out_stride + off */
  int32_t K8;               /* padded input channels / 8                      */
  int32_t Cout, CoutPad;
  int32_t ntaps;
  int32_t tap_dy[MI_MAX_TAPS], tap_dx[MI_MAX_TAPS], tap_w[MI_MAX_TAPS];
  int32_t flags;
  int32_t TH, TW;           /* pixel tile; 0 => chosen by the launcher        */
  int32_t KC, BN;           /* k-chunk / cout tile; 0 => chosen by launcher   */
  int32_t stats_slots;      /* 1..MI_BN_SLOTS accumulator slots (0 => MI_BN_SLOTS) */
  int32_t TPS;              /* taps multiplied per main-loop step (divides ntaps); 0 => chosen by launcher */
  /* MI_CONV_BNBWD only: raw conv output of the producing layer (bf16 NHWC, same N x outH x outW x Cout as y),
   * its BatchNorm scale / shift / mean / invstd [Cout] and activation (1 = SiLU, 0 = none) */
  const void* bn_y;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  int32_t bn_ldy, bn_act;
  /* BatchNorm(train) + activation of the INPUT, applied by this launch ("BN in the consumer", csrc/conv_bn.h): x is the
   * RAW output of the producing convolution and xf a DEVICE record (mi_bnx) with that layer's accumulators, parameters and
   * the activated tensor `a`; the launch derives scale / shift itself, transforms the tiles it fetched in LDS and - when
   * xf_write - stores a for the tensor's later readers, records scale / shift / mean / invstd and updates the running
   * statistics: mi_bn_act_fwd(x -> a) followed by this convolution on a, bit for bit, in one launch.  Forward
   * convolutions with stats_acc on the streaming 1x1 / weight-stationary 3x3 kernels only (mi_conv2d returns MI_EINVAL
   * otherwise); jobs of one launch that read the same x share the record and exactly one of them sets xf_write. */
  const void* xf;
  int32_t xf_write, xf_C;   /* xf_C: the record's channel count (== K8 * 8) */
} mi_conv_desc;

/* device record behind mi_conv_desc.xf (host-built, uploaded by the caller; all pointers are device pointers) */
typedef struct mi_bnx {
  const void* res_unused; void* a;
  const float* gamma; const float* beta; float* rmean; float* rvar; int64_t* nbt;
  float* scale; float* shift; float* mean; float* invstd;
  int32_t ldres_unused, lda, act, pad0_;
  double inv_count, unbias;
  float eps, momentum;
  const double* acc;        /* the producer's fp64 (sum, sumsq) accumulators [nslots][sld / 2][2] */
  int32_t sld, nslots, C, pad1_;
} mi_bnx;

int mi_conv2d(const mi_conv_desc* d, mi_stream_t s);
/* fills TH/TW/KC/BN/TPS if zero; returns number of pixel tiles or <0 */
int mi_conv2d_plan(mi_conv_desc* d);
/* the kernel family mi_conv2d runs for this descriptor: 0 tile kernel (conv_igemm), 1 mi_conv1x1_stream, 2 mi_conv3x3_ws;
 * -1 when the descriptor carries an input BatchNorm (xf) and neither 1 nor 2 applies (mi_conv2d would refuse it) */
int mi_conv2d_route(const mi_conv_desc* d);

/* several independent convolutions in ONE launch (the FPN levels of the head: the 40x40 / 20x20 launches are
 * latency-bound alone).  All jobs run the template configuration chosen for the job with the most output pixels, so
 * their K must be a multiple of its k-chunk, CoutPad of its cout tile, and they must agree on ntaps-divisibility and
 * on MI_CONV_ACCUM / MI_CONV_BNBWD.  _plan() writes the device job table into table_host (caller uploads it; NULL / 0
 * to query meta->table_bytes); _run() launches it. */
#define MI_CONV_MAX_GROUP 8
typedef struct mi_conv_group {
  int32_t njobs, nblocks, lds_bytes;
  int32_t KC, BN, TPIX, TPS, EPI;
  int64_t starts_off, table_bytes;
  int64_t priv[288];        /* KC == -1: the jobs are 1x1 convolutions of one input and run as ONE streaming launch
                               (mi_conv1x1_stream); KC == -2: 3x3 K -> K convolutions as ONE weight-stationary launch
                               (mi_conv3x3_ws); the launch record lives here, no device table is read */
} mi_conv_group;
int mi_conv2d_group_plan(const mi_conv_desc* descs, int n, void* table_host, int64_t table_cap, mi_conv_group* meta);
int mi_conv2d_group_run(const mi_conv_group* meta, const void* table_dev, mi_stream_t s);

/* Convolution + train-mode BatchNorm + activation (+ residual) as ONE launch: the whole BaseConv.forward
 * (backbone/layers/wrappers.py:76-83) and the Bottleneck shortcut (:119-123).  descs[j] must carry stats_acc and bn[j]
 * must be exactly the mi_bn_act_fwd that would follow it (y / ldy / C / acc / nslots / npix == count == N*H*W of the
 * conv's output).  When the convolutions run on the streaming 1x1 kernel (<= 2 convs of one input) or on the weight-
 * stationary 3x3 kernel (<= 8 jobs), the BatchNorm pass is the second phase of the SAME persistent launch behind a grid
 * barrier: every block applies scale / shift / SiLU to the output pieces it stored itself.  Bit-identical to
 * mi_conv2d followed by mi_bn_act_fwd.
 * _plan: returns 1 and fills meta (run it with mi_conv2d_group_run(meta, NULL, s)) when such a launch exists, 0 when the
 *        caller has to keep the two launches (tile-kernel shapes, MI_CONV_BN_FUSE=0), < 0 on bad arguments.
 * _fwd:  plan + run; MI_EINVAL when no fused launch exists (it never falls back).
 * mi_conv_bn_barrier_status: bit 0 / 1 set when a block of a streaming / weight-stationary launch ever gave up waiting at
 *        the grid barrier (its outputs were written as NaN); synchronises the device. */
struct mi_bn_job;
int mi_conv2d_bn_plan(const mi_conv_desc* descs, const struct mi_bn_job* bn, int n, mi_conv_group* meta);
int mi_conv2d_bn_fwd(const mi_conv_desc* descs, const struct mi_bn_job* bn, int n, mi_stream_t s);
int mi_conv_bn_barrier_status(uint32_t* flags);

/* streaming 1x1 convolution (csrc/conv1x1_stream.h): n >= 1 bf16 1x1 stride-1 convolutions that read the SAME input
 * view (x, ldx, N, H, W, K8 equal; K in {32, 64, 128, 256, 512}; Cout == CoutPad, a multiple of 32; all with stats_acc,
 * all with MI_CONV_ACCUM, or all plain) as one persistent launch: weights stay in registers, the input is read once,
 * BatchNorm statistics leave as one set of atomics per block.  Round 6: descriptors with an fp32 bias, MI_CONV_RELU,
 * MI_CONV_ADDRELU or MI_CONV_RELUMASK (the same flags for all n; bn_y / bn_ldy = the second tensor) are served too, and -
 * for the launches without statistics - pixel counts N * H * W that are no multiple of the pixel tile (>= 8192 pixels) and
 * ONE convolution with more than 512 output channels (a multiple of 512: 512 at a time).  Replaces the 1x1 nn.Conv2d forward / data
 * gradient of CSPLayer / Bottleneck / SPP / head stems (layers/wrappers.py:60-83,150-197).  mi_conv2d and
 * mi_conv2d_group_plan take this path by themselves for eligible descriptors (MI_CONV_STREAM=0 disables that);
 * this entry returns MI_EINVAL instead of falling back. */
int mi_conv1x1_stream(const mi_conv_desc* descs, int n, mi_stream_t s);

/* weight-stationary 3x3 convolution (csrc/conv3x3_ws.h): 1..8 bf16 3x3 stride-1 convolutions with Cin == Cout == K in
 * {32, 64, 128} (the same K for all; their own tensors and shapes), no bias, all plain / all with stats_acc / all
 * MI_CONV_ACCUM, as ONE launch of persistent blocks: each wave keeps its 32 output channels x 9 taps x K of weights in
 * registers, only the input halo tiles move (LDS-DMA, double-buffered).  Forward and data gradient (any tap order) of
 * Bottleneck conv2 and of the YOLOXHead cls / reg towers (layers/wrappers.py:105-123, head/yolox_head.py:73-102).
 * mi_conv2d and mi_conv2d_group_plan take this path by themselves (MI_CONV_WS=0 disables that); this entry returns
 * MI_EINVAL instead of falling back. */
int mi_conv3x3_ws(const mi_conv_desc* descs, int n, mi_stream_t s);

/* weight gradient: g[co][ci][tap] (fp32 OIHW, the nn.Parameter gradient layout; overwritten, or += if
 * `accumulate`) = sum_pixels dy[p][co] * x[p*stride + tap][ci].  replaces conv wgrad of the same modules.
 * Split-K over pixel tiles with a caller-owned fp32 workspace (no atomics: the split partials are summed
 * in a fixed order, so the result is bit-reproducible); mi_conv2d_wgrad_plan() returns the bytes needed. */
typedef struct mi_wgrad_desc {
  const void* x;  /* bf16 NHWC input view  (CinPad channels readable)        */
  const void* dy; /* bf16 NHWC out-grad view (CoutPad channels readable)     */
  float* gw;      /* fp32 [Cout][Cin][ntaps]                                  */
  void* ws;       /* split-K workspace, 16-byte aligned                       */
  int64_t ws_bytes;
  int32_t ldx, ldy;
  int32_t N, H, W, outH, outW;
  int32_t stride;
  int32_t Cin, Cout;        /* real channel counts written to gw              */
  int32_t CinPad, CoutPad;  /* readable (zero-padded) channels, CinPad%16==0, CoutPad%32==0 */
  int32_t ntaps;            /* 1 or 9 */
  int32_t tap_dy[MI_MAX_TAPS], tap_dx[MI_MAX_TAPS];
  int32_t accumulate;
  int32_t TH, TW, splitk, cfg_tp, cfg_ns; /* 0 => chosen by launcher (pixel tile, split-K, tile pixels, LDS stages) */
  const float* row_scale; /* NULL, or fp32 [Cout]: gw[co] (+)= row_scale[co] * dW[co] - the gradient of a weight whose
                           * image carried a folded per-Cout factor (mi_pack_conv_weight_scaled), applied to the fp32
                           * sum in the split-K reduction */
  float* gbias;           /* NULL, or fp32 [Cout]: the BIAS gradient = column sums of dy over all pixels, written by the
                           * same two launches (the blocks of input-channel tile 0 add up the dy rows they stream, the
                           * split-K reduction adds the splits in a fixed order) instead of a separate column-sum launch;
                           * mi_conv2d_wgrad only (single layers: nn.Linear / biased nn.Conv2d of DETR and SparseInst);
                           * mi_conv2d_wgrad_plan then returns the workspace INCLUDING nsplit * CoutPad floats for it */
} mi_wgrad_desc;
int mi_conv2d_wgrad(const mi_wgrad_desc* d, mi_stream_t s);
/* workspace bytes mi_conv2d_wgrad needs for this descriptor (pointers may be NULL), or <0 */
int64_t mi_conv2d_wgrad_plan(const mi_wgrad_desc* d);

/* grouped form: all layers of a step in one grid per tile configuration + one reduce grid (weight gradients are
 * not consumed before the optimizer step, so they can all run at the end of backward; each layer keeps its own
 * out-gradient buffer).  _plan() lays out per-layer workspaces from ws_base, writes the device job table into
 * table_host (caller uploads it; pass NULL/0 to query sizes) and fills *meta; _run() launches it. */
#define MI_WGRAD_MAX_GROUPS 32
typedef struct mi_wgrad_group {
  int32_t ngroups, nred, red_blocks, pad_;
  struct {
    int32_t cfg[6];
    int32_t njobs, nblocks, lds_bytes, fixup; /* fixup = 1: the launch sums its split-K partials itself (MI_WG_FIXUP=1;
                                               * the table then starts with the tile counters and carries no reduce jobs) */
    int64_t job_off, starts_off;
  } g[MI_WGRAD_MAX_GROUPS];
  int64_t red_off, red_starts_off, table_bytes, ws_bytes;
  /* second reduce grid: the 3x3 layers (one 576-thread block per 16x16 fragment tile x 9 taps) */
  int64_t red9_off, red9_starts_off;
  int32_t nred9, red9_blocks;
} mi_wgrad_group;
int mi_conv2d_wgrad_group_plan(const mi_wgrad_desc* descs, int n, void* ws_base, void* table_host,
                               int64_t table_cap, mi_wgrad_group* meta);
int mi_conv2d_wgrad_group_run(const mi_wgrad_group* meta, const void* table_dev, mi_stream_t s);
/* (round 6) jobs of a group may carry gbias: the bias gradient (column sums of dy) then leaves with the group's launches -
 * partial rows behind the job's split slabs, summed in a fixed order by extra blocks of the reduce grid.  Used by the
 * per-layer grouped weight gradients of the transformer / ResNet blocks (yolov7_d2_amd/ops.py WgradBatch). */
/* asynchronous host -> device copy of a job table on stream s (hipMemcpyAsync; `src_pinned` must be page-locked and must
 * stay unchanged while a captured graph that recorded this copy can replay) */
int mi_upload_async(void* dst_dev, const void* src_pinned, int64_t nbytes, mi_stream_t s);

/* OIHW fp32 master -> packed bf16 images.  wf: forward [KH*KW][CinPad/8][CoutPad][8];
 * wd: dgrad  [KH*KW][CoutPadK/8][CinPadN][8] (roles swapped).  Either may be NULL. */
int mi_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW,
                        void* wf, int CinPad, int CoutPad,
                        void* wd, int CoutPadK, int CinPadN, mi_stream_t s);

/* the same with a per-output-channel fp32 factor folded in before the bf16 rounding: detectron2's Conv2d(norm =
 * FrozenBatchNorm2d) as the reference's DETR / SparseInst backbones build it (W * scale[co]; un-vendored d2
 * layers/batch_norm.py FrozenBatchNorm2d + layers/wrappers.py Conv2d) without a folded fp32 copy of the weight */
int mi_pack_conv_weight_scaled(const float* w_oihw, const float* cout_scale, int Cout, int Cin, int KH, int KW,
                               void* wf, int CinPad, int CoutPad,
                               void* wd, int CoutPadK, int CinPadN, mi_stream_t s);
/* out[r][:] = g[r][:] * scale[r], fp32 [rows][rowlen]: the weight gradient of such a layer (dW = scale[co] * dW') */
int mi_scale_rows_f32(const float* g, const float* scale, float* out, int rows, int rowlen, mi_stream_t s);

/* all layers of a step in ONE launch: jobs_dev is a device array of njobs records (same meaning as the
 * arguments of mi_pack_conv_weight) */
typedef struct mi_pack_job {
  const float* w;
  void* wf;
  void* wd;
  int32_t Cout, Cin, KK, CinPad, CoutPad, CoutPadK, CinPadN;
  int32_t blk0; /* first block of this job in the flat launch: filled by mi_pack_jobs_layout */
  const float* scale; /* NULL, or fp32 [Cout] folded into both images as mi_pack_conv_weight_scaled does (W * scale[co]) */
} mi_pack_job;
/* validates the (host) job table, fills blk0 and returns the number of blocks of the flat launch (<0 on error) */
int mi_pack_jobs_layout(mi_pack_job* jobs_host, int njobs);
/* kk_max: the largest tap count (KK) of the table: sizes the kernel's LDS tile */
int mi_pack_conv_weights_batch(const mi_pack_job* jobs_dev, int njobs, int total_blocks, int kk_max, mi_stream_t s);

/* ---- depthwise 3x3 convolution ------------------------------------------------
 * the `dconv` of DWConv (backbone/layers/wrappers.py:86-102: BaseConv(C, C, ksize, stride, groups=C); the reference
 * builds it with ksize 3 everywhere, darknetx.py:113-160): NHWC bf16, pad 1, stride 1 or 2, C % 8 == 0.
 * w: the fp32 parameter tensor [C][1][3][3] (rounded to bf16 in the kernel like every conv operand).
 * fwd: y = dwconv(x); stats_acc (may be NULL): fp64 [nslots][rup(C,32)][2] (sum, sumsq) accumulators of the following
 *      BatchNorm, added to atomically exactly as the dense conv's epilogue does (caller zeroes them once per step).
 * dgrad: dx (+)= dwconv^T(dy).  wgrad: dw[C][1][3][3] fp32 = correlation(x, dy), overwritten, deterministic
 *      (block partials in ws, mi_dwconv3x3_wgrad_ws_bytes(C) bytes, summed in a fixed order). */
int mi_dwconv3x3_fwd(const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, int C, int stride,
                     int outH, int outW, double* stats_acc, int nslots, mi_stream_t s);
int mi_dwconv3x3_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx, int N, int H, int W, int C,
                       int stride, int outH, int outW, int accumulate, mi_stream_t s);
int64_t mi_dwconv3x3_wgrad_ws_bytes(int C);
int mi_dwconv3x3_wgrad(const void* x, int ldx, const void* dy, int lddy, int N, int H, int W, int C, int stride,
                       int outH, int outW, float* ws, int64_t ws_bytes, float* dw, mi_stream_t s);

/* ---- BatchNorm(train) + SiLU (+ residual) -----------------------------
 * replaces nn.BatchNorm2d + nn.SiLU of BaseConv (wrappers.py:76-80) and the
 * Bottleneck add (wrappers.py:119-123). */
/* the same BatchNorm pass of several independent layers in ONE launch (FPN levels of the head).
 * kind 0 = mi_bn_act_fwd, 1 = mi_bn_act_bwd_reduce, 2 = mi_bn_act_bwd_apply, 3 = mi_bn_act_bwd_fused; the fields of a job mean what the
 * arguments of those calls mean (acc = stats_acc / dacc).  All jobs share `act`. */
#define MI_BN_MAX_GROUP 8
typedef struct mi_bn_job {
  const void* y; const void* res; void* a;
  const void* da; void* dy; void* dres;
  double* acc;
  const float* gamma; const float* beta; float* rmean; float* rvar; int64_t* nbt;
  float* scale; float* shift; float* mean; float* invstd; float* dgamma; float* dbeta;
  int64_t npix, count;
  int32_t ldy, ldres, lda, ldda, lddy, lddres, dres_accum, C, nslots, nblk, act, pad_;
  float eps, momentum;
  uint32_t* bar; /* kind 3: the layer's barrier words (see mi_bn_act_bwd_fused) */
} mi_bn_job;
typedef struct mi_bn_group {
  int32_t kind, njobs, nblocks, act;
  int64_t starts_off, table_bytes;
} mi_bn_group;
int mi_bn_group_plan(int kind, const mi_bn_job* jobs, int n, void* table_host, int64_t table_cap, mi_bn_group* meta);
int mi_bn_group_run(const mi_bn_group* meta, const void* table_dev, mi_stream_t s);

/* eval mode: scale/shift from running statistics */
int mi_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int C, float* scale, float* shift,
                      mi_stream_t s);
/* a = act(y*scale+shift) (+res); act: 1 silu, 0 identity.
 * train mode (stats_acc != NULL): scale/shift are first derived, in the kernel prologue, from the fp64
 * accumulators the conv epilogue filled ([nslots][C][2], `count` elements per channel); scale/shift/mean/
 * invstd [C] are written for the backward pass and the running statistics are updated exactly as
 * nn.BatchNorm2d does (momentum, unbiased running variance, num_batches_tracked).
 * eval mode (stats_acc == NULL): scale/shift are inputs (mi_bn_eval_affine). */
int mi_bn_act_fwd(const void* y, int ldy, const double* stats_acc, int nslots, int64_t count, const float* gamma,
                  const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                  int64_t* num_batches_tracked, float* scale, float* shift, float* mean, float* invstd,
                  const void* res, int ldres, void* a, int lda, int64_t npix, int C, int act, mi_stream_t s);
/* pass 1: per-channel sums of dz and dz*xhat over pixels, atomically added to dacc[nslots][C][2] (fp64,
 * caller zeroes once per step); nblk = number of blocks to launch */
int mi_bn_act_bwd_reduce(const void* da, int ldda, const void* y, int ldy, const float* scale,
                         const float* shift, const float* mean, const float* invstd,
                         double* dacc, int nslots, int nblk, int64_t npix, int C, int act, mi_stream_t s);
/* pass 2: dy = gamma*invstd*(dz - c1 - xhat*c2), c1/c2 from dacc (prologue); writes dgamma/dbeta (overwrite);
 * optional dres (+)= da */
int mi_bn_act_bwd_apply(const void* da, int ldda, const void* y, int ldy, const float* scale,
                        const float* shift, const float* mean, const float* invstd,
                        const float* gamma, const double* dacc, int nslots, int64_t count, float* dgamma,
                        float* dbeta,
                        void* dy, int lddy, void* dres, int lddres, int dres_accum, int64_t npix, int C, int act,
                        mi_stream_t s);
/* both backward passes in ONE launch: every block stays resident, keeps its share of (da, y) in registers across a
 * grid-wide barrier and writes dy from them - 3 tensor passes instead of 5 (layers beyond the register capacity stream the
 * excess twice, as the two-pass form does).  Same arguments and results as reduce + apply; dacc must be zero on entry.
 * barrier_words: MI_BN_BAR_WORDS x uint32 of device memory (256-byte aligned) owned by this layer, zero before the first
 * use, not shared between launches that may run concurrently ([2] != 0 afterwards means a wait gave up: the launch was
 * not fully resident).
 * Must not run concurrently with another fused launch on the same device (both would wait for blocks that cannot start). */
#define MI_BN_BAR_GROUPS 16
#define MI_BN_BAR_WORDS (64 * (1 + 2 * MI_BN_BAR_GROUPS))
/* resident-block budget of the fused launches: 0 = everything the device runs at once (default); a data-parallel
 * trainer leaves headroom for the collective's kernels.  Returns the budget in effect. */
int mi_bn_fused_set_capacity(int blocks);
int mi_bn_act_bwd_fused(const void* da, int ldda, const void* y, int ldy, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* gamma, double* dacc, int nslots,
                        int64_t count, float* dgamma, float* dbeta, void* dy, int lddy, void* dres, int lddres,
                        int dres_accum, int64_t npix, int C, int act, uint32_t* barrier_words, mi_stream_t s);

/* ---- data movement ops -------------------------------------------------- */
/* Focus space-to-depth (wrappers.py:202-220) fused with fp32 NCHW -> bf16 NHWC and 12->16 ch pad */
int mi_focus_pack(const float* img_nchw, int N, int H, int W, void* out, int ldo, mi_stream_t s);
/* same from the uint8 NCHW image of the data loader (preprocess_image's .type(torch.float), yolox.py:96-99, fused) */
int mi_focus_pack_u8(const uint8_t* img_nchw, int N, int H, int W, void* out, int ldo, mi_stream_t s);
/* nn.Upsample(2,"nearest") (yolo_pafpn.py:28) into a concat slice, and its gradient */
int mi_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C,
                      mi_stream_t s);
int mi_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int accumulate, int N, int H,
                      int W, int C, mi_stream_t s);
/* SPP max-pools k=5,9,13 s1 (wrappers.py:150-153): three outputs + argmax codes; idx: 6*N*H*W*C bytes
 * (separable argmax: 3 planes of vertical codes, 3 of horizontal codes); H*W <= 512 for the backward */
int mi_spp_pool_fwd(const void* x, int ldx, void* y5, void* y9, void* y13, int ldy, uint8_t* idx,
                    int N, int H, int W, int C, mi_stream_t s);
int mi_spp_pool_bwd(const void* dy5, const void* dy9, const void* dy13, int lddy,
                    const uint8_t* idx, void* dx, int lddx, int accumulate, int N, int H, int W,
                    int C, mi_stream_t s);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of detectron2's ResNet stem (BasicStem.forward; d2 upstream), bf16
 * NHWC views, C %% 8 == 0.  outH = (H + 1) / 2.  Backward routes each output gradient to the FIRST maximum of its
 * window in row-major order (ATen's tie rule): dx (+)= sum over the <= 4 windows that contain the pixel. */
int mi_maxpool3x3s2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, mi_stream_t s);
/* the same pooling recording, per output element, the window position (0..8, row-major) of its first maximum in
 * code[N][Ho][Wo][C] (uint8, 8-byte aligned), and the backward that reads the codes instead of re-deriving them from x:
 * identical results (torch's MaxPool2d gradient goes to the first maximum), a twentieth of the time on the stem's map */
int mi_maxpool3x3s2_fwd_idx(const void* x, int ldx, void* y, int ldy, uint8_t* code, int N, int H, int W, int C, mi_stream_t s);
int mi_maxpool3x3s2_bwd_idx(const uint8_t* code, const void* dy, int lddy, void* dx, int lddx, int accumulate, int N, int H,
                            int W, int C, mi_stream_t s);
int mi_maxpool3x3s2_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int accumulate, int N,
                        int H, int W, int C, mi_stream_t s);
/* generic strided bf16 NHWC copy / accumulate (dst (+)= src) */
int mi_copy_bf16(const void* src, int lds_, void* dst, int ldd, int accumulate, int64_t npix, int C,
                 mi_stream_t s);
/* column sums of a bf16 NHWC view -> fp32 [C] (bias gradient of the prediction convs) */
int mi_colsum_bf16(const void* x, int ldx, int64_t npix, int C, float* out, int accumulate, float* ws,
                   mi_stream_t s);
/* the same for any channel count (a multiple of 8 readable) in one launch pair; ws: mi_colsum_wide_ws_bytes(C) bytes.
 * Bias gradients of nn.Linear in the transformer (backbone/detr_backbone.py:140-230: 256 .. 2048 output channels). */
int64_t mi_colsum_wide_ws_bytes(int C);
int mi_colsum_bf16_wide(const void* x, int ldx, int64_t npix, int C, float* out, int accumulate, float* ws, mi_stream_t s);
/* the same over the SQUARES of the values (fp32 products): column norms^2 */
int mi_colsumsq_bf16_wide(const void* x, int ldx, int64_t npix, int C, float* out, int accumulate, float* ws, mi_stream_t s); /* ws: >= 128*128 floats of scratch (two-stage, fixed summation order) */

/* ---- YOLOX head: decode + SimOTA + losses ---------------------------------
 * replaces YOLOXHead.get_output_and_grid / get_losses / get_assignments /
 * get_in_boxes_info / dynamic_k_matching (head/yolox_head.py:226-669),
 * bboxes_iou and IOUloss (utils/boxes.py:57-81,125-168). */
typedef struct mi_yolox_loss_desc {
  const float* preds;   /* [B][A][5+ncls] raw head outputs (xy,wh undecoded; logits)        */
  const float* labels;  /* [B][max_labels][5] (cls,cx,cy,w,h), zero rows = padding          */
  const float* anchors; /* [A][3] (grid_x, grid_y, stride)                                   */
  int32_t B, A, ncls, max_labels;
  int32_t gmax;         /* upper bound on valid labels per image (<= max_labels)             */
  /* workspaces (caller-owned) */
  float* cost;          /* [B][gmax][A]                                                       */
  float* iou;           /* [B][gmax][A]                                                       */
  uint8_t* match;       /* unused since round 2 (the match matrix became a per-anchor counter); may be NULL       */
  int32_t* ngt;         /* [B]                                                                */
  /* assignment outputs */
  uint8_t* fg;          /* [B][A] final foreground mask                                       */
  int32_t* matched_gt;  /* [B][A] index of matched gt or -1                                   */
  float* matched_iou;   /* [B][A]                                                             */
  float* partial;       /* [nblk][4] block partial sums (iou, obj, cls, nfg); nblk = B*ceil(A/256) */
  float* out;           /* [8]: total, 5*iou, obj, cls, l1, num_fg/num_gt, num_fg, num_gt */
  /* head.use_l1 (yolox_head.py:131, switched on by the meta-arch after INPUT.MOSAIC_AND_MIXUP.DISABLE_AT_ITER,
   * meta_arch/yolox.py:105-118): adds sum|raw_reg - get_l1_target| / num_fg to the total; 0 = the default */
  int32_t use_l1, rsv_;
  float* partial_l1;    /* [nblk] (use_l1 only) */
  /* the YOLOv6 head's form of this loss (ComputeLoss, head/yolov6_head.py:315-754) differs in constants only; a zero
   * selects the YOLOX value: centre radius 2.5, SimOTA cost weights cls 1 / iou 3, box-loss weight 5 */
  float center_radius, cls_weight, iou_weight, reg_weight;
  int32_t iou_type;     /* box loss: 0 IOUloss "iou" (1 - iou^2); 1..4 IOUlossV6 giou / diou / ciou / siou, eps 1e-7 */
  int32_t rsv2_;
} mi_yolox_loss_desc;
int mi_yolox_loss_fwd(const mi_yolox_loss_desc* d, mi_stream_t s);
/* gradient wrt raw preds; gw[4] = upstream grads of (total, 5*iou, obj, cls) on device (+ gw[4] = of l1 iff use_l1).
 * dpreds fp32 [B][A][5+ncls] and/or bf16 per-level NHWC maps (see plan builder). */
int mi_yolox_loss_bwd(const mi_yolox_loss_desc* d, const float* gw, float* dpreds, mi_stream_t s);
/* extract channels [c0, c0+nc) of dpreds fp32 [B][A][nch] for anchors a0..a0+HW into a bf16
 * NHWC map dst [B][HW][ld] (channels >= nc zero-filled): the out-gradient of one prediction conv. */
int mi_yolox_split_dpreds(const float* dpreds, int B, int A, int nch, int a0, int HW, int c0, int nc,
                          void* dst, int ld, mi_stream_t s);
/* bias gradients of all prediction convs (nn.Conv2d bias of yolox_head.py:103-129) in two launches:
 * out[c] = sum_b sum_{a0 <= a < a0+HW} dpreds[b][a][c0 + c]; jobs is a HOST array (<= 16);
 * ws: 16*512*128 floats of scratch = 16 slabs; a call uses (distinct (a0, HW) levels) x ceil(nch / 128) of them
 * (3 levels: nch <= 640, i.e. up to 635 classes) */
typedef struct mi_bias_job {
  float* out;
  int32_t a0, HW, c0, nc;
} mi_bias_job;
int mi_yolox_bias_grads(const float* dpreds, int B, int A, int nch, const mi_bias_job* jobs, int njobs, float* ws,
                        mi_stream_t s);
/* the out-gradient maps of ALL prediction convs in one launch (the per-conv form is mi_yolox_split_dpreds):
 * dst[b][pix][0..ld) (bf16, ld % 8 == 0) = dpreds[b][a0 + pix][c0 .. c0 + nc), pad channels zero. */
typedef struct mi_split_job {
  void* dst;
  int32_t a0, HW, c0, nc, ld, rsv_;
} mi_split_job;
int mi_yolox_split_dpreds_batch(const float* dpreds, int B, int A, int nch, const mi_split_job* jobs, int njobs,
                                mi_stream_t s);
/* mi_yolox_loss_bwd + mi_yolox_split_dpreds_batch + mi_yolox_bias_grads in ONE pass over the gradient: every value is
 * computed once and written as the bf16 out-gradient map of its prediction conv, into per-block column sums and - only when
 * dpreds is not NULL (tests / diagnostics) - as fp32 dpreds [B][A][5 + ncls]; per-block column sums (bias gradients; a second small launch adds the blocks).  The split jobs must tile [A][5 + ncls]
 * exactly (each job = the box columns, the objectness column or class columns of one level); every bias job must name the
 * (a0, HW, c0, nc) of one split job.  ws: scratch of ws_floats floats (>= njobs * max ld; 16 * 512 * 128 serves all plans). */
int mi_yolox_loss_bwd_fused(const mi_yolox_loss_desc* d, const float* gw, float* dpreds, const mi_split_job* split_jobs,
                            int nsplit, const mi_bias_job* bias_jobs, int nbias, float* ws, int64_t ws_floats, mi_stream_t s);
/* eval decode (yolox_head.py:247-272): in-place on preds; obj/cls sigmoid applied */
int mi_yolox_decode(float* preds, const float* anchors, int B, int A, int ncls, mi_stream_t s);
/* the ONNX-export layout of decode_outputs (yolox_head.py:263-269) from the decoded predictions:
 * out [B][A][6+ncls] = (xy, wh, conf, argmax(prob) as float - first maximum, prob) */
int mi_yolox_onnx_layout(const float* decoded, float* out, int B, int A, int ncls, mi_stream_t s);

/* ---- DETR set matching (config 4) --------------------------------------------------
 * replaces HungarianMatcher.forward (utils/detr_utils.py:37-91): matching cost
 * C = w_bbox*L1(cdist) + w_class*(-softmax(logits)[label]) + w_giou*(-GIoU) per image, then
 * scipy.optimize.linear_sum_assignment (detr_utils.py:89) - which the reference runs on the CPU.
 * logits [B][Q][NC], boxes [B][Q][4] cxcywh, tgt_labels int64 [T], tgt_boxes [T][4] cxcywh,
 * tgt_off int32 [B+1] (targets of image b = [tgt_off[b], tgt_off[b+1])), gmax >= max targets per image.
 * cost (workspace / output) fp32 [B][Q][gmax]; match_q / match_t int64 [B][gmax]: the first nmatch[b] =
 * min(Q, G_b) entries are the (query, target) pairs sorted by query index, exactly what scipy returns.
 * Q, gmax <= 1024. */
int mi_hungarian_match(const float* logits, const float* boxes, const int64_t* tgt_labels,
                       const float* tgt_boxes, const int32_t* tgt_off, int B, int Q, int NC, int gmax,
                       float w_class, float w_bbox, float w_giou, float* cost, int64_t* match_q,
                       int64_t* match_t, int32_t* nmatch, mi_stream_t s);
/* the assignment alone on a caller-provided cost matrix (same layout) */
int mi_lsap(const float* cost, const int32_t* tgt_off, int B, int Q, int gmax, int64_t* match_q,
            int64_t* match_t, int32_t* nmatch, mi_stream_t s);

/* DETR set-prediction losses for the match indices above, forward + backward.
 * replaces SetCriterion.loss_labels / loss_cardinality / loss_boxes (modeling/meta_arch/detr.py:504-556) and their
 * autograd backward.  NC = num_classes + 1 (last class = no-object, weight eos_coef); num_boxes as detr.py:615-619
 * computes it (host).  losses[8] = loss_ce, class_error, cardinality_error, loss_bbox, loss_giou, sum of CE weights,
 * matched pairs, 0.  rowstate: caller workspace of B*Q*16 floats, written by fwd and read by bwd.
 * bwd: gw[3] on device = upstream gradients of (loss_ce, loss_bbox, loss_giou); dlogits [B][Q][NC], dboxes [B][Q][4].
 * B <= 256. */
typedef struct mi_detr_loss_desc {
  const float* logits;
  const float* boxes;
  const int64_t* tgt_labels;
  const float* tgt_boxes;
  const int32_t* tgt_off;
  const int64_t* match_q;
  const int64_t* match_t;
  const int32_t* nmatch;
  int32_t B, Q, NC, gmax;
  float eos_coef, num_boxes;
  float* losses;
  float* rowstate;
} mi_detr_loss_desc;
int mi_detr_set_loss_fwd(const mi_detr_loss_desc* d, mi_stream_t s);
int mi_detr_set_loss_bwd(const mi_detr_loss_desc* d, const float* gw, float* dlogits, float* dboxes, mi_stream_t s);

/* ---- multi-head attention core (DETR, config 4) -------------------------------------
 * O = softmax(scale * Q K^T + key_padding_mask) V per (batch, head); head_dim = 32 (DETR: 256 = 8 x 32).
 * replaces the attention inside nn.MultiheadAttention as called by the transformer layers
 * (modeling/backbone/detr_backbone.py:140,155-157,200-202,222-230).  Tensors are bf16 [L][B][E] (sequence first,
 * E = H*32: the layout nn.MultiheadAttention's in-projection produces; = NHWC "pixel l*B+b, channel h*32+d", so
 * the projections are 1x1 convolutions of this library).  key_padding_mask uint8 [B][Lk] (1 = ignore) or NULL;
 * lse fp32 [B][H][Lq] is saved for the backward; delta_ws: fp32 [B][H][Lq] scratch.  No attention-weight dropout. */
int mi_mha_fwd(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, void* o, float* lse,
               int B, int H, int Lq, int Lk, int E, float scale, mi_stream_t s);
int mi_mha_bwd(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, const void* o,
               const float* lse, const void* dout, float* delta_ws, void* dq, void* dk, void* dv, int B, int H,
               int Lq, int Lk, int E, float scale, mi_stream_t s);
/* the same with attention-weight dropout (nn.MultiheadAttention(dropout=p) in training mode, detr_backbone.py:140):
 * survivors scaled by 1/(1-p); the keep mask is a pure function of (seed, ((b*H+h)*Lq+q)*Lk+key), recomputed by the
 * backward - pass the SAME p and seed to both.  mi_mha_dropout_mask writes that mask (uint8 [B][H][Lq][Lk]) for tests. */
int mi_mha_fwd_dropout(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, void* o, float* lse,
                       int B, int H, int Lq, int Lk, int E, float scale, float drop_p, uint64_t seed, mi_stream_t s);
int mi_mha_bwd_dropout(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, const void* o,
                       const float* lse, const void* dout, float* delta_ws, void* dq, void* dk, void* dv, int B, int H,
                       int Lq, int Lk, int E, float scale, float drop_p, uint64_t seed, mi_stream_t s);
int mi_mha_dropout_mask(uint8_t* out, int B, int H, int Lq, int Lk, float drop_p, uint64_t seed, mi_stream_t s);
/* the same again with an fp32 copy of the output, o_f32 [Lq][B][E] (NULL = the forms above): the forward writes it, the
 * backward takes delta = rowsum(dO o O) from it.  dS = P o (dP - delta) cancels to a few percent of its operands whenever
 * the values of a row's keys are alike (every attention of a freshly initialised DETR); the 2^-9 rounding of a bf16 O is
 * coherent over the row and leaves that difference as a 30 - 80 % error of dq (measured on device operands, tools/
 * attn_bwd_error.py: dq rel 0.78 -> 0.0014).  4 bytes per output element, read once by the backward. */
int mi_mha_fwd_dropout_o32(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, void* o, float* o_f32,
                           float* lse, int B, int H, int Lq, int Lk, int E, float scale, float drop_p, uint64_t seed,
                           mi_stream_t s);
int mi_mha_bwd_dropout_o32(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, const void* o,
                           const float* o_f32, const float* lse, const void* dout, float* delta_ws, void* dq, void* dk,
                           void* dv, int B, int H, int Lq, int Lk, int E, float scale, float drop_p, uint64_t seed,
                           mi_stream_t s);
/* and with row strides for q / dq (ldq) and k / dk (ldk), in elements, >= E: a self-attention whose query and key are the
 * same tensor (the encoder layers and the decoder's self-attention, detr_backbone.py:155-157,222-224: q = k = x + pos) takes
 * q and k from ONE [T, 2E] projection (k = q + E, ldq = ldk = 2E) and hands dq | dk back the same way - one projection, one
 * data-gradient and one weight-gradient launch instead of two each.  v, o, dout, dv, o_f32 keep row stride E. */
int mi_mha_fwd_dropout_ld(const void* q, int ldq, const void* k, int ldk, const void* v, const uint8_t* key_padding_mask,
                          void* o, float* o_f32, float* lse, int B, int H, int Lq, int Lk, int E, float scale, float drop_p,
                          uint64_t seed, mi_stream_t s);
int mi_mha_bwd_dropout_ld(const void* q, int ldq, const void* k, int ldk, const void* v, const uint8_t* key_padding_mask,
                          const void* o, const float* o_f32, const float* lse, const void* dout, float* delta_ws, void* dq,
                          void* dk, void* dv, int B, int H, int Lq, int Lk, int E, float scale, float drop_p, uint64_t seed,
                          mi_stream_t s);
/* elementwise dropout of a bf16 tensor (F.dropout of detr_backbone.py:147-150,163-167,...): out[i] = keep(seed, i) ?
 * x[i] / (1-p) : 0; applying it with the same (p, seed) to the output gradient IS the backward. n %% 8 == 0. */
int mi_dropout_bf16(const void* x, void* out, int64_t n, float drop_p, uint64_t seed, mi_stream_t s);
/* out = res + dropout(x): the residual add that follows every F.dropout of the transformer layers
 * (detr_backbone.py:163,167,235,239,243: `src = src + self.dropout1(src2)`) in the same pass; the dropped value is rounded
 * to bf16 before the fp32 add, so the result equals mi_dropout_bf16 followed by mi_ew_bf16(op 0) bit for bit.  res NULL:
 * mi_dropout_bf16. */
int mi_dropout_add_bf16(const void* x, const void* res, void* out, int64_t n, float drop_p, uint64_t seed, mi_stream_t s);
/* The post-norm residual of the transformer layers in one pass (detr_backbone.py:163-168,235-243:
 * `src = self.norm1(src + self.dropout1(src2))`): sum_out = bf16(res + dropout(x)) - the bits mi_dropout_add_bf16 writes -
 * and y / mean / rstd = mi_layernorm_fwd(sum_out).  mi_layernorm_bwd_dropout is mi_layernorm_bwd that ALSO writes
 * dx_drop = dropout(dx) with the forward's (p, seed): the gradient of the dropped branch (dx itself is the residual's).
 * dx_drop NULL: mi_layernorm_bwd. */
int mi_dropout_add_layernorm_fwd(const void* x, const void* res, void* sum_out, const float* gamma, const float* beta, void* y,
                                 float* mean, float* rstd, int T, int E, float eps, float drop_p, uint64_t seed, mi_stream_t s);
int mi_layernorm_bwd_dropout(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                             void* dx_drop, float* dgamma, float* dbeta, float* ws, int T, int E, float drop_p, uint64_t seed,
                             mi_stream_t s);
/* A step captured as a hipGraph bakes its seeds into the kernel arguments.  With a device word registered here every
 * dropout kernel launched afterwards (mi_dropout_bf16, mi_mha_*_dropout, forward and backward alike) uses seed + *dev_word,
 * read at RUN time: advance the word once per replay (after the backward) and every replay draws fresh masks while the
 * backward of a step still recomputes its own forward's.  NULL (default) switches it off.  Process-global, not per stream. */
int mi_dropout_seed_offset(const uint64_t* dev_word);

/* ---- row-wise ops of DETR's transformer layers (detr_backbone.py:135-278) -------------------
 * nn.LayerNorm(E) forward / backward over bf16 [T][E] token rows (fp32 gamma/beta/mean/rstd), eps 1e-5;
 * E %% 64 == 0, E <= 1024; ws (backward): fp32 [ceil(T/16)][E][2].  mi_ew_bf16: op 0 out = a + b (residual),
 * op 1 out = relu(a), op 2 out = a * (b > 0) (ReLU backward: a = dy, b = forward output), op 3 out = sigmoid(a),
 * op 4 out = a * b * (1 - b) (sigmoid backward: a = dy, b = forward output); op 5 / 6 swish and its backward; op 7 out =
 * relu(a + b) (the tail of a ResNet bottleneck, bit-identical to op 0 followed by op 1); n %% 8 == 0. */
int mi_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int T,
                     int E, float eps, mi_stream_t s);
int mi_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                     float* dgamma, float* dbeta, float* ws, int T, int E, mi_stream_t s);
int mi_ew_bf16(const void* a, const void* b, void* out, int64_t n, int op, mi_stream_t s);

/* ---- IoU-family regression losses (CIoU / DIoU / GIoU / SIoU / IoU) with gradient ------
 * replaces IOUlossV6.__call__ + its autograd backward (utils/boxes.py:666-752; YOLOv6 head, yolov6_head.py:346,512):
 * loss[n] = 1 - iou_variant(pred[n], target[n]) (reduction "none"), dpred[n][4] = dloss[n] * d loss / d pred
 * (dloss NULL => ones; CIoU's alpha is a constant for the gradient, as under torch.no_grad()).
 * pred/target fp32 [n][4] in (cx,cy,w,h) (box_xyxy 0) or (x1,y1,x2,y2); iou_type: 0 iou, 1 giou, 2 diou, 3 ciou, 4 siou */
int mi_iou_loss_v6(const float* pred, const float* target, int n, int iou_type, int box_xyxy, float eps,
                   const float* dloss, float* loss, float* dpred, mi_stream_t s);

/* IOUloss of the YOLOX head as a standalone op (utils/boxes.py:125-168): boxes (cx,cy,w,h) fp32 [n][4];
 * loss_type 0 "iou" (1 - iou^2), 1 "giou" (1 - clamp(giou, -1, 1)); loss[n] (reduction "none") and
 * dpred[n][4] = dloss[n] * d loss / d pred (dloss NULL => ones).  Either output may be NULL. */
int mi_yolox_iou_loss(const float* pred, const float* target, int n, int loss_type, const float* dloss, float* loss,
                      float* dpred, mi_stream_t s);
/* pairwise IoU matrix out[N][M]: bboxes_iou (utils/boxes.py:57-81) and pairwise_bbox_iou (utils/boxes.py:755-779);
 * boxes (cx,cy,w,h) (box_xyxy 0) or (x1,y1,x2,y2) */
int mi_pairwise_bbox_iou(const float* box1, const float* box2, int N, int M, int box_xyxy, float* out, mi_stream_t s);

/* ---- BiFPN neck (neck/bifpn.py:184-395, build_resnet_bifpn_backbone) -------------------
 * GroupNorm = detectron2 get_norm("GN") = nn.GroupNorm(32, C) (MODEL.BIFPN.NORM default): NHWC bf16, x [N][HW][ldx],
 * C % 8 == 0, C % G == 0 (C / G need not be a multiple of 8), eps as nn.GroupNorm (1e-5).  mean_rstd [N][G][2] is written
 * by fwd and read by bwd; ws: mi_groupnorm_ws_bytes(N, C) bytes of scratch (zeroed inside).  bwd overwrites dgamma / dbeta. */
int64_t mi_groupnorm_ws_bytes(int N, int C);
int mi_groupnorm_fwd(const void* x, int ldx, int N, int HW, int C, int G, const float* gamma, const float* beta, float eps,
                     void* y, int ldy, float* mean_rstd, double* ws, mi_stream_t s);
int mi_groupnorm_bwd(const void* dy, int lddy, const void* x, int ldx, int N, int HW, int C, int G, const float* gamma,
                     const float* mean_rstd, void* dx, int lddx, float* dgamma, float* dbeta, double* ws, mi_stream_t s);
/* nn.MaxPool2d(2, 2) of ResampleFeatureMap (bifpn.py:151-155), floor mode; bwd recomputes the arg-max from x (first
 * maximum in row-major window order); with odd H / W the caller zero-fills dx first */
int mi_maxpool2x2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, mi_stream_t s);
int mi_maxpool2x2_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C,
                      mi_stream_t s);
/* FpnCombine "fastattn" (bifpn.py:221-236): out = sum_i x_i * relu(w_i) / (sum relu(w) + 1e-4) over nin = 2 or 3
 * contiguous bf16 tensors of n elements (n % 8 == 0); bwd: dx_i (dxs[i] may be NULL) and d edge_weights [nin]
 * (deterministic: per-block partials in ws, mi_fastattn_ws_bytes() bytes) */
int64_t mi_fastattn_ws_bytes(void);
int mi_fastattn_fwd(const void* const* xs, int nin, const float* edge_weights, void* out, int64_t n, mi_stream_t s);
int mi_fastattn_bwd(const void* const* xs, int nin, const float* edge_weights, const void* g, void* const* dxs, float* dedge,
                    float* ws, int64_t n, mi_stream_t s);

/* ---- input pipeline: mosaic + random_perspective + pad on the GPU ---------------
 * replaces, for a whole batch, the cv2.resize / canvas paste of MyDatasetMapper2.__call__ (data/dataset_mapper.py:523-598),
 * the cv2.warpAffine of random_perspective (data/transforms/data_augment.py:67-75) and the pad-to-batch with 114 of
 * YOLOX.preprocess_image (meta_arch/yolox.py:95-130).  uint8, OpenCV's fixed-point INTER_LINEAR arithmetic.
 * paste job: source image src [h0][w0][3] (HWC) resized to rw x rh; canvas[y1a:y2a, x1a:x2a] = resized[y1b.., x1b..]
 *            (canvas row pitch cw pixels, pre-filled with 114 by the caller).
 * warp job:  out[c][y][x] (planes of Hp x Wp bytes, y < h, x < w) = INTER_LINEAR sample of the ch x cw canvas at
 *            minv * (x, y, 1)  (minv: the INVERTED 2x3 matrix, row-major; taps outside the canvas read `border`).
 * mi_mosaic_jobs_layout validates both (host) tables and fills blk0; returns max(blocks of the paste launch, blocks of the
 * warp launch); the launches take the device copies of the tables and their own block counts. */
typedef struct mi_mosaic_paste_job {
  const void* src;
  void* canvas;
  int32_t h0, w0, rh, rw, cw, x1a, y1a, x2a, y2a, x1b, y1b, blk0;
  int32_t fsrc, pad_;        /* fsrc: the loaded image is float32 in the reference (YOLOFRandomDistortion ran): cv2.resize takes
                                its FLOAT path (float32 coefficients, products and sums) and the assignment into the uint8
                                canvas truncates (dataset_mapper.py:531-567) */
} mi_mosaic_paste_job;
typedef struct mi_warp_job {
  const void* canvas;
  void* out;
  double minv[6];
  int32_t ch, cw, h, w, Hp, Wp, border, blk0;
} mi_warp_job;
int mi_mosaic_jobs_layout(mi_mosaic_paste_job* paste_host, int npaste, mi_warp_job* warp_host, int nwarp);
int mi_mosaic_paste(const mi_mosaic_paste_job* jobs_dev, int njobs, int total_blocks, mi_stream_t s);
int mi_warp_affine_u8(const mi_warp_job* jobs_dev, int njobs, int total_blocks, mi_stream_t s);
/* MyDatasetMapper2.mixup (data/dataset_mapper.py:686-768) in place on a warped sample: out[c][y][x] (y < th, x < tw) =
 * uint8(0.5 * out + 0.5 * partner), partner = the pool image src [h0][w0][3] resized to rw1 x rh1 (8-bit fixed point) into
 * the corner of a 114.0 canvas dh x dw, that canvas resized to ow x oh (float64 path), optionally mirrored, zero-padded to the
 * target, cropped at (x_off, y_off), truncated to uint8.  mi_mixup_jobs_layout fills blk0 and returns the block count. */
typedef struct mi_mixup_job {
  const void* src;
  void* out;
  int32_t h0, w0, rh1, rw1, dh, dw, oh, ow, flip, x_off, y_off, th, tw, Hp, Wp, blk0;
  int32_t fsrc, pad_;        /* fsrc: the pool image is float32 in the reference: the FIRST resize takes cv2's float path and is
                                not rounded (dataset_mapper.py:706-711) */
} mi_mixup_job;
int mi_mixup_jobs_layout(mi_mixup_job* jobs_host, int njobs);
int mi_mixup_blend(const mi_mixup_job* jobs_dev, int njobs, int total_blocks, mi_stream_t s);

/* ---- the detectron2 T.* front of the reference's input pipeline --------------------------------------------------
 * build_normal_augmentation (yolov7/data/detection_utils.py:37-86) as MyDatasetMapper2._load_image_with_annos applies it
 * to every image it loads (yolov7/data/dataset_mapper.py:642-683): T.ResizeShortestEdge = PIL.Image.resize(BILINEAR) of
 * the uint8 image (Pillow libImaging/Resample.c: fp64 coefficients -> 22-bit fixed point, horizontal pass rounded to 8
 * bits, then the vertical pass), T.RandomFlip horizontal / vertical, YOLOFRandomShift (data/transforms/transform.py:341-388:
 * zeros where the shifted image does not reach).  One job per image: src HWC uint8 [h0][w0][3] (rows src_ld bytes apart, so a
 * T.RandomCrop window is an offset pointer; optionally mirrored first, as DetrDatasetMapper's flip-then-resize order needs,
 * data/dataset_mapper.py:777-800) -> nh x nw, element
 * (c, y, x) at dst + c dsc + y dsy + x dsx bytes (HWC: 1, 3 nw, 3; a sample of a padded NCHW batch: Hp Wp, Wp, 1);
 * tmp = [h0][nw][3] scratch of the horizontal pass (unused when nw == w0).  RandomSaturation / RandomBrightness (detectron2
 * BlendTransform, numpy's fp64 / fp32 arithmetic) and YOLOFRandomDistortion (OpenCV's 8-bit RGB2HSV_b / HSV2RGB_b, restated:
 * no cv2 here to pin them) are applied per pixel between the flips and the shift.
 * mi_pil_resize_jobs_layout validates the (host) table, fills blk0h / blk0v and returns the block counts of the two flat
 * launches; the launches take the device copy of the table.  Down-scaling factors up to 8. */
typedef struct mi_pil_resize_job {
  const void* src;
  void* tmp;
  void* dst;
  int64_t dsc, dsy, dsx;
  int64_t src_ld;            /* bytes between source rows (3 w0 for a whole image; the parent's for a crop window) */
  double sat_src;            /* RandomSaturation: 1 - w (fp64, the grey image's weight) */
  int32_t h0, w0, nh, nw;
  int32_t hflip, vflip, shift_x, shift_y;
  int32_t src_hflip;         /* mirror the source BEFORE the resampling (DetrDatasetMapper: T.RandomFlip, then the resizes) */
  int32_t color;             /* bit 0 RandomSaturation, bit 1 RandomBrightness (d2 BlendTransform on the uint8 image), bit 2
                                YOLOFRandomDistortion (data/transforms/transform.py:250-308: cv2's 8-bit RGB <-> HSV around
                                three float32 scalings); in that order, after the flips */
  float sat_dst, bri_dst;    /* w as float32 */
  float dis_hue, dis_sat, dis_exp;   /* float32(dhue * 179 / 255.), float32(dsat), float32(dexp) */
  int32_t dis_pos;           /* dhue > 0 */
  int32_t blk0h, blk0v;
} mi_pil_resize_job;
int mi_pil_resize_jobs_layout(mi_pil_resize_job* jobs_host, int njobs, int32_t* blocks_h, int32_t* blocks_v);
int mi_pil_resize_h(const mi_pil_resize_job* jobs_dev, int njobs, int total_blocks, mi_stream_t s);
int mi_pil_resize_v(const mi_pil_resize_job* jobs_dev, int njobs, int total_blocks, mi_stream_t s);

/* ---- image decoding of the input pipeline: baseline JPEG ---------------------------------------------------------
 * What detectron2's utils.read_image(file, format="BGR") does for MyDatasetMapper2._load_image_with_annos
 * (yolov7/data/dataset_mapper.py:646-648; d2 un-vendored): PIL.Image.open -> EXIF orientation -> convert("RGB") -> BGR,
 * i.e. Pillow's libjpeg(-turbo) defaults - sequential Huffman decoding (jdhuff.c), JDCT_ISLOW (jidctint.c), fancy
 * up-sampling (jdsample.c), YCbCr -> RGB (jdcolor.c).  Served: SOF0 / SOF1 sequential and SOF2 progressive Huffman files
 * (jdphuff.c; any scan script, tables redefined between scans), 8 bit, grey or three components, sampling factors 1 or 2
 * (4:4:4, 4:2:2, 4:2:0, 4:4:0), restart intervals, Adobe transform 0, EXIF orientations 1-8.  Refused with MI_EINVAL:
 * lossless / hierarchical / arithmetic-coded / 12-bit / CMYK files.
 * HOST functions (no GPU needed): mi_jpeg_parse reads the markers up to the first scan; mi_jpeg_huffman decodes every scan's
 * entropy-coded data into int16 coefficient blocks (natural order, not de-quantised; info->coef_count values, component c's blocks_h[c] x
 * blocks_w[c] blocks of 64 from coef_off[c]) - the caller copies them to the device; mi_jpeg_job_fill builds one image's
 * device job (planes: sum over components of blocks * 64 bytes of scratch; out: HWC uint8 [h][w][3], h and w swapped for
 * orientations 5-8 when apply_orientation); mi_jpeg_jobs_layout lays a batch out.  DEVICE: mi_jpeg_idct (one thread per
 * 8x8 block), then mi_jpeg_color (one thread per output pixel), over the device copy of the job table. */
typedef struct mi_jpeg_info {
  int32_t width, height, ncomp, restart_interval, orientation, adobe_transform;   /* adobe_transform -1: no Adobe marker */
  int32_t progressive, pad0_;
  int32_t comp_id[3], hs[3], vs[3], tq[3];
  int32_t hmax, vmax, mcu_w, mcu_h;
  int32_t blocks_w[3], blocks_h[3];
  int64_t coef_off[3], coef_count, sos_pos;   /* sos_pos: offset of the first SOS segment's length field */
  uint16_t qt[4][64];                  /* natural (row-major) order */
  uint8_t dc_bits[4][17], dc_vals[4][256], ac_bits[4][17], ac_vals[4][256];     /* the tables defined before the first scan */
  uint8_t have_qt[4], have_dc[4], have_ac[4], pad_[4];
} mi_jpeg_info;
typedef struct mi_jpeg_job {
  const int16_t* coef;
  void* planes;
  void* out;
  int64_t coef_off[3], plane_off[3];
  int32_t width, height, ncomp, orientation, bgr, ycc;
  int32_t hs[3], vs[3], blocks_w[3], blocks_h[3];
  int32_t hmax, vmax;
  int32_t blk0_idct, blk0_pix;
  uint16_t qt[3][64];
} mi_jpeg_job;
int mi_jpeg_parse(const uint8_t* data, int64_t len, mi_jpeg_info* info);
int mi_jpeg_huffman(const uint8_t* data, int64_t len, const mi_jpeg_info* info, int16_t* coef_host);
/* n files on `threads` host threads of the library; rcs[k] = file k's return code, the call returns the first non-zero one */
int mi_jpeg_huffman_batch(const uint8_t* const* datas, const int64_t* lens, const mi_jpeg_info* infos, int16_t* const* coefs_host,
                          int n, int threads, int32_t* rcs);
int mi_jpeg_job_fill(const mi_jpeg_info* info, const void* coef_dev, void* planes_dev, void* out_dev, int bgr,
                     int apply_orientation, mi_jpeg_job* job);
int mi_jpeg_jobs_layout(mi_jpeg_job* jobs_host, int njobs, int32_t* blocks_idct, int32_t* blocks_pix);
int mi_jpeg_idct(const mi_jpeg_job* jobs_dev, int njobs, int total_blocks, mi_stream_t s);
int mi_jpeg_color(const mi_jpeg_job* jobs_dev, int njobs, int total_blocks, mi_stream_t s);

/* ---- COCO run-length encoding of masks (evaluation output format) ----------------------
 * what pycocotools.mask.encode does for instances_to_coco_json (evaluation/coco_evaluation.py:38-50; the algorithm is
 * cocoapi's maskApi.c rleEncode / rleToString, un-vendored): masks uint8 [n][H][W] (device, non-zero = foreground) ->
 * counts[n][max_runs] = lengths of the alternating runs of the COLUMN-major scan, starting with a (possibly empty) run of
 * zeros; nruns[n] = number of runs, or -(needed runs) when max_runs is too small (that mask's counts are undefined).
 * mi_rle_to_string is host code: counts -> the compact ASCII "counts" string; returns its length (NUL-terminated). */
int mi_rle_encode(const uint8_t* masks, int n, int H, int W, int max_runs, uint32_t* counts, int32_t* nruns, mi_stream_t s);
int mi_rle_to_string(const uint32_t* counts, int nruns, char* out, int out_cap);

/* ---- batched NMS -------------------------------------------------------------
 * replaces torchvision.ops.batched_nms as called by postprocess (utils/boxes.py:199).
 * boxes xyxy fp32 [n][4], scores [n], idxs (class id as float, as the reference passes) [n].
 * keep[n] receives the kept indices in descending score order, *n_keep their count.
 * workspaces: order int32[n], mask uint64[n*ceil(n/64)], sboxes float[5*n+1]. */
int mi_batched_nms(const float* boxes, const float* scores, const float* idxs, int n, float iou_thr,
                   int32_t* order, uint64_t* mask, float* sboxes, int64_t* keep, int32_t* n_keep,
                   mi_stream_t s);
/* the same with the IoU arithmetic chosen by the caller: -1 torchvision's choice by size, 0 class by class on the raw
 * coordinates, 1 the per-class coordinate offsets.  0 is also what batched_clusternms (meta_arch/utils.py:66-95) computes:
 * its fixed-point iteration over the IoU matrix converges to the greedy result. */
int mi_batched_nms_ex(const float* boxes, const float* scores, const float* idxs, int n, float iou_thr, int arithmetic,
                      int32_t* order, uint64_t* mask, float* sboxes, int64_t* keep, int32_t* n_keep, mi_stream_t s);
/* class-aware Soft-NMS (batched_softnms, meta_arch/utils.py:33-63): scores are rescaled IN PLACE, keep[] receives the
 * indices with score > score_threshold in descending score order (ties: ascending index).  sigma is the reference's
 * iou_threshold argument; linear 0 = "gaussian" exp(-iou^2 / sigma), 1 = "linear".  n <= 16384. */
int mi_batched_softnms(const float* boxes, float* scores, const float* idxs, int n, float sigma, float score_threshold,
                       int linear, int64_t* keep, int32_t* n_keep, mi_stream_t s);
/* Matrix NMS (utils/solov2_utils.py:160-206) from the mask-intersection matrix inter[n][n] = masks @ masks^T
 * (candidates in descending score order), sum_masks[n], labels[n] (as float): out_scores[n] = scores * decay.
 * comp_ws: n floats of scratch. */
/* greedy mask NMS (utils/solov2_utils.py:209-236) on the same intersection matrix: keep[n] (1 = kept) */
int mi_mask_nms(const float* inter, const float* sum_masks, const float* labels, int n, float nms_thr, uint8_t* keep,
                mi_stream_t s);
int mi_matrix_nms(const float* inter, const float* sum_masks, const float* labels, const float* scores, int n, float sigma,
                  int linear, float* comp_ws, float* out_scores, mi_stream_t s);

/* ---- fused SGD(momentum, weight-decay) over a flat parameter arena -----------
 * replaces torch.optim.SGD.step as built by detectron2's build_optimizer
 * (train_det.py:73 -> DefaultTrainer). seg table: per segment (offset,count,wd,lr) */
typedef struct mi_sgd_seg {
  int64_t offset, count;
  float weight_decay, lr;
} mi_sgd_seg;
int mi_sgd_momentum_step(float* params, const float* grads, float* momentum_buf,
                         const mi_sgd_seg* segs_dev, int nseg, float momentum, float grad_scale,
                         int first_step, mi_stream_t s);

/* fp32 sigmoid of the DETR box head (meta_arch/detr.py:452): forward (x -> y; dy = dx = NULL) or backward
 * (dx = dy * y * (1 - y); x may be NULL) */
int mi_sigmoid_f32(const float* x, const float* dy, float* y, float* dx, int64_t n, mi_stream_t s);
/* fused AdamW (decoupled weight decay, bias-corrected moments) over the flat arena: replaces torch.optim.AdamW.step of
 * the DETR / SparseInst trainers (train_transformer.py, train_inseg.py -> optimizer/build.py); `step` is the 1-based
 * update count, seg table as for SGD */
int mi_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const mi_sgd_seg* segs_dev,
                  int nseg, float beta1, float beta2, float eps, int64_t step, float grad_scale, mi_stream_t s);
/* the same update over SEPARATELY ALLOCATED tensors (an eager nn.Module tree: no flat arena) in one launch.  tensors_dev:
 * device table of (param, grad, exp_avg, exp_avg_sq, count, lr, weight_decay); chunks_dev: device table of (tensor index,
 * element offset, count <= 16384), one block each; step_dev: device int64 holding the 1-based update count of THIS step
 * (the caller advances it).  Everything the kernel dereferences is read at run time, so a captured hipGraph stays valid
 * when the tables are refilled (gradient addresses are only known once the capture has ended). */
typedef struct mi_adamw_tensor {
  float* p; const float* g; float* m; float* v;
  int64_t count;
  float lr, weight_decay;
} mi_adamw_tensor;
typedef struct mi_adamw_chunk {
  int32_t tensor, count;
  int64_t offset;
} mi_adamw_chunk;
int mi_adamw_step_multi(const mi_adamw_tensor* tensors_dev, const mi_adamw_chunk* chunks_dev, int nchunks, float beta1,
                        float beta2, float eps, const int64_t* step_dev, float grad_scale, mi_stream_t s);
/* FullModelGradientClippingOptimizer.step (yolov7/optimizer/build.py:206-223: clip_grad_norm_ over all parameters, then
 * the update) over the same tables without touching the host: mi_grad_norm_multi leaves coef_norm_out[0] = min(1, max_norm
 * / (||g||_2 + 1e-6)) and [1] = the norm on the device (partial_dev: nchunks doubles of scratch, summed in index order);
 * (of the gradients times grad_scale); mi_adamw_step_multi_clip multiplies every gradient by grad_scale * *grad_scale_dev
 * (NULL: grad_scale alone) */
int mi_grad_norm_multi(const mi_adamw_tensor* tensors_dev, const mi_adamw_chunk* chunks_dev, int nchunks,
                       double* partial_dev, float max_norm, float grad_scale, float* coef_norm_out, mi_stream_t s);
/* data-parallel form of the captured step (train_transformer.py:188-203 / train_inseg.py:63-77 -> d2 create_ddp_model): the
 * gradients of every tensor copied into ONE flat fp32 buffer (tensor k at element offset flat_off_dev[k]) in one launch -
 * the buffer is all-reduced in a few large messages and the update reads it (a table whose g fields point into it) with
 * grad_scale = 1 / world_size; mi_grad_norm_multi's grad_scale makes the clipped norm the AVERAGED gradient's */
int mi_grad_gather_multi(const mi_adamw_tensor* tensors_dev, const mi_adamw_chunk* chunks_dev, int nchunks,
                         const int64_t* flat_off_dev, float* flat, mi_stream_t s);
int mi_adamw_step_multi_clip(const mi_adamw_tensor* tensors_dev, const mi_adamw_chunk* chunks_dev, int nchunks, float beta1,
                             float beta2, float eps, const int64_t* step_dev, float grad_scale,
                             const float* grad_scale_dev, mi_stream_t s);
/* full-model gradient clipping (FullModelGradientClippingOptimizer, yolov7/optimizer/build.py:206-223 =
 * torch.nn.utils.clip_grad_norm_ over all parameters): grads *= min(1, max_norm / (||grads||_2 + 1e-6)) without a host
 * synchronisation; ws: 1024 doubles of scratch; norm_out (optional, device) receives the norm */
int mi_grad_clip_full_model(float* grads, int64_t n, float max_norm, double* ws, float* norm_out, mi_stream_t s);
/* PositionEmbeddingSine.forward (modeling/backbone/detr_backbone.py:309-375): mask [B][H][W] bytes (non-zero = padding)
 * -> pos fp32 [B][2*num_pos_feats][H][W] */
int mi_pos_embed_sine(const uint8_t* mask, int B, int H, int W, int num_pos_feats, float temperature, int normalize,
                      float scale, int centered, float* out, mi_stream_t s);

/* ---- SparseInst (config 5): bilinear resize (align_corners False) of bf16 NHWC maps and the mask losses of the matched
 * (prediction, target) pairs of SparseInstCriterion (loss/sparseinst_loss.py:123-187).  See csrc/sparseinst_ops.hip. */
int mi_bilinear_resize_bf16(const void* x, int ldx, int N, int H, int W, int C, void* y, int ldy, int Ho, int Wo,
                            mi_stream_t s);
int mi_bilinear_resize_bwd_bf16(const void* dy, int lddy, int N, int H, int W, int C, void* dx, int lddx, int Ho, int Wo,
                                float* acc_ws_zeroed, mi_stream_t s);
/* PyramidPoolingModule's pooling stages (transcoders/encoder_sparseinst.py:18-62: F.avg_pool2d(x, kernel = (kh, kw), stride =
 * kernel, ceil_mode=False) per stage) of one bf16 NHWC map x [N][H][W][C] (ldx) in ONE launch: y[s] bf16 [N][H / kh[s]][W / kw[s]][C]
 * dense; and the backward: dx = sum over the stages of dy[s] spread over its windows / (kh kw) (dy[s] NULL: no gradient).
 * kh / kw / y / dy are HOST arrays of ns <= MI_PYR_MAX_STAGES entries. */
#define MI_PYR_MAX_STAGES 8
int mi_pyramid_pool_fwd(const void* x, int ldx, int N, int H, int W, int C, int ns, const int* kh, const int* kw, void* const* y,
                        mi_stream_t s);
int mi_pyramid_pool_bwd(void* const* dy, int N, int H, int W, int C, int ns, const int* kh, const int* kw, void* dx, int lddx,
                        mi_stream_t s);
/* masks bf16 logits [B][P][ldm] (instance = channel), targets fp32 [T][P], pairs int32 [K][3] = (b, n, t);
 * stats fp32 [K][8] = sum BCE, sum sig*t, sum sig^2, sum t^2, |sig>=.4 & t>.5|, |sig>=.4|, |t>.5|, 0 (rows with b < 0: zeros).
 * ws: mi_sparseinst_mask_stats_ws_floats(K, P) floats of block partials, summed in a fixed order (bit-reproducible; no
 * float atomics) */
int64_t mi_sparseinst_mask_stats_ws_floats(int K, int P);
int mi_sparseinst_mask_stats(const void* masks, int ldm, int P, const float* targets, const int32_t* pairs, int K,
                             float* stats, float* ws, mi_stream_t s);
int mi_sparseinst_mask_grad(const void* masks, int ldm, int P, const float* targets, const int32_t* pairs, int K,
                            const float* stats, float c_bce, float c_dice, void* dmasks_zeroed, mi_stream_t s);
/* the form a captured step uses: pair rows whose image index is < 0 are skipped by both kernels (a fixed-capacity pair table
 * filled by the device-side matching), and the two upstream gradients come from the device (coef_dev[0], coef_dev[1]) */
int mi_sparseinst_mask_grad_dev(const void* masks, int ldm, int P, const float* targets, const int32_t* pairs, int K,
                                const float* stats, const float* coef_dev, void* dmasks, mi_stream_t s);

/* SparseInstCriterion's scalar half and SparseInstMatcher's cost matrix (csrc/sparseinst_loss.hip; loss/sparseinst_loss.py:
 * 190-297, 300-354) - four launches for what the reference spells as ~150 small torch calls.
 * mi_sparseinst_match_cost: cost[b][n][t] = -(dice^alpha * prob^beta), dice = 2 num[b][n][t] / (s2[b][n] + t2[b][t] + 1e-4),
 *   prob = sigmoid(logits[b][n][labels[b][t]]); num fp32 [B][Np][cap], s2 [B][Np], t2 [B][cap], logits fp32 [B][N][C],
 *   labels int64 [B][cap] -> cost fp32 [B][N][cap].
 * The three criterion calls share one descriptor (all device pointers; fixed capacity `cap` pairs per image):
 *   _pairs:        match_q / match_t int64 [B][cap], nmatch int32 [B] (mi_lsap), labels -> pairs int32 [B*cap][3] = (b or -1, q,
 *                  b*cap+t), valid fp32 [B*cap], row_cls / row_pair int32 [B][N] (class / pair of the query's match or -1),
 *                  kdev[0] = max(sum nmatch, 1)
 *   _head_loss:    logits, scores fp32 [B][N] (objectness logits), stats fp32 [B*cap][8] (mi_sparseinst_mask_stats), inv_num[0]
 *                  -> losses[4] = weighted loss_ce, loss_mask, loss_dice, loss_objectness
 *   _head_loss_bwd: gup[4] upstream gradients -> dlogits [B][N][C], dscores [B][N], coef[2] (mi_sparseinst_mask_grad_dev's) */
typedef struct mi_sparseinst_loss_desc {
  const float* logits;
  const float* scores;
  const int64_t* labels;
  const int64_t* match_q;
  const int64_t* match_t;
  const int32_t* nmatch;
  const float* inv_num;
  const float* stats;
  const float* gup;
  int32_t* pairs;
  float* valid;
  int32_t* row_cls;
  int32_t* row_pair;
  float* kdev;
  float* losses;
  float* dlogits;
  float* dscores;
  float* coef;
  int B, N, C, cap, P, use_labels, use_masks, pad_;
  float alpha, gamma, w_ce, w_mask, w_dice, w_obj;
} mi_sparseinst_loss_desc;
int mi_sparseinst_match_cost(const float* num, const float* s2, const float* t2, const float* logits, const int64_t* labels,
                             int B, int N, int Np, int C, int cap, float alpha, float beta, float* cost, mi_stream_t s);
int mi_sparseinst_pairs(const mi_sparseinst_loss_desc* d, mi_stream_t s);
int mi_sparseinst_head_loss(const mi_sparseinst_loss_desc* d, mi_stream_t s);
int mi_sparseinst_head_loss_bwd(const mi_sparseinst_loss_desc* d, mi_stream_t s);

/* ---- box utilities of the DETR path (yolov7/utils/boxes.py:28-37,85-122) ------------------------------------------
 * mi_box_convert: n boxes [n][4] fp32; to_cxcywh 0 = box_cxcywh_to_xyxy, 1 = box_xyxy_to_cxcywh.
 * mi_box_iou_pairwise: box_iou (iou and union, [n][m]) and, when giou != NULL, generalized_box_iou of xyxy boxes in the
 * reference's operation order; *degenerate (may be NULL, caller zeroes it) gets bit 0 / 1 set when a box of the first /
 * second set has x1 < x0 or y1 < y0 - the condition generalized_box_iou asserts on. */
int mi_box_convert(const float* in, float* out, int64_t n, int to_cxcywh, mi_stream_t s);
int mi_box_iou_pairwise(const float* boxes1, int n, const float* boxes2, int m, float* iou, float* uni, float* giou,
                        int32_t* degenerate, mi_stream_t s);

/* sizeof() of the public structs as compiled into the library (binding self-check): 0 mi_conv_desc, 1 mi_wgrad_desc,
 * 2 mi_wgrad_group, 3 mi_pack_job, 4 mi_bias_job, 5 mi_yolox_loss_desc, 6 mi_detr_loss_desc, 7 mi_sgd_seg, 8 mi_cmd,
 * 9 mi_conv_group, 10 mi_bn_job, 11 mi_bn_group, 12 mi_pil_resize_job, 13 mi_jpeg_info, 14 mi_jpeg_job; -1 for an unknown id */
int mi_abi_sizeof(int which);

/* ---- command list executor -----------------------------------------------------
 * A step (forward / backward / update) is a flat list of mi_cmd records built once
 * by the host; mi_cmdlist_run() issues them back-to-back on one stream from C++
 * (no per-launch Python), and mi_graph_* capture/replay the same list as a hipGraph. */
typedef struct mi_cmd {
  int32_t op;          /* MI_OP_* */
  int32_t i[40];
  float f[8];
  void* p[16];
  int64_t l[4];
} mi_cmd;

enum {
  MI_OP_NOP = 0,
  MI_OP_CONV = 1,
  MI_OP_WGRAD = 2,
  MI_OP_PACK_W = 3,
  MI_OP_RESERVED4 = 4, /* was UNPACK_WG: wgrad now writes OIHW directly */
  MI_OP_RESERVED5 = 5, /* was BN_FINALIZE: folded into BN_ACT_FWD */
  MI_OP_BN_ACT_FWD = 6,
  MI_OP_BN_BWD_REDUCE = 7,
  MI_OP_RESERVED8 = 8, /* was BN_BWD_FINALIZE: folded into BN_BWD_APPLY */
  MI_OP_BN_BWD_APPLY = 9,
  MI_OP_FOCUS = 10,
  MI_OP_UPSAMPLE_FWD = 11,
  MI_OP_UPSAMPLE_BWD = 12,
  MI_OP_SPP_FWD = 13,
  MI_OP_SPP_BWD = 14,
  MI_OP_COPY = 15,
  MI_OP_COLSUM = 16,
  MI_OP_LOSS_FWD = 17,
  MI_OP_LOSS_BWD = 18,
  MI_OP_SPLIT_DPREDS = 19,
  MI_OP_MEMSET = 20,
  MI_OP_SGD = 21,
  MI_OP_BN_EVAL_AFFINE = 22,
  MI_OP_DECODE = 23,
  MI_OP_PACK_W_BATCH = 24,
  MI_OP_WGRAD_GROUP = 25,
  MI_OP_STREAM = 26, /* i[0] = stream id for the following commands (0 = the caller's stream, 1..MI_MAX_AUX = aux) */
  MI_OP_FORK = 27,   /* aux stream i[0] waits for everything issued so far on the caller's stream               */
  MI_OP_JOIN = 28,   /* the caller's stream waits for everything issued so far on aux stream i[0]               */
  MI_OP_BIAS_GRADS = 29,
  MI_OP_CONV_GROUP = 30,   /* p0 = mi_conv_group* (host), p1 = device job table */
  MI_OP_BN_GROUP = 31,     /* p0 = mi_bn_group* (host),   p1 = device job table */
  MI_OP_SPLIT_DPREDS_BATCH = 32, /* p0 = mi_split_job* (host), p1 = dpreds, i = B, A, nch, njobs */
  MI_OP_BN_BWD_FUSED = 33, /* BN_BWD_APPLY's arguments + p[12] = barrier words */
  /* depthwise 3x3: i = ldx/lddy, ldy/lddx, N, H, W, C, stride, outH, outW, nslots/accumulate */
  MI_OP_DWCONV_FWD = 34,   /* p = x, w, y, stats_acc */
  MI_OP_DWCONV_DGRAD = 35, /* p = dy, w, dx */
  MI_OP_DWCONV_WGRAD = 36, /* p = x, dy, ws, dw; l0 = ws bytes */
  MI_OP_LOSS_BWD_FUSED = 37, /* p = loss desc, gw, dpreds, split jobs (host), bias jobs (host), ws; i = nsplit, nbias; l0 = ws floats */
  MI_OP_COUNT
};

/* Independent chains (e.g. the three FPN levels of the YOLOX head) may be issued on auxiliary HIP streams owned by
 * the library (STREAM / FORK / JOIN commands): under hipGraph capture they become parallel branches of the graph,
 * which the MI355X runs concurrently (measured: two 64-block chains overlap ~2x). */
#define MI_MAX_AUX 4
int mi_cmdlist_run(const mi_cmd* cmds, int n, mi_stream_t s);
/* A HIP stream whose kernels run only on the compute units of `mask` (hipExtStreamCreateWithCUMask; nwords 32-bit words,
 * bit i of the mask = logical CU i; on the 8-XCD MI355X the driver deals logical CUs round-robin over the XCDs, so the low
 * 64 bits are 8 CUs of EVERY XCD - measured with the census probe of include/mi355_debug.h).  The caller owns the stream
 * (mi_stream_destroy).  Used for the weight-gradient side queue beside the backward chain (engine.NativeTrainer). */
int mi_stream_create_cu_mask(const uint32_t* mask, int nwords, mi_stream_t* out);
int mi_stream_destroy(mi_stream_t s);
/* replace auxiliary stream `sid` (1..MI_MAX_AUX) of the executor's STREAM / FORK / JOIN commands by a caller-owned stream
 * (NULL: back to the library's own); takes effect for command lists run or captured afterwards */
int mi_aux_stream_set(int sid, mi_stream_t s);
/* returns an opaque handle (>0) or <0 */
int64_t mi_graph_capture(const mi_cmd* cmds, int n, mi_stream_t s);
int mi_graph_launch(int64_t handle, mi_stream_t s);
int mi_graph_destroy(int64_t handle);

/* time one command list with HIP events on stream s: runs it `iters` times, returns avg ms in *ms;
 * if per_cmd_ms != NULL (n floats) each command is additionally timed alone (event pair per cmd). */
int mi_cmdlist_time(const mi_cmd* cmds, int n, int iters, float* ms, float* per_cmd_ms, mi_stream_t s);

/* ---- the batch-building half of the DETR / SparseInst steps (csrc/host_feed.hip): one launch per batch where the reference
 * runs a handful of torch calls per image.  The per-image records are passed BY VALUE inside the kernel arguments (`jobs` is
 * a HOST array; nothing is uploaded, nothing has to stay alive), MI_FEED_MAX_IMAGES images per launch.
 *
 * mi_normalize_pad_batch: yolov7/modeling/meta_arch/detr.py:273-278 and meta_arch/sparseinst.py:95-98 -
 *   dst[b][c][y][x] = (img_b[c][y][x] - mean[c]) / std[c] for y < h_b, x < w_b, else 0; dst fp32 [B][3][Hp][Wp] (16-byte
 *   stores when Wp % 4 == 0 and dst is 16-byte aligned), images CHW dense on the device, dtype 0 = fp32, 1 = uint8.
 * mi_mask_targets_batch: yolov7/utils/misc.py:148-170 (nested_masks_from_list) + yolov7/modeling/loss/sparseinst_loss.py:
 *   149-151, 326-328 (F.interpolate bilinear, align_corners=False) - image b's M_b masks [M_b][h_b][w_b] (fp32, or 1-byte
 *   bool / uint8 read as 0 / 1) zero-extended to (Hi, Wi), resized to (Ho, Wo), written as fp32 rows tgt[(b * cap + j)][Ho * Wo]
 *   (rows j >= M_b zero), as their bf16 transpose tgtT[b][Ho * Wo][cap] (NULL: skipped) and labels[b][cap] (int64, 0 beyond
 *   M_b; NULL: skipped; a job's `labels` may be NULL when M == 0).  cap % 8 == 0, 64 * cap * 2 bytes of LDS.
 *   t2 (NULL: skipped): fp32 [B][cap] = sum over the pixels of tgt^2 per row, summed in a fixed order through t2_ws
 *   (B * cap * ceil(Ho * Wo / 64) floats): wave sums of 64 pixels, then the chunks in sequence (a second small launch). */
#define MI_FEED_MAX_IMAGES 32
#define MI_FEED_MAX_LEVELS 8
typedef struct mi_image_job {
  const void* src;
  int h, w, dtype, pad_;
} mi_image_job;
typedef struct mi_mask_job {
  const void* masks;
  const int64_t* labels;
  int M, h, w, dtype;
} mi_mask_job;
int mi_normalize_pad_batch(const mi_image_job* jobs, int B, float* dst, int Hp, int Wp, const float* mean3, const float* std3,
                           mi_stream_t s);
/* mi_padding_masks: MaskedBackbone.mask_out_padding (yolov7/modeling/meta_arch/detr.py:385-403) for all feature levels in one
 *   launch, from the device copy of the image sizes (int64 [B][2] = (h, w)): out[l] uint8 / bool [B][H[l]][W[l]] = 0 where
 *   y < ceil(h / stride[l]) and x < ceil(w / stride[l]), else 1.  out / H / W / stride are HOST arrays of nlev entries. */
int mi_padding_masks(const int64_t* sizes_dev, int B, int nlev, void* const* out, const int* H, const int* W, const int* stride,
                     mi_stream_t s);
int mi_mask_targets_batch(const mi_mask_job* jobs, int B, int cap, int Hi, int Wi, int Ho, int Wo, float* tgt, void* tgtT_bf16,
                          int64_t* labels, float* t2, float* t2_ws, mi_stream_t s);

/* ---- hardware probes used by tests (lane layouts of MFMA / LDS transpose read) */
int mi_probe_mfma32(const void* a_bf16_32x16, const void* b_bf16_16x32, float* d_32x32, mi_stream_t s);
int mi_probe_mfma16(const void* a_bf16_16x32, const void* b_bf16_32x16, float* d_16x16, mi_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* MI355_DET_H */
