// Implicit-GEMM convolution on the gfx950 matrix cores, im2col-free.
//
// One kernel template covers 1x1 / 3x3-s1 / 3x3-s2 forward and their data gradients
// through a tap table (see include/mi355_det.h, mi_conv_desc).  Replaces the ATen/cuDNN
// conv2d the reference reaches from BaseConv (yolov7/modeling/backbone/layers/wrappers.py:60-83)
// and the prediction convs of YOLOXHead (yolov7/modeling/head/yolox_head.py:103-129).
//
// Design (MI355X): block = WM x WN waves, output tile = TPIX (64/128) pixels (TH x TW) x BN output channels.
//   * per k-chunk (KC channels) the input halo tile is streamed HBM/L2 -> LDS ONCE by LDS-DMA
//     (global_load_lds_dwordx4, no VGPR staging) as a pixel-major image [halo pixel][KC] and reused
//     by all taps: a 3x3 conv reads each input element ~1.4x, never 9x; nothing is materialised.
//     16-byte chunks of a row are XOR-permuted on the SOURCE address (an LDS-DMA image is lane-linear)
//     so the ds_read_b128 of 16 consecutive pixels hits 16 distinct bank groups.
//   * weights are pre-packed [tap][K/8][CoutPad][8]: a (tap, k-chunk) slab is a run of contiguous
//     16-byte rows, also LDS-DMA'd.  A step multiplies one k-chunk against `tps` taps (1 .. all of them);
//     the next step's weight slabs and the next halo chunk are in flight (double-buffered) meanwhile;
//     one barrier per step.  Steps are latency-bound (a barrier + the wait for the previous step's DMA), so
//     the launcher makes them as fat as LDS allows: a 3x3 K=32 conv or a 1x1 K<=128 conv is ONE step.
//   * v_mfma_f32_32x32x16_bf16 with A = weights (M = cout), B = pixels (N = pixel): each
//     lane then owns 4 consecutive couts x 4 groups of ONE pixel -> 8-byte NHWC stores.
//   * epilogue optionally emits per-tile per-channel (sum, sumsq) from the fp32 accumulators:
//     the BatchNorm batch statistics cost no extra pass over the conv output.  A data-gradient launch can
//     instead emit the BatchNorm BACKWARD sums (sum dz, sum dz*xhat) of the layer that produced its output
//     tensor (MI_CONV_BNBWD): the final da tile is in registers, only that layer's raw conv output is re-read.
//   * small feature maps (20x20, 40x40) get 64-pixel tiles / narrower cout tiles so that every
//     launch has >= ~2 blocks per CU.
#include <string.h>
#include <string.h>
#include "common.h"
#include "conv_igemm_kernel.h"
#include "conv1x1_stream.h"

// streaming 1x1 path (conv1x1_stream.hip): eligible descriptors leave the tile kernel
bool c1s_try_launch(const mi_conv_desc* ds, int n, hipStream_t s, int* rc);
bool c1s_try_plan(const mi_conv_desc* ds, int n, C1Launch* l);
int c1s_run_planned(const C1Launch* l, hipStream_t s);
bool c1s_try_plan_bn(const mi_conv_desc* ds, const mi_bn_job* bn, int n, C1Launch* l);
int c1s_barrier_status(unsigned* flag);
static_assert(sizeof(C1Launch) <= sizeof(((mi_conv_group*)0)->priv), "mi_conv_group.priv holds the stream launch record");
#include "conv3x3_ws.h"
// weight-stationary 3x3 path (conv3x3_ws.hip)
bool w3_try_launch(const mi_conv_desc* ds, int n, hipStream_t s, int* rc);
bool w3_try_plan(const mi_conv_desc* ds, int n, W3Launch* l);
int w3_run_planned(const W3Launch* l, hipStream_t s);
bool w3_try_plan_bn(const mi_conv_desc* ds, const mi_bn_job* bn, int n, W3Launch* l);
int w3_barrier_status(unsigned* flag);
static_assert(sizeof(W3Launch) <= sizeof(((mi_conv_group*)0)->priv), "mi_conv_group.priv holds the 3x3 launch record");

#define MI_DECL_KC(KCv)                                                                      \
  int conv_launch_kc##KCv(const ConvK& k, int BN, int TPIX, size_t lds, hipStream_t s);       \
  int conv_group_launch_kc##KCv(const mi_conv_group* m, const ConvK* jobs, const int* starts, hipStream_t s);
MI_DECL_KC(16) MI_DECL_KC(32) MI_DECL_KC(64) MI_DECL_KC(128)
#undef MI_DECL_KC

// ---------------------------------------------------------------- host side
static void choose_tile(int TPIX, int gridH, int gridW, int* TH, int* TW) {
  const int cands[] = {gridW, 64, 32, 16, 8, 4};
  long best = -1;
  int bh = 8, bw = 16;
  for (int c : cands) {
    if (c <= 0 || c > TPIX) continue;
    int tw = c;
    int th = TPIX / tw;
    if (th > gridH) th = gridH;
    if (th < 1) th = 1;
    long tiles = (long)mi_cdiv(gridH, th) * mi_cdiv(gridW, tw);
    long halo = (long)(th + 2) * (tw + 2);
    long score = tiles * 100000 + halo;
    if (best < 0 || score < best) {
      best = score;
      bh = th;
      bw = tw;
    }
  }
  *TH = bh;
  *TW = bw;
}

struct ConvCfg {
  int KC, BN, TPIX, TPS;
};

#ifdef MI_CONV_TIMELINE
static long long* g_conv_tl = nullptr;
extern "C" void mi_debug_conv_timeline(long long* dev_buf) { g_conv_tl = dev_buf; }
#endif
static int conv_fill(const mi_conv_desc* d, ConvK* k, ConvCfg* c, size_t* ldsBytes) {
  MI_REQUIRE(d->x && d->w && d->y, "conv: null pointer");
  MI_REQUIRE(d->ntaps >= 1 && d->ntaps <= MI_MAX_TAPS, "conv: ntaps %d", d->ntaps);
  MI_REQUIRE(d->K8 >= 2 && (d->K8 % 2) == 0, "conv: K8 %d must be even", d->K8);
  MI_REQUIRE(d->CoutPad % 32 == 0 && d->Cout <= d->CoutPad && d->Cout > 0, "conv: Cout %d pad %d",
             d->Cout, d->CoutPad);
  MI_REQUIRE(d->ldx % 8 == 0 && ((uintptr_t)d->x % 16) == 0, "conv: x must be 16B aligned (ldx %d)", d->ldx);
  MI_REQUIRE(d->in_stride == 1 || d->in_stride == 2, "conv: in_stride");
  MI_REQUIRE(d->out_stride == 1 || d->out_stride == 2, "conv: out_stride");
  MI_REQUIRE((long long)d->H * d->W * d->ldx < (1LL << 30), "conv: image plane too large for 32-bit offsets");
  if (!(d->flags & MI_CONV_OUT_F32))
    MI_REQUIRE((d->Cout % 4 != 0) || (d->ldy % 4 == 0 && ((uintptr_t)d->y % 8) == 0),
               "conv: bf16 y needs 8B alignment (ldy %d)", d->ldy);
  k->x = (const __bf16*)d->x;
  k->w = (const u32x4*)d->w;
  k->y = d->y;
  k->bias = d->bias;
  k->stats = d->stats_acc;
  k->nslots = (d->stats_slots >= 1 && d->stats_slots <= MI_BN_SLOTS) ? d->stats_slots : MI_BN_SLOTS;
  k->ynstride = d->y_nstride > 0 ? (long long)d->y_nstride : (long long)d->outH * d->outW * d->ldy;
  k->ldx = d->ldx; k->ldy = d->ldy; k->N = d->N; k->H = d->H; k->W = d->W;
  k->outH = d->outH; k->outW = d->outW; k->gridH = d->gridH; k->gridW = d->gridW;
  k->is = d->in_stride; k->os = d->out_stride; k->ooy = d->out_oy; k->oox = d->out_ox;
  k->K8 = d->K8; k->Cout = d->Cout; k->CoutPad = d->CoutPad; k->ntaps = d->ntaps;
  int dymin = 1 << 30, dymax = -(1 << 30), dxmin = 1 << 30, dxmax = -(1 << 30);
  for (int t = 0; t < d->ntaps; ++t) {
    k->tw[t] = d->tap_w[t];
    if (d->tap_dy[t] < dymin) dymin = d->tap_dy[t];
    if (d->tap_dy[t] > dymax) dymax = d->tap_dy[t];
    if (d->tap_dx[t] < dxmin) dxmin = d->tap_dx[t];
    if (d->tap_dx[t] > dxmax) dxmax = d->tap_dx[t];
  }
  k->flags = d->flags;
  {
    static const int noatom = getenv("MI_DEBUG_NOATOM") ? atoi(getenv("MI_DEBUG_NOATOM")) : 0;
    if (noatom) k->flags |= 1024;
  }
  MI_REQUIRE(!(d->flags & MI_CONV_RELU) || !((d->flags & (MI_CONV_ACCUM | MI_CONV_BNBWD | MI_CONV_OUT_F32 | MI_CONV_RELUMASK | MI_CONV_ADDRELU)) || d->stats_acc),
             "conv: MI_CONV_RELU is a plain forward epilogue (bf16 output, no accumulate / statistics)");
  if (d->flags & MI_CONV_BNBWD) {
    MI_REQUIRE(d->stats_acc && d->bn_y && d->bn_scale && d->bn_shift && d->bn_mean && d->bn_invstd,
               "conv: MI_CONV_BNBWD needs stats_acc and the producing layer's bn_* arrays");
    MI_REQUIRE(!(d->flags & MI_CONV_OUT_F32) && d->Cout % 8 == 0 && d->bn_ldy % 8 == 0 && ((uintptr_t)d->bn_y % 16) == 0,
               "conv: MI_CONV_BNBWD needs a bf16 output with Cout %% 8 == 0 (staged epilogue) and 16B-aligned bn_y");
    k->bn_y = (const __bf16*)d->bn_y; k->bn_scale = d->bn_scale; k->bn_shift = d->bn_shift;
    k->bn_mean = d->bn_mean; k->bn_invstd = d->bn_invstd; k->bn_ldy = d->bn_ldy; k->bn_act = d->bn_act;
  } else if (d->flags & (MI_CONV_RELUMASK | MI_CONV_ADDRELU)) {
    MI_REQUIRE((d->flags & (MI_CONV_RELUMASK | MI_CONV_ADDRELU)) != (MI_CONV_RELUMASK | MI_CONV_ADDRELU) && !d->stats_acc,
               "conv: MI_CONV_RELUMASK and MI_CONV_ADDRELU exclude each other and the forward statistics");
    MI_REQUIRE(d->bn_y && !(d->flags & MI_CONV_OUT_F32) && d->Cout % 8 == 0 && d->bn_ldy % 8 == 0 && d->bn_ldy >= d->Cout &&
               ((uintptr_t)d->bn_y % 16) == 0,
               "conv: the aux-tensor epilogue needs bn_y (16B-aligned, bn_ldy %% 8 == 0) and a bf16 output with Cout %% 8 == 0");
    k->bn_y = (const __bf16*)d->bn_y; k->bn_ldy = d->bn_ldy;
    k->bn_scale = k->bn_shift = k->bn_mean = k->bn_invstd = nullptr; k->bn_act = 0;
  } else {
    k->bn_y = nullptr; k->bn_scale = k->bn_shift = k->bn_mean = k->bn_invstd = nullptr; k->bn_ldy = 0; k->bn_act = 0;
  }
  // ---- tile configuration: largest tile that still gives >= ~2 blocks per CU
  const int Kp = d->K8 * 8;
  int BNmax = (d->CoutPad % 128 == 0) ? 128 : (d->CoutPad % 64 == 0) ? 64 : 32;
  int BN = d->BN, TPIX = 0, TH = d->TH, TW = d->TW;
  if (TH > 0 && TW > 0) TPIX = (TH * TW <= 64) ? 64 : 128;
  if (BN <= 0 || TPIX == 0) {
    // preference order: big tiles first; take the first (pixel tile, cout tile) pair that fills the chip once
    const long want = 256;
    const int order[5][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {64, 32}};
    int bestBN = BNmax, bestTP = TPIX ? TPIX : 128;
    for (int o = 0; o < 5; ++o) {
      const int tp = order[o][0];
      int bn = order[o][1];
      if (bn > BNmax) bn = BNmax;
      if (TPIX && tp != TPIX) continue;
      if (d->BN > 0 && bn != d->BN) continue;
      int th, tw;
      choose_tile(tp, d->gridH, d->gridW, &th, &tw);
      const long tiles = (long)d->N * mi_cdiv(d->gridH, th) * mi_cdiv(d->gridW, tw);
      bestBN = bn; bestTP = tp;
      if (tiles * (d->CoutPad / bn) >= want) break;
    }
    BN = bestBN; TPIX = bestTP;
  }
  MI_REQUIRE(BN == 32 || BN == 64 || BN == 128, "conv: BN %d", BN);
  MI_REQUIRE(d->CoutPad % BN == 0, "conv: CoutPad %d %% BN %d", d->CoutPad, BN);
  if (TH <= 0 || TW <= 0) choose_tile(TPIX, d->gridH, d->gridW, &TH, &TW);
  MI_REQUIRE(TH * TW <= TPIX && TH >= 1 && TW >= 1, "conv: tile %dx%d", TH, TW);
  k->TH = TH; k->TW = TW;
  k->tilesY = mi_cdiv(d->gridH, TH); k->tilesX = mi_cdiv(d->gridW, TW);
  k->dymin = dymin; k->dxmin = dxmin;
  const int haloH = (TH - 1) * d->in_stride + (dymax - dymin) + 1;
  k->haloW = (TW - 1) * d->in_stride + (dxmax - dxmin) + 1;
  k->npixh = haloH * k->haloW;
  for (int t = 0; t < d->ntaps; ++t) k->toff[t] = (d->tap_dy[t] - dymin) * k->haloW + (d->tap_dx[t] - dxmin);
  // ---- k-chunk and taps per step.  A step costs a barrier + the wait for the previous step's DMA (~0.25 us), a
  // (chunk, tap) iteration inside it exposes one LDS round trip (~0.09 us); co-resident blocks (LDS-, register- and
  // grid-limited) hide each other's latencies.  Constants fitted to tools/conv_sweep.py on MI355X; the plan-level
  // autotuner (plan.py) replaces this estimate by measurements.
  k->nco = d->CoutPad / BN;
  const long nblocks = (long)d->N * k->tilesY * k->tilesX * k->nco;
  int KC = d->KC, tps = d->TPS;
  auto lds_need = [&](int kc, int tp) {
    const size_t xb = (size_t)mi_cdiv(k->npixh, 64 / (kc / 8)) * 1024;
    const size_t wb = (size_t)tp * (kc / 8) * BN * 16;
    const int nch = Kp / kc, nst = nch * (d->ntaps / tp);
    return xb * (nch > 1 ? 2 : 1) + wb * (nst > 1 ? 2 : 1);
  };
  if (KC <= 0 || tps <= 0) {
    static const int maxkc = getenv("MI_CONV_MAXKC") ? atoi(getenv("MI_CONV_MAXKC")) : 128;
    static const int cap2 = getenv("MI_CONV_LDSCAP") ? atoi(getenv("MI_CONV_LDSCAP")) : 80;
    static const int cap1 = getenv("MI_CONV_LDSCAP1") ? atoi(getenv("MI_CONV_LDSCAP1")) : 160;
    static const double cS = getenv("MI_CONV_S") ? atof(getenv("MI_CONV_S")) : 0.25;
    static const double cT = getenv("MI_CONV_T") ? atof(getenv("MI_CONV_T")) : 0.3;
    static const int useocc = getenv("MI_CONV_OCC") ? atoi(getenv("MI_CONV_OCC")) : 1;
    const size_t cap = (size_t)(nblocks > 256 ? cap2 : cap1) * 1024;
    double bestCost = -1.0;
    int bk = 16, bt = 1;
    size_t bestLds = 0;
    const double vocc = BN == 32 ? 8.0 : (BN == 64 ? (TPIX == 128 ? 4.0 : 8.0) : (TPIX == 128 ? 3.0 : 5.0));
    const double gocc = nblocks > 256 ? nblocks / 256.0 : 1.0;
    const int kcs[4] = {128, 64, 32, 16};
    for (int ci = 0; ci < 4; ++ci) {
      const int kc = kcs[ci];
      if (Kp % kc != 0 || (d->KC > 0 && kc != d->KC) || (d->KC <= 0 && kc > maxkc)) continue;
      for (int tp = d->ntaps; tp >= 1; --tp) {
        if (d->ntaps % tp != 0 || (d->TPS > 0 && tp != d->TPS)) continue;
        const size_t need = lds_need(kc, tp);
        if (need > cap && !(kc == 16 && tp == 1)) continue;
        double occ = (double)((160 * 1024) / (need > 20480 ? need : 20480));
        if (occ < 1.0) occ = 1.0;
        if (occ > vocc) occ = vocc;
        if (occ > gocc) occ = gocc;
        if (!useocc) occ = 1.0;
        const double steps = (double)(Kp / kc) * (d->ntaps / tp), iters = (double)(Kp / kc) * d->ntaps;
        const double cost = (steps * cS + iters * cT) / occ;
        if (bestCost < 0 || cost < bestCost * 0.999 || (cost <= bestCost * 1.001 && need < bestLds)) {
          bestCost = cost; bk = kc; bt = tp; bestLds = need;
        }
      }
    }
    KC = bk; tps = bt;
  }
  MI_REQUIRE((KC == 16 || KC == 32 || KC == 64 || KC == 128) && Kp % KC == 0, "conv: KC %d for K %d", KC, Kp);
  MI_REQUIRE(tps >= 1 && d->ntaps % tps == 0, "conv: TPS %d for %d taps", tps, d->ntaps);
  const int RPI = 64 / (KC / 8);
  k->nqx = mi_cdiv(k->npixh, RPI);
  k->xbytes = k->nqx * 1024;
  MI_REQUIRE((long)k->nqx * RPI * k->haloW < (1 << 20) && k->nqx * RPI < 4096, "conv: halo too large for the row decode");
  k->mTW = ((1u << 20) + TW - 1) / TW;
  k->mHW = ((1u << 20) + k->haloW - 1) / k->haloW;
  const int nch = Kp / KC, nst = nch * (d->ntaps / tps);
  k->tps = tps;
  k->xstride = nch > 1 ? k->xbytes : 0;
  k->wstride = nst > 1 ? tps * (KC / 8) * BN * 16 : 0;
  c->KC = KC; c->BN = BN; c->TPIX = TPIX; c->TPS = tps;
  {
    static const int xm = getenv("MI_CONV_XMAP") ? atoi(getenv("MI_CONV_XMAP")) : 2;
    // 1: launches with several cout tiles per pixel tile; 2: also multi-tap launches (vertically adjacent tiles share
    // halo rows); 3: every launch
    const bool on = xm >= 3 || (xm == 2 && (k->nco > 1 || d->ntaps > 1)) || (xm == 1 && k->nco > 1);
    k->xmap = (on && nblocks % 8 == 0) ? (int)(nblocks / 8) : 0;
#ifdef MI_CONV_TIMELINE
    k->tl = g_conv_tl;
#endif
  }
  *ldsBytes = lds_need(KC, tps);
  const size_t stage = (size_t)TPIX * (BN * 2 + 16);          // staged epilogue tile
  const size_t red = (size_t)16 * BN * 8;                     // statistics rows of up to 16 waves
  if (*ldsBytes < stage) *ldsBytes = stage;
  if (*ldsBytes < red) *ldsBytes = red;
  if (*ldsBytes < 4 * BN * 8) *ldsBytes = 4 * BN * 8;
  MI_REQUIRE(*ldsBytes <= 160 * 1024, "conv: LDS %zu too large", *ldsBytes);
  return MI_OK;
}

// which kernel family mi_conv2d runs for this descriptor: 0 tile (conv_igemm), 1 streaming 1x1, 2 weight-stationary 3x3
extern "C" int mi_conv2d_route(const mi_conv_desc* d) {
  MI_REQUIRE(d, "conv2d_route: null");
  C1Launch cl;
  if (c1s_try_plan(d, 1, &cl)) return 1;
  W3Launch wl;
  if (w3_try_plan(d, 1, &wl)) return 2;
  return d->xf ? -1 : 0;   // (no kernel applies the input BatchNorm on the tile path: -1 without an error record)
}

extern "C" int mi_conv2d_plan(mi_conv_desc* d) {
  ConvK k;
  ConvCfg c;
  size_t lds;
  int rc = conv_fill(d, &k, &c, &lds);
  if (rc) return rc;
  d->TH = k.TH; d->TW = k.TW; d->KC = c.KC; d->BN = c.BN; d->TPS = c.TPS;
  return d->N * k.tilesY * k.tilesX;
}

extern "C" int mi_conv2d(const mi_conv_desc* d, mi_stream_t st) {
  {
    int rc = MI_OK;
    if (d && c1s_try_launch(d, 1, (hipStream_t)st, &rc)) return rc;
    if (d && w3_try_launch(d, 1, (hipStream_t)st, &rc)) return rc;
  }
  MI_REQUIRE(!d || !d->xf, "conv: the BatchNorm of the input (xf) runs on the streaming 1x1 / weight-stationary 3x3 kernels only");
  ConvK k;
  ConvCfg c;
  size_t lds;
  int rc = conv_fill(d, &k, &c, &lds);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)st;
  // 8-wave form of the same tile: half the 32x32 MFMA tiles, half the staging / store rows and ~40 % fewer registers per
  // wave.  The blocks are latency-bound per wave (LDS round trips of the fragments, the serial epilogue), so twice the
  // waves per SIMD on the same data is worth 1.5 % (fast boxes) to 6 % (slow boxes) of the step; 16 waves lose again.
  // MI_CONV_W8 / MI_CONV_W8G: largest grid (blocks) of a single / grouped launch that takes it (0: never).
  static const long w8max = getenv("MI_CONV_W8") ? atol(getenv("MI_CONV_W8")) : (1L << 40);
  static const int w8mask = getenv("MI_CONV_W8MASK") ? atoi(getenv("MI_CONV_W8MASK")) : 7;
  const long nblocks = (long)k.N * k.tilesY * k.tilesX * k.nco;
  int tp = c.TPIX;
  const int cls = (c.TPIX == 128 && c.BN == 64) ? 1 : (c.TPIX == 128 && c.BN == 128) ? 2 : (c.TPIX == 64 && c.BN == 128) ? 4 : 0;
  if (nblocks <= w8max && (cls & w8mask) && !(d->flags & MI_CONV_OUT_F32) && (d->Cout & 7) == 0) {
    tp |= 1024;
  }
  switch (c.KC) {
    case 16: return conv_launch_kc16(k, c.BN, tp, lds, s);
    case 32: return conv_launch_kc32(k, c.BN, tp, lds, s);
    case 64: return conv_launch_kc64(k, c.BN, tp, lds, s);
    case 128: return conv_launch_kc128(k, c.BN, tp, lds, s);
  }
  MI_FAIL(MI_EINVAL, "conv: no kernel for KC %d BN %d TPIX %d", c.KC, c.BN, c.TPIX);
}

// ---------------------------------------------------------------- grouped launch
extern "C" int mi_conv2d_group_plan(const mi_conv_desc* descs, int n, void* table_host, int64_t table_cap,
                                    mi_conv_group* meta) {
  MI_REQUIRE(descs && meta && n >= 1 && n <= MI_CONV_MAX_GROUP, "conv_group_plan: 1..%d jobs", MI_CONV_MAX_GROUP);
  {
    // 1x1 convolutions of ONE input tensor (CSP conv1 + conv2): a single streaming launch that reads it once
    C1Launch cl;
    if (c1s_try_plan(descs, n, &cl)) {
      memset(meta, 0, sizeof(*meta));
      meta->njobs = n; meta->nblocks = cl.grid; meta->lds_bytes = cl.lds;
      meta->KC = -1; meta->BN = cl.WM * 32; meta->TPIX = cl.PT; meta->TPS = cl.NBUF; meta->EPI = cl.MODE;
      meta->starts_off = 0; meta->table_bytes = 16;   // no device table: the launch record travels in meta->priv
      memcpy(meta->priv, &cl, sizeof(cl));
      if (table_host) {
        MI_REQUIRE(table_cap >= meta->table_bytes, "conv_group_plan: table too small");
        memset(table_host, 0, (size_t)meta->table_bytes);
      }
      return MI_OK;
    }
    // 3x3 K -> K convolutions (the head's levels x towers): persistent weight-stationary blocks, one launch
    W3Launch wl;
    if (w3_try_plan(descs, n, &wl)) {
      memset(meta, 0, sizeof(*meta));
      meta->njobs = n; meta->nblocks = wl.grid; meta->lds_bytes = wl.lds;
      meta->KC = -2; meta->BN = wl.K; meta->TPIX = 128; meta->TPS = 9; meta->EPI = wl.MODE;
      meta->starts_off = 0; meta->table_bytes = 16;
      memcpy(meta->priv, &wl, sizeof(wl));
      if (table_host) {
        MI_REQUIRE(table_cap >= meta->table_bytes, "conv_group_plan: table too small");
        memset(table_host, 0, (size_t)meta->table_bytes);
      }
      return MI_OK;
    }
  }
  for (int j = 0; j < n; ++j)
    MI_REQUIRE(!descs[j].xf, "conv_group_plan: the BatchNorm of the input (xf) runs on the streaming 1x1 / weight-stationary 3x3 kernels only");
  // the job with the most output pixels picks the configuration; the others are forced onto its template
  int big = 0;
  for (int j = 1; j < n; ++j)
    if ((long)descs[j].N * descs[j].gridH * descs[j].gridW > (long)descs[big].N * descs[big].gridH * descs[big].gridW) big = j;
  ConvK kb;
  ConvCfg cb;
  size_t lb;
  int rc = conv_fill(&descs[big], &kb, &cb, &lb);
  if (rc) return rc;
  ConvK ks[MI_CONV_MAX_GROUP];
  int starts[MI_CONV_MAX_GROUP + 1];
  size_t lds = 0;
  int epi = -1;
  starts[0] = 0;
  for (int j = 0; j < n; ++j) {
    mi_conv_desc d = descs[j];
    MI_REQUIRE((d.K8 * 8) % cb.KC == 0 && d.CoutPad % cb.BN == 0 && d.ntaps % cb.TPS == 0,
               "conv_group_plan: job %d does not fit the group's configuration (KC %d BN %d TPS %d)", j, cb.KC, cb.BN, cb.TPS);
    d.KC = cb.KC; d.BN = cb.BN; d.TPS = cb.TPS;
    ConvCfg c;
    size_t l = 0;
    if (j == big) {
      d.TH = kb.TH; d.TW = kb.TW;
      rc = conv_fill(&d, &ks[j], &c, &l);
      if (rc) return rc;
    } else {
      // same pixel-tile CLASS (64 / 128 pixels) as the group; among the tile shapes of that class take the one with
      // the fewest tiles whose LDS footprint does not exceed the leading job's (a wider halo - e.g. 3x40 tiles of a
      // 40x40 map next to 8x16 tiles of an 80x80 map - would cut the occupancy of EVERY block of the launch)
      const size_t cap = lb > 80 * 1024 ? lb : 80 * 1024;
      const int cands[6][2] = {{0, 0}, {8, 16}, {4, 32}, {16, 8}, {8, 8}, {4, 16}};
      long bestTiles = -1;
      for (int q = 0; q < 6; ++q) {
        mi_conv_desc t = d;
        if (q == 0) {
          int th, tw;
          choose_tile(cb.TPIX, d.gridH, d.gridW, &th, &tw);
          t.TH = th; t.TW = tw;
        } else {
          t.TH = cands[q][0]; t.TW = cands[q][1];
        }
        if ((t.TH * t.TW <= 64 ? 64 : 128) != cb.TPIX) continue;
        ConvK kt;
        ConvCfg ct;
        size_t lt;
        if (conv_fill(&t, &kt, &ct, &lt) != MI_OK) continue;
        if (lt > cap || ct.KC != cb.KC || ct.BN != cb.BN || ct.TPS != cb.TPS) continue;
        const long tiles = (long)kt.N * kt.tilesY * kt.tilesX;
        if (bestTiles < 0 || tiles < bestTiles) { bestTiles = tiles; ks[j] = kt; c = ct; l = lt; }
      }
      MI_REQUIRE(bestTiles > 0, "conv_group_plan: job %d has no tile shape within the group's LDS footprint", j);
    }
    MI_REQUIRE(c.KC == cb.KC && c.BN == cb.BN && c.TPIX == cb.TPIX && c.TPS == cb.TPS,
               "conv_group_plan: job %d resolved to another configuration", j);
    MI_REQUIRE(!(ks[j].flags & (MI_CONV_RELUMASK | MI_CONV_ADDRELU)), "conv_group_plan: the aux-tensor epilogue (RELUMASK / ADDRELU) is a single-launch feature");
    const int e = (ks[j].flags & (MI_CONV_ACCUM | MI_CONV_BNBWD)) != 0;
    MI_REQUIRE(epi < 0 || epi == e, "conv_group_plan: jobs mix accumulating and plain launches");
    epi = e;
    if (l > lds) lds = l;
    if (starts[j] % 8) ks[j].xmap = 0;   // the XCD of a block is its GLOBAL id % 8: the remap needs the job to start on XCD 0
    starts[j + 1] = starts[j] + ks[j].N * ks[j].tilesY * ks[j].tilesX * ks[j].nco;
  }
  int tpix = cb.TPIX;
  {
    static const long w8max = getenv("MI_CONV_W8G") ? atol(getenv("MI_CONV_W8G")) : (1L << 40);
    static const int w8mask = getenv("MI_CONV_W8MASK") ? atoi(getenv("MI_CONV_W8MASK")) : 7;
    const int cls = (cb.TPIX == 128 && cb.BN == 64) ? 1 : (cb.TPIX == 128 && cb.BN == 128) ? 2 : (cb.TPIX == 64 && cb.BN == 128) ? 4 : 0;
    bool staged = true;
    for (int j = 0; j < n; ++j) staged = staged && !(descs[j].flags & MI_CONV_OUT_F32) && (descs[j].Cout & 7) == 0;
    if (starts[n] <= w8max && (cls & w8mask) && staged) {
      tpix |= 1024;   // 8-wave form (see mi_conv2d)
    }
  }
  meta->njobs = n; meta->nblocks = starts[n]; meta->lds_bytes = (int32_t)lds;
  meta->KC = cb.KC; meta->BN = cb.BN; meta->TPIX = tpix; meta->TPS = cb.TPS; meta->EPI = epi;
  meta->starts_off = (int64_t)sizeof(ConvK) * n;
  meta->table_bytes = meta->starts_off + (int64_t)sizeof(int) * (n + 1);
  if (table_host) {
    MI_REQUIRE(table_cap >= meta->table_bytes, "conv_group_plan: table too small");
    memcpy(table_host, ks, sizeof(ConvK) * n);
    memcpy((char*)table_host + meta->starts_off, starts, sizeof(int) * (n + 1));
  }
  return MI_OK;
}

// ---------------------------------------------------------------- convolution + BatchNorm(train) + activation, one launch
extern "C" int mi_conv2d_bn_plan(const mi_conv_desc* descs, const mi_bn_job* bn, int n, mi_conv_group* meta) {
  MI_REQUIRE(descs && bn && meta && n >= 1 && n <= MI_CONV_MAX_GROUP, "conv_bn_plan: 1..%d jobs", MI_CONV_MAX_GROUP);
  C1Launch cl;
  if (c1s_try_plan_bn(descs, bn, n, &cl)) {
    memset(meta, 0, sizeof(*meta));
    meta->njobs = n; meta->nblocks = cl.grid; meta->lds_bytes = cl.lds;
    meta->KC = -1; meta->BN = cl.WM * 32; meta->TPIX = cl.PT; meta->TPS = cl.NBUF; meta->EPI = cl.MODE;
    meta->table_bytes = 16;
    memcpy(meta->priv, &cl, sizeof(cl));
    return 1;
  }
  W3Launch wl;
  if (w3_try_plan_bn(descs, bn, n, &wl)) {
    memset(meta, 0, sizeof(*meta));
    meta->njobs = n; meta->nblocks = wl.grid; meta->lds_bytes = wl.lds;
    meta->KC = -2; meta->BN = wl.K; meta->TPIX = 128; meta->TPS = 9; meta->EPI = wl.MODE;
    meta->table_bytes = 16;
    memcpy(meta->priv, &wl, sizeof(wl));
    return 1;
  }
  return 0;
}

extern "C" int mi_conv2d_bn_fwd(const mi_conv_desc* descs, const mi_bn_job* bn, int n, mi_stream_t st) {
  mi_conv_group meta;
  const int rc = mi_conv2d_bn_plan(descs, bn, n, &meta);
  if (rc < 0) return rc;
  MI_REQUIRE(rc == 1, "conv_bn_fwd: needs convolutions that run on the streaming 1x1 (at most %d of one input) or the weight-"
             "stationary 3x3 kernel with stats_acc, each followed by the train-mode BatchNorm of exactly its output", C1_MAX_BN);
  if (meta.KC == -1) return c1s_run_planned((const C1Launch*)meta.priv, (hipStream_t)st);
  return w3_run_planned((const W3Launch*)meta.priv, (hipStream_t)st);
}

extern "C" int mi_conv_bn_barrier_status(uint32_t* flags) {
  MI_REQUIRE(flags, "conv_bn_barrier_status: null");
  unsigned a = 0, b = 0;
  int rc = c1s_barrier_status(&a);
  if (rc) return rc;
  rc = w3_barrier_status(&b);
  if (rc) return rc;
  *flags = (a ? 1u : 0u) | (b ? 2u : 0u);
  return MI_OK;
}

extern "C" int mi_conv2d_group_run(const mi_conv_group* m, const void* table_dev, mi_stream_t st) {
  MI_REQUIRE(m && (table_dev || m->KC < 0) && m->njobs >= 1, "conv_group_run: null");
  if (m->KC == -1) return c1s_run_planned((const C1Launch*)m->priv, (hipStream_t)st);
  if (m->KC == -2) return w3_run_planned((const W3Launch*)m->priv, (hipStream_t)st);
  const ConvK* jobs = (const ConvK*)table_dev;
  const int* starts = (const int*)((const char*)table_dev + m->starts_off);
  hipStream_t s = (hipStream_t)st;
  switch (m->KC) {
    case 16: return conv_group_launch_kc16(m, jobs, starts, s);
    case 32: return conv_group_launch_kc32(m, jobs, starts, s);
    case 64: return conv_group_launch_kc64(m, jobs, starts, s);
    case 128: return conv_group_launch_kc128(m, jobs, starts, s);
  }
  MI_FAIL(MI_EINVAL, "conv_group: no kernel for KC %d BN %d TPIX %d", m->KC, m->BN, m->TPIX);
}

// ================================================================= weight (un)packing
__device__ __forceinline__ void pack_w_body(const float* __restrict__ w, int Cout, int Cin, int KK, __bf16* wf,
                                            int CinPad, int CoutPad, __bf16* wd, int CoutPadK, int CinPadN,
                                            const float* __restrict__ cs = nullptr) {   // cs: per-Cout factor (fp32 product, then the bf16 rounding)
  // forward image: wf[tap][ci/8][co][ci%8]
  const int64_t nf = wf ? (int64_t)KK * CinPad * CoutPad : 0;
  const int64_t nd = wd ? (int64_t)KK * CoutPadK * CinPadN : 0;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < nf + nd;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx < nf) {
      const int e = idx & 7;
      int64_t r = idx >> 3;
      const int co = r % CoutPad; r /= CoutPad;
      const int k8 = r % (CinPad / 8);
      const int tap = r / (CinPad / 8);
      const int ci = k8 * 8 + e;
      float v = 0.f;
      if (co < Cout && ci < Cin) v = w[((int64_t)co * Cin + ci) * KK + tap] * (cs ? cs[co] : 1.f);
      wf[idx] = (__bf16)v;
    } else {
      // dgrad image: wd[tap][co/8][ci][co%8]   (k = cout, "n" = cin)
      const int64_t i2 = idx - nf;
      const int e = i2 & 7;
      int64_t r = i2 >> 3;
      const int ci = r % CinPadN; r /= CinPadN;
      const int k8 = r % (CoutPadK / 8);
      const int tap = r / (CoutPadK / 8);
      const int co = k8 * 8 + e;
      float v = 0.f;
      if (co < Cout && ci < Cin) v = w[((int64_t)co * Cin + ci) * KK + tap] * (cs ? cs[co] : 1.f);
      wd[i2] = (__bf16)v;
    }
  }
}

__global__ void pack_w_kernel(const float* __restrict__ w, int Cout, int Cin, int KK, __bf16* wf, int CinPad,
                              int CoutPad, __bf16* wd, int CoutPadK, int CinPadN) {
  pack_w_body(w, Cout, Cin, KK, wf, CinPad, CoutPad, wd, CoutPadK, CinPadN);
}
__global__ void pack_w_scaled_kernel(const float* __restrict__ w, const float* __restrict__ cs, int Cout, int Cin, int KK,
                                     __bf16* wf, int CinPad, int CoutPad, __bf16* wd, int CoutPadK, int CinPadN) {
  pack_w_body(w, Cout, Cin, KK, wf, CinPad, CoutPad, wd, CoutPadK, CinPadN, cs);
}
// all layers in one flat launch, one block per (32 output channels x 64 input channels x all taps) tile of one layer:
// the tile's fp32 weights are read in memory order (coalesced), rounded to bf16 into LDS, and both packed images are written
// from there with 16-byte stores whose neighbours belong to neighbouring threads - forward image wf[tap][ci/8][co][ci%8]
// along co, data-gradient image wd[tap][co/8][ci][co%8] along ci.  (A thread-per-item gather read 4 bytes of 64 different
// cache lines per instruction: 100 us per step for 36 MB of weights; this transpose runs at the copy rate.)
// blk0 = first block of the job (prefix sum of its tile count, mi_pack_jobs_layout); a block finds its job by bisection.
#define PK_CO 32
#define PK_CI 64
__global__ __launch_bounds__(256) void pack_w_batch_kernel(const mi_pack_job* __restrict__ jobs, int njobs, int kk_max) {
  extern __shared__ __attribute__((aligned(16))) __bf16 pk_s[];   // [PK_CO][PK_CI * kk_max + 2]
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const mi_pack_job j = jobs[lo];
  const int KK = j.KK, tid = threadIdx.x;
  const int coP = j.wd && j.CoutPadK > j.CoutPad ? j.CoutPadK : (j.wf ? j.CoutPad : j.CoutPadK);
  const int nco_t = (coP + PK_CO - 1) / PK_CO;
  const int t = (int)blockIdx.x - j.blk0;
  const int co0 = (t % nco_t) * PK_CO, ci0 = (t / nco_t) * PK_CI;
  const int rs = PK_CI * kk_max + 2;                       // LDS row stride (elements): odd word count, no bank conflicts
  int nci = j.Cin - ci0;
  nci = nci < 0 ? 0 : (nci > PK_CI ? PK_CI : nci);
  const int rowlen = PK_CI * KK;
  // full tiles of 16-byte aligned rows (every tile of the YOLOX / ResNet / transformer weights but the ragged edges): a wave
  // takes whole output-channel rows, a lane 4 consecutive floats per trip - one 16-byte load and no integer division per
  // element (the scalar loop below spent ~25 instructions per 4-byte load: 44 us for the 36 MB of YOLOX-s masters)
  const float* const wrow0 = j.w + ((size_t)co0 * j.Cin + ci0) * KK;
  const bool fast = nci == PK_CI && co0 + PK_CO <= j.Cout && ((j.Cin * KK) & 3) == 0 && (((uintptr_t)wrow0) & 15) == 0 &&
                    (rowlen & 3) == 0;
  if (fast) {
    const int wave = tid >> 6, lane = tid & 63;
    for (int co_l = wave; co_l < PK_CO; co_l += 4) {
      const float4* src = (const float4*)(wrow0 + (size_t)co_l * j.Cin * KK);
      const float sc = j.scale ? j.scale[co0 + co_l] : 1.f;      // (x * 1.f is exact: one code path)
      unsigned* dst = (unsigned*)(pk_s + co_l * rs);             // rs even: 4-byte aligned rows
      for (int q4 = lane; q4 < rowlen / 4; q4 += 64) {
        const float4 v = src[q4];
        const __bf16 b0 = (__bf16)(v.x * sc), b1 = (__bf16)(v.y * sc), b2 = (__bf16)(v.z * sc), b3 = (__bf16)(v.w * sc);
        dst[2 * q4] = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
        dst[2 * q4 + 1] = (unsigned)__builtin_bit_cast(unsigned short, b2) | ((unsigned)__builtin_bit_cast(unsigned short, b3) << 16);
      }
    }
  } else
  for (int idx = tid; idx < PK_CO * rowlen; idx += 256) {
    const int co_l = idx / rowlen, q = idx - co_l * rowlen;
    const int row = co0 + co_l;
    float v = 0.f;
    if (row < j.Cout && q < nci * KK) {
      v = j.w[((size_t)row * j.Cin + ci0) * KK + q];
      if (j.scale) v *= j.scale[row];                      // (fp32 product, then the bf16 rounding: pack_w_body's order)
    }
    pk_s[co_l * rs + q] = (__bf16)v;
  }
  __syncthreads();
  const int K8f = j.CinPad / 8, K8d = j.CoutPadK / 8;
  if (j.wf) {
    for (int it = tid; it < KK * 256; it += 256) {
      const int tap = it >> 8, k8_l = (it >> 5) & 7, co_l = it & 31;
      const int k8 = ci0 / 8 + k8_l, co = co0 + co_l;
      if (k8 >= K8f || co >= j.CoutPad) continue;
      unsigned short h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = __builtin_bit_cast(unsigned short, pk_s[co_l * rs + (k8_l * 8 + e) * KK + tap]);
      ((uint4*)j.wf)[((size_t)tap * K8f + k8) * j.CoutPad + co] =
          make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16),
                     h[6] | ((uint32_t)h[7] << 16));
    }
  }
  if (j.wd) {
    for (int it = tid; it < KK * 256; it += 256) {
      const int tap = it >> 8, co8_l = (it >> 6) & 3, ci_l = it & 63;
      const int co8 = co0 / 8 + co8_l, ci = ci0 + ci_l;
      if (co8 >= K8d || ci >= j.CinPadN) continue;
      unsigned short h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = __builtin_bit_cast(unsigned short, pk_s[(co8_l * 8 + e) * rs + ci_l * KK + tap]);
      ((uint4*)j.wd)[((size_t)tap * K8d + co8) * j.CinPadN + ci] =
          make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16),
                     h[6] | ((uint32_t)h[7] << 16));
    }
  }
}
static int pack_tiles(const mi_pack_job& j) {
  const int coP = j.wd && j.CoutPadK > j.CoutPad ? j.CoutPadK : (j.wf ? j.CoutPad : j.CoutPadK);
  const int ciP = j.wd && j.CinPadN > j.CinPad ? j.CinPadN : (j.wf ? j.CinPad : j.CinPadN);
  return ((coP + PK_CO - 1) / PK_CO) * ((ciP + PK_CI - 1) / PK_CI);
}
extern "C" int mi_pack_jobs_layout(mi_pack_job* jobs_host, int njobs) {
  MI_REQUIRE(jobs_host && njobs > 0, "pack_jobs_layout: args");
  int64_t blk = 0;
  for (int k = 0; k < njobs; ++k) {
    mi_pack_job& j = jobs_host[k];
    MI_REQUIRE(j.w && (j.wf || j.wd) && j.KK >= 1 && j.KK <= MI_MAX_TAPS, "pack_jobs_layout: job %d: null / taps", k);
    if (j.wf) MI_REQUIRE(j.CinPad % 8 == 0 && j.CinPad >= j.Cin && j.CoutPad >= j.Cout, "pack_jobs_layout: job %d: fwd pads", k);
    if (j.wd) MI_REQUIRE(j.CoutPadK % 8 == 0 && j.CoutPadK >= j.Cout && j.CinPadN >= j.Cin, "pack_jobs_layout: job %d: dgrad pads", k);
    j.blk0 = (int32_t)blk;
    blk += pack_tiles(j);
    MI_REQUIRE(blk < (1LL << 30), "pack_jobs_layout: too many blocks");
  }
  return (int)blk;
}
extern "C" int mi_pack_conv_weights_batch(const mi_pack_job* jobs_dev, int njobs, int total_blocks, int kk_max,
                                          mi_stream_t st) {
  MI_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && kk_max >= 1 && kk_max <= MI_MAX_TAPS, "pack_w_batch: args");
  const size_t lds = (size_t)PK_CO * (PK_CI * kk_max + 2) * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)pack_w_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(pack_w_batch_kernel, dim3(total_blocks), dim3(256), lds, (hipStream_t)st, jobs_dev, njobs, kk_max);
  MI_CHECK_LAUNCH("pack_w_batch");
  return MI_OK;
}

extern "C" int mi_pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, void* wf, int CinPad,
                                   int CoutPad, void* wd, int CoutPadK, int CinPadN, mi_stream_t st) {
  MI_REQUIRE(w && (wf || wd), "pack_w: null");
  if (wf) MI_REQUIRE(CinPad % 8 == 0 && CinPad >= Cin && CoutPad >= Cout, "pack_w: fwd pads");
  if (wd) MI_REQUIRE(CoutPadK % 8 == 0 && CoutPadK >= Cout && CinPadN >= Cin, "pack_w: dgrad pads");
  const int KK = KH * KW;
  const int64_t n = (wf ? (int64_t)KK * CinPad * CoutPad : 0) + (wd ? (int64_t)KK * CoutPadK * CinPadN : 0);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_w_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, w, Cout, Cin, KK, (__bf16*)wf,
                     CinPad, CoutPad, (__bf16*)wd, CoutPadK, CinPadN);
  MI_CHECK_LAUNCH("pack_w");
  return MI_OK;
}

// the same with a per-output-channel factor folded in (detectron2 Conv2d + FrozenBatchNorm2d: W * scale[co])
extern "C" int mi_pack_conv_weight_scaled(const float* w, const float* cout_scale, int Cout, int Cin, int KH, int KW, void* wf, int CinPad,
                                   int CoutPad, void* wd, int CoutPadK, int CinPadN, mi_stream_t st) {
  MI_REQUIRE(w && cout_scale && (wf || wd), "pack_w_scaled: null");
  if (wf) MI_REQUIRE(CinPad % 8 == 0 && CinPad >= Cin && CoutPad >= Cout, "pack_w: fwd pads");
  if (wd) MI_REQUIRE(CoutPadK % 8 == 0 && CoutPadK >= Cout && CinPadN >= Cin, "pack_w: dgrad pads");
  const int KK = KH * KW;
  const int64_t n = (wf ? (int64_t)KK * CinPad * CoutPad : 0) + (wd ? (int64_t)KK * CoutPadK * CinPadN : 0);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_w_scaled_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, w, cout_scale, Cout, Cin, KK, (__bf16*)wf,
                     CinPad, CoutPad, (__bf16*)wd, CoutPadK, CinPadN);
  MI_CHECK_LAUNCH("pack_w_scaled");
  return MI_OK;
}
