// Host side of the streaming 1x1 convolution (conv1x1_stream.h): eligibility, slice table, launch configuration.
// mi_conv2d / mi_conv2d_group_plan route eligible descriptors here (MI_CONV_STREAM=0 keeps everything on the tile kernel);
// mi_conv1x1_stream is the explicit entry (it fails instead of falling back).
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "conv1x1_stream.h"

int c1s_launch_mode0(const C1Launch& l, hipStream_t s);
int c1s_launch_mode1(const C1Launch& l, hipStream_t s);
int c1s_launch_mode1x(const C1Launch& l, hipStream_t s);
int c1s_launch_mode2(const C1Launch& l, hipStream_t s);
int c1s_launch_mode3(const C1Launch& l, hipStream_t s);
int c1s_launch_mode4(const C1Launch& l, hipStream_t s);
int c1s_launch_mode5(const C1Launch& l, hipStream_t s);
int c1s_launch_mode6(const C1Launch& l, hipStream_t s);
int c1s_bar_status(unsigned* flag);

static int c1s_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// what the kernel cannot do stays on the tile kernel: taps, strides, fp32 outputs, ragged channel counts
static bool c1s_desc_ok(const mi_conv_desc* d) {
  const int K = d->K8 * 8;
  if (d->ntaps != 1 || d->tap_dy[0] != 0 || d->tap_dx[0] != 0) return false;
  if (d->in_stride != 1 || d->out_stride != 1 || d->out_oy || d->out_ox) return false;
  if (d->gridH != d->outH || d->gridW != d->outW || d->outH != d->H || d->outW != d->W) return false;
  // round 6: bias / ReLU (MODE 4) and the aux-tensor epilogues (MODE 5) - MI_CONV_STREAM_EPI=0 keeps those on the tile kernel
  static const int epi_ok = getenv("MI_CONV_STREAM_EPI") ? atoi(getenv("MI_CONV_STREAM_EPI")) : 1;
  const int epf = MI_CONV_RELU | MI_CONV_ADDRELU | MI_CONV_RELUMASK;
  if (d->flags & ~(MI_CONV_ACCUM | epf)) return false;
  const bool accmask = (d->flags & (MI_CONV_ACCUM | epf)) == (MI_CONV_ACCUM | MI_CONV_RELUMASK) && !d->bias;   // MODE 6
  if ((d->flags & epf) || d->bias) {
    if (!epi_ok || ((d->flags & MI_CONV_ACCUM) && !accmask) || d->stats_acc || d->xf) return false;
    const int aux = d->flags & (MI_CONV_ADDRELU | MI_CONV_RELUMASK);
    if (aux == (MI_CONV_ADDRELU | MI_CONV_RELUMASK) || (aux && (d->flags & MI_CONV_RELU))) return false;
    if (aux && (!d->bn_y || d->bn_ldy % 8 || ((uintptr_t)d->bn_y & 15) ||
                (long long)d->N * d->H * d->W * d->bn_ldy * 2 >= (1LL << 31))) return false;
  }
  if (!(K == 32 || K == 64 || K == 128 || K == 256 || K == 512)) return false;
  if (d->Cout != d->CoutPad || d->Cout % 32) return false;
  if (d->ldx % 8 || d->ldy % 8 || ((uintptr_t)d->x & 15) || ((uintptr_t)d->y & 15) || ((uintptr_t)d->w & 15)) return false;
  const long long npix = (long long)d->N * d->H * d->W;
  if (d->y_nstride && (long long)d->y_nstride != (long long)d->outH * d->outW * d->ldy) return false;
  if (npix * d->ldx * 2 >= (1LL << 31) || npix * d->ldy * 2 >= (1LL << 31)) return false;
  if ((d->flags & MI_CONV_ACCUM) && d->stats_acc) return false;
  return true;
}

// the BatchNorm pass that follows convolution d, as the kernel's phase 2; false when the job does not describe exactly that
bool cbn_from_job(const mi_conv_desc& d, const mi_bn_job& j, CBnFwd* o) {
  const long long npix = (long long)d.N * d.outH * d.outW;
  if (j.y != d.y || j.ldy != d.ldy || j.C != d.Cout || j.acc != d.stats_acc || !j.acc || !j.a) return false;
  if (j.npix != npix || j.count != npix) return false;
  const int nsl = (d.stats_slots >= 1 && d.stats_slots <= MI_BN_SLOTS) ? d.stats_slots : MI_BN_SLOTS;
  if (j.nslots != nsl) return false;
  if (!j.gamma || !j.beta || !j.scale || !j.shift || !j.mean || !j.invstd) return false;
  if (j.lda % 8 || ((uintptr_t)j.a & 15) || npix * j.lda * 2 >= (1LL << 31)) return false;
  if (j.res && (j.ldres % 8 || ((uintptr_t)j.res & 15) || npix * j.ldres * 2 >= (1LL << 31))) return false;
  memset(o, 0, sizeof(*o));
  o->res = (const __bf16*)j.res; o->a = (__bf16*)j.a;
  o->gamma = j.gamma; o->beta = j.beta; o->rmean = j.rmean; o->rvar = j.rvar; o->nbt = (long long*)j.nbt;
  o->scale = j.scale; o->shift = j.shift; o->mean = j.mean; o->invstd = j.invstd;
  o->ldres = j.ldres; o->lda = j.lda; o->act = j.act;
  o->inv_count = 1.0 / (double)j.count;                                    // (as mi_bn_act_fwd derives them)
  o->unbias = j.count > 1 ? (double)j.count / (double)(j.count - 1) : 1.0;
  o->eps = j.eps; o->momentum = j.momentum;
  return true;
}

// n descriptors that read the same tensor -> one launch; returns false when the stream kernel does not apply.
// bn (may be NULL): the BatchNorm jobs of the n convolutions -> MODE 3
// c0 / cn (single descriptor only, cn > 0): the launch computes output channels [c0, c0 + cn) of the convolution - a layer
// with more than 512 output channels runs as several launches over the same input (c1s_try_launch)
static bool c1s_fill(const mi_conv_desc* ds, int n, C1Launch* l, const mi_bn_job* bn = nullptr, int c0 = 0, int cn = 0) {
  if (n < 1) return false;
  if (bn && n > C1_MAX_BN) return false;
  if (cn > 0 && (n != 1 || bn || cn % 32 || c0 % 32 || c0 + cn > ds[0].Cout)) return false;
  const mi_conv_desc& d0 = ds[0];
  int ns = 0;
  for (int j = 0; j < n; ++j) {
    const mi_conv_desc& d = ds[j];
    if (!c1s_desc_ok(&d)) return false;
    if (d.x != d0.x || d.ldx != d0.ldx || d.N != d0.N || d.H != d0.H || d.W != d0.W || d.K8 != d0.K8) return false;
    if (d.flags != d0.flags || (d.stats_acc != nullptr) != (d0.stats_acc != nullptr)) return false;
    if (d.xf != d0.xf || (d.xf && (d.xf_C != d.K8 * 8 || ((uintptr_t)d.xf & 7)))) return false;   // one input tensor, one record
    ns += (cn > 0 ? cn : d.Cout) / 32;
  }
  if (ns > C1_MAX_SLICES) return false;
  const int K = d0.K8 * 8;
  const long long npix = (long long)d0.N * d0.H * d0.W;
  const int WM = ns % 4 == 0 ? 4 : (ns % 2 == 0 ? 2 : 1);
  const int nco = ns / WM;
  // pixel tile: 128 (WM 4 / 2), 256 (WM 1); WM = 4 falls to 64 pixels when the ring would not fit LDS, when the map is
  // small (the chip wants >= ~2 tiles per CU in flight) or when the pixel count asks for it
  int PT = 1, tpix = WM == 4 ? 128 : (WM == 2 ? 128 : 256);
  if (WM == 4) {
    PT = 2;
    const bool fits = c1s_valid(K, 128), small = (npix / 128) * nco < 2 * c1s_cus();
    if (!fits || npix % 128 || (small && c1s_valid(K, 64))) { PT = 1; tpix = 64; }
  }
  if (!c1s_valid(K, tpix)) return false;
  memset(l, 0, sizeof(*l));
  l->K = K; l->WM = WM; l->PT = PT; l->NBUF = c1s_nbuf(K, tpix);
  l->MODE = (d0.flags & MI_CONV_ACCUM) ? 2 : (d0.stats_acc ? 1 : 0);
  bool any_bias = false;
  for (int j = 0; j < n; ++j) any_bias = any_bias || ds[j].bias != nullptr;
  if ((d0.flags & (MI_CONV_ACCUM | MI_CONV_RELUMASK)) == (MI_CONV_ACCUM | MI_CONV_RELUMASK)) l->MODE = 6;
  else if (d0.flags & (MI_CONV_ADDRELU | MI_CONV_RELUMASK)) l->MODE = 5;
  else if (any_bias || (d0.flags & MI_CONV_RELU)) l->MODE = 4;
  // a partial last pixel tile (round 6; the modes without statistics): maps whose pixel count is no multiple of the tile -
  // detectron2's ResNet at 800 x 1333 (res4: 4 x 50 x 84 = 16 800 pixels).  Small ragged maps stay on the tile kernel
  // (MI_C1S_RAGGED_MINPIX, default 8192: the transformer's token-row GEMMs, T = 4 368, measured no faster here)
  bool ragged = false;
  if (npix % tpix) {
    static const long minpix = getenv("MI_C1S_RAGGED_MINPIX") ? atol(getenv("MI_C1S_RAGGED_MINPIX")) : 8192;
    const bool mode_ok = l->MODE == 0 || l->MODE == 2 || l->MODE >= 4;
    if (!mode_ok || bn || d0.xf || minpix < 0 || npix < minpix || npix < tpix) return false;
    ragged = true;
  }
  if (bn) {
    if (l->MODE != 1) return false;
    l->MODE = 3;
  }
  // BatchNorm + activation of the input inside this launch (conv_bn.h, BnXf): forward-with-statistics launches only
  const bool xf = d0.xf != nullptr;
  if (xf && (l->MODE != 1 || bn)) return false;
  l->XF = xf ? 1 : 0;
  l->lds = l->NBUF * tpix * K * 2 + (xf ? 2 * K * 4 : 0);
  if (l->lds > 160 * 1024) return false;
  C1K& k = l->k;
  k.x = (const __bf16*)d0.x;
  k.ldx = d0.ldx;
  k.xf = (const BnXf*)d0.xf;
  k.relu = (d0.flags & MI_CONV_RELU) ? 1 : 0;
  k.epi = (d0.flags & MI_CONV_ADDRELU) ? 1 : ((d0.flags & MI_CONV_RELUMASK) ? 2 : 0);
  for (int j = 0; j < n; ++j) k.xfw |= (xf && ds[j].xf_write) ? 1 : 0;
  k.ntiles = (int)((npix + tpix - 1) / tpix);
  k.npix = (int)npix;
  k.ragged = ragged ? 1 : 0;
  k.nco = nco;
  // persistent grid: as many blocks as are resident at once (LDS-limited, at most 2 per CU), never more than tiles
  int per_cu = (160 * 1024) / l->lds;
  if (per_cu > 1) per_cu = 1;   // (measured: two blocks per CU lose 0.5 % of the step - twice the statistics atomics, no gain in bandwidth)
  if (per_cu < 1) per_cu = 1;
  static const int ovr = getenv("MI_C1S_PERCU") ? atoi(getenv("MI_C1S_PERCU")) : 0;
  if (ovr > 0 && !bn) per_cu = ovr;   // (MODE 3: the grid barrier needs every block resident - one per CU always is)
  int nb = (c1s_cus() * per_cu) / nco;
  if (nb > k.ntiles) nb = k.ntiles;
  if (nb < 1) nb = 1;
  if (nco > 1 && nb >= 8) nb &= ~7;   // cot * nb + b: the cout tiles of a pixel tile meet in one XCD's L2
  k.nb = nb;
  k.xcd_order = (nco > 1 && nb % 8 == 0) ? 1 : 0;
  l->grid = nb * nco;
  static const int dbg = getenv("MI_DEBUG_NOATOM") ? atoi(getenv("MI_DEBUG_NOATOM")) : 0;
  k.dbg = dbg;
  int si = 0;
  for (int j = 0; j < n; ++j) {
    const mi_conv_desc& d = ds[j];
    const int nsl = (d.stats_slots >= 1 && d.stats_slots <= MI_BN_SLOTS) ? d.stats_slots : MI_BN_SLOTS;
    for (int c = c0; c < (cn > 0 ? c0 + cn : d.Cout); c += 32, ++si) {
      C1Slice& s = k.s[si];
      s.w = (const u32x4*)d.w + c;
      s.wld = d.CoutPad;
      s.y = (__bf16*)d.y + c;
      s.ldy = d.ldy;
      s.stats = d.stats_acc ? d.stats_acc + (size_t)c * 2 : nullptr;
      s.sld = d.CoutPad * 2;
      s.nslots = nsl;
      s.bnj = j;
      s.c0 = c;
      s.bias = d.bias ? d.bias + c : nullptr;
      s.aux = (d.flags & (MI_CONV_ADDRELU | MI_CONV_RELUMASK)) ? (const __bf16*)d.bn_y + c : nullptr;
      s.ldaux = d.bn_ldy;
    }
    if (bn && !cbn_from_job(d, bn[j], &k.bn[j])) return false;
  }
  return true;
}

static int c1s_run(const C1Launch& l, hipStream_t s) {
  switch (l.MODE) {
    case 0: return c1s_launch_mode0(l, s);
    case 1: return l.XF ? c1s_launch_mode1x(l, s) : c1s_launch_mode1(l, s);
    case 3: return c1s_launch_mode3(l, s);
    case 4: return c1s_launch_mode4(l, s);
    case 5: return c1s_launch_mode5(l, s);
    case 6: return c1s_launch_mode6(l, s);
    default: return c1s_launch_mode2(l, s);
  }
}

// 0: off, 1: on (default).  Read per call: the tests flip it inside one process.
static bool c1s_enabled() {
  const char* e = getenv("MI_CONV_STREAM");
  return !e || atoi(e) != 0;
}

// internal entry points for conv_igemm.hip
static bool c1s_launch_any(const mi_conv_desc* ds, int n, hipStream_t s, int* rc) {
  C1Launch l;
  if (n == 1 && ds[0].Cout > 32 * C1_MAX_SLICES && ds[0].Cout % (32 * C1_MAX_SLICES) == 0 && !ds[0].stats_acc && !ds[0].xf) {
    // more output channels than one launch's slice table holds (ResNet conv3 / shortcut: 1024, 2048): 512 at a time, each
    // launch streaming the (4 - 16 x smaller) input again.  All or nothing: the first chunk decides
    const int step = 32 * C1_MAX_SLICES;
    if (!c1s_fill(ds, 1, &l, nullptr, 0, step)) return false;
    *rc = c1s_run(l, s);
    for (int c = step; c < ds[0].Cout && *rc == MI_OK; c += step) {
      if (!c1s_fill(ds, 1, &l, nullptr, c, step)) { *rc = MI_EINVAL; break; }
      *rc = c1s_run(l, s);
    }
    return true;
  }
  if (!c1s_fill(ds, n, &l)) return false;
  *rc = c1s_run(l, s);
  return true;
}
bool c1s_try_launch(const mi_conv_desc* ds, int n, hipStream_t s, int* rc) { return c1s_enabled() && c1s_launch_any(ds, n, s, rc); }
bool c1s_try_plan(const mi_conv_desc* ds, int n, C1Launch* l) { return c1s_enabled() && c1s_fill(ds, n, l); }
bool c1s_try_plan_bn(const mi_conv_desc* ds, const mi_bn_job* bn, int n, C1Launch* l) { return c1s_enabled() && c1s_fill(ds, n, l, bn); }
int c1s_barrier_status(unsigned* flag) { return c1s_bar_status(flag); }
int c1s_run_planned(const C1Launch* l, hipStream_t s) { return c1s_run(*l, s); }

extern "C" int mi_conv1x1_stream(const mi_conv_desc* descs, int n, mi_stream_t st) {
  MI_REQUIRE(descs && n >= 1, "conv1x1_stream: null");
  int rc = MI_OK;
  const bool ok = c1s_launch_any(descs, n, (hipStream_t)st, &rc);      // (the explicit entry does not honour the routing switch)
  MI_REQUIRE(ok,
             "conv1x1_stream: needs 1x1 stride-1 bf16 convs of one input with K in {32..512}, Cout %% 32 == 0 (at most 512 together, "
             "or one convolution with a multiple of 512), statistics / BatchNorm modes with N*H*W a multiple of the pixel tile");
  return rc;
}
