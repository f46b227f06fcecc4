// Host side of the weight-stationary 3x3 convolution (conv3x3_ws.h): eligibility, job table, block partition.
// mi_conv2d / mi_conv2d_group_plan route eligible descriptors here (MI_CONV_WS=0 keeps them on the tile kernel);
// mi_conv3x3_ws is the explicit entry (it fails instead of falling back).
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "conv3x3_ws.h"

int w3_launch_mode0(const W3Launch& l, hipStream_t s);
int w3_launch_mode1(const W3Launch& l, hipStream_t s);
int w3_launch_mode1x(const W3Launch& l, hipStream_t s);
int w3_launch_mode2(const W3Launch& l, hipStream_t s);
int w3_launch_mode3(const W3Launch& l, hipStream_t s);
int w3_mode3_blocks_per_cu(int K, int lds);
int w3_bar_status(unsigned* flag);
bool cbn_from_job(const mi_conv_desc& d, const mi_bn_job& j, CBnFwd* o);   // conv1x1_stream.hip

static int w3_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// 3x3, K in {32, 64, 128} input channels, bf16 in / out, no bias; stride 1: K -> K; stride 2 (forward only): K -> a multiple
// of the block's output channels (64 for K = 32, else 128), pad 1.  tw[] = weight slab per tap position
static bool w3_desc_ok(const mi_conv_desc* d, int* tw) {
  const int K = d->K8 * 8;
  if (d->ntaps != 9 || (d->in_stride != 1 && d->in_stride != 2) || d->out_stride != 1 || d->out_oy || d->out_ox) return false;
  if (d->gridH != d->outH || d->gridW != d->outW) return false;
  if (d->flags & ~MI_CONV_ACCUM) return false;
  if (d->bias) return false;
  if (!(K == 32 || K == 64 || K == 128) || d->Cout != d->CoutPad) return false;
  if (d->in_stride == 1) {
    if (d->outH != d->H || d->outW != d->W || d->Cout != K) return false;
  } else {
    const char* s2 = getenv("MI_CONV_WS_S2");     // (read per call: tests compare the two kernels in one process)
    if ((s2 && atoi(s2) == 0) || (d->flags & MI_CONV_ACCUM)) return false;
    if (d->outH != (d->H - 1) / 2 + 1 || d->outW != (d->W - 1) / 2 + 1) return false;
    const int bc = w3_block_cout(2, K);
    if (d->Cout % bc || d->Cout / bc > W3_MAX_JOBS) return false;
  }
  if (d->ldx % 8 || d->ldy % 8 || ((uintptr_t)d->x & 15) || ((uintptr_t)d->y & 15) || ((uintptr_t)d->w & 15)) return false;
  if (d->y_nstride && (long long)d->y_nstride != (long long)d->outH * d->outW * d->ldy) return false;
  if ((long long)d->H * d->W * d->ldx * 2 >= (1LL << 31)) return false;   // per-lane 32-bit offsets inside one image
  if ((d->flags & MI_CONV_ACCUM) && d->stats_acc) return false;
  int seen = 0;
  for (int t = 0; t < 9; ++t) {
    const int dy = d->tap_dy[t], dx = d->tap_dx[t];
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
    const int pos = (dy + 1) * 3 + dx + 1;
    if (seen & (1 << pos)) return false;
    seen |= 1 << pos;
    tw[pos] = d->tap_w[t];
  }
  return seen == 0x1ff;
}

// bn (may be NULL): the BatchNorm jobs of the n convolutions -> MODE 3
static bool w3_fill(const mi_conv_desc* ds, int n, W3Launch* l, const mi_bn_job* bn = nullptr) {
  if (n < 1 || n > W3_MAX_JOBS) return false;
  memset(l, 0, sizeof(*l));
  const int K = ds[0].K8 * 8, S = ds[0].in_stride;
  if (bn && ((ds[0].flags & MI_CONV_ACCUM) || !ds[0].stats_acc || S != 1)) return false;
  const int mode = (ds[0].flags & MI_CONV_ACCUM) ? 2 : (ds[0].stats_acc ? (bn ? 3 : 1) : 0);
  // BatchNorm + activation of the INPUT inside this launch (conv_bn.h, BnXf): forward-with-statistics launches only, every
  // job carries its record (jobs over one input tensor share it; exactly one of them has xf_write)
  const bool xf = ds[0].xf != nullptr;
  if (xf && mode != 1) return false;
  const int TH = w3_tile_h(S, K);
  const int hrows = ((S == 1 ? (TH + 2) * (W3_TW + 2) : (2 * TH + 1) * (2 * W3_TW + 1)) + 63) / 64 * 64;
  const int lds = 2 * (K / 8) * hrows * 16 + ((mode == 1 || mode == 3) ? 256 * 32 * 4 : 0) + (xf ? 2 * K * 4 : 0);   // halo ring (+ per-lane BatchNorm sums) (+ input scale / shift)
  long long total = 0;
  int nj = 0;
  for (int j = 0; j < n; ++j) {
    const mi_conv_desc& d = ds[j];
    int tw[9];
    if (!w3_desc_ok(&d, tw) || d.K8 * 8 != K || d.in_stride != S) return false;
    if ((d.flags & MI_CONV_ACCUM) != (ds[0].flags & MI_CONV_ACCUM) || (d.stats_acc != nullptr) != (ds[0].stats_acc != nullptr)) return false;
    if ((d.xf != nullptr) != xf || (xf && (d.xf_C != K || ((uintptr_t)d.xf & 7)))) return false;
    // a block computes w3_block_cout output channels: a wider convolution (stride 2: K -> 2 K) is that many jobs over one input
    const int bc = w3_block_cout(S, K);
    for (int c0 = 0; c0 < d.Cout; c0 += bc, ++nj) {
      if (nj >= W3_MAX_JOBS) return false;
      W3Job& jb = l->k.j[nj];
      memcpy(jb.tw, tw, sizeof(tw));
      jb.x = (const __bf16*)d.x; jb.w = (const u32x4*)d.w + c0; jb.y = (__bf16*)d.y + c0;
      jb.stats = d.stats_acc ? d.stats_acc + (size_t)c0 * 2 : nullptr;
      jb.ldx = d.ldx; jb.ldy = d.ldy; jb.N = d.N; jb.H = d.outH; jb.W = d.outW; jb.inH = d.H; jb.inW = d.W;
      jb.tilesY = mi_cdiv(d.outH, TH); jb.tilesX = mi_cdiv(d.outW, W3_TW);
      jb.ntiles = d.N * jb.tilesY * jb.tilesX;
      jb.wld = d.CoutPad;
      jb.nslots = (d.stats_slots >= 1 && d.stats_slots <= MI_BN_SLOTS) ? d.stats_slots : MI_BN_SLOTS;
      jb.sld = d.CoutPad * 2;
      total += jb.ntiles;
      if (bn && !cbn_from_job(d, bn[j], &jb.bn)) return false;
      jb.xf = (const BnXf*)d.xf;
      jb.xfw = (xf && d.xf_write && c0 == 0) ? 1 : 0;
    }
  }
  n = nj;
  // persistent blocks: K = 128 one per CU (288 VGPRs of weights per wave), K = 64 two, K = 32 four; shared between
  // the jobs in proportion to their tiles (every job gets at least one block, none more blocks than tiles)
  static const int ovr = getenv("MI_W3_PERCU") ? atoi(getenv("MI_W3_PERCU")) : 0;
  int per_cu = ovr > 0 ? ovr : (K == 128 ? 1 : (K == 64 ? 2 : 4));
  if (S == 2 && ovr <= 0) per_cu = (160 * 1024) / lds >= 2 && K != 128 ? 2 : 1;    // (the stride-2 halo ring is 41 - 98 KB)
  if (mode == 3) {   // the grid barrier needs every block resident
    static int occ[3] = {-1, -1, -1};
    int& o = occ[K == 128 ? 0 : (K == 64 ? 1 : 2)];
    if (o < 0) o = w3_mode3_blocks_per_cu(K, lds);
    if (o < 1) return false;
    if (per_cu > o) per_cu = o;
  }
  long long cap = (long long)w3_cus() * per_cu;
  if (cap > total) cap = total;
  int used = 0;
  for (int j = 0; j < n; ++j) {
    W3Job& jb = l->k.j[j];
    long long nb = (cap * jb.ntiles) / total;
    if (nb < 1) nb = 1;
    if (nb > jb.ntiles) nb = jb.ntiles;
    jb.nblk = (int)nb;
    used += jb.nblk;
  }
  // leftover blocks go to the jobs with the most tiles per block
  while (used < cap) {
    int best = -1;
    double br = 0;
    for (int j = 0; j < n; ++j) {
      const W3Job& jb = l->k.j[j];
      const double r = (double)jb.ntiles / jb.nblk;
      if (jb.nblk < jb.ntiles && r > br) { br = r; best = j; }
    }
    if (best < 0) break;
    l->k.j[best].nblk++;
    used++;
  }
  int blk = 0;
  for (int j = 0; j < n; ++j) {
    l->k.j[j].blk0 = blk;
    blk += l->k.j[j].nblk;
  }
  l->k.njobs = n;
  // (timing experiments; read per call.  1: no statistics atomics, 2: no halo traffic after the first tile, 4: no main loop)
  l->k.dbg = (getenv("MI_DEBUG_NOATOM") ? atoi(getenv("MI_DEBUG_NOATOM")) & 1 : 0) | (getenv("MI_W3_DBG") ? atoi(getenv("MI_W3_DBG")) & 6 : 0);
  l->K = K;
  l->S = S;
  l->MODE = mode;
  l->XF = xf ? 1 : 0;
  l->grid = blk;
  l->lds = lds;
  return true;
}

static int w3_run(const W3Launch& l, hipStream_t s) {
  switch (l.MODE) {
    case 0: return w3_launch_mode0(l, s);
    case 1: return l.XF ? w3_launch_mode1x(l, s) : w3_launch_mode1(l, s);
    case 3: return w3_launch_mode3(l, s);
    default: return w3_launch_mode2(l, s);
  }
}

static bool w3_enabled() {
  const char* e = getenv("MI_CONV_WS");
  return !e || atoi(e) != 0;
}

bool w3_try_launch(const mi_conv_desc* ds, int n, hipStream_t s, int* rc) {
  if (!w3_enabled()) return false;
  W3Launch l;
  if (!w3_fill(ds, n, &l)) return false;
  *rc = w3_run(l, s);
  return true;
}
bool w3_try_plan(const mi_conv_desc* ds, int n, W3Launch* l) { return w3_enabled() && w3_fill(ds, n, l); }
bool w3_try_plan_bn(const mi_conv_desc* ds, const mi_bn_job* bn, int n, W3Launch* l) { return w3_enabled() && w3_fill(ds, n, l, bn); }
int w3_barrier_status(unsigned* flag) { return w3_bar_status(flag); }
int w3_run_planned(const W3Launch* l, hipStream_t s) { return w3_run(*l, s); }

extern "C" int mi_conv3x3_ws(const mi_conv_desc* descs, int n, mi_stream_t st) {
  MI_REQUIRE(descs && n >= 1, "conv3x3_ws: null");
  W3Launch l;
  MI_REQUIRE(w3_fill(descs, n, &l),
             "conv3x3_ws: needs 1..%d bf16 3x3 stride-1 convs with K == Cout in {32, 64, 128} (all the same K), no bias, "
             "all plain / all with stats_acc / all MI_CONV_ACCUM", W3_MAX_JOBS);
  return w3_run(l, (hipStream_t)st);
}
