// Implicit-GEMM convolution on the gfx950 matrix cores, im2col-free.
//
// One kernel template covers 1x1 / 3x3-s1 / 3x3-s2 forward and their data gradients
// through a tap table (see include/mi355_det.h, mi_conv_desc).  Replaces the ATen/cuDNN
// conv2d the reference reaches from BaseConv (yolov7/modeling/backbone/layers/wrappers.py:60-83)
// and the prediction convs of YOLOXHead (yolov7/modeling/head/yolox_head.py:103-129).
//
// Design (MI355X): block = 4 waves, output tile = 128 pixels (TH x TW) x BN output channels.
//   * the input halo tile for one k-chunk (KC channels) is staged ONCE into LDS in a k8-major
//     image [KC/8][halo pixels][8 ch] (16-byte units) and reused by all taps: the 3x3 conv
//     reads each input element from HBM/L2 ~1.4x, never 9x, and nothing is materialised.
//   * weights are pre-packed [tap][K/8][CoutPad][8] so a (tap, k-chunk) slab is a run of
//     contiguous 16-byte rows; slabs are register-prefetched and double-buffered in LDS.
//   * v_mfma_f32_32x32x16_bf16 with A = weights (M = cout), B = pixels (N = pixel): each
//     lane then owns 4 consecutive couts x 4 groups of ONE pixel -> 8-byte NHWC stores.
//   * epilogue optionally emits per-tile per-channel (sum, sumsq) from the fp32 accumulators:
//     the BatchNorm batch statistics cost no extra pass over the conv output.
#include "common.h"

struct ConvK {
  const __bf16* x;
  const u32x4* w;
  void* y;
  const float* bias;
  float* stats;
  int ldx, ldy, N, H, W, outH, outW, gridH, gridW, is, os, ooy, oox, K8, Cout, CoutPad, ntaps;
  long long ynstride;
  int tdy[MI_MAX_TAPS], tdx[MI_MAX_TAPS], tw[MI_MAX_TAPS];
  int flags, TH, TW, tilesY, tilesX, nco;
  int dymin, dxmin, haloH, haloW, npixh;
};

template <int KC, int BN, int WM, int WN, int CT, int PT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvK p) {
  static_assert(WM * WN == 4, "4 waves");
  static_assert(WM * CT * 32 == BN, "cout tiling");
  static_assert(WN * PT * 32 == 128, "pixel tiling");
  constexpr int KC8 = KC / 8;
  constexpr int KS = KC / 16;
  constexpr int WCH = KC8 * BN;           // 16-byte rows per weight slab
  constexpr int WPT = (WCH + 255) / 256;  // rows per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* Xs = (u32x4*)smem;               // [KC8][npixh]
  u32x4* Ws = Xs + KC8 * p.npixh;         // [2][KC8][BN]
  float* Ss = (float*)(Ws + 2 * WCH);     // [WN][BN][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const int cot = blockIdx.x % p.nco;
  const int tile = blockIdx.x / p.nco;
  const int tpi = p.tilesY * p.tilesX;
  const int img = tile / tpi;
  const int rem = tile - img * tpi;
  const int ty0 = (rem / p.tilesX) * p.TH, tx0 = (rem % p.tilesX) * p.TW;
  const int co0 = cot * BN;
  const int iy0 = ty0 * p.is + p.dymin, ix0 = tx0 * p.is + p.dxmin;
  const int TP = p.TH * p.TW;
  const int npixh = p.npixh;

  int pixbase[PT], gy[PT], gx[PT];
  bool pvalid[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int P = (wn * PT + j) * 32 + l31;
    const bool v = P < TP;
    const int ty = v ? P / p.TW : 0;
    const int tx = v ? P - ty * p.TW : 0;
    pixbase[j] = ty * p.is * p.haloW + tx * p.is;
    gy[j] = ty0 + ty;
    gx[j] = tx0 + tx;
    pvalid[j] = v && gy[j] < p.gridH && gx[j] < p.gridW;
  }

  f32x16 acc[CT][PT];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < PT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = p.K8 / KC8;
  const int nsteps = nchunks * p.ntaps;
  u32x4 wreg[WPT];

  auto load_w = [&](int step) {
    const int kc = step / p.ntaps, t = step - kc * p.ntaps;
    const int slab = p.tw[t];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * 256;
      if (idx < WCH) {
        const int c8 = idx / BN, co = idx % BN;
        wreg[i] = p.w[((size_t)(slab * p.K8 + kc * KC8 + c8)) * p.CoutPad + co0 + co];
      }
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * 256;
      if (idx < WCH) Ws[buf * WCH + idx] = wreg[i];
    }
  };

  load_w(0);
  store_w(0);
  for (int step = 0; step < nsteps; ++step) {
    const int kc = step / p.ntaps, t = step - kc * p.ntaps;
    if (t == 0) {
      if (step > 0) __syncthreads();  // every wave finished reading the previous halo slab
      const size_t imgbase = (size_t)img * p.H;
      for (int idx = tid; idx < KC8 * npixh; idx += 256) {
        const int c = idx & (KC8 - 1), hp = idx / KC8;
        const int hy = hp / p.haloW, hx = hp - hy * p.haloW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
          v = *(const u32x4*)(p.x + ((imgbase + iy) * p.W + ix) * (size_t)p.ldx + kc * KC + c * 8);
        Xs[c * npixh + hp] = v;
      }
    }
    if (step + 1 < nsteps) load_w(step + 1);
    __syncthreads();
    const int buf = step & 1;
    const int toff = (p.tdy[t] - p.dymin) * p.haloW + (p.tdx[t] - p.dxmin);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k8 = ks * 2 + h;
      bf16x8 a[CT], b[PT];
#pragma unroll
      for (int i = 0; i < CT; ++i)
        a[i] = __builtin_bit_cast(bf16x8, Ws[buf * WCH + k8 * BN + (wm * CT + i) * 32 + l31]);
#pragma unroll
      for (int j = 0; j < PT; ++j)
        b[j] = __builtin_bit_cast(bf16x8, Xs[k8 * npixh + pixbase[j] + toff]);
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (step + 1 < nsteps) store_w(buf ^ 1);
  }

  // ---- epilogue: D[m = cout][n = pixel]; lane (n = l31, h) holds couts 8q+4h+{0..3}, q=0..3
  const bool do_stats = p.stats != nullptr;
  const bool accum = (p.flags & MI_CONV_ACCUM) != 0;
  const bool outf32 = (p.flags & MI_CONV_OUT_F32) != 0;
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int cbase = co0 + (wm * CT + i) * 32;
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      if (!pvalid[j]) continue;
      const int oy = gy[j] * p.os + p.ooy, ox = gx[j] * p.os + p.oox;
      const size_t po = (size_t)img * (size_t)p.ynstride + ((size_t)oy * p.outW + ox) * (size_t)p.ldy;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = cbase + 8 * q + 4 * h;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < p.Cout) v[e] += p.bias[c + e];
        }
        if (outf32) {
          float* yp = (float*)p.y + po + c;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < p.Cout) {
              if (accum) v[e] += yp[e];
              yp[e] = v[e];
            }
        } else {
          __bf16* yp = (__bf16*)p.y + po + c;
          if (c + 3 < p.Cout) {
            if (accum) {
              const bf16x4 o = *(const bf16x4*)yp;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)o[e];
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            *(bf16x4*)yp = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c + e < p.Cout) {
                if (accum) v[e] += (float)yp[e];
                yp[e] = (__bf16)v[e];
              }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[4 * q + e] += v[e];
          s2[4 * q + e] += v[e] * v[e];
        }
      }
    }
    if (do_stats) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a1 = s1[r], a2 = s2[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          a1 += __shfl_xor(a1, o, 64);
          a2 += __shfl_xor(a2, o, 64);
        }
        if (l31 == 0) {
          const int cl = (wm * CT + i) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
          Ss[(wn * BN + cl) * 2 + 0] = a1;
          Ss[(wn * BN + cl) * 2 + 1] = a2;
        }
      }
    }
  }
  if (do_stats) {
    __syncthreads();
    if (tid < BN) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) {
        a1 += Ss[(w * BN + tid) * 2 + 0];
        a2 += Ss[(w * BN + tid) * 2 + 1];
      }
      float* sp = p.stats + ((size_t)tile * p.CoutPad + co0 + tid) * 2;
      sp[0] = a1;
      sp[1] = a2;
    }
  }
}

// ---------------------------------------------------------------- host side
static void choose_tile(int gridH, int gridW, int* TH, int* TW) {
  const int cands[] = {gridW, 64, 32, 16, 8, 4};
  long best = -1;
  int bh = 8, bw = 16;
  for (int c : cands) {
    if (c <= 0 || c > 128) continue;
    int tw = c;
    int th = 128 / tw;
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

static int conv_fill(const mi_conv_desc* d, ConvK* k, int* KCo, int* BNo, size_t* ldsBytes) {
  MI_REQUIRE(d->x && d->w && d->y, "conv: null pointer");
  MI_REQUIRE(d->ntaps >= 1 && d->ntaps <= MI_MAX_TAPS, "conv: ntaps %d", d->ntaps);
  MI_REQUIRE(d->K8 >= 2 && (d->K8 % 2) == 0, "conv: K8 %d must be even", d->K8);
  MI_REQUIRE(d->CoutPad % 32 == 0 && d->Cout <= d->CoutPad && d->Cout > 0, "conv: Cout %d pad %d",
             d->Cout, d->CoutPad);
  MI_REQUIRE(d->ldx % 8 == 0 && ((uintptr_t)d->x % 16) == 0, "conv: x must be 16B aligned (ldx %d)", d->ldx);
  MI_REQUIRE(d->in_stride == 1 || d->in_stride == 2, "conv: in_stride");
  MI_REQUIRE(d->out_stride == 1 || d->out_stride == 2, "conv: out_stride");
  if (!(d->flags & MI_CONV_OUT_F32))
    MI_REQUIRE((d->Cout % 4 != 0) || (d->ldy % 4 == 0 && ((uintptr_t)d->y % 8) == 0),
               "conv: bf16 y needs 8B alignment (ldy %d)", d->ldy);
  k->x = (const __bf16*)d->x;
  k->w = (const u32x4*)d->w;
  k->y = d->y;
  k->bias = d->bias;
  k->stats = d->stats_partial;
  k->ynstride = d->y_nstride > 0 ? (long long)d->y_nstride : (long long)d->outH * d->outW * d->ldy;
  k->ldx = d->ldx; k->ldy = d->ldy; k->N = d->N; k->H = d->H; k->W = d->W;
  k->outH = d->outH; k->outW = d->outW; k->gridH = d->gridH; k->gridW = d->gridW;
  k->is = d->in_stride; k->os = d->out_stride; k->ooy = d->out_oy; k->oox = d->out_ox;
  k->K8 = d->K8; k->Cout = d->Cout; k->CoutPad = d->CoutPad; k->ntaps = d->ntaps;
  int dymin = 1 << 30, dymax = -(1 << 30), dxmin = 1 << 30, dxmax = -(1 << 30);
  for (int t = 0; t < d->ntaps; ++t) {
    k->tdy[t] = d->tap_dy[t]; k->tdx[t] = d->tap_dx[t]; k->tw[t] = d->tap_w[t];
    if (d->tap_dy[t] < dymin) dymin = d->tap_dy[t];
    if (d->tap_dy[t] > dymax) dymax = d->tap_dy[t];
    if (d->tap_dx[t] < dxmin) dxmin = d->tap_dx[t];
    if (d->tap_dx[t] > dxmax) dxmax = d->tap_dx[t];
  }
  k->flags = d->flags;
  int TH = d->TH, TW = d->TW;
  if (TH <= 0 || TW <= 0) choose_tile(d->gridH, d->gridW, &TH, &TW);
  MI_REQUIRE(TH * TW <= 128 && TH >= 1 && TW >= 1, "conv: tile %dx%d", TH, TW);
  k->TH = TH; k->TW = TW;
  k->tilesY = mi_cdiv(d->gridH, TH); k->tilesX = mi_cdiv(d->gridW, TW);
  k->dymin = dymin; k->dxmin = dxmin;
  k->haloH = (TH - 1) * d->in_stride + (dymax - dymin) + 1;
  k->haloW = (TW - 1) * d->in_stride + (dxmax - dxmin) + 1;
  k->npixh = k->haloH * k->haloW;
  int BN = d->BN;
  if (BN <= 0) BN = (d->CoutPad % 128 == 0) ? 128 : (d->CoutPad % 64 == 0) ? 64 : 32;
  MI_REQUIRE(BN == 32 || BN == 64 || BN == 128, "conv: BN %d", BN);
  MI_REQUIRE(d->CoutPad % BN == 0, "conv: CoutPad %d %% BN %d", d->CoutPad, BN);
  int KC = d->KC;
  const int Kp = d->K8 * 8;
  if (KC <= 0) {
    KC = (Kp % 64 == 0) ? 64 : (Kp % 32 == 0) ? 32 : 16;
    // keep two blocks per CU: shrink the k-chunk when the halo slab is large (stride-2 tiles)
    while (KC > 16) {
      size_t b = ((size_t)(KC / 8) * k->npixh + 2 * (size_t)(KC / 8) * BN) * 16;
      if (b <= 72 * 1024) break;
      KC /= 2;
    }
  }
  MI_REQUIRE((KC == 16 || KC == 32 || KC == 64) && Kp % KC == 0, "conv: KC %d for K %d", KC, Kp);
  k->nco = d->CoutPad / BN;
  *KCo = KC; *BNo = BN;
  *ldsBytes = ((size_t)(KC / 8) * k->npixh + 2 * (size_t)(KC / 8) * BN) * 16 + 4 * BN * 2 * sizeof(float);
  MI_REQUIRE(*ldsBytes <= 160 * 1024, "conv: LDS %zu too large", *ldsBytes);
  return MI_OK;
}

extern "C" int mi_conv2d_plan(mi_conv_desc* d) {
  ConvK k;
  int KC, BN;
  size_t lds;
  int rc = conv_fill(d, &k, &KC, &BN, &lds);
  if (rc) return rc;
  d->TH = k.TH; d->TW = k.TW; d->KC = KC; d->BN = BN;
  return d->N * k.tilesY * k.tilesX;
}

template <int KC, int BN, int WM, int WN, int CT, int PT>
static int launch_cfg(const ConvK& k, size_t lds, hipStream_t s) {
  auto fn = conv_igemm_kernel<KC, BN, WM, WN, CT, PT>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  dim3 grid((unsigned)(k.N * k.tilesY * k.tilesX * k.nco));
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, s, k);
  MI_CHECK_LAUNCH("conv_igemm");
  return MI_OK;
}

extern "C" int mi_conv2d(const mi_conv_desc* d, mi_stream_t st) {
  ConvK k;
  int KC, BN;
  size_t lds;
  int rc = conv_fill(d, &k, &KC, &BN, &lds);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)st;
#define MI_DISPATCH(KCv)                                                          \
  if (KC == KCv) {                                                                \
    if (BN == 32) return launch_cfg<KCv, 32, 1, 4, 1, 1>(k, lds, s);              \
    if (BN == 64) return launch_cfg<KCv, 64, 2, 2, 1, 2>(k, lds, s);              \
    return launch_cfg<KCv, 128, 2, 2, 2, 2>(k, lds, s);                           \
  }
  MI_DISPATCH(16)
  MI_DISPATCH(32)
  MI_DISPATCH(64)
#undef MI_DISPATCH
  MI_FAIL(MI_EINVAL, "conv: no kernel for KC %d BN %d", KC, BN);
}

// ================================================================= weight (un)packing
__global__ void pack_w_kernel(const float* __restrict__ w, int Cout, int Cin, int KK, __bf16* wf,
                              int CinPad, int CoutPad, __bf16* wd, int CoutPadK, int CinPadN) {
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
      if (co < Cout && ci < Cin) v = w[((int64_t)co * Cin + ci) * KK + tap];
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
      if (co < Cout && ci < Cin) v = w[((int64_t)co * Cin + ci) * KK + tap];
      wd[i2] = (__bf16)v;
    }
  }
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
