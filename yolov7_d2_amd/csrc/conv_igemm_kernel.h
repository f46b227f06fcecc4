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
// (kernel template + per-k-chunk launchers; instantiated once per k-chunk in conv_igemm_kc{16,32,64,128}.hip so that the
//  four translation units compile in parallel)
#pragma once
#include <string.h>
#include "common.h"

struct ConvK {
  const __bf16* x;
  const u32x4* w;
  void* y;
  const float* bias;
  double* stats;  // [MI_BN_SLOTS][CoutPad][2] fp64 accumulators
  int ldx, ldy, N, H, W, outH, outW, gridH, gridW, is, os, ooy, oox, K8, Cout, CoutPad, ntaps;
  long long ynstride;
  int toff[MI_MAX_TAPS], tw[MI_MAX_TAPS];
  int flags, TH, TW, tilesY, tilesX, nco, nslots;
  int dymin, dxmin, haloW, npixh, nqx, xbytes;
  int tps, xstride, wstride;  // taps per weight slab; byte strides of the (double) halo / slab buffers (0: single)
  // MI_CONV_BNBWD: BatchNorm-backward sums of the layer that produced this launch's output tensor
  const __bf16* bn_y;
  const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
  int bn_ldy, bn_act;
  unsigned mTW, mHW;  // ceil(2^20 / TW), ceil(2^20 / haloW)
  int xmap;           // XCD-aware block order: blocks / 8 when the launch has several cout tiles and blocks % 8 == 0, else 0
#ifdef MI_CONV_TIMELINE
  long long* tl;      // [blocks][8] phase timestamps (tools/conv_timeline.py; diagnostic builds only)
#endif
};
#ifdef MI_CONV_TIMELINE
#define CONV_TL(k) do { if (p.tl && threadIdx.x == 0 && bid < 8192) p.tl[bid * 8 + (k)] = wall_clock64(); } while (0)
#else
#define CONV_TL(k)
#endif

static __device__ uint4 g_conv_zero_page[4];

// 16-byte LDS-DMA: LDS[lds_off + lane*16 .. +16) = *g (lds_off wave-uniform).  Inline asm keeps the compiler from
// fencing every later ds_read with vmcnt(0); the step loop waits explicitly before its barrier.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g),
               "s"(__builtin_amdgcn_readfirstlane(lds_off))
               : "memory", "m0");
}

// TPS: taps per step as a compile-time constant (1: the classic one-tap step, fully scheduled by the compiler) or
// 0: run-time p.tps (multi-tap steps of the small-K / stride-2 / parity-class launches)
// EPI: 1 = the staged epilogue may accumulate into y (MI_CONV_ACCUM) and / or take the BatchNorm-backward sums
// (MI_CONV_BNBWD) - its global operands are prefetched into registers; 0 = plain store (+ forward statistics), which
// keeps the forward kernels' register count (occupancy) low; 2 = EPI 1 + a second tensor at the output pixels used as a
// ReLU mask or as a residual under a ReLU (MI_CONV_RELUMASK / MI_CONV_ADDRELU; single launches only)
// PK: ConvK (kernel argument) or an address-space-4 (constant) ConvK for a job table entry: constant-address-space
// loads are invariant, so the compiler keeps the fields in SGPRs across the "memory"-clobbering LDS-DMA asm and the
// stores instead of re-loading them at every use
typedef const __attribute__((address_space(4))) ConvK ConvKC;
template <int KC, int BN, int WM, int WN, int CT, int PT, int TPS, int EPI, class PK>
__device__ __forceinline__ void conv_igemm_body(PK& p, const int bid) {
  static_assert(WM * CT * 32 == BN, "cout tiling");
  constexpr int NW = WM * WN;
  constexpr int TPIX = WN * PT * 32;
  constexpr int KC8 = KC / 8, KS = KC / 16, R = KC * 2;
  constexpr int RBSH = (KC == 128) ? 0 : (KC == 64) ? 1 : (KC == 32) ? 2 : 3;  // log2(rows per 256-byte bank row)
  constexpr int RPI = 64 / KC8;                              // halo rows per LDS-DMA instruction
  constexpr int WCH = KC8 * BN;                              // 16-byte rows per weight slab
  constexpr int WQ = WCH / 64;                               // LDS-DMA instructions per weight slab
  static_assert(WCH % 64 == 0, "weight slab");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [x0][x1 (if > 1 k-chunk)][w0][w1 (if > 1 step)]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int xbytes = p.xbytes;
  const int wbase = xbytes + p.xstride;
  char* const Wb = smem + wbase;
  float* Ss = (float*)smem;  // [WN][BN][2] (direct epilogue only; aliases the halo buffer after the last step)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  // blocks of one pixel tile (its nco cout tiles) read the same halo: give each XCD (block id % 8) a contiguous range of
  // (tile, cout tile) pairs so that they meet in one L2 instead of fetching the tile once per XCD (p.xmap = blocks / 8)
  const int lb = p.xmap ? (bid & 7) * p.xmap + (bid >> 3) : bid;
  const int cot = lb % p.nco;
  const int tile = lb / p.nco;
  const int tpi = p.tilesY * p.tilesX;
  const int img = tile / tpi;
  const int rem = tile - img * tpi;
  const int tyq = rem / p.tilesX;
  const int ty0 = tyq * p.TH, tx0 = (rem - tyq * p.tilesX) * p.TW;
  const int co0 = cot * BN;
  const int iy0 = ty0 * p.is + p.dymin, ix0 = tx0 * p.is + p.dxmin;
  const int TP = p.TH * p.TW;

  int pixbase[PT], gy[PT], gx[PT];
  bool pvalid[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int P = (wn * PT + j) * 32 + l31;
    const bool v = P < TP;
    const int ty = v ? (int)(((unsigned)P * p.mTW) >> 20) : 0;
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
  const int tps = TPS ? TPS : p.tps;
  const int ngr = TPS == 1 ? p.ntaps : p.ntaps / tps;  // tap groups per k-chunk
  const int nsteps = nchunks * ngr;
  const char* const zero = (const char*)g_conv_zero_page;
  const char* const xb = (const char*)(p.x + ((size_t)img * p.H * p.W) * (size_t)p.ldx);

  auto issue_w = [&](int step) {
    const int kc = step / ngr, t0 = (step - kc * ngr) * tps;
    const unsigned dst = lds0 + wbase + (step & 1) * p.wstride;
    for (int q = wave; q < WQ * tps; q += NW) {
      const int tt = TPS == 1 ? 0 : q / WQ, qq = q - tt * WQ;
      const u32x4* src = p.w + ((size_t)(p.tw[t0 + tt] * p.K8 + kc * KC8)) * p.CoutPad + co0;
      const int idx = qq * 64 + lane;
      const int c8 = idx / BN, co = idx % BN;
      glds16(src + (size_t)c8 * p.CoutPad + co, dst + q * 1024);
    }
  };
  auto issue_x = [&](int kc) {
    const unsigned dst = lds0 + (kc & 1) * p.xstride;
    for (int q = wave; q < p.nqx; q += NW) {
      const int row = q * RPI + lane / KC8;
      const int chunk = (lane % KC8) ^ ((row >> RBSH) & (KC8 - 1));
      const int hy = (int)(((unsigned)row * p.mHW) >> 20);
      const int hx = row - hy * p.haloW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool v = (row < p.npixh) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const unsigned off = (unsigned)(((iy * p.W + ix) * p.ldx + kc * KC + chunk * 8) * 2);
      glds16(v ? xb + off : zero, dst + q * 1024);
    }
  };

  const int nsteps_run = (p.flags & 256) ? 0 : nsteps;
  CONV_TL(0);
  if (nsteps_run) { issue_w(0); issue_x(0); }
  CONV_TL(1);
  for (int step = 0; step < nsteps_run; ++step) {
    const int kc = step / ngr, g = step - kc * ngr;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // slabs `step` and chunk `kc` landed; the other buffers are no longer read
    if (step == 0) CONV_TL(2);
    if (step + 1 < nsteps) issue_w(step + 1);
    if (g == 0 && kc + 1 < nchunks) issue_x(kc + 1);
    const char* Xs = smem + (kc & 1) * p.xstride;
    for (int tt = 0; tt < tps; ++tt) {
    const u32x4* Ws = (const u32x4*)(Wb + (step & 1) * p.wstride) + tt * WCH;
    const int toff = p.toff[g * tps + tt];
    int xrow[PT], xsw[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const int row = pixbase[j] + toff;
      xrow[j] = row * R;
      xsw[j] = (row >> RBSH) & (KC8 - 1);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k8 = ks * 2 + h;
      bf16x8 a[CT], b[PT];
#pragma unroll
      for (int i = 0; i < CT; ++i)
        a[i] = __builtin_bit_cast(bf16x8, Ws[k8 * BN + (wm * CT + i) * 32 + l31]);
#pragma unroll
      for (int j = 0; j < PT; ++j)
        b[j] = __builtin_bit_cast(bf16x8, *(const u32x4*)(Xs + xrow[j] + ((k8 ^ xsw[j]) << 4)));
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    }
  }

  CONV_TL(3);
  if (p.flags & 512) {
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) z += acc[i][j][r];
    if (z == 1.2345f) ((float*)p.y)[0] = z;
    return;
  }
  const bool do_stats = p.stats != nullptr;
  const bool accum = (p.flags & MI_CONV_ACCUM) != 0;
  const bool outf32 = (p.flags & MI_CONV_OUT_F32) != 0;
  if (!outf32 && (p.Cout & 7) == 0) {
    // ---- staged epilogue: accumulators -> bf16 tile in LDS [pixel][BN] -> 16-byte row-contiguous NHWC stores;
    // the BatchNorm partial sums are taken from the very values that are stored (bf16-rounded), in a fixed order.
    constexpr int NTH = NW * 64;
    constexpr int RS = BN * 2 + 16;  // staging row stride: +16 B keeps the 8-byte fragment writes conflict-free
    constexpr int C8N = BN / 8, PPI = NTH / C8N;
    static_assert(NTH % C8N == 0, "epilogue thread mapping");
    // this thread's output rows: NP pixels x one 8-channel group
    constexpr int NP = TPIX / PPI;
    static_assert(TPIX % PPI == 0, "epilogue rows");
    const int c8 = tid % C8N, pr = tid / C8N;
    const int cbase = co0 + c8 * 8;
    const bool cvalid = cbase < p.Cout;
    __bf16* const yb = (__bf16*)p.y + (size_t)img * (size_t)p.ynstride + cbase;
    auto out_pixel = [&](int P) {   // linear output pixel of tile row P, or -1
      const int ty = (int)(((unsigned)P * p.mTW) >> 20);
      const int tx = P - ty * p.TW;
      const int gyy = ty0 + ty, gxx = tx0 + tx;
      const bool v = (P < TP) & (gyy < p.gridH) & (gxx < p.gridW) & cvalid;
      return v ? (gyy * p.os + p.ooy) * p.outW + gxx * p.os + p.oox : -1;
    };
    auto stage = [&]() {
      __syncthreads();  // every wave is done with the halo / weight buffers
      char* Tb = smem;
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const int row = (wn * PT + j) * 32 + l31;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cl = (wm * CT + i) * 32 + 8 * q + 4 * h;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[i][j][4 * q + e];
              if (p.bias && co0 + cl + e < p.Cout) v += p.bias[co0 + cl + e];
              if (p.flags & MI_CONV_RELU) v = fmaxf(v, 0.f);
              o[e] = (__bf16)v;
            }
            *(bf16x4*)(Tb + row * RS + cl * 2) = o;
          }
        }
      __syncthreads();
    };
    const char* const Tb = smem;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if constexpr (EPI == 0) {
      // (measured dead end: taking the statistics from the staged tile FIRST and issuing the atomics before the stores
      //  - so that their latency overlaps the store phase - is 0.6 % slower than this order)
      stage();
      CONV_TL(4);
#pragma unroll 2
      for (int P = pr; P < TPIX; P += PPI) {
        const int op = out_pixel(P);
        if (op >= 0) {
          const bf16x8 v = *(const bf16x8*)(Tb + P * RS + c8 * 16);
          *(bf16x8*)(yb + (size_t)op * (size_t)p.ldy) = v;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            s1[e] += f;
            s2[e] += f * f;
          }
        }
      }
    } else {
      // the global operands of the store loop (old values of an accumulating launch, the producing layer's conv
      // output of a BNBWD launch) are requested BEFORE the staging, so their latency overlaps it instead of
      // serialising NP round trips in the loop
      const bool bnb = (p.flags & MI_CONV_BNBWD) != 0;
      // EPI 2 (its own instantiations: the EPI 1 kernels keep their code): the second tensor at the output pixels is a ReLU
      // OUTPUT whose sign masks the result (MI_CONV_RELUMASK: the ReLU backward of the layer this data gradient flows into)
      // or a residual that is added before a ReLU (MI_CONV_ADDRELU: conv3 + shortcut + ReLU of a bottleneck block)
      bool aux = false;
      if constexpr (EPI == 2) aux = (p.flags & (MI_CONV_RELUMASK | MI_CONV_ADDRELU)) != 0;
      const __bf16* const byb = p.bn_y + (size_t)img * (size_t)p.outH * p.outW * p.bn_ldy + cbase;
      int opix[NP];
      bf16x8 oldv[NP], yv[NP];
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        opix[it] = out_pixel(pr + it * PPI);
        if (opix[it] >= 0 && accum) oldv[it] = *(const bf16x8*)(yb + (size_t)opix[it] * (size_t)p.ldy);
        if (opix[it] >= 0 && (bnb || aux)) yv[it] = *(const bf16x8*)(byb + (size_t)opix[it] * (size_t)p.bn_ldy);
      }
      float bsc[8], bsh[8], bmu[8], bis[8];
      if (bnb && cvalid) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bsc[e] = p.bn_scale[cbase + e]; bsh[e] = p.bn_shift[cbase + e];
          bmu[e] = p.bn_mean[cbase + e];  bis[e] = p.bn_invstd[cbase + e];
        }
      }
      stage();
      CONV_TL(4);
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        if (opix[it] >= 0) {
          const int P = pr + it * PPI;
          bf16x8 v = *(const bf16x8*)(Tb + P * RS + c8 * 16);
          if (accum) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] + (float)oldv[it][e]);
          }
          if constexpr (EPI == 2) {
            if (p.flags & MI_CONV_ADDRELU) {          // relu(bf16(conv + residual)): the rounding of mi_ew_bf16 op 7
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (__bf16)fmaxf((float)(__bf16)((float)v[e] + (float)yv[it][e]), 0.f);
            } else if (p.flags & MI_CONV_RELUMASK) {   // dy * (a > 0): mi_ew_bf16 op 2
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (float)yv[it][e] > 0.f ? v[e] : (__bf16)0.f;
            }
          }
          *(bf16x8*)(yb + (size_t)opix[it] * (size_t)p.ldy) = v;
          if (bnb) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float yy = (float)yv[it][e];
              const float z = yy * bsc[e] + bsh[e];
              float g = 1.f;
              if (p.bn_act) {
                const float sg = sigmoidf_(z);
                g = sg * (1.f + z * (1.f - sg));
              }
              const float dz = (float)v[e] * g;
              s1[e] += dz;
              s2[e] += dz * ((yy - bmu[e]) * bis[e]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float f = (float)v[e];
              s1[e] += f;
              s2[e] += f * f;
            }
          }
        }
      }
    }
    CONV_TL(5);
    if (do_stats) {
      // threads of a wave that own the same 8 channels (lane % C8N) fold their sums by lane exchange; one row per wave
      // goes through LDS, and BN threads add the NW rows and issue the atomics
#pragma unroll
      for (int off = C8N; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += __shfl_xor(s1[e], off, 64);
          s2[e] += __shfl_xor(s2[e], off, 64);
        }
      __syncthreads();  // staging tile fully consumed
      float* Rs = (float*)smem;  // [NW][BN][2]
      if (lane < C8N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          Rs[(wave * BN + c8 * 8 + e) * 2 + 0] = s1[e];
          Rs[(wave * BN + c8 * 8 + e) * 2 + 1] = s2[e];
        }
      }
      __syncthreads();
      if (tid < BN) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
          a1 += Rs[(q * BN + tid) * 2 + 0];
          a2 += Rs[(q * BN + tid) * 2 + 1];
        }
        double* sp = p.stats + ((size_t)(tile % p.nslots) * p.CoutPad + co0 + tid) * 2;
        if (!(p.flags & 1024)) {   // (1024: timing experiments without the atomics)
          atomicAdd(sp, (double)a1);
          atomicAdd(sp + 1, (double)a2);
        }
      }
    }
    CONV_TL(6);
    return;
  }
  // ---- direct epilogue (fp32 prediction maps / ragged channel counts): D[m = cout][n = pixel]
  if (do_stats) __syncthreads();  // Ss aliases the halo buffer
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
        if (p.flags & MI_CONV_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
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
      double* sp = p.stats + ((size_t)(tile % p.nslots) * p.CoutPad + co0 + tid) * 2;
      atomicAdd(sp, (double)a1);
      atomicAdd(sp + 1, (double)a2);
    }
  }
}

template <int KC, int BN, int WM, int WN, int CT, int PT, int TPS, int EPI>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv_igemm_kernel(const ConvK p) {
  conv_igemm_body<KC, BN, WM, WN, CT, PT, TPS, EPI, const ConvK>(p, blockIdx.x);
}

// several independent convolutions (same template configuration, their own shapes / tensors) in ONE launch: the
// block looks its job up in a device table.  Used for the three FPN levels of the head, whose 40x40 / 20x20
// launches are latency-bound on their own and ride along with the 80x80 level here.
template <int KC, int BN, int WM, int WN, int CT, int PT, int TPS, int EPI>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv_igemm_group_kernel(const ConvK* __restrict__ jobs,
                                                                          const int* __restrict__ starts, int njobs) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= starts[j + 1]) ++j;
  j = __builtin_amdgcn_readfirstlane(j);
  ConvKC* pj = (ConvKC*)(uintptr_t)(jobs + j);
  conv_igemm_body<KC, BN, WM, WN, CT, PT, TPS, EPI, ConvKC>(*pj, (int)blockIdx.x - starts[j]);
}

// ---------------------------------------------------------------- launchers (one k-chunk per translation unit)
template <int KC, int BN, int WM, int WN, int CT, int PT, int TPS, int EPI>
static int launch_one(const ConvK& k, size_t lds, hipStream_t s) {
  auto fn = conv_igemm_kernel<KC, BN, WM, WN, CT, PT, TPS, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  dim3 grid((unsigned)(k.N * k.tilesY * k.tilesX * k.nco));
  hipLaunchKernelGGL(fn, grid, dim3(WM * WN * 64), lds, s, k);
  MI_CHECK_LAUNCH("conv_igemm");
  return MI_OK;
}
template <int KC, int BN, int WM, int WN, int CT, int PT>
static int launch_cfg(const ConvK& k, size_t lds, hipStream_t s) {
  const bool epi = (k.flags & (MI_CONV_ACCUM | MI_CONV_BNBWD)) != 0;
  if (k.flags & (MI_CONV_RELUMASK | MI_CONV_ADDRELU))      // the aux-tensor epilogue: separate instantiations
    return k.tps == 1 ? launch_one<KC, BN, WM, WN, CT, PT, 1, 2>(k, lds, s) : launch_one<KC, BN, WM, WN, CT, PT, 0, 2>(k, lds, s);
  if (k.tps == 1) return epi ? launch_one<KC, BN, WM, WN, CT, PT, 1, 1>(k, lds, s) : launch_one<KC, BN, WM, WN, CT, PT, 1, 0>(k, lds, s);
  return epi ? launch_one<KC, BN, WM, WN, CT, PT, 0, 1>(k, lds, s) : launch_one<KC, BN, WM, WN, CT, PT, 0, 0>(k, lds, s);
}

template <int KC, int BN, int WM, int WN, int CT, int PT, int TPS, int EPI>
static int launch_group_one(const ConvK* jobs, const int* starts, int njobs, int nblocks, size_t lds, hipStream_t s) {
  auto fn = conv_igemm_group_kernel<KC, BN, WM, WN, CT, PT, TPS, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)nblocks), dim3(WM * WN * 64), lds, s, jobs, starts, njobs);
  MI_CHECK_LAUNCH("conv_igemm_group");
  return MI_OK;
}
template <int KC, int BN, int WM, int WN, int CT, int PT>
static int launch_group_cfg(const mi_conv_group* m, const ConvK* jobs, const int* starts, hipStream_t s) {
  const size_t lds = (size_t)m->lds_bytes;
  if (m->TPS == 1)
    return m->EPI ? launch_group_one<KC, BN, WM, WN, CT, PT, 1, 1>(jobs, starts, m->njobs, m->nblocks, lds, s)
                  : launch_group_one<KC, BN, WM, WN, CT, PT, 1, 0>(jobs, starts, m->njobs, m->nblocks, lds, s);
  return m->EPI ? launch_group_one<KC, BN, WM, WN, CT, PT, 0, 1>(jobs, starts, m->njobs, m->nblocks, lds, s)
                : launch_group_one<KC, BN, WM, WN, CT, PT, 0, 0>(jobs, starts, m->njobs, m->nblocks, lds, s);
}

template <int KCv>
static int conv_launch_kc(const ConvK& k, int BN, int TPIX, size_t lds, hipStream_t s) {
  if (TPIX & 1024) {   // 8 waves on the same tile
    TPIX &= 1023;
    if (TPIX == 128 && BN == 64) return launch_cfg<KCv, 64, 2, 4, 1, 1>(k, lds, s);
    if (TPIX == 128 && BN == 128) return launch_cfg<KCv, 128, 2, 4, 2, 1>(k, lds, s);
    if (TPIX == 64 && BN == 128) return launch_cfg<KCv, 128, 4, 2, 1, 1>(k, lds, s);
  }
  if (TPIX == 128) {
    if (BN == 32) return launch_cfg<KCv, 32, 1, 4, 1, 1>(k, lds, s);
    if (BN == 64) return launch_cfg<KCv, 64, 2, 2, 1, 2>(k, lds, s);
    return launch_cfg<KCv, 128, 2, 2, 2, 2>(k, lds, s);
  }
  if (BN == 32) return launch_cfg<KCv, 32, 1, 2, 1, 1>(k, lds, s);
  if (BN == 64) return launch_cfg<KCv, 64, 2, 2, 1, 1>(k, lds, s);
  return launch_cfg<KCv, 128, 2, 2, 2, 1>(k, lds, s);
}
template <int KCv>
static int conv_group_launch_kc(const mi_conv_group* m, const ConvK* jobs, const int* starts, hipStream_t s) {
  if (m->TPIX & 1024) {   // 8 waves on the same tile
    const int tp = m->TPIX & 1023;
    if (tp == 128 && m->BN == 64) return launch_group_cfg<KCv, 64, 2, 4, 1, 1>(m, jobs, starts, s);
    if (tp == 128 && m->BN == 128) return launch_group_cfg<KCv, 128, 2, 4, 2, 1>(m, jobs, starts, s);
    if (tp == 64 && m->BN == 128) return launch_group_cfg<KCv, 128, 4, 2, 1, 1>(m, jobs, starts, s);
  }
  if (m->TPIX == 128) {
    if (m->BN == 32) return launch_group_cfg<KCv, 32, 1, 4, 1, 1>(m, jobs, starts, s);
    if (m->BN == 64) return launch_group_cfg<KCv, 64, 2, 2, 1, 2>(m, jobs, starts, s);
    return launch_group_cfg<KCv, 128, 2, 2, 2, 2>(m, jobs, starts, s);
  }
  if (m->BN == 32) return launch_group_cfg<KCv, 32, 1, 2, 1, 1>(m, jobs, starts, s);
  if (m->BN == 64) return launch_group_cfg<KCv, 64, 2, 2, 1, 1>(m, jobs, starts, s);
  return launch_group_cfg<KCv, 128, 2, 2, 2, 1>(m, jobs, starts, s);
}
