// Depthwise 3x3 convolution (groups = channels), NHWC bf16, stride 1 or 2, pad 1: the `dconv` of the reference's DWConv
// (yolov7/modeling/backbone/layers/wrappers.py:86-102; MODEL.DARKNET.DEPTH_WISE True, darknetx.py:113,210) - forward with
// the BatchNorm statistics of the following BaseConv.bn, data gradient and weight gradient.
//
// 9 multiply-adds per output element: HBM-bound by a wide margin, so no matrix cores and no LDS tiling - a thread owns one
// 8-channel group (16-byte accesses), keeps its 72 weights in registers and walks strips of 4 output pixels of a row, so
// that the 3 x (4*stride + 2) input vectors of a strip are each loaded once; neighbouring strips overlap in the caches.
// Weights are read from the fp32 parameter tensor [C][1][3][3] and rounded to bf16 like every other conv operand.
#include "common.h"

#define DW_STRIP 4

struct DwK {
  const __bf16* x;   // [N][H][W][ldx]
  __bf16* y;         // [N][Ho][Wo][ldy]
  const float* w;    // [C][9] fp32
  double* stats;     // [nslots][CA][2] or NULL
  int ldx, ldy, N, H, W, Ho, Wo, C8, stride, nslots, accumulate;
};

__device__ __forceinline__ void dw_load_w(const float* __restrict__ w, int c8, float (&wr)[9][8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][e] = (float)(__bf16)w[(size_t)(c8 * 8 + e) * 9 + t];
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void dwconv3x3_fwd_kernel(const DwK p) {
  __shared__ float red[256 * 16];
  const int tid = threadIdx.x;
  const int C8 = p.C8, PL = 256 / C8;
  const bool active = tid < PL * C8;
  const int c8 = tid % C8, pl = active ? tid / C8 : 0;
  const int S = p.stride;
  float wr[9][8];
  dw_load_w(p.w, c8, wr);
  const int wstrips = (p.Wo + DW_STRIP - 1) / DW_STRIP;
  const int64_t nstrips = (int64_t)p.N * p.Ho * wstrips;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  for (int64_t sidx = active ? (int64_t)blockIdx.x * PL + pl : nstrips; sidx < nstrips; sidx += (int64_t)gridDim.x * PL) {
    const int ws = (int)(sidx % wstrips);
    const int64_t row = sidx / wstrips;
    const int oy = (int)(row % p.Ho), n = (int)(row / p.Ho);
    const int ox0 = ws * DW_STRIP;
    float acc[DW_STRIP][8];
#pragma unroll
    for (int j = 0; j < DW_STRIP; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = oy * S + r - 1;
      if (iy < 0 || iy >= p.H) continue;
      const __bf16* xrow = p.x + ((size_t)((size_t)n * p.H + iy) * p.W) * p.ldx + c8 * 8;
      const int ncol = (DW_STRIP - 1) * S + 3;
      for (int col = 0; col < ncol; ++col) {
        const int ix = ox0 * S + col - 1;
        if (ix < 0 || ix >= p.W) continue;
        const bf16x8 xv = *(const bf16x8*)(xrow + (size_t)ix * p.ldx);
#pragma unroll
        for (int j = 0; j < DW_STRIP; ++j) {
          const int s = col - j * S;
          if (s >= 0 && s < 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] += (float)xv[e] * wr[r * 3 + s][e];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DW_STRIP; ++j) {
      const int ox = ox0 + j;
      if (ox >= p.Wo) break;
      const bf16x8 ob = pack8(acc[j]);
      *(bf16x8*)(p.y + ((size_t)((size_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy + c8 * 8) = ob;
      // the statistics of the values that are STORED (bf16-rounded), like the dense conv's epilogue: the BatchNorm
      // that follows normalises exactly this tensor
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)ob[e];
        s1[e] += f;
        s2[e] += f * f;
      }
    }
  }
  if (!p.stats) return;
  if (active) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(pl * C8 + c8) * 16 + e] = s1[e];
      red[(pl * C8 + c8) * 16 + 8 + e] = s2[e];
    }
  }
  __syncthreads();
  const int nout = C8 * 16, C = C8 * 8, CA = (C + 31) / 32 * 32;
  double* slot = p.stats + (size_t)(blockIdx.x % p.nslots) * CA * 2;
  for (int j = tid; j < nout; j += 256) {
    float a = 0.f;
    for (int q = 0; q < PL; ++q) a += red[q * nout + j];
    const int cc8 = j / 16, v = j % 16;
    atomicAdd(slot + (cc8 * 8 + (v & 7)) * 2 + (v >> 3), (double)a);
  }
}

// ------------------------------------------------------------------------------------------------ data gradient
// dx[n][iy][ix][c] (+)= sum_{r,s} dy[n][oy][ox][c] * w[c][r][s]  over (oy, ox) with oy*S + r - 1 == iy, ox*S + s - 1 == ix
__global__ __launch_bounds__(256) void dwconv3x3_dgrad_kernel(const DwK p) {   // x = dy (Ho x Wo), y = dx (H x W)
  const int tid = threadIdx.x;
  const int C8 = p.C8, PL = 256 / C8;
  if (tid >= PL * C8) return;
  const int c8 = tid % C8, pl = tid / C8;
  const int S = p.stride;
  float wr[9][8];
  dw_load_w(p.w, c8, wr);
  const int wstrips = (p.W + DW_STRIP - 1) / DW_STRIP;
  const int64_t nstrips = (int64_t)p.N * p.H * wstrips;
  for (int64_t sidx = (int64_t)blockIdx.x * PL + pl; sidx < nstrips; sidx += (int64_t)gridDim.x * PL) {
    const int ws = (int)(sidx % wstrips);
    const int64_t row = sidx / wstrips;
    const int iy = (int)(row % p.H), n = (int)(row / p.H);
    const int ix0 = ws * DW_STRIP;
    float acc[DW_STRIP][8];
#pragma unroll
    for (int j = 0; j < DW_STRIP; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int t = iy + 1 - r;
      if (t < 0 || t % S) continue;
      const int oy = t / S;
      if (oy >= p.Ho) continue;
      const __bf16* drow = p.x + ((size_t)((size_t)n * p.Ho + oy) * p.Wo) * p.ldx + c8 * 8;
      // output columns whose taps reach [ix0, ix0 + DW_STRIP): ox * S in [ix0 - 1, ix0 + DW_STRIP]
      int oxa = ix0 - 1;
      oxa = oxa <= 0 ? 0 : (oxa + S - 1) / S;
      int oxb = (ix0 + DW_STRIP) / S;
      if (oxb > p.Wo - 1) oxb = p.Wo - 1;
      for (int ox = oxa; ox <= oxb; ++ox) {
        const bf16x8 dv = *(const bf16x8*)(drow + (size_t)ox * p.ldx);
#pragma unroll
        for (int j = 0; j < DW_STRIP; ++j) {
          const int s = ix0 + j + 1 - ox * S;
          if (s >= 0 && s < 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] += (float)dv[e] * wr[r * 3 + s][e];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DW_STRIP; ++j) {
      const int ix = ix0 + j;
      if (ix >= p.W) break;
      __bf16* dp = p.y + ((size_t)((size_t)n * p.H + iy) * p.W + ix) * p.ldy + c8 * 8;
      if (p.accumulate) {
        const bf16x8 o = *(const bf16x8*)dp;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] += (float)o[e];
      }
      *(bf16x8*)dp = pack8(acc[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dw[c][r][s] = sum_{n,oy,ox} x[n][oy*S+r-1][ox*S+s-1][c] * dy[n][oy][ox][c]: block partials ws[blk][9][C], then a
// fixed-order sum over the blocks (deterministic)
struct DwWgK {
  const __bf16* x;
  const __bf16* dy;
  float* ws;
  int ldx, lddy, N, H, W, Ho, Wo, C8, stride;
};
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const DwWgK p) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const int C8 = p.C8, PL = 256 / C8;
  const bool active = tid < PL * C8;
  const int c8 = tid % C8, pl = active ? tid / C8 : 0;
  const int S = p.stride;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  const int64_t npix = (int64_t)p.N * p.Ho * p.Wo;
  for (int64_t pix = active ? (int64_t)blockIdx.x * PL + pl : npix; pix < npix; pix += (int64_t)gridDim.x * PL) {
    const int ox = (int)(pix % p.Wo);
    const int64_t row = pix / p.Wo;
    const int oy = (int)(row % p.Ho), n = (int)(row / p.Ho);
    const bf16x8 dv = *(const bf16x8*)(p.dy + (size_t)pix * p.lddy + c8 * 8);
    float d[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = (float)dv[e];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = oy * S + r - 1;
      if (iy < 0 || iy >= p.H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ix = ox * S + s - 1;
        if (ix < 0 || ix >= p.W) continue;
        const bf16x8 xv = *(const bf16x8*)(p.x + ((size_t)((size_t)n * p.H + iy) * p.W + ix) * p.ldx + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[r * 3 + s][e] += (float)xv[e] * d[e];
      }
    }
  }
  const int C = C8 * 8;
  float* out = p.ws + (size_t)blockIdx.x * 9 * C;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(pl * C8 + c8) * 8 + e] = acc[t][e];
    }
    __syncthreads();
    for (int j = tid; j < C; j += 256) {
      float a = 0.f;
      for (int q = 0; q < PL; ++q) a += red[q * C + j];
      out[t * C + j] = a;
    }
  }
}
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_reduce_kernel(const float* __restrict__ ws, int nblk, int C,
                                                                     float* __restrict__ dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // i = t * C + c
  if (i >= 9 * C) return;
  float a = 0.f;
  for (int b = 0; b < nblk; ++b) a += ws[(size_t)b * 9 * C + i];
  const int t = i / C, c = i % C;
  dw[(size_t)c * 9 + t] = a;
}

// ------------------------------------------------------------------------------------------------ launchers
static int dw_blocks(int64_t units, int C8, int cap) {
  const int PL = 256 / C8;
  int64_t b = (units + PL - 1) / PL;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}
static int dw_check(const char* what, int C, int stride, int H, int W, int Ho, int Wo) {
  MI_REQUIRE(C > 0 && C % 8 == 0 && C <= 2048, "%s: C %d (need C %% 8 == 0, C <= 2048)", what, C);
  MI_REQUIRE(stride == 1 || stride == 2, "%s: stride %d", what, stride);
  MI_REQUIRE(Ho == (H + 2 - 3) / stride + 1 && Wo == (W + 2 - 3) / stride + 1, "%s: output size %dx%d for %dx%d stride %d",
             what, Ho, Wo, H, W, stride);
  return MI_OK;
}

extern "C" int mi_dwconv3x3_fwd(const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, int C,
                                int stride, int outH, int outW, double* stats_acc, int nslots, mi_stream_t st) {
  MI_REQUIRE(x && w && y && N > 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "dwconv_fwd: args");
  int rc = dw_check("dwconv_fwd", C, stride, H, W, outH, outW);
  if (rc) return rc;
  DwK k;
  k.x = (const __bf16*)x; k.y = (__bf16*)y; k.w = w; k.stats = stats_acc; k.ldx = ldx; k.ldy = ldy; k.N = N; k.H = H;
  k.W = W; k.Ho = outH; k.Wo = outW; k.C8 = C / 8; k.stride = stride; k.accumulate = 0;
  k.nslots = (nslots >= 1 && nslots <= MI_BN_SLOTS) ? nslots : MI_BN_SLOTS;
  const int64_t strips = (int64_t)N * outH * ((outW + DW_STRIP - 1) / DW_STRIP);
  hipLaunchKernelGGL(dwconv3x3_fwd_kernel, dim3(dw_blocks(strips, C / 8, 2048)), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("dwconv_fwd");
  return MI_OK;
}

extern "C" int mi_dwconv3x3_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx, int N, int H, int W, int C,
                                  int stride, int outH, int outW, int accumulate, mi_stream_t st) {
  MI_REQUIRE(dy && w && dx && N > 0 && lddy % 8 == 0 && lddx % 8 == 0 && lddy >= C && lddx >= C, "dwconv_dgrad: args");
  int rc = dw_check("dwconv_dgrad", C, stride, H, W, outH, outW);
  if (rc) return rc;
  DwK k;
  k.x = (const __bf16*)dy; k.y = (__bf16*)dx; k.w = w; k.stats = nullptr; k.ldx = lddy; k.ldy = lddx; k.N = N; k.H = H;
  k.W = W; k.Ho = outH; k.Wo = outW; k.C8 = C / 8; k.stride = stride; k.accumulate = accumulate; k.nslots = 1;
  const int64_t strips = (int64_t)N * H * ((W + DW_STRIP - 1) / DW_STRIP);
  hipLaunchKernelGGL(dwconv3x3_dgrad_kernel, dim3(dw_blocks(strips, C / 8, 2048)), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("dwconv_dgrad");
  return MI_OK;
}

#define DW_WG_BLOCKS 512
extern "C" int64_t mi_dwconv3x3_wgrad_ws_bytes(int C) { return (int64_t)DW_WG_BLOCKS * 9 * C * 4; }
extern "C" int mi_dwconv3x3_wgrad(const void* x, int ldx, const void* dy, int lddy, int N, int H, int W, int C, int stride,
                                  int outH, int outW, float* ws, int64_t ws_bytes, float* dw, mi_stream_t st) {
  MI_REQUIRE(x && dy && ws && dw && N > 0 && ldx % 8 == 0 && lddy % 8 == 0 && ldx >= C && lddy >= C, "dwconv_wgrad: args");
  int rc = dw_check("dwconv_wgrad", C, stride, H, W, outH, outW);
  if (rc) return rc;
  MI_REQUIRE(ws_bytes >= mi_dwconv3x3_wgrad_ws_bytes(C), "dwconv_wgrad: workspace of %lld bytes, need %lld", (long long)ws_bytes,
             (long long)mi_dwconv3x3_wgrad_ws_bytes(C));
  DwWgK k;
  k.x = (const __bf16*)x; k.dy = (const __bf16*)dy; k.ws = ws; k.ldx = ldx; k.lddy = lddy; k.N = N; k.H = H; k.W = W;
  k.Ho = outH; k.Wo = outW; k.C8 = C / 8; k.stride = stride;
  const int nb = dw_blocks(((int64_t)N * outH * outW + 7) / 8, C / 8, DW_WG_BLOCKS);   // >= 8 pixels per thread
  hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3(nb), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("dwconv_wgrad");
  hipLaunchKernelGGL(dwconv3x3_wgrad_reduce_kernel, dim3(mi_cdiv(9 * C, 256)), dim3(256), 0, (hipStream_t)st, ws, nb, C, dw);
  MI_CHECK_LAUNCH("dwconv_wgrad_reduce");
  return MI_OK;
}
