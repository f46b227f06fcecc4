// Kernels of the BiFPN neck (yolov7/modeling/neck/bifpn.py:184-395; MODEL.BACKBONE.NAME build_resnet_bifpn_backbone) that
// the YOLOX path did not need: GroupNorm (detectron2 get_norm("GN") = nn.GroupNorm(32, C), the default MODEL.BIFPN.NORM),
// MaxPool2d(2, 2) of ResampleFeatureMap, and the "fastattn" weighted feature fusion of FpnCombine.  NHWC bf16, fp32 math.
// All HBM streams: 16-byte accesses, one channel group of 8 per thread.
#include "common.h"

// ------------------------------------------------------------------------------------------------ GroupNorm
// Statistics are per (image, group) over H*W*(C/G) elements.  Pass 1 adds per-(image, channel) sums into fp64 accumulators
// acc[N][C][2] (sum, sumsq; zeroed by the launcher); pass 2 folds the C/G channels of a group in its prologue and applies
// y = (x - mean_g) * rstd_g * gamma_c + beta_c.  C / G need not be a multiple of 8 (BiFPN: 160 / 32 = 5).
struct GnK {
  const __bf16* x;
  const __bf16* dy;
  __bf16* y;
  double* acc;        // [N][C][2]
  const float* gamma;
  const float* beta;
  float* mean_rstd;   // [N][G][2]
  float* dgamma;
  float* dbeta;
  int ldx, ldy, lddy, C8, G, HW, N;
  float eps;
};
#define GN_MAXC 2048

// per-(image, channel) sums of (v, v*w) over the pixels: MODE 0: (x, x*x); MODE 1: (dy, dy * xhat)
template <int MODE>
__global__ __launch_bounds__(256) void gn_sums_kernel(const GnK p) {
  __shared__ float red[256 * 16];
  __shared__ float s_mu[GN_MAXC], s_rs[GN_MAXC];
  const int tid = threadIdx.x, n = blockIdx.y;
  const int C8 = p.C8, C = C8 * 8, PL = 256 / C8;
  const bool active = tid < PL * C8;
  const int c8 = tid % C8, pl = active ? tid / C8 : 0;
  if (MODE == 1) {
    const int cpg = C / p.G;
    for (int c = tid; c < C; c += 256) {
      s_mu[c] = p.mean_rstd[((size_t)n * p.G + c / cpg) * 2];
      s_rs[c] = p.mean_rstd[((size_t)n * p.G + c / cpg) * 2 + 1];
    }
    __syncthreads();
  }
  float s1[8], s2[8], mu[8], rs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s1[e] = s2[e] = 0.f;
    mu[e] = MODE ? s_mu[c8 * 8 + e] : 0.f;
    rs[e] = MODE ? s_rs[c8 * 8 + e] : 0.f;
  }
  const __bf16* xb = p.x + (size_t)n * p.HW * p.ldx + c8 * 8;
  const __bf16* db = MODE ? p.dy + (size_t)n * p.HW * p.lddy + c8 * 8 : nullptr;
  for (int px = active ? blockIdx.x * PL + pl : p.HW; px < p.HW; px += gridDim.x * PL) {
    const bf16x8 xv = *(const bf16x8*)(xb + (size_t)px * p.ldx);
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)xv[e];
        s1[e] += f;
        s2[e] += f * f;
      }
    } else {
      const bf16x8 dv = *(const bf16x8*)(db + (size_t)px * p.lddy);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)dv[e];
        s1[e] += d;
        s2[e] += d * (((float)xv[e] - mu[e]) * rs[e]);
      }
    }
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(pl * C8 + c8) * 16 + e] = s1[e];
      red[(pl * C8 + c8) * 16 + 8 + e] = s2[e];
    }
  }
  __syncthreads();
  const int nout = C8 * 16;
  double* a = p.acc + (size_t)n * C * 2;
  for (int j = tid; j < nout; j += 256) {
    float v = 0.f;
    for (int q = 0; q < PL; ++q) v += red[q * nout + j];
    const int cc8 = j / 16, w = j % 16;
    atomicAdd(a + (cc8 * 8 + (w & 7)) * 2 + (w >> 3), (double)v);
  }
}

__global__ __launch_bounds__(256) void gn_fwd_apply_kernel(const GnK p) {
  __shared__ float s_sc[GN_MAXC], s_sh[GN_MAXC];
  const int tid = threadIdx.x, n = blockIdx.y;
  const int C8 = p.C8, C = C8 * 8, cpg = C / p.G;
  const double cnt = (double)p.HW * cpg;
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    double t1 = 0.0, t2 = 0.0;
    for (int q = 0; q < cpg; ++q) {
      t1 += p.acc[((size_t)n * C + g * cpg + q) * 2];
      t2 += p.acc[((size_t)n * C + g * cpg + q) * 2 + 1];
    }
    const double mean = t1 / cnt;
    double var = t2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)p.eps);
    s_sc[c] = (float)((double)p.gamma[c] * rstd);
    s_sh[c] = (float)((double)p.beta[c] - mean * (double)p.gamma[c] * rstd);
    if (blockIdx.x == 0 && c == g * cpg) {
      p.mean_rstd[((size_t)n * p.G + g) * 2] = (float)mean;
      p.mean_rstd[((size_t)n * p.G + g) * 2 + 1] = (float)rstd;
    }
  }
  __syncthreads();
  const int TPB = (256 / C8) * C8;
  if (tid >= TPB) return;
  const int c8 = tid % C8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = s_sc[c8 * 8 + e];
    sh[e] = s_sh[c8 * 8 + e];
  }
  const int64_t total = (int64_t)p.HW * C8;
  const __bf16* xb = p.x + (size_t)n * p.HW * p.ldx;
  __bf16* yb = p.y + (size_t)n * p.HW * p.ldy;
  for (int64_t i = (int64_t)blockIdx.x * TPB + tid; i < total; i += (int64_t)gridDim.x * TPB) {
    const int64_t px = i / C8;
    const bf16x8 xv = *(const bf16x8*)(xb + px * p.ldx + c8 * 8);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)xv[e] * sc[e] + sh[e];
    *(bf16x8*)(yb + px * p.ldy + c8 * 8) = pack8(o);
  }
}

// dx = rstd_g * (dy * gamma_c - S1_g / m - xhat * S2_g / m),  S1_g = sum_{c in g} gamma_c * A_c,  S2_g likewise with B_c,
// m = HW * C/G;  A_c = sum_p dy, B_c = sum_p dy * xhat (pass 1, MODE 1).  y = dx here.
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const GnK p) {
  __shared__ float s_k1[GN_MAXC], s_k2[GN_MAXC], s_mu[GN_MAXC], s_rs[GN_MAXC];
  const int tid = threadIdx.x, n = blockIdx.y;
  const int C8 = p.C8, C = C8 * 8, cpg = C / p.G;
  const double m = (double)p.HW * cpg;
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    double t1 = 0.0, t2 = 0.0;
    for (int q = 0; q < cpg; ++q) {
      const int cc = g * cpg + q;
      t1 += (double)p.gamma[cc] * p.acc[((size_t)n * C + cc) * 2];
      t2 += (double)p.gamma[cc] * p.acc[((size_t)n * C + cc) * 2 + 1];
    }
    s_k1[c] = (float)(t1 / m);
    s_k2[c] = (float)(t2 / m);
    s_mu[c] = p.mean_rstd[((size_t)n * p.G + g) * 2];
    s_rs[c] = p.mean_rstd[((size_t)n * p.G + g) * 2 + 1];
  }
  if (blockIdx.x == 0 && n == 0) {   // parameter gradients: sums over the images, fixed order
    for (int c = tid; c < C; c += 256) {
      double a = 0.0, b = 0.0;
      for (int i = 0; i < p.N; ++i) {
        a += p.acc[((size_t)i * C + c) * 2];
        b += p.acc[((size_t)i * C + c) * 2 + 1];
      }
      if (p.dbeta) p.dbeta[c] = (float)a;
      if (p.dgamma) p.dgamma[c] = (float)b;
    }
  }
  __syncthreads();
  const int TPB = (256 / C8) * C8;
  if (tid >= TPB) return;
  const int c8 = tid % C8;
  float k1[8], k2[8], mu[8], rs[8], ga[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    k1[e] = s_k1[c]; k2[e] = s_k2[c]; mu[e] = s_mu[c]; rs[e] = s_rs[c]; ga[e] = p.gamma[c];
  }
  const int64_t total = (int64_t)p.HW * C8;
  const __bf16* xb = p.x + (size_t)n * p.HW * p.ldx;
  const __bf16* db = p.dy + (size_t)n * p.HW * p.lddy;
  __bf16* ob = p.y + (size_t)n * p.HW * p.ldy;
  for (int64_t i = (int64_t)blockIdx.x * TPB + tid; i < total; i += (int64_t)gridDim.x * TPB) {
    const int64_t px = i / C8;
    const bf16x8 xv = *(const bf16x8*)(xb + px * p.ldx + c8 * 8);
    const bf16x8 dv = *(const bf16x8*)(db + px * p.lddy + c8 * 8);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = ((float)xv[e] - mu[e]) * rs[e];
      o[e] = rs[e] * ((float)dv[e] * ga[e] - k1[e] - xh * k2[e]);
    }
    *(bf16x8*)(ob + px * p.ldy + c8 * 8) = pack8(o);
  }
}

static int gn_blocks(int HW, int C8) {
  int b = (int)(((int64_t)HW * C8 + 2047) / 2048);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return b;
}
static int gn_check(const char* what, int N, int HW, int C, int G) {
  MI_REQUIRE(N > 0 && HW > 0 && C > 0 && C % 8 == 0 && C <= GN_MAXC && G > 0 && C % G == 0 && N <= 65535,
             "%s: N %d HW %d C %d G %d (C %% 8 == 0, C %% G == 0, C <= %d)", what, N, HW, C, G, GN_MAXC);
  return MI_OK;
}

extern "C" int64_t mi_groupnorm_ws_bytes(int N, int C) { return (int64_t)N * C * 2 * 8; }

extern "C" int mi_groupnorm_fwd(const void* x, int ldx, int N, int HW, int C, int G, const float* gamma, const float* beta,
                                float eps, void* y, int ldy, float* mean_rstd, double* ws, mi_stream_t st) {
  MI_REQUIRE(x && y && gamma && beta && mean_rstd && ws && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "groupnorm_fwd: args");
  int rc = gn_check("groupnorm_fwd", N, HW, C, G);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)st;
  if (hipMemsetAsync(ws, 0, (size_t)mi_groupnorm_ws_bytes(N, C), s) != hipSuccess) MI_FAIL(MI_ELAUNCH, "groupnorm_fwd: memset");
  GnK k;
  k.x = (const __bf16*)x; k.dy = nullptr; k.y = (__bf16*)y; k.acc = ws; k.gamma = gamma; k.beta = beta;
  k.mean_rstd = mean_rstd; k.dgamma = k.dbeta = nullptr; k.ldx = ldx; k.ldy = ldy; k.lddy = 0; k.C8 = C / 8; k.G = G;
  k.HW = HW; k.N = N; k.eps = eps;
  const dim3 g((unsigned)gn_blocks(HW, C / 8), (unsigned)N);
  hipLaunchKernelGGL(gn_sums_kernel<0>, g, dim3(256), 0, s, k);
  hipLaunchKernelGGL(gn_fwd_apply_kernel, g, dim3(256), 0, s, k);
  MI_CHECK_LAUNCH("groupnorm_fwd");
  return MI_OK;
}

extern "C" int mi_groupnorm_bwd(const void* dy, int lddy, const void* x, int ldx, int N, int HW, int C, int G,
                                const float* gamma, const float* mean_rstd, void* dx, int lddx, float* dgamma, float* dbeta,
                                double* ws, mi_stream_t st) {
  MI_REQUIRE(dy && x && gamma && mean_rstd && dx && ws && lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0, "groupnorm_bwd: args");
  int rc = gn_check("groupnorm_bwd", N, HW, C, G);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)st;
  if (hipMemsetAsync(ws, 0, (size_t)mi_groupnorm_ws_bytes(N, C), s) != hipSuccess) MI_FAIL(MI_ELAUNCH, "groupnorm_bwd: memset");
  GnK k;
  k.x = (const __bf16*)x; k.dy = (const __bf16*)dy; k.y = (__bf16*)dx; k.acc = ws; k.gamma = gamma; k.beta = nullptr;
  k.mean_rstd = (float*)mean_rstd; k.dgamma = dgamma; k.dbeta = dbeta; k.ldx = ldx; k.ldy = lddx; k.lddy = lddy;
  k.C8 = C / 8; k.G = G; k.HW = HW; k.N = N; k.eps = 0.f;
  const dim3 g((unsigned)gn_blocks(HW, C / 8), (unsigned)N);
  hipLaunchKernelGGL(gn_sums_kernel<1>, g, dim3(256), 0, s, k);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, g, dim3(256), 0, s, k);
  MI_CHECK_LAUNCH("groupnorm_bwd");
  return MI_OK;
}

// ------------------------------------------------------------------------------------------------ MaxPool2d(2, 2)
// nn.MaxPool2d(kernel_size=2, stride=2) of ResampleFeatureMap (bifpn.py:151-155): Ho = H / 2, Wo = W / 2 (floor).
// Backward recomputes the arg-max (first maximum in row-major window order, as ATen) from x.
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const __bf16* __restrict__ x, int ldx, const __bf16* __restrict__ dy,
                                                         int lddy, __bf16* __restrict__ out, int ldo, int N, int H, int W,
                                                         int C8, int backward) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    int64_t r = idx / C8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    bf16x8 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = *(const bf16x8*)(x + (((int64_t)n * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * ldx + c8 * 8);
    if (!backward) {
      float m[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = fmaxf(fmaxf((float)v[0][e], (float)v[1][e]), fmaxf((float)v[2][e], (float)v[3][e]));
      *(bf16x8*)(out + (((int64_t)n * Ho + oy) * Wo + ox) * ldo + c8 * 8) = pack8(m);
    } else {
      const bf16x8 g = *(const bf16x8*)(dy + (((int64_t)n * Ho + oy) * Wo + ox) * lddy + c8 * 8);
      float o[4][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int am = 0;
        float bv = (float)v[0][e];
#pragma unroll
        for (int q = 1; q < 4; ++q)
          if ((float)v[q][e] > bv) { bv = (float)v[q][e]; am = q; }
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q][e] = q == am ? (float)g[e] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(bf16x8*)(out + (((int64_t)n * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * ldo + c8 * 8) = pack8(o[q]);
    }
  }
}
static int mp_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
extern "C" int mi_maxpool2x2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, mi_stream_t st) {
  MI_REQUIRE(x && y && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && H >= 2 && W >= 2 && N > 0, "maxpool2x2_fwd: args");
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(mp_blocks((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, (hipStream_t)st,
                     (const __bf16*)x, ldx, nullptr, 0, (__bf16*)y, ldy, N, H, W, C / 8, 0);
  MI_CHECK_LAUNCH("maxpool2x2_fwd");
  return MI_OK;
}
// dx must be zero-initialised by the caller when H or W is odd (the last row / column is outside every window)
extern "C" int mi_maxpool2x2_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int N, int H, int W,
                                 int C, mi_stream_t st) {
  MI_REQUIRE(x && dy && dx && C % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && H >= 2 && W >= 2 && N > 0,
             "maxpool2x2_bwd: args");
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(mp_blocks((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, (hipStream_t)st,
                     (const __bf16*)x, ldx, (const __bf16*)dy, lddy, (__bf16*)dx, lddx, N, H, W, C / 8, 1);
  MI_CHECK_LAUNCH("maxpool2x2_bwd");
  return MI_OK;
}

// ------------------------------------------------------------------------------------------------ fastattn fusion
// FpnCombine, weight_method "fastattn" (bifpn.py:221-236): w = relu(edge_weights), out = sum_i x_i * w_i / (sum w + 1e-4).
// forward: nin (2 or 3) inputs -> out.  backward: dx_i = g * w_i / S and d edge_weight_i = [w_i > 0] * sum_p g * (x_i - out) / S
// (out recomputed), the sums as per-block partials + a fixed-order final sum.
struct FaK {
  const __bf16* x[3];
  const __bf16* g;
  __bf16* out;
  __bf16* dx[3];
  const float* ew;   // raw edge weights [nin]
  float* part;       // [blocks][3]
  int nin;
  int64_t n8;
};
__device__ __forceinline__ void fa_weights(const FaK& p, float* wn, float* invS) {
  float w[3], S = 0.f;
  for (int i = 0; i < 3; ++i) {
    w[i] = i < p.nin ? fmaxf(p.ew[i], 0.f) : 0.f;
    S += w[i];
  }
  *invS = 1.f / (S + 0.0001f);
  for (int i = 0; i < 3; ++i) wn[i] = w[i] * *invS;
}
__global__ __launch_bounds__(256) void fastattn_fwd_kernel(const FaK p) {
  float wn[3], invS;
  fa_weights(p, wn, &invS);
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < p.n8; i += (int64_t)gridDim.x * 256) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int k = 0; k < p.nin; ++k) {
      const bf16x8 v = *(const bf16x8*)(p.x[k] + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += (float)v[e] * wn[k];
    }
    *(bf16x8*)(p.out + i * 8) = pack8(o);
  }
}
__global__ __launch_bounds__(256) void fastattn_bwd_kernel(const FaK p) {
  __shared__ float red[4][3];
  float wn[3], invS;
  fa_weights(p, wn, &invS);
  float acc[3] = {0.f, 0.f, 0.f};
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < p.n8; i += (int64_t)gridDim.x * 256) {
    const bf16x8 g = *(const bf16x8*)(p.g + i * 8);
    bf16x8 v[3];
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int k = 0; k < p.nin; ++k) {
      v[k] = *(const bf16x8*)(p.x[k] + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += (float)v[k][e] * wn[k];
    }
    for (int k = 0; k < p.nin; ++k) {
      float d[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        d[e] = (float)g[e] * wn[k];
        acc[k] += (float)g[e] * ((float)v[k][e] - o[e]);
      }
      if (p.dx[k]) *(bf16x8*)(p.dx[k] + i * 8) = pack8(d);
    }
  }
  for (int k = 0; k < 3; ++k) {
    float a = acc[k];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = a;
  }
  __syncthreads();
  if (threadIdx.x < 3) p.part[(size_t)blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(64) void fastattn_dw_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ ew,
                                                         int nin, float* __restrict__ dew) {
  const int k = threadIdx.x;
  if (k >= nin) return;
  float S = 0.f;
  for (int i = 0; i < nin; ++i) S += fmaxf(ew[i], 0.f);
  double a = 0.0;
  for (int b = 0; b < nblk; ++b) a += (double)part[(size_t)b * 3 + k];
  dew[k] = ew[k] > 0.f ? (float)(a / (double)(S + 0.0001f)) : 0.f;
}
#define FA_BLOCKS 1024
extern "C" int64_t mi_fastattn_ws_bytes() { return (int64_t)FA_BLOCKS * 3 * 4; }
extern "C" int mi_fastattn_fwd(const void* const* xs, int nin, const float* edge_weights, void* out, int64_t n, mi_stream_t st) {
  MI_REQUIRE(xs && edge_weights && out && (nin == 2 || nin == 3) && n > 0 && n % 8 == 0, "fastattn_fwd: args");
  FaK k;
  for (int i = 0; i < 3; ++i) { k.x[i] = i < nin ? (const __bf16*)xs[i] : nullptr; k.dx[i] = nullptr; }
  for (int i = 0; i < nin; ++i) MI_REQUIRE(xs[i], "fastattn_fwd: input %d null", i);
  k.g = nullptr; k.out = (__bf16*)out; k.ew = edge_weights; k.part = nullptr; k.nin = nin; k.n8 = n / 8;
  int64_t b = (k.n8 + 255) / 256;
  if (b > FA_BLOCKS) b = FA_BLOCKS;
  hipLaunchKernelGGL(fastattn_fwd_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("fastattn_fwd");
  return MI_OK;
}
extern "C" int mi_fastattn_bwd(const void* const* xs, int nin, const float* edge_weights, const void* g, void* const* dxs,
                               float* dedge, float* ws, int64_t n, mi_stream_t st) {
  MI_REQUIRE(xs && edge_weights && g && dxs && dedge && ws && (nin == 2 || nin == 3) && n > 0 && n % 8 == 0, "fastattn_bwd: args");
  FaK k;
  for (int i = 0; i < 3; ++i) { k.x[i] = i < nin ? (const __bf16*)xs[i] : nullptr; k.dx[i] = i < nin ? (__bf16*)dxs[i] : nullptr; }
  for (int i = 0; i < nin; ++i) MI_REQUIRE(xs[i], "fastattn_bwd: input %d null", i);
  k.g = (const __bf16*)g; k.out = nullptr; k.ew = edge_weights; k.part = ws; k.nin = nin; k.n8 = n / 8;
  int64_t b = (k.n8 + 255) / 256;
  if (b > FA_BLOCKS) b = FA_BLOCKS;
  hipStream_t s = (hipStream_t)st;
  hipLaunchKernelGGL(fastattn_bwd_kernel, dim3((int)b), dim3(256), 0, s, k);
  hipLaunchKernelGGL(fastattn_dw_kernel, dim3(1), dim3(64), 0, s, ws, (int)b, edge_weights, nin, dedge);
  MI_CHECK_LAUNCH("fastattn_bwd");
  return MI_OK;
}
