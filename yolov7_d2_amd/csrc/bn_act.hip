// Train-mode BatchNorm + SiLU (+ residual add) around the conv kernels, NHWC bf16.
// Replaces nn.BatchNorm2d / nn.SiLU of BaseConv (yolov7/modeling/backbone/layers/wrappers.py:76-80,
// eps/momentum patched at yolov7/modeling/meta_arch/yolox.py:85-90) and the Bottleneck
// shortcut add (wrappers.py:119-123).  Batch statistics come from the conv epilogue partials.
// All kernels are pure HBM streams: 16-byte (8 x bf16) accesses, fp32 math.
#include "common.h"

// ------------------------------------------------------------------ forward statistics
__device__ __forceinline__ void block_sum2_d(double& s1, double& s2) {
  // 256-thread block reduction of two doubles (fixed order)
  __shared__ double red[2 * 4];
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave * 2 + 0] = s1;
    red[wave * 2 + 1] = s2;
  }
  __syncthreads();
  s1 = (red[0] + red[2]) + (red[4] + red[6]);
  s2 = (red[1] + red[3]) + (red[5] + red[7]);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int ntiles, int C, int CPad,
                                                         double inv_count, double unbias, const float* gamma,
                                                         const float* beta, float eps, float momentum, float* rmean,
                                                         float* rvar, int64_t* nbt, float* scale, float* shift,
                                                         float* mean_o, float* invstd_o) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  const f32x2* pp = (const f32x2*)partial + c;
  int t = lane;
  for (; t + 768 < ntiles; t += 1024) {  // 4 independent 8-byte loads in flight per thread
    const f32x2 a = pp[(size_t)t * CPad], b = pp[(size_t)(t + 256) * CPad], cc = pp[(size_t)(t + 512) * CPad],
                d = pp[(size_t)(t + 768) * CPad];
    s1 += ((double)a[0] + (double)b[0]) + ((double)cc[0] + (double)d[0]);
    s2 += ((double)a[1] + (double)b[1]) + ((double)cc[1] + (double)d[1]);
  }
  for (; t < ntiles; t += 256) {
    const f32x2 a = pp[(size_t)t * CPad];
    s1 += (double)a[0];
    s2 += (double)a[1];
  }
  block_sum2_d(s1, s2);
  if (lane == 0) {
    const double mean = s1 * inv_count;
    double var = s2 * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma[c], b = beta[c];
    const float sc = (float)((double)g * invstd);
    scale[c] = sc;
    shift[c] = (float)((double)b - mean * (double)g * invstd);
    mean_o[c] = (float)mean;
    invstd_o[c] = (float)invstd;
    if (rmean) {
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * unbias);
    }
    if (c == 0 && nbt) *nbt += 1;
  }
}

extern "C" int mi_bn_finalize(const float* partial, int ntiles, int C, int CPad, int64_t count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float* scale, float* shift, float* mean, float* invstd,
                              mi_stream_t st) {
  MI_REQUIRE(partial && gamma && beta && scale && shift && mean && invstd, "bn_finalize: null");
  MI_REQUIRE(C > 0 && CPad >= C && ntiles > 0 && count > 0, "bn_finalize: sizes");
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)st, partial, ntiles, C, CPad,
                     1.0 / (double)count, unbias, gamma, beta, eps, momentum, running_mean, running_var,
                     num_batches_tracked, scale, shift, mean, invstd);
  MI_CHECK_LAUNCH("bn_finalize");
  return MI_OK;
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rv[c] + eps);
  scale[c] = gamma[c] * invstd;
  shift[c] = beta[c] - rm[c] * gamma[c] * invstd;
}

extern "C" int mi_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int C, float* scale, float* shift,
                                 mi_stream_t st) {
  MI_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_eval_affine: args");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(mi_cdiv(C, 128)), dim3(128), 0, (hipStream_t)st, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  MI_CHECK_LAUNCH("bn_eval_affine");
  return MI_OK;
}

// ------------------------------------------------------------------ forward apply
template <int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const __bf16* __restrict__ y, int ldy,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const __bf16* res, int ldres,
                                                         __bf16* a, int lda, int64_t npix, int C8) {
  const int64_t total = npix * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t pix = idx / C8;
    const int c8 = (int)(idx - pix * C8);
    const bf16x8 v = *(const bf16x8*)(y + pix * ldy + c8 * 8);
    float f[8], o[8];
    unpack8(v, f);
    const f32x4 sc0 = *(const f32x4*)(scale + c8 * 8), sc1 = *(const f32x4*)(scale + c8 * 8 + 4);
    const f32x4 sh0 = *(const f32x4*)(shift + c8 * 8), sh1 = *(const f32x4*)(shift + c8 * 8 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sc = e < 4 ? sc0[e] : sc1[e - 4];
      const float sh = e < 4 ? sh0[e] : sh1[e - 4];
      const float z = f[e] * sc + sh;
      o[e] = ACT ? z * sigmoidf_(z) : z;
    }
    if (res) {
      const bf16x8 r = *(const bf16x8*)(res + pix * ldres + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += (float)r[e];
    }
    *(bf16x8*)(a + pix * lda + c8 * 8) = pack8(o);
  }
}

static int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int mi_bn_act_fwd(const void* y, int ldy, const float* scale, const float* shift, const void* res,
                             int ldres, void* a, int lda, int64_t npix, int C, int act, mi_stream_t st) {
  MI_REQUIRE(y && scale && shift && a, "bn_act_fwd: null");
  MI_REQUIRE(C % 8 == 0 && ldy % 8 == 0 && lda % 8 == 0 && (!res || ldres % 8 == 0), "bn_act_fwd: C/ld %% 8");
  MI_REQUIRE(((uintptr_t)y % 16) == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)res % 16) == 0, "bn_act_fwd: align");
  const int64_t total = npix * (C / 8);
  if (act)
    hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, (const __bf16*)y,
                       ldy, scale, shift, (const __bf16*)res, ldres, (__bf16*)a, lda, npix, C / 8);
  else
    hipLaunchKernelGGL(bn_act_fwd_kernel<0>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, (const __bf16*)y,
                       ldy, scale, shift, (const __bf16*)res, ldres, (__bf16*)a, lda, npix, C / 8);
  MI_CHECK_LAUNCH("bn_act_fwd");
  return MI_OK;
}

// ------------------------------------------------------------------ backward
// dz = da * act'(z), z = y*scale+shift, xhat = (y-mean)*invstd
__device__ __forceinline__ float act_grad(float z, int act) {
  if (!act) return 1.f;
  const float s = sigmoidf_(z);
  return s * (1.f + z * (1.f - s));
}

// pass 1: block partial sums of (dz, dz*xhat) per channel.  256 % C8 == 0 so a thread's channel
// group is fixed (c8 = tid % C8) and its pixel lane is tid / C8.
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const __bf16* __restrict__ da, int ldda,
                                                            const __bf16* __restrict__ y, int ldy,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* partial,
                                                            int64_t npix, int C8) {
  __shared__ float red[256 * 16];
  const int tid = threadIdx.x;
  const int c8 = tid % C8, pl = tid / C8, PL = 256 / C8;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = scale[c8 * 8 + e];
    sh[e] = shift[c8 * 8 + e];
    mu[e] = mean[c8 * 8 + e];
    is[e] = invstd[c8 * 8 + e];
    s1[e] = s2[e] = 0.f;
  }
  for (int64_t pix = (int64_t)blockIdx.x * PL + pl; pix < npix; pix += (int64_t)gridDim.x * PL) {
    const bf16x8 dv = *(const bf16x8*)(da + pix * ldda + c8 * 8);
    const bf16x8 yv = *(const bf16x8*)(y + pix * ldy + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float yy = (float)yv[e];
      const float z = yy * sc[e] + sh[e];
      const float dz = (float)dv[e] * act_grad(z, ACT);
      s1[e] += dz;
      s2[e] += dz * ((yy - mu[e]) * is[e]);
    }
  }
  // red[pl][c8][16]
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[(pl * C8 + c8) * 16 + e] = s1[e];
    red[(pl * C8 + c8) * 16 + 8 + e] = s2[e];
  }
  __syncthreads();
  const int nout = C8 * 16;
  for (int j = tid; j < nout; j += 256) {
    float acc = 0.f;
    for (int q = 0; q < PL; ++q) acc += red[q * nout + j];
    const int cc8 = j / 16, v = j % 16;
    const int c = cc8 * 8 + (v & 7), which = v >> 3;
    partial[((size_t)blockIdx.x * (C8 * 8) + c) * 2 + which] = acc;
  }
}

extern "C" int mi_bn_act_bwd_reduce(const void* da, int ldda, const void* y, int ldy, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, float* partial,
                                    int nblk, int64_t npix, int C, int act, mi_stream_t st) {
  MI_REQUIRE(da && y && scale && shift && mean && invstd && partial, "bn_bwd_reduce: null");
  MI_REQUIRE(C % 8 == 0 && C <= 2048 && (256 % (C / 8)) == 0, "bn_bwd_reduce: C %d (need 256 %% (C/8) == 0)", C);
  MI_REQUIRE(ldda % 8 == 0 && ldy % 8 == 0 && nblk > 0, "bn_bwd_reduce: ld");
  if (act)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(nblk), dim3(256), 0, (hipStream_t)st, (const __bf16*)da, ldda,
                       (const __bf16*)y, ldy, scale, shift, mean, invstd, partial, npix, C / 8);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<0>, dim3(nblk), dim3(256), 0, (hipStream_t)st, (const __bf16*)da, ldda,
                       (const __bf16*)y, ldy, scale, shift, mean, invstd, partial, npix, C / 8);
  MI_CHECK_LAUNCH("bn_bwd_reduce");
  return MI_OK;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                             double inv_count, float* dgamma, float* dbeta, float* c1,
                                                             float* c2) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  const f32x2* pp = (const f32x2*)partial + c;
  int t = lane;
  for (; t + 768 < nblk; t += 1024) {
    const f32x2 a = pp[(size_t)t * C], b = pp[(size_t)(t + 256) * C], cc = pp[(size_t)(t + 512) * C],
                d = pp[(size_t)(t + 768) * C];
    s1 += ((double)a[0] + (double)b[0]) + ((double)cc[0] + (double)d[0]);
    s2 += ((double)a[1] + (double)b[1]) + ((double)cc[1] + (double)d[1]);
  }
  for (; t < nblk; t += 256) {
    const f32x2 a = pp[(size_t)t * C];
    s1 += (double)a[0];
    s2 += (double)a[1];
  }
  block_sum2_d(s1, s2);
  if (lane == 0) {
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    c1[c] = (float)(s1 * inv_count);
    c2[c] = (float)(s2 * inv_count);
  }
}

extern "C" int mi_bn_bwd_finalize(const float* partial, int nblk, int C, int64_t count, float* dgamma, float* dbeta,
                                  float* c1, float* c2, mi_stream_t st) {
  MI_REQUIRE(partial && c1 && c2 && nblk > 0 && C > 0 && count > 0, "bn_bwd_finalize: args");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)st, partial, nblk, C,
                     1.0 / (double)count, dgamma, dbeta, c1, c2);
  MI_CHECK_LAUNCH("bn_bwd_finalize");
  return MI_OK;
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const __bf16* __restrict__ da, int ldda,
                                                           const __bf16* __restrict__ y, int ldy,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ c1, const float* __restrict__ c2,
                                                           __bf16* dy, int lddy, __bf16* dres, int lddres,
                                                           int dres_accum, int64_t npix, int C8) {
  const int64_t total = npix * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t pix = idx / C8;
    const int c8 = (int)(idx - pix * C8);
    const bf16x8 dv = *(const bf16x8*)(da + pix * ldda + c8 * 8);
    const bf16x8 yv = *(const bf16x8*)(y + pix * ldy + c8 * 8);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 * 8 + e;
      const float yy = (float)yv[e];
      const float z = yy * scale[c] + shift[c];
      const float dz = (float)dv[e] * act_grad(z, ACT);
      const float xh = (yy - mean[c]) * invstd[c];
      o[e] = gamma[c] * invstd[c] * (dz - c1[c] - xh * c2[c]);
    }
    *(bf16x8*)(dy + pix * lddy + c8 * 8) = pack8(o);
    if (dres) {
      __bf16* rp = dres + pix * lddres + c8 * 8;
      if (dres_accum) {
        const bf16x8 r = *(const bf16x8*)rp;
        float q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = (float)r[e] + (float)dv[e];
        *(bf16x8*)rp = pack8(q);
      } else {
        *(bf16x8*)rp = dv;
      }
    }
  }
}

extern "C" int mi_bn_act_bwd_apply(const void* da, int ldda, const void* y, int ldy, const float* scale,
                                   const float* shift, const float* mean, const float* invstd, const float* gamma,
                                   const float* c1, const float* c2, void* dy, int lddy, void* dres, int lddres,
                                   int dres_accum, int64_t npix, int C, int act, mi_stream_t st) {
  MI_REQUIRE(da && y && scale && shift && mean && invstd && gamma && c1 && c2 && dy, "bn_bwd_apply: null");
  MI_REQUIRE(C % 8 == 0 && ldda % 8 == 0 && ldy % 8 == 0 && lddy % 8 == 0 && (!dres || lddres % 8 == 0),
             "bn_bwd_apply: C/ld");
  const int64_t total = npix * (C / 8);
  if (act)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st,
                       (const __bf16*)da, ldda, (const __bf16*)y, ldy, scale, shift, mean, invstd, gamma, c1, c2,
                       (__bf16*)dy, lddy, (__bf16*)dres, lddres, dres_accum, npix, C / 8);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st,
                       (const __bf16*)da, ldda, (const __bf16*)y, ldy, scale, shift, mean, invstd, gamma, c1, c2,
                       (__bf16*)dy, lddy, (__bf16*)dres, lddres, dres_accum, npix, C / 8);
  MI_CHECK_LAUNCH("bn_bwd_apply");
  return MI_OK;
}
