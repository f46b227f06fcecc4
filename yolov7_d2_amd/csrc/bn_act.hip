// Train-mode BatchNorm + SiLU (+ residual add) around the conv kernels, NHWC bf16.
// Replaces nn.BatchNorm2d / nn.SiLU of BaseConv (yolov7/modeling/backbone/layers/wrappers.py:76-80,
// eps/momentum patched at yolov7/modeling/meta_arch/yolox.py:85-90) and the Bottleneck
// shortcut add (wrappers.py:119-123).
//
// Channel counts: any multiple of 8 up to BN_MAXC.  A block uses TPB = (256 / C8) * C8 of its 256 threads for the
// streaming loops (C8 = C / 8 channel groups), so that a thread keeps ONE channel group across the grid stride and
// its scale / shift live in registers: all 256 threads when C8 is a power of two (YOLOX-s / -l), 240-252 of them for the
// 0.375 / 0.75 / 1.25 width multipliers (C8 = 3, 6, 12, 24, 48, 10, 20, 40, 80, 160).
//
// Three streaming kernels per layer and NO separate "finalize" launches:
//   * the conv epilogue adds its per-tile (sum, sumsq) into fp64 accumulators acc[MI_BN_SLOTS][C][2]
//     (hardware global_atomic_add_f64, slot = tile % MI_BN_SLOTS to spread same-address traffic);
//   * bn_act_fwd reduces the slots in its prologue (every block, 2 x MI_BN_SLOTS loads per channel, fixed
//     order), block 0 also publishes scale/shift/mean/invstd for the backward pass and updates the running
//     statistics exactly as nn.BatchNorm2d does (momentum, unbiased running variance, num_batches_tracked);
//   * bn_bwd_reduce accumulates (sum dz, sum dz*xhat) the same way and bn_bwd_apply finalizes in its prologue.
// fp64 accumulation of fp32 tile sums makes the result independent of the atomic arrival order except in
// astronomically rare rounding ties.  All kernels are HBM streams: 16-byte (8 x bf16) accesses, fp32 math.
#include <string.h>
#include <stdlib.h>
#include "common.h"
#include "conv_bn.h"

#define BN_MAXC 2048
// accumulator layout: [slot][CA][2] with CA = C rounded up to 32 - the CoutPad the conv epilogue adds its tile sums with
#define BN_ACC_C(C) (((C) + 31) / 32 * 32)

// ------------------------------------------------------------------ eval-mode affine
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

// ------------------------------------------------------------------ forward apply (+ statistics finalize)
struct BnFwdK {
  const __bf16* y;
  const __bf16* res;
  __bf16* a;
  const double* acc;  // [MI_BN_SLOTS][C][2] or NULL (eval: scale/shift are inputs)
  const float* gamma;
  const float* beta;
  float* rmean;
  float* rvar;
  int64_t* nbt;
  float* scale;
  float* shift;
  float* mean;
  float* invstd;
  int ldy, ldres, lda, C8, nslots;
  int64_t npix;
  double inv_count, unbias;
  float eps, momentum;
};

template <int ACT, class PK>
__device__ __forceinline__ void bn_act_fwd_body(PK& p, const int bid, const int nb) {
  __shared__ float s_sc[BN_MAXC], s_sh[BN_MAXC];
  const int C = p.C8 * 8, CA = BN_ACC_C(C);
  if (p.acc) {
    for (int c = threadIdx.x; c < C; c += 256) {
      // all slot loads are issued before the first add (a rolled loop serialises 16 L2 round trips: ~4.5 us)
      f64x2 v[MI_BN_SLOTS];
#pragma unroll
      for (int k = 0; k < MI_BN_SLOTS; ++k)
        v[k] = k < p.nslots ? *(const f64x2*)(p.acc + ((size_t)k * CA + c) * 2) : f64x2{0.0, 0.0};
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < MI_BN_SLOTS; ++k) {
        s1 += v[k][0];
        s2 += v[k][1];
      }
      const double mean = s1 * p.inv_count;
      double var = s2 * p.inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)p.eps);
      const float g = p.gamma[c], b = p.beta[c];
      const float sc = (float)((double)g * invstd);
      const float sh = (float)((double)b - mean * (double)g * invstd);
      s_sc[c] = sc;
      s_sh[c] = sh;
      if (bid == 0) {
        p.scale[c] = sc;
        p.shift[c] = sh;
        p.mean[c] = (float)mean;
        p.invstd[c] = (float)invstd;
        if (p.rmean) {
          p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * (float)mean;
          p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * (float)(var * p.unbias);
        }
        if (c == 0 && p.nbt) *p.nbt += 1;
      }
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {
      s_sc[c] = p.scale[c];
      s_sh[c] = p.shift[c];
    }
  }
  __syncthreads();
  const int C8 = p.C8;
  const int64_t total = p.npix * C8;
  // TPB % C8 == 0, so a thread keeps its channel group: hoist its 8 scale/shift pairs
  const int TPB = (256 / C8) * C8;
  if ((int)threadIdx.x >= TPB) return;
  const int c8 = (int)threadIdx.x % C8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = s_sc[c8 * 8 + e];
    sh[e] = s_sh[c8 * 8 + e];
  }
  // 4 grid-stride elements per trip, all loads issued before the first use (one load in flight per wave is
  // latency-bound: ~1 us per trip)
  // TPB % C8 == 0: the thread's items are the pixels pix0 + k * pstep of its channel group - no division per item (a
  // 64-bit one costs as much as the item's arithmetic)
  const int64_t stride = (int64_t)nb * TPB;
  const int64_t pstep = (int64_t)nb * (TPB / C8);
  int64_t pixb = (int64_t)bid * (TPB / C8) + (int)threadIdx.x / C8;
  for (int64_t idx = (int64_t)bid * TPB + threadIdx.x; idx < total; idx += 4 * stride, pixb += 4 * pstep) {
    bf16x8 v[4], r[4];
    int64_t pix[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = idx + u * stride;
      pix[u] = pixb + u * pstep;
      if (i < total) {
        v[u] = *(const bf16x8*)(p.y + pix[u] * p.ldy + c8 * 8);
        if (p.res) r[u] = *(const bf16x8*)(p.res + pix[u] * p.ldres + c8 * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (idx + u * stride >= total) break;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = (float)v[u][e] * sc[e] + sh[e];
        o[e] = ACT ? z * sigmoidf_(z) : z;
      }
      if (p.res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += (float)r[u][e];
      }
      *(bf16x8*)(p.a + pix[u] * p.lda + c8 * 8) = pack8(o);
    }
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const BnFwdK p) {
  bn_act_fwd_body<ACT, const BnFwdK>(p, blockIdx.x, gridDim.x);
}
// several layers in one launch (the FPN levels of the head): the block looks its layer up in a device table
__device__ __forceinline__ int bn_group_job(const int* __restrict__ starts, int njobs) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= starts[j + 1]) ++j;
  return __builtin_amdgcn_readfirstlane(j);
}
template <int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_group_kernel(const BnFwdK* __restrict__ jobs, const int* __restrict__ starts,
                                                               int njobs) {
  const int j = bn_group_job(starts, njobs);
  // constant-address-space view of the table entry: invariant loads, the fields stay in SGPRs across the stores
  typedef const __attribute__((address_space(4))) BnFwdK KC4;
  KC4* pj = (KC4*)(uintptr_t)(jobs + j);
  bn_act_fwd_body<ACT, KC4>(*pj, (int)blockIdx.x - starts[j], starts[j + 1] - starts[j]);
}

// every block pays the statistics prologue (2 x MI_BN_SLOTS fp64 loads + an fp64 rsqrt per channel), so never more than
// 4 blocks per CU; 512 items (two 16-byte items per thread) per block: the 20x20 / 40x40 maps then spread over 400-800 blocks
// instead of 100-200 with the 2048 of rounds 1-3 (same-box A/B, round 4: 5.405 / 5.419 -> 5.384 / 5.385 ms and 5.506 /
// 5.511 / 5.515 -> 5.481 / 5.477 / 5.490 ms per step; 256 and 1024 are in between)
static int ew_blocks(int64_t total) {
  static const int per = getenv("MI_BN_EW_ITEMS") ? atoi(getenv("MI_BN_EW_ITEMS")) : 512;   // items per block (A/B knob)
  const int64_t ipb = per >= 256 ? per : 512;
  int64_t b = (total + ipb - 1) / ipb;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int mi_bn_act_fwd(const void* y, int ldy, const double* stats_acc, int nslots, int64_t count, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, float* scale, float* shift, float* mean, float* invstd,
                             const void* res, int ldres, void* a, int lda, int64_t npix, int C, int act,
                             mi_stream_t st) {
  MI_REQUIRE(y && scale && shift && a, "bn_act_fwd: null");
  MI_REQUIRE(!stats_acc || (gamma && beta && mean && invstd && count > 0), "bn_act_fwd: train mode needs gamma/beta/mean/invstd");
  MI_REQUIRE(C % 8 == 0 && C > 0 && C <= BN_MAXC, "bn_act_fwd: C %d (need C %% 8 == 0, C <= %d)", C, BN_MAXC);
  MI_REQUIRE(ldy % 8 == 0 && lda % 8 == 0 && (!res || ldres % 8 == 0), "bn_act_fwd: ld %% 8");
  MI_REQUIRE(((uintptr_t)y % 16) == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)res % 16) == 0, "bn_act_fwd: align");
  BnFwdK k;
  k.y = (const __bf16*)y; k.res = (const __bf16*)res; k.a = (__bf16*)a; k.acc = stats_acc; k.gamma = gamma;
  k.beta = beta; k.rmean = running_mean; k.rvar = running_var; k.nbt = num_batches_tracked; k.scale = scale;
  k.shift = shift; k.mean = mean; k.invstd = invstd; k.ldy = ldy; k.ldres = ldres; k.lda = lda; k.C8 = C / 8;
  k.npix = npix; k.inv_count = count > 0 ? 1.0 / (double)count : 0.0;
  k.unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  k.eps = eps; k.momentum = momentum;
  k.nslots = (nslots >= 1 && nslots <= MI_BN_SLOTS) ? nslots : MI_BN_SLOTS;
  const int64_t total = npix * (C / 8);
  if (act)
    hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, k);
  else
    hipLaunchKernelGGL(bn_act_fwd_kernel<0>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, k);
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

// pass 1: per-channel sums of (dz, dz*xhat) -> fp64 accumulators dacc[MI_BN_SLOTS][CA][2].  PL = 256 / C8 pixel lanes: a
// thread's channel group is fixed (c8 = tid % C8) and its pixel lane is tid / C8.
struct BnRedK {
  const __bf16* da;
  const __bf16* y;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  double* dacc;
  int ldda, ldy, nslots, C8;
  int64_t npix;
};
template <int ACT, class PK>
__device__ __forceinline__ void bn_bwd_reduce_body(PK& p, const int bid, const int nb) {
  __shared__ float red[256 * 16];
  const int tid = threadIdx.x;
  const int C8 = p.C8;
  const int PL = 256 / C8;                       // pixel lanes; threads beyond PL * C8 only take part in the barriers
  const bool active = tid < PL * C8;
  const int c8 = tid % C8, pl = active ? tid / C8 : 0;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = p.scale[c8 * 8 + e];
    sh[e] = p.shift[c8 * 8 + e];
    mu[e] = p.mean[c8 * 8 + e];
    is[e] = p.invstd[c8 * 8 + e];
    s1[e] = s2[e] = 0.f;
  }
  const int64_t pstride = (int64_t)nb * PL;
  for (int64_t pix0 = active ? (int64_t)bid * PL + pl : p.npix; pix0 < p.npix; pix0 += 4 * pstride) {
    bf16x8 dv[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t pix = pix0 + u * pstride;
      if (pix < p.npix) {
        dv[u] = *(const bf16x8*)(p.da + pix * p.ldda + c8 * 8);
        yv[u] = *(const bf16x8*)(p.y + pix * p.ldy + c8 * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (pix0 + u * pstride >= p.npix) break;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yy = (float)yv[u][e];
        const float z = yy * sc[e] + sh[e];
        const float dz = (float)dv[u][e] * act_grad(z, ACT);
        s1[e] += dz;
        s2[e] += dz * ((yy - mu[e]) * is[e]);
      }
    }
  }
  // red[pl][c8][16]
  if (active) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(pl * C8 + c8) * 16 + e] = s1[e];
      red[(pl * C8 + c8) * 16 + 8 + e] = s2[e];
    }
  }
  __syncthreads();
  const int nout = C8 * 16, C = C8 * 8;
  double* slot = p.dacc + (size_t)(bid % p.nslots) * BN_ACC_C(C) * 2;
  for (int j = tid; j < nout; j += 256) {
    float acc = 0.f;
    for (int q = 0; q < PL; ++q) acc += red[q * nout + j];
    const int cc8 = j / 16, v = j % 16;
    const int c = cc8 * 8 + (v & 7), which = v >> 3;
    atomicAdd(slot + c * 2 + which, (double)acc);
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnRedK p) {
  bn_bwd_reduce_body<ACT, const BnRedK>(p, blockIdx.x, gridDim.x);
}
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_group_kernel(const BnRedK* __restrict__ jobs,
                                                                  const int* __restrict__ starts, int njobs) {
  const int j = bn_group_job(starts, njobs);
  // constant-address-space view of the table entry: invariant loads, the fields stay in SGPRs across the stores
  typedef const __attribute__((address_space(4))) BnRedK KC4;
  KC4* pj = (KC4*)(uintptr_t)(jobs + j);
  bn_bwd_reduce_body<ACT, KC4>(*pj, (int)blockIdx.x - starts[j], starts[j + 1] - starts[j]);
}

extern "C" int mi_bn_act_bwd_reduce(const void* da, int ldda, const void* y, int ldy, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, double* dacc,
                                    int nslots, int nblk, int64_t npix, int C, int act, mi_stream_t st) {
  if (nslots < 1 || nslots > MI_BN_SLOTS) nslots = MI_BN_SLOTS;
  MI_REQUIRE(da && y && scale && shift && mean && invstd && dacc, "bn_bwd_reduce: null");
  MI_REQUIRE(C % 8 == 0 && C > 0 && C <= BN_MAXC, "bn_bwd_reduce: C %d (need C %% 8 == 0, C <= %d)", C, BN_MAXC);
  MI_REQUIRE(ldda % 8 == 0 && ldy % 8 == 0 && nblk > 0, "bn_bwd_reduce: ld");
  BnRedK k;
  k.da = (const __bf16*)da; k.y = (const __bf16*)y; k.scale = scale; k.shift = shift; k.mean = mean; k.invstd = invstd;
  k.dacc = dacc; k.ldda = ldda; k.ldy = ldy; k.nslots = nslots; k.C8 = C / 8; k.npix = npix;
  if (act)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(nblk), dim3(256), 0, (hipStream_t)st, k);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<0>, dim3(nblk), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("bn_bwd_reduce");
  return MI_OK;
}

// pass 2: dy = gamma*invstd*(dz - c1 - xhat*c2) with c1 = sum(dz)/count, c2 = sum(dz*xhat)/count taken from the
// accumulators in the prologue; block 0 also writes dgamma = sum(dz*xhat), dbeta = sum(dz).  optional dres (+)= da.
struct BnBwdK {
  const __bf16* da;
  const __bf16* y;
  __bf16* dy;
  __bf16* dres;
  const double* dacc;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  const float* gamma;
  float* dgamma;
  float* dbeta;
  int ldda, ldy, lddy, lddres, dres_accum, C8, nslots;
  int64_t npix;
  double inv_count;
};

template <int ACT, class PK>
__device__ __forceinline__ void bn_bwd_apply_body(PK& p, const int bid, const int nb) {
  __shared__ float s_c1[BN_MAXC], s_c2[BN_MAXC];
  const int C8 = p.C8, C = C8 * 8, CA = BN_ACC_C(C);
  for (int c = threadIdx.x; c < C; c += 256) {
    f64x2 v[MI_BN_SLOTS];
#pragma unroll
    for (int k = 0; k < MI_BN_SLOTS; ++k)
      v[k] = k < p.nslots ? *(const f64x2*)(p.dacc + ((size_t)k * CA + c) * 2) : f64x2{0.0, 0.0};
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < MI_BN_SLOTS; ++k) {
      s1 += v[k][0];
      s2 += v[k][1];
    }
    s_c1[c] = (float)(s1 * p.inv_count);
    s_c2[c] = (float)(s2 * p.inv_count);
    if (bid == 0) {
      if (p.dbeta) p.dbeta[c] = (float)s1;
      if (p.dgamma) p.dgamma[c] = (float)s2;
    }
  }
  __syncthreads();
  const int64_t total = p.npix * C8;
  const int TPB = (256 / C8) * C8;
  if ((int)threadIdx.x >= TPB) return;
  const int c8 = (int)threadIdx.x % C8;
  float sc[8], sh[8], mu[8], is[8], gi[8], k1[8], k2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    sc[e] = p.scale[c];
    sh[e] = p.shift[c];
    mu[e] = p.mean[c];
    is[e] = p.invstd[c];
    gi[e] = p.gamma[c] * is[e];
    k1[e] = s_c1[c];
    k2[e] = s_c2[c];
  }
  const int64_t stride = (int64_t)nb * TPB;
  const int64_t pstep = (int64_t)nb * (TPB / C8);   // (no division per item: see bn_act_fwd_body)
  int64_t pixb = (int64_t)bid * (TPB / C8) + (int)threadIdx.x / C8;
  for (int64_t idx = (int64_t)bid * TPB + threadIdx.x; idx < total; idx += 2 * stride, pixb += 2 * pstep) {
    bf16x8 dv[2], yv[2], rv[2];
    int64_t pix[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t i = idx + u * stride;
      pix[u] = pixb + u * pstep;
      if (i < total) {
        dv[u] = *(const bf16x8*)(p.da + pix[u] * p.ldda + c8 * 8);
        yv[u] = *(const bf16x8*)(p.y + pix[u] * p.ldy + c8 * 8);
        if (p.dres && p.dres_accum) rv[u] = *(const bf16x8*)(p.dres + pix[u] * p.lddres + c8 * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (idx + u * stride >= total) break;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yy = (float)yv[u][e];
        const float z = yy * sc[e] + sh[e];
        const float dz = (float)dv[u][e] * act_grad(z, ACT);
        const float xh = (yy - mu[e]) * is[e];
        o[e] = gi[e] * (dz - k1[e] - xh * k2[e]);
      }
      *(bf16x8*)(p.dy + pix[u] * p.lddy + c8 * 8) = pack8(o);
      if (p.dres) {
        __bf16* rp = p.dres + pix[u] * p.lddres + c8 * 8;
        if (p.dres_accum) {
          float q[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) q[e] = (float)rv[u][e] + (float)dv[u][e];
          *(bf16x8*)rp = pack8(q);
        } else {
          *(bf16x8*)rp = dv[u];
        }
      }
    }
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdK p) {
  bn_bwd_apply_body<ACT, const BnBwdK>(p, blockIdx.x, gridDim.x);
}
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_group_kernel(const BnBwdK* __restrict__ jobs,
                                                                 const int* __restrict__ starts, int njobs) {
  const int j = bn_group_job(starts, njobs);
  // constant-address-space view of the table entry: invariant loads, the fields stay in SGPRs across the stores
  typedef const __attribute__((address_space(4))) BnBwdK KC4;
  KC4* pj = (KC4*)(uintptr_t)(jobs + j);
  bn_bwd_apply_body<ACT, KC4>(*pj, (int)blockIdx.x - starts[j], starts[j + 1] - starts[j]);
}

extern "C" int mi_bn_act_bwd_apply(const void* da, int ldda, const void* y, int ldy, const float* scale,
                                   const float* shift, const float* mean, const float* invstd, const float* gamma,
                                   const double* dacc, int nslots, int64_t count, float* dgamma, float* dbeta, void* dy,
                                   int lddy,
                                   void* dres, int lddres, int dres_accum, int64_t npix, int C, int act,
                                   mi_stream_t st) {
  MI_REQUIRE(da && y && scale && shift && mean && invstd && gamma && dacc && dy && count > 0, "bn_bwd_apply: null");
  MI_REQUIRE(C % 8 == 0 && C > 0 && C <= BN_MAXC, "bn_bwd_apply: C %d", C);
  MI_REQUIRE(ldda % 8 == 0 && ldy % 8 == 0 && lddy % 8 == 0 && (!dres || lddres % 8 == 0), "bn_bwd_apply: ld");
  BnBwdK k;
  k.da = (const __bf16*)da; k.y = (const __bf16*)y; k.dy = (__bf16*)dy; k.dres = (__bf16*)dres; k.dacc = dacc;
  k.scale = scale; k.shift = shift; k.mean = mean; k.invstd = invstd; k.gamma = gamma; k.dgamma = dgamma;
  k.dbeta = dbeta; k.ldda = ldda; k.ldy = ldy; k.lddy = lddy; k.lddres = lddres; k.dres_accum = dres_accum;
  k.C8 = C / 8; k.npix = npix; k.inv_count = 1.0 / (double)count;
  k.nslots = (nslots >= 1 && nslots <= MI_BN_SLOTS) ? nslots : MI_BN_SLOTS;
  const int64_t total = npix * (C / 8);
  if (act)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, k);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("bn_bwd_apply");
  return MI_OK;
}

// ------------------------------------------------------------------ backward, both passes in ONE launch
// The two-pass form reads da and y twice (5 tensor passes: 2 + 2 reads, 1 write).  The chip holds far more than one
// layer's (da, y) on-chip: 256 CUs x 512 KB of vector registers.  bn_bwd_fused keeps every block resident (grid <= what
// the GPU runs concurrently), loads its share of da and y into REGISTERS (14-16 16-byte items of each per thread),
// adds the block's channel sums to the fp64 accumulators, crosses a grid-wide barrier (agent-scope atomics on a counter
// + generation word), reads the finished sums and writes dy from the registers: 3 tensor passes and one launch.  Items
// beyond the register capacity (the 160x160 / 320x320 maps) are streamed in both phases like the two-pass kernels do.
struct BnFusK {
  const __bf16* da;
  const __bf16* y;
  __bf16* dy;
  __bf16* dres;
  double* dacc;
  unsigned* bar;  // MI_BN_BAR_WORDS barrier words (bn_grid_barrier); [2] is set when a wait gave up (never in a healthy run)
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  const float* gamma;
  float* dgamma;
  float* dbeta;
  int ldda, ldy, lddy, lddres, dres_accum, C8, nslots, rsv_;   // rsv_ 1: round 4's form of phase 2 (every block reads every slot; A/B only)
  int64_t npix;
  double inv_count;
};

// MODE 0: da and y stay in registers as loaded (bf16, 8 VGPRs per item, 16 items); phase 2 recomputes the activation
//         gradient.  MODE 1: dz stays in fp32 registers and y in LDS (14 items, 72 KB per block); phase 2 is 5 flops per
//         element.
template <int ACT, int MODE, class PK>
__device__ __forceinline__ void bn_bwd_fused_body(PK& p, const int bid, const int nb) {
  constexpr int R = MODE ? 14 : 16;
  // LDS: 16 KB block reduction scratch (re-used for the finished sums after the barrier) + (MODE 1) the retained y items
  __shared__ float red[256 * 16];
  __shared__ bf16x8 ylds[MODE ? R : 1][MODE ? 256 : 1];
  float* s_c1 = red;
  float* s_c2 = red + BN_MAXC;
  const int tid = threadIdx.x;
  const int C8 = p.C8, C = C8 * 8, CA = BN_ACC_C(C);
  const int PL = 256 / C8, TPB = PL * C8;
  const bool active = tid < TPB;
  const int c8 = tid % C8, pl = active ? tid / C8 : 0;
  unsigned gen0 = 0;   // thread 0 only
  if (tid == 0)
    gen0 = __hip_atomic_load(bn_bar_gen(p.bar, bid % (nb < BN_BAR_G ? nb : BN_BAR_G)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // item = (pixel, channel group); TPB % C8 == 0: the thread's pixels are pix0 + k * pstep.  Addresses are a uniform
  // 64-bit base (+ k * uniform step) plus ONE 32-bit per-thread byte offset per tensor (host checks the 4 GB bound), so
  // the retained items cost no address registers.
  const int64_t pstep = (int64_t)nb * PL;
  const int64_t pix0 = (int64_t)bid * PL + pl;
  const int nmine = (active && pix0 < p.npix) ? (int)((p.npix - pix0 + pstep - 1) / pstep) : 0;  // this thread's items
  const char* dab = (const char*)p.da;
  const char* yb = (const char*)p.y;
  const unsigned oda = (unsigned)((pix0 * p.ldda + c8 * 8) * 2), oy = (unsigned)((pix0 * p.ldy + c8 * 8) * 2);
  const int64_t sda = pstep * p.ldda * 2, sy = pstep * p.ldy * 2;
  // the residual gradient (dres (+)= da) does not depend on the sums: written in phase 1, while da is at hand
  char* drb = (char*)p.dres;
  const unsigned odr = (unsigned)((pix0 * p.lddres + c8 * 8) * 2);
  const int64_t sdr = pstep * p.lddres * 2;
  const bool racc = drb && p.dres_accum;
  auto pass_res = [&](const int k, const bf16x8& d) {
    if (!drb) return;
    bf16x8* rp = (bf16x8*)(drb + k * sdr + odr);
    if (racc) {
      const bf16x8 r = *rp;
      float q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = (float)r[e] + (float)d[e];
      *rp = pack8(q);
    } else {
      *rp = d;
    }
  };
  bf16x8 dv[R], yv[R];          // MODE 0: live across the barrier
  float dzr[MODE ? R : 1][8];   // MODE 1
  float s1[8], s2[8];
  {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (u < nmine) {
        dv[u] = *(const bf16x8*)(dab + u * sda + oda);
        yv[u] = *(const bf16x8*)(yb + u * sy + oy);
      }
    }
    float sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = p.scale[c8 * 8 + e];
      sh[e] = p.shift[c8 * 8 + e];
      mu[e] = p.mean[c8 * 8 + e];
      is[e] = p.invstd[c8 * 8 + e];
      s1[e] = s2[e] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (u < nmine) {
        pass_res(u, dv[u]);
        if (MODE) ylds[MODE ? u : 0][MODE ? tid : 0] = yv[u];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yy = (float)yv[u][e];
          const float z = yy * sc[e] + sh[e];
          const float dz = (float)dv[u][e] * act_grad(z, ACT);
          if (MODE) dzr[MODE ? u : 0][e] = dz;
          s1[e] += dz;
          s2[e] += dz * ((yy - mu[e]) * is[e]);
        }
      }
    }
    // beyond the on-chip capacity: stream (4 items in flight)
    for (int k0 = R; k0 < nmine; k0 += 4) {
      bf16x8 d4[4], y4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k0 + u < nmine) {
          d4[u] = *(const bf16x8*)(dab + (k0 + u) * sda + oda);
          y4[u] = *(const bf16x8*)(yb + (k0 + u) * sy + oy);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k0 + u >= nmine) break;
        pass_res(k0 + u, d4[u]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yy = (float)y4[u][e];
          const float z = yy * sc[e] + sh[e];
          const float dz = (float)d4[u][e] * act_grad(z, ACT);
          s1[e] += dz;
          s2[e] += dz * ((yy - mu[e]) * is[e]);
        }
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
  {
    const int nout = C8 * 16;
    double* slot = p.dacc + (size_t)(bid % p.nslots) * CA * 2;
    for (int j = tid; j < nout; j += 256) {
      float acc = 0.f;
      for (int q = 0; q < PL; ++q) acc += red[q * nout + j];
      const int cc8 = j / 16, v = j % 16;
      atomicAdd(slot + (cc8 * 8 + (v & 7)) * 2 + (v >> 3), (double)acc);
    }
  }
  __shared__ int s_gave_up, s_last;
  // The grid barrier in two halves (conv_bn.h): the block that arrives last folds the nslots accumulator slots of every
  // channel into slot 0 (same fixed order as before: identical totals) and records dgamma / dbeta BEFORE it releases the
  // others, which then read two fp64 totals per channel.  Round 4 had EVERY block read every slot after the barrier:
  // C x nslots x 2 agent-scope 8-byte loads per block (they bypass the XCD's L2: one memory-side request each), 2 M requests
  // for a 256-channel layer on 512 blocks - measured on the same pattern in round 5's first conv prologue: ~9 us for
  // C = 512 (profiles/r05_bn_bwd_last_arriver_ab.txt).
  bn_bar_arrive(p.bar, bid, nb, &s_gave_up, &s_last);   // (its leading __syncthreads also ends the reads of red[])
  const bool legacy = p.rsv_ != 0;
  if (s_last && !legacy) {
    for (int c = tid; c < C; c += 256) {
      double v1[MI_BN_SLOTS], v2[MI_BN_SLOTS];
#pragma unroll
      for (int k = 0; k < MI_BN_SLOTS; ++k) {
        if (k < p.nslots) {
          v1[k] = __hip_atomic_load(p.dacc + ((size_t)k * CA + c) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v2[k] = __hip_atomic_load(p.dacc + ((size_t)k * CA + c) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          v1[k] = v2[k] = 0.0;
        }
      }
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int k = 0; k < MI_BN_SLOTS; ++k) {
        t1 += v1[k];
        t2 += v2[k];
      }
      __hip_atomic_store(p.dacc + (size_t)c * 2, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.dacc + (size_t)c * 2 + 1, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p.dbeta) p.dbeta[c] = (float)t1;
      if (p.dgamma) p.dgamma[c] = (float)t2;
    }
  }
  bn_bar_finish(p.bar, gen0, bid, nb, &s_gave_up, &s_last);
  // a block whose wait timed out holds incomplete sums: everything it writes from here on is NaN (dy of its items), and
  // the host finds word 2 of the barrier record set (Plan.check_bn_barriers)
  const float poison = s_gave_up ? __builtin_nanf("") : 0.f;
  // ---- phase 2: the finished totals (agent-scope loads: written by a block of another XCD)
  for (int c = tid; c < C; c += 256) {
    double t1 = __hip_atomic_load(p.dacc + (size_t)c * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double t2 = __hip_atomic_load(p.dacc + (size_t)c * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (legacy) {   // (slot 0 still holds its own partial: add the others)
      for (int k = 1; k < p.nslots; ++k) {
        t1 += __hip_atomic_load(p.dacc + ((size_t)k * CA + c) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2 += __hip_atomic_load(p.dacc + ((size_t)k * CA + c) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (bid == 0) {
        if (p.dbeta) p.dbeta[c] = (float)t1 + poison;
        if (p.dgamma) p.dgamma[c] = (float)t2 + poison;
      }
    }
    s_c1[c] = (float)(t1 * p.inv_count) + poison;
    s_c2[c] = (float)(t2 * p.inv_count) + poison;
  }
  __syncthreads();
  if (!active) return;
  float sc[8], sh[8], mu[8], is[8], gi[8], k1[8], k2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    sc[e] = p.scale[c];
    sh[e] = p.shift[c];
    mu[e] = p.mean[c];
    is[e] = p.invstd[c];
    gi[e] = p.gamma[c] * is[e];
    k1[e] = s_c1[c];
    k2[e] = s_c2[c];
  }
  char* dyb = (char*)p.dy;
  const unsigned ody = (unsigned)((pix0 * p.lddy + c8 * 8) * 2);
  const int64_t sdy = pstep * p.lddy * 2;
  auto full = [&](const int k, const bf16x8& d, const bf16x8& yy8) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float yy = (float)yy8[e];
      const float z = yy * sc[e] + sh[e];
      const float dz = (float)d[e] * act_grad(z, ACT);
      const float xh = (yy - mu[e]) * is[e];
      o[e] = gi[e] * (dz - k1[e] - xh * k2[e]);
    }
    *(bf16x8*)(dyb + k * sdy + ody) = pack8(o);
  };
#pragma unroll
  for (int u = 0; u < R; ++u) {
    if (u < nmine) {
      if (MODE) {
        const bf16x8 yk = ylds[MODE ? u : 0][MODE ? tid : 0];
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = ((float)yk[e] - mu[e]) * is[e];
          o[e] = gi[e] * (dzr[MODE ? u : 0][e] - k1[e] - xh * k2[e]);
        }
        *(bf16x8*)(dyb + u * sdy + ody) = pack8(o);
      } else {
        full(u, dv[u], yv[u]);
      }
    }
  }
  for (int k0 = R; k0 < nmine; k0 += 2) {
    bf16x8 d2[2], y2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (k0 + u < nmine) {
        d2[u] = *(const bf16x8*)(dab + (k0 + u) * sda + oda);
        y2[u] = *(const bf16x8*)(yb + (k0 + u) * sy + oy);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (k0 + u < nmine) full(k0 + u, d2[u], y2[u]);
  }
}

template <int ACT, int MODE>
__global__ __launch_bounds__(256, 2) void bn_bwd_fused_kernel(const BnFusK p) {
  bn_bwd_fused_body<ACT, MODE, const BnFusK>(p, blockIdx.x, gridDim.x);
}
template <int ACT, int MODE>
__global__ __launch_bounds__(256, 2) void bn_bwd_fused_group_kernel(const BnFusK* __restrict__ jobs,
                                                                    const int* __restrict__ starts, int njobs) {
  const int j = bn_group_job(starts, njobs);
  typedef const __attribute__((address_space(4))) BnFusK KC4;
  KC4* pj = (KC4*)(uintptr_t)(jobs + j);
  bn_bwd_fused_body<ACT, MODE, KC4>(*pj, (int)blockIdx.x - starts[j], starts[j + 1] - starts[j]);
}
// MI_BN_BWD_LAST=0: the round-4 form of the phase between the two halves (A/B runs); read once
static int bn_fused_every_block_reads() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MI_BN_BWD_LAST");
    v = (e && atoi(e) == 0) ? 1 : 0;
  }
  return v;
}
static int bn_fused_mode() {
  static int m = -1;
  if (m < 0) {
    const char* e = getenv("MI_BN_FUSED_MODE");
    m = e ? atoi(e) : 1;
    if (m != 0) m = 1;
  }
  return m;
}

// number of 256-thread blocks of the fused kernel the GPU keeps resident at once (the grid barrier needs all of them)
static int g_fused_cap_user = 0;
static int bn_fused_capacity_hw() {
  static int cap = 0;
  if (cap) return cap;
  int dev = 0, ncu = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      ncu <= 0) {
    (void)hipGetLastError();
    return 512;  // no device (host-side planning of a dry run): the MI355X figure, 256 CUs x 2
  }
  // the grid size is shared by all four instantiations (MI_BN_FUSED_MODE, activation on / off) and their grouped forms: the
  // one with the fewest resident blocks per CU bounds it (MODE 0 keeps 32 16-byte items in registers)
  per = 2;
  const void* variants[] = {(const void*)bn_bwd_fused_kernel<1, 1>, (const void*)bn_bwd_fused_kernel<1, 0>,
                            (const void*)bn_bwd_fused_kernel<0, 1>, (const void*)bn_bwd_fused_kernel<0, 0>,
                            (const void*)bn_bwd_fused_group_kernel<1, 1>, (const void*)bn_bwd_fused_group_kernel<1, 0>,
                            (const void*)bn_bwd_fused_group_kernel<0, 1>, (const void*)bn_bwd_fused_group_kernel<0, 0>};
  for (const void* fn : variants) {
    int q = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, fn, 256, 0) != hipSuccess || q < 1) q = 1;
    if (q < per) per = q;
  }
  cap = ncu * per;
  return cap;
}
static int bn_fused_capacity() {
  const int hw = bn_fused_capacity_hw();
  return (g_fused_cap_user > 0 && g_fused_cap_user < hw) ? g_fused_cap_user : hw;
}
extern "C" int mi_bn_fused_set_capacity(int blocks) {
  g_fused_cap_user = blocks > 0 ? blocks : 0;
  return bn_fused_capacity();
}
static bool bn_fused_fits(int64_t npix, int ldda, int ldy, int lddy, int lddres) {
  int ld = ldda > ldy ? ldda : ldy;
  if (lddy > ld) ld = lddy;
  if (lddres > ld) ld = lddres;
  return npix * ld * 2 < (1LL << 32) - 4096;   // 32-bit per-thread byte offsets
}
static int bn_fused_blocks(int64_t npix, int C8, int cap) {
  // at least ~4 items per thread (the barrier costs one atomic round trip per block), never more than `cap` blocks
  const int64_t TPB = (256 / C8) * C8;
  static const int ipt_e = getenv("MI_BN_FUSED_ITEMS") ? atoi(getenv("MI_BN_FUSED_ITEMS")) : 4;   // A/B knob
  const int64_t ipt = ipt_e >= 1 && ipt_e <= 16 ? ipt_e : 4;
  int64_t b = (npix * C8 + TPB * ipt - 1) / (TPB * ipt);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int mi_bn_act_bwd_fused(const void* da, int ldda, const void* y, int ldy, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, const float* gamma, double* dacc, int nslots,
                                   int64_t count, float* dgamma, float* dbeta, void* dy, int lddy, void* dres, int lddres,
                                   int dres_accum, int64_t npix, int C, int act, uint32_t* barrier_words, mi_stream_t st) {
  MI_REQUIRE(da && y && scale && shift && mean && invstd && gamma && dacc && dy && barrier_words && count > 0, "bn_bwd_fused: null");
  MI_REQUIRE(C % 8 == 0 && C > 0 && C <= BN_MAXC, "bn_bwd_fused: C %d", C);
  MI_REQUIRE(ldda % 8 == 0 && ldy % 8 == 0 && lddy % 8 == 0 && (!dres || lddres % 8 == 0), "bn_bwd_fused: ld");
  MI_REQUIRE(bn_fused_fits(npix, ldda, ldy, lddy, dres ? lddres : 0), "bn_bwd_fused: tensors beyond 4 GB (use reduce + apply)");
  BnFusK k;
  k.da = (const __bf16*)da; k.y = (const __bf16*)y; k.dy = (__bf16*)dy; k.dres = (__bf16*)dres; k.dacc = dacc;
  k.bar = barrier_words; k.scale = scale; k.shift = shift; k.mean = mean; k.invstd = invstd; k.gamma = gamma;
  k.dgamma = dgamma; k.dbeta = dbeta; k.ldda = ldda; k.ldy = ldy; k.lddy = lddy; k.lddres = lddres;
  k.dres_accum = dres_accum; k.C8 = C / 8; k.rsv_ = bn_fused_every_block_reads(); k.npix = npix; k.inv_count = 1.0 / (double)count;
  k.nslots = (nslots >= 1 && nslots <= MI_BN_SLOTS) ? nslots : MI_BN_SLOTS;
  const int nb = bn_fused_blocks(npix, C / 8, bn_fused_capacity());
  const int mode = bn_fused_mode();
  if (act && mode) hipLaunchKernelGGL((bn_bwd_fused_kernel<1, 1>), dim3(nb), dim3(256), 0, (hipStream_t)st, k);
  else if (act) hipLaunchKernelGGL((bn_bwd_fused_kernel<1, 0>), dim3(nb), dim3(256), 0, (hipStream_t)st, k);
  else if (mode) hipLaunchKernelGGL((bn_bwd_fused_kernel<0, 1>), dim3(nb), dim3(256), 0, (hipStream_t)st, k);
  else hipLaunchKernelGGL((bn_bwd_fused_kernel<0, 0>), dim3(nb), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("bn_bwd_fused");
  return MI_OK;
}

// ------------------------------------------------------------------ grouped launches
// one launch for the same BatchNorm pass of several independent layers (the FPN levels of the head)
extern "C" int mi_bn_group_plan(int kind, const mi_bn_job* jobs, int n, void* table_host, int64_t table_cap,
                                mi_bn_group* meta) {
  MI_REQUIRE(jobs && meta && n >= 1 && n <= MI_BN_MAX_GROUP && kind >= 0 && kind <= 3, "bn_group_plan: args");
  int starts[MI_BN_MAX_GROUP + 1];
  starts[0] = 0;
  BnFwdK kf[MI_BN_MAX_GROUP];
  BnRedK kr[MI_BN_MAX_GROUP];
  BnBwdK ka[MI_BN_MAX_GROUP];
  BnFusK ku[MI_BN_MAX_GROUP];
  // fused backward: the whole launch must be resident at once - the jobs share the capacity in proportion to their size
  const int cap = kind == 3 ? bn_fused_capacity() : 0;
  int64_t want = 0;
  if (kind == 3) {
    MI_REQUIRE(n <= cap, "bn_group_plan: more fused jobs than resident blocks");
    for (int j = 0; j < n; ++j)
      if (jobs[j].C > 0 && jobs[j].C % 8 == 0) want += bn_fused_blocks(jobs[j].npix, jobs[j].C / 8, cap);
  }
  for (int j = 0; j < n; ++j) {
    const mi_bn_job& b = jobs[j];
    const int C = b.C;
    MI_REQUIRE(C % 8 == 0 && C > 0 && C <= BN_MAXC && b.npix > 0, "bn_group_plan: job %d C %d", j, C);
    MI_REQUIRE(b.act == jobs[0].act, "bn_group_plan: jobs must agree on the activation");
    const int nslots = (b.nslots >= 1 && b.nslots <= MI_BN_SLOTS) ? b.nslots : MI_BN_SLOTS;
    const int64_t total = b.npix * (C / 8);
    int nb = ew_blocks(total);
    if (kind == 0) {
      MI_REQUIRE(b.y && b.a && b.scale && b.shift && b.ldy % 8 == 0 && b.lda % 8 == 0 && (!b.res || b.ldres % 8 == 0),
                 "bn_group_plan: fwd job %d", j);
      BnFwdK& k = kf[j];
      k.y = (const __bf16*)b.y; k.res = (const __bf16*)b.res; k.a = (__bf16*)b.a; k.acc = b.acc; k.gamma = b.gamma;
      k.beta = b.beta; k.rmean = b.rmean; k.rvar = b.rvar; k.nbt = b.nbt; k.scale = b.scale; k.shift = b.shift;
      k.mean = b.mean; k.invstd = b.invstd; k.ldy = b.ldy; k.ldres = b.ldres; k.lda = b.lda; k.C8 = C / 8;
      k.npix = b.npix; k.nslots = nslots;
      if (b.acc) {
        MI_REQUIRE(b.gamma && b.beta && b.mean && b.invstd && b.count > 0, "bn_group_plan: train-mode fwd job %d", j);
        k.inv_count = 1.0 / (double)b.count;
        k.unbias = b.count > 1 ? (double)b.count / (double)(b.count - 1) : 1.0;
      } else {
        k.inv_count = 0.0; k.unbias = 1.0;
      }
      k.eps = b.eps; k.momentum = b.momentum;
    } else if (kind == 1) {
      MI_REQUIRE(b.da && b.y && b.scale && b.shift && b.mean && b.invstd && b.acc && b.nblk > 0, "bn_group_plan: reduce job %d", j);
      BnRedK& k = kr[j];
      k.da = (const __bf16*)b.da; k.y = (const __bf16*)b.y; k.scale = b.scale; k.shift = b.shift; k.mean = b.mean;
      k.invstd = b.invstd; k.dacc = b.acc; k.ldda = b.ldda; k.ldy = b.ldy; k.nslots = nslots; k.C8 = C / 8; k.npix = b.npix;
      nb = b.nblk;
    } else if (kind == 3) {
      MI_REQUIRE(b.da && b.y && b.scale && b.shift && b.mean && b.invstd && b.gamma && b.acc && b.dy && b.bar && b.count > 0,
                 "bn_group_plan: fused job %d", j);
      MI_REQUIRE(bn_fused_fits(b.npix, b.ldda, b.ldy, b.lddy, b.dres ? b.lddres : 0), "bn_group_plan: fused job %d beyond 4 GB", j);
      BnFusK& k = ku[j];
      k.da = (const __bf16*)b.da; k.y = (const __bf16*)b.y; k.dy = (__bf16*)b.dy; k.dres = (__bf16*)b.dres; k.dacc = b.acc;
      k.bar = b.bar; k.scale = b.scale; k.shift = b.shift; k.mean = b.mean; k.invstd = b.invstd; k.gamma = b.gamma;
      k.dgamma = b.dgamma; k.dbeta = b.dbeta; k.ldda = b.ldda; k.ldy = b.ldy; k.lddy = b.lddy; k.lddres = b.lddres;
      k.dres_accum = b.dres_accum; k.C8 = C / 8; k.rsv_ = bn_fused_every_block_reads(); k.npix = b.npix; k.inv_count = 1.0 / (double)b.count;
      k.nslots = nslots;
      nb = bn_fused_blocks(b.npix, C / 8, cap);
      if (want > cap) {
        nb = (int)((int64_t)nb * (cap - n) / want);   // sum <= cap - n, + 1 each below
        nb += 1;
      }
    } else {
      MI_REQUIRE(b.da && b.y && b.scale && b.shift && b.mean && b.invstd && b.gamma && b.acc && b.dy && b.count > 0,
                 "bn_group_plan: apply job %d", j);
      BnBwdK& k = ka[j];
      k.da = (const __bf16*)b.da; k.y = (const __bf16*)b.y; k.dy = (__bf16*)b.dy; k.dres = (__bf16*)b.dres; k.dacc = b.acc;
      k.scale = b.scale; k.shift = b.shift; k.mean = b.mean; k.invstd = b.invstd; k.gamma = b.gamma; k.dgamma = b.dgamma;
      k.dbeta = b.dbeta; k.ldda = b.ldda; k.ldy = b.ldy; k.lddy = b.lddy; k.lddres = b.lddres; k.dres_accum = b.dres_accum;
      k.C8 = C / 8; k.npix = b.npix; k.inv_count = 1.0 / (double)b.count; k.nslots = nslots;
    }
    starts[j + 1] = starts[j] + nb;
  }
  const size_t rec = kind == 0 ? sizeof(BnFwdK) : (kind == 1 ? sizeof(BnRedK) : (kind == 2 ? sizeof(BnBwdK) : sizeof(BnFusK)));
  const void* src = kind == 0 ? (const void*)kf : (kind == 1 ? (const void*)kr : (kind == 2 ? (const void*)ka : (const void*)ku));
  meta->kind = kind; meta->njobs = n; meta->nblocks = starts[n]; meta->act = jobs[0].act;
  meta->starts_off = (int64_t)(rec * n);
  meta->table_bytes = meta->starts_off + (int64_t)sizeof(int) * (n + 1);
  if (table_host) {
    MI_REQUIRE(table_cap >= meta->table_bytes, "bn_group_plan: table too small");
    memcpy(table_host, src, rec * n);
    memcpy((char*)table_host + meta->starts_off, starts, sizeof(int) * (n + 1));
  }
  return MI_OK;
}

extern "C" int mi_bn_group_run(const mi_bn_group* m, const void* table_dev, mi_stream_t st) {
  MI_REQUIRE(m && table_dev && m->njobs >= 1 && m->nblocks >= 1, "bn_group_run: args");
  const int* starts = (const int*)((const char*)table_dev + m->starts_off);
  hipStream_t s = (hipStream_t)st;
  const dim3 g((unsigned)m->nblocks), b(256);
  if (m->kind == 0) {
    if (m->act) hipLaunchKernelGGL(bn_act_fwd_group_kernel<1>, g, b, 0, s, (const BnFwdK*)table_dev, starts, m->njobs);
    else hipLaunchKernelGGL(bn_act_fwd_group_kernel<0>, g, b, 0, s, (const BnFwdK*)table_dev, starts, m->njobs);
  } else if (m->kind == 1) {
    if (m->act) hipLaunchKernelGGL(bn_bwd_reduce_group_kernel<1>, g, b, 0, s, (const BnRedK*)table_dev, starts, m->njobs);
    else hipLaunchKernelGGL(bn_bwd_reduce_group_kernel<0>, g, b, 0, s, (const BnRedK*)table_dev, starts, m->njobs);
  } else if (m->kind == 2) {
    if (m->act) hipLaunchKernelGGL(bn_bwd_apply_group_kernel<1>, g, b, 0, s, (const BnBwdK*)table_dev, starts, m->njobs);
    else hipLaunchKernelGGL(bn_bwd_apply_group_kernel<0>, g, b, 0, s, (const BnBwdK*)table_dev, starts, m->njobs);
  } else {
    MI_REQUIRE(m->nblocks <= bn_fused_capacity(), "bn_group_run: fused launch of %d blocks exceeds the resident capacity", m->nblocks);
    const int mode = bn_fused_mode();
    const BnFusK* jt = (const BnFusK*)table_dev;
    if (m->act && mode) hipLaunchKernelGGL((bn_bwd_fused_group_kernel<1, 1>), g, b, 0, s, jt, starts, m->njobs);
    else if (m->act) hipLaunchKernelGGL((bn_bwd_fused_group_kernel<1, 0>), g, b, 0, s, jt, starts, m->njobs);
    else if (mode) hipLaunchKernelGGL((bn_bwd_fused_group_kernel<0, 1>), g, b, 0, s, jt, starts, m->njobs);
    else hipLaunchKernelGGL((bn_bwd_fused_group_kernel<0, 0>), g, b, 0, s, jt, starts, m->njobs);
  }
  MI_CHECK_LAUNCH("bn_group");
  return MI_OK;
}
