// Grid-wide barrier over the resident blocks of one launch + the BatchNorm(train) + activation (+ residual) pass that a
// persistent convolution kernel runs over ITS OWN output tiles behind that barrier (conv1x1_stream.h / conv3x3_ws.h,
// MODE 3): the second half of BaseConv.forward (yolov7/modeling/backbone/layers/wrappers.py:76-83) without a second
// launch.  Used by bn_act.hip (fused BatchNorm backward) and by the two convolution kernels.
#pragma once
#include "common.h"

#define BN_FUS_SPIN_LIMIT (1 << 22)
// Grid-wide barrier over nb resident blocks.  Agent-scope atomics are performed memory-side, one after the other per
// address (~0.1 us each: 512 arrivals on ONE counter cost ~50 us, measured), so arrivals go through a two-level tree -
// MI_BN_BAR_GROUPS group counters (block % groups), whose last arrivers meet on the top counter - and every group waits
// on its own generation word; each word sits on its own 256-byte line.  Word layout (uint32):
//   [0] top arrivals  [2] give-up flag  [64 * (1 + g)] generation of group g  [64 * (1 + G + g)] arrivals of group g
#define BN_BAR_G MI_BN_BAR_GROUPS
__device__ __forceinline__ unsigned* bn_bar_gen(unsigned* bar, int g) { return bar + 64 * (1 + g); }
__device__ __forceinline__ unsigned* bn_bar_cnt(unsigned* bar, int g) { return bar + 64 * (1 + BN_BAR_G + g); }
// No cache maintenance: an agent-scope acquire / release (or __threadfence) writes back and invalidates the XCD's whole L2
// - issued by hundreds of polling blocks that stalls every block still streaming (measured: +90 us per launch).  It is not
// needed here: the only data that crosses blocks are the fp64 sums, added by memory-side atomics that have been
// acknowledged when the block passes its __syncthreads (workgroup release = s_waitcnt vmcnt(0)), and read back after the
// barrier by agent-scope atomic loads, which bypass the non-coherent caches.
// *gave_up (LDS) = 1 when this block's wait timed out: its sums are incomplete and it must poison what it writes.
__device__ __forceinline__ void bn_grid_barrier(unsigned* bar, const unsigned gen0, const int bid, const int nb, int* gave_up) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (see bn_bar_arrive)
  __syncthreads();
  if (threadIdx.x == 0) {
    *gave_up = 0;
    const int G = nb < BN_BAR_G ? nb : BN_BAR_G;
    const int g = bid % G;
    const unsigned gsize = (unsigned)((nb - g + G - 1) / G);
    bool released = false;
    if (__hip_atomic_fetch_add(bn_bar_cnt(bar, g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
      __hip_atomic_store(bn_bar_cnt(bar, g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G - 1u) {
        __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = 0; q < G; ++q)
          __hip_atomic_fetch_add(bn_bar_gen(bar, q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        released = true;
      }
    }
    if (!released) {
      int spins = 0;
      while (__hip_atomic_load(bn_bar_gen(bar, g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > BN_FUS_SPIN_LIMIT) {  // a block that never became resident: report instead of hanging the GPU
          __hip_atomic_store(bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *gave_up = 1;
          break;
        }
      }
    }
  }
  __syncthreads();
}

// The same barrier in two halves, so that the block that arrives LAST can do work for everybody before it releases the
// others (bn_bwd_fused: it folds the accumulator slots into one total per channel - one block's worth of loads instead of
// every block's; see bn_act.hip).  *is_last / *gave_up live in LDS.
__device__ __forceinline__ void bn_bar_arrive(unsigned* bar, const int bid, const int nb, int* gave_up, int* is_last) {
  // every thread's fp64 atomicAdds must have been acknowledged before thread 0 counts the arrival: stated explicitly - the
  // memory model does not promise that a workgroup-scope barrier waits for vmcnt(0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    *gave_up = 0;
    *is_last = 0;
    const int G = nb < BN_BAR_G ? nb : BN_BAR_G;
    const int g = bid % G;
    const unsigned gsize = (unsigned)((nb - g + G - 1) / G);
    if (__hip_atomic_fetch_add(bn_bar_cnt(bar, g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
      __hip_atomic_store(bn_bar_cnt(bar, g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G - 1u) {
        __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *is_last = 1;
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void bn_bar_finish(unsigned* bar, const unsigned gen0, const int bid, const int nb, int* gave_up,
                                              const int* is_last) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last block's agent-scope stores of the folded totals: acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const int G = nb < BN_BAR_G ? nb : BN_BAR_G;
    const int g = bid % G;
    if (*is_last) {
      for (int q = 0; q < G; ++q)
        __hip_atomic_fetch_add(bn_bar_gen(bar, q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      int spins = 0;
      while (__hip_atomic_load(bn_bar_gen(bar, g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > BN_FUS_SPIN_LIMIT) {  // a block that never became resident: report instead of hanging the GPU
          __hip_atomic_store(bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *gave_up = 1;
          break;
        }
      }
    }
  }
  __syncthreads();
}


// ---- BatchNorm(train) forward of a convolution, phase 2 of the convolution launch
struct CBnFwd {
  const __bf16* res;   // residual added after the activation (Bottleneck shortcut) or NULL
  __bf16* a;           // activated output
  const float* gamma;
  const float* beta;
  float* rmean;        // running statistics (may be NULL)
  float* rvar;
  long long* nbt;
  float* scale;        // written for the backward pass: scale / shift / mean / invstd [C]
  float* shift;
  float* mean;
  float* invstd;
  int ldres, lda, act, pad_;
  double inv_count, unbias;
  float eps, momentum;
};

// channel c's (scale, shift) from ITS fp64 accumulators acc[slot * sld + {0, 1}] (acc points at the channel) - the arithmetic of
// bn_act_fwd_body (bn_act.hip), bit for bit.  Agent-scope loads: the sums were added memory-side by every block.
// `writer`: this lane also records scale / shift / mean / invstd and updates the running statistics (one lane per channel
// of the launch); `poison` (NaN after a barrier timeout, else 0) marks results computed from incomplete sums.
__device__ __forceinline__ void cbn_finalize(const CBnFwd& bn, const double* acc, int sld, int nslots, int c, bool writer,
                                             float poison, float& sc_out, float& sh_out) {
  double v1[MI_BN_SLOTS], v2[MI_BN_SLOTS];
#pragma unroll
  for (int k = 0; k < MI_BN_SLOTS; ++k) {
    if (k < nslots) {
      v1[k] = __hip_atomic_load(acc + (size_t)k * sld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v2[k] = __hip_atomic_load(acc + (size_t)k * sld + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      v1[k] = v2[k] = 0.0;
    }
  }
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < MI_BN_SLOTS; ++k) {
    s1 += v1[k];
    s2 += v2[k];
  }
  const double mean = s1 * bn.inv_count;
  double var = s2 * bn.inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)bn.eps);
  const float g = bn.gamma[c], b = bn.beta[c];
  const float sc = (float)((double)g * invstd) + poison;
  const float sh = (float)((double)b - mean * (double)g * invstd) + poison;
  sc_out = sc;
  sh_out = sh;
  if (writer) {
    bn.scale[c] = sc;
    bn.shift[c] = sh;
    bn.mean[c] = (float)mean + poison;
    bn.invstd[c] = (float)invstd + poison;
    if (bn.rmean) {
      bn.rmean[c] = (1.f - bn.momentum) * bn.rmean[c] + bn.momentum * (float)mean;
      bn.rvar[c] = (1.f - bn.momentum) * bn.rvar[c] + bn.momentum * (float)(var * bn.unbias);
    }
    if (c == 0 && bn.nbt) *bn.nbt += 1;
  }
}

// 8 channels of one pixel: a = act(y * scale + shift) (+ res), the expression of bn_act_fwd_body
__device__ __forceinline__ u32x4 cbn_apply8(const u32x4 yv, const float* sc, const float* sh, const int act, const bool has_res,
                                            const u32x4 rv) {
  const bf16x8 v = __builtin_bit_cast(bf16x8, yv), r = __builtin_bit_cast(bf16x8, rv);
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float z = (float)v[e] * sc[e] + sh[e];
    o[e] = act ? z * sigmoidf_(z) : z;
  }
  if (has_res) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += (float)r[e];
  }
  return __builtin_bit_cast(u32x4, pack8(o));
}

// ---- BatchNorm(train) + activation of a convolution's INPUT, applied by the consuming launch ("BN in the consumer")
// The producer left the raw output y and its fp64 (sum, sumsq) accumulators; instead of a bn_act_fwd launch that reads y
// and writes a = act(y * scale + shift), the consumer fetches y tiles (LDS-DMA), every wave rewrites the 16-byte pieces it
// fetched itself IN LDS with cbn_apply8 (the expression and the rounding point of bn_act_fwd_body: the fragments the
// matrix cores see are bit-identical to the two-launch form) and the launch's writer job stores the same values to `a`
// for the later readers of the tensor (weight gradient, residuals, pooling).  BaseConv.forward's norm + act
// (yolov7/modeling/backbone/layers/wrappers.py:76-83) without a launch of its own and without re-reading y.
struct BnXf {
  CBnFwd bn;           // gamma .. invstd, running statistics, eps / momentum / counts; a / lda: the activated tensor; res unused
  const double* acc;   // the producer's accumulators [nslots][sld / 2][2]
  int sld, nslots, C, pad_;
};
static_assert(sizeof(BnXf) == sizeof(mi_bnx) && __builtin_offsetof(BnXf, acc) == __builtin_offsetof(mi_bnx, acc) &&
              __builtin_offsetof(BnXf, bn.inv_count) == __builtin_offsetof(mi_bnx, inv_count) &&
              __builtin_offsetof(BnXf, bn.lda) == __builtin_offsetof(mi_bnx, lda), "mi_bnx (include/mi355_det.h) mirrors BnXf");
// every block: (scale, shift) of all C input channels into LDS tables - the arithmetic of bn_act_fwd_body's prologue, bit for
// bit, with ITS loads: plain 16-byte loads, thread c next to thread c + 1 (the sums were completed by the previous launch;
// cbn_finalize's agent-scope 8-byte atomic loads are for readers BEHIND a grid barrier and cost one L2 request each: with
// every block of a launch asking for the same C x nslots x 16 bytes they took ~9 us of a 512-channel 20x20 launch).
// `rec`: this block also records scale / shift / mean / invstd for the backward pass and updates the running statistics
__device__ __forceinline__ void bnx_tables(const BnXf* __restrict__ xf, float* s_sc, float* s_sh, const int nthreads, const bool rec) {
  const int C = xf->C, nslots = xf->nslots;
  const size_t sld2 = (size_t)(xf->sld >> 1);   // f64x2 elements between slots
  const CBnFwd& bn = xf->bn;
  for (int c = threadIdx.x; c < C; c += nthreads) {
    const f64x2* const ap = (const f64x2*)xf->acc + c;
    f64x2 v[MI_BN_SLOTS];
#pragma unroll
    for (int k = 0; k < MI_BN_SLOTS; ++k) v[k] = ap[(k < nslots ? (size_t)k : 0) * sld2];   // (unconditional: all in flight together)
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < MI_BN_SLOTS; ++k) {
      s1 += k < nslots ? v[k][0] : 0.0;
      s2 += k < nslots ? v[k][1] : 0.0;
    }
    const double mean = s1 * bn.inv_count;
    double var = s2 * bn.inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)bn.eps);
    const float g = bn.gamma[c], b = bn.beta[c];
    const float sc = (float)((double)g * invstd);
    const float sh = (float)((double)b - mean * (double)g * invstd);
    s_sc[c] = sc;
    s_sh[c] = sh;
    if (rec) {
      bn.scale[c] = sc;
      bn.shift[c] = sh;
      bn.mean[c] = (float)mean;
      bn.invstd[c] = (float)invstd;
      if (bn.rmean) {
        bn.rmean[c] = (1.f - bn.momentum) * bn.rmean[c] + bn.momentum * (float)mean;
        bn.rvar[c] = (1.f - bn.momentum) * bn.rvar[c] + bn.momentum * (float)(var * bn.unbias);
      }
      if (c == 0 && bn.nbt) *bn.nbt += 1;
    }
  }
}
// one 16-byte piece (8 channels cg * 8 .. of one pixel), in place; returns what it wrote
__device__ __forceinline__ u32x4 bnx_apply_lds(char* sp, const float* s_sc, const float* s_sh, const int cg, const int act, const bool valid) {
  const u32x4 raw = *(const u32x4*)sp;
  const f32x4 c0 = *(const f32x4*)(s_sc + cg * 8), c1 = *(const f32x4*)(s_sc + cg * 8 + 4);
  const f32x4 h0 = *(const f32x4*)(s_sh + cg * 8), h1 = *(const f32x4*)(s_sh + cg * 8 + 4);
  const float sc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
  const float sh[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  u32x4 o = cbn_apply8(raw, sc, sh, act, false, u32x4{0u, 0u, 0u, 0u});
  if (!valid) o = u32x4{0u, 0u, 0u, 0u};   // zero padding stays zero (act(shift) is not)
  *(u32x4*)sp = o;
  return o;
}
