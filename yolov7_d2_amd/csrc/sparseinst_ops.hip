// SparseInst (BASELINE.json config 5) - the ops that are not convolutions:
//   * bilinear resize (align_corners = False) of bf16 NHWC maps, forward + backward: F.interpolate of the FPN outputs
//     (transcoders/encoder_sparseinst.py:120-124), of the PPM priors (:63-70) and of the predicted masks / IAMs
//     (transcoders/decoder_sparseinst.py:141-160)
//   * the mask part of SparseInstCriterion (loss/sparseinst_loss.py:123-187): for every matched (prediction, target)
//     pair ONE pass over the pair's pixels yields the BCE sum, the dice terms and the thresholded mask IoU
//     (compute_mask_iou :19-28, dice_loss :38-47); a second pass writes d(loss)/d(mask logits).
// The convolutions, the IAM aggregation / dynamic mask "bmm"s and the matcher's dice-score matmul run on the conv /
// wgrad MFMA kernels (modeling/sparseinst.py), the assignment on mi_lsap.
#include <stdlib.h>
#include <string.h>
#include "common.h"

// ---------------------------------------------------------------- bilinear resize, NHWC bf16
struct ResizeK {
  const __bf16* x;
  __bf16* y;
  float* acc;     // backward: fp32 [N][H][W][C] accumulator
  int ldx, ldy, N, H, W, Ho, Wo, C8;
  float sh, sw;   // H / Ho, W / Wo
};

__device__ __forceinline__ void src_index(int o, float scale, int n, int* i0, int* i1, float* l1) {
  // ATen area_pixel_compute_source_index (align_corners = False): src = scale * (dst + 0.5) - 0.5, clamped at 0
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  const int a = (int)s;
  *i0 = a < n - 1 ? a : n - 1;
  *i1 = a < n - 1 ? a + 1 : n - 1;
  *l1 = s - (float)a;
}

__global__ __launch_bounds__(256) void resize_fwd_kernel(const ResizeK p) {
  const int64_t total = (int64_t)p.N * p.Ho * p.Wo * p.C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % p.C8);
    int64_t r = idx / p.C8;
    const int ox = (int)(r % p.Wo); r /= p.Wo;
    const int oy = (int)(r % p.Ho);
    const int n = (int)(r / p.Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(oy, p.sh, p.H, &y0, &y1, &ly);
    src_index(ox, p.sw, p.W, &x0, &x1, &lx);
    const __bf16* b = p.x + ((int64_t)n * p.H * p.W) * p.ldx + c8 * 8;
    const bf16x8 v00 = *(const bf16x8*)(b + ((int64_t)y0 * p.W + x0) * p.ldx), v01 = *(const bf16x8*)(b + ((int64_t)y0 * p.W + x1) * p.ldx);
    const bf16x8 v10 = *(const bf16x8*)(b + ((int64_t)y1 * p.W + x0) * p.ldx), v11 = *(const bf16x8*)(b + ((int64_t)y1 * p.W + x1) * p.ldx);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = w00 * (float)v00[e] + w01 * (float)v01[e] + w10 * (float)v10[e] + w11 * (float)v11[e];
    *(bf16x8*)(p.y + (((int64_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy + c8 * 8) = pack8(o);
  }
}

// backward: every output-gradient pixel adds its four weighted contributions into the fp32 accumulator (hardware
// float atomics), a second launch rounds the accumulator to bf16
__global__ __launch_bounds__(256) void resize_bwd_kernel(const ResizeK p) {   // p.x = dy (Ho x Wo), p.acc = dx accumulator
  const int64_t total = (int64_t)p.N * p.Ho * p.Wo * p.C8;
  const int C = p.C8 * 8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % p.C8);
    int64_t r = idx / p.C8;
    const int ox = (int)(r % p.Wo); r /= p.Wo;
    const int oy = (int)(r % p.Ho);
    const int n = (int)(r / p.Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(oy, p.sh, p.H, &y0, &y1, &ly);
    src_index(ox, p.sw, p.W, &x0, &x1, &lx);
    const bf16x8 g = *(const bf16x8*)(p.x + (((int64_t)n * p.Ho + oy) * p.Wo + ox) * p.ldx + c8 * 8);
    float* a = p.acc + ((int64_t)n * p.H * p.W) * C + c8 * 8;
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float ge = (float)g[e];
      atomicAdd(a + ((int64_t)y0 * p.W + x0) * C + e, w00 * ge);
      atomicAdd(a + ((int64_t)y0 * p.W + x1) * C + e, w01 * ge);
      atomicAdd(a + ((int64_t)y1 * p.W + x0) * C + e, w10 * ge);
      atomicAdd(a + ((int64_t)y1 * p.W + x1) * C + e, w11 * ge);
    }
  }
}
// gather form: an input pixel collects, in a fixed order, the weighted out-gradients of every output pixel whose bilinear
// footprint contains it - no atomics, no fp32 accumulator tensor, no rounding pass, deterministic.  (The scatter form above
// was 32 float atomics per output element: 1.03 ms per call, a quarter of the SparseInst step.)  The candidate output rows /
// columns of input index i are those with source coordinate in (i - 1, i + 1); each candidate's (i0, i1, l1) is recomputed
// with the forward's own src_index, so the two passes agree bit for bit on which pixels touch which.
__device__ __forceinline__ float resize_w(int o, float scale, int n, int i) {
  int i0, i1;
  float l1;
  src_index(o, scale, n, &i0, &i1, &l1);
  return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}
__global__ __launch_bounds__(256) void resize_bwd_gather_kernel(const ResizeK p) {   // p.x = dy (Ho x Wo, ldx), p.y = dx (H x W, ldy)
  const int64_t total = (int64_t)p.N * p.H * p.W * p.C8;
  const float ish = 1.f / p.sh, isw = 1.f / p.sw;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % p.C8);
    int64_t r = idx / p.C8;
    const int ix = (int)(r % p.W); r /= p.W;
    const int iy = (int)(r % p.H);
    const int n = (int)(r / p.H);
    int oy0 = (int)floorf(((float)iy - 0.5f) * ish - 0.5f) - 1, oy1 = (int)ceilf(((float)iy + 1.5f) * ish - 0.5f) + 1;
    int ox0 = (int)floorf(((float)ix - 0.5f) * isw - 0.5f) - 1, ox1 = (int)ceilf(((float)ix + 1.5f) * isw - 0.5f) + 1;
    oy0 = oy0 < 0 ? 0 : oy0; ox0 = ox0 < 0 ? 0 : ox0;
    oy1 = oy1 > p.Ho - 1 ? p.Ho - 1 : oy1; ox1 = ox1 > p.Wo - 1 ? p.Wo - 1 : ox1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const __bf16* g0 = p.x + ((int64_t)n * p.Ho * p.Wo) * p.ldx + c8 * 8;
    // the column weights once per input pixel (they do not depend on the row): the candidate window carries a margin of
    // zero-weight columns on both sides, trimmed here; windows wider than RW_MAX (the 1 x 1 .. 6 x 6 pooled maps blown up
    // to 20 x 20) keep the per-candidate evaluation.  Same products in the same order as before: bit-identical.
    constexpr int RW_MAX = 12;
    float wxs[RW_MAX];
    const bool cached = ox1 - ox0 < RW_MAX;
    if (cached) {
#pragma unroll
      for (int k = 0; k < RW_MAX; ++k) wxs[k] = ox0 + k <= ox1 ? resize_w(ox0 + k, p.sw, p.W, ix) : 0.f;
    }
    for (int oy = oy0; oy <= oy1; ++oy) {
      const float wy = resize_w(oy, p.sh, p.H, iy);
      if (wy == 0.f) continue;
      if (cached) {
#pragma unroll
        for (int k = 0; k < RW_MAX; ++k) {
          const float w = wy * wxs[k];
          if (w == 0.f) continue;
          const bf16x8 g = *(const bf16x8*)(g0 + ((int64_t)oy * p.Wo + ox0 + k) * p.ldx);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w * (float)g[e];
        }
        continue;
      }
      for (int ox = ox0; ox <= ox1; ++ox) {
        const float w = wy * resize_w(ox, p.sw, p.W, ix);
        if (w == 0.f) continue;
        const bf16x8 g = *(const bf16x8*)(g0 + ((int64_t)oy * p.Wo + ox) * p.ldx);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += w * (float)g[e];
      }
    }
    *(bf16x8*)(p.y + (((int64_t)n * p.H + iy) * p.W + ix) * p.ldy + c8 * 8) = pack8(acc);
  }
}
// The gather form for SMALL input maps with WIDE windows (the pyramid-pooling priors: 1 x 1 .. 6 x 6 maps blown up to 20 x 20,
// every input pixel collects 100 - 400 out-gradients; the 20 x 20 -> 80 x 80 fusion resize: ~120): one thread per
// (pixel, channel group) walked its whole window alone, and the launch was one or two blocks - 112 - 171 us of serial loads
// for a few KB of output (4 of the 7 resize backward launches of a SparseInst step, 0.55 ms).  Here a BLOCK owns one input
// pixel: 256 / C8 slices share the window (candidate q goes to slice q mod nslices), every slice accumulates its 8 channels
// in fp32, the slices are added in index order through LDS - fixed order, no atomics.
__global__ __launch_bounds__(256) void resize_bwd_pixel_kernel(const ResizeK p) {
  __shared__ float red[256 * 8];
  const int C8 = p.C8, nsl = 256 / C8;
  const int c8 = threadIdx.x % C8, sl = threadIdx.x / C8;
  int64_t r = blockIdx.x;
  const int ix = (int)(r % p.W); r /= p.W;
  const int iy = (int)(r % p.H);
  const int n = (int)(r / p.H);
  const float ish = 1.f / p.sh, isw = 1.f / p.sw;
  int oy0 = (int)floorf(((float)iy - 0.5f) * ish - 0.5f) - 1, oy1 = (int)ceilf(((float)iy + 1.5f) * ish - 0.5f) + 1;
  int ox0 = (int)floorf(((float)ix - 0.5f) * isw - 0.5f) - 1, ox1 = (int)ceilf(((float)ix + 1.5f) * isw - 0.5f) + 1;
  oy0 = oy0 < 0 ? 0 : oy0; ox0 = ox0 < 0 ? 0 : ox0;
  oy1 = oy1 > p.Ho - 1 ? p.Ho - 1 : oy1; ox1 = ox1 > p.Wo - 1 ? p.Wo - 1 : ox1;
  const int ww = ox1 - ox0 + 1, nq = (oy1 - oy0 + 1) * ww;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (sl < nsl) {
    const __bf16* g0 = p.x + ((int64_t)n * p.Ho * p.Wo) * p.ldx + c8 * 8;
    for (int q = sl; q < nq; q += nsl) {
      const int oy = oy0 + q / ww, ox = ox0 + q % ww;
      const float w = resize_w(oy, p.sh, p.H, iy) * resize_w(ox, p.sw, p.W, ix);
      if (w == 0.f) continue;
      const bf16x8 g = *(const bf16x8*)(g0 + ((int64_t)oy * p.Wo + ox) * p.ldx);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += w * (float)g[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(sl * C8 + c8) * 8 + e] = acc[e];
  }
  __syncthreads();
  if (sl == 0) {
    for (int k = 1; k < nsl; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += red[(k * C8 + c8) * 8 + e];
    *(bf16x8*)(p.y + (((int64_t)n * p.H + iy) * p.W + ix) * p.ldy + c8 * 8) = pack8(acc);
  }
}
__global__ __launch_bounds__(256) void f32_to_bf16_rows_kernel(const float* __restrict__ a, __bf16* o, int ldo, int64_t npix, int C8) {
  const int64_t total = npix * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    const int64_t px = idx / C8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a[px * C8 * 8 + c8 * 8 + e];
    *(bf16x8*)(o + px * ldo + c8 * 8) = pack8(v);
  }
}

static int nblocks(int64_t total) {
  int64_t b = (total + 1023) / 1024;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int mi_bilinear_resize_bf16(const void* x, int ldx, int N, int H, int W, int C, void* y, int ldy, int Ho, int Wo,
                                       mi_stream_t st) {
  MI_REQUIRE(x && y && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0,
             "bilinear_resize: args");
  ResizeK k;
  memset(&k, 0, sizeof(k));
  k.x = (const __bf16*)x; k.y = (__bf16*)y; k.ldx = ldx; k.ldy = ldy; k.N = N; k.H = H; k.W = W; k.Ho = Ho; k.Wo = Wo;
  k.C8 = C / 8; k.sh = (float)H / (float)Ho; k.sw = (float)W / (float)Wo;
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(nblocks((int64_t)N * Ho * Wo * k.C8)), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("bilinear_resize");
  return MI_OK;
}
// dy: [N][Ho][Wo][C] (lddy) -> dx: [N][H][W][C] (lddx); acc_ws: unused (may be NULL) - only MI_RESIZE_BWD_SCATTER=1, the
// first atomic form, needs it: fp32 N*H*W*C, zeroed by the caller
extern "C" int mi_bilinear_resize_bwd_bf16(const void* dy, int lddy, int N, int H, int W, int C, void* dx, int lddx, int Ho,
                                           int Wo, float* acc_ws, mi_stream_t st) {
  MI_REQUIRE(dy && dx && C % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "bilinear_resize_bwd: args");
  ResizeK k;
  memset(&k, 0, sizeof(k));
  k.x = (const __bf16*)dy; k.acc = acc_ws; k.ldx = lddy; k.N = N; k.H = H; k.W = W; k.Ho = Ho; k.Wo = Wo; k.C8 = C / 8;
  k.sh = (float)H / (float)Ho; k.sw = (float)W / (float)Wo;
  hipStream_t s = (hipStream_t)st;
  const char* sc = getenv("MI_RESIZE_BWD_SCATTER");      // (the first, atomic form: kept for comparison; it needs acc_ws zeroed)
  if (!(sc && atoi(sc))) {
    k.y = (__bf16*)dx; k.ldy = lddx;
    // window of one input pixel ~ (2 Ho / H + 3) x (2 Wo / W + 3) candidates (clamped to the map)
    const float wy = 2.f / k.sh + 3.f, wx = 2.f / k.sw + 3.f;
    const float cand = (wy < (float)Ho ? wy : (float)Ho) * (wx < (float)Wo ? wx : (float)Wo);
    static const int px_mode = getenv("MI_RESIZE_BWD_PIXEL") ? atoi(getenv("MI_RESIZE_BWD_PIXEL")) : 1;   // 0: round-4 form everywhere
    if (px_mode && cand >= 64.f && (int64_t)N * H * W <= 65536 && k.C8 <= 256) {
      hipLaunchKernelGGL(resize_bwd_pixel_kernel, dim3((unsigned)((int64_t)N * H * W)), dim3(256), 0, s, k);
      MI_CHECK_LAUNCH("bilinear_resize_bwd (block per pixel)");
      return MI_OK;
    }
    hipLaunchKernelGGL(resize_bwd_gather_kernel, dim3(nblocks((int64_t)N * H * W * k.C8)), dim3(256), 0, s, k);
    MI_CHECK_LAUNCH("bilinear_resize_bwd");
    return MI_OK;
  }
  MI_REQUIRE(acc_ws, "bilinear_resize_bwd: the scatter form needs the fp32 accumulator");
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(nblocks((int64_t)N * Ho * Wo * k.C8)), dim3(256), 0, s, k);
  MI_CHECK_LAUNCH("bilinear_resize_bwd");
  hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3(nblocks((int64_t)N * H * W * k.C8)), dim3(256), 0, s, acc_ws,
                     (__bf16*)dx, lddx, (int64_t)N * H * W, k.C8);
  MI_CHECK_LAUNCH("bilinear_resize_bwd_round");
  return MI_OK;
}

// ---------------------------------------------------------------- mask losses of the matched pairs
// masks: bf16 logits [B][P][ldm] (pixel-major, instance n = channel); targets: fp32 [T][P]; pairs: int32 [K][3] =
// (image b, instance n, target row t).  stats[k][8] = (sum BCE, A = sum sig*t, S = sum sig^2, Tt = sum t^2,
// |bin(sig >= 0.4) & (t > 0.5)|, |sig >= 0.4|, |t > 0.5|, 0)
struct MaskLossK {
  const __bf16* masks;
  const float* tgt;
  const int* pairs;
  float* stats;
  float* part;       // mask_stats: block partials [K][blocks][8]
  __bf16* dmasks;
  int K, P, ldm;
  float c_bce, c_dice;   // backward: upstream-gradient-scaled weights: c_bce = g_mask * w / (K * P), c_dice = g_dice * w / num_inst
  const float* coef;     // optional device pair (c_bce, c_dice) that replaces the two launch constants
};

__global__ __launch_bounds__(256) void mask_stats_kernel(const MaskLossK p) {
  __shared__ float red[4][8];
  const int k = blockIdx.y;
  const int b = p.pairs[k * 3], n = p.pairs[k * 3 + 1], t = p.pairs[k * 3 + 2];
  if (b < 0) return;   // an unused row of a fixed-capacity pair table (device-side matching): its statistics stay zero
  const __bf16* m = p.masks + (size_t)b * p.P * p.ldm + n;
  const float* tg = p.tgt + (size_t)t * p.P;
  float a[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int px = blockIdx.x * 256 + threadIdx.x; px < p.P; px += gridDim.x * 256) {
    const float x = (float)m[(size_t)px * p.ldm], tt = tg[px];
    const float sg = 1.f / (1.f + __expf(-x));
    // binary_cross_entropy_with_logits: max(x, 0) - x t + log(1 + exp(-|x|))
    a[0] += fmaxf(x, 0.f) - x * tt + log1pf(__expf(-fabsf(x)));
    a[1] += sg * tt; a[2] += sg * sg; a[3] += tt * tt;
    const float bs = sg >= 0.4f ? 1.f : 0.f, bt = tt > 0.5f ? 1.f : 0.f;
    a[4] += bs * bt; a[5] += bs; a[6] += bt;
  }
#pragma unroll
  for (int e = 0; e < 7; ++e) a[e] = wave_sum(a[e]);
  if ((threadIdx.x & 63) == 0)
    for (int e = 0; e < 7; ++e) red[threadIdx.x >> 6][e] = a[e];
  __syncthreads();
  // block partial -> workspace [k][block][8]; mask_stats_final_kernel adds the blocks in a fixed order.  (Round 4: this was an
  // fp32 atomicAdd into stats - 13 blocks per pair arriving in any order - and the captured SparseInst step was not
  // reproducible run to run: two final losses alternated.  Same two launches: the final pass replaces the zero fill.)
  if (threadIdx.x < 8)
    p.part[((size_t)k * gridDim.x + blockIdx.x) * 8 + threadIdx.x] =
        threadIdx.x < 7 ? (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]) : 0.f;
}
__global__ __launch_bounds__(256) void mask_stats_final_kernel(const MaskLossK p, int nblk) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.K * 8) return;
  const int k = i >> 3, e = i & 7;
  float s = 0.f;
  if (p.pairs[k * 3] >= 0)          // (an unused row's blocks returned before writing: its statistics are zero)
    for (int b = 0; b < nblk; ++b) s += p.part[((size_t)k * nblk + b) * 8 + e];
  p.stats[i] = s;
}

// d(loss)/d(logit) of pair k at pixel px: c_bce * (sig - t) + c_dice * (-2 t / D + 4 A sig / D^2) * sig (1 - sig), D = S + Tt + 1e-4
__global__ __launch_bounds__(256) void mask_grad_kernel(const MaskLossK p) {
  const int k = blockIdx.y;
  const int b = p.pairs[k * 3], n = p.pairs[k * 3 + 1], t = p.pairs[k * 3 + 2];
  if (b < 0) return;
  const __bf16* m = p.masks + (size_t)b * p.P * p.ldm + n;
  __bf16* dm = p.dmasks + (size_t)b * p.P * p.ldm + n;
  const float* tg = p.tgt + (size_t)t * p.P;
  const float A = p.stats[k * 8 + 1], D = p.stats[k * 8 + 2] + p.stats[k * 8 + 3] + 1e-4f;
  // the two upstream gradients: launch constants, or read from the device (a captured step has no host value of them)
  const float c_bce = p.coef ? p.coef[0] : p.c_bce, c_dice = p.coef ? p.coef[1] : p.c_dice;
  for (int px = blockIdx.x * 256 + threadIdx.x; px < p.P; px += gridDim.x * 256) {
    const float x = (float)m[(size_t)px * p.ldm], tt = tg[px];
    const float sg = 1.f / (1.f + __expf(-x));
    const float g = c_bce * (sg - tt) + c_dice * (-2.f * tt / D + 4.f * A * sg / (D * D)) * sg * (1.f - sg);
    dm[(size_t)px * p.ldm] = (__bf16)g;
  }
}

static int mask_stats_blocks(int P) {
  int bx = (P + 2047) / 2048;
  return bx > 64 ? 64 : bx;
}
extern "C" int64_t mi_sparseinst_mask_stats_ws_floats(int K, int P) {
  return K > 0 && P > 0 ? (int64_t)K * mask_stats_blocks(P) * 8 : -1;
}
extern "C" int mi_sparseinst_mask_stats(const void* masks, int ldm, int P, const float* targets, const int32_t* pairs, int K,
                                        float* stats, float* ws, mi_stream_t st) {
  MI_REQUIRE(masks && targets && pairs && stats && ws && K > 0 && P > 0 && ldm > 0, "sparseinst_mask_stats: args");
  MaskLossK k;
  memset(&k, 0, sizeof(k));
  k.masks = (const __bf16*)masks; k.tgt = targets; k.pairs = pairs; k.stats = stats; k.part = ws; k.K = K; k.P = P; k.ldm = ldm;
  hipStream_t s = (hipStream_t)st;
  const int bx = mask_stats_blocks(P);
  hipLaunchKernelGGL(mask_stats_kernel, dim3(bx, K), dim3(256), 0, s, k);
  MI_CHECK_LAUNCH("sparseinst_mask_stats");
  // (no hipMemsetAsync anywhere near this: inside the captured SparseInst step - a hipGraph of ~3000 nodes from a shared memory
  //  pool - memset nodes were the one node kind whose effect went missing on replays after the first; tools/si_graph_debug*.py)
  hipLaunchKernelGGL(mask_stats_final_kernel, dim3((K * 8 + 255) / 256), dim3(256), 0, s, k, bx);
  MI_CHECK_LAUNCH("sparseinst_mask_stats final");
  return MI_OK;
}
// dmasks: bf16 [B][P][ldm], zeroed by the caller (only the matched instances' columns are written)
static int mask_grad_launch(const void* masks, int ldm, int P, const float* targets, const int32_t* pairs, int K,
                            const float* stats, float c_bce, float c_dice, const float* coef_dev, void* dmasks, mi_stream_t st);
extern "C" int mi_sparseinst_mask_grad(const void* masks, int ldm, int P, const float* targets, const int32_t* pairs, int K,
                                       const float* stats, float c_bce, float c_dice, void* dmasks, mi_stream_t st) {
  return mask_grad_launch(masks, ldm, P, targets, pairs, K, stats, c_bce, c_dice, nullptr, dmasks, st);
}
// the same with the two upstream gradients read from the device (coef_dev[0] = d / d(sum of BCE sums), [1] = d / d(sum of
// dice losses)) and pair rows with image index < 0 skipped: the form a captured step uses (fixed-capacity pair table)
extern "C" int mi_sparseinst_mask_grad_dev(const void* masks, int ldm, int P, const float* targets, const int32_t* pairs,
                                           int K, const float* stats, const float* coef_dev, void* dmasks, mi_stream_t st) {
  MI_REQUIRE(coef_dev, "sparseinst_mask_grad_dev: coef_dev");
  return mask_grad_launch(masks, ldm, P, targets, pairs, K, stats, 0.f, 0.f, coef_dev, dmasks, st);
}
static int mask_grad_launch(const void* masks, int ldm, int P, const float* targets, const int32_t* pairs, int K,
                            const float* stats, float c_bce, float c_dice, const float* coef_dev, void* dmasks, mi_stream_t st) {
  MI_REQUIRE(masks && targets && pairs && stats && dmasks && K > 0 && P > 0, "sparseinst_mask_grad: args");
  MaskLossK k;
  memset(&k, 0, sizeof(k));
  k.masks = (const __bf16*)masks; k.tgt = targets; k.pairs = pairs; k.stats = (float*)stats; k.dmasks = (__bf16*)dmasks;
  k.K = K; k.P = P; k.ldm = ldm; k.c_bce = c_bce; k.c_dice = c_dice; k.coef = coef_dev;
  int bx = (P + 1023) / 1024;
  if (bx > 128) bx = 128;
  hipLaunchKernelGGL(mask_grad_kernel, dim3(bx, K), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("sparseinst_mask_grad");
  return MI_OK;
}

// ---------------------------------------------------------------- pyramid pooling (encoder_sparseinst.py:18-62)
// PyramidPoolingModule's MyAdaptiveAvgPool2d stages - F.avg_pool2d(x, kernel = (ceil(H / sz), ceil(W / sz)), stride = kernel,
// ceil_mode=False) for sz in (1, 2, 3, 6) - of ONE bf16 NHWC map in one launch, and their backward (the sum of the four
// stages' window-spread gradients) in one launch.  As torch calls each stage was float() + two avg_pool2d + to(bf16) forward
// and the mirror image + an accumulation backward: ~35 launches of 4 - 12 us on a [8, 20, 20, 256] map.
// fp32 window sums in a fixed order, one division by the window size (avg_pool2d's divisor with no padding).
struct PyrPoolK {
  const __bf16* x;      // [N][H][W][C] (ldx); backward: unused
  __bf16* dx;           // backward: [N][H][W][C] (lddx)
  __bf16* y[MI_PYR_MAX_STAGES];          // forward out / backward in: [N][oh][ow][C] dense
  int kh[MI_PYR_MAX_STAGES], kw[MI_PYR_MAX_STAGES], oh[MI_PYR_MAX_STAGES], ow[MI_PYR_MAX_STAGES], first[MI_PYR_MAX_STAGES + 1];
  int N, H, W, C8, ldx, ns;
};
// one BLOCK per (image, output pixel of any stage): thread (channel group c8 = tid % C8, part = tid / C8) sums every
// (256 / C8)-th element of the window, the parts meet in LDS in a fixed order.  (One thread per output walks a 20 x 20
// window alone: 400 dependent loads - the form torch's avg_pool2d has, 105 us for the 1-bin stage.)
__global__ __launch_bounds__(256) void pyr_pool_fwd_kernel(const PyrPoolK p) {
  __shared__ float red[256 * 8];
  const int per_img = p.first[p.ns];
  const int n = blockIdx.x / per_img, o = blockIdx.x - n * per_img;
  int s = 0;
  while (s + 1 < p.ns && o >= p.first[s + 1]) ++s;
  const int oo = o - p.first[s], oy = oo / p.ow[s], ox = oo - oy * p.ow[s];
  const int kh = p.kh[s], kw = p.kw[s], win = kh * kw;
  const int C8 = p.C8, parts = 256 / C8;
  const int tid = threadIdx.x, c8 = tid % C8, part = tid / C8;
  float a[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = 0.f;
  if (part < parts)
    for (int w = part; w < win; w += parts) {
      const int y = oy * kh + w / kw, x = ox * kw + w % kw;
      const bf16x8 v = *(const bf16x8*)(p.x + (((int64_t)n * p.H + y) * p.W + x) * p.ldx + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid * 8 + e] = a[e];
  __syncthreads();
  if (tid < C8) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = 0.f;
    for (int q = 0; q < parts; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] += red[(q * C8 + tid) * 8 + e];
    const float d = (float)win;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = t[e] / d;
    *(bf16x8*)(p.y[s] + (((int64_t)n * p.oh[s] + oy) * p.ow[s] + ox) * (C8 * 8) + tid * 8) = pack8(t);
  }
}
__global__ __launch_bounds__(256) void pyr_pool_bwd_kernel(const PyrPoolK p) {
  const int64_t total = (int64_t)p.N * p.H * p.W * p.C8;
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % p.C8);
    int64_t r = i / p.C8;
    const int x = (int)(r % p.W); r /= p.W;
    const int y = (int)(r % p.H), n = (int)(r / p.H);
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    for (int s = 0; s < p.ns; ++s) {
      const int oy = y / p.kh[s], ox = x / p.kw[s];
      if (oy >= p.oh[s] || ox >= p.ow[s] || !p.y[s]) continue;      // (ceil_mode False: the remainder rows / columns feed no window)
      const bf16x8 g = *(const bf16x8*)(p.y[s] + (((int64_t)n * p.oh[s] + oy) * p.ow[s] + ox) * (p.C8 * 8) + c8 * 8);
      const float d = (float)(p.kh[s] * p.kw[s]);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += (float)g[e] / d;
    }
    *(bf16x8*)(p.dx + (((int64_t)n * p.H + y) * p.W + x) * p.ldx + c8 * 8) = pack8(a);
  }
}
static int pyr_fill(PyrPoolK* k, int N, int H, int W, int C, int ns, const int* kh, const int* kw, void* const* y) {
  MI_REQUIRE(N >= 1 && H >= 1 && W >= 1 && C % 8 == 0 && ns >= 1 && ns <= MI_PYR_MAX_STAGES && kh && kw && y, "pyramid_pool: args");
  memset(k, 0, sizeof(*k));
  k->N = N; k->H = H; k->W = W; k->C8 = C / 8; k->ns = ns;
  int off = 0;
  for (int s = 0; s < ns; ++s) {
    MI_REQUIRE(kh[s] >= 1 && kw[s] >= 1 && kh[s] <= H && kw[s] <= W, "pyramid_pool: stage %d window %d x %d on a %d x %d map", s, kh[s], kw[s], H, W);
    k->kh[s] = kh[s]; k->kw[s] = kw[s]; k->oh[s] = H / kh[s]; k->ow[s] = W / kw[s];
    k->y[s] = (__bf16*)y[s];
    k->first[s] = off;
    off += k->oh[s] * k->ow[s];
  }
  k->first[ns] = off;
  return MI_OK;
}
extern "C" int mi_pyramid_pool_fwd(const void* x, int ldx, int N, int H, int W, int C, int ns, const int* kh, const int* kw,
                                   void* const* y, mi_stream_t st) {
  PyrPoolK k;
  int rc = pyr_fill(&k, N, H, W, C, ns, kh, kw, y);
  if (rc) return rc;
  MI_REQUIRE(x && ldx % 8 == 0 && ldx >= C, "pyramid_pool_fwd: x");
  for (int s = 0; s < ns; ++s) MI_REQUIRE(y[s], "pyramid_pool_fwd: output %d", s);
  k.x = (const __bf16*)x; k.ldx = ldx;
  MI_REQUIRE(k.C8 <= 256, "pyramid_pool_fwd: at most 2048 channels");
  hipLaunchKernelGGL(pyr_pool_fwd_kernel, dim3((unsigned)(N * k.first[ns])), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("pyramid_pool_fwd");
  return MI_OK;
}
extern "C" int mi_pyramid_pool_bwd(void* const* dy, int N, int H, int W, int C, int ns, const int* kh, const int* kw, void* dx,
                                   int lddx, mi_stream_t st) {
  PyrPoolK k;
  int rc = pyr_fill(&k, N, H, W, C, ns, kh, kw, dy);      // (a NULL dy[s]: that stage received no gradient)
  if (rc) return rc;
  MI_REQUIRE(dx && lddx % 8 == 0 && lddx >= C, "pyramid_pool_bwd: dx");
  k.dx = (__bf16*)dx; k.ldx = lddx;
  hipLaunchKernelGGL(pyr_pool_bwd_kernel, dim3(nblocks((int64_t)N * H * W * k.C8)), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("pyramid_pool_bwd");
  return MI_OK;
}
