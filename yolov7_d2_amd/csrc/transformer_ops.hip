// Row-wise ops around the attention core of DETR's transformer layers (modeling/backbone/detr_backbone.py:135-278):
// LayerNorm forward / backward (nn.LayerNorm(d_model), eps 1e-5, biased variance), ReLU backward mask, residual add.
// bf16 [T][E] token tensors (T = L*B rows), fp32 statistics and parameters.  All HBM streams; one wave per row.
#include "common.h"

// ---- LayerNorm forward: y = (x - mean) * rstd * gamma + beta ; saves mean / rstd [T]
// res != NULL: the layer's input is res + dropout(x) (the post-norm residual of detr_backbone.py:163-168,235-243:
// `src = self.norm1(src + self.dropout1(src2))`), formed here, rounded to bf16 and written to sum_out - the bits
// mi_dropout_add_bf16 would have written - so that the row is read once and the step has one launch less per norm
extern const unsigned long long* g_mi_seed_off;   // runtime.hip
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const __bf16* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, __bf16* y, float* mean,
                                                            float* rstd, int T, int E, float eps,
                                                            const __bf16* __restrict__ res, __bf16* sum_out, unsigned thr,
                                                            float dscale, unsigned long long seed,
                                                            const unsigned long long* seed_off) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= T) return;
  const __bf16* xr = x + (size_t)row * E;
  float v[16];  // E <= 1024: 16 values per lane
  const int per = E / 64;
  float s = 0.f;
  if (res) {
    if (seed_off) seed += *seed_off;
    const unsigned long long rk = mi_rng_key(seed);
    for (int j = 0; j < per; ++j) {
      const size_t i = (size_t)row * E + lane + 64 * j;
      const __bf16 d = (__bf16)(mi_rng32k(rk, (unsigned long long)i) >= thr ? (float)xr[lane + 64 * j] * dscale : 0.f);
      const __bf16 t = (__bf16)((float)res[i] + (float)d);
      sum_out[i] = t;
      v[j] = (float)t;
      s += v[j];
    }
  } else
  for (int j = 0; j < per; ++j) { v[j] = (float)xr[lane + 64 * j]; s += v[j]; }
  const float mu = wave_sum(s) / (float)E;
  float q = 0.f;
  for (int j = 0; j < per; ++j) { const float d = v[j] - mu; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) / (float)E + eps);
  for (int j = 0; j < per; ++j) {
    const int c = lane + 64 * j;
    y[(size_t)row * E + c] = (__bf16)((v[j] - mu) * rs * gamma[c] + beta[c]);
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// ---- LayerNorm backward: dx per row; dgamma / dbeta as block partials [nblk][E][2] (second stage below).
// A lane owns PER = E / 64 CONSECUTIVE channels (one 2 PER-byte load per tensor and row); a block takes rows_per_block rows,
// chosen by the launcher so that the grid is ~2 blocks per CU (T = 4200 tokens: 16 rows, 263 blocks; the first version's
// fixed 64 rows gave 66 blocks of scalar 2-byte loads: 37 us per call).
template <int PER>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, __bf16* dx, float* part,
                                                            int T, int E, int rows_per_block, __bf16* dx_drop, unsigned thr,
                                                            float dscale, unsigned long long seed,
                                                            const unsigned long long* seed_off) {
  // dx_drop != NULL: also dropout(dx) with the forward's (p, seed) - the gradient of the dropped branch of
  // res + dropout(x) in front of this norm (what mi_dropout_bf16 applied to dx would write)
  if (dx_drop && seed_off) seed += *seed_off;
  const unsigned long long rk = dx_drop ? mi_rng_key(seed) : 0ull;
  extern __shared__ float sacc[];  // [4 waves][E][2]
  typedef __attribute__((ext_vector_type(PER))) __bf16 bvec;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = lane * PER;
  float ag[PER], ab[PER], gm[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) { ag[j] = ab[j] = 0.f; gm[j] = gamma[c0 + j]; }
  const int r0 = blockIdx.x * rows_per_block;
  for (int rr = wave; rr < rows_per_block; rr += 4) {
    const int row = r0 + rr;
    if (row >= T) break;
    const float mu = mean[row], rs = rstd[row];
    const bvec xv = *(const bvec*)(x + (size_t)row * E + c0), dv = *(const bvec*)(dy + (size_t)row * E + c0);
    float xh[PER], g[PER];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const float d = (float)dv[j];
      xh[j] = ((float)xv[j] - mu) * rs;
      g[j] = d * gm[j];
      s1 += g[j];
      s2 += g[j] * xh[j];
      ag[j] += d * xh[j];
      ab[j] += d;
    }
    s1 = wave_sum(s1) / (float)E;
    s2 = wave_sum(s2) / (float)E;
    bvec o;
#pragma unroll
    for (int j = 0; j < PER; ++j) o[j] = (__bf16)(rs * (g[j] - s1 - xh[j] * s2));
    *(bvec*)(dx + (size_t)row * E + c0) = o;
    if (dx_drop) {
      bvec od;
#pragma unroll
      for (int j = 0; j < PER; ++j)
        od[j] = (__bf16)(mi_rng32k(rk, (unsigned long long)((size_t)row * E + c0 + j)) >= thr ? (float)o[j] * dscale : 0.f);
      *(bvec*)(dx_drop + (size_t)row * E + c0) = od;
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    sacc[(wave * E + c0 + j) * 2 + 0] = ag[j];
    sacc[(wave * E + c0 + j) * 2 + 1] = ab[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < E; c += 256) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < 4; ++w) { a += sacc[(w * E + c) * 2]; b += sacc[(w * E + c) * 2 + 1]; }
    part[((size_t)blockIdx.x * E + c) * 2 + 0] = a;
    part[((size_t)blockIdx.x * E + c) * 2 + 1] = b;
  }
}
// dgamma / dbeta: sum of the block partials.  One block = 32 channels x 8 row groups (thread (g, c) sums the rows g, g + 8,
// ... with four independent chains, the groups are combined through LDS in a fixed order).  The first form - one thread
// per channel walking ALL partial rows - was a chain of nblk / 4 dependent L2 round trips in a single block: 10 us for the
// 263 rows of a DETR encoder layer (T = 4 200), 36 launches per step.
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const float* __restrict__ part, int nblk, int E,
                                                                   float* dgamma, float* dbeta) {
  __shared__ float sa[8][32], sb[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < E) {
    int k = g;
    for (; k + 24 < nblk; k += 32)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x2 v = *(const f32x2*)(part + ((size_t)(k + 8 * u) * E + c) * 2);
        a[u] += v[0];
        b[u] += v[1];
      }
    for (; k < nblk; k += 8) {
      const f32x2 v = *(const f32x2*)(part + ((size_t)k * E + c) * 2);
      a[0] += v[0];
      b[0] += v[1];
    }
  }
  sa[g][cl] = (a[0] + a[1]) + (a[2] + a[3]);
  sb[g][cl] = (b[0] + b[1]) + (b[2] + b[3]);
  __syncthreads();
  if (g == 0 && c < E) {
    float x = sa[0][cl], y = sb[0][cl];
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      x += sa[q][cl];
      y += sb[q][cl];
    }
    dgamma[c] = x;
    dbeta[c] = y;
  }
}

static unsigned drop_threshold(float drop_p) {
  unsigned thr = drop_p > 0.f ? (unsigned)((double)drop_p * 4294967296.0) : 0u;
  if (drop_p > 0.f && thr == 0u) thr = 1u;
  return thr;
}
extern "C" int mi_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                int T, int E, float eps, mi_stream_t st) {
  MI_REQUIRE(x && gamma && beta && y && mean && rstd && T > 0, "layernorm_fwd: args");
  MI_REQUIRE(E % 64 == 0 && E <= 1024, "layernorm_fwd: E %d (multiple of 64, <= 1024)", E);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(mi_cdiv(T, 4)), dim3(256), 0, (hipStream_t)st, (const __bf16*)x, gamma, beta,
                     (__bf16*)y, mean, rstd, T, E, eps, (const __bf16*)nullptr, (__bf16*)nullptr, 0u, 1.f, 0ull,
                     (const unsigned long long*)nullptr);
  MI_CHECK_LAUNCH("layernorm_fwd");
  return MI_OK;
}
extern "C" int mi_dropout_add_layernorm_fwd(const void* x, const void* res, void* sum_out, const float* gamma, const float* beta,
                                            void* y, float* mean, float* rstd, int T, int E, float eps, float drop_p,
                                            uint64_t seed, mi_stream_t st) {
  MI_REQUIRE(x && res && sum_out && gamma && beta && y && mean && rstd && T > 0, "dropout_add_layernorm_fwd: args");
  MI_REQUIRE(E % 64 == 0 && E <= 1024, "dropout_add_layernorm_fwd: E %d (multiple of 64, <= 1024)", E);
  MI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "dropout_add_layernorm_fwd: p %f", drop_p);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(mi_cdiv(T, 4)), dim3(256), 0, (hipStream_t)st, (const __bf16*)x, gamma, beta,
                     (__bf16*)y, mean, rstd, T, E, eps, (const __bf16*)res, (__bf16*)sum_out, drop_threshold(drop_p),
                     1.f / (1.f - drop_p), (unsigned long long)seed, g_mi_seed_off);
  MI_CHECK_LAUNCH("dropout_add_layernorm_fwd");
  return MI_OK;
}
/* ws: fp32 [ceil(T/16)][E][2] (one partial row per block; a block takes >= 16 token rows) */
extern "C" int mi_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                                void* dx, float* dgamma, float* dbeta, float* ws, int T, int E, mi_stream_t st) {
  return mi_layernorm_bwd_dropout(x, dy, gamma, mean, rstd, dx, nullptr, dgamma, dbeta, ws, T, E, 0.f, 0, st);
}
extern "C" int mi_layernorm_bwd_dropout(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                                        void* dx, void* dx_drop, float* dgamma, float* dbeta, float* ws, int T, int E,
                                        float drop_p, uint64_t seed, mi_stream_t st) {
  MI_REQUIRE(x && dy && gamma && mean && rstd && dx && dgamma && dbeta && ws && T > 0, "layernorm_bwd: args");
  MI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "layernorm_bwd: p %f", drop_p);
  const unsigned thr = drop_threshold(drop_p);
  const float dscale = 1.f / (1.f - drop_p);
  MI_REQUIRE(E % 64 == 0 && E <= 1024, "layernorm_bwd: E %d", E);
  // rows per block: ~512 blocks for long sequences, 16 rows at least (the workspace contract: ceil(T / 16) partial rows) -
  // T = 4200 (DETR's encoder at 800 x 1333, bs 4) runs 263 blocks of 16 rows
  int rpb = mi_cdiv(mi_cdiv(T, 512), 4) * 4;
  if (rpb < 16) rpb = 16;
  const int nblk = mi_cdiv(T, rpb);
  hipStream_t s = (hipStream_t)st;
  const size_t lds = (size_t)4 * E * 2 * sizeof(float);
#define MI_LN_BWD(PERv)                                                                                                  \
  hipLaunchKernelGGL(layernorm_bwd_kernel<PERv>, dim3(nblk), dim3(256), lds, s, (const __bf16*)x, (const __bf16*)dy, gamma, \
                     mean, rstd, (__bf16*)dx, ws, T, E, rpb, (__bf16*)dx_drop, thr, dscale, (unsigned long long)seed,          \
                     g_mi_seed_off)
  switch (E / 64) {
    case 1: MI_LN_BWD(1); break;
    case 2: MI_LN_BWD(2); break;
    case 4: MI_LN_BWD(4); break;
    case 8: MI_LN_BWD(8); break;
    case 16: MI_LN_BWD(16); break;
    default: MI_FAIL(MI_EINVAL, "layernorm_bwd: E %d (64, 128, 256, 512 or 1024)", E);
  }
#undef MI_LN_BWD
  MI_CHECK_LAUNCH("layernorm_bwd");
  hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3(mi_cdiv(E, 32)), dim3(256), 0, s, ws, nblk, E, dgamma, dbeta);
  MI_CHECK_LAUNCH("layernorm_bwd_params");
  return MI_OK;
}

// ---- elementwise: out = a + b ; out = relu(a) ; dx = dy * (y > 0)        (n multiple of 8, 16-byte aligned)
__global__ __launch_bounds__(256) void ew_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, __bf16* o,
                                                 int64_t n8, int op) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const bf16x8 va = *(const bf16x8*)(a + i * 8);
    bf16x8 vb = va;
    if (b) vb = *(const bf16x8*)(b + i * 8);
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (float)va[e], y = (float)vb[e];
      float z;
      if (op == 0) z = x + y;
      else if (op == 1) z = fmaxf(x, 0.f);
      else if (op == 2) z = y > 0.f ? x : 0.f;  // a = dy, b = forward output
      else if (op == 3) z = 1.f / (1.f + __expf(-x));
      else if (op == 4) z = x * y * (1.f - y); // sigmoid backward, a = dy, b = forward output
      else if (op == 5) z = x / (1.f + __expf(-x));   // swish / SiLU (BiFPN's Swish, neck/bifpn.py:49-61)
      else if (op == 7) z = fmaxf((float)(__bf16)(x + y), 0.f);   // relu(a + b): the tail of a ResNet bottleneck in one pass
      else {                                   // op 6: swish backward, a = dy, b = forward INPUT
        const float sg = 1.f / (1.f + __expf(-y));
        z = x * sg * (1.f + y * (1.f - sg));
      }
      r[e] = (__bf16)z;
    }
    *(bf16x8*)(o + i * 8) = r;
  }
}
extern "C" int mi_ew_bf16(const void* a, const void* b, void* out, int64_t n, int op, mi_stream_t st) {
  MI_REQUIRE(a && out && n > 0 && n % 8 == 0 && op >= 0 && op <= 7 && (op == 1 || op == 3 || op == 5 || b), "ew_bf16: args");
  MI_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)out % 16) == 0, "ew_bf16: alignment");
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ew_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)st, (const __bf16*)a, (const __bf16*)b,
                     (__bf16*)out, n / 8, op);
  MI_CHECK_LAUNCH("ew_bf16");
  return MI_OK;
}

// ---- fp32 sigmoid with gradient: the box head's final activation (meta_arch/detr.py:452, outputs_coord.sigmoid());
// dx may be NULL (forward) - with dy given, dx = dy * y * (1 - y) from the stored output y
__global__ __launch_bounds__(256) void sigmoid_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ y, float* __restrict__ dx, int64_t n) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (dy) {
      const float v = y[i];
      dx[i] = dy[i] * v * (1.f - v);
    } else {
      y[i] = 1.f / (1.f + expf(-x[i]));
    }
  }
}
extern "C" int mi_sigmoid_f32(const float* x, const float* dy, float* y, float* dx, int64_t n, mi_stream_t st) {
  MI_REQUIRE(y && n > 0 && ((x && !dy && !dx) || (dy && dx)), "sigmoid_f32: forward needs (x, y), backward (dy, y, dx)");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sigmoid_f32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)st, x, dy, y, dx, n);
  MI_CHECK_LAUNCH("sigmoid_f32");
  return MI_OK;
}

// ---- elementwise dropout (F.dropout in the transformer layers, detr_backbone.py:147-150,163-167,212-217,235-243)
extern const unsigned long long* g_mi_seed_off;   // runtime.hip
// res != NULL: o = res + dropout(x), the residual add of the same layer in the same pass (the dropped value is rounded to
// bf16 before the fp32 add, as the two-launch form dropout -> mi_ew_bf16 add rounds it: identical bits)
__global__ __launch_bounds__(256) void dropout_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ res, __bf16* o,
                                                      int64_t n8, unsigned thr, float scale, unsigned long long seed,
                                                      const unsigned long long* seed_off) {
  if (seed_off) seed += *seed_off;   // (a captured step: the word is advanced once per replay)
  const unsigned long long rk = mi_rng_key(seed);
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const bf16x8 v = *(const bf16x8*)(x + i * 8);
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      r[e] = (__bf16)(mi_rng32k(rk, (unsigned long long)(i * 8 + e)) >= thr ? (float)v[e] * scale : 0.f);
    if (res) {
      const bf16x8 a = *(const bf16x8*)(res + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = (__bf16)((float)a[e] + (float)r[e]);
    }
    *(bf16x8*)(o + i * 8) = r;
  }
}
extern "C" int mi_dropout_bf16(const void* x, void* out, int64_t n, float drop_p, uint64_t seed, mi_stream_t st) {
  return mi_dropout_add_bf16(x, nullptr, out, n, drop_p, seed, st);
}
extern "C" int mi_dropout_add_bf16(const void* x, const void* res, void* out, int64_t n, float drop_p, uint64_t seed,
                                   mi_stream_t st) {
  MI_REQUIRE(x && out && n > 0 && n % 8 == 0 && drop_p >= 0.f && drop_p < 1.f, "dropout: args (n %lld, p %f)", (long long)n,
             drop_p);
  MI_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)res % 16) == 0, "dropout: alignment");
  unsigned thr = drop_p > 0.f ? (unsigned)((double)drop_p * 4294967296.0) : 0u;
  if (drop_p > 0.f && thr == 0u) thr = 1u;
  int64_t nb = (n / 8 + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)st, (const __bf16*)x, (const __bf16*)res,
                     (__bf16*)out, n / 8, thr, 1.f / (1.f - drop_p), (unsigned long long)seed, g_mi_seed_off);
  MI_CHECK_LAUNCH("dropout");
  return MI_OK;
}
