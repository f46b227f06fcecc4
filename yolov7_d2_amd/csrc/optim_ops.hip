// Optimizer-side kernels over the flat parameter / gradient arena (the trainer loop around the path)
// and the DETR positional encoding.
//   mi_adamw_step            torch.optim.AdamW.step as train_transformer.py builds it for DETR (decoupled weight decay,
//                            bias-corrected moments), one launch over all parameter segments
//   mi_grad_clip_full_model  FullModelGradientClippingOptimizer (yolov7/optimizer/build.py:206-223): one global L2 norm
//                            over every gradient, grads *= min(1, max_norm / (norm + 1e-6)); no host synchronisation
//   mi_pos_embed_sine        PositionEmbeddingSine.forward (modeling/backbone/detr_backbone.py:309-375)
#include "common.h"

// ---------------------------------------------------------------- AdamW
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* __restrict__ g, float* m, float* v,
                                                    const mi_sgd_seg* __restrict__ segs, float beta1, float beta2,
                                                    float eps, float bc1, float bc2_sqrt, float grad_scale) {
  const mi_sgd_seg sg = segs[blockIdx.x];
  const float wd = sg.weight_decay, lr = sg.lr;
  const float step_size = lr / bc1;
  for (int64_t i = threadIdx.x; i < sg.count; i += 256) {
    const int64_t k = sg.offset + i;
    const float gr = g[k] * grad_scale;
    float pv = p[k];
    pv *= 1.f - lr * wd;                       // decoupled weight decay
    const float mk = beta1 * m[k] + (1.f - beta1) * gr;
    const float vk = beta2 * v[k] + (1.f - beta2) * gr * gr;
    m[k] = mk;
    v[k] = vk;
    const float denom = sqrtf(vk) / bc2_sqrt + eps;
    p[k] = pv - step_size * (mk / denom);
  }
}

extern "C" int mi_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                             const mi_sgd_seg* segs_dev, int nseg, float beta1, float beta2, float eps, int64_t step,
                             float grad_scale, mi_stream_t st) {
  MI_REQUIRE(params && grads && exp_avg && exp_avg_sq && segs_dev && nseg > 0 && step >= 1, "adamw: args");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)st, params, grads, exp_avg, exp_avg_sq,
                     segs_dev, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
  MI_CHECK_LAUNCH("adamw");
  return MI_OK;
}

// ---------------------------------------------------------------- AdamW over separately allocated tensors
// torch.optim.AdamW(capturable=True) inside a captured step is ~470 small launches per DETR step (41 M parameters in 235
// tensors); this is one.  Every pointer and the update count live in DEVICE tables: a hipGraph records only the tables'
// addresses, so the gradient pointers (known when the capture ends) are filled in afterwards and the count advances per
// replay.  A block owns one chunk (<= 16 k elements) of one tensor.
__global__ __launch_bounds__(256) void adamw_multi_kernel(const mi_adamw_tensor* __restrict__ tensors,
                                                          const mi_adamw_chunk* __restrict__ chunks, float beta1, float beta2,
                                                          float eps, const long long* __restrict__ step_dev, float grad_scale,
                                                          const float* __restrict__ grad_scale_dev) {
  const mi_adamw_chunk ch = chunks[blockIdx.x];
  const mi_adamw_tensor t = tensors[ch.tensor];
  if (grad_scale_dev) grad_scale *= *grad_scale_dev;   // the clip coefficient of THIS step (mi_grad_norm_multi), read at run time
  const double step = (double)*step_dev;
  const float bc1 = (float)(1.0 - pow((double)beta1, step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
  const float wd = t.weight_decay, lr = t.lr;
  const float step_size = lr / bc1;
  float* p = t.p + ch.offset;
  const float* g = t.g + ch.offset;
  float* m = t.m + ch.offset;
  float* v = t.v + ch.offset;
  for (int i = threadIdx.x; i < ch.count; i += 256) {
    const float gr = g[i] * grad_scale;
    float pv = p[i];
    pv *= 1.f - lr * wd;                       // decoupled weight decay
    const float mk = beta1 * m[i] + (1.f - beta1) * gr;
    const float vk = beta2 * v[i] + (1.f - beta2) * gr * gr;
    m[i] = mk;
    v[i] = vk;
    const float denom = sqrtf(vk) / bc2_sqrt + eps;
    p[i] = pv - step_size * (mk / denom);
  }
}
extern "C" int mi_adamw_step_multi_clip(const mi_adamw_tensor* tensors_dev, const mi_adamw_chunk* chunks_dev, int nchunks,
                                        float beta1, float beta2, float eps, const int64_t* step_dev, float grad_scale,
                                        const float* grad_scale_dev, mi_stream_t st) {
  MI_REQUIRE(tensors_dev && chunks_dev && step_dev && nchunks > 0, "adamw_multi: args");
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)st, tensors_dev, chunks_dev, beta1, beta2,
                     eps, (const long long*)step_dev, grad_scale, grad_scale_dev);
  MI_CHECK_LAUNCH("adamw_multi");
  return MI_OK;
}
extern "C" int mi_adamw_step_multi(const mi_adamw_tensor* tensors_dev, const mi_adamw_chunk* chunks_dev, int nchunks,
                                   float beta1, float beta2, float eps, const int64_t* step_dev, float grad_scale,
                                   mi_stream_t st) {
  return mi_adamw_step_multi_clip(tensors_dev, chunks_dev, nchunks, beta1, beta2, eps, step_dev, grad_scale, nullptr, st);
}

// full-model gradient norm over the SAME tables (FullModelGradientClippingOptimizer.step = clip_grad_norm_ over every
// parameter, then the update: yolov7/optimizer/build.py:206-223).  One block per chunk leaves an fp64 partial; a single
// block adds them in index order (deterministic) and writes coef = min(1, max_norm / (norm + 1e-6)) and the norm to the
// device - the update kernel multiplies every gradient by it, so the clipped gradients are never written back and nothing
// of the step touches the host: the whole clip + update is capturable.
__global__ __launch_bounds__(256) void sqsum_multi_kernel(const mi_adamw_tensor* __restrict__ tensors,
                                                          const mi_adamw_chunk* __restrict__ chunks, double* __restrict__ partial) {
  __shared__ double red[4];
  const mi_adamw_chunk ch = chunks[blockIdx.x];
  const float* g = tensors[ch.tensor].g + ch.offset;
  double s = 0.0;
  for (int i = threadIdx.x; i < ch.count; i += 256) {
    const float a = g[i];
    s += (double)a * (double)a;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ partial, int n, float max_norm,
                                                        float grad_scale, float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    // grad_scale: the factor the update will apply to every gradient BEFORE clipping (1 / world_size of a data-parallel
    // step whose buffers hold the all-reduced SUM): the norm that is clipped is the averaged gradient's
    const float norm = grad_scale * (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
    const float c = max_norm / (norm + 1e-6f);
    // torch.nn.utils.clip_grad_norm_ (the reference's FullModelGradientClippingOptimizer, optimizer/build.py:206-223):
    // clamp(max_norm / (norm + 1e-6), max = 1) - a NaN norm gives a NaN coefficient and the divergence shows in the very
    // next loss, an infinite norm gives 0.  (`c < 1 ? c : 1` alone turned NaN into 1: a silent step on garbage.)
    out[0] = (norm != norm) ? norm : (c < 1.f ? c : 1.f);
    out[1] = norm;
  }
}
extern "C" int mi_grad_norm_multi(const mi_adamw_tensor* tensors_dev, const mi_adamw_chunk* chunks_dev, int nchunks,
                                  double* partial_dev, float max_norm, float grad_scale, float* coef_norm_out,
                                  mi_stream_t st) {
  MI_REQUIRE(tensors_dev && chunks_dev && partial_dev && coef_norm_out && nchunks > 0 && max_norm > 0.f && grad_scale > 0.f,
             "grad_norm_multi: args");
  hipLaunchKernelGGL(sqsum_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)st, tensors_dev, chunks_dev, partial_dev);
  MI_CHECK_LAUNCH("grad_sqsum_multi");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)st, partial_dev, nchunks, max_norm, grad_scale,
                     coef_norm_out);
  MI_CHECK_LAUNCH("clip_coef");
  return MI_OK;
}

// the gradients of an eager module tree -> ONE flat fp32 buffer (tensor k at element offset flat_off[k]), one launch over the
// same chunk table: what a data-parallel step all-reduces in a few large messages (train_transformer.py:188-203 -> d2
// create_ddp_model's buckets) and what the update then reads at addresses that do not depend on the captured graph's pool
__global__ __launch_bounds__(256) void grad_gather_kernel(const mi_adamw_tensor* __restrict__ tensors,
                                                          const mi_adamw_chunk* __restrict__ chunks,
                                                          const long long* __restrict__ flat_off, float* __restrict__ flat) {
  const mi_adamw_chunk ch = chunks[blockIdx.x];
  const float* g = tensors[ch.tensor].g + ch.offset;
  float* o = flat + flat_off[ch.tensor] + ch.offset;
  for (int i = threadIdx.x; i < ch.count; i += 256) o[i] = g[i];
}
extern "C" int mi_grad_gather_multi(const mi_adamw_tensor* tensors_dev, const mi_adamw_chunk* chunks_dev, int nchunks,
                                    const int64_t* flat_off_dev, float* flat, mi_stream_t st) {
  MI_REQUIRE(tensors_dev && chunks_dev && flat_off_dev && flat && nchunks > 0, "grad_gather_multi: args");
  hipLaunchKernelGGL(grad_gather_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)st, tensors_dev, chunks_dev,
                     (const long long*)flat_off_dev, flat);
  MI_CHECK_LAUNCH("grad_gather_multi");
  return MI_OK;
}

// ---------------------------------------------------------------- full-model gradient clipping
#define CLIP_BLOCKS 1024
__global__ __launch_bounds__(256) void sqsum_kernel(const float* __restrict__ g, int64_t n, double* partial) {
  __shared__ double red[4];
  double s = 0.0;
  const int64_t n4 = n >> 2;
  const f32x4* g4 = (const f32x4*)g;
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 a = g4[i];
    s += (double)(a[0] * a[0] + a[1] * a[1]) + (double)(a[2] * a[2] + a[3] * a[3]);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += (double)g[i] * (double)g[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void clip_scale_kernel(float* g, int64_t n, const double* __restrict__ partial, int nb,
                                                         float max_norm, float* norm_out) {
  __shared__ float s_coef;
  if (threadIdx.x < 64) {   // every block re-reduces the (<= 1024) partials in the same fixed order
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) {
      const float norm = (float)sqrt(s);
      float c = max_norm / (norm + 1e-6f);
      s_coef = c < 1.f ? c : 1.f;
      if (blockIdx.x == 0 && norm_out) *norm_out = norm;
    }
  }
  __syncthreads();
  const float c = s_coef;
  if (c >= 1.f) return;
  const int64_t n4 = n >> 2;
  f32x4* g4 = (f32x4*)g;
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) g4[i] = g4[i] * c;
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) g[i] *= c;
}

extern "C" int mi_grad_clip_full_model(float* grads, int64_t n, float max_norm, double* ws, float* norm_out,
                                       mi_stream_t st) {
  MI_REQUIRE(grads && ws && n > 0 && max_norm > 0.f && ((uintptr_t)grads % 16) == 0, "grad_clip: args");
  int nb = (int)((n / 4 + 255) / 256);
  if (nb > CLIP_BLOCKS) nb = CLIP_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sqsum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)st, grads, n, ws);
  MI_CHECK_LAUNCH("grad_sqsum");
  hipLaunchKernelGGL(clip_scale_kernel, dim3(nb), dim3(256), 0, (hipStream_t)st, grads, n, ws, nb, max_norm, norm_out);
  MI_CHECK_LAUNCH("grad_clip_scale");
  return MI_OK;
}

// ---------------------------------------------------------------- sine positional embedding
// mask [B][H][W] bytes (non-zero = padding) -> pos fp32 [B][2N][H][W]: channels [0,N) from the row count of valid cells
// (y), [N,2N) from the column count (x); channel 2p = sin(e / T^(2p/N)), 2p+1 = cos(e / T^(2p/N))
#define POS_CG 8
__global__ __launch_bounds__(256) void pos_embed_sine_kernel(const uint8_t* __restrict__ mask, int B, int H, int W, int N,
                                                             float temperature, int normalize, float scale,
                                                             int centered, float* __restrict__ out) {
  // blockIdx.z = a group of POS_CG channels: the 2 N powf / sinf / cosf of a cell were one thread's serial work (102 us per
  // DETR step for 4 200 cells on 17 blocks); the same calls spread over N / POS_CG times as many threads - identical results
  const int b = blockIdx.y;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= H * W) return;
  const int i0 = (int)blockIdx.z * POS_CG, i1 = i0 + POS_CG < N ? i0 + POS_CG : N;
  const int h = cell / W, w = cell - h * W;
  const uint8_t* mb = mask + (size_t)b * H * W;
  float ye = 0.f, xe = 0.f, ylast = 0.f, xlast = 0.f;
  for (int r = 0; r < H; ++r) {
    const float nv = mb[r * W + w] ? 0.f : 1.f;
    ylast += nv;
    if (r <= h) ye += nv;
  }
  for (int c = 0; c < W; ++c) {
    const float nv = mb[h * W + c] ? 0.f : 1.f;
    xlast += nv;
    if (c <= w) xe += nv;
  }
  if (normalize) {
    const float eps = 1e-6f;
    if (centered) {
      ye = (ye - 0.5f) / (ylast + eps) * scale;
      xe = (xe - 0.5f) / (xlast + eps) * scale;
    } else {
      ye = ye / (ylast + eps) * scale;
      xe = xe / (xlast + eps) * scale;
    }
  }
  float* ob = out + (size_t)b * 2 * N * H * W + cell;
  for (int i = i0; i < i1; ++i) {
    const float dim_t = powf(temperature, (float)(2 * (i / 2)) / (float)N);
    const float py = ye / dim_t, px = xe / dim_t;
    ob[(size_t)i * H * W] = (i & 1) ? cosf(py) : sinf(py);
    ob[(size_t)(N + i) * H * W] = (i & 1) ? cosf(px) : sinf(px);
  }
}

extern "C" int mi_pos_embed_sine(const uint8_t* mask, int B, int H, int W, int num_pos_feats, float temperature,
                                 int normalize, float scale, int centered, float* out, mi_stream_t st) {
  MI_REQUIRE(mask && out && B > 0 && H > 0 && W > 0 && num_pos_feats > 0 && num_pos_feats % 2 == 0, "pos_embed_sine: args");
  hipLaunchKernelGGL(pos_embed_sine_kernel, dim3(mi_cdiv(H * W, 256), B, mi_cdiv(num_pos_feats, POS_CG)), dim3(256), 0, (hipStream_t)st, mask, B, H, W,
                     num_pos_feats, temperature, normalize, scale, centered, out);
  MI_CHECK_LAUNCH("pos_embed_sine");
  return MI_OK;
}
