// SparseInstCriterion's scalar half and the matcher's cost matrix as four launches (round 6).
//
// yolov7/modeling/loss/sparseinst_loss.py:190-297 (SparseInstCriterion: loss_labels = sigmoid focal loss on a one-hot target,
// loss_masks_with_iou_objectness = BCE / dice from the matched pairs' mask statistics + BCE of the objectness logit against
// the pairs' mask IoU) and :300-354 (SparseInstMatcher: cost = dice^alpha * prob^beta).  A captured SparseInst-R50 step
// spent ~150 torch launches of ~5 us on this arithmetic over [8, 100, 80] / [8, 96] tensors - index_put (a radix sort),
// gathers, a dozen elementwise kernels per loss and their autograd twins: 0.75 of a 12 ms step.  Here:
//   mi_sparseinst_match_cost     cost[b][n][t] = -(dice[b][n][t]^alpha * sigmoid(logit[b][n][label[b][t]])^beta)
//   mi_sparseinst_pairs          (match_q, match_t, nmatch) -> the pair table the mask kernels read, the validity flags, every
//                                query's matched class / pair, K = max(sum nmatch, 1)
//   mi_sparseinst_head_loss      the four weighted losses from the logits, the objectness scores and the pairs' mask statistics
//   mi_sparseinst_head_loss_bwd  d logits, d scores and the two coefficients mi_sparseinst_mask_grad_dev takes, scaled by the
//                                four upstream gradients
// Sums run in ONE block in a fixed order with fp64 accumulators: bit-reproducible, no atomics.
// Formulas (fp32, the operations of the torch calls they replace):
//   BCE-with-logits  (1 - y) x + max(-x, 0) + log1p(exp(-|x|))                      (ATen binary_cross_entropy_with_logits)
//   focal            a_t * BCE * (1 - p_t)^gamma, p_t = p t + (1 - p)(1 - t), a_t = a t + (1 - a)(1 - t)   (fvcore, un-vendored)
//   d focal / d x    -a_t (2 t - 1) (1 - p_t)^gamma (gamma p_t BCE + (1 - p_t))       for t in {0, 1}
#include "common.h"
#include <cstring>

__device__ __forceinline__ float si_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float si_bce(float x, float y) { return (1.f - y) * x + fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float si_pow(float v, float e) { return e == 2.f ? v * v : (e == 1.f ? v : powf(v, e)); }

// ---------------------------------------------------------------------------------------------------------------- matcher
struct SiCostK {
  const float *num, *s2, *t2, *logits;
  const int64_t* labels;
  float* cost;
  int B, N, Np, C, cap;
  float alpha, beta;
};
__global__ __launch_bounds__(256) void si_match_cost_kernel(const SiCostK p) {
  const int total = p.B * p.N * p.cap;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int t = i % p.cap, r = i / p.cap;
    const int n = r % p.N, b = r / p.N;
    const float num = p.num[((int64_t)b * p.Np + n) * p.cap + t];
    const float score = (2.f * num) / (p.s2[b * p.Np + n] + p.t2[b * p.cap + t] + 1e-4f);
    const int64_t lab = p.labels[b * p.cap + t];
    const float x = p.logits[((int64_t)b * p.N + n) * p.C + (int)lab];
    const float prob = 1.f / (1.f + expf(-x));
    p.cost[i] = -(powf(score, p.alpha) * powf(prob, p.beta));
  }
}
extern "C" int mi_sparseinst_match_cost(const float* num, const float* s2, const float* t2, const float* logits,
                                        const int64_t* labels, int B, int N, int Np, int C, int cap, float alpha, float beta,
                                        float* cost, mi_stream_t st) {
  MI_REQUIRE(num && s2 && t2 && logits && labels && cost && B >= 1 && N >= 1 && Np >= N && C >= 1 && cap >= 1, "sparseinst_match_cost: args");
  SiCostK k;
  k.num = num; k.s2 = s2; k.t2 = t2; k.logits = logits; k.labels = labels; k.cost = cost;
  k.B = B; k.N = N; k.Np = Np; k.C = C; k.cap = cap; k.alpha = alpha; k.beta = beta;
  const int total = B * N * cap;
  hipLaunchKernelGGL(si_match_cost_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("sparseinst_match_cost");
  return MI_OK;
}

// -------------------------------------------------------------------------------------------------------------- criterion
__global__ __launch_bounds__(1024) void si_pairs_kernel(const mi_sparseinst_loss_desc d) {
  const int tid = threadIdx.x;
  for (int i = tid; i < d.B * d.N; i += 1024) {
    d.row_cls[i] = -1;
    d.row_pair[i] = -1;
  }
  __syncthreads();
  for (int i = tid; i < d.B * d.cap; i += 1024) {
    const int b = i / d.cap, j = i - b * d.cap;
    const int nm = d.nmatch[b] > 0 ? d.nmatch[b] : 0;      // (an invalid cost matrix - nmatch < 0 - contributes no pair)
    const bool v = j < nm;
    const int q = (int)d.match_q[i], t = (int)d.match_t[i];
    d.pairs[i * 3 + 0] = v ? b : -1;
    d.pairs[i * 3 + 1] = q;
    d.pairs[i * 3 + 2] = b * d.cap + t;
    d.valid[i] = v ? 1.f : 0.f;
    if (v && q >= 0 && q < d.N) {            // the assignment is one-to-one: a query is written at most once
      d.row_cls[b * d.N + q] = (int)d.labels[b * d.cap + t];
      d.row_pair[b * d.N + q] = i;
    }
  }
  if (tid == 0) {
    int K = 0;
    for (int b = 0; b < d.B; ++b) K += d.nmatch[b] > 0 ? d.nmatch[b] : 0;
    d.kdev[0] = K > 1 ? (float)K : 1.f;
  }
}

__device__ __forceinline__ double si_block_sum(double v, double* red) {      // 1024 threads, fixed order; every thread gets the total
  const int tid = threadIdx.x;
  __syncthreads();
  red[tid] = v;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  return red[0];
}

__device__ __forceinline__ float si_pair_iou(const float* st) { return st[4] / (st[6] + st[5] - st[4] + 1e-6f); }

__global__ __launch_bounds__(1024) void si_head_loss_kernel(const mi_sparseinst_loss_desc d) {
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  const float inv_num = d.inv_num[0], kdev = d.kdev[0];
  double ce = 0.0, bce = 0.0, dice = 0.0, obj = 0.0;
  if (d.use_labels) {
    const int total = d.B * d.N * d.C;
    for (int i = tid; i < total; i += 1024) {
      const int c = i % d.C, row = i / d.C;
      const float x = d.logits[i];
      const float t = d.row_cls[row] == c ? 1.f : 0.f;
      const float p = si_sigmoid(x);
      const float pt = p * t + (1.f - p) * (1.f - t);
      const float at = d.alpha * t + (1.f - d.alpha) * (1.f - t);
      ce += (double)(at * (si_bce(x, t) * si_pow(1.f - pt, d.gamma)));
    }
  }
  if (d.use_masks) {
    for (int k = tid; k < d.B * d.cap; k += 1024) {
      const float* st = d.stats + (int64_t)k * 8;
      const float v = d.valid[k];
      bce += (double)st[0];
      dice += (double)((1.f - 2.f * st[1] / (st[2] + st[3] + 1e-4f)) * v);
      if (v != 0.f) {
        const int b = k / d.cap;
        const float s = d.scores[b * d.N + (int)d.match_q[k]];
        obj += (double)si_bce(s, si_pair_iou(st));
      }
    }
  }
  ce = si_block_sum(ce, red);
  bce = si_block_sum(bce, red);
  dice = si_block_sum(dice, red);
  obj = si_block_sum(obj, red);
  if (tid == 0) {
    d.losses[0] = d.use_labels ? (float)ce * inv_num * d.w_ce : 0.f;
    d.losses[1] = d.use_masks ? (float)bce / (kdev * (float)d.P) * d.w_mask : 0.f;
    d.losses[2] = d.use_masks ? (float)dice * inv_num * d.w_dice : 0.f;
    d.losses[3] = d.use_masks ? (float)obj / kdev * d.w_obj : 0.f;
  }
}

__global__ __launch_bounds__(256) void si_head_loss_bwd_kernel(const mi_sparseinst_loss_desc d) {
  const float inv_num = d.inv_num[0], kdev = d.kdev[0];
  const float g_ce = d.gup[0] * d.w_ce * inv_num;
  const int total = d.B * d.N * d.C;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int c = i % d.C, row = i / d.C;
    float dl = 0.f;
    if (d.use_labels) {
      const float x = d.logits[i];
      const float t = d.row_cls[row] == c ? 1.f : 0.f;
      const float p = si_sigmoid(x);
      const float pt = p * t + (1.f - p) * (1.f - t);
      const float at = d.alpha * t + (1.f - d.alpha) * (1.f - t);
      const float om = 1.f - pt;
      dl = g_ce * (-at * (2.f * t - 1.f) * si_pow(om, d.gamma) * (d.gamma * pt * si_bce(x, t) + om));
    }
    d.dlogits[i] = dl;
    if (c == 0) {
      float ds = 0.f;
      const int k = d.row_pair[row];
      if (d.use_masks && k >= 0) ds = d.gup[3] * d.w_obj / kdev * (si_sigmoid(d.scores[row]) - si_pair_iou(d.stats + (int64_t)k * 8));
      d.dscores[row] = ds;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    d.coef[0] = d.gup[1] * d.w_mask / (kdev * (float)d.P);
    d.coef[1] = d.gup[2] * d.w_dice * inv_num;
  }
}

static int si_check(const mi_sparseinst_loss_desc* d, const char* what) {
  MI_REQUIRE(d && d->B >= 1 && d->N >= 1 && d->C >= 1 && d->cap >= 1 && d->P >= 1, "%s: sizes", what);
  MI_REQUIRE(d->match_q && d->match_t && d->nmatch && d->labels && d->pairs && d->valid && d->row_cls && d->row_pair && d->kdev, "%s: pair tables", what);
  return MI_OK;
}
extern "C" int mi_sparseinst_pairs(const mi_sparseinst_loss_desc* d, mi_stream_t st) {
  if (si_check(d, "sparseinst_pairs") != MI_OK) return MI_EINVAL;
  hipLaunchKernelGGL(si_pairs_kernel, dim3(1), dim3(1024), 0, (hipStream_t)st, *d);
  MI_CHECK_LAUNCH("sparseinst_pairs");
  return MI_OK;
}
extern "C" int mi_sparseinst_head_loss(const mi_sparseinst_loss_desc* d, mi_stream_t st) {
  if (si_check(d, "sparseinst_head_loss") != MI_OK) return MI_EINVAL;
  MI_REQUIRE(d->losses && d->inv_num && (!d->use_labels || d->logits) && (!d->use_masks || (d->stats && d->scores)), "sparseinst_head_loss: args");
  hipLaunchKernelGGL(si_head_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)st, *d);
  MI_CHECK_LAUNCH("sparseinst_head_loss");
  return MI_OK;
}
extern "C" int mi_sparseinst_head_loss_bwd(const mi_sparseinst_loss_desc* d, mi_stream_t st) {
  if (si_check(d, "sparseinst_head_loss_bwd") != MI_OK) return MI_EINVAL;
  MI_REQUIRE(d->gup && d->dlogits && d->dscores && d->coef && d->inv_num && d->logits && d->scores && (!d->use_masks || d->stats),
             "sparseinst_head_loss_bwd: args");
  const int total = d->B * d->N * d->C;
  hipLaunchKernelGGL(si_head_loss_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)st, *d);
  MI_CHECK_LAUNCH("sparseinst_head_loss_bwd");
  return MI_OK;
}
