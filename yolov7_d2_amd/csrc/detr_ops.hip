// DETR set-prediction matching on the GPU, no host round trip: matching cost matrix + linear sum assignment.
// Replaces HungarianMatcher.forward (yolov7/utils/detr_utils.py:37-91): softmax class cost, L1 cdist, GIoU cost
// (utils/boxes.py:28-31,85-122) and scipy.optimize.linear_sum_assignment (detr_utils.py:89), which the reference
// runs on the CPU after a device->host copy of the cost matrix, once per decoder layer per step.
//
// The assignment is the shortest-augmenting-path (Jonker-Volgenant) algorithm scipy implements
// (scipy/optimize/rectangular_lsap, un-vendored dependency, version not pinned by the reference): fp64 duals,
// rows = the shorter side (scipy transposes a tall matrix), r = minVal + cost - u[i] - v[j] evaluated in the same
// order, so on tie-free inputs the (unique) optimum and therefore the returned index pairs are identical.
// One wave per image: the 64 lanes scan the columns, a wave reduction picks the next column.
#include <string.h>
#include "common.h"

#define LSAP_MAXN 1024  // longer side (queries / targets per image) supported by the LDS-resident state

struct MatchK {
  const float* logits;   // [B][Q][NC]
  const float* boxes;    // [B][Q][4] cxcywh
  const int64_t* tlab;   // [T]
  const float* tbox;     // [T][4] cxcywh
  const int32_t* toff;   // [B+1]
  int B, Q, NC, gmax;
  float wc, wb, wg;
  float* cost;           // [B][Q][gmax]
  int64_t* mq;           // [B][gmax]
  int64_t* mt;           // [B][gmax]
  int32_t* nmatch;       // [B]
};

// ---- cost matrix: one block per (image, query), threads over the image's targets
__global__ __launch_bounds__(64) void match_cost_kernel(const MatchK p) {
  const int b = blockIdx.y, q = blockIdx.x, lane = threadIdx.x;
  const int t0 = p.toff[b], G = p.toff[b + 1] - t0;
  if (G <= 0) return;
  const float* lg = p.logits + ((size_t)b * p.Q + q) * p.NC;
  // softmax statistics (torch.softmax: exp(x - max) / sum exp(x - max), fp32)
  float mx = -INFINITY;
  for (int c = lane; c < p.NC; c += 64) mx = fmaxf(mx, lg[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sm = 0.f;
  for (int c = lane; c < p.NC; c += 64) sm += expf(lg[c] - mx);
  sm = wave_sum(sm);
  const float* bx = p.boxes + ((size_t)b * p.Q + q) * 4;
  const float cx = bx[0], cy = bx[1], w = bx[2], h = bx[3];
  const float x0 = cx - 0.5f * w, y0 = cy - 0.5f * h, x1 = cx + 0.5f * w, y1 = cy + 0.5f * h;
  const float area1 = (x1 - x0) * (y1 - y0);
  for (int g = lane; g < G; g += 64) {
    const float* tb = p.tbox + (size_t)(t0 + g) * 4;
    const float tcx = tb[0], tcy = tb[1], tw = tb[2], th = tb[3];
    const int lab = (int)p.tlab[t0 + g];
    const float cost_class = -(expf(lg[lab] - mx) / sm);
    float cost_bbox = fabsf(cx - tcx);
    cost_bbox += fabsf(cy - tcy);
    cost_bbox += fabsf(w - tw);
    cost_bbox += fabsf(h - th);
    const float u0 = tcx - 0.5f * tw, v0 = tcy - 0.5f * th, u1 = tcx + 0.5f * tw, v1 = tcy + 0.5f * th;
    const float area2 = (u1 - u0) * (v1 - v0);
    const float iw = fmaxf(fminf(x1, u1) - fmaxf(x0, u0), 0.f), ih = fmaxf(fminf(y1, v1) - fmaxf(y0, v0), 0.f);
    const float inter = iw * ih;
    const float uni = area1 + area2 - inter;
    const float iou = inter / uni;
    const float ew = fmaxf(fmaxf(x1, u1) - fminf(x0, u0), 0.f), eh = fmaxf(fmaxf(y1, v1) - fminf(y0, v0), 0.f);
    const float earea = ew * eh;
    const float giou = iou - (earea - uni) / earea;
    const float cost_giou = -giou;
    float c = p.wb * cost_bbox + p.wc * cost_class;
    c = c + p.wg * cost_giou;
    p.cost[((size_t)b * p.Q + q) * p.gmax + g] = c;
  }
}

// ---- linear sum assignment, one wave per image
__global__ __launch_bounds__(64) void lsap_kernel(const MatchK p) {
  __shared__ double u[LSAP_MAXN], v[LSAP_MAXN], sp[LSAP_MAXN];
  __shared__ int path[LSAP_MAXN], row4col[LSAP_MAXN], col4row[LSAP_MAXN];
  __shared__ unsigned char SR[LSAP_MAXN], SC[LSAP_MAXN];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int G = p.toff[b + 1] - p.toff[b];
  const bool tr = p.Q > G;              // scipy transposes a tall matrix: rows = targets, cols = queries
  const int nr = tr ? G : p.Q, nc = tr ? p.Q : G;
  const float* Cb = p.cost + (size_t)b * p.Q * p.gmax;
  auto cost = [&](int i, int j) -> double {  // i in [0,nr), j in [0,nc)
    return tr ? (double)Cb[(size_t)j * p.gmax + i] : (double)Cb[(size_t)i * p.gmax + j];
  };
  if (lane == 0) p.nmatch[b] = nr;
  if (nr <= 0) return;
  for (int k = lane; k < LSAP_MAXN; k += 64) {
    u[k] = 0.0; v[k] = 0.0; row4col[k] = -1; col4row[k] = -1; path[k] = -1;
  }
  __syncthreads();
  for (int cur = 0; cur < nr; ++cur) {
    for (int k = lane; k < nc; k += 64) { sp[k] = INFINITY; SC[k] = 0; }
    for (int k = lane; k < nr; k += 64) SR[k] = 0;
    __syncthreads();
    double minVal = 0.0;
    int i = cur, sink = -1;
    while (sink < 0) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      double best = INFINITY;
      int bestj = -1, bestfree = 0;
      for (int j = lane; j < nc; j += 64) {
        if (SC[j]) continue;
        const double r = minVal + cost(i, j) - ui - v[j];
        if (r < sp[j]) { sp[j] = r; path[j] = i; }
        const double s = sp[j];
        const int fr = row4col[j] < 0;
        if (s < best || (s == best && (fr > bestfree || (fr == bestfree && j > bestj)))) { best = s; bestj = j; bestfree = fr; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oj = __shfl_xor(bestj, o, 64), of = __shfl_xor(bestfree, o, 64);
        if (oj >= 0 && (bestj < 0 || ob < best || (ob == best && (of > bestfree || (of == bestfree && oj > bestj))))) {
          best = ob; bestj = oj; bestfree = of;
        }
      }
      minVal = best;
      const int j = bestj;
      if (j < 0) {
        // no finite candidate: NaN / inf costs (diverged logits or boxes).  scipy.optimize.linear_sum_assignment raises
        // ValueError("matrix contains invalid numeric entries") here; the kernel flags the image (nmatch = -1, checked
        // on the host by HungarianMatcher) instead of indexing row4col[-1] and spinning.  bestj is wave-uniform after
        // the reduction, so the whole (one-wave) block leaves together.
        if (lane == 0) p.nmatch[b] = -1;
        return;
      }
      if (row4col[j] < 0) sink = j; else i = row4col[j];
      __syncthreads();
      if (lane == 0) SC[j] = 1;
      __syncthreads();
    }
    // dual updates (same expressions as scipy)
    if (lane == 0) u[cur] += minVal;
    __syncthreads();
    for (int k = lane; k < nr; k += 64)
      if (SR[k] && k != cur) u[k] += minVal - sp[col4row[k]];
    for (int k = lane; k < nc; k += 64)
      if (SC[k]) v[k] -= minVal - sp[k];
    __syncthreads();
    if (lane == 0) {  // augment
      int j = sink;
      while (true) {
        const int ii = path[j];
        row4col[j] = ii;
        const int t = col4row[ii];
        col4row[ii] = j;
        j = t;
        if (ii == cur) break;
      }
    }
    __syncthreads();
  }
  // output pairs sorted by query index (scipy returns sorted row indices)
  int64_t* mq = p.mq + (size_t)b * p.gmax;
  int64_t* mt = p.mt + (size_t)b * p.gmax;
  for (int k = lane; k < nr; k += 64) {
    const int qk = tr ? col4row[k] : k, tk = tr ? k : col4row[k];
    int rank = 0;
    for (int m = 0; m < nr; ++m) {
      const int qm = tr ? col4row[m] : m;
      rank += qm < qk;
    }
    mq[rank] = qk;
    mt[rank] = tk;
  }
}

extern "C" int mi_hungarian_match(const float* logits, const float* boxes, const int64_t* tgt_labels,
                                  const float* tgt_boxes, const int32_t* tgt_off, int B, int Q, int NC, int gmax,
                                  float w_class, float w_bbox, float w_giou, float* cost, int64_t* match_q,
                                  int64_t* match_t, int32_t* nmatch, mi_stream_t st) {
  MI_REQUIRE(logits && boxes && tgt_labels && tgt_boxes && tgt_off && cost && match_q && match_t && nmatch,
             "hungarian_match: null");
  MI_REQUIRE(B > 0 && Q > 0 && Q <= LSAP_MAXN && NC > 0 && gmax > 0 && gmax <= LSAP_MAXN,
             "hungarian_match: Q %d / gmax %d (<= %d)", Q, gmax, LSAP_MAXN);
  MatchK k;
  k.logits = logits; k.boxes = boxes; k.tlab = tgt_labels; k.tbox = tgt_boxes; k.toff = tgt_off;
  k.B = B; k.Q = Q; k.NC = NC; k.gmax = gmax; k.wc = w_class; k.wb = w_bbox; k.wg = w_giou;
  k.cost = cost; k.mq = match_q; k.mt = match_t; k.nmatch = nmatch;
  hipStream_t s = (hipStream_t)st;
  hipLaunchKernelGGL(match_cost_kernel, dim3(Q, B), dim3(64), 0, s, k);
  MI_CHECK_LAUNCH("match_cost");
  hipLaunchKernelGGL(lsap_kernel, dim3(B), dim3(64), 0, s, k);
  MI_CHECK_LAUNCH("lsap");
  return MI_OK;
}

// assignment only, on a caller-provided cost matrix [B][Q][gmax] with ng[b] valid columns (tests, other matchers)
extern "C" int mi_lsap(const float* cost, const int32_t* tgt_off, int B, int Q, int gmax, int64_t* match_q,
                       int64_t* match_t, int32_t* nmatch, mi_stream_t st) {
  MI_REQUIRE(cost && tgt_off && match_q && match_t && nmatch, "lsap: null");
  MI_REQUIRE(B > 0 && Q > 0 && Q <= LSAP_MAXN && gmax > 0 && gmax <= LSAP_MAXN, "lsap: Q %d / gmax %d (<= %d)", Q, gmax,
             LSAP_MAXN);
  MatchK k;
  memset(&k, 0, sizeof(k));
  k.toff = tgt_off; k.B = B; k.Q = Q; k.gmax = gmax; k.cost = (float*)cost; k.mq = match_q; k.mt = match_t;
  k.nmatch = nmatch;
  hipLaunchKernelGGL(lsap_kernel, dim3(B), dim3(64), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("lsap");
  return MI_OK;
}

// ------------------------------------------------------------------ box utilities of the DETR path
// box_cxcywh_to_xyxy / box_xyxy_to_cxcywh / box_iou / generalized_box_iou (yolov7/utils/boxes.py:28-37,85-122) as
// elementwise / pairwise kernels in the reference's operation order (this file is built with -ffp-contract=off).  The
// training step never calls them (matching cost and GIoU loss are fused in mi_hungarian_match / mi_detr_set_loss_*);
// they serve target preparation, inference and callers of the reference's API.
__global__ void box_convert_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int to_cxcywh) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 b = ((const float4*)in)[i];
  float4 o;
  if (to_cxcywh) {   // (x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0
    o.x = (b.x + b.z) / 2.f; o.y = (b.y + b.w) / 2.f; o.z = b.z - b.x; o.w = b.w - b.y;
  } else {           // x_c - 0.5 w, y_c - 0.5 h, x_c + 0.5 w, y_c + 0.5 h
    o.x = b.x - 0.5f * b.z; o.y = b.y - 0.5f * b.w; o.z = b.x + 0.5f * b.z; o.w = b.y + 0.5f * b.w;
  }
  ((float4*)out)[i] = o;
}
extern "C" int mi_box_convert(const float* in, float* out, int64_t n, int to_cxcywh, mi_stream_t st) {
  MI_REQUIRE(in && out && n >= 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "box_convert: null / unaligned");
  if (n == 0) return MI_OK;
  hipLaunchKernelGGL(box_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)st, in, out, n, to_cxcywh);
  MI_CHECK_LAUNCH("box_convert");
  return MI_OK;
}

// iou[n][m], uni[n][m] (may be null), giou[n][m] (may be null); *degenerate |= 1 / 2 when a box of the first / second
// set has x1 < x0 or y1 < y0 (what generalized_box_iou asserts on the host)
__global__ void box_iou_pairwise_kernel(const float* __restrict__ b1, int n, const float* __restrict__ b2, int m,
                                        float* __restrict__ iou, float* __restrict__ uni, float* __restrict__ giou,
                                        int* __restrict__ degenerate) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * m) return;
  const int i = (int)(idx / m), j = (int)(idx - (int64_t)i * m);
  const float4 a = ((const float4*)b1)[i], b = ((const float4*)b2)[j];
  if (degenerate) {
    if (j == 0 && !(a.z >= a.x && a.w >= a.y)) atomicOr(degenerate, 1);
    if (i == 0 && !(b.z >= b.x && b.w >= b.y)) atomicOr(degenerate, 2);
  }
  const float area1 = (a.z - a.x) * (a.w - a.y), area2 = (b.z - b.x) * (b.w - b.y);
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f), h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float inter = w * h;
  const float un = area1 + area2 - inter;
  const float io = inter / un;
  iou[idx] = io;
  if (uni) uni[idx] = un;
  if (giou) {
    const float ew = fmaxf(fmaxf(a.z, b.z) - fminf(a.x, b.x), 0.f), eh = fmaxf(fmaxf(a.w, b.w) - fminf(a.y, b.y), 0.f);
    const float ea = ew * eh;
    giou[idx] = io - (ea - un) / ea;
  }
}
extern "C" int mi_box_iou_pairwise(const float* boxes1, int n, const float* boxes2, int m, float* iou, float* uni, float* giou,
                                   int32_t* degenerate, mi_stream_t st) {
  MI_REQUIRE(n >= 0 && m >= 0, "box_iou_pairwise: sizes");
  if (n == 0 || m == 0) return MI_OK;
  MI_REQUIRE(boxes1 && boxes2 && iou && ((uintptr_t)boxes1 % 16) == 0 && ((uintptr_t)boxes2 % 16) == 0, "box_iou_pairwise: null / unaligned");
  const int64_t tot = (int64_t)n * m;
  hipLaunchKernelGGL(box_iou_pairwise_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)st, boxes1, n, boxes2, m,
                     iou, uni, giou, degenerate);
  MI_CHECK_LAUNCH("box_iou_pairwise");
  return MI_OK;
}
