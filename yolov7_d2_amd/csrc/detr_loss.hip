// DETR set-prediction losses for given match indices, forward + backward, without leaving the device.
// Replaces SetCriterion.loss_labels / loss_cardinality / loss_boxes (yolov7/modeling/meta_arch/detr.py:504-556) and
// their autograd backward: weighted cross entropy (no-object weight eos_coef, 'mean' = sum(w nll) / sum(w)),
// class_error (utils/misc.py:212-227, top-1), cardinality error, L1 and GIoU (utils/boxes.py:85-122) over the matched
// (query, target) pairs, both divided by num_boxes.  Consumes mi_hungarian_match's match_q / match_t / nmatch directly.
//   fwd: one wave per (image, query) row -> row state; one block reduces the rows in a fixed order -> losses[8]
//   bwd: dlogits = g_ce * w_row * (softmax - onehot) / sum_w;  dboxes = g_bbox * dL1 + g_giou * dGIoU  (dual numbers)
#include "common.h"
#include "dual4.h"

#define DL_ROW 16     // floats of state per row
#define DL_MAXB 256   // images per call (LDS-resident per-image partials)

struct DetrLossK {
  const float* logits;   // [B][Q][NC]
  const float* boxes;    // [B][Q][4] cxcywh
  const int64_t* tlab;   // [T]
  const float* tbox;     // [T][4]
  const int32_t* toff;   // [B+1]
  const int64_t* mq;     // [B][gmax]
  const int64_t* mt;     // [B][gmax]
  const int32_t* nmatch; // [B]
  int B, Q, NC, gmax;
  float eos, num_boxes;
  float* losses;         // [8]
  float* rows;           // [B*Q][DL_ROW]
};

__global__ __launch_bounds__(256) void detr_loss_rows_kernel(const DetrLossK p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.B * p.Q) return;
  const int b = row / p.Q, q = row - b * p.Q;
  // which target (if any) this query is matched to
  const int n = p.nmatch[b];
  int hit = -1;
  for (int e = lane; e < n; e += 64)
    if ((int)p.mq[(size_t)b * p.gmax + e] == q) hit = e;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hit = max(hit, __shfl_xor(hit, o, 64));
  const int t = hit >= 0 ? p.toff[b] + (int)p.mt[(size_t)b * p.gmax + hit] : -1;
  const int tc = t >= 0 ? (int)p.tlab[t] : p.NC - 1;
  const float* lg = p.logits + (size_t)row * p.NC;
  // log-softmax statistics + first arg max
  float mx = -INFINITY;
  int am = 0x7fffffff;
  for (int c = lane; c < p.NC; c += 64) {
    const float v = lg[c];
    if (v > mx) { mx = v; am = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(am, o, 64);
    if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
  }
  float sm = 0.f;
  for (int c = lane; c < p.NC; c += 64) sm += expf(lg[c] - mx);
  sm = wave_sum(sm);
  if (lane != 0) return;
  const float lsm = logf(sm);
  const float nll = -((lg[tc] - mx) - lsm);
  const float w = tc == p.NC - 1 ? p.eos : 1.f;
  float* r = p.rows + (size_t)row * DL_ROW;
  r[0] = w * nll;
  r[1] = w;
  r[2] = am != p.NC - 1 ? 1.f : 0.f;
  r[3] = (t >= 0 && am == tc) ? 1.f : 0.f;
  r[6] = mx + lsm;
  r[7] = __int_as_float(tc);
  float l1 = 0.f, gl = 0.f, d1[4] = {0.f, 0.f, 0.f, 0.f}, dg[4] = {0.f, 0.f, 0.f, 0.f};
  if (t >= 0) {
    const float* sb = p.boxes + (size_t)row * 4;
    const float* tb = p.tbox + (size_t)t * 4;
    const float inv = 1.f / p.num_boxes;
    for (int k = 0; k < 4; ++k) {
      const float df = sb[k] - tb[k];
      l1 += fabsf(df);
      d1[k] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv;
    }
    const D4 cx = dvar(sb[0], 0), cy = dvar(sb[1], 1), bw = dvar(sb[2], 2), bh = dvar(sb[3], 3);
    const D4 x0 = cx - bw * 0.5f, y0 = cy - bh * 0.5f, x1 = cx + bw * 0.5f, y1 = cy + bh * 0.5f;
    const D4 u0 = dconst(tb[0] - 0.5f * tb[2]), v0 = dconst(tb[1] - 0.5f * tb[3]);
    const D4 u1 = dconst(tb[0] + 0.5f * tb[2]), v1 = dconst(tb[1] + 0.5f * tb[3]);
    const D4 area1 = (x1 - x0) * (y1 - y0);
    const float area2 = (u1.v - u0.v) * (v1.v - v0.v);
    const D4 inter = dclamp0(dmin(x1, u1) - dmax(x0, u0)) * dclamp0(dmin(y1, v1) - dmax(y0, v0));
    const D4 uni = area1 + area2 - inter;
    const D4 iou = inter / uni;
    const D4 earea = dclamp0(dmax(x1, u1) - dmin(x0, u0)) * dclamp0(dmax(y1, v1) - dmin(y0, v0));
    const D4 giou = iou - (earea - uni) / earea;
    gl = 1.f - giou.v;
    for (int k = 0; k < 4; ++k) dg[k] = -giou.d[k] * inv;
  }
  r[4] = l1;
  r[5] = gl;
  for (int k = 0; k < 4; ++k) { r[8 + k] = d1[k]; r[12 + k] = dg[k]; }
}

// fixed-order reduction: per image over the queries (one wave per image), then over the images (wave 0)
__global__ __launch_bounds__(256) void detr_loss_reduce_kernel(const DetrLossK p) {
  __shared__ float part[7][DL_MAXB];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int b = wv; b < p.B; b += 4) {
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = lane; q < p.Q; q += 64) {
      const float* r = p.rows + ((size_t)b * p.Q + q) * DL_ROW;
      for (int k = 0; k < 6; ++k) s[k] += r[k];
    }
    for (int k = 0; k < 6; ++k) s[k] = wave_sum(s[k]);
    if (lane == 0) {
      const float G = (float)(p.toff[b + 1] - p.toff[b]);
      part[0][b] = s[0]; part[1][b] = s[1]; part[2][b] = fabsf(s[2] - G);
      part[3][b] = s[3]; part[4][b] = s[4]; part[5][b] = s[5]; part[6][b] = (float)p.nmatch[b];
    }
  }
  __syncthreads();
  if (wv != 0) return;
  float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = lane; b < p.B; b += 64)
    for (int k = 0; k < 7; ++k) s[k] += part[k][b];
  for (int k = 0; k < 7; ++k) s[k] = wave_sum(s[k]);
  if (lane == 0) {
    p.losses[0] = s[0] / s[1];
    p.losses[1] = s[6] > 0.f ? 100.f - s[3] * (100.f / s[6]) : 100.f;
    p.losses[2] = s[2] / (float)p.B;
    p.losses[3] = s[4] / p.num_boxes;
    p.losses[4] = s[5] / p.num_boxes;
    p.losses[5] = s[1];
    p.losses[6] = s[6];
    p.losses[7] = 0.f;
  }
}

__global__ __launch_bounds__(256) void detr_loss_bwd_kernel(const DetrLossK p, const float* __restrict__ gw,
                                                            float* __restrict__ dlogits, float* __restrict__ dboxes) {
  const int nrow = p.B * p.Q;
  const size_t nl = (size_t)nrow * p.NC;
  const float gce = gw[0] / p.losses[5], gb = gw[1], gg = gw[2];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nl + (size_t)nrow * 4; i += (size_t)gridDim.x * 256) {
    if (i < nl) {
      const int row = (int)(i / p.NC), c = (int)(i - (size_t)row * p.NC);
      const float* r = p.rows + (size_t)row * DL_ROW;
      const float sm = expf(p.logits[i] - r[6]);
      dlogits[i] = gce * r[1] * (sm - (c == __float_as_int(r[7]) ? 1.f : 0.f));
    } else {
      const size_t j = i - nl;
      const float* r = p.rows + (j >> 2) * DL_ROW;
      dboxes[j] = gb * r[8 + (j & 3)] + gg * r[12 + (j & 3)];
    }
  }
}

static int fill(const mi_detr_loss_desc* d, DetrLossK& k) {
  MI_REQUIRE(d && d->logits && d->boxes && d->tgt_labels && d->tgt_boxes && d->tgt_off && d->match_q && d->match_t &&
             d->nmatch && d->losses && d->rowstate, "detr_set_loss: null");
  MI_REQUIRE(d->B >= 1 && d->B <= DL_MAXB && d->Q >= 1 && d->NC >= 2 && d->gmax >= 1,
             "detr_set_loss: B %d (1..%d), Q %d, NC %d, gmax %d", d->B, DL_MAXB, d->Q, d->NC, d->gmax);
  MI_REQUIRE(d->num_boxes > 0.f, "detr_set_loss: num_boxes must be > 0");
  k.logits = d->logits; k.boxes = d->boxes; k.tlab = d->tgt_labels; k.tbox = d->tgt_boxes; k.toff = d->tgt_off;
  k.mq = d->match_q; k.mt = d->match_t; k.nmatch = d->nmatch;
  k.B = d->B; k.Q = d->Q; k.NC = d->NC; k.gmax = d->gmax; k.eos = d->eos_coef; k.num_boxes = d->num_boxes;
  k.losses = d->losses; k.rows = d->rowstate;
  return MI_OK;
}

extern "C" int mi_detr_set_loss_fwd(const mi_detr_loss_desc* d, mi_stream_t s) {
  DetrLossK k;
  if (int rc = fill(d, k)) return rc;
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(detr_loss_rows_kernel, dim3((k.B * k.Q + 3) / 4), dim3(256), 0, st, k);
  MI_CHECK_LAUNCH("detr_loss_rows");
  hipLaunchKernelGGL(detr_loss_reduce_kernel, dim3(1), dim3(256), 0, st, k);
  MI_CHECK_LAUNCH("detr_loss_reduce");
  return MI_OK;
}

extern "C" int mi_detr_set_loss_bwd(const mi_detr_loss_desc* d, const float* gw, float* dlogits, float* dboxes,
                                    mi_stream_t s) {
  DetrLossK k;
  if (int rc = fill(d, k)) return rc;
  MI_REQUIRE(gw && dlogits && dboxes, "detr_set_loss_bwd: null");
  const size_t n = (size_t)k.B * k.Q * (k.NC + 4);
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(detr_loss_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, k, gw, dlogits, dboxes);
  MI_CHECK_LAUNCH("detr_loss_bwd");
  return MI_OK;
}
