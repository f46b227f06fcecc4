// Class-aware greedy NMS with torchvision.ops.batched_nms semantics, as called from
// postprocess (yolov7/utils/boxes.py:171-210, call at :199).  torchvision is not vendored in the
// reference; the arithmetic restated here is torchvision's nms kernel:
//   area = (x2-x1)*(y2-y1); inter = max(0,xx2-xx1)*max(0,yy2-yy1); suppress iff inter/(a+b-inter) > thr;
//   candidates visited in descending score order; output in descending score order.
// batched_nms has two branches (coordinate trick for <= 4000 box coordinates, per-class loop above);
// `mode` selects which arithmetic is reproduced (they differ only by fp32 rounding of the offsets).
//
// gfx950 design: (1) rank sort by score (O(n^2) compares, trivially parallel), (2) 64x64 tiled
// suppression bit-matrix, one 64-bit word per (box, 64-box column chunk) = one bit per wave lane,
// (3) a single-wavefront scan: per 64-box chunk lane k owns box k; the intra-chunk dependency is
// resolved with scalar 64-bit mask ops, then the kept rows are OR-ed into the running removed-bitmap
// that lives distributed over the 64 lanes.
#include "common.h"

__global__ __launch_bounds__(256) void nms_rank_kernel(const float* __restrict__ scores, int n, int32_t* order) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  __shared__ float ss[256];
  const float si = i < n ? scores[i] : 0.f;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + threadIdx.x;
    __syncthreads();
    ss[threadIdx.x] = j < n ? scores[j] : -INFINITY;
    __syncthreads();
    const int lim = min(256, n - j0);
    if (i < n)
      for (int t = 0; t < lim; ++t) {
        const float sj = ss[t];
        rank += (sj > si) || (sj == si && (j0 + t) < i);
      }
  }
  if (i < n) order[rank] = i;
}

// gather boxes in sorted order; mode 1 adds the batched_nms coordinate-trick offset idx*(max+1)
__global__ __launch_bounds__(256) void nms_gather_kernel(const float* __restrict__ boxes, const float* __restrict__ idxs,
                                                         const int32_t* __restrict__ order, int n, int mode,
                                                         const float* maxc, float* sboxes, float* scls) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int o = order[i];
  float off = 0.f;
  if (mode == 1) off = idxs[o] * (maxc[0] + 1.0f);
#pragma unroll
  for (int q = 0; q < 4; ++q) sboxes[i * 4 + q] = boxes[o * 4 + q] + off;
  scls[i] = idxs[o];
}

__global__ __launch_bounds__(256) void nms_max_kernel(const float* __restrict__ boxes, int n4, float* maxc) {
  __shared__ float sm[4];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n4; i += 256) m = fmaxf(m, boxes[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) maxc[0] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// mask[i][w] bit k: box (w*64+k) is suppressed by box i (only k with w*64+k > i)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sboxes, const float* __restrict__ scls,
                                                      int n, float thr, int mode, uint64_t* mask, int nw) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  __shared__ float cbx[64][5];
  const int lane = threadIdx.x;
  const int cj = cb * 64 + lane;
  if (cj < n) {
#pragma unroll
    for (int q = 0; q < 4; ++q) cbx[lane][q] = sboxes[cj * 4 + q];
    cbx[lane][4] = scls[cj];
  }
  __syncthreads();
  const int i = rb * 64 + lane;
  if (i >= n) return;
  const float x1 = sboxes[i * 4 + 0], y1 = sboxes[i * 4 + 1], x2 = sboxes[i * 4 + 2], y2 = sboxes[i * 4 + 3];
  const float ci = scls[i];
  const float ai = (x2 - x1) * (y2 - y1);
  const int lim = min(64, n - cb * 64);
  uint64_t bits = 0;
  for (int k = (rb == cb ? lane + 1 : 0); k < lim; ++k) {
    if (mode == 0 && cbx[k][4] != ci) continue;
    const float xx1 = fmaxf(x1, cbx[k][0]), yy1 = fmaxf(y1, cbx[k][1]);
    const float xx2 = fminf(x2, cbx[k][2]), yy2 = fminf(y2, cbx[k][3]);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float aj = (cbx[k][2] - cbx[k][0]) * (cbx[k][3] - cbx[k][1]);
    const float ovr = inter / (ai + aj - inter);
    if (ovr > thr) bits |= 1ull << k;
  }
  mask[(size_t)i * nw + cb] = bits;
}

// single-wavefront greedy scan
__global__ __launch_bounds__(64) void nms_scan_kernel(const uint64_t* __restrict__ mask, const int32_t* __restrict__ order,
                                                      int n, int nw, int64_t* keep, int32_t* n_keep) {
  extern __shared__ uint64_t removed[];  // [nw]
  const int lane = threadIdx.x;
  for (int w = lane; w < nw; w += 64) removed[w] = 0;
  __syncthreads();
  int nk = 0;
  for (int c = 0; c < nw; ++c) {
    const int i = c * 64 + lane;
    // diagonal word of my row: which later boxes of this chunk I would suppress
    const uint64_t diag = (i < n) ? mask[(size_t)i * nw + c] : 0ull;
    uint64_t rem = removed[c];
    if (n - c * 64 < 64) rem |= ~0ull << (n - c * 64);  // lanes beyond n are "removed"
    // resolve the chunk sequentially with wave-uniform scalar ops
    uint64_t kept = 0;
    for (int k = 0; k < 64; ++k) {
      const uint64_t dk = __shfl(diag, k, 64);
      if (!((rem >> k) & 1ull)) {
        kept |= 1ull << k;
        rem |= dk;
      }
    }
    // emit kept boxes in order
    const bool me = (kept >> lane) & 1ull;
    const int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
    if (me) keep[pos] = (int64_t)order[i];
    nk += __popcll(kept);
    // OR the kept rows into the later words of the removed bitmap (lane-distributed)
    for (int w = c + 1 + lane; w < nw; w += 64) {
      uint64_t r = removed[w];
      uint64_t kk = kept;
      while (kk) {
        const int k = __ffsll((long long)kk) - 1;
        kk &= kk - 1;
        r |= mask[(size_t)(c * 64 + k) * nw + w];
      }
      removed[w] = r;
    }
    __syncthreads();
  }
  if (lane == 0) *n_keep = nk;
}

extern "C" int mi_batched_nms(const float* boxes, const float* scores, const float* idxs, int n, float iou_thr,
                              int32_t* order, uint64_t* mask, float* sboxes, int64_t* keep, int32_t* n_keep,
                              mi_stream_t st) {
  return mi_batched_nms_ex(boxes, scores, idxs, n, iou_thr, -1, order, mask, sboxes, keep, n_keep, st);
}
extern "C" int mi_batched_nms_ex(const float* boxes, const float* scores, const float* idxs, int n, float iou_thr,
                                 int arithmetic, int32_t* order, uint64_t* mask, float* sboxes, int64_t* keep,
                                 int32_t* n_keep, mi_stream_t st) {
  MI_REQUIRE(n_keep && (n == 0 || (boxes && scores && idxs && order && mask && sboxes && keep)), "nms: null");
  MI_REQUIRE(arithmetic >= -1 && arithmetic <= 1, "nms: arithmetic %d (-1 torchvision's choice, 0 per class, 1 offsets)", arithmetic);
  hipStream_t s = (hipStream_t)st;
  if (n == 0) {
    if (hipMemsetAsync(n_keep, 0, sizeof(int32_t), s) != hipSuccess) MI_FAIL(MI_ELAUNCH, "nms: memset");
    return MI_OK;
  }
  MI_REQUIRE(n <= 65536, "nms: n %d too large", n);
  const int nw = mi_cdiv(n, 64);
  // torchvision: coordinate trick when boxes.numel() <= 4000, per-class NMS above
  const int mode = arithmetic >= 0 ? arithmetic : ((4 * (int64_t)n > 4000) ? 0 : 1);
  float* scls = sboxes + (size_t)n * 4;   // caller provides n*5 + 1 floats
  float* maxc = scls + n;
  const int nb = mi_cdiv(n, 256);
  hipLaunchKernelGGL(nms_rank_kernel, dim3(nb), dim3(256), 0, s, scores, n, order);
  if (mode == 1) hipLaunchKernelGGL(nms_max_kernel, dim3(1), dim3(256), 0, s, boxes, 4 * n, maxc);
  hipLaunchKernelGGL(nms_gather_kernel, dim3(nb), dim3(256), 0, s, boxes, idxs, order, n, mode, maxc, sboxes, scls);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nw, nw), dim3(64), 0, s, sboxes, scls, n, iou_thr, mode, mask, nw);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), nw * sizeof(uint64_t), s, mask, order, n, nw, keep, n_keep);
  MI_CHECK_LAUNCH("nms");
  return MI_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Soft-NMS, class-aware (batched_softnms / softnms / scale_by_iou, yolov7/modeling/meta_arch/utils.py:7-63):
// repeatedly take the highest-scoring undone box, rescale the scores of the undone boxes of ITS class by
// exp(-iou^2 / sigma) ("gaussian") or (1 - iou where iou >= sigma) ("linear"), retire what fell below the score threshold.
// The reference loops class by class; classes do not interact, so one global sequence of "take the maximum" visits
// every class's boxes in that class's own order and gives the same scores.  A class's last undone box is never
// "taken" by the reference (while undone.sum() > 1); taking it here touches nothing.
// One block of 1024 threads, SNMS_PER boxes per thread in registers (n <= 16384): an iteration is a block-wide arg-max
// (score descending, index ascending = torch.argmax's first maximum) and one pass over the registers.
#define SNMS_T 1024
#define SNMS_PER 16
__global__ __launch_bounds__(SNMS_T) void softnms_kernel(const float* __restrict__ boxes, float* __restrict__ scores,
                                                         const float* __restrict__ idxs, int n, float sigma, float thr,
                                                         int linear) {
  __shared__ float s_v[SNMS_T / 64];
  __shared__ int s_i[SNMS_T / 64];
  __shared__ float s_top[6];
  __shared__ int s_topi;
  const int tid = threadIdx.x;
  float bx[SNMS_PER][4], sc[SNMS_PER], cl[SNMS_PER];
  bool undone[SNMS_PER];
#pragma unroll
  for (int k = 0; k < SNMS_PER; ++k) {
    const int i = k * SNMS_T + tid;
    undone[k] = false;
    if (i < n) {
      for (int c = 0; c < 4; ++c) bx[k][c] = boxes[(size_t)i * 4 + c];
      sc[k] = scores[i];
      cl[k] = idxs[i];
      undone[k] = sc[k] >= thr;
    }
  }
  for (int iter = 0; iter < n; ++iter) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < SNMS_PER; ++k) {
      const int i = k * SNMS_T + tid;
      if (undone[k] && (sc[k] > bv || (sc[k] == bv && i < bi))) { bv = sc[k]; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      float v = s_v[0];
      int ix = s_i[0];
      for (int w = 1; w < SNMS_T / 64; ++w)
        if (s_v[w] > v || (s_v[w] == v && s_i[w] < ix)) { v = s_v[w]; ix = s_i[w]; }
      s_topi = ix;
    }
    __syncthreads();
    const int top = s_topi;
    if (top == 0x7fffffff) break;   // nothing undone
    if ((top % SNMS_T) == tid) {    // the owner publishes the box and retires it
      const int k = top / SNMS_T;
#pragma unroll
      for (int q = 0; q < SNMS_PER; ++q)
        if (q == k) {
          s_top[0] = bx[q][0]; s_top[1] = bx[q][1]; s_top[2] = bx[q][2]; s_top[3] = bx[q][3]; s_top[4] = cl[q];
          undone[q] = false;
        }
    }
    __syncthreads();
    const float t0 = s_top[0], t1 = s_top[1], t2 = s_top[2], t3 = s_top[3], tc = s_top[4];
    const float tarea = (t2 - t0) * (t3 - t1);
#pragma unroll
    for (int k = 0; k < SNMS_PER; ++k) {
      if (undone[k] && cl[k] == tc) {
        // iou(): clamp the candidate's corners into the top box (utils.py:7-18)
        const float x1 = fmaxf(bx[k][0], t0), y1 = fmaxf(bx[k][1], t1), x2 = fminf(bx[k][2], t2), y2 = fminf(bx[k][3], t3);
        const float inter = fmaxf(x2 - x1, 0.f) * fmaxf(y2 - y1, 0.f);
        const float area = (bx[k][2] - bx[k][0]) * (bx[k][3] - bx[k][1]);
        const float iou = inter / (tarea + area - inter);
        const float scale = linear ? (iou >= sigma ? 1.f - iou : 1.f) : expf(-(iou * iou) / sigma);
        sc[k] *= scale;
        if (sc[k] < thr) undone[k] = false;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < SNMS_PER; ++k) {
    const int i = k * SNMS_T + tid;
    if (i < n) scores[i] = sc[k];
  }
}
// keep = indices with score > thr, descending score (ties: ascending index)
__global__ __launch_bounds__(256) void softnms_keep_kernel(const float* __restrict__ scores, int n, float thr,
                                                           int64_t* __restrict__ keep, int32_t* __restrict__ n_keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = scores[i];
  if (!(v > thr)) return;
  int rank = 0;
  for (int j = 0; j < n; ++j) {
    const float u = scores[j];
    rank += (u > thr) && (u > v || (u == v && j < i));
  }
  keep[rank] = i;
  atomicAdd(n_keep, 1);
}
extern "C" int mi_batched_softnms(const float* boxes, float* scores, const float* idxs, int n, float sigma,
                                  float score_threshold, int linear, int64_t* keep, int32_t* n_keep, mi_stream_t st) {
  MI_REQUIRE(n_keep && (n == 0 || (boxes && scores && idxs && keep)), "softnms: null");
  MI_REQUIRE(n >= 0 && n <= SNMS_T * SNMS_PER, "softnms: n %d (at most %d boxes)", n, SNMS_T * SNMS_PER);
  hipStream_t s = (hipStream_t)st;
  if (hipMemsetAsync(n_keep, 0, sizeof(int32_t), s) != hipSuccess) MI_FAIL(MI_ELAUNCH, "softnms: memset");
  if (n == 0) return MI_OK;
  hipLaunchKernelGGL(softnms_kernel, dim3(1), dim3(SNMS_T), 0, s, boxes, scores, idxs, n, sigma, score_threshold, linear);
  hipLaunchKernelGGL(softnms_keep_kernel, dim3(mi_cdiv(n, 256)), dim3(256), 0, s, scores, n, score_threshold, keep, n_keep);
  MI_CHECK_LAUNCH("softnms");
  return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix NMS (SOLOv2; yolov7/utils/solov2_utils.py:160-206) on the mask-intersection matrix inter[n][n] =
// masks @ masks^T (exact in any float format: 0 / 1 masks): score decay of candidate j by every higher-ranked candidate i
// of the same class, compensated by how much i itself was overlapped.  Candidates are in descending score order.
//   iou[i][j] = inter / (sum_i + sum_j - inter) for i < j, else 0;   delay = iou where the labels agree, else 0
//   comp[i] = max_k delay[k][i];   coef[j] = min over ALL i of  f(delay[i][j]) / f(comp[i])
//   f = exp(-sigma x^2) (gaussian) or 1 - x (linear);   score[j] *= coef[j]
__global__ __launch_bounds__(256) void matrix_nms_comp_kernel(const float* __restrict__ inter, const float* __restrict__ sums,
                                                              const float* __restrict__ labels, int n, float* comp) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // column
  if (i >= n) return;
  float m = 0.f;                                   // the masked matrix holds zeros: the maximum is at least 0
  const float li = labels[i], si = sums[i];
  for (int k = 0; k < i; ++k) {
    if (labels[k] != li) continue;
    const float in = inter[(size_t)k * n + i];
    m = fmaxf(m, in / (sums[k] + si - in));
  }
  comp[i] = m;
}
__global__ __launch_bounds__(256) void matrix_nms_decay_kernel(const float* __restrict__ inter, const float* __restrict__ sums,
                                                               const float* __restrict__ labels, const float* __restrict__ comp,
                                                               const float* __restrict__ scores, int n, float sigma, int linear,
                                                               float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float lj = labels[j], sj = sums[j];
  float coef = INFINITY;
  for (int i = 0; i < n; ++i) {
    float d = 0.f;
    if (i < j && labels[i] == lj) {
      const float in = inter[(size_t)i * n + j];
      d = in / (sums[i] + sj - in);
    }
    const float c = comp[i];
    const float r = linear ? (1.f - d) / (1.f - c) : expf(-sigma * d * d) / expf(-sigma * c * c);
    coef = fminf(coef, r);
  }
  out[j] = scores[j] * coef;
}
extern "C" int mi_matrix_nms(const float* inter, const float* sum_masks, const float* labels, const float* scores, int n,
                             float sigma, int linear, float* comp_ws, float* out_scores, mi_stream_t st) {
  MI_REQUIRE(n >= 0, "matrix_nms: n");
  if (n == 0) return MI_OK;
  MI_REQUIRE(inter && sum_masks && labels && scores && comp_ws && out_scores, "matrix_nms: null");
  hipStream_t s = (hipStream_t)st;
  hipLaunchKernelGGL(matrix_nms_comp_kernel, dim3(mi_cdiv(n, 256)), dim3(256), 0, s, inter, sum_masks, labels, n, comp_ws);
  hipLaunchKernelGGL(matrix_nms_decay_kernel, dim3(mi_cdiv(n, 256)), dim3(256), 0, s, inter, sum_masks, labels, comp_ws,
                     scores, n, sigma, linear, out_scores);
  MI_CHECK_LAUNCH("matrix_nms");
  return MI_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Greedy mask NMS (mask_nms, yolov7/utils/solov2_utils.py:209-236) on the mask-intersection matrix: candidates in
// descending score order; a kept candidate i removes every later j of the same label whose mask IoU
// inter / (sum_i + sum_j - inter) exceeds thr - and every later same-label j whose union is 0.  One block: the scan over i
// is sequential (a candidate's fate depends on the kept ones before it), the sweep over j is parallel.
__global__ __launch_bounds__(1024) void mask_nms_kernel(const float* __restrict__ inter, const float* __restrict__ sums,
                                                        const float* __restrict__ labels, int n, float thr,
                                                        uint8_t* __restrict__ keep) {
  extern __shared__ uint8_t s_keep[];
  for (int j = threadIdx.x; j < n; j += 1024) s_keep[j] = 1;
  __syncthreads();
  for (int i = 0; i + 1 < n; ++i) {
    if (s_keep[i]) {   // (uniform: read from LDS after the barrier)
      const float li = labels[i], si = sums[i];
      for (int j = i + 1 + threadIdx.x; j < n; j += 1024) {
        if (!s_keep[j] || labels[j] != li) continue;
        const float in = inter[(size_t)i * n + j];
        const float uni = si + sums[j] - in;
        if (uni > 0.f) {
          if (in / uni > thr) s_keep[j] = 0;
        } else {
          s_keep[j] = 0;
        }
      }
    }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < n; j += 1024) keep[j] = s_keep[j];
}
extern "C" int mi_mask_nms(const float* inter, const float* sum_masks, const float* labels, int n, float nms_thr,
                           uint8_t* keep, mi_stream_t st) {
  MI_REQUIRE(n >= 0 && n <= 65536, "mask_nms: n %d", n);
  if (n == 0) return MI_OK;
  MI_REQUIRE(inter && sum_masks && labels && keep, "mask_nms: null");
  hipLaunchKernelGGL(mask_nms_kernel, dim3(1), dim3(1024), (size_t)n, (hipStream_t)st, inter, sum_masks, labels, n, nms_thr,
                     keep);
  MI_CHECK_LAUNCH("mask_nms");
  return MI_OK;
}
