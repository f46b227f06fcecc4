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
  MI_REQUIRE(n_keep && (n == 0 || (boxes && scores && idxs && order && mask && sboxes && keep)), "nms: null");
  hipStream_t s = (hipStream_t)st;
  if (n == 0) {
    if (hipMemsetAsync(n_keep, 0, sizeof(int32_t), s) != hipSuccess) MI_FAIL(MI_ELAUNCH, "nms: memset");
    return MI_OK;
  }
  MI_REQUIRE(n <= 65536, "nms: n %d too large", n);
  const int nw = mi_cdiv(n, 64);
  // torchvision: coordinate trick when boxes.numel() <= 4000, per-class NMS above
  const int mode = (4 * (int64_t)n > 4000) ? 0 : 1;
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
