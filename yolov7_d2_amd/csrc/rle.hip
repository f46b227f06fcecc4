// COCO run-length encoding of binary masks (what `pycocotools.mask.encode` does for instances_to_coco_json,
// yolov7/evaluation/coco_evaluation.py:38-50; algorithm: cocoapi common/maskApi.c rleEncode / rleToString, un-vendored).
// A mask [H][W] is scanned in COLUMN-major order (index = x * H + y); counts[] are the lengths of the alternating runs,
// starting with a (possibly empty) run of zeros.  Byte / integer work, bit-exact by construction.
//
// gfx950: one block per mask.  Each thread owns a contiguous piece of the column-major sequence, counts the positions
// where the value differs from its predecessor (position 0 counts iff the mask starts with a one... see below), a block
// scan turns the counts into output offsets, and the thread writes the run START positions; run lengths are differences
// of consecutive starts, taken in a second sweep over the (short) start list.
#include <string.h>
#include "common.h"

#define RLE_T 256

// value of the mask at column-major position i
__device__ __forceinline__ int rle_at(const uint8_t* __restrict__ m, int H, int W, int64_t i) {
  const int x = (int)(i / H), y = (int)(i - (int64_t)x * H);
  return m[(size_t)y * W + x] != 0;
}

__global__ __launch_bounds__(RLE_T) void rle_encode_kernel(const uint8_t* __restrict__ masks, int H, int W, int max_runs,
                                                           uint32_t* __restrict__ counts, int32_t* __restrict__ nruns) {
  __shared__ int s_cnt[RLE_T];
  __shared__ int s_total;
  const int tid = threadIdx.x;
  const uint8_t* m = masks + (size_t)blockIdx.x * H * W;
  uint32_t* out = counts + (size_t)blockIdx.x * max_runs;
  const int64_t n = (int64_t)H * W;
  const int64_t per = (n + RLE_T - 1) / RLE_T;
  const int64_t i0 = (int64_t)tid * per, i1 = (i0 + per < n) ? i0 + per : n;
  // run starts: position 0 always starts the (possibly empty) leading zero run - it is implicit (start 0); every position
  // i >= 1 whose value differs from position i-1 starts a new run; if the mask begins with a one, a zero-length run of
  // zeros precedes it: an extra start at 0.
  int c = 0;
  if (i0 < n) {
    int prev = i0 > 0 ? rle_at(m, H, W, i0 - 1) : 0;   // virtual predecessor of position 0 is 0
    for (int64_t i = i0; i < i1; ++i) {
      const int v = rle_at(m, H, W, i);
      c += (v != prev);
      prev = v;
    }
  }
  s_cnt[tid] = c;
  __syncthreads();
  if (tid == 0) {   // exclusive scan (256 entries: serial is fine)
    int acc = 0;
    for (int t = 0; t < RLE_T; ++t) { const int v = s_cnt[t]; s_cnt[t] = acc; acc += v; }
    s_total = acc;
  }
  __syncthreads();
  const int total = s_total;            // transitions; runs = total + 1 (the leading zero run has start 0)
  // starts[k + 1] for the k-th transition; starts[0] = 0.  Stored temporarily in `out` as positions.
  if (total + 1 <= max_runs) {
    if (tid == 0) out[0] = 0;
    if (i0 < n) {
      int k = s_cnt[tid];
      int prev = i0 > 0 ? rle_at(m, H, W, i0 - 1) : 0;
      for (int64_t i = i0; i < i1; ++i) {
        const int v = rle_at(m, H, W, i);
        if (v != prev) out[1 + k++] = (uint32_t)i;
        prev = v;
      }
    }
  }
  __syncthreads();
  if (tid == 0) nruns[blockIdx.x] = (total + 1 <= max_runs) ? total + 1 : -(total + 1);
  if (total + 1 > max_runs) return;
  // starts -> lengths, in place: length[r] = start[r+1] - start[r] (last: n - start).  Each thread reads its entries and
  // their successors before anyone writes.
  const int R = total + 1;
  uint32_t len[8];
  int nmine = 0;
  for (int r = tid; r < R && nmine < 8; r += RLE_T) {
    const uint32_t s0 = out[r], s1 = (r + 1 < R) ? out[r + 1] : (uint32_t)n;
    len[nmine++] = s1 - s0;
  }
  const bool fits = R <= 8 * RLE_T;
  __syncthreads();
  if (fits) {
    int q = 0;
    for (int r = tid; r < R; r += RLE_T) out[r] = len[q++];
  } else if (tid == 0) {   // very fragmented mask: serial in-place conversion (front to back is safe)
    for (int r = 0; r < R; ++r) out[r] = ((r + 1 < R) ? out[r + 1] : (uint32_t)n) - out[r];
  }
}

extern "C" int mi_rle_encode(const uint8_t* masks, int n, int H, int W, int max_runs, uint32_t* counts, int32_t* nruns,
                             mi_stream_t st) {
  MI_REQUIRE(n >= 0 && H > 0 && W > 0 && max_runs >= 1 && (int64_t)H * W < (1LL << 31), "rle_encode: sizes");
  if (n == 0) return MI_OK;
  MI_REQUIRE(masks && counts && nruns, "rle_encode: null");
  hipLaunchKernelGGL(rle_encode_kernel, dim3(n), dim3(RLE_T), 0, (hipStream_t)st, masks, H, W, max_runs, counts, nruns);
  MI_CHECK_LAUNCH("rle_encode");
  return MI_OK;
}

// host: counts -> the compact ASCII string of the COCO format (maskApi.c rleToString): each count (from the third on, the
// difference to the count two places before) in 5-bit groups, low group first, bit 5 = "more", + 48.
extern "C" int mi_rle_to_string(const uint32_t* counts, int nruns, char* out, int out_cap) {
  MI_REQUIRE(counts && out && nruns >= 0 && out_cap >= 1, "rle_to_string: args");
  int p = 0;
  for (int i = 0; i < nruns; ++i) {
    long x = (long)counts[i];
    if (i > 2) x -= (long)counts[i - 2];
    bool more = true;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      c += 48;
      if (p + 1 >= out_cap) MI_FAIL(MI_EINVAL, "rle_to_string: output buffer of %d bytes too small", out_cap);
      out[p++] = c;
    }
  }
  out[p] = 0;
  return p;
}
