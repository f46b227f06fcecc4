// YOLOX head on the GPU with no host round trip: decode, SimOTA assignment, IoU / objectness /
// class losses and their gradient with respect to the raw head outputs.
//
// Follows yolov7/modeling/head/yolox_head.py:
//   get_output_and_grid :226-245   (decode xy/wh)
//   get_losses          :274-441   (targets, loss sums, normalisation by num_fg)
//   get_assignments     :450-547   (cost = cls + 3*iou + 1e5*~(in_box & in_centre))
//   get_in_boxes_info   :549-633   (strict > 0 tests, centre radius 2.5 strides)
//   dynamic_k_matching  :635-669   (k = max(1,int(sum top10 iou)), k smallest cost, conflicts -> argmin cost)
// and yolov7/utils/boxes.py: bboxes_iou :57-81 (no eps), IOUloss :125-168 (eps 1e-16, 1 - iou^2).
// The reference loops over images and ground truths on the host with .item() syncs; here the
// batch is processed by four launches.  Compile with -ffp-contract=off: the float expressions keep
// the reference's operation order so near-ties resolve the same way.
#include <stdlib.h>
#include "common.h"
#include "iou_v6.h"

#define SIMOTA_INF 3.0e38f

struct LossK {
  const float* preds;
  const float* labels;
  const float* anchors;
  int B, A, ncls, max_labels, gmax, nch;
  float* cost;
  float* iou;
  uint8_t* match;
  int32_t* ngt;
  uint8_t* fg;
  int32_t* matched_gt;
  float* matched_iou;
  float* partial;
  float* out;
  int use_l1;          // get_l1_target + nn.L1Loss on the raw regression outputs (yolox_head.py:389-427, 443-448)
  float* partial_l1;   // [nblk] block sums of the L1 term
  // the YOLOv6 head's form of the same loss (ComputeLoss, head/yolov6_head.py:315-754): configurable SimOTA weights /
  // centre radius and an IOUlossV6 box loss; the YOLOX values are 2.5, 1, 3, 5 and box loss 1 - iou^2 (iou_type 0)
  float center_radius, cls_weight, iou_weight, reg_weight;
  int dbg;             // timing experiments (MI_SIMOTA_DBG): early exits of the dynamic-k kernel
  int iou_type;        // 0: IOUloss "iou" of the YOLOX head; 1..4: IOUlossV6 giou / diou / ciou / siou (eps 1e-7)
};

// l1 target of one foreground anchor (yolox_head.py:443-448; eps = 1e-8)
__device__ __forceinline__ void l1_target(const float* lab, float gx_, float gy_, float st, float* t) {
  t[0] = lab[1] / st - gx_;
  t[1] = lab[2] / st - gy_;
  t[2] = logf(lab[3] / st + 1e-8f);
  t[3] = logf(lab[4] / st + 1e-8f);
}

__device__ __forceinline__ float clamp_log(float v) { return fmaxf(logf(v), -100.0f); }
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// number of valid labels: rows with sum > 0 (yolox_head.py:295)
__device__ int count_labels(const float* lab, int max_labels) {
  int n = 0;
  for (int r = 0; r < max_labels; ++r) {
    float s = lab[r * 5 + 0] + lab[r * 5 + 1];
    s = s + lab[r * 5 + 2];
    s = s + lab[r * 5 + 3];
    s = s + lab[r * 5 + 4];
    if (s > 0.f) ++n;
  }
  return n;
}

// ---- kernel 1: candidates, pairwise IoU and cost
// what the cost of one (candidate anchor, gt) pair needs of the anchor
struct CandRec { float xc, yc, st, px, py, pw, ph, so, S; const float* pr; };
// decode (yolox_head.py:243-244) and anchor centre (yolox_head.py:558-570)
__device__ __forceinline__ void simota_cand_decode(const LossK& p, int b, int a, CandRec& r) {
  r.pr = p.preds + ((size_t)b * p.A + a) * p.nch;
  const float gxs = p.anchors[a * 3 + 0], gys = p.anchors[a * 3 + 1], st = p.anchors[a * 3 + 2];
  r.st = st;
  r.xc = gxs * st + 0.5f * st;
  r.yc = gys * st + 0.5f * st;
  r.px = (r.pr[0] + gxs) * st;
  r.py = (r.pr[1] + gys) * st;
  r.pw = expf(r.pr[2]) * st;
  r.ph = expf(r.pr[3]) * st;
  r.so = sigmoid_ref(r.pr[4]);
}
// sum_c max(log(1 - p_c), -100): the row is read in batches of 16 logits issued together (one load per iteration is a
// chain of ncls dependent round trips), summed in class order
__device__ __forceinline__ float simota_class_sum(const LossK& p, const float* pr, float so) {
  float S = 0.f;
  for (int c0 = 0; c0 < p.ncls; c0 += 16) {
    float lg[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) lg[u] = (c0 + u < p.ncls) ? pr[5 + c0 + u] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (c0 + u < p.ncls) {
        const float pc = sqrtf(sigmoid_ref(lg[u]) * so);
        S += clamp_log(1.0f - pc);
      }
    }
  }
  return S;
}
// cost / IoU of one pair, stored at [g][a] (yolox_head.py:487-547)
__device__ __forceinline__ void simota_cost_pair(const LossK& p, const float* slab, int g, const CandRec& r, float* costp, float* ioup) {
  const float xc = r.xc, yc = r.yc, st = r.st, px = r.px, py = r.py, pw = r.pw, ph = r.ph;
  const float area_b = pw * ph;
  const float gcls = slab[g * 5 + 0];
  const float gcx = slab[g * 5 + 1], gcy = slab[g * 5 + 2], gw = slab[g * 5 + 3], gh = slab[g * 5 + 4];
  const float bl = xc - (gcx - 0.5f * gw), br = (gcx + 0.5f * gw) - xc;
  const float bt = yc - (gcy - 0.5f * gh), bb = (gcy + 0.5f * gh) - yc;
  const bool inb = fminf(fminf(bl, bt), fminf(br, bb)) > 0.0f;
  const float cl = xc - (gcx - p.center_radius * st), cr = (gcx + p.center_radius * st) - xc;
  const float ct = yc - (gcy - p.center_radius * st), cb = (gcy + p.center_radius * st) - yc;
  const bool inc = fminf(fminf(cl, ct), fminf(cr, cb)) > 0.0f;
  // bboxes_iou(gt, pred, xyxy=False) boxes.py:66-81
  const float tlx = fmaxf(gcx - gw / 2, px - pw / 2), tly = fmaxf(gcy - gh / 2, py - ph / 2);
  const float brx = fminf(gcx + gw / 2, px + pw / 2), bry = fminf(gcy + gh / 2, py + ph / 2);
  const float area_a = gw * gh;
  const float en = (tlx < brx ? 1.f : 0.f) * (tly < bry ? 1.f : 0.f);
  const float area_i = ((brx - tlx) * (bry - tly)) * en;
  const float iou = area_i / (area_a + area_b - area_i);
  const float iou_loss = -logf(iou + 1e-8f);
  const int gc = (int)gcls;
  const float pg = sqrtf(sigmoid_ref(r.pr[5 + gc]) * r.so);
  const float lp = clamp_log(pg), l1p = clamp_log(1.0f - pg);
  const float cls_loss = -lp - (r.S - l1p);
  float cost = p.cls_weight * cls_loss + p.iou_weight * iou_loss;
  cost = cost + 100000.0f * ((inb && inc) ? 0.f : 1.f);
  *costp = cost;
  *ioup = iou;
}
// one candidate anchor, the round-5 way: everything by its own thread
__device__ __forceinline__ void simota_cost_candidate(const LossK& p, const float* slab, int G, int b, int a) {
  CandRec r;
  simota_cand_decode(p, b, a, r);
  r.S = simota_class_sum(p, r.pr, r.so);
  const size_t rowstride = (size_t)p.A;
  float* costp = p.cost + (size_t)b * p.gmax * rowstride + a;
  float* ioup = p.iou + (size_t)b * p.gmax * rowstride + a;
  for (int g = 0; g < G; ++g) simota_cost_pair(p, slab, g, r, costp + g * rowstride, ioup + g * rowstride);
}

// FORM 0: the round-5 kernel (thread 0 counts the labels; a candidate's thread does all of its work - in waves a third
// of whose lanes hold a candidate with COCO-sized boxes).
// FORM 1: the label count across the threads; the block's candidates gathered into a list, the first ncand threads take
// one each.
// FORM 2: FORM 1, and (a) the class term is formed with the CLASSES across the lanes - 16 lanes take one candidate's row
// (64 contiguous bytes per load instead of 64 lanes in 64 different rows 340 bytes apart), the terms go to LDS and the
// candidate's thread adds them up in class order: the same float sum as the loop over its own row; (b) the (candidate, gt)
// PAIRS are dealt to all 256 threads (an image's blocks hold 30 - 250 candidates and up to gmax ground truths).
// Same expressions per class term / pair: identical outputs.
#define SIMOTA_TERM_FLOATS 10240
template <int FORM, int NT = 256>
__global__ __launch_bounds__(NT) void simota_cost_kernel(const LossK p) {
  constexpr int NW = NT / 64;
  extern __shared__ float slab[];  // [gmax][5] (FORM 2: + [SIMOTA_TERM_FLOATS] class terms)
  __shared__ int s_ngt;
  __shared__ CandRec s_rec[FORM == 2 ? NT : 1];
  __shared__ int s_wn[NW];
  __shared__ unsigned short s_list[NT];
  const int b = blockIdx.y, tid = threadIdx.x;
  const float* lab = p.labels + (size_t)b * p.max_labels * 5;
  if constexpr (FORM == 0) {
    if (tid == 0) s_ngt = count_labels(lab, p.max_labels);
  } else {
    // number of valid labels: rows with sum > 0 (yolox_head.py:295; count_labels across the threads)
    if (tid == 0) s_ngt = 0;
    __syncthreads();
    int n = 0;
    for (int r = tid; r < p.max_labels; r += NT) {
      float sm = lab[r * 5 + 0] + lab[r * 5 + 1];
      sm = sm + lab[r * 5 + 2];
      sm = sm + lab[r * 5 + 3];
      sm = sm + lab[r * 5 + 4];
      n += sm > 0.f ? 1 : 0;
    }
    if (n) atomicAdd(&s_ngt, n);
  }
  for (int i = tid; i < p.gmax * 5; i += NT) slab[i] = lab[i];
  __syncthreads();
  const int G = s_ngt < p.gmax ? s_ngt : p.gmax;
  if (tid == 0 && blockIdx.x == 0) p.ngt[b] = G;
  if (p.dbg == 4) return;
  const int a = blockIdx.x * NT + tid;
  bool cand = false;
  if (a < p.A) {
    // per-anchor match counter / matched gt, filled by the dynamic-k kernels with atomics and consumed by the resolve
    // kernel, which then stores the final values in the same words (the count lives in matched_iou's bits until then)
    p.matched_gt[(size_t)b * p.A + a] = -1;
    ((int32_t*)p.matched_iou)[(size_t)b * p.A + a] = 0;
    const float gxs = p.anchors[a * 3 + 0], gys = p.anchors[a * 3 + 1], st = p.anchors[a * 3 + 2];
    const float xc = gxs * st + 0.5f * st;
    const float yc = gys * st + 0.5f * st;
    for (int g = 0; g < G; ++g) {
      const float gcx = slab[g * 5 + 1], gcy = slab[g * 5 + 2], gw = slab[g * 5 + 3], gh = slab[g * 5 + 4];
      const float bl = xc - (gcx - 0.5f * gw), br = (gcx + 0.5f * gw) - xc;
      const float bt = yc - (gcy - 0.5f * gh), bb = (gcy + 0.5f * gh) - yc;
      const bool inb = fminf(fminf(bl, bt), fminf(br, bb)) > 0.0f;
      const float cl = xc - (gcx - p.center_radius * st), cr = (gcx + p.center_radius * st) - xc;
      const float ct = yc - (gcy - p.center_radius * st), cb = (gcy + p.center_radius * st) - yc;
      const bool inc = fminf(fminf(cl, ct), fminf(cr, cb)) > 0.0f;
      cand = cand || inb || inc;
    }
    if (!cand) {
      const size_t rowstride = (size_t)p.A;
      float* costp = p.cost + (size_t)b * p.gmax * rowstride + a;
      float* ioup = p.iou + (size_t)b * p.gmax * rowstride + a;
      for (int g = 0; g < G; ++g) {
        costp[g * rowstride] = SIMOTA_INF;
        ioup[g * rowstride] = -1.0f;
      }
    }
  }
  if (p.dbg == 5) return;
  if constexpr (FORM == 0) {
    if (cand) simota_cost_candidate(p, slab, G, b, a);
  } else {
    const unsigned long long m = __ballot(cand);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) s_wn[wave] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      if (w < wave) off += s_wn[w];
      total += s_wn[w];
    }
    if (cand) s_list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tid;
    __syncthreads();
    if constexpr (FORM == 1) {
      if (tid < total) simota_cost_candidate(p, slab, G, b, blockIdx.x * NT + s_list[tid]);
    } else {
      float* const s_term = slab + p.gmax * 5;
      const int ncls = p.ncls, ld = ncls + 1, CH = SIMOTA_TERM_FLOATS / ld;   // (host: CH >= 16)
      CandRec rec;
      if (tid < total) {
        simota_cand_decode(p, b, blockIdx.x * NT + s_list[tid], rec);
        s_rec[tid].so = rec.so;
        s_rec[tid].pr = rec.pr;
      }
      __syncthreads();
      const int grp = tid >> 4, gl = tid & 15;
      for (int c0 = 0; c0 < total; c0 += CH) {
        const int n = total - c0 < CH ? total - c0 : CH;
        // four candidates x eight logits per 16-lane group are loaded before the first is used: one load per iteration is a
        // chain of dependent HBM round trips (a group walks ~8 candidates x 5 loads)
        for (int ci0 = grp; ci0 < n; ci0 += NT / 4) {
          for (int cb = 0; cb < ncls; cb += 128) {
            float lg[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int ci = ci0 + (NT / 16) * u;
              const float* pr = s_rec[c0 + (ci < n ? ci : 0)].pr + 5;
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int c = cb + gl + 16 * q;
                lg[u][q] = (ci < n && c < ncls) ? pr[c] : 0.f;
              }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int ci = ci0 + (NT / 16) * u;
              const float so = s_rec[c0 + (ci < n ? ci : 0)].so;
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int c = cb + gl + 16 * q;
                if (ci < n && c < ncls) {
                  const float pc = sqrtf(sigmoid_ref(lg[u][q]) * so);
                  s_term[ci * ld + c] = clamp_log(1.0f - pc);
                }
              }
            }
          }
        }
        __syncthreads();
        if (tid >= c0 && tid < c0 + n) {
          float S = 0.f;
          for (int c = 0; c < ncls; ++c) S += s_term[(tid - c0) * ld + c];
          rec.S = S;
        }
        __syncthreads();
      }
      if (p.dbg == 6) return;
      if (tid < total) s_rec[tid] = rec;
      __syncthreads();
      // pairs: idx = g * total + ci (exact split: (idx + 0.5) / total is at least 0.5 / 256 away from an integer)
      const float rt = 1.0f / (float)total;
      const size_t rowstride = (size_t)p.A;
      float* const cost0 = p.cost + (size_t)b * p.gmax * rowstride + (size_t)blockIdx.x * NT;
      float* const iou0 = p.iou + (size_t)b * p.gmax * rowstride + (size_t)blockIdx.x * NT;
      for (int idx = tid; idx < total * G; idx += NT) {
        const int g = (int)(((float)idx + 0.5f) * rt);
        const int ci = idx - g * total;
        const size_t o = (size_t)g * rowstride + s_list[ci];
        simota_cost_pair(p, slab, g, s_rec[ci], cost0 + o, iou0 + o);
      }
    }
  }
}

// ---- kernel 2: dynamic-k per (image, gt): block-wide ordered selection
struct KV { float v; int i; };
// order: larger v first (DESC) or smaller v first (ASC); ties -> smaller index first
template <bool DESC>
__device__ __forceinline__ bool kv_better(float v, int i, float bv, int bi) {
  if (bi < 0) return true;
  if (DESC) return v > bv || (v == bv && i < bi);
  return v < bv || (v == bv && i < bi);
}
template <bool DESC>
__device__ __forceinline__ bool kv_after(float v, int i, float lv, int li) {
  // strictly after the last selected element in the total order
  if (li < 0) return true;
  if (DESC) return v < lv || (v == lv && i > li);
  return v > lv || (v == lv && i > li);
}
template <bool DESC>
__device__ KV block_select(const float* row, int A, float lv, int li, float invalid_lo, float invalid_hi, KV* sred) {
  float bv = 0.f;
  int bi = -1;
  for (int a = threadIdx.x; a < A; a += 256) {
    const float v = row[a];
    if (v <= invalid_lo || v >= invalid_hi) continue;
    if (!kv_after<DESC>(v, a, lv, li)) continue;
    if (kv_better<DESC>(v, a, bv, bi)) { bv = v; bi = a; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (oi >= 0 && kv_better<DESC>(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sred[wave].v = bv; sred[wave].i = bi; }
  __syncthreads();
  KV best = sred[0];
  for (int w = 1; w < 4; ++w)
    if (sred[w].i >= 0 && kv_better<DESC>(sred[w].v, sred[w].i, best.v, best.i)) best = sred[w];
  return best;
}

// register-resident variant: the row is read from HBM/L2 once (NV values per thread, element a = tid + 256*j),
// the ~20 ordered selections then scan registers.  Same total order as block_select (ties -> smaller index).
template <bool DESC, int NV>
__device__ __forceinline__ KV block_select_reg(const float (&vals)[NV], int A, float lv, int li, float invalid_lo,
                                               float invalid_hi, KV* sred) {
  float bv = 0.f;
  int bi = -1;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int a = threadIdx.x + 256 * j;
    const float v = vals[j];
    if (a >= A || v <= invalid_lo || v >= invalid_hi) continue;
    if (!kv_after<DESC>(v, a, lv, li)) continue;
    if (kv_better<DESC>(v, a, bv, bi)) { bv = v; bi = a; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (oi >= 0 && kv_better<DESC>(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sred[wave].v = bv; sred[wave].i = bi; }
  __syncthreads();
  KV best = sred[0];
  for (int w = 1; w < 4; ++w)
    if (sred[w].i >= 0 && kv_better<DESC>(sred[w].v, sred[w].i, best.v, best.i)) best = sred[w];
  return best;
}

template <int NV>
__global__ __launch_bounds__(256) void simota_dynk_reg_kernel(const LossK p) {
  __shared__ KV sred[4];
  const int g = blockIdx.x, b = blockIdx.y;
  if (g >= p.ngt[b]) return;
  const float* iour = p.iou + ((size_t)b * p.gmax + g) * p.A;
  const float* costr = p.cost + ((size_t)b * p.gmax + g) * p.A;
  int32_t* cntr = (int32_t*)p.matched_iou + (size_t)b * p.A;
  int32_t* gselr = p.matched_gt + (size_t)b * p.A;
  float vi[NV], vc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int a = threadIdx.x + 256 * j;
    vi[j] = a < p.A ? iour[a] : -1.0f;
    vc[j] = a < p.A ? costr[a] : SIMOTA_INF;
  }
  float sum = 0.f, lv = 0.f;
  int li = -1;
  for (int r = 0; r < 10; ++r) {
    const KV s = block_select_reg<true, NV>(vi, p.A, lv, li, -0.5f, INFINITY, sred);
    if (s.i < 0) break;
    sum += s.v;
    lv = s.v;
    li = s.i;
  }
  int k = (int)sum;
  if (k < 1) k = 1;
  lv = 0.f;
  li = -1;
  for (int r = 0; r < k; ++r) {
    const KV s = block_select_reg<false, NV>(vc, p.A, lv, li, -INFINITY, SIMOTA_INF, sred);
    if (s.i < 0) break;
    if (threadIdx.x == 0) {   // matching_matrix[g][s.i] = 1: count the anchor's matches, remember one of its gts
      atomicAdd(cntr + s.i, 1);
      atomicMax(gselr + s.i, g);
    }
    lv = s.v;
    li = s.i;
  }
}

// Pre-filtered, rank-sorted variant of simota_dynk_reg_kernel.  The ~20 block-wide selection rounds (each a scan of NV
// registers per thread, a six-step shuffle chain and two barriers: ~2.4 us per round with one wave per SIMD) become:
//  1. per-thread bests; their 10th best (per-wave rank sort of the 64 lane bests -> 4 x 10 entries -> rank sort of those)
//     bounds the 10th best ELEMENT from below, so every element of the top 10 is at or before it in the total order;
//  2. those elements (typically 10 - 40 per row) go to a list in LDS;
//  3. every listed element finds its RANK (the number of listed elements before it: one loop of broadcast LDS reads, no
//     dependent chain): ranks 0 .. 9 of the IoU list are the top 10 in order - summed in that order -, ranks 0 .. k - 1 of
//     the cost list are the matches (k <= 10: the sum of ten IoUs), each written by its own thread.
// Same total order (ties -> smaller index), same summation order: identical results; a list longer than the block (or
// k > 10) falls back to the block-wide rounds.
#define DYNK_CAP 256
// (bitwise | and &: no short-circuit branches inside the rank loops)
template <bool DESC>
__device__ __forceinline__ bool kv_valid(float v) {
  if (DESC) return !((v <= -0.5f) | (v >= INFINITY));
  return !((v <= -INFINITY) | (v >= SIMOTA_INF));
}
// strictly before (v, i) in the total order
template <bool DESC>
__device__ __forceinline__ bool kv_before(float ov, int oi, float v, int i) {
  if (DESC) return (ov > v) | ((ov == v) & (oi < i));
  return (ov < v) | ((ov == v) & (oi < i));
}
// rank of (v, i) among the valid entries of list[0 .. M)
template <bool DESC>
__device__ __forceinline__ int kv_rank(const KV* list, int M, float v, int i) {
  int r = 0;
#pragma unroll 8
  for (int m = 0; m < M; ++m) {   // (unrolled: the broadcast reads of eight entries are in flight together)
    const KV e = list[m];
    r += (int)((e.i >= 0) & kv_before<DESC>(e.v, e.i, v, i));
  }
  return r;
}

template <int NV>
__global__ __launch_bounds__(256) void simota_dynk_pre_kernel(const LossK p) {
  __shared__ KV s_best[2][256];
  __shared__ KV s_wtop[2][40];
  __shared__ KV s_list[2][DYNK_CAP];
  __shared__ KV s_thr[2];
  __shared__ int s_cnt[2];
  __shared__ float s_top[10];
  __shared__ KV sred[4];
  const int g = blockIdx.x, b = blockIdx.y;
  if (g >= p.ngt[b]) return;
  const float* iour = p.iou + ((size_t)b * p.gmax + g) * p.A;
  const float* costr = p.cost + ((size_t)b * p.gmax + g) * p.A;
  int32_t* cntr = (int32_t*)p.matched_iou + (size_t)b * p.A;
  int32_t* gselr = p.matched_gt + (size_t)b * p.A;
  const int tid = threadIdx.x, wave = tid >> 6;
  float vi[NV], vc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int a = tid + 256 * j;
    vi[j] = a < p.A ? iour[a] : -1.0f;
    vc[j] = a < p.A ? costr[a] : SIMOTA_INF;
  }
  // ---- per-thread bests
  float bvi = 0.f, bvc = 0.f;
  int bii = -1, bic = -1;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int a = tid + 256 * j;
    const bool ti = (a < p.A) & kv_valid<true>(vi[j]) & ((bii < 0) | kv_before<true>(vi[j], a, bvi, bii));
    const bool tc = (a < p.A) & kv_valid<false>(vc[j]) & ((bic < 0) | kv_before<false>(vc[j], a, bvc, bic));
    bvi = ti ? vi[j] : bvi; bii = ti ? a : bii;
    bvc = tc ? vc[j] : bvc; bic = tc ? a : bic;
  }
  s_best[0][tid].v = bvi; s_best[0][tid].i = bii;
  s_best[1][tid].v = bvc; s_best[1][tid].i = bic;
  if (tid < 80) s_wtop[tid / 40][tid % 40].i = -1;
  if (tid < 2) { s_cnt[tid] = 0; s_thr[tid].i = -1; }
  __syncthreads();
  if (p.dbg == 1) return;
  // ---- the bounds: the 10th largest of the IoU bests, the 10th smallest of the cost bests (i < 0: fewer than ten threads hold
  // a valid element - then all valid elements are listed)
  {
    const int ri = bii >= 0 ? kv_rank<true>(s_best[0] + wave * 64, 64, bvi, bii) : 64;
    const int rc = bic >= 0 ? kv_rank<false>(s_best[1] + wave * 64, 64, bvc, bic) : 64;
    if (ri < 10) { s_wtop[0][wave * 10 + ri].v = bvi; s_wtop[0][wave * 10 + ri].i = bii; }
    if (rc < 10) { s_wtop[1][wave * 10 + rc].v = bvc; s_wtop[1][wave * 10 + rc].i = bic; }
  }
  __syncthreads();
  if (wave < 2 && (tid & 63) < 40) {
    const KV e = s_wtop[wave][tid & 63];
    if (e.i >= 0) {
      const int r = wave == 0 ? kv_rank<true>(s_wtop[0], 40, e.v, e.i) : kv_rank<false>(s_wtop[1], 40, e.v, e.i);
      if (r == 9) s_thr[wave] = e;
    }
  }
  __syncthreads();
  if (p.dbg == 2) return;
  {
    const KV ti = s_thr[0], tc = s_thr[1];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int a = tid + 256 * j;
      if (a >= p.A) continue;
      if (kv_valid<true>(vi[j]) && (ti.i < 0 || !kv_after<true>(vi[j], a, ti.v, ti.i))) {
        const int pos = atomicAdd(&s_cnt[0], 1);
        if (pos < DYNK_CAP) { s_list[0][pos].v = vi[j]; s_list[0][pos].i = a; }
      }
      if (kv_valid<false>(vc[j]) && (tc.i < 0 || !kv_after<false>(vc[j], a, tc.v, tc.i))) {
        const int pos = atomicAdd(&s_cnt[1], 1);
        if (pos < DYNK_CAP) { s_list[1][pos].v = vc[j]; s_list[1][pos].i = a; }
      }
    }
  }
  __syncthreads();
  const int Mi = s_cnt[0], Mc = s_cnt[1];
  if (p.dbg == 3) return;
  if (Mi <= DYNK_CAP && Mc <= DYNK_CAP) {
    KV ec;
    ec.v = 0.f;
    ec.i = -1;
    int rc = DYNK_CAP;
    if (tid < Mi) {
      const KV e = s_list[0][tid];
      const int r = kv_rank<true>(s_list[0], Mi, e.v, e.i);
      if (r < 10) s_top[r] = e.v;
    }
    // (the cost list's ranks by the upper half of the block when it fits: the two rank loops then run on different SIMDs)
    const int tc_ = Mc <= 128 ? tid - 128 : tid;
    if (tc_ >= 0 && tc_ < Mc) {
      ec = s_list[1][tc_];
      rc = kv_rank<false>(s_list[1], Mc, ec.v, ec.i);
    }
    __syncthreads();
    // top-10 IoU among candidates, summed in descending order (yolox_head.py:640-642)
    float sum = 0.f;
    const int ntop = Mi < 10 ? Mi : 10;
    for (int r = 0; r < ntop; ++r) sum += s_top[r];
    int k = (int)sum;
    if (k < 1) k = 1;
    if (k <= 10) {
      if (rc < k) {   // matching_matrix[g][a] = 1: count the anchor's matches, remember one of its gts
        atomicAdd(cntr + ec.i, 1);
        atomicMax(gselr + ec.i, g);
      }
      return;
    }
    // (k > 10 cannot come out of ten IoUs <= 1; kept exact anyway)
    float lv = 0.f;
    int li = -1;
    for (int r = 0; r < k; ++r) {
      const KV s = block_select_reg<false, NV>(vc, p.A, lv, li, -INFINITY, SIMOTA_INF, sred);
      if (s.i < 0) break;
      if (tid == 0) {
        atomicAdd(cntr + s.i, 1);
        atomicMax(gselr + s.i, g);
      }
      lv = s.v;
      li = s.i;
    }
    return;
  }
  // ---- a list overflowed: the block-wide rounds
  float sum = 0.f, lv = 0.f;
  int li = -1;
  for (int r = 0; r < 10; ++r) {
    const KV s = block_select_reg<true, NV>(vi, p.A, lv, li, -0.5f, INFINITY, sred);
    if (s.i < 0) break;
    sum += s.v;
    lv = s.v;
    li = s.i;
  }
  int k = (int)sum;
  if (k < 1) k = 1;
  lv = 0.f;
  li = -1;
  for (int r = 0; r < k; ++r) {
    const KV s = block_select_reg<false, NV>(vc, p.A, lv, li, -INFINITY, SIMOTA_INF, sred);
    if (s.i < 0) break;
    if (tid == 0) {
      atomicAdd(cntr + s.i, 1);
      atomicMax(gselr + s.i, g);
    }
    lv = s.v;
    li = s.i;
  }
}

__global__ __launch_bounds__(256) void simota_dynk_kernel(const LossK p) {
  __shared__ KV sred[4];
  const int g = blockIdx.x, b = blockIdx.y;
  if (g >= p.ngt[b]) return;
  const float* iour = p.iou + ((size_t)b * p.gmax + g) * p.A;
  const float* costr = p.cost + ((size_t)b * p.gmax + g) * p.A;
  int32_t* cntr = (int32_t*)p.matched_iou + (size_t)b * p.A;
  int32_t* gselr = p.matched_gt + (size_t)b * p.A;
  // top-10 IoU among candidates (iou >= 0), summed in descending order
  float sum = 0.f, lv = 0.f;
  int li = -1;
  for (int r = 0; r < 10; ++r) {
    const KV s = block_select<true>(iour, p.A, lv, li, -0.5f, INFINITY, sred);
    if (s.i < 0) break;
    sum += s.v;
    lv = s.v;
    li = s.i;
  }
  int k = (int)sum;
  if (k < 1) k = 1;
  lv = 0.f;
  li = -1;
  for (int r = 0; r < k; ++r) {
    const KV s = block_select<false>(costr, p.A, lv, li, -INFINITY, SIMOTA_INF, sred);
    if (s.i < 0) break;
    if (threadIdx.x == 0) {   // matching_matrix[g][s.i] = 1: count the anchor's matches, remember one of its gts
      atomicAdd(cntr + s.i, 1);
      atomicMax(gselr + s.i, g);
    }
    lv = s.v;
    li = s.i;
  }
}

// shared decode of one prediction row
struct Box { float x, y, w, h; };
__device__ __forceinline__ Box decode_box(const float* pr, float gxs, float gys, float st) {
  Box b;
  b.x = (pr[0] + gxs) * st;
  b.y = (pr[1] + gys) * st;
  b.w = expf(pr[2]) * st;
  b.h = expf(pr[3]) * st;
  return b;
}
__device__ __forceinline__ float bce_logits(float x, float t) {
  // F.binary_cross_entropy_with_logits: (1-t)*x + max(-x,0) + log1p(exp(-|x|))
  return (1.f - t) * x + fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));
}

// ---- kernel 3: resolve conflicts, write assignment, accumulate loss sums
__global__ __launch_bounds__(256) void simota_resolve_loss_kernel(const LossK p) {
  __shared__ float sred[4][5];
  const int b = blockIdx.y;
  const int a = blockIdx.x * 256 + threadIdx.x;
  const int G = p.ngt[b];
  float l_iou = 0.f, l_obj = 0.f, l_cls = 0.f, nfg = 0.f, l_l1 = 0.f;
  float miou_ = 0.f;
  int gcls = -1;
  bool fg_ = false;
  if (a < p.A) {
    const size_t rs = (size_t)p.A;
    const float* costp = p.cost + (size_t)b * p.gmax * rs + a;
    const int cnt = ((const int32_t*)p.matched_iou)[(size_t)b * p.A + a];   // anchor_matching_gt = matching_matrix.sum(0)
    int gsel = p.matched_gt[(size_t)b * p.A + a];                          // its gt when it has exactly one
    if (cnt > 1) {  // yolox_head.py:653-657: argmin of cost over ALL gts, first minimum
      float bv = costp[0];
      gsel = 0;
      for (int g = 1; g < G; ++g) {
        const float v = costp[g * rs];
        if (v < bv) { bv = v; gsel = g; }
      }
    }
    const bool fg = cnt > 0;
    const size_t o = (size_t)b * p.A + a;
    p.fg[o] = fg ? 1 : 0;
    p.matched_gt[o] = fg ? gsel : -1;
    const float miou = fg ? p.iou[((size_t)b * p.gmax + gsel) * rs + a] : 0.f;
    p.matched_iou[o] = miou;
    fg_ = fg;
    miou_ = miou;
    const float* pr = p.preds + o * p.nch;
    l_obj = bce_logits(pr[4], fg ? 1.f : 0.f);
    if (fg) {
      nfg = 1.f;
      const float* lab = p.labels + ((size_t)b * p.max_labels + gsel) * 5;
      const Box pb = decode_box(pr, p.anchors[a * 3 + 0], p.anchors[a * 3 + 1], p.anchors[a * 3 + 2]);
      const float gx = lab[1], gy = lab[2], gw = lab[3], gh = lab[4];
      // IOUloss boxes.py:131-151
      const float tlx = fmaxf(pb.x - pb.w / 2, gx - gw / 2), tly = fmaxf(pb.y - pb.h / 2, gy - gh / 2);
      const float brx = fminf(pb.x + pb.w / 2, gx + gw / 2), bry = fminf(pb.y + pb.h / 2, gy + gh / 2);
      const float area_p = pb.w * pb.h, area_g = gw * gh;
      const float en = (tlx < brx ? 1.f : 0.f) * (tly < bry ? 1.f : 0.f);
      const float area_i = ((brx - tlx) * (bry - tly)) * en;
      const float iou = area_i / (area_p + area_g - area_i + 1e-16f);
      l_iou = 1.f - iou * iou;
      if (p.iou_type) {   // IOUlossV6 on the decoded box (yolov6_head.py:346,512)
        const float pbox[4] = {pb.x, pb.y, pb.w, pb.h};
        l_iou = 1.f - iou_v6_dual(pbox, lab + 1, p.iou_type, 0, 1e-7f).v;
      }
      gcls = (int)lab[0];
      if (p.use_l1) {
        float t[4];
        l1_target(lab, p.anchors[a * 3 + 0], p.anchors[a * 3 + 1], p.anchors[a * 3 + 2], t);
        l_l1 = fabsf(pr[0] - t[0]) + fabsf(pr[1] - t[1]) + fabsf(pr[2] - t[2]) + fabsf(pr[3] - t[3]);
      }
    }
  }
  // class BCE of the foreground anchors (1-2 % of the anchors, but 60 % of the waves hold one): the wave takes its
  // foreground rows one at a time with the CLASSES across the lanes (coalesced row read, two trips for 80 classes) instead
  // of one lane walking 80 classes while 63 wait
  {
    const int lane = threadIdx.x & 63;
    unsigned long long m = __ballot(fg_);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int as = __shfl(a, src, 64), gcs = __shfl(gcls, src, 64);
      const float mi = __shfl(miou_, src, 64);
      const float* prs = p.preds + ((size_t)b * p.A + as) * p.nch + 5;
      for (int c = lane; c < p.ncls; c += 64) l_cls += bce_logits(prs[c], c == gcs ? mi : 0.f);
    }
  }
  l_iou = wave_sum(l_iou); l_obj = wave_sum(l_obj); l_cls = wave_sum(l_cls); nfg = wave_sum(nfg);
  if (p.use_l1) l_l1 = wave_sum(l_l1);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sred[wave][0] = l_iou; sred[wave][1] = l_obj; sred[wave][2] = l_cls; sred[wave][3] = nfg; sred[wave][4] = l_l1;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float v = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x];
    p.partial[((size_t)b * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = v;
  }
  if (threadIdx.x == 4 && p.use_l1)
    p.partial_l1[(size_t)b * gridDim.x + blockIdx.x] = sred[0][4] + sred[1][4] + sred[2][4] + sred[3][4];
}

__global__ __launch_bounds__(64) void loss_final_kernel(const float* partial, const float* partial_l1, int nblk,
                                                        const int32_t* ngt, int B, float reg_weight, float* out) {
  const int lane = threadIdx.x;
  double s[4] = {0, 0, 0, 0};
  double s_l1 = 0;
  // (four blocks' partials in flight per trip - 16-byte loads -, added in the same order: one dependent L2 round trip per
  //  block partial was most of this launch's 4.8 us)
  int t = lane;
  for (; t + 192 < nblk; t += 256) {
    float4 v[4];
    float l1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = *(const float4*)(partial + (size_t)(t + 64 * u) * 4);
      if (partial_l1) l1[u] = partial_l1[t + 64 * u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s[0] += (double)v[u].x; s[1] += (double)v[u].y; s[2] += (double)v[u].z; s[3] += (double)v[u].w;
      if (partial_l1) s_l1 += (double)l1[u];
    }
  }
  for (; t < nblk; t += 64) {
    for (int q = 0; q < 4; ++q) s[q] += (double)partial[(size_t)t * 4 + q];
    if (partial_l1) s_l1 += (double)partial_l1[t];
  }
  double g = 0;
  for (int b = lane; b < B; b += 64) g += (double)ngt[b];
  for (int q = 0; q < 4; ++q) s[q] = wave_sum_d(s[q]);
  s_l1 = wave_sum_d(s_l1);
  g = wave_sum_d(g);
  if (lane == 0) {
    const double nfg = s[3];
    const double N = nfg > 1.0 ? nfg : 1.0;
    const float li = (float)(s[0] / N), lo = (float)(s[1] / N), lc = (float)(s[2] / N);
    out[1] = reg_weight * li;
    out[2] = lo;
    out[3] = lc;
    const float l1 = (float)(s_l1 / N);   // 0 unless use_l1
    out[0] = reg_weight * li + lo + lc + l1;
    out[4] = l1;
    out[5] = (float)(N / (g > 1.0 ? g : 1.0));
    out[6] = (float)nfg;
    out[7] = (float)g;
  }
}

static int loss_fill(const mi_yolox_loss_desc* d, LossK* k) {
  MI_REQUIRE(d->preds && d->labels && d->anchors && d->cost && d->iou && d->ngt && d->fg &&
                 d->matched_gt && d->matched_iou && d->partial && d->out,
             "yolox_loss: null pointer");
  MI_REQUIRE(d->B > 0 && d->A > 0 && d->ncls > 0 && d->gmax > 0 && d->gmax <= d->max_labels, "yolox_loss: sizes");
  k->preds = d->preds; k->labels = d->labels; k->anchors = d->anchors;
  k->B = d->B; k->A = d->A; k->ncls = d->ncls; k->max_labels = d->max_labels; k->gmax = d->gmax;
  k->nch = 5 + d->ncls;
  k->cost = d->cost; k->iou = d->iou; k->match = d->match; k->ngt = d->ngt; k->fg = d->fg;
  k->matched_gt = d->matched_gt; k->matched_iou = d->matched_iou; k->partial = d->partial; k->out = d->out;
  k->use_l1 = d->use_l1 != 0; k->partial_l1 = d->partial_l1;
  k->center_radius = d->center_radius > 0.f ? d->center_radius : 2.5f;
  k->cls_weight = d->cls_weight > 0.f ? d->cls_weight : 1.0f;
  k->iou_weight = d->iou_weight > 0.f ? d->iou_weight : 3.0f;
  k->reg_weight = d->reg_weight > 0.f ? d->reg_weight : 5.0f;
  k->iou_type = d->iou_type;
  k->dbg = 0;
  MI_REQUIRE(d->iou_type >= 0 && d->iou_type <= 4, "yolox_loss: iou_type %d", d->iou_type);
  MI_REQUIRE(!k->use_l1 || d->partial_l1, "yolox_loss: use_l1 needs partial_l1");
  return MI_OK;
}

extern "C" int mi_yolox_loss_fwd(const mi_yolox_loss_desc* d, mi_stream_t st) {
  LossK k;
  int rc = loss_fill(d, &k);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)st;
  const int nb = mi_cdiv(d->A, 256);
  // MI_SIMOTA_COMPACT=0 (1: compaction only) / MI_SIMOTA_PREFILTER=0: the round-5 forms of the two kernels (read per call: the tests compare
  // both forms in one process; identical outputs)
  const char* e1 = getenv("MI_SIMOTA_COMPACT");
  const char* e2 = getenv("MI_SIMOTA_PREFILTER");
  int form = e1 ? atoi(e1) : 2;
  const bool pre = !e2 || atoi(e2) != 0;
  if (form == 2 && SIMOTA_TERM_FLOATS / (d->ncls + 1) < 16) form = 1;   // (very many classes: the terms of 16 candidates do not fit)
  const char* e3 = getenv("MI_SIMOTA_DBG");
  k.dbg = e3 ? atoi(e3) : 0;
  const size_t slab_bytes = d->gmax * 5 * sizeof(float);
  // (128-anchor blocks - simota_cost_kernel<2, 128> - measured slower: 91.8 vs 83.3 us for the four launches)
  if (form == 2)
    hipLaunchKernelGGL(simota_cost_kernel<2>, dim3(nb, d->B), dim3(256), slab_bytes + SIMOTA_TERM_FLOATS * sizeof(float), s, k);
  else if (form == 1)
    hipLaunchKernelGGL(simota_cost_kernel<1>, dim3(nb, d->B), dim3(256), slab_bytes, s, k);
  else
    hipLaunchKernelGGL(simota_cost_kernel<0>, dim3(nb, d->B), dim3(256), slab_bytes, s, k);
  MI_CHECK_LAUNCH("simota_cost");
  if (d->A <= 256 * 9) {
    if (pre) hipLaunchKernelGGL(simota_dynk_pre_kernel<9>, dim3(d->gmax, d->B), dim3(256), 0, s, k);
    else hipLaunchKernelGGL(simota_dynk_reg_kernel<9>, dim3(d->gmax, d->B), dim3(256), 0, s, k);
  } else if (d->A <= 256 * 33) {
    if (pre) hipLaunchKernelGGL(simota_dynk_pre_kernel<33>, dim3(d->gmax, d->B), dim3(256), 0, s, k);
    else hipLaunchKernelGGL(simota_dynk_reg_kernel<33>, dim3(d->gmax, d->B), dim3(256), 0, s, k);
  } else {
    hipLaunchKernelGGL(simota_dynk_kernel, dim3(d->gmax, d->B), dim3(256), 0, s, k);
  }
  MI_CHECK_LAUNCH("simota_dynk");
  hipLaunchKernelGGL(simota_resolve_loss_kernel, dim3(nb, d->B), dim3(256), 0, s, k);
  MI_CHECK_LAUNCH("simota_resolve_loss");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, d->partial, k.use_l1 ? d->partial_l1 : nullptr,
                     nb * d->B, d->ngt, d->B, k.reg_weight, d->out);
  MI_CHECK_LAUNCH("loss_final");
  return MI_OK;
}

// ---- backward: gradient with respect to the raw head outputs, two launches:
//   (1) an elementwise pass over [B][A][5+ncls] with the channel fastest (coalesced 4-byte stores; 99 % of the rows are
//       background: zero except the objectness column, and their logits are not even read) - objectness and class terms;
//   (2) the box columns (IoU / IOUlossV6 and L1 terms) of the foreground anchors only.
// (One thread per anchor writing its own 85-float row touched 64 cache lines per store instruction: 54 us for 45 MB.)
__global__ __launch_bounds__(256) void yolox_loss_bwd_cls_kernel(const LossK p, const float* gw, float* dpreds) {
  const float nfg = p.out[6];
  const float N = nfg > 1.f ? nfg : 1.f;
  const float w_obj = (gw[0] + gw[2]) / N;
  const float w_cls = (gw[0] + gw[3]) / N;
  // (32-bit indices: the launcher checks B * A * nch < 2^31; a 64-bit division per element costs more than the store)
  const unsigned total = (unsigned)p.B * (unsigned)p.A * (unsigned)p.nch, nchu = (unsigned)p.nch;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned o = idx / nchu;
    const int c = (int)(idx - o * nchu);
    const bool fg = p.fg[o] != 0;
    float v = 0.f;
    if (c == 4) {
      v = w_obj * (sigmoid_ref(p.preds[idx]) - (fg ? 1.f : 0.f));
    } else if (c > 4 && fg) {
      const int b = (int)(o / p.A);
      const float* lab = p.labels + ((size_t)b * p.max_labels + p.matched_gt[o]) * 5;
      v = w_cls * (sigmoid_ref(p.preds[idx]) - ((c - 5) == (int)lab[0] ? p.matched_iou[o] : 0.f));
    }
    dpreds[idx] = v;
  }
}
// box-column gradient of one foreground anchor: dp[0..3] = d loss / d raw (x, y, w, h)
__device__ __forceinline__ void box_grad(const LossK& p, const float* gw, const size_t o, const int a, const int b, float* dp) {
  const float nfg = p.out[6];
  const float N = nfg > 1.f ? nfg : 1.f;
  const float w_iou = p.reg_weight * (gw[0] + gw[1]) / N;
  const float w_l1 = p.use_l1 ? (gw[0] + gw[4]) / N : 0.f;   // gw has a fifth entry (upstream of l1_loss) iff use_l1
  const float* pr = p.preds + o * p.nch;
  const float* lab = p.labels + ((size_t)b * p.max_labels + p.matched_gt[o]) * 5;
  const float st = p.anchors[a * 3 + 2];
  const Box pb = decode_box(pr, p.anchors[a * 3 + 0], p.anchors[a * 3 + 1], st);
  const float gx = lab[1], gy = lab[2], gw_ = lab[3], gh = lab[4];
  const float ptlx = pb.x - pb.w / 2, ptly = pb.y - pb.h / 2, pbrx = pb.x + pb.w / 2, pbry = pb.y + pb.h / 2;
  const float gtlx = gx - gw_ / 2, gtly = gy - gh / 2, gbrx = gx + gw_ / 2, gbry = gy + gh / 2;
  const float tlx = fmaxf(ptlx, gtlx), tly = fmaxf(ptly, gtly), brx = fminf(pbrx, gbrx), bry = fminf(pbry, gbry);
  const float en = (tlx < brx ? 1.f : 0.f) * (tly < bry ? 1.f : 0.f);
  const float wi = brx - tlx, hi = bry - tly;
  const float I = wi * hi * en;
  const float D = pb.w * pb.h + gw_ * gh - I + 1e-16f;
  const float u = I / D;
  const float dLdu = -2.f * u;
  const float dLdI = dLdu * (D + I) / (D * D);
  const float dLdAp = dLdu * (-I / (D * D));
  // max/min sub-gradients (ties split evenly, as ATen's maximum/minimum backward does)
  const float tlx_p = ptlx > gtlx ? 1.f : (ptlx == gtlx ? 0.5f : 0.f);
  const float tly_p = ptly > gtly ? 1.f : (ptly == gtly ? 0.5f : 0.f);
  const float brx_p = pbrx < gbrx ? 1.f : (pbrx == gbrx ? 0.5f : 0.f);
  const float bry_p = pbry < gbry ? 1.f : (pbry == gbry ? 0.5f : 0.f);
  const float dI_dtlx = -hi * en, dI_dbrx = hi * en, dI_dtly = -wi * en, dI_dbry = wi * en;
  const float dpx = dLdI * (dI_dtlx * tlx_p + dI_dbrx * brx_p);
  const float dpy = dLdI * (dI_dtly * tly_p + dI_dbry * bry_p);
  const float dpw = dLdI * (dI_dtlx * (-0.5f) * tlx_p + dI_dbrx * 0.5f * brx_p) + dLdAp * pb.h;
  const float dph = dLdI * (dI_dtly * (-0.5f) * tly_p + dI_dbry * 0.5f * bry_p) + dLdAp * pb.w;
  dp[0] = w_iou * dpx * st;
  dp[1] = w_iou * dpy * st;
  dp[2] = w_iou * dpw * pb.w;
  dp[3] = w_iou * dph * pb.h;
  if (p.iou_type) {   // d(1 - iou_variant)/d(decoded box), chained through the decode (x, y: * stride; w, h: * themselves)
    const float pbox[4] = {pb.x, pb.y, pb.w, pb.h};
    const D4 v = iou_v6_dual(pbox, lab + 1, p.iou_type, 0, 1e-7f);
    dp[0] = w_iou * -v.d[0] * st;
    dp[1] = w_iou * -v.d[1] * st;
    dp[2] = w_iou * -v.d[2] * pb.w;
    dp[3] = w_iou * -v.d[3] * pb.h;
  }
  if (p.use_l1) {   // d|x - t| = sign(x - t) (0 at x == t, as ATen's l1 backward)
    float t[4];
    l1_target(lab, p.anchors[a * 3 + 0], p.anchors[a * 3 + 1], st, t);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float e = pr[q] - t[q];
      dp[q] += w_l1 * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
    }
  }
}
__global__ __launch_bounds__(256) void yolox_loss_bwd_box_kernel(const LossK p, const float* gw, float* dpreds) {
  const int b = blockIdx.y;
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= p.A) return;
  const size_t o = (size_t)b * p.A + a;
  if (!p.fg[o]) return;
  float dp[4];
  box_grad(p, gw, o, a, b, dp);
  float* d = dpreds + o * p.nch;
  d[0] = dp[0]; d[1] = dp[1]; d[2] = dp[2]; d[3] = dp[3];
}

// ---- the same gradient written ONCE in every form the backward pass needs (MI_LOSS_BWD_FUSED, default on): the fp32
// [B][A][5 + ncls] tensor, the bf16 NHWC out-gradient map of every prediction conv (what mi_yolox_split_dpreds_batch
// produced from a second read of that tensor) and per-block column sums for the convs' bias gradients (what
// bias_grads_stage1 produced from a third read).  One thread item = (prediction conv, image, pixel, 8-channel group).
#define MI_LBF_MAX_JOBS 16
struct LossBwdFusedK {
  LossK l;
  const float* gw;
  float* dpreds;
  float* ws;        // [njobs][gridDim.x][ldmax]
  int njobs, ldmax;
  mi_split_job jobs[MI_LBF_MAX_JOBS];
  float* bias_out[MI_LBF_MAX_JOBS];
};
__global__ __launch_bounds__(256) void yolox_loss_bwd_fused_kernel(const LossBwdFusedK q) {
  __shared__ float red[256 * 8];
  const LossK& p = q.l;
  const mi_split_job j = q.jobs[blockIdx.y];
  const int ld8 = j.ld >> 3;
  const int TPB = (256 / ld8) * ld8;          // active threads: a thread keeps its channel group across its items
  const bool active = (int)threadIdx.x < TPB;
  const int j8 = active ? (int)threadIdx.x % ld8 : 0;
  const float nfg = p.out[6];
  const float N = nfg > 1.f ? nfg : 1.f;
  const float w_obj = (q.gw[0] + q.gw[2]) / N;
  const float w_cls = (q.gw[0] + q.gw[3]) / N;
  const unsigned total = (unsigned)p.B * (unsigned)j.HW * (unsigned)ld8;
  uint4* dst = (uint4*)j.dst;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  for (unsigned idx = blockIdx.x * (unsigned)TPB + threadIdx.x; active && idx < total; idx += gridDim.x * (unsigned)TPB) {
    const unsigned bp = idx / (unsigned)ld8;
    const int b = (int)(bp / (unsigned)j.HW), pidx = (int)(bp - (unsigned)b * j.HW);
    const int a = j.a0 + pidx;
    const size_t o = (size_t)b * p.A + a;
    const bool fg = p.fg[o] != 0;
    const int cb = j.c0 + j8 * 8;             // first prediction channel of this item
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (cb < 4) {                              // the regression conv: box columns, foreground only
      if (fg) {
        float dp[4];
        box_grad(p, q.gw, o, a, b, dp);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cb + e < 4 && j8 * 8 + e < j.nc) v[e] = dp[cb + e];
      }
      // reg_preds + obj_preds as one 5-channel convolution (columns 0 .. 4 in one job): the objectness column rides along
      if (cb == 0 && j.nc == 5) v[4] = w_obj * (sigmoid_ref(p.preds[o * p.nch + 4]) - (fg ? 1.f : 0.f));
    } else if (cb == 4 && j.nc == 1) {         // the objectness conv
      v[0] = w_obj * (sigmoid_ref(p.preds[o * p.nch + 4]) - (fg ? 1.f : 0.f));
    } else if (fg) {                           // the class conv (background rows: zero, their logits are not read)
      const float* lab = p.labels + ((size_t)b * p.max_labels + p.matched_gt[o]) * 5;
      const int gc = (int)lab[0];
      const float miou = p.matched_iou[o];
      const float* pr = p.preds + o * p.nch + cb;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (j8 * 8 + e < j.nc) v[e] = w_cls * (sigmoid_ref(pr[e]) - ((cb + e - 5) == gc ? miou : 0.f));
    }
    unsigned short h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] += v[e];
      h[e] = __builtin_bit_cast(unsigned short, (__bf16)v[e]);
    }
    if (q.dpreds) {   // (diagnostic: 4-byte stores 32 bytes apart across the lanes - this alone doubles the kernel's time)
      float* dpr = q.dpreds + o * p.nch + cb;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (j8 * 8 + e < j.nc) dpr[e] = v[e];
    }
    dst[idx] = make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16),
                          h[6] | ((uint32_t)h[7] << 16));
  }
  if (!q.bias_out[blockIdx.y]) return;         // (uniform per block)
  // column sums of this block: threads with the same j8 are tid = j8 + k * ld8
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
  __syncthreads();
  for (int c = threadIdx.x; c < j.ld; c += 256) {
    const int g = c >> 3, e = c & 7;
    float t = 0.f;
    for (int k = g; k < TPB; k += ld8) t += red[k * 8 + e];
    q.ws[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * q.ldmax + c] = t;
  }
}
// bias gradient of one prediction conv = sum of its blocks' column sums: 128 channels x 8 parts of the block list per trip,
// four loads in flight per thread (a rolled loop over 512 partials is a chain of 512 L2 round trips: ~100 us)
__global__ __launch_bounds__(1024) void yolox_loss_bwd_bias_kernel(const LossBwdFusedK q, int nblk) {
  __shared__ float red[1024];
  const mi_split_job j = q.jobs[blockIdx.x];
  float* out = q.bias_out[blockIdx.x];
  if (!out) return;
  const float* w = q.ws + (size_t)blockIdx.x * nblk * q.ldmax;
  const int lane = threadIdx.x & 127, part = threadIdx.x >> 7;
  for (int c0 = 0; c0 < j.nc; c0 += 128) {
    const int c = c0 + lane;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (c < j.nc) {
      int b = part;
      // (16 loads in flight, then the additions in the order of the 4-load loop below: one L2 round trip per 4 loads was
      //  ~6 of this launch's 7.6 us)
      for (; b + 120 < nblk; b += 128) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = w[(size_t)(b + 8 * u) * q.ldmax + c];
#pragma unroll
        for (int u = 0; u < 16; u += 4) { t0 += v[u]; t1 += v[u + 1]; t2 += v[u + 2]; t3 += v[u + 3]; }
      }
      for (; b + 24 < nblk; b += 32) {
        const float v0 = w[(size_t)b * q.ldmax + c], v1 = w[(size_t)(b + 8) * q.ldmax + c];
        const float v2 = w[(size_t)(b + 16) * q.ldmax + c], v3 = w[(size_t)(b + 24) * q.ldmax + c];
        t0 += v0; t1 += v1; t2 += v2; t3 += v3;
      }
      for (; b < nblk; b += 8) t0 += w[(size_t)b * q.ldmax + c];
    }
    red[threadIdx.x] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (part == 0 && c < j.nc) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k * 128 + lane];
      out[c] = t;
    }
    __syncthreads();
  }
}

extern "C" int mi_yolox_loss_bwd(const mi_yolox_loss_desc* d, const float* gw, float* dpreds, mi_stream_t st) {
  LossK k;
  int rc = loss_fill(d, &k);
  if (rc) return rc;
  MI_REQUIRE(gw && dpreds, "yolox_loss_bwd: null");
  const int64_t total = (int64_t)d->B * d->A * k.nch;
  MI_REQUIRE(total < (1LL << 31) - (1 << 22), "yolox_loss_bwd: B * A * (5 + classes) = %lld exceeds the 32-bit element index", (long long)total);
  int64_t nb = (total + 1023) / 1024;   // 4 elements per thread
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(yolox_loss_bwd_cls_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)st, k, gw, dpreds);
  hipLaunchKernelGGL(yolox_loss_bwd_box_kernel, dim3(mi_cdiv(d->A, 256), d->B), dim3(256), 0, (hipStream_t)st, k, gw, dpreds);
  MI_CHECK_LAUNCH("yolox_loss_bwd");
  return MI_OK;
}

// ---- extract one prediction conv's out-gradient: channels [c0,c0+nc) -> bf16 NHWC map, pads zero
__global__ __launch_bounds__(256) void split_dpreds_kernel(const float* __restrict__ dpreds, int B, int A, int nch,
                                                           int a0, int HW, int c0, int nc, __bf16* dst, int ld) {
  const int64_t total = (int64_t)B * HW * ld;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int j = (int)(idx % ld);
    const int64_t bp = idx / ld;
    const int b = (int)(bp / HW), pidx = (int)(bp % HW);
    float v = 0.f;
    if (j < nc) v = dpreds[((size_t)b * A + a0 + pidx) * nch + c0 + j];
    dst[idx] = (__bf16)v;
  }
}
extern "C" int mi_yolox_split_dpreds(const float* dpreds, int B, int A, int nch, int a0, int HW, int c0, int nc,
                                     void* dst, int ld, mi_stream_t st) {
  MI_REQUIRE(dpreds && dst && ld >= nc && c0 >= 0 && c0 + nc <= nch && a0 >= 0 && a0 + HW <= A, "split_dpreds: args");
  const int64_t total = (int64_t)B * HW * ld;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_dpreds_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)st, dpreds, B, A, nch, a0, HW,
                     c0, nc, (__bf16*)dst, ld);
  MI_CHECK_LAUNCH("split_dpreds");
  return MI_OK;
}

// all prediction convs' out-gradient maps in ONE launch: blockIdx.y = job
#define MI_SPLIT_MAX_JOBS 16
struct SplitK {
  const float* dpreds;
  int B, A, nch, njobs;
  mi_split_job jobs[MI_SPLIT_MAX_JOBS];
};
__global__ __launch_bounds__(256) void split_dpreds_batch_kernel(const SplitK p) {
  const mi_split_job j = p.jobs[blockIdx.y];
  const int ld8 = j.ld >> 3;
  const int64_t total = (int64_t)p.B * j.HW * ld8;  // one 16-byte vector (8 channels) per item
  uint4* dst = (uint4*)j.dst;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int j8 = (int)(idx % ld8);
    const int64_t bp = idx / ld8;
    const int b = (int)(bp / j.HW), pidx = (int)(bp % j.HW);
    const float* src = p.dpreds + ((size_t)b * p.A + j.a0 + pidx) * p.nch + j.c0 + j8 * 8;
    unsigned short h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (j8 * 8 + e < j.nc) ? src[e] : 0.f;
      h[e] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
    dst[idx] = make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16),
                          h[6] | ((uint32_t)h[7] << 16));
  }
}
extern "C" int mi_yolox_split_dpreds_batch(const float* dpreds, int B, int A, int nch, const mi_split_job* jobs, int njobs,
                                           mi_stream_t st) {
  MI_REQUIRE(dpreds && jobs && njobs > 0 && njobs <= MI_SPLIT_MAX_JOBS, "split_dpreds_batch: args");
  SplitK k;
  k.dpreds = dpreds; k.B = B; k.A = A; k.nch = nch; k.njobs = njobs;
  int64_t most = 0;
  for (int n = 0; n < njobs; ++n) {
    const mi_split_job& j = jobs[n];
    MI_REQUIRE(j.dst && j.ld % 8 == 0 && j.ld >= j.nc && j.c0 >= 0 && j.c0 + j.nc <= nch && j.a0 >= 0 && j.a0 + j.HW <= A,
               "split_dpreds_batch: job %d", n);
    k.jobs[n] = j;
    const int64_t t = (int64_t)B * j.HW * (j.ld / 8);
    if (t > most) most = t;
  }
  int64_t blocks = (most + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(split_dpreds_batch_kernel, dim3((int)blocks, njobs), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("split_dpreds_batch");
  return MI_OK;
}

// ---- bias gradients of all prediction convs in two launches (fixed summation order):
// grad_bias[job][c] = sum_b sum_{a in [a0, a0+HW)} dpreds[b][a][c0 + c]
// stage 1 sums ALL nch columns of every distinct anchor range (FPN level) with fully coalesced row reads;
// stage 2 combines the block partials and scatters each job's channel slice.
#define MI_BIAS_MAX_JOBS 16
#define MI_BIAS_BLOCKS 512
struct BiasK {
  const float* dpreds;
  float* ws;  // [nlev][chunks][MI_BIAS_BLOCKS][128]
  int B, A, nch, njobs, nlev, chunks;
  int lev_a0[MI_BIAS_MAX_JOBS], lev_hw[MI_BIAS_MAX_JOBS], job_lev[MI_BIAS_MAX_JOBS];
  mi_bias_job jobs[MI_BIAS_MAX_JOBS];
};
__global__ __launch_bounds__(256) void bias_grads_stage1_kernel(const BiasK p) {
  __shared__ float red[256];
  const int a0 = p.lev_a0[blockIdx.y], HW = p.lev_hw[blockIdx.y];
  const int c = blockIdx.z * 128 + (threadIdx.x & 127), rl = threadIdx.x >> 7;  // 128 channel lanes x 2 row lanes
  const long rows = (long)p.B * HW;
  float acc = 0.f;
  if (c < p.nch) {
    const long step = (long)gridDim.x * 2;
    long r = (long)blockIdx.x * 2 + rl;
    float a0_ = 0.f, a1_ = 0.f, a2_ = 0.f, a3_ = 0.f;
    auto at = [&](long rr) {
      const long b = rr / HW, a = rr - b * HW;
      return p.dpreds[((size_t)b * p.A + a0 + a) * p.nch + c];
    };
    for (; r + 3 * step < rows; r += 4 * step) {  // 4 independent loads in flight
      const float v0 = at(r), v1 = at(r + step), v2 = at(r + 2 * step), v3 = at(r + 3 * step);
      a0_ += v0; a1_ += v1; a2_ += v2; a3_ += v3;
    }
    for (; r < rows; r += step) a0_ += at(r);
    acc = (a0_ + a1_) + (a2_ + a3_);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 128)
    p.ws[(((size_t)blockIdx.y * p.chunks + blockIdx.z) * MI_BIAS_BLOCKS + blockIdx.x) * 128 + threadIdx.x] =
        red[threadIdx.x] + red[threadIdx.x + 128];
}
__global__ __launch_bounds__(128) void bias_grads_stage2_kernel(const BiasK p) {
  const mi_bias_job j = p.jobs[blockIdx.x];
  const int lev = p.job_lev[blockIdx.x];
  for (int c = threadIdx.x; c < j.nc; c += 128) {
    const int cc = j.c0 + c;
    const float* w = p.ws + ((size_t)lev * p.chunks + (cc >> 7)) * MI_BIAS_BLOCKS * 128 + (cc & 127);
    float s = 0.f;
    for (int b = 0; b < MI_BIAS_BLOCKS; ++b) s += w[(size_t)b * 128];
    j.out[c] = s;
  }
}
extern "C" int mi_yolox_bias_grads(const float* dpreds, int B, int A, int nch, const mi_bias_job* jobs, int njobs,
                                   float* ws, mi_stream_t st) {
  MI_REQUIRE(dpreds && jobs && ws && njobs > 0 && njobs <= MI_BIAS_MAX_JOBS && nch > 0, "bias_grads: args");
  BiasK k;
  k.dpreds = dpreds; k.ws = ws; k.B = B; k.A = A; k.nch = nch; k.njobs = njobs; k.nlev = 0; k.chunks = (nch + 127) / 128;
  for (int i = 0; i < njobs; ++i) {
    MI_REQUIRE(jobs[i].out && jobs[i].nc >= 1 && jobs[i].c0 >= 0 && jobs[i].c0 + jobs[i].nc <= nch &&
                   jobs[i].a0 >= 0 && jobs[i].a0 + jobs[i].HW <= A, "bias_grads: job %d", i);
    k.jobs[i] = jobs[i];
    int lev = -1;
    for (int l = 0; l < k.nlev; ++l)
      if (k.lev_a0[l] == jobs[i].a0 && k.lev_hw[l] == jobs[i].HW) lev = l;
    if (lev < 0) {
      lev = k.nlev++;
      k.lev_a0[lev] = jobs[i].a0;
      k.lev_hw[lev] = jobs[i].HW;
    }
    k.job_lev[i] = lev;
  }
  MI_REQUIRE(k.nlev * k.chunks <= 16, "bias_grads: %d levels x %d channel chunks of 128 exceed the 16-slab scratch", k.nlev, k.chunks);
  hipLaunchKernelGGL(bias_grads_stage1_kernel, dim3(MI_BIAS_BLOCKS, k.nlev, k.chunks), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("bias_grads1");
  hipLaunchKernelGGL(bias_grads_stage2_kernel, dim3(njobs), dim3(128), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("bias_grads2");
  return MI_OK;
}

extern "C" int mi_yolox_loss_bwd_fused(const mi_yolox_loss_desc* d, const float* gw, float* dpreds, const mi_split_job* split_jobs,
                                       int nsplit, const mi_bias_job* bias_jobs, int nbias, float* ws, int64_t ws_floats,
                                       mi_stream_t st) {
  LossBwdFusedK k;
  int rc = loss_fill(d, &k.l);
  if (rc) return rc;
  MI_REQUIRE(gw && split_jobs && nsplit > 0 && nsplit <= MI_LBF_MAX_JOBS && nbias >= 0 && (nbias == 0 || (bias_jobs && ws)),
             "yolox_loss_bwd_fused: args");
  const int nch = k.l.nch;
  MI_REQUIRE((int64_t)d->B * d->A * nch < (1LL << 31) - (1 << 22), "yolox_loss_bwd_fused: element index exceeds 32 bits");
  k.gw = gw; k.dpreds = dpreds; k.ws = ws; k.njobs = nsplit; k.ldmax = 0;
  int64_t most = 0, cover = 0;
  for (int n = 0; n < nsplit; ++n) {
    const mi_split_job& j = split_jobs[n];
    MI_REQUIRE(j.dst && j.ld % 8 == 0 && j.ld >= j.nc && j.ld <= 2048 && j.nc >= 1 && j.c0 >= 0 && j.c0 + j.nc <= nch && j.a0 >= 0 &&
                   j.a0 + j.HW <= d->A && (j.c0 >= 5 || j.c0 + j.nc <= 5) && (j.c0 >= 4 || j.c0 + j.nc <= 4 || (j.c0 == 0 && j.nc == 5)),
               "yolox_loss_bwd_fused: job %d (a job is the box columns, the objectness column, both (0 .. 4) or class columns)", n);
    k.jobs[n] = j;
    k.bias_out[n] = nullptr;
    if (j.ld > k.ldmax) k.ldmax = j.ld;
    const int64_t t = (int64_t)d->B * j.HW * (j.ld / 8);
    if (t > most) most = t;
    cover += (int64_t)j.HW * j.nc;
  }
  // the jobs must tile [A][nch] exactly once: dpreds is WRITTEN here, not zero-filled first
  MI_REQUIRE(cover == (int64_t)d->A * nch, "yolox_loss_bwd_fused: the jobs cover %lld of %lld (anchor, channel) cells", (long long)cover,
             (long long)d->A * nch);
  for (int n = 0; n < nbias; ++n) {
    int hit = -1;
    for (int m = 0; m < nsplit; ++m)
      if (split_jobs[m].a0 == bias_jobs[n].a0 && split_jobs[m].HW == bias_jobs[n].HW && split_jobs[m].c0 == bias_jobs[n].c0 &&
          split_jobs[m].nc == bias_jobs[n].nc) hit = m;
    MI_REQUIRE(hit >= 0 && bias_jobs[n].out, "yolox_loss_bwd_fused: bias job %d matches no out-gradient map", n);
    k.bias_out[hit] = bias_jobs[n].out;
  }
  int64_t blocks = (most + 255) / 256;
  if (blocks > 512) blocks = 512;
  if (nbias && blocks * nsplit * k.ldmax > ws_floats) blocks = ws_floats / ((int64_t)nsplit * k.ldmax);
  MI_REQUIRE(blocks >= 1, "yolox_loss_bwd_fused: scratch of %lld floats too small", (long long)ws_floats);
  hipStream_t s = (hipStream_t)st;
  hipLaunchKernelGGL(yolox_loss_bwd_fused_kernel, dim3((int)blocks, nsplit), dim3(256), 0, s, k);
  if (nbias) hipLaunchKernelGGL(yolox_loss_bwd_bias_kernel, dim3(nsplit), dim3(1024), 0, s, k, (int)blocks);
  MI_CHECK_LAUNCH("yolox_loss_bwd_fused");
  return MI_OK;
}

// ---- eval decode (yolox_head.py:197-224,247-272): sigmoid(obj, cls), xy/wh decode, in place
__global__ __launch_bounds__(256) void yolox_decode_kernel(float* preds, const float* anchors, int B, int A, int nch) {
  const int64_t total = (int64_t)B * A;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int a = (int)(idx % A);
    float* pr = preds + idx * nch;
    const float gxs = anchors[a * 3 + 0], gys = anchors[a * 3 + 1], st = anchors[a * 3 + 2];
    pr[0] = (pr[0] + gxs) * st;
    pr[1] = (pr[1] + gys) * st;
    pr[2] = expf(pr[2]) * st;
    pr[3] = expf(pr[3]) * st;
    for (int c = 4; c < nch; ++c) pr[c] = sigmoid_ref(pr[c]);
  }
}
extern "C" int mi_yolox_decode(float* preds, const float* anchors, int B, int A, int ncls, mi_stream_t st) {
  MI_REQUIRE(preds && anchors && B > 0 && A > 0 && ncls > 0, "decode: args");
  const int64_t total = (int64_t)B * A;
  hipLaunchKernelGGL(yolox_decode_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)st, preds,
                     anchors, B, A, ncls + 5);
  MI_CHECK_LAUNCH("decode");
  return MI_OK;
}


// ---- decode_outputs, ONNX-export layout (yolox_head.py:263-269): from the DECODED predictions [B][A][5+ncls]
// (mi_yolox_decode: boxes in pixels, sigmoid on obj / cls) to [B][A][6+ncls] = (xy, wh, conf, argmax(prob) as float, prob);
// torch.argmax: the first maximal class
__global__ __launch_bounds__(256) void yolox_onnx_layout_kernel(const float* __restrict__ dec, float* __restrict__ out,
                                                                int64_t rows, int ncls) {
  const int64_t r = blockIdx.x * 256LL + threadIdx.x;
  if (r >= rows) return;
  const float* p = dec + r * (5 + ncls);
  float* o = out + r * (6 + ncls);
  for (int c = 0; c < 5; ++c) o[c] = p[c];
  int best = 0;
  float bv = p[5];
  for (int c = 0; c < ncls; ++c) {
    const float v = p[5 + c];
    o[6 + c] = v;
    if (v > bv) { bv = v; best = c; }
  }
  o[5] = (float)best;
}
extern "C" int mi_yolox_onnx_layout(const float* decoded, float* out, int B, int A, int ncls, mi_stream_t st) {
  MI_REQUIRE(decoded && out && B > 0 && A > 0 && ncls > 0, "onnx_layout: args");
  const int64_t rows = (int64_t)B * A;
  hipLaunchKernelGGL(yolox_onnx_layout_kernel, dim3((int)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)st, decoded, out,
                     rows, ncls);
  MI_CHECK_LAUNCH("onnx_layout");
  return MI_OK;
}
