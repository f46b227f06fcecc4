// Input pipeline on the GPU: the pixel work of the reference's mosaic branch
//   MyDatasetMapper2.__call__  yolov7/data/dataset_mapper.py:523-598  (cv2.resize of four images, paste on a 114 canvas)
//   random_perspective         yolov7/data/transforms/data_augment.py:67-75 (cv2.warpAffine, border 114)
//   YOLOX.preprocess_image     yolov7/modeling/meta_arch/yolox.py:95-130 (pad to the batch size with 114, NCHW)
// for a whole batch in two launches over device-resident decoded images.  Byte / integer arithmetic only: OpenCV's
// fixed-point INTER_LINEAR resize (11-bit coefficients, its 8-bit vertical pass) and warpAffine (source coordinates in
// 1/32 px, 15-bit weight table), exactly as oracle/augment_oracle.py restates them - the tests hold the two bit-identical.
// Compiled with -ffp-contract=off: the fp64 / fp32 coordinate expressions must round as the restatement's do.
#include "common.h"

struct PasteJob {   // mirrors mi_mosaic_paste_job
  const unsigned char* src;
  unsigned char* canvas;
  int h0, w0, rh, rw, cw, x1a, y1a, x2a, y2a, x1b, y1b, blk0;
  int fsrc, pad_;   // fsrc: the reference's image is float32 here (YOLOFRandomDistortion): cv2.resize's float path, then truncation
};
struct WarpJob {    // mirrors mi_warp_job
  const unsigned char* canvas;
  unsigned char* out;          // plane 0 of this sample: [3][Hp][Wp]
  double minv[6];
  int ch, cw, h, w, Hp, Wp, border, blk0;
};

__device__ __forceinline__ void resize_coef(int d, double scale, int src, int* s, int* a0, int* a1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int si = (int)floorf(f);
  f = f - (float)si;
  if (si < 0) { f = 0.f; si = 0; }
  if (si >= src - 1) { f = 0.f; si = src - 1; }
  float v0 = rintf((1.0f - f) * 2048.f), v1 = rintf(f * 2048.f);
  *a0 = (int)fminf(fmaxf(v0, -32768.f), 32767.f);
  *a1 = (int)fminf(fmaxf(v1, -32768.f), 32767.f);
  *s = si;
}

__device__ __forceinline__ void lin_coef_f(int d, double scale, int src, int* s, float* f0, float* f1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int si = (int)floorf(f);
  f = f - (float)si;
  if (si < 0) { f = 0.f; si = 0; }
  if (si >= src - 1) { f = 0.f; si = src - 1; }
  *s = si; *f0 = 1.0f - f; *f1 = f;
}

__global__ __launch_bounds__(256) void mosaic_paste_kernel(const PasteJob* __restrict__ jobs, int njobs) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
  const PasteJob p = jobs[j];
  const int rw_ = p.x2a - p.x1a, rh_ = p.y2a - p.y1a;
  const int idx = ((int)blockIdx.x - p.blk0) * 256 + threadIdx.x;
  if (idx >= rw_ * rh_) return;
  const int yy = idx / rw_, xx = idx - yy * rw_;
  const int rx = xx + p.x1b, ry = yy + p.y1b;
  int sx, ax0, ax1, sy, by0, by1;
  resize_coef(rx, (double)p.w0 / (double)p.rw, p.w0, &sx, &ax0, &ax1);
  resize_coef(ry, (double)p.h0 / (double)p.rh, p.h0, &sy, &by0, &by1);
  const int sx1 = min(sx + 1, p.w0 - 1), sy1 = min(sy + 1, p.h0 - 1);
  const unsigned char* r0 = p.src + (size_t)sy * p.w0 * 3;
  const unsigned char* r1 = p.src + (size_t)sy1 * p.w0 * 3;
  unsigned char* d = p.canvas + ((size_t)(p.y1a + yy) * p.cw + p.x1a + xx) * 3;
  if (p.fsrc) {
    // cv2.resize of a float32 image (resize.cpp generic path, WT = float): D = S[sx] * a0 + S[sx + 1] * a1 per row, then
    // D0 * b0 + D1 * b1, every product and sum in float32, no fused multiply-add (-ffp-contract=off); img4[...] = img
    // truncates (oracle/augment_oracle.py::resize_linear_f32: build-dependent in the real library on flat regions)
    if (p.rw == p.w0 && p.rh == p.h0) {      // (same size: cv2.resize copies)
#pragma unroll
      for (int c = 0; c < 3; ++c) d[c] = p.src[((size_t)ry * p.w0 + rx) * 3 + c];
      return;
    }
    int fsx, fsy;
    float fx0, fx1, fy0, fy1;
    lin_coef_f(rx, (double)p.w0 / (double)p.rw, p.w0, &fsx, &fx0, &fx1);
    lin_coef_f(ry, (double)p.h0 / (double)p.rh, p.h0, &fsy, &fy0, &fy1);
    const int fsx1 = min(fsx + 1, p.w0 - 1), fsy1 = min(fsy + 1, p.h0 - 1);
    const unsigned char* q0 = p.src + (size_t)fsy * p.w0 * 3;
    const unsigned char* q1 = p.src + (size_t)fsy1 * p.w0 * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float h0 = (float)q0[fsx * 3 + c] * fx0 + (float)q0[fsx1 * 3 + c] * fx1;
      const float h1 = (float)q1[fsx * 3 + c] * fx0 + (float)q1[fsx1 * 3 + c] * fx1;
      float v = h0 * fy0 + h1 * fy1;
      v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
      d[c] = (unsigned char)v;
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = (int)r0[sx * 3 + c] * ax0 + (int)r0[sx1 * 3 + c] * ax1;
    const int h1 = (int)r1[sx * 3 + c] * ax0 + (int)r1[sx1 * 3 + c] * ax1;
    int v = ((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16);
    v = (v + 2) >> 2;
    d[c] = (unsigned char)min(max(v, 0), 255);
  }
}

// ---- mixup (MyDatasetMapper2.mixup, dataset_mapper.py:686-768), in place on the warped sample
struct MixJob {     // mirrors mi_mixup_job
  const unsigned char* src;
  unsigned char* out;
  int h0, w0, rh1, rw1, dh, dw, oh, ow, flip, x_off, y_off, th, tw, Hp, Wp, blk0;
  int fsrc, pad_;   // fsrc: the pool image is float32 in the reference: the first resize runs cv2's float path, unrounded
};
__global__ __launch_bounds__(256) void mixup_blend_kernel(const MixJob* __restrict__ jobs, int njobs) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
  const MixJob p = jobs[j];
  const int idx = ((int)blockIdx.x - p.blk0) * 256 + threadIdx.x;
  if (idx >= p.tw * p.th) return;
  const int y = idx / p.tw, x = idx - y * p.tw;
  const int py = y + p.y_off, px = x + p.x_off;
  int pad[3] = {0, 0, 0};
  if (py < p.oh && px < p.ow) {
    const int u = p.flip ? p.ow - 1 - px : px;
    // second resize (float64 image, float32 coefficients): canvas dh x dw -> oh x ow
    int sx, sy;
    float fx0, fx1, fy0, fy1;
    lin_coef_f(u, (double)p.dw / (double)p.ow, p.dw, &sx, &fx0, &fx1);
    lin_coef_f(py, (double)p.dh / (double)p.oh, p.dh, &sy, &fy0, &fy1);
    const int xs[2] = {sx, min(sx + 1, p.dw - 1)}, ys[2] = {sy, min(sy + 1, p.dh - 1)};
    // first resize (uint8 fixed point): source h0 x w0 -> rh1 x rw1, in the canvas' top-left corner; 114.0 elsewhere
    int cxs[2], cxa0[2], cxa1[2], cys[2], cyb0[2], cyb1[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      resize_coef(min(xs[k], p.rw1 - 1), (double)p.w0 / (double)p.rw1, p.w0, &cxs[k], &cxa0[k], &cxa1[k]);
      resize_coef(min(ys[k], p.rh1 - 1), (double)p.h0 / (double)p.rh1, p.h0, &cys[k], &cyb0[k], &cyb1[k]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double S[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (ys[a] < p.rh1 && xs[b] < p.rw1 && p.fsrc) {
            // float32 source (YOLOFRandomDistortion): cv2.resize's float path, the result kept as it is (.astype(float32))
            const int cx = min(xs[b], p.rw1 - 1), cy = min(ys[a], p.rh1 - 1);
            if (p.rw1 == p.w0 && p.rh1 == p.h0) {
              S[a][b] = (double)p.src[((size_t)cy * p.w0 + cx) * 3 + c];
            } else {
              int fsx, fsy;
              float gx0, gx1, gy0, gy1;
              lin_coef_f(cx, (double)p.w0 / (double)p.rw1, p.w0, &fsx, &gx0, &gx1);
              lin_coef_f(cy, (double)p.h0 / (double)p.rh1, p.h0, &fsy, &gy0, &gy1);
              const int fsx1 = min(fsx + 1, p.w0 - 1), fsy1 = min(fsy + 1, p.h0 - 1);
              const unsigned char* q0 = p.src + (size_t)fsy * p.w0 * 3;
              const unsigned char* q1 = p.src + (size_t)fsy1 * p.w0 * 3;
              const float h0 = (float)q0[fsx * 3 + c] * gx0 + (float)q0[fsx1 * 3 + c] * gx1;
              const float h1 = (float)q1[fsx * 3 + c] * gx0 + (float)q1[fsx1 * 3 + c] * gx1;
              S[a][b] = (double)(h0 * gy0 + h1 * gy1);
            }
          } else if (ys[a] < p.rh1 && xs[b] < p.rw1) {
            const int x0 = cxs[b], x1 = min(x0 + 1, p.w0 - 1), y0 = cys[a], y1 = min(y0 + 1, p.h0 - 1);
            const unsigned char* r0 = p.src + (size_t)y0 * p.w0 * 3;
            const unsigned char* r1 = p.src + (size_t)y1 * p.w0 * 3;
            const int h0 = (int)r0[x0 * 3 + c] * cxa0[b] + (int)r0[x1 * 3 + c] * cxa1[b];
            const int h1 = (int)r1[x0 * 3 + c] * cxa0[b] + (int)r1[x1 * 3 + c] * cxa1[b];
            int v = ((cyb0[a] * (h0 >> 4)) >> 16) + ((cyb1[a] * (h1 >> 4)) >> 16);
            v = min(max((v + 2) >> 2, 0), 255);
            S[a][b] = (double)v;
          } else {
            S[a][b] = 114.0;
          }
        }
      const double r0 = S[0][0] * (double)fx0 + S[0][1] * (double)fx1;
      const double r1 = S[1][0] * (double)fx0 + S[1][1] * (double)fx1;
      const double v = r0 * (double)fy0 + r1 * (double)fy1;
      pad[c] = (int)(unsigned char)v;
    }
  }
  unsigned char* d = p.out + (size_t)y * p.Wp + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    unsigned char* q = d + (size_t)c * p.Hp * p.Wp;
    const float f = 0.5f * (float)(*q) + 0.5f * (float)pad[c];
    *q = (unsigned char)f;
  }
}

__device__ __forceinline__ long long sat_i32(double v) {
  v = rint(v);
  if (v < -2147483648.0) v = -2147483648.0;
  if (v > 2147483647.0) v = 2147483647.0;
  return (long long)v;
}

__global__ __launch_bounds__(256) void warp_affine_kernel(const WarpJob* __restrict__ jobs, int njobs,
                                                          const int* __restrict__ tab) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
  const WarpJob& p = jobs[j];
  const int idx = ((int)blockIdx.x - p.blk0) * 256 + threadIdx.x;
  if (idx >= p.w * p.h) return;
  const int y = idx / p.w, x = idx - y * p.w;
  const long long ax = sat_i32(p.minv[0] * (double)x * 1024.0), bx = sat_i32(p.minv[3] * (double)x * 1024.0);
  const long long ay = sat_i32((p.minv[1] * (double)y + p.minv[2]) * 1024.0) + 16;
  const long long by = sat_i32((p.minv[4] * (double)y + p.minv[5]) * 1024.0) + 16;
  const long long X = (ax + ay) >> 5, Y = (bx + by) >> 5;
  long long sxl = X >> 5, syl = Y >> 5;
  sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
  syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
  const int sx = (int)sxl, sy = (int)syl;
  const int* w = tab + (((int)(Y & 31) * 32) + (int)(X & 31)) * 4;
  const bool x0 = sx >= 0 && sx < p.cw, x1 = sx + 1 >= 0 && sx + 1 < p.cw;
  const bool y0 = sy >= 0 && sy < p.ch, y1 = sy + 1 >= 0 && sy + 1 < p.ch;
  const unsigned char* r0 = p.canvas + (size_t)(y0 ? sy : 0) * p.cw * 3;
  const unsigned char* r1 = p.canvas + (size_t)(y1 ? sy + 1 : 0) * p.cw * 3;
  const int o0 = (x0 ? sx : 0) * 3, o1 = (x1 ? sx + 1 : 0) * 3;
  unsigned char* d = p.out + (size_t)y * p.Wp + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int t00 = (y0 && x0) ? r0[o0 + c] : p.border, t01 = (y0 && x1) ? r0[o1 + c] : p.border;
    const int t10 = (y1 && x0) ? r1[o0 + c] : p.border, t11 = (y1 && x1) ? r1[o1 + c] : p.border;
    int acc = t00 * w[0] + t01 * w[1] + t10 * w[2] + t11 * w[3];
    acc = (acc + (1 << 14)) >> 15;
    d[(size_t)c * p.Hp * p.Wp] = (unsigned char)min(max(acc, 0), 255);
  }
}

// imgwarp.cpp initInterTab2D(INTER_LINEAR, fixed point): see oracle/augment_oracle.py::bilinear_tab.  Kept as int: the
// cell of an integer source position carries the full weight 32768, which a short cannot hold.
static void build_bilinear_tab(int* tab) {
  float t1[32][2];
  for (int i = 0; i < 32; ++i) {
    const float x = (float)i * (1.0f / 32);
    t1[i][0] = 1.0f - x;
    t1[i][1] = x;
  }
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      const float w[4] = {t1[i][0] * t1[j][0], t1[i][0] * t1[j][1], t1[i][1] * t1[j][0], t1[i][1] * t1[j][1]};
      int it[4], sum = 0;
      for (int k = 0; k < 4; ++k) {
        float v = rintf(w[k] * 32768.f);
        if (v > 32767.f) v = 32767.f;
        it[k] = (int)v;
        sum += it[k];
      }
      const int diff = sum - 32768;
      int hi = 0, lo = 0;
      for (int k = 1; k < 4; ++k) {
        if (it[k] > it[hi]) hi = k;
        if (it[k] < it[lo]) lo = k;
      }
      if (diff < 0) it[hi] -= diff;
      else if (diff > 0) it[lo] -= diff;
      for (int k = 0; k < 4; ++k) tab[(i * 32 + j) * 4 + k] = it[k];
    }
}

static int* g_tab_dev = nullptr;

extern "C" int mi_mosaic_jobs_layout(mi_mosaic_paste_job* paste, int npaste, mi_warp_job* warp, int nwarp) {
  MI_REQUIRE((paste || !npaste) && (warp || !nwarp) && npaste >= 0 && nwarp >= 0, "mosaic_jobs_layout: args");
  int blk = 0;
  for (int j = 0; j < npaste; ++j) {
    mi_mosaic_paste_job& p = paste[j];
    MI_REQUIRE(p.h0 > 0 && p.w0 > 0 && p.rh > 0 && p.rw > 0 && p.cw > 0 && p.x2a >= p.x1a && p.y2a >= p.y1a && p.x1b >= 0 &&
                   p.y1b >= 0 && p.x1b + (p.x2a - p.x1a) <= p.rw && p.y1b + (p.y2a - p.y1a) <= p.rh && p.x1a >= 0 && p.x2a <= p.cw,
               "mosaic_jobs_layout: paste job %d geometry", j);
    p.blk0 = blk;
    blk += (int)(((long long)(p.x2a - p.x1a) * (p.y2a - p.y1a) + 255) / 256);
  }
  const int pb = blk;
  blk = 0;
  for (int j = 0; j < nwarp; ++j) {
    mi_warp_job& w = warp[j];
    MI_REQUIRE(w.h > 0 && w.w > 0 && w.ch > 0 && w.cw > 0 && w.Hp >= w.h && w.Wp >= w.w, "mosaic_jobs_layout: warp job %d geometry", j);
    w.blk0 = blk;
    blk += (int)(((long long)w.w * w.h + 255) / 256);
  }
  return pb > blk ? pb : blk;
}

extern "C" int mi_mixup_jobs_layout(mi_mixup_job* jobs, int njobs) {
  MI_REQUIRE(jobs && njobs > 0, "mixup_jobs_layout: args");
  int blk = 0;
  for (int j = 0; j < njobs; ++j) {
    mi_mixup_job& p = jobs[j];
    MI_REQUIRE(p.src && p.out && p.h0 > 0 && p.w0 > 0 && p.rh1 > 0 && p.rw1 > 0 && p.rh1 <= p.dh && p.rw1 <= p.dw && p.oh > 0 && p.ow > 0 &&
                   p.th > 0 && p.tw > 0 && p.Hp >= p.th && p.Wp >= p.tw && p.x_off >= 0 && p.y_off >= 0,
               "mixup_jobs_layout: job %d geometry", j);
    p.blk0 = blk;
    blk += (int)(((long long)p.tw * p.th + 255) / 256);
  }
  return blk;
}
extern "C" int mi_mixup_blend(const mi_mixup_job* jobs_dev, int njobs, int total_blocks, mi_stream_t st) {
  MI_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, "mixup_blend: args");
  static_assert(sizeof(MixJob) == sizeof(mi_mixup_job), "job layout");
  hipLaunchKernelGGL(mixup_blend_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)st, (const MixJob*)jobs_dev, njobs);
  MI_CHECK_LAUNCH("mixup_blend");
  return MI_OK;
}

extern "C" int mi_mosaic_paste(const mi_mosaic_paste_job* jobs_dev, int njobs, int total_blocks, mi_stream_t st) {
  MI_REQUIRE(jobs_dev && njobs > 0 && total_blocks >= 0, "mosaic_paste: args");
  static_assert(sizeof(PasteJob) == sizeof(mi_mosaic_paste_job), "job layout");
  if (total_blocks == 0) return MI_OK;
  hipLaunchKernelGGL(mosaic_paste_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)st, (const PasteJob*)jobs_dev, njobs);
  MI_CHECK_LAUNCH("mosaic_paste");
  return MI_OK;
}

extern "C" int mi_warp_affine_u8(const mi_warp_job* jobs_dev, int njobs, int total_blocks, mi_stream_t st) {
  MI_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, "warp_affine_u8: args");
  static_assert(sizeof(WarpJob) == sizeof(mi_warp_job), "job layout");
  if (!g_tab_dev) {
    int host[32 * 32 * 4];
    build_bilinear_tab(host);
    if (hipMalloc(&g_tab_dev, sizeof(host)) != hipSuccess) MI_FAIL(MI_ELAUNCH, "warp_affine_u8: table allocation");
    if (hipMemcpy(g_tab_dev, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess) MI_FAIL(MI_ELAUNCH, "warp_affine_u8: table copy");
  }
  hipLaunchKernelGGL(warp_affine_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)st, (const WarpJob*)jobs_dev, njobs,
                     (const int*)g_tab_dev);
  MI_CHECK_LAUNCH("warp_affine_u8");
  return MI_OK;
}


// ================================================================= detectron2 T.* front of the input pipeline
// T.ResizeShortestEdge (Pillow's 8-bit bilinear resampling, pil_resize_core.h) + T.RandomFlip (horizontal, vertical) +
// YOLOFRandomShift of every image the mapper loads (yolov7/data/detection_utils.py:37-86, dataset_mapper.py:642-683): two
// flat launches over a job table - the horizontal pass into a scratch image, then vertical pass + flips + shift + the
// caller's destination strides (an HWC image for the mosaic pool, or a plane triple of the padded NCHW batch).
#include "pil_resize_core.h"
static_assert(sizeof(PilJob) == sizeof(mi_pil_resize_job), "PilJob mirrors mi_pil_resize_job");

__global__ __launch_bounds__(256) void pil_resize_h_kernel(const PilJob* __restrict__ jobs, int njobs) {
  pil_h_thread(jobs, njobs, (int)blockIdx.x, (int)threadIdx.x);
}
__global__ __launch_bounds__(256) void pil_resize_v_kernel(const PilJob* __restrict__ jobs, int njobs) {
  pil_v_thread(jobs, njobs, (int)blockIdx.x, (int)threadIdx.x);
}

extern "C" int mi_pil_resize_jobs_layout(mi_pil_resize_job* jobs, int n, int32_t* blocks_h, int32_t* blocks_v) {
  MI_REQUIRE(jobs && n > 0 && blocks_h && blocks_v, "pil_resize_layout: args");
  int64_t bh = 0, bv = 0;
  for (int i = 0; i < n; ++i) {
    mi_pil_resize_job& j = jobs[i];
    MI_REQUIRE(j.src && j.dst && j.h0 > 0 && j.w0 > 0 && j.nh > 0 && j.nw > 0, "pil_resize_layout: job %d: null / empty", i);
    MI_REQUIRE(j.nw == j.w0 || j.tmp, "pil_resize_layout: job %d needs the scratch image of the horizontal pass", i);
    MI_REQUIRE(j.src_ld >= (int64_t)j.w0 * 3, "pil_resize_layout: job %d: source row stride %lld < 3 w0", i, (long long)j.src_ld);
    const int kx = (int)((j.w0 + j.nw - 1) / j.nw), ky = (int)((j.h0 + j.nh - 1) / j.nh);    // ceil(scale)
    MI_REQUIRE(kx * 2 + 1 <= PIL_MAX_TAPS && ky * 2 + 1 <= PIL_MAX_TAPS, "pil_resize_layout: job %d shrinks by more than 8", i);
    MI_REQUIRE(j.shift_x > -j.nw && j.shift_x < j.nw && j.shift_y > -j.nh && j.shift_y < j.nh, "pil_resize_layout: job %d: shift", i);
    j.blk0h = (int32_t)bh;
    j.blk0v = (int32_t)bv;
    if (j.nw != j.w0) bh += ((int64_t)j.h0 * j.nw + 255) / 256;
    bv += ((int64_t)j.nh * j.nw + 255) / 256;
    MI_REQUIRE(bh < (1LL << 30) && bv < (1LL << 30), "pil_resize_layout: too many blocks");
  }
  *blocks_h = (int32_t)bh;
  *blocks_v = (int32_t)bv;
  return MI_OK;
}
extern "C" int mi_pil_resize_h(const mi_pil_resize_job* jobs_dev, int n, int total_blocks, mi_stream_t st) {
  MI_REQUIRE(jobs_dev && n > 0 && total_blocks >= 0, "pil_resize_h: args");
  if (total_blocks == 0) return MI_OK;                        // every job keeps its width
  hipLaunchKernelGGL(pil_resize_h_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)st, (const PilJob*)jobs_dev, n);
  MI_CHECK_LAUNCH("pil_resize_h");
  return MI_OK;
}
extern "C" int mi_pil_resize_v(const mi_pil_resize_job* jobs_dev, int n, int total_blocks, mi_stream_t st) {
  MI_REQUIRE(jobs_dev && n > 0 && total_blocks > 0, "pil_resize_v: args");
  hipLaunchKernelGGL(pil_resize_v_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)st, (const PilJob*)jobs_dev, n);
  MI_CHECK_LAUNCH("pil_resize_v");
  return MI_OK;
}
