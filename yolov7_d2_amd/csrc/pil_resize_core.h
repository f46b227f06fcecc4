// Pillow's 8-bit bilinear resampling (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
// ImagingResampleHorizontal_8bpc / Vertical_8bpc) as per-output-pixel functions: what detectron2's ResizeTransform.apply_image
// runs for a uint8 image (PIL.Image.resize(BILINEAR)) inside T.ResizeShortestEdge - the first augmentation of the
// reference's build_normal_augmentation (yolov7/data/detection_utils.py:37-86), applied to every image
// MyDatasetMapper2._load_image_with_annos loads (yolov7/data/dataset_mapper.py:642-683).
// Plain C++ with no HIP types: the kernels in augment.hip call these on the device, tests/native/pil_resize_host.cpp
// compiles the SAME functions for the host and the CPU test holds them bit-identical to the installed Pillow.
// Coefficients are fp64 exactly as the C library computes them (translation units including this are built with
// -ffp-contract=off), then 22-bit fixed point; the horizontal pass rounds to 8 bits before the vertical pass reads it.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define MI_HD __host__ __device__ __forceinline__
#else
#define MI_HD static inline
#endif

#define PIL_PRECISION_BITS 22          // 32 - 8 - 2
#define PIL_MAX_TAPS 17                // ceil(scale) * 2 + 1: down-scaling factors up to 8

struct PilJob {   // mirrors mi_pil_resize_job (include/mi355_det.h)
  const unsigned char* src;    // HWC uint8, rows src_ld bytes apart (a crop window = an offset pointer + the parent's row stride)
  unsigned char* tmp;
  unsigned char* dst;
  int64_t dsc, dsy, dsx;
  int64_t src_ld;
  double sat_src;              // RandomSaturation: 1 - w as Python computes it (the weight of the grey image, fp64)
  int32_t h0, w0, nh, nw;
  int32_t hflip, vflip, shift_x, shift_y;
  int32_t src_hflip;           // the source is mirrored left-right BEFORE the resampling (T.RandomFlip ahead of the resize)
  int32_t color;               // bit 0: RandomSaturation, bit 1: RandomBrightness (detectron2 BlendTransform), bit 2:
                               // YOLOFRandomDistortion (cv2's 8-bit RGB <-> HSV around three float32 scalings); after the flips
  float sat_dst, bri_dst;      // w as float32 (numpy multiplies the float32 image by the weak Python scalar in float32)
  float dis_hue, dis_sat, dis_exp;   // float32(dhue * 179 / 255.), float32(dsat), float32(dexp) (transform.py:268-279)
  int32_t dis_pos;             // dhue > 0: H > 1 wraps down; else H < 0 wraps up
  int32_t blk0h, blk0v;
};

struct PilTaps {
  int x0, n;
  int k[PIL_MAX_TAPS];
};

// taps of output index xx of an axis resampled from in_size to out_size (full-axis box)
MI_HD void pil_taps(int in_size, int out_size, int xx, PilTaps* t) {
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;                  // bilinear: filterp->support = 1.0
  const double center = (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > PIL_MAX_TAPS) xmax = PIL_MAX_TAPS;               // (excluded by mi_pil_resize_jobs_layout)
  double w[PIL_MAX_TAPS];
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = (x + xmin - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    w[x] = a < 1.0 ? 1.0 - a : 0.0;
    ww += w[x];
  }
  for (int x = 0; x < xmax; ++x) {
    double v = w[x];
    if (ww != 0.0) v /= ww;
    t->k[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << PIL_PRECISION_BITS)) : (int)(0.5 + v * (double)(1 << PIL_PRECISION_BITS));
  }
  t->x0 = xmin;
  t->n = xmax;
}

MI_HD unsigned char pil_clip8(int v) {
  v >>= PIL_PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[y][xo][0..2] from src row y (called only when nw != w0)
MI_HD void pil_h_pixel(const PilJob& j, int y, int xo, unsigned char out[3]) {
  PilTaps t;
  pil_taps(j.w0, j.nw, xo, &t);
  const unsigned char* row = j.src + (int64_t)y * j.src_ld;
  int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < t.n; ++x) {
    const int sx = j.src_hflip ? j.w0 - 1 - (t.x0 + x) : t.x0 + x;
    const unsigned char* r = row + (int64_t)sx * 3;
    s0 += (int)r[0] * t.k[x];
    s1 += (int)r[1] * t.k[x];
    s2 += (int)r[2] * t.k[x];
  }
  out[0] = pil_clip8(s0); out[1] = pil_clip8(s1); out[2] = pil_clip8(s2);
}

// detectron2 RandomSaturation / RandomBrightness = BlendTransform.apply_image on a uint8 image, numpy's dtypes exactly:
//   saturation: grey = img.dot([0.299, 0.587, 0.114]) (fp64, the channel order as stored), out = (1 - w) * grey [fp64] +
//               w * float32(img) [fp32 product, then widened]; brightness: out = w * float32(img) in fp32;
//   np.clip(out, 0, 255).astype(np.uint8) truncates.  No fused multiply-add (-ffp-contract=off).
MI_HD void pil_distort(const PilJob& j, unsigned char o[3]);
MI_HD void pil_color(const PilJob& j, unsigned char o[3]) {
  if (j.color & 1) {
    const double grey = ((double)o[0] * 0.299 + (double)o[1] * 0.587) + (double)o[2] * 0.114;
    const double g = j.sat_src * grey;
    for (int c = 0; c < 3; ++c) {
      const float u = j.sat_dst * (float)o[c];
      double v = g + (double)u;
      v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
      o[c] = (unsigned char)v;
    }
  }
  if (j.color & 2) {
    for (int c = 0; c < 3; ++c) {
      float v = j.bri_dst * (float)o[c];
      v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
      o[c] = (unsigned char)v;
    }
  }
  if (j.color & 4) pil_distort(j, o);
}

// YOLOFDistortTransform.apply_image (yolov7/data/transforms/transform.py:272-288) on one pixel: cv2.cvtColor(RGB2HSV) of the
// uint8 pixel - OpenCV's RGB2HSV_b: integer arithmetic over two 12-bit reciprocal tables, H in [0, 180), channel 0 taken as
// R whatever the image's order is -, numpy's float32 steps (/ 255., S *= dsat, V *= dexp, H += dhue * 179 / 255. with one
// wrap, * 255, clip, truncate to uint8), cv2.cvtColor(HSV2RGB) - HSV2RGB_b: the float formula on (h, s / 255, v / 255), x 255,
// rounded to nearest-even, saturated.  The reference returns these integers as a float32 image (see mosaic_paste's fsrc).
// oracle/augment_oracle.py::distort_image restates the same (cv2 itself is not installed: parity unpinned).
MI_HD unsigned char pil_trunc8(float v) {
  v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
  return (unsigned char)v;
}
MI_HD unsigned char pil_round8(float v) {
  v = rintf(v);
  return (unsigned char)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
}
MI_HD void pil_distort(const PilJob& j, unsigned char o[3]) {
  const int r = o[0], g = o[1], b = o[2];
  int v = b > g ? b : g;
  v = v > r ? v : r;
  int vmin = b < g ? b : g;
  vmin = vmin < r ? vmin : r;
  const int diff = v - vmin;
  const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
  const int sdiv = v ? (int)rint((double)(255 << 12) / (1.0 * (double)v)) : 0;          // sdiv_table[v]
  const int hdiv = diff ? (int)rint((double)(180 << 12) / (6.0 * (double)diff)) : 0;    // hdiv_table180[diff]
  const int s = (diff * sdiv + (1 << 11)) >> 12;
  int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
  h = (h * hdiv + (1 << 11)) >> 12;
  h += h < 0 ? 180 : 0;
  h = h < 0 ? 0 : (h > 255 ? 255 : h);
  // numpy, float32
  float x0 = (float)h / 255.f, x1 = (float)(s & 255) / 255.f, x2 = (float)v / 255.f;
  x1 *= j.dis_sat;
  x2 *= j.dis_exp;
  float H = x0 + j.dis_hue;
  if (j.dis_pos) { if (H > 1.0f) H -= 1.0f; }
  else { if (H < 0.0f) H += 1.0f; }
  const unsigned char h8 = pil_trunc8(H * 255.f), s8 = pil_trunc8(x1 * 255.f), v8 = pil_trunc8(x2 * 255.f);
  // HSV2RGB_b
  float hf = (float)h8;
  const float sf = (float)s8 * (1.f / 255.f), vf = (float)v8 * (1.f / 255.f);
  float bb, gg, rr;
  if (sf == 0.f) {
    bb = gg = rr = vf;
  } else {
    hf *= (6.f / 180.f);
    if (hf >= 6.f) hf -= 6.f;
    int sector = (int)floorf(hf);
    hf -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; hf = 0.f; }
    float tab[4];
    tab[0] = vf;
    tab[1] = vf * (1.f - sf);
    tab[2] = vf * (1.f - sf * hf);
    tab[3] = vf * (1.f - sf * (1.f - hf));
    const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    bb = tab[sd[sector][0]]; gg = tab[sd[sector][1]]; rr = tab[sd[sector][2]];
  }
  o[0] = pil_round8(rr * 255.f); o[1] = pil_round8(gg * 255.f); o[2] = pil_round8(bb * 255.f);
}

// destination pixel (yd, xd) of the nh x nw result after the vertical pass, HFlipTransform, VFlipTransform and
// the colour blends (pil_color) and YOLOFShiftTransform (zeros where the shifted image does not reach; transform.py:355-388),
// in that order.
// Reads the horizontally resampled image tmp [h0][nw][3], or the source itself when nw == w0.
MI_HD void pil_v_pixel(const PilJob& j, int yd, int xd, unsigned char out[3]) {
  const bool direct = j.nw == j.w0;                            // no horizontal pass: the vertical pass reads the source itself
  const unsigned char* h_img = direct ? j.src : j.tmp;
  const int64_t h_ld = direct ? j.src_ld : (int64_t)j.nw * 3;
  int ys = yd - j.shift_y, xs = xd - j.shift_x;
  if (ys < 0 || ys >= j.nh || xs < 0 || xs >= j.nw) {
    out[0] = out[1] = out[2] = 0;
    return;
  }
  if (j.vflip) ys = j.nh - 1 - ys;
  if (j.hflip) xs = j.nw - 1 - xs;
  if (direct && j.src_hflip) xs = j.w0 - 1 - xs;
  if (j.nh == j.h0) {                                          // no vertical pass: the row passes through unchanged
    const unsigned char* p = h_img + (int64_t)ys * h_ld + (int64_t)xs * 3;
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
    if (j.color) pil_color(j, out);
    return;
  }
  PilTaps t;
  pil_taps(j.h0, j.nh, ys, &t);
  int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < t.n; ++y) {
    const unsigned char* p = h_img + (int64_t)(t.x0 + y) * h_ld + (int64_t)xs * 3;
    s0 += (int)p[0] * t.k[y];
    s1 += (int)p[1] * t.k[y];
    s2 += (int)p[2] * t.k[y];
  }
  out[0] = pil_clip8(s0); out[1] = pil_clip8(s1); out[2] = pil_clip8(s2);
  if (j.color) pil_color(j, out);
}

// one thread of the two flat launches (block = 256 threads; a job's first block is blk0h / blk0v, filled by
// mi_pil_resize_jobs_layout): the kernels pass blockIdx.x / threadIdx.x, the host test build walks the same grid in loops
MI_HD void pil_h_thread(const PilJob* jobs, int njobs, int block, int thread) {
  int j = 0;
  while (j + 1 < njobs && block >= jobs[j + 1].blk0h) ++j;
  const PilJob p = jobs[j];
  if (p.nw == p.w0) return;                                    // no horizontal pass: the job owns no blocks of this launch
  const int64_t idx = ((int64_t)block - p.blk0h) * 256 + thread;
  if (idx >= (int64_t)p.h0 * p.nw) return;
  const int y = (int)(idx / p.nw), x = (int)(idx - (int64_t)y * p.nw);
  unsigned char o[3];
  pil_h_pixel(p, y, x, o);
  unsigned char* d = p.tmp + idx * 3;
  d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
}
MI_HD void pil_v_thread(const PilJob* jobs, int njobs, int block, int thread) {
  int j = 0;
  while (j + 1 < njobs && block >= jobs[j + 1].blk0v) ++j;
  const PilJob p = jobs[j];
  const int64_t idx = ((int64_t)block - p.blk0v) * 256 + thread;
  if (idx >= (int64_t)p.nh * p.nw) return;
  const int y = (int)(idx / p.nw), x = (int)(idx - (int64_t)y * p.nw);
  unsigned char o[3];
  pil_v_pixel(p, y, x, o);
  unsigned char* d = p.dst + (int64_t)y * p.dsy + (int64_t)x * p.dsx;
  d[0] = o[0]; d[p.dsc] = o[1]; d[2 * p.dsc] = o[2];
}
