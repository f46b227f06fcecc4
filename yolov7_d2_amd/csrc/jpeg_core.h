// Baseline JPEG decoding as the reference's mapper gets it: detectron2 utils.read_image(file, "BGR") = PIL.Image.open ->
// EXIF orientation -> convert("RGB") -> [:, :, ::-1] (yolov7/data/dataset_mapper.py:646-648; d2 un-vendored), i.e. Pillow's
// libjpeg(-turbo) defaults: JDCT_ISLOW (jidctint.c), fancy up-sampling (jdsample.c), YCbCr -> RGB (jdcolor.c).
// The entropy decoding is sequential and runs on the host (jpeg_host.cpp); everything per block / per pixel is here, as
// plain C++ with no HIP types: the kernels in jpeg.hip call these on the device, tests/native/jpeg_host_test.cpp compiles the
// SAME functions for the host and the CPU suite holds them bit-identical to the installed Pillow.  Integer arithmetic only.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MJ_HD __host__ __device__ __forceinline__
#else
#define MJ_HD static inline
#endif

struct JpegJob {   // mirrors mi_jpeg_job (include/mi355_det.h)
  const int16_t* coef;       // this image's coefficient blocks, natural order, component after component
  unsigned char* planes;     // scratch: the components' sample planes [blocks_h * 8][blocks_w * 8]
  unsigned char* out;        // HWC uint8 [oh][ow][3] (after the EXIF orientation)
  int64_t coef_off[3], plane_off[3];
  int32_t width, height, ncomp, orientation, bgr, ycc;
  int32_t hs[3], vs[3], blocks_w[3], blocks_h[3];
  int32_t hmax, vmax;
  int32_t blk0_idct, blk0_pix;
  uint16_t qt[3][64];
};

// ---- jidctint.c, 8x8, CONST_BITS 13, PASS1_BITS 2
#define MJ_DESCALE(x, n) (((x) + ((int64_t)1 << ((n) - 1))) >> (n))
MJ_HD void mj_idct_1d(const int64_t d[8], int shift, int64_t o[8]) {
  int64_t z2 = d[2], z3 = d[6];
  int64_t z1 = (z2 + z3) * 4433;
  const int64_t tmp2 = z1 + z3 * (-15137);
  const int64_t tmp3 = z1 + z2 * 6270;
  z2 = d[0]; z3 = d[4];
  const int64_t tmp0 = (z2 + z3) << 13;
  const int64_t tmp1 = (z2 - z3) << 13;
  const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  int64_t t0 = d[7], t1 = d[5], t2 = d[3], t3 = d[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
  int64_t z4 = t1 + t3;
  const int64_t z5 = (z3 + z4) * 9633;
  t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
  z1 *= -7373; z2 *= -20995; z3 = z3 * (-16069) + z5; z4 = z4 * (-3196) + z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  o[0] = MJ_DESCALE(tmp10 + t3, shift); o[7] = MJ_DESCALE(tmp10 - t3, shift);
  o[1] = MJ_DESCALE(tmp11 + t2, shift); o[6] = MJ_DESCALE(tmp11 - t2, shift);
  o[2] = MJ_DESCALE(tmp12 + t1, shift); o[5] = MJ_DESCALE(tmp12 - t1, shift);
  o[3] = MJ_DESCALE(tmp13 + t0, shift); o[4] = MJ_DESCALE(tmp13 - t0, shift);
}
// one block: coefficients (natural order) x quantisation table -> 8 rows of 8 samples at out (row stride in bytes)
MJ_HD void mj_idct_block(const int16_t* c, const uint16_t* qt, unsigned char* out, int64_t stride) {
  int32_t ws[64];
  for (int x = 0; x < 8; ++x) {            // pass 1: columns
    int64_t d[8], o[8];
    for (int y = 0; y < 8; ++y) d[y] = (int64_t)c[y * 8 + x] * (int64_t)qt[y * 8 + x];
    mj_idct_1d(d, 13 - 2, o);
    for (int y = 0; y < 8; ++y) ws[y * 8 + x] = (int32_t)o[y];
  }
  for (int y = 0; y < 8; ++y) {            // pass 2: rows
    int64_t d[8], o[8];
    for (int x = 0; x < 8; ++x) d[x] = ws[y * 8 + x];
    mj_idct_1d(d, 13 + 2 + 3, o);
    for (int x = 0; x < 8; ++x) {
      const int64_t v = o[x] + 128;       // range_limit: clamp (files whose samples leave [-384, 639] are corrupt)
      out[y * stride + x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

// ---- jdsample.c: the full-resolution sample of component c at (sy, sx)
MJ_HD int mj_sample(const JpegJob& j, int c, int sy, int sx) {
  const unsigned char* p = j.planes + j.plane_off[c];
  const int64_t ld = (int64_t)j.blocks_w[c] * 8;
  const int hx = j.hmax / j.hs[c], vx = j.vmax / j.vs[c];
  if (hx == 1 && vx == 1) return p[(int64_t)sy * ld + sx];
  const int dw = (int)(((int64_t)j.width * j.hs[c] + j.hmax - 1) / j.hmax);      // downsampled_width / _height
  const int dh = (int)(((int64_t)j.height * j.vs[c] + j.vmax - 1) / j.vmax);
  if (hx == 2 && vx == 1) {
    const unsigned char* r = p + (int64_t)sy * ld;
    const int i = sx >> 1;
    if (dw <= 2) return r[i];
    if (sx & 1) return i == dw - 1 ? r[i] : (3 * r[i] + r[i + 1] + 2) >> 2;
    return i == 0 ? r[0] : (3 * r[i] + r[i - 1] + 1) >> 2;
  }
  const int r0 = sy >> 1, v = sy & 1;
  int rn = v ? r0 + 1 : r0 - 1;                                 // the nearer neighbour row, replicated at the image edges
  if (rn < 0) rn = 0;
  if (rn > dh - 1) rn = dh - 1;
  const unsigned char* a = p + (int64_t)r0 * ld;
  const unsigned char* b = p + (int64_t)rn * ld;
  if (hx == 1 && vx == 2) return (3 * a[sx] + b[sx] + (v ? 2 : 1)) >> 2;
  // h2v2
  const int i = sx >> 1;
  if (dw <= 2) return a[i];
  const int cs = 3 * a[i] + b[i];
  if (sx & 1) return i == dw - 1 ? (cs * 4 + 7) >> 4 : (cs * 3 + 3 * a[i + 1] + b[i + 1] + 7) >> 4;
  return i == 0 ? (cs * 4 + 8) >> 4 : (cs * 3 + 3 * a[i - 1] + b[i - 1] + 8) >> 4;
}

MJ_HD unsigned char mj_clamp(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// output pixel (oy, ox) of the oriented image: up-sampling, jdcolor.c's YCbCr -> RGB (16-bit fixed point), channel order
MJ_HD void mj_pixel(const JpegJob& j, int oy, int ox, unsigned char out[3]) {
  const int W = j.width, H = j.height;
  int sy, sx;
  switch (j.orientation) {                                     // PIL.ImageOps.exif_transpose
    case 2: sy = oy; sx = W - 1 - ox; break;
    case 3: sy = H - 1 - oy; sx = W - 1 - ox; break;
    case 4: sy = H - 1 - oy; sx = ox; break;
    case 5: sy = ox; sx = oy; break;
    case 6: sy = H - 1 - ox; sx = oy; break;
    case 7: sy = H - 1 - ox; sx = W - 1 - oy; break;
    case 8: sy = ox; sx = W - 1 - oy; break;
    default: sy = oy; sx = ox; break;
  }
  int r, g, b;
  if (j.ncomp == 1) {
    r = g = b = mj_sample(j, 0, sy, sx);
  } else {
    const int y = mj_sample(j, 0, sy, sx), c1 = mj_sample(j, 1, sy, sx), c2 = mj_sample(j, 2, sy, sx);
    if (j.ycc) {
      const int cb = c1 - 128, cr = c2 - 128;
      r = mj_clamp(y + ((91881 * cr + 32768) >> 16));
      g = mj_clamp(y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
      b = mj_clamp(y + ((116130 * cb + 32768) >> 16));
    } else {
      r = y; g = c1; b = c2;
    }
  }
  if (j.bgr) { out[0] = (unsigned char)b; out[1] = (unsigned char)g; out[2] = (unsigned char)r; }
  else       { out[0] = (unsigned char)r; out[1] = (unsigned char)g; out[2] = (unsigned char)b; }
}

// ---- one thread of the two flat launches (256 threads per block; a job's first block: blk0_idct / blk0_pix)
MJ_HD void mj_idct_thread(const JpegJob* jobs, int njobs, int block, int thread) {
  int k = 0;
  while (k + 1 < njobs && block >= jobs[k + 1].blk0_idct) ++k;
  const JpegJob& j = jobs[k];
  int64_t idx = ((int64_t)block - j.blk0_idct) * 256 + thread;      // block index within the image, component after component
  for (int c = 0; c < j.ncomp; ++c) {
    const int64_t nb = (int64_t)j.blocks_w[c] * j.blocks_h[c];
    if (idx < nb) {
      const int by = (int)(idx / j.blocks_w[c]), bx = (int)(idx - (int64_t)by * j.blocks_w[c]);
      const int64_t ld = (int64_t)j.blocks_w[c] * 8;
      mj_idct_block(j.coef + j.coef_off[c] + idx * 64, j.qt[c], j.planes + j.plane_off[c] + (int64_t)by * 8 * ld + bx * 8, ld);
      return;
    }
    idx -= nb;
  }
}
MJ_HD void mj_pixel_thread(const JpegJob* jobs, int njobs, int block, int thread) {
  int k = 0;
  while (k + 1 < njobs && block >= jobs[k + 1].blk0_pix) ++k;
  const JpegJob& j = jobs[k];
  const int transposed = j.orientation >= 5 && j.orientation <= 8;
  const int ow = transposed ? j.height : j.width, oh = transposed ? j.width : j.height;
  const int64_t idx = ((int64_t)block - j.blk0_pix) * 256 + thread;
  if (idx >= (int64_t)ow * oh) return;
  const int oy = (int)(idx / ow), ox = (int)(idx - (int64_t)oy * ow);
  unsigned char o[3];
  mj_pixel(j, oy, ox, o);
  unsigned char* d = j.out + idx * 3;
  d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
}
