// Image decoding for the input pipeline: what detectron2's utils.read_image does for the reference's
// mapper (yolov7/data/dataset_mapper.py:646-648) - PIL.Image.open -> EXIF orientation -> RGB -> BGR - for baseline JPEGs.
// Host half (this file, plain C++): marker parsing and the sequential Huffman decoding of jdhuff.c into int16 coefficient
// blocks (natural order, not dequantised).  Device half: de-quantisation + jidctint.c's ISLOW IDCT, one thread per block, then
// one thread per output pixel for fancy up-sampling (jdsample.c), YCbCr -> RGB (jdcolor.c), the EXIF transpose and the
// channel order (jpeg_core.h).  Two flat launches for a batch of images.  Progressive / arithmetic-coded / 12-bit / CMYK
// files are refused with MI_EINVAL - there is no CPU decode to fall back to.
#include <string.h>
#include "common.h"
#include "jpeg_core.h"
static_assert(sizeof(JpegJob) == sizeof(mi_jpeg_job), "JpegJob mirrors mi_jpeg_job");

static const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27,
                                          20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                          58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static int exif_orientation(const uint8_t* t, int64_t n) {
  if (n < 8) return 1;
  const bool le = t[0] == 'I' && t[1] == 'I';
  auto u16 = [&](int64_t o) -> uint32_t { return le ? (uint32_t)t[o] | ((uint32_t)t[o + 1] << 8) : ((uint32_t)t[o] << 8) | t[o + 1]; };
  auto u32 = [&](int64_t o) -> uint32_t { return le ? u16(o) | (u16(o + 2) << 16) : (u16(o) << 16) | u16(o + 2); };
  const int64_t ifd = u32(4);
  if (ifd + 2 > n) return 1;
  const int cnt = (int)u16(ifd);
  for (int k = 0; k < cnt; ++k) {
    const int64_t e = ifd + 2 + 12 * (int64_t)k;
    if (e + 12 > n) break;
    if (u16(e) == 0x0112) {
      const int o = (int)u16(e + 8);
      return (o >= 1 && o <= 8) ? o : 1;
    }
  }
  return 1;
}

extern "C" int mi_jpeg_parse(const uint8_t* b, int64_t len, mi_jpeg_info* info) {
  MI_REQUIRE(b && info && len >= 4, "jpeg_parse: args");
  MI_REQUIRE(b[0] == 0xFF && b[1] == 0xD8, "jpeg_parse: not a JPEG (no SOI)");
  memset(info, 0, sizeof(*info));
  info->orientation = 1;
  info->adobe_transform = -1;
  int comp_id[3] = {0, 0, 0};
  int64_t p = 2;
  bool have_frame = false;
  while (true) {
    while (p < len && b[p] != 0xFF) ++p;
    while (p < len && b[p] == 0xFF) ++p;
    MI_REQUIRE(p < len, "jpeg_parse: ran off the end before a scan");
    const int m = b[p++];
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    MI_REQUIRE(m != 0xD9, "jpeg_parse: EOI before a scan");
    MI_REQUIRE(p + 2 <= len, "jpeg_parse: truncated segment");
    const int64_t L = ((int64_t)b[p] << 8) | b[p + 1];
    MI_REQUIRE(L >= 2 && p + L <= len, "jpeg_parse: truncated segment");
    const uint8_t* seg = b + p + 2;
    const int64_t n = L - 2;
    if (m == 0xDB) {
      int64_t q = 0;
      while (q < n) {
        const int pq = seg[q] >> 4, tq = seg[q] & 15;
        MI_REQUIRE(pq == 0 && tq < 4 && q + 65 <= n, "jpeg_parse: quantisation table (16-bit tables are not served)");
        for (int i = 0; i < 64; ++i) info->qt[tq][kZigzag[i]] = seg[q + 1 + i];
        info->have_qt[tq] = 1;
        q += 65;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      MI_REQUIRE(n >= 6 && seg[0] == 8, "jpeg_parse: sample precision %d (8-bit files only)", n >= 1 ? seg[0] : -1);
      info->height = (seg[1] << 8) | seg[2];
      info->width = (seg[3] << 8) | seg[4];
      info->ncomp = seg[5];
      MI_REQUIRE((info->ncomp == 1 || info->ncomp == 3) && n >= 6 + 3 * info->ncomp, "jpeg_parse: %d components (grey or three-component files only)", info->ncomp);
      MI_REQUIRE(info->width > 0 && info->height > 0, "jpeg_parse: empty frame");
      for (int i = 0; i < info->ncomp; ++i) {
        comp_id[i] = seg[6 + 3 * i];
        info->hs[i] = seg[7 + 3 * i] >> 4;
        info->vs[i] = seg[7 + 3 * i] & 15;
        info->tq[i] = seg[8 + 3 * i];
        MI_REQUIRE(info->tq[i] < 4, "jpeg_parse: quantisation table index");
      }
      have_frame = true;
    } else if ((m >= 0xC2 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      MI_FAIL(MI_EINVAL, "jpeg_parse: SOF marker 0x%02X (progressive / lossless / arithmetic coding is not served)", m);
    } else if (m == 0xC4) {
      int64_t q = 0;
      while (q < n) {
        MI_REQUIRE(q + 17 <= n, "jpeg_parse: truncated Huffman table");
        const int tc = seg[q] >> 4, th = seg[q] & 15;
        MI_REQUIRE(tc < 2 && th < 4, "jpeg_parse: Huffman table id");
        uint8_t* bits = tc ? info->ac_bits[th] : info->dc_bits[th];
        uint8_t* vals = tc ? info->ac_vals[th] : info->dc_vals[th];
        int nv = 0;
        bits[0] = 0;
        for (int i = 1; i <= 16; ++i) { bits[i] = seg[q + i]; nv += bits[i]; }
        MI_REQUIRE(nv <= 256 && q + 17 + nv <= n, "jpeg_parse: Huffman table size");
        for (int i = 0; i < nv; ++i) vals[i] = seg[q + 17 + i];
        (tc ? info->have_ac : info->have_dc)[th] = 1;
        q += 17 + nv;
      }
    } else if (m == 0xDD) {
      MI_REQUIRE(n >= 2, "jpeg_parse: DRI");
      info->restart_interval = (seg[0] << 8) | seg[1];
    } else if (m == 0xE1 && n >= 6 && memcmp(seg, "Exif\0\0", 6) == 0) {
      info->orientation = exif_orientation(seg + 6, n - 6);
    } else if (m == 0xEE && n >= 12 && memcmp(seg, "Adobe", 5) == 0) {
      info->adobe_transform = seg[11];
    } else if (m == 0xDA) {
      MI_REQUIRE(have_frame, "jpeg_parse: scan before the frame header");
      MI_REQUIRE(n >= 1 && seg[0] == info->ncomp && n >= 1 + 2 * info->ncomp, "jpeg_parse: non-interleaved scans are not served");
      for (int i = 0; i < info->ncomp; ++i) {
        MI_REQUIRE(seg[1 + 2 * i] == comp_id[i], "jpeg_parse: scan component order");
        info->td[i] = seg[2 + 2 * i] >> 4;
        info->ta[i] = seg[2 + 2 * i] & 15;
        MI_REQUIRE(info->td[i] < 4 && info->ta[i] < 4 && info->have_dc[info->td[i]] && info->have_ac[info->ta[i]] && info->have_qt[info->tq[i]],
                   "jpeg_parse: the scan names a table the file does not define");
      }
      info->scan_start = p + L;
      break;
    }
    p += L;
  }
  int hmax = 1, vmax = 1;
  for (int i = 0; i < info->ncomp; ++i) {
    MI_REQUIRE(info->hs[i] >= 1 && info->hs[i] <= 2 && info->vs[i] >= 1 && info->vs[i] <= 2, "jpeg_parse: sampling factors %dx%d", info->hs[i], info->vs[i]);
    if (info->hs[i] > hmax) hmax = info->hs[i];
    if (info->vs[i] > vmax) vmax = info->vs[i];
  }
  if (info->ncomp == 1) { info->hs[0] = info->vs[0] = 1; hmax = vmax = 1; }     // a single-component scan is not interleaved: 8x8 MCUs
  for (int i = 0; i < info->ncomp; ++i)
    MI_REQUIRE(hmax % info->hs[i] == 0 && vmax % info->vs[i] == 0, "jpeg_parse: fractional sampling ratios");
  info->hmax = hmax; info->vmax = vmax;
  info->mcu_w = (info->width + 8 * hmax - 1) / (8 * hmax);
  info->mcu_h = (info->height + 8 * vmax - 1) / (8 * vmax);
  int64_t off = 0;
  for (int i = 0; i < info->ncomp; ++i) {
    info->blocks_w[i] = info->mcu_w * info->hs[i];
    info->blocks_h[i] = info->mcu_h * info->vs[i];
    info->coef_off[i] = off;
    off += (int64_t)info->blocks_w[i] * info->blocks_h[i] * 64;
  }
  info->coef_count = off;
  return MI_OK;
}

namespace {
struct HuffTab {       // jdhuff.c jpeg_make_d_derived_tbl
  int32_t maxcode[18], valptr[17], mincode[17];
  const uint8_t* vals;
};
bool make_tab(const uint8_t* bits, const uint8_t* vals, HuffTab* t) {
  int huffsize[257], huffcode[257], n = 0;
  for (int l = 1; l <= 16; ++l)
    for (int i = 0; i < bits[l]; ++i) { if (n >= 256) return false; huffsize[n++] = l; }
  int code = 0, si = n ? huffsize[0] : 0, k = 0;
  while (k < n) {
    while (k < n && huffsize[k] == si) huffcode[k++] = code++;
    code <<= 1;
    ++si;
  }
  k = 0;
  for (int l = 1; l <= 16; ++l) {
    t->maxcode[l] = -1; t->valptr[l] = 0; t->mincode[l] = 0;
    if (bits[l]) {
      t->valptr[l] = k;
      t->mincode[l] = huffcode[k];
      k += bits[l];
      t->maxcode[l] = huffcode[k - 1];
    }
  }
  t->maxcode[17] = 0xFFFFF;
  t->vals = vals;
  return true;
}
struct BitReader {
  const uint8_t* d;
  int64_t p, len;
  uint64_t acc;
  int n;
  void fill() {
    while (n <= 48) {
      int c = p < len ? d[p] : 0;
      if (c == 0xFF) {
        const int nx = p + 1 < len ? d[p + 1] : 0xD9;
        if (nx == 0) p += 2;
        else c = 0;                  // a marker: feed zeros, do not advance (jdhuff.c does the same once it has hit one)
      } else {
        ++p;
      }
      acc = (acc << 8) | (uint64_t)c;
      n += 8;
    }
  }
  int get(int k) {
    if (k == 0) return 0;
    if (n < k) fill();
    n -= k;
    return (int)((acc >> n) & ((1u << k) - 1));
  }
  int decode(const HuffTab& t) {
    int code = get(1), l = 1;
    while (code > t.maxcode[l]) { code = (code << 1) | get(1); ++l; }
    if (l > 16) return 0;
    return t.vals[(t.valptr[l] + code - t.mincode[l]) & 255];
  }
  void restart() {                   // byte-align, skip the RSTn marker
    acc = 0; n = 0;
    while (p + 1 < len && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) ++p;
    p += 2;
  }
};
inline int extend(int v, int s) { return v >= (1 << (s - 1)) ? v : v - (1 << s) + 1; }
}  // namespace

extern "C" int mi_jpeg_huffman(const uint8_t* data, int64_t len, const mi_jpeg_info* info, int16_t* coef) {
  MI_REQUIRE(data && info && coef && info->coef_count > 0 && info->scan_start > 0 && info->scan_start <= len, "jpeg_huffman: args");
  memset(coef, 0, (size_t)info->coef_count * sizeof(int16_t));
  HuffTab dc[4], ac[4];
  for (int i = 0; i < 4; ++i) {
    if (info->have_dc[i]) MI_REQUIRE(make_tab(info->dc_bits[i], info->dc_vals[i], &dc[i]), "jpeg_huffman: DC table %d", i);
    if (info->have_ac[i]) MI_REQUIRE(make_tab(info->ac_bits[i], info->ac_vals[i], &ac[i]), "jpeg_huffman: AC table %d", i);
  }
  BitReader br{data, info->scan_start, len, 0, 0};
  int pred[3] = {0, 0, 0};
  const int dri = info->restart_interval;
  int todo = dri;
  for (int my = 0; my < info->mcu_h; ++my)
    for (int mx = 0; mx < info->mcu_w; ++mx) {
      if (dri) {
        if (todo == 0) {
          br.restart();
          pred[0] = pred[1] = pred[2] = 0;
          todo = dri;
        }
        --todo;
      }
      for (int ci = 0; ci < info->ncomp; ++ci) {
        const int h = info->hs[ci], v = info->vs[ci];
        for (int by = 0; by < v; ++by)
          for (int bx = 0; bx < h; ++bx) {
            int16_t* blk = coef + info->coef_off[ci] + ((int64_t)(my * v + by) * info->blocks_w[ci] + (mx * h + bx)) * 64;
            int s = br.decode(dc[info->td[ci]]);
            if (s > 15) s = 15;
            pred[ci] += s ? extend(br.get(s), s) : 0;
            blk[0] = (int16_t)pred[ci];
            int k = 1;
            while (k < 64) {
              const int rs = br.decode(ac[info->ta[ci]]);
              const int r = rs >> 4, sz = rs & 15;
              if (sz) {
                k += r;
                const int val = extend(br.get(sz), sz);
                if (k < 64) blk[kZigzag[k]] = (int16_t)val;
                ++k;
              } else {
                if (r != 15) break;
                k += 16;
              }
            }
          }
      }
    }
  return MI_OK;
}

// ---------------------------------------------------------------- device half
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JpegJob* __restrict__ jobs, int njobs) {
  mj_idct_thread(jobs, njobs, (int)blockIdx.x, (int)threadIdx.x);
}
__global__ __launch_bounds__(256) void jpeg_pixel_kernel(const JpegJob* __restrict__ jobs, int njobs) {
  mj_pixel_thread(jobs, njobs, (int)blockIdx.x, (int)threadIdx.x);
}

extern "C" int mi_jpeg_job_fill(const mi_jpeg_info* info, const void* coef_dev, void* planes_dev, void* out_dev, int bgr,
                                int apply_orientation, mi_jpeg_job* job) {
  MI_REQUIRE(info && job && coef_dev && planes_dev && out_dev && info->coef_count > 0, "jpeg_job_fill: args");
  memset(job, 0, sizeof(*job));
  job->coef = (const int16_t*)coef_dev;
  job->planes = (unsigned char*)planes_dev;
  job->out = (unsigned char*)out_dev;
  job->width = info->width; job->height = info->height; job->ncomp = info->ncomp;
  job->orientation = apply_orientation ? info->orientation : 1;
  job->bgr = bgr ? 1 : 0;
  job->ycc = (info->ncomp == 3 && info->adobe_transform != 0) ? 1 : 0;
  job->hmax = info->hmax; job->vmax = info->vmax;
  int64_t poff = 0;
  for (int c = 0; c < info->ncomp; ++c) {
    job->hs[c] = info->hs[c]; job->vs[c] = info->vs[c];
    job->blocks_w[c] = info->blocks_w[c]; job->blocks_h[c] = info->blocks_h[c];
    job->coef_off[c] = info->coef_off[c];
    job->plane_off[c] = poff;
    poff += (int64_t)info->blocks_w[c] * info->blocks_h[c] * 64;
    for (int i = 0; i < 64; ++i) job->qt[c][i] = info->qt[info->tq[c]][i];
  }
  return MI_OK;
}
extern "C" int mi_jpeg_jobs_layout(mi_jpeg_job* jobs, int n, int32_t* blocks_idct, int32_t* blocks_pix) {
  MI_REQUIRE(jobs && n > 0 && blocks_idct && blocks_pix, "jpeg_jobs_layout: args");
  int64_t bi = 0, bp = 0;
  for (int i = 0; i < n; ++i) {
    mi_jpeg_job& j = jobs[i];
    MI_REQUIRE(j.coef && j.planes && j.out && j.width > 0 && j.height > 0 && (j.ncomp == 1 || j.ncomp == 3), "jpeg_jobs_layout: job %d", i);
    int64_t nb = 0;
    for (int c = 0; c < j.ncomp; ++c) {
      MI_REQUIRE(j.hs[c] >= 1 && j.vs[c] >= 1 && j.hmax % j.hs[c] == 0 && j.vmax % j.vs[c] == 0 && j.hmax <= 2 && j.vmax <= 2,
                 "jpeg_jobs_layout: job %d sampling factors", i);
      nb += (int64_t)j.blocks_w[c] * j.blocks_h[c];
    }
    j.blk0_idct = (int32_t)bi;
    j.blk0_pix = (int32_t)bp;
    bi += (nb + 255) / 256;
    bp += ((int64_t)j.width * j.height + 255) / 256;
    MI_REQUIRE(bi < (1LL << 30) && bp < (1LL << 30), "jpeg_jobs_layout: too many blocks");
  }
  *blocks_idct = (int32_t)bi;
  *blocks_pix = (int32_t)bp;
  return MI_OK;
}
extern "C" int mi_jpeg_idct(const mi_jpeg_job* jobs_dev, int n, int total_blocks, mi_stream_t st) {
  MI_REQUIRE(jobs_dev && n > 0 && total_blocks > 0, "jpeg_idct: args");
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)st, (const JpegJob*)jobs_dev, n);
  MI_CHECK_LAUNCH("jpeg_idct");
  return MI_OK;
}
extern "C" int mi_jpeg_color(const mi_jpeg_job* jobs_dev, int n, int total_blocks, mi_stream_t st) {
  MI_REQUIRE(jobs_dev && n > 0 && total_blocks > 0, "jpeg_color: args");
  hipLaunchKernelGGL(jpeg_pixel_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)st, (const JpegJob*)jobs_dev, n);
  MI_CHECK_LAUNCH("jpeg_color");
  return MI_OK;
}
