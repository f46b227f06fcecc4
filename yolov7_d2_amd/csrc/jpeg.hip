// Image decoding for the input pipeline: what detectron2's utils.read_image does for the reference's
// mapper (yolov7/data/dataset_mapper.py:646-648) - PIL.Image.open -> EXIF orientation -> RGB -> BGR - for baseline JPEGs.
// Host half (this file, plain C++): marker parsing and the Huffman decoding of every scan - sequential (jdhuff.c) and
// progressive (jdphuff.c: DC / AC, first / refinement passes, end-of-band runs) - into int16 coefficient blocks (natural
// order, not dequantised).  Device half: de-quantisation + jidctint.c's ISLOW IDCT, one thread per block, then
// one thread per output pixel for fancy up-sampling (jdsample.c), YCbCr -> RGB (jdcolor.c), the EXIF transpose and the
// channel order (jpeg_core.h).  Two flat launches for a batch of images.  Arithmetic-coded / lossless / 12-bit / CMYK files
// are refused with MI_EINVAL - there is no CPU decode to fall back to.
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
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
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      info->progressive = m == 0xC2;
      MI_REQUIRE(n >= 6 && seg[0] == 8, "jpeg_parse: sample precision %d (8-bit files only)", n >= 1 ? seg[0] : -1);
      info->height = (seg[1] << 8) | seg[2];
      info->width = (seg[3] << 8) | seg[4];
      info->ncomp = seg[5];
      MI_REQUIRE((info->ncomp == 1 || info->ncomp == 3) && n >= 6 + 3 * info->ncomp, "jpeg_parse: %d components (grey or three-component files only)", info->ncomp);
      MI_REQUIRE(info->width > 0 && info->height > 0, "jpeg_parse: empty frame");
      for (int i = 0; i < info->ncomp; ++i) {
        info->comp_id[i] = seg[6 + 3 * i];
        info->hs[i] = seg[7 + 3 * i] >> 4;
        info->vs[i] = seg[7 + 3 * i] & 15;
        info->tq[i] = seg[8 + 3 * i];
        MI_REQUIRE(info->tq[i] < 4, "jpeg_parse: quantisation table index");
      }
      have_frame = true;
    } else if ((m >= 0xC3 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      MI_FAIL(MI_EINVAL, "jpeg_parse: SOF marker 0x%02X (lossless / hierarchical / arithmetic coding is not served)", m);
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
      info->sos_pos = p;               // the length field of the first SOS: mi_jpeg_huffman walks the scans from here
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
constexpr int kLook = 10;    // codes of up to kLook bits resolve in one table read (jdhuff.c's HUFF_LOOKAHEAD idea)
struct HuffTab {       // jdhuff.c jpeg_make_d_derived_tbl
  int32_t maxcode[18], valptr[17], mincode[17];
  const uint8_t* vals;
  uint16_t look[1 << kLook];   // (code length << 8) | symbol for the codes of <= kLook bits, 0 otherwise
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
  memset(t->look, 0, sizeof(t->look));
  for (int i = 0; i < n; ++i) {
    const int l = huffsize[i];
    if (l > kLook) continue;
    const int first = huffcode[i] << (kLook - l), cnt = 1 << (kLook - l);
    if (first + cnt > (1 << kLook)) return false;
    for (int q = 0; q < cnt; ++q) t->look[first + q] = (uint16_t)((l << 8) | vals[i]);
  }
  return true;
}
struct BitReader {
  const uint8_t* d;
  int64_t p, len;
  uint64_t acc;
  int n;
  void fill() {
    while (n <= 32 && p + 4 <= len && d[p] != 0xFF && d[p + 1] != 0xFF && d[p + 2] != 0xFF && d[p + 3] != 0xFF) {
      acc = (acc << 32) | ((uint64_t)d[p] << 24) | ((uint64_t)d[p + 1] << 16) | ((uint64_t)d[p + 2] << 8) | (uint64_t)d[p + 3];
      p += 4;
      n += 32;
    }
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
    if (n < 16) fill();
    const uint16_t e = t.look[(acc >> (n - kLook)) & ((1u << kLook) - 1)];
    if (e) {
      n -= e >> 8;
      return e & 255;
    }
    int l = kLook + 1;
    int code = (int)((acc >> (n - l)) & ((1u << l) - 1));
    while (l <= 16 && code > t.maxcode[l]) {
      ++l;
      code = (int)((acc >> (n - l)) & ((1u << l) - 1));
    }
    if (l > 16) { n -= 16; return 0; }
    n -= l;
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

namespace {
struct Tables {
  uint8_t bits[2][4][17], vals[2][4][256];
  bool have[2][4];
};
// one DHT segment -> the table set (tables may be redefined between scans)
bool read_dht(const uint8_t* seg, int64_t n, Tables* t) {
  int64_t q = 0;
  while (q < n) {
    if (q + 17 > n) return false;
    const int tc = seg[q] >> 4, th = seg[q] & 15;
    if (tc > 1 || th > 3) return false;
    int nv = 0;
    t->bits[tc][th][0] = 0;
    for (int i = 1; i <= 16; ++i) { t->bits[tc][th][i] = seg[q + i]; nv += seg[q + i]; }
    if (nv > 256 || q + 17 + nv > n) return false;
    for (int i = 0; i < nv; ++i) t->vals[tc][th][i] = seg[q + 17 + i];
    t->have[tc][th] = true;
    q += 17 + nv;
  }
  return true;
}
inline void refine(int16_t* c, BitReader& br, int p1, int m1) {      // jdphuff.c: correction bit of an already-nonzero coefficient
  if (br.get(1) && ((int)*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
}
}  // namespace

extern "C" int mi_jpeg_huffman(const uint8_t* data, int64_t len, const mi_jpeg_info* info, int16_t* coef) {
  MI_REQUIRE(data && info && coef && info->coef_count > 0 && info->sos_pos > 0 && info->sos_pos + 2 <= len, "jpeg_huffman: args");
  memset(coef, 0, (size_t)info->coef_count * sizeof(int16_t));
  Tables tb;
  memset(&tb, 0, sizeof(tb));
  for (int i = 0; i < 4; ++i) {
    memcpy(tb.bits[0][i], info->dc_bits[i], 17); memcpy(tb.vals[0][i], info->dc_vals[i], 256); tb.have[0][i] = info->have_dc[i];
    memcpy(tb.bits[1][i], info->ac_bits[i], 17); memcpy(tb.vals[1][i], info->ac_vals[i], 256); tb.have[1][i] = info->have_ac[i];
  }
  int dri = info->restart_interval;
  const int nc = info->ncomp, W = info->width, H = info->height;
  int64_t p = info->sos_pos;
  for (int nscan = 0; nscan < 1024; ++nscan) {
    const int64_t L = ((int64_t)data[p] << 8) | data[p + 1];
    MI_REQUIRE(L >= 6 && p + L <= len, "jpeg_huffman: truncated scan header");
    const uint8_t* seg = data + p + 2;
    const int n = seg[0];
    MI_REQUIRE(n >= 1 && n <= nc && L >= 2 + 1 + 2 * n + 3, "jpeg_huffman: scan header");
    int sci[3], std_[3], sta[3];
    for (int i = 0; i < n; ++i) {
      int ci = -1;
      for (int c = 0; c < nc; ++c) if (info->comp_id[c] == seg[1 + 2 * i]) ci = c;
      MI_REQUIRE(ci >= 0, "jpeg_huffman: the scan names a component the frame does not have");
      sci[i] = ci; std_[i] = seg[2 + 2 * i] >> 4; sta[i] = seg[2 + 2 * i] & 15;
      MI_REQUIRE(std_[i] < 4 && sta[i] < 4, "jpeg_huffman: table index");
    }
    int Ss = seg[1 + 2 * n], Se = seg[2 + 2 * n], Ah = seg[3 + 2 * n] >> 4, Al = seg[3 + 2 * n] & 15;
    if (!info->progressive) { Ss = 0; Se = 63; Ah = 0; Al = 0; }
    MI_REQUIRE(Ss <= Se && Se <= 63 && Al <= 13 && (Ss > 0 ? n == 1 : true) && (info->progressive && Ss == 0 ? Se == 0 : true),
               "jpeg_huffman: spectral selection / approximation parameters");
    HuffTab dct[3], act[3];
    for (int i = 0; i < n; ++i) {
      if (Ss == 0 && Ah == 0) {
        MI_REQUIRE(tb.have[0][std_[i]] && make_tab(tb.bits[0][std_[i]], tb.vals[0][std_[i]], &dct[i]), "jpeg_huffman: DC table %d", std_[i]);
      }
      if (Se > 0) {
        MI_REQUIRE(tb.have[1][sta[i]] && make_tab(tb.bits[1][sta[i]], tb.vals[1][sta[i]], &act[i]), "jpeg_huffman: AC table %d", sta[i]);
      }
    }
    // units of the scan: MCUs when interleaved, the component's own 8x8 blocks (not padded to the MCU grid) otherwise
    int uw, uh;
    if (n > 1) { uw = info->mcu_w; uh = info->mcu_h; }
    else {
      const int c = sci[0];
      const int dw = (int)(((int64_t)W * info->hs[c] + info->hmax - 1) / info->hmax), dh = (int)(((int64_t)H * info->vs[c] + info->vmax - 1) / info->vmax);
      uw = (dw + 7) / 8; uh = (dh + 7) / 8;
    }
    BitReader br{data, p + L, len, 0, 0};
    int pred[3] = {0, 0, 0};
    int eobrun = 0, todo = dri;
    const int p1 = 1 << Al, m1 = -(1 << Al);
    for (int uy = 0; uy < uh; ++uy)
      for (int ux = 0; ux < uw; ++ux) {
        if (dri) {
          if (todo == 0) {
            br.restart();
            pred[0] = pred[1] = pred[2] = 0;
            eobrun = 0;
            todo = dri;
          }
          --todo;
        }
        for (int i = 0; i < n; ++i) {
          const int ci = sci[i];
          const int h = n > 1 ? info->hs[ci] : 1, v = n > 1 ? info->vs[ci] : 1;
          for (int by = 0; by < v; ++by)
            for (int bx = 0; bx < h; ++bx) {
              int16_t* blk = coef + info->coef_off[ci] + ((int64_t)(uy * v + by) * info->blocks_w[ci] + (ux * h + bx)) * 64;
              if (Ss == 0) {
                if (Ah == 0) {
                  int s = br.decode(dct[i]);
                  if (s > 15) s = 15;
                  pred[ci] += s ? extend(br.get(s), s) : 0;
                  blk[0] = (int16_t)(pred[ci] * p1);
                } else if (br.get(1)) {
                  blk[0] = (int16_t)(blk[0] | p1);
                }
                if (Se == 0) continue;
              }
              int k = Ss > 1 ? Ss : 1;
              if (Ah == 0) {                                   // sequential, or an AC first pass
                if (eobrun > 0) { --eobrun; continue; }
                while (k <= Se) {
                  const int rs = br.decode(act[i]);
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz) {
                    k += r;
                    const int val = extend(br.get(sz), sz);
                    if (k <= 63) blk[kZigzag[k]] = (int16_t)(val * p1);
                    ++k;
                  } else if (r == 15) {
                    k += 16;
                  } else {
                    if (info->progressive) eobrun = (1 << r) + (r ? br.get(r) : 0) - 1;
                    break;
                  }
                }
                continue;
              }
              if (eobrun == 0) {                               // AC refinement (jdphuff.c decode_mcu_AC_refine)
                while (k <= Se) {
                  const int rs = br.decode(act[i]);
                  int r = rs >> 4, sz = rs & 15;
                  if (sz) {
                    sz = br.get(1) ? p1 : m1;
                  } else if (r != 15) {
                    eobrun = (1 << r) + (r ? br.get(r) : 0);
                    break;
                  }
                  while (k <= Se) {
                    int16_t* c = blk + kZigzag[k];
                    if (*c != 0) refine(c, br, p1, m1);
                    else if (--r < 0) break;
                    ++k;
                  }
                  if (sz && k <= 63) blk[kZigzag[k]] = (int16_t)sz;
                  ++k;
                }
              }
              if (eobrun > 0) {
                for (; k <= Se; ++k) {
                  int16_t* c = blk + kZigzag[k];
                  if (*c != 0) refine(c, br, p1, m1);
                }
                --eobrun;
              }
            }
        }
      }
    // the marker segments after the scan: tables and the restart interval may be redefined between scans
    p = br.p;
    while (true) {
      while (p + 1 < len && !(data[p] == 0xFF && data[p + 1] != 0x00 && data[p + 1] != 0xFF && !(data[p + 1] >= 0xD0 && data[p + 1] <= 0xD7))) ++p;
      if (p + 1 >= len) return MI_OK;                          // no EOI: what was decoded stands (as libjpeg's premature-end warning)
      const int m = data[p + 1];
      p += 2;
      if (m == 0xD9) return MI_OK;
      MI_REQUIRE(p + 2 <= len, "jpeg_huffman: truncated segment");
      const int64_t L2 = ((int64_t)data[p] << 8) | data[p + 1];
      MI_REQUIRE(L2 >= 2 && p + L2 <= len, "jpeg_huffman: truncated segment");
      if (m == 0xDA) break;
      if (m == 0xC4) MI_REQUIRE(read_dht(data + p + 2, L2 - 2, &tb), "jpeg_huffman: Huffman table between scans");
      else if (m == 0xDD && L2 >= 4) dri = (data[p + 2] << 8) | data[p + 3];
      p += L2;
    }
  }
  MI_FAIL(MI_EINVAL, "jpeg_huffman: more than 1024 scans");
}

// a batch of files on `threads` host threads (the library's own: no Python thread per file, no GIL hand-offs); rcs[k] receives
// file k's return code, the call returns the first non-zero one
extern "C" int mi_jpeg_huffman_batch(const uint8_t* const* datas, const int64_t* lens, const mi_jpeg_info* infos, int16_t* const* coefs,
                                     int n, int threads, int32_t* rcs) {
  MI_REQUIRE(datas && lens && infos && coefs && rcs && n > 0, "jpeg_huffman_batch: args");
  if (threads < 1) threads = 1;
  if (threads > n) threads = n;
  std::atomic<int> next(0);
  auto work = [&]() {
    for (int k = next.fetch_add(1); k < n; k = next.fetch_add(1)) rcs[k] = mi_jpeg_huffman(datas[k], lens[k], &infos[k], coefs[k]);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  for (int k = 0; k < n; ++k)
    if (rcs[k] != MI_OK) return rcs[k];
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
