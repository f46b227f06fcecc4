// TEST INFRASTRUCTURE (compiled by tests/test_pil_resize_host.py with g++ -ffp-contract=off): runs the per-pixel functions
// the HIP kernels call (yolov7_d2_amd/csrc/pil_resize_core.h) over a whole image on the host, so the CPU suite can hold the
// arithmetic bit-identical to the installed Pillow without a GPU.  Not part of libmi355det.so.
#include "../../yolov7_d2_amd/csrc/pil_resize_core.h"

extern "C" int pil_front_host(const unsigned char* src, int h0, int w0, int nh, int nw, int hflip, int vflip, int shift_x,
                              int shift_y, unsigned char* tmp, unsigned char* dst, long long dsc, long long dsy, long long dsx) {
  PilJob j;
  j.src = src; j.tmp = tmp; j.dst = dst; j.dsc = dsc; j.dsy = dsy; j.dsx = dsx; j.src_ld = (long long)w0 * 3;
  j.h0 = h0; j.w0 = w0; j.nh = nh; j.nw = nw; j.hflip = hflip; j.vflip = vflip; j.shift_x = shift_x; j.shift_y = shift_y;
  j.src_hflip = 0; j.color = 0; j.sat_src = 0.0; j.sat_dst = j.bri_dst = 0.f;
  j.dis_hue = j.dis_sat = j.dis_exp = 0.f; j.dis_pos = 0;
  j.blk0h = j.blk0v = 0;
  if (nw != w0) {
    for (int y = 0; y < h0; ++y)
      for (int x = 0; x < nw; ++x) pil_h_pixel(j, y, x, tmp + ((long long)y * nw + x) * 3);
  }
  for (int y = 0; y < nh; ++y)
    for (int x = 0; x < nw; ++x) {
      unsigned char o[3];
      pil_v_pixel(j, y, x, o);
      for (int c = 0; c < 3; ++c) dst[c * dsc + y * dsy + x * dsx] = o[c];
    }
  return 0;
}

// the two flat launches, walked block by block and thread by thread on the host: the job table is the one the product's host
// code built and mi_pil_resize_jobs_layout (libmi355det.so, host code) laid out
extern "C" int pil_emulate_launches(const PilJob* jobs, int njobs, int blocks_h, int blocks_v) {
  for (int b = 0; b < blocks_h; ++b)
    for (int t = 0; t < 256; ++t) pil_h_thread(jobs, njobs, b, t);
  for (int b = 0; b < blocks_v; ++b)
    for (int t = 0; t < 256; ++t) pil_v_thread(jobs, njobs, b, t);
  return 0;
}
