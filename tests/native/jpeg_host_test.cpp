// TEST INFRASTRUCTURE (compiled by tests/test_jpeg_decode.py with g++): walks the two flat launches of the JPEG decoder -
// the thread bodies the HIP kernels call (yolov7_d2_amd/csrc/jpeg_core.h) - block by block and thread by thread on the host,
// over a job table that libmi355det.so's own host functions built (mi_jpeg_parse / _huffman / _job_fill / _jobs_layout), so
// the CPU suite can hold the whole decoder bit-identical to the installed Pillow without a GPU.  Not part of the library.
#include "../../yolov7_d2_amd/csrc/jpeg_core.h"

extern "C" int jpeg_emulate_launches(const JpegJob* jobs, int njobs, int blocks_idct, int blocks_pix) {
  for (int b = 0; b < blocks_idct; ++b)
    for (int t = 0; t < 256; ++t) mj_idct_thread(jobs, njobs, b, t);
  for (int b = 0; b < blocks_pix; ++b)
    for (int t = 0; t < 256; ++t) mj_pixel_thread(jobs, njobs, b, t);
  return 0;
}
extern "C" int jpeg_job_size() { return (int)sizeof(JpegJob); }
