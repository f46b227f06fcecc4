// explicit instantiations of the streaming 1x1 convolution, MODE 3 (convolution + BatchNorm + activation in one launch)
#include "conv1x1_stream.h"
int c1s_launch_mode3(const C1Launch& l, hipStream_t s) {
  switch (l.K) {
    case 32: return c1s_launch_k<32, 3>(l, s);
    case 64: return c1s_launch_k<64, 3>(l, s);
    case 128: return c1s_launch_k<128, 3>(l, s);
    case 256: return c1s_launch_k<256, 3>(l, s);
    case 512: return c1s_launch_k<512, 3>(l, s);
  }
  MI_FAIL(MI_EINVAL, "conv1x1_stream: K %d", l.K);
}
// word 2 of the barrier record: set when a block's wait timed out (never in a healthy run)
int c1s_bar_status(unsigned* flag) {
  unsigned w[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_c1_bar), sizeof(w), 0, hipMemcpyDeviceToHost) != hipSuccess) MI_FAIL(MI_ELAUNCH, "conv1x1_stream: barrier status");
  *flag = w[2];
  return MI_OK;
}
