// explicit instantiations of the streaming 1x1 convolution, MODE 6 (one translation unit per mode: they compile in parallel)
#include "conv1x1_stream.h"
int c1s_launch_mode6(const C1Launch& l, hipStream_t s) {
  switch (l.K) {
    case 32: return c1s_launch_k<32, 6>(l, s);
    case 64: return c1s_launch_k<64, 6>(l, s);
    case 128: return c1s_launch_k<128, 6>(l, s);
    case 256: return c1s_launch_k<256, 6>(l, s);
    case 512: return c1s_launch_k<512, 6>(l, s);
  }
  MI_FAIL(MI_EINVAL, "conv1x1_stream: K %d", l.K);
}
