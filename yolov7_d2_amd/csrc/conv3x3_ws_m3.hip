// explicit instantiations of the weight-stationary 3x3 convolution, MODE 3 (convolution + BatchNorm + activation in one launch)
#include "conv3x3_ws.h"
int w3_launch_mode3(const W3Launch& l, hipStream_t s) { return w3_launch_mode<3>(l, s); }
// resident MODE 3 blocks per CU for K channels and `lds` bytes of dynamic LDS (the grid barrier needs the whole grid resident)
int w3_mode3_blocks_per_cu(int K, int lds) {
  int n = 0;
  hipError_t e = hipErrorInvalidValue;
  switch (K) {
    case 128: {
      auto fn = w3_kernel<128, 4, 1, 4, 3>;
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, (size_t)lds);
      break;
    }
    case 64: {
      auto fn = w3_kernel<64, 2, 2, 2, 3>;
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, (size_t)lds);
      break;
    }
    case 32: {
      auto fn = w3_kernel<32, 1, 4, 1, 3>;
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, (size_t)lds);
      break;
    }
  }
  return e == hipSuccess ? n : 0;
}
int w3_bar_status(unsigned* flag) {
  unsigned w[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_w3_bar), sizeof(w), 0, hipMemcpyDeviceToHost) != hipSuccess) MI_FAIL(MI_ELAUNCH, "conv3x3_ws: barrier status");
  *flag = w[2];
  return MI_OK;
}
