// conv_igemm kernels with a 64-channel k-chunk (see conv_igemm_kernel.h / conv_igemm.hip)
#include "conv_igemm_kernel.h"
int conv_launch_kc64(const ConvK& k, int BN, int TPIX, size_t lds, hipStream_t s) { return conv_launch_kc<64>(k, BN, TPIX, lds, s); }
int conv_group_launch_kc64(const mi_conv_group* m, const ConvK* jobs, const int* starts, hipStream_t s) {
  return conv_group_launch_kc<64>(m, jobs, starts, s);
}
