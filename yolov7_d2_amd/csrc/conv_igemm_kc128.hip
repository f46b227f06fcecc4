// conv_igemm kernels with a 128-channel k-chunk (see conv_igemm_kernel.h / conv_igemm.hip)
#include "conv_igemm_kernel.h"
int conv_launch_kc128(const ConvK& k, int BN, int TPIX, size_t lds, hipStream_t s) { return conv_launch_kc<128>(k, BN, TPIX, lds, s); }
int conv_group_launch_kc128(const mi_conv_group* m, const ConvK* jobs, const int* starts, hipStream_t s) {
  return conv_group_launch_kc<128>(m, jobs, starts, s);
}
