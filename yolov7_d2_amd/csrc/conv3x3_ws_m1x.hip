// explicit instantiations of the weight-stationary 3x3 convolution, MODE 1 with the BatchNorm + SiLU of the INPUT (XF)
#include "conv3x3_ws.h"
int w3_launch_mode1x(const W3Launch& l, hipStream_t s) { return w3_launch_mode<1, 1>(l, s); }
