// explicit instantiations of the weight-stationary 3x3 convolution, MODE 1
#include "conv3x3_ws.h"
int w3_launch_mode1(const W3Launch& l, hipStream_t s) { return w3_launch_mode<1>(l, s); }
