// explicit instantiations of the weight-stationary 3x3 convolution, MODE 2
#include "conv3x3_ws.h"
int w3_launch_mode2(const W3Launch& l, hipStream_t s) { return w3_launch_mode<2>(l, s); }
