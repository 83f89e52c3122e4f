#define CONV_T float
#define CONV_FN chap_conv_launch_f32
#include "conv_dispatch.inc"
