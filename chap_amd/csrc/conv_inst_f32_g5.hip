#define CONV_T float
#define CONV_GEOM 5
#define CONV_FN chap_conv_launch_f32_g5
#include "conv_dispatch.inc"
