#define CONV_T float
#define CONV_GEOM 1
#define CONV_FN chap_conv_launch_f32_g1
#include "conv_dispatch.inc"
