#define CONV_T float
#define CONV_GEOM 2
#define CONV_FN chap_conv_launch_f32_g2
#include "conv_dispatch.inc"
