#define CONV_T float
#define CONV_GEOM 3
#define CONV_FN chap_conv_launch_f32_g3
#include "conv_dispatch.inc"
