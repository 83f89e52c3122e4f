#define CONV_T float
#define CONV_GEOM 4
#define CONV_FN chap_conv_launch_f32_g4
#include "conv_dispatch.inc"
