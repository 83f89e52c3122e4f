#define CONV_T bf16_t
#define CONV_GEOM 1
#define CONV_FN chap_conv_launch_bf16_g1
#include "conv_dispatch.inc"
