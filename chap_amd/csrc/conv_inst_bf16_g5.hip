#define CONV_T bf16_t
#define CONV_GEOM 5
#define CONV_FN chap_conv_launch_bf16_g5
#include "conv_dispatch.inc"
