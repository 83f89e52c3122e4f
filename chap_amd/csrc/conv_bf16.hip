#define CONV_T bf16_t
#define CONV_FN chap_conv_launch_bf16
#include "conv_dispatch.inc"
