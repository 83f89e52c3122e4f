#define WG_T bf16_t
#define WG_FN chap_wgrad_launch_bf16
#include "wgrad_dispatch.inc"
