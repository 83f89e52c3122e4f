#define WG_T float
#define WG_FN chap_wgrad_launch_f32
#include "wgrad_dispatch.inc"
