// Launcher of conv_kpar_kernel (conv_kpar.h): the deep, small 3x3(x3) layers in bf16.  conv_api.hip decides eligibility.
#include "conv_kpar.h"
#include <atomic>

template <bool D3, int KC, int NT, int CPAR, bool ONE>
static int kpar_launch(const chap_conv_params* p, hipStream_t stream) {
    typedef conv_geom<3, 1, D3, D3 ? 1 : 2> G;
    const void* kern = chap_kernel<chap_conv_params, conv_kpar_kernel<bf16_t, D3, KC, NT, CPAR, ONE>, 256, 2>();
    const size_t lds = conv_kpar_lds_bytes<bf16_t, D3, KC, CPAR>(NT);
    static std::atomic<int> attr_set[16];                       // per device; setting the attribute twice (a race) is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { chap_set_error("conv(kpar): hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e)); return CHAP_ELAUNCH; }
        }
        attr_set[dev].store(1, std::memory_order_release);
    }
    const long ntiles = (long)p->N * p->D * cdiv(p->H, G::TH) * cdiv(p->W, G::TW);
    const long gx = ntiles < CHAP_STATS_MAX_SLOTS ? ntiles : CHAP_STATS_MAX_SLOTS;      // one statistics slot per block (chap_hip.h)
    const int gy = cdiv((p->Cout + 15) / 16, NT);
    return chap_launch_ptr<chap_conv_params>(kern, dim3((unsigned)gx, gy), dim3(256), lds, stream, *p, "chap_conv_fwd(kpar)");
}

template <bool D3, int KC, int CPAR>
static int kpar_one(const chap_conv_params* p, hipStream_t s) {
    if (p->nsrc == 1) return kpar_launch<D3, KC, 2, CPAR, true>(p, s);
    return kpar_launch<D3, KC, 2, CPAR, false>(p, s);
}

// KC: the K-chunk the weights were packed with (16 / 32); cpar: chunks side by side (2 / 4)
int chap_conv_launch_kpar_bf16(const chap_conv_params* p, int KC, int cpar, hipStream_t s) {
    const bool d3 = p->dims == 3;
    if (KC == 32) {
        if (cpar == 4) return d3 ? kpar_one<true, 32, 4>(p, s) : kpar_one<false, 32, 4>(p, s);
        return d3 ? kpar_one<true, 32, 2>(p, s) : kpar_one<false, 32, 2>(p, s);
    }
    if (cpar == 4) return d3 ? kpar_one<true, 16, 4>(p, s) : kpar_one<false, 16, 4>(p, s);
    return d3 ? kpar_one<true, 16, 2>(p, s) : kpar_one<false, 16, 2>(p, s);
}
