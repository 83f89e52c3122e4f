// Launcher of conv_kpar_kernel (conv_kpar.h): the deep, small 3x3(x3) layers in bf16.  conv_api.hip decides eligibility.
#include "conv_kpar.h"
#include "conv_kpar2d.h"
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

// 2D, 32-channel chunks: the round-4 kernel (conv_kpar2d.h)
template <int CPAR, bool ONE, bool KEEPM, bool SINGLE>
static int kpar2d_launch(const chap_conv_params* p, hipStream_t stream) {
    typedef conv_geom<3, 1, false, 2> G;
    constexpr int NT = 2;
    const void* kern = chap_kernel<chap_conv_params, conv_kpar2d_kernel<NT, CPAR, ONE, KEEPM, SINGLE>, 256, 2>();
    const size_t lds = conv_kpar_lds_bytes<bf16_t, false, 32, CPAR>(NT);
    static std::atomic<int> attr_set[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { chap_set_error("conv(kpar2d): hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e)); return CHAP_ELAUNCH; }
        }
        attr_set[dev].store(1, std::memory_order_release);
    }
    const long ntiles = (long)p->N * p->D * cdiv(p->H, G::TH) * cdiv(p->W, G::TW);
    const long gx = ntiles < CHAP_STATS_MAX_SLOTS ? ntiles : CHAP_STATS_MAX_SLOTS;
    const int gy = cdiv((p->Cout + 15) / 16, NT);
    return chap_launch_ptr<chap_conv_params>(kern, dim3((unsigned)gx, gy), dim3(256), lds, stream, *p, "chap_conv_fwd(kpar2d)");
}
template <int CPAR>
static int kpar2d_one(const chap_conv_params* p, hipStream_t s) {
    const int Ctot = p->src[0].C + (p->nsrc > 1 ? p->src[1].C : 0);
    const bool single = Ctot == 32 * CPAR, keep = p->src[0].keep || (p->nsrc > 1 && p->src[1].keep), one = p->nsrc == 1;
#define CHAP_K2D(O, K, S) if (one == O && keep == K && single == S) return kpar2d_launch<CPAR, O, K, S>(p, s)
    CHAP_K2D(true, false, true); CHAP_K2D(true, true, true); CHAP_K2D(false, false, true); CHAP_K2D(false, true, true);
    CHAP_K2D(true, false, false); CHAP_K2D(true, true, false); CHAP_K2D(false, false, false); CHAP_K2D(false, true, false);
#undef CHAP_K2D
    return CHAP_EUNSUPPORTED;
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
        if (cpar == 4) return d3 ? kpar_one<true, 32, 4>(p, s) : kpar2d_one<4>(p, s);
        return d3 ? kpar_one<true, 32, 2>(p, s) : kpar2d_one<2>(p, s);
    }
    if (cpar == 4) return d3 ? kpar_one<true, 16, 4>(p, s) : kpar_one<false, 16, 4>(p, s);
    return d3 ? kpar_one<true, 16, 2>(p, s) : kpar_one<false, 16, 2>(p, s);
}
