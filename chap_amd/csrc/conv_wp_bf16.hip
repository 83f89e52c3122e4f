// Launcher of conv_wp_kernel (conv_wp.h): the 2D full-resolution 3x3 layers in bf16, wave-private pipelines.  conv_api.hip decides eligibility.
#include "conv_wp.h"
#include <atomic>
#include <cstdlib>

template <int KC, int NT, bool LANESEL>
static int conv_wp_launch(const chap_conv_params* p, hipStream_t stream) {
    typedef conv_geom<3, 1, false, 1, false> G;
    constexpr int MINW = KC == 16 ? (NT == 1 ? 4 : 3) : 2;      // waves per SIMD the registers allow without spilling (128 / 168 / 256 VGPRs)
    const void* kern = chap_kernel<chap_conv_params, conv_wp_kernel<KC, NT, LANESEL>, 256, MINW>();
    const size_t lds = conv_wp_lds_bytes<KC, NT>();
    static std::atomic<int> attr_set[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { chap_set_error("conv(wp): hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e)); return CHAP_ELAUNCH; }
        }
        attr_set[dev].store(1, std::memory_order_release);
    }
    const long ntiles = (long)p->N * cdiv(p->H, G::TH) * cdiv(p->W, G::TW);
    const int gy = cdiv((p->Cout + 15) / 16, NT);
    // persistent blocks of four wave pipelines: CHAP_CONV_WP_BPC blocks per CU (lab knob; default 4 with 16-channel chunks, 2 with 32), at least two tiles per wave
    const char* e = getenv("CHAP_CONV_WP_BPC");
    const long bpc = (e && atol(e) > 0) ? atol(e) : (KC == 16 ? 4 : 2);
    long gx = 256 * bpc / gy;
    const long need = (ntiles + 7) / 8;                         // two tiles per wave
    if (gx > need) gx = need;
    if (gx > CHAP_STATS_MAX_SLOTS) gx = CHAP_STATS_MAX_SLOTS;
    gx = gx >= 8 ? gx / 8 * 8 : (gx < 1 ? 1 : gx);
    return chap_launch_ptr<chap_conv_params>(kern, dim3((unsigned)gx, gy), dim3(256), lds, stream, *p, "chap_conv_fwd(wp)");
}

// KC = all input channels (16 or 32) in one chunk; two concatenated sources inside it: per-lane source select
int chap_conv_launch_wp_bf16(const chap_conv_params* p, int KC, hipStream_t s) {
    const int nt = (p->Cout + 15) / 16;
    const bool two = p->nsrc == 2;
    if (KC == 16) {
        if (two) { chap_set_error("chap_conv_fwd(wp): two sources need a 32-channel chunk"); return CHAP_EUNSUPPORTED; }
        return nt == 1 ? conv_wp_launch<16, 1, false>(p, s) : conv_wp_launch<16, 2, false>(p, s);
    }
    if (nt == 1) return two ? conv_wp_launch<32, 1, true>(p, s) : conv_wp_launch<32, 1, false>(p, s);
    return two ? conv_wp_launch<32, 2, true>(p, s) : conv_wp_launch<32, 2, false>(p, s);      // (64 output channels: two block rows; four 16-channel tiles per wave spill)
}
