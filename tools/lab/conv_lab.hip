// Stand-alone ablation harness for conv_fwd_kernel (build with -DCHAP_ABLATE=<bits>):
//   1 skip MFMA (LDS fragment reads kept alive)   2 skip the whole tap loop
//   4 force the plain (no transform) commit path  8 skip global stores   16 skip halo global loads
#include "conv_kernel.h"
#include "conv_kpar.h"
#include "conv_kpar2d.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
bool chap_group_recording() { return false; }
int chap_group_record(const chap_pending&, hipStream_t) { return 0; }
void chap_set_error(const char* fmt, ...) {}

template <int KS, int KC, int NT, int MR, bool WLDS, bool D3 = false, bool ZW = false>
static void run(const char* name, int N, int H, int W, int Cin, int Cout, bool prologue, bool stats, int D = 1) {
    typedef uint16_t T;
    typedef conv_geom<KS, 1, D3, MR, ZW> G;
    constexpr int GPT = KC / 8, NP = G::NTAPS * GPT, STEPS = (NP + 3) / 4;
    const int nchunks = Cin / KC, ntile16 = (Cout + 15) / 16;
    size_t nin = (size_t)N * D * H * W * Cin, nout = (size_t)N * D * H * W * Cout;
    T *x, *y, *wp; float *sc, *sh, *st;
    hipMalloc(&x, nin * 2); hipMalloc(&y, nout * 2);
    size_t wbytes_all = (size_t)nchunks * STEPS * ntile16 * 64 * 8 * 2;
    hipMalloc(&wp, wbytes_all); hipMalloc(&sc, Cin * 4); hipMalloc(&sh, Cin * 4); hipMalloc(&st, (CHAP_STATS_HDR + (size_t)CHAP_STATS_MAX_SLOTS * 2 * Cout) * 4);
    std::vector<T> hx(nin); for (size_t i = 0; i < nin; ++i) hx[i] = 0x3c00 + (rand() & 0x3ff);
    hipMemcpy(x, hx.data(), nin * 2, hipMemcpyHostToDevice);
    std::vector<T> hw(wbytes_all / 2); for (auto& v : hw) v = 0x3800 + (rand() & 0xff);
    hipMemcpy(wp, hw.data(), wbytes_all, hipMemcpyHostToDevice);
    std::vector<float> hs(Cin, 1.01f); hipMemcpy(sc, hs.data(), Cin * 4, hipMemcpyHostToDevice); hipMemcpy(sh, hs.data(), Cin * 4, hipMemcpyHostToDevice);
    
    chap_conv_params P = {};
    P.src[0].ptr = x; P.src[0].C = Cin; P.src[0].ld = Cin; P.src[0].slope = 0.01f; P.src[0].keep_scale = 1.f;
    if (prologue) { P.src[0].scale = sc; P.src[0].shift = sh; P.src[0].act = 1; }
    P.nsrc = 1; P.N = N; P.D = D; P.H = H; P.W = W; P.ID = D; P.IH = H; P.IW = W; P.ksize = KS; P.stride = 1; P.dims = D3 ? 3 : 2;
    P.wpacked = wp; P.out = y; P.Cout = Cout; P.out_ld = Cout; P.stats = stats ? st : nullptr; P.stats_shift = nullptr; P.dtype = CHAP_BF16;
    // the conv body runs behind launch.h's trampoline (round 3): one argument block = group 0
    auto kern = chap_grouped<chap_conv_params, conv_fwd_kernel<T, KS, 1, D3, KC, NT, MR, false, WLDS, ZW, true>, 256, conv_min_waves<KC, D3>()>;
    chap_group<chap_conv_params> PG; for (int i = 0; i < CHAP_MAX_GROUP; ++i) PG.p[i] = P;
    size_t lds_fixed_only = conv_lds_fixed_bytes<T, KS, 1, D3, KC, MR, ZW>(NT) + 2 * CONV_MAX_AFFINE_C * 4;
    const bool fits = conv_wstaged<T, KS, 1, D3, KC, NT, MR, ZW>();
    const int wstage = WLDS || fits;
    size_t lds = conv_lds_fixed_bytes<T, KS, 1, D3, KC, MR, ZW>(NT) + 2 * CONV_MAX_AFFINE_C * 4 + (WLDS ? (size_t)nchunks * STEPS * NT * 1024 : (wstage ? 2 * (size_t)STEPS * NT * 1024 : 0));
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 256, lds);
    long ntiles = (long)N * ((D + G::TD - 1) / G::TD) * ((H + G::TH - 1) / G::TH) * ((W + 15) / 16);
    int gy = (ntile16 + NT - 1) / NT;
    for (int bpc = 1; bpc <= occ; bpc = (bpc * 2 <= occ || bpc == occ) ? bpc * 2 : occ) {
        long gx = std::min<long>((ntiles + 7) / 8 * 8, (long)256 * bpc / gy / 8 * 8);
        if (gx < 8) gx = 8;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), lds, 0, PG);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), lds, 0, PG);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1e3 / 20, gb = (double)(nin + nout) * 2 / us / 1e3;
#ifdef CHAP_CONV_TRACE
        if (gx >= 1024 || D3 || getenv("LAB_TRACE_ALL")) {
            unsigned long long h[4][64][8];
            hipMemcpyFromSymbol(h, HIP_SYMBOL(chap_trace), sizeof(h));
            for (int b = 0; b < 2; ++b) for (int w = 0; w < 2; ++w) {
                printf("  block %d wave %d (100 MHz ticks -> us):", b * 257, w);
                for (int it = 0; it < 4; ++it) {
                    unsigned long long* r = h[b][it * 4 + w];
                    printf(" | it%d issue %.2f mfma %.2f epi %.2f commit %.2f bar %.2f", it, (r[1]-r[0])/100., (r[2]-r[1])/100., (r[3]-r[2])/100., (r[4]-r[3])/100., (r[5]-r[4])/100.);
                }
                printf("\n");
            }
            unsigned long long hp[4][4][8];
            hipMemcpyFromSymbol(hp, HIP_SYMBOL(chap_trace_p), sizeof(hp));
            unsigned long long t0 = hp[0][0][0];
            for (int b = 0; b < 4; ++b) { unsigned long long* r = hp[b][0];
                printf("  block %d: start %+.2f | setup %.2f units+xoff %.2f rest-to-loads-issued %.2f | loads issued %.2f | barrier %.2f | first commit %.2f | item loop %.2f | stats flush %.2f | total %.2f\n", b * 257, ((double)r[0] - (double)t0) / 100., (r[6]-r[0])/100., (r[7]-r[6])/100., (r[1]-r[7])/100., (r[1]-r[0])/100., (r[2]-r[1])/100., (r[3]-r[2])/100., (r[4]-r[3])/100., (r[5]-r[4])/100., (r[5]-r[0])/100.); }
        }
#endif
        printf("mw=%d abl=%2d %-28s occ=%d grid=%5ldx%d lds=%6zu : %8.1f us %7.1f GB/s\n", CHAP_CONV_MINWAVES, CHAP_ABLATE, name, occ, gx, gy, lds, us, gb);
    }
    hipFree(x); hipFree(y); hipFree(wp); hipFree(sc); hipFree(sh); hipFree(st);
}

// conv_kpar_kernel (K-chunks side by side) on a 2D deep layer, with its per-wave phase stamps under -DCHAP_CONV_TRACE
template <int KC, int NT, int CPAR, bool NEW2D = false>
static void run_kpar(const char* name, int N, int H, int W, int Cin, int Cout, int bpc_max = 2) {
    typedef uint16_t T;
    typedef conv_geom<3, 1, false, 2> G;
    constexpr int GPT = KC / 8, NP = G::NTAPS * GPT, STEPS = (NP + 3) / 4;
    const int nchunks = Cin / KC, ntile16 = (Cout + 15) / 16;
    size_t nin = (size_t)N * H * W * Cin, nout = (size_t)N * H * W * Cout;
    T *x, *y, *wp; float *sc, *sh, *st;
    hipMalloc(&x, nin * 2); hipMalloc(&y, nout * 2);
    size_t wbytes_all = (size_t)nchunks * STEPS * ntile16 * 64 * 8 * 2;
    hipMalloc(&wp, wbytes_all); hipMalloc(&sc, Cin * 4); hipMalloc(&sh, Cin * 4); hipMalloc(&st, (CHAP_STATS_HDR + (size_t)CHAP_STATS_MAX_SLOTS * 2 * Cout) * 4);
    std::vector<T> hx(nin); for (size_t i = 0; i < nin; ++i) hx[i] = 0x3c00 + (rand() & 0x3ff);
    hipMemcpy(x, hx.data(), nin * 2, hipMemcpyHostToDevice);
    std::vector<T> hw(wbytes_all / 2); for (auto& v : hw) v = 0x3800 + (rand() & 0xff);
    hipMemcpy(wp, hw.data(), wbytes_all, hipMemcpyHostToDevice);
    std::vector<float> hs(Cin, 1.01f); hipMemcpy(sc, hs.data(), Cin * 4, hipMemcpyHostToDevice); hipMemcpy(sh, hs.data(), Cin * 4, hipMemcpyHostToDevice);
    chap_conv_params P = {};
    P.src[0].ptr = x; P.src[0].C = Cin; P.src[0].ld = Cin; P.src[0].slope = 0.01f; P.src[0].keep_scale = 1.f;
    P.src[0].scale = sc; P.src[0].shift = sh; P.src[0].act = 1;
    P.nsrc = 1; P.N = N; P.D = 1; P.H = H; P.W = W; P.ID = 1; P.IH = H; P.IW = W; P.ksize = 3; P.stride = 1; P.dims = 2;
    P.wpacked = wp; P.out = y; P.Cout = Cout; P.out_ld = Cout; P.stats = st; P.stats_shift = nullptr; P.dtype = CHAP_BF16;
    void (*kern)(const chap_group<chap_conv_params>) = nullptr;
    if constexpr (NEW2D) {
        if (Cin == 32 * CPAR) kern = chap_grouped<chap_conv_params, conv_kpar2d_kernel<NT, CPAR, true, false, true>, 256, 2>;
        else kern = chap_grouped<chap_conv_params, conv_kpar2d_kernel<NT, CPAR, true, false, false>, 256, 2>;
    } else kern = chap_grouped<chap_conv_params, conv_kpar_kernel<T, false, KC, NT, CPAR, true>, 256, 2>;
    chap_group<chap_conv_params> PG; for (int i = 0; i < CHAP_MAX_GROUP; ++i) PG.p[i] = P;
    const size_t lds = conv_kpar_lds_bytes<T, false, KC, CPAR>(NT);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 256, lds);
    long ntiles = (long)N * ((H + G::TH - 1) / G::TH) * ((W + 15) / 16);
    int gy = (ntile16 + NT - 1) / NT;
    long gx = ntiles;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), lds, 0, PG);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), lds, 0, PG);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kpar %-28s occ=%d grid=%5ldx%d lds=%6zu : %8.1f us per launch (back to back)\n", name, occ, gx, gy, lds, ms * 1e3 / 20);
#ifdef CHAP_CONV_TRACE
    unsigned long long h[4][4][16];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(chap_trace_k), sizeof(h));
    const char* ph[10] = {"setup", "issue", "bar0", "commit", "bar1", "taps", "bar2", "red", "store", "stats"};
    for (int b = 0; b < 4; ++b) for (int w = 0; w < 4; ++w) {
        unsigned long long* r = h[b][w];
        printf("  block %3d wave %d start %+6.2f us:", b * 97, w, ((double)r[0] - (double)h[0][0][0]) / 100.);
        for (int k = 0; k < 10; ++k) printf(" %s %.2f", ph[k], ((double)r[k + 1] - (double)r[k]) / 100.);
        printf(" | total %.2f\n", ((double)r[10] - (double)r[0]) / 100.);
    }
#endif
    hipFree(x); hipFree(y); hipFree(wp); hipFree(sc); hipFree(sh); hipFree(st);
}

template <int KC, bool D3, bool ZW, int MR> static void sweep_nt(const char* tag, int N, int D, int H, int W, int C) {
    char nm[64];
    snprintf(nm, 64, "%s MR%d NT4", tag, MR); run<3, KC, 4, MR, false, D3, ZW>(nm, N, H, W, C, C, true, true, D);
    snprintf(nm, 64, "%s MR%d NT2", tag, MR); run<3, KC, 2, MR, false, D3, ZW>(nm, N, H, W, C, C, true, true, D);
    snprintf(nm, 64, "%s MR%d NT1", tag, MR); run<3, KC, 1, MR, false, D3, ZW>(nm, N, H, W, C, C, true, true, D);
}
int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 's') {       // blocking sweep on the deep layers
        for (int N : {12, 24}) {
            char t[32];
            snprintf(t, 32, "64@64 N%d", N);  sweep_nt<32, false, false, 4>(t, N, 1, 64, 64, 64); sweep_nt<32, false, false, 2>(t, N, 1, 64, 64, 64); sweep_nt<32, false, false, 1>(t, N, 1, 64, 64, 64);
            snprintf(t, 32, "128@32 N%d", N); sweep_nt<32, false, false, 4>(t, N, 1, 32, 32, 128); sweep_nt<32, false, false, 2>(t, N, 1, 32, 32, 128); sweep_nt<32, false, false, 1>(t, N, 1, 32, 32, 128);
            snprintf(t, 32, "256@16 N%d", N); sweep_nt<32, false, false, 4>(t, N, 1, 16, 16, 256); sweep_nt<32, false, false, 2>(t, N, 1, 16, 16, 256); sweep_nt<32, false, false, 1>(t, N, 1, 16, 16, 256);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'l') {       // 3D level 1 (32 channels at 40x56x56): bricks with KC16, NT / weight residency variants
        run<3, 16, 2, 4, true, true, true>("3D 32->32@40x56x56 ZW KC16 NT2 WLDS", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 16, 2, 4, false, true, true>("3D 32->32@40x56x56 ZW KC16 NT2 stg", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 16, 1, 4, true, true, true>("3D 32->32@40x56x56 ZW KC16 NT1 WLDS", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 16, 1, 4, false, true, true>("3D 32->32@40x56x56 ZW KC16 NT1 stg", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 16, 2, 4, true, true, true>("3D 64->32@40x56x56 ZW KC16 NT2 WLDS", 2, 56, 56, 64, 32, true, true, 40);
        run<3, 16, 2, 4, false, true, true>("3D 64->32@40x56x56 ZW KC16 NT2 stg", 2, 56, 56, 64, 32, true, true, 40);
        run<3, 16, 1, 4, true, true, true>("3D 64->32@40x56x56 ZW KC16 NT1 WLDS", 2, 56, 56, 64, 32, true, true, 40);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'k') {       // 3D: KC16 bricks vs KC32
        run<3, 16, 2, 4, true, true, true>("3D 32->32@40x56x56 ZW KC16 NT2 WLDS", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 16, 2, 4, false, true, true>("3D 32->32@40x56x56 ZW KC16 NT2 stg", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 32, 2, 4, false, true, true>("3D 32->32@40x56x56 ZW KC32 NT2", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 16, 1, 4, true, true, true>("3D 32->16@80x112x112 ZW KC16 NT1 WLDS", 2, 112, 112, 32, 16, true, true, 80);
        run<3, 32, 1, 4, false, true, true>("3D 32->16@80x112x112 ZW KC32 NT1", 2, 112, 112, 32, 16, true, true, 80);
        for (int N : {2, 4}) {
            char t[48];
            snprintf(t, 48, "3D 64@20x28x28 N%d ZW KC16 NT2", N); run<3, 16, 2, 4, false, true, true>(t, N, 28, 28, 64, 64, true, true, 20);
            snprintf(t, 48, "3D 64@20x28x28 N%d ZW KC32 NT4", N); run<3, 32, 4, 4, false, true, true>(t, N, 28, 28, 64, 64, true, true, 20);
            snprintf(t, 48, "3D 128@10x14x14 N%d ZW KC16 NT2", N); run<3, 16, 2, 4, false, true, true>(t, N, 14, 14, 128, 128, true, true, 10);
            snprintf(t, 48, "3D 128@10x14x14 N%d ZW KC32 NT2", N); run<3, 32, 2, 4, false, true, true>(t, N, 14, 14, 128, 128, true, true, 10);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'h') {       // 3D half-resolution layer
        run<3, 32, 2, 4, false, true, true>("3D 32->32@40x56x56 ZW NT2", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 32, 2, 1, false, true, false>("3D 32->32@40x56x56 MR1 NT2", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 32, 1, 4, false, true, true>("3D 32->16@80x112x112 ZW NT1", 2, 112, 112, 32, 16, true, true, 80);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 't') {       // 3D deep layers
        for (int N : {2, 4}) {
            char t[32];
            snprintf(t, 32, "3D 64@20x28x28 N%d", N);  sweep_nt<32, true, false, 1>(t, N, 20, 28, 28, 64); sweep_nt<32, true, true, 4>(t, N, 20, 28, 28, 64);
            snprintf(t, 32, "3D 128@10x14x14 N%d", N); sweep_nt<32, true, false, 1>(t, N, 10, 14, 14, 128); sweep_nt<32, true, true, 4>(t, N, 10, 14, 14, 128);
            snprintf(t, 32, "3D 256@5x7x7 N%d", N);    sweep_nt<32, true, false, 1>(t, N, 5, 7, 7, 256); sweep_nt<32, true, true, 4>(t, N, 5, 7, 7, 256);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'p') {       // 2D deep layers on the K-parallel kernel (conv_kpar.h), phase stamps per wave
        run_kpar<32, 2, 4>("128->128@32 N12 kpar NT2 C4", 12, 32, 32, 128, 128);
        run_kpar<32, 2, 4>("256->256@16 N12 kpar NT2 C4", 12, 16, 16, 256, 256);
        run_kpar<32, 2, 2>("64->64@64 N12 kpar NT2 C2", 12, 64, 64, 64, 64);
        run_kpar<32, 2, 4>("128->128@32 N24 kpar NT2 C4", 24, 32, 32, 128, 128);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'q') {       // the round-4 form of the 2D K-parallel kernel (conv_kpar2d.h), phase stamps per wave
        run_kpar<32, 2, 4, true>("128->128@32 N12 kpar2d C4", 12, 32, 32, 128, 128);
        run_kpar<32, 2, 4, true>("256->256@16 N12 kpar2d C4 (2 rounds)", 12, 16, 16, 256, 256);
        run_kpar<32, 2, 2, true>("64->64@64 N12 kpar2d C2", 12, 64, 64, 64, 64);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'd') {       // 2D deep layers with the library's blocking (KC32 NT2 MR2, staged weights)
        run<3, 32, 2, 2, false>("128->128@32 N12 NT2 MR2", 12, 32, 32, 128, 128, true, true);
        run<3, 32, 2, 2, false>("256->256@16 N12 NT2 MR2", 12, 16, 16, 256, 256, true, true);
        run<3, 32, 2, 2, false>("64->64@64 N12 NT2 MR2", 12, 64, 64, 64, 64, true, true);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'e') {       // 3D deep layers, 1 x 4 x 16 slabs: KC32 (weights streamed from L2 per wave) vs KC16 (staged through LDS)
        run<3, 32, 2, 1, false, true, false>("3D 64@20x28x28 N2 slab KC32 NT2", 2, 28, 28, 64, 64, true, true, 20);
        run<3, 16, 2, 1, false, true, false>("3D 64@20x28x28 N2 slab KC16 NT2", 2, 28, 28, 64, 64, true, true, 20);
        run<3, 16, 4, 1, false, true, false>("3D 64@20x28x28 N2 slab KC16 NT4", 2, 28, 28, 64, 64, true, true, 20);
        run<3, 16, 2, 4, false, true, true>("3D 64@20x28x28 N2 brick KC16 NT2", 2, 28, 28, 64, 64, true, true, 20);
        run<3, 32, 2, 1, false, true, false>("3D 128@10x14x14 N2 slab KC32 NT2", 2, 14, 14, 128, 128, true, true, 10);
        run<3, 16, 2, 1, false, true, false>("3D 128@10x14x14 N2 slab KC16 NT2", 2, 14, 14, 128, 128, true, true, 10);
        run<3, 16, 4, 1, false, true, false>("3D 128@10x14x14 N2 slab KC16 NT4", 2, 14, 14, 128, 128, true, true, 10);
        run<3, 32, 2, 1, false, true, false>("3D 256@5x7x7 N2 slab KC32 NT2", 2, 7, 7, 256, 256, true, true, 5);
        run<3, 16, 2, 1, false, true, false>("3D 256@5x7x7 N2 slab KC16 NT2", 2, 7, 7, 256, 256, true, true, 5);
        run<3, 16, 4, 1, false, true, false>("3D 256@5x7x7 N2 slab KC16 NT4", 2, 7, 7, 256, 256, true, true, 5);
        return 0;
    }
    if (argc > 1 && argv[1][0] == '3') {
        run<3, 16, 1, 4, true, true, true>("3D 16->16@80x112x112 ZW", 2, 112, 112, 16, 16, true, true, 80);
        run<3, 32, 2, 4, true, true, true>("3D 32->32@40x56x56 ZW", 2, 56, 56, 32, 32, true, true, 40);
        run<3, 32, 4, 1, false, true, false>("3D 64->64@20x28x28 MR1", 4, 28, 28, 64, 64, true, true, 20);
        run<3, 32, 4, 4, false, true, true>("3D 64->64@20x28x28 ZW", 4, 28, 28, 64, 64, true, true, 20);
        run<3, 32, 4, 1, false, true, false>("3D 128->128@10x14x14 MR1", 4, 14, 14, 128, 128, true, true, 10);
        return 0;
    }
    run<3, 16, 1, 4, true>("16->16@256 MR4 pro+stats", 12, 256, 256, 16, 16, true, true);
    run<3, 16, 1, 2, true>("16->16@256 MR2 pro+stats", 12, 256, 256, 16, 16, true, true);
    run<3, 16, 1, 8, true>("16->16@256 MR8 pro+stats", 12, 256, 256, 16, 16, true, true);
    run<3, 32, 2, 4, true>("32->32@128 MR4 pro+stats", 12, 128, 128, 32, 32, true, true);
    run<3, 32, 4, 2, true>("64->64@64 MR2 pro+stats", 12, 64, 64, 64, 64, true, true);
    run<3, 32, 4, 2, false>("128->128@32 MR2", 12, 32, 32, 128, 128, true, true);
    return 0;
}
