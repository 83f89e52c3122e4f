// First conv of the networks (Cin = 1, 3x3 / 3x3x3 "same", 16 output channels; unet.py:50 with in_chns = 1, vnet.py:19 with n_channels = 1), bf16 mode.
//
// Rounds 1-3 ran it on the generic MFMA kernel over the input zero-padded to 16 channels: `chap_planar_to_cl` wrote 32 B per voxel of which 30 are
// zeros (3D: 64 MB per pass for a 4 MB image, 20-35 us), and the 16->16 conv read them back (65 us at 112x112x80, N = 2) -- nine passes per iteration.
// Here the taps ARE the GEMM's K dimension: out[cout][pixel] = sum_tap W[cout][tap] * x[pixel + tap], K = 9 or 27 padded to 32 = ONE
// `v_mfma_f32_16x16x32_bf16` per 16 pixels x 16 channels.  The block stages a halo of the fp32 image (rounded to bf16, exactly what the padded path
// fed the MFMA) in LDS; a lane gathers its 8 taps of one pixel with 8 two-byte LDS reads (taps beyond the kernel re-read tap 0 against a zero
// weight).  Same products as the padded path, summed in another order; bias, shifted BatchNorm statistics (one partial slot per block, the conv
// kernels' layout) and the channel-last bf16 output as there.  Bound: the 32 B per voxel it writes.
#pragma once
#include "conv_kernel.h"

// block tile: 8 rows (2D: 8 y; 3D: 2 z x 4 y) x TXT 16-pixel tiles along x; the 8 * TXT (row, x-tile) units are dealt to the four waves
template <bool D3, int TXT>
struct c1m_geom {
    static constexpr int TZ = D3 ? 2 : 1, TY = D3 ? 4 : 8, TX = 16 * TXT;
    static constexpr int HZ = D3 ? TZ + 2 : 1, HY = TY + 2, HX = TX + 2;
    static constexpr int HALO = HZ * HY * HX, NL = (HALO + 255) / 256;
    static constexpr int TAPS = D3 ? 27 : 9;
};

template <bool D3, int TXT>
__device__ __forceinline__ void conv_c1_mfma_kernel(const chap_conv_c1_params& P) {
    typedef c1m_geom<D3, TXT> G;
    typedef bf16_t T;
    __shared__ __attribute__((aligned(16))) T halo[G::HALO + 8];
    __shared__ float bstat[4 * 2 * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const int tiles_x = (P.W + G::TX - 1) / G::TX, tiles_y = (P.H + G::TY - 1) / G::TY, tiles_z = (P.D + G::TZ - 1) / G::TZ;
    const long ntiles = (long)P.N * tiles_z * tiles_y * tiles_x;
    const bool do_stats = P.stats != nullptr;

    // this lane's weight fragment: 8 taps (k = 8g .. 8g + 7) of output channel px, zero beyond the kernel; bias / statistics shift of channels 4g .. 4g + 3
    uint4 wf;
    {
        float w8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k = 8 * g + j; w8[j] = k < G::TAPS ? P.w[px * G::TAPS + k] : 0.f; }
        wf = frag<T>::pack(w8);
    }
    float bj[4], cj[4], ssum[4], ssq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bj[j] = P.bias ? P.bias[4 * g + j] : 0.f;
        cj[j] = (do_stats && P.stats_shift) ? P.stats_shift[4 * g + j] : 0.f;
        ssum[j] = 0.f; ssq[j] = 0.f;
    }
    // LDS element offset of tap k relative to the pixel's halo origin (taps beyond the kernel: tap 0 -- finite data against a zero weight)
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j, kk = k < G::TAPS ? k : 0;
        const int dx = kk % 3, dy = (kk / 3) % 3, dz = kk / 9;
        toff[j] = (dz * G::HY + dy) * G::HX + dx;
    }
    // halo elements of this thread: e = tid + 256 i  ->  (hz, hy, hx)
    int hrel[G::NL], hzyx[G::NL];
#pragma unroll
    for (int i = 0; i < G::NL; ++i) {
        const int e = threadIdx.x + 256 * i, ee = e < G::HALO ? e : 0;
        const int hx = ee % G::HX, hy = (ee / G::HX) % G::HY, hz = ee / (G::HX * G::HY);
        hzyx[i] = e < G::HALO ? ((hz << 20) | (hy << 10) | hx) : -1;
        hrel[i] = (hz * P.H + hy) * P.W + hx;
    }
    float pre[G::NL];
    auto coords = [&](long t, int& n, int& z0, int& y0, int& x0) __attribute__((always_inline)) {
        const unsigned u = (unsigned)t;
        x0 = (int)(u % (unsigned)tiles_x) * G::TX; unsigned r = u / (unsigned)tiles_x;
        y0 = (int)(r % (unsigned)tiles_y) * G::TY; r /= (unsigned)tiles_y;
        z0 = (int)(r % (unsigned)tiles_z) * G::TZ; n = (int)(r / (unsigned)tiles_z);
    };
    auto issue = [&](long t) __attribute__((always_inline)) {
        int n, z0, y0, x0;
        coords(t, n, z0, y0, x0);
        const int gz0 = z0 - (D3 ? 1 : 0), gy0 = y0 - 1, gx0 = x0 - 1;
        const long base = (((long)n * P.D + gz0) * P.H + gy0) * P.W + gx0;
#pragma unroll
        for (int i = 0; i < G::NL; ++i) {
            const int c = hzyx[i];
            const int zz = gz0 + (c >> 20), yy = gy0 + ((c >> 10) & 1023), xx = gx0 + (c & 1023);
            const bool ok = c >= 0 && (unsigned)zz < (unsigned)P.D && (unsigned)yy < (unsigned)P.H && (unsigned)xx < (unsigned)P.W;
            pre[i] = ok ? P.x[base + hrel[i]] : 0.f;
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G::NL; ++i) {
            const int e = threadIdx.x + 256 * i;
            halo[e < G::HALO ? e : G::HALO] = elem<T>::put(pre[i]);
        }
    };
    long t = blockIdx.x;
    if (t < ntiles) issue(t);
    for (; t < ntiles; t += gridDim.x) {
        __syncthreads();                                        // the previous tile's gathers are done
        commit();
        __syncthreads();
        int n, z0, y0, x0;
        coords(t, n, z0, y0, x0);
        const long tn = t + gridDim.x;
        if (tn < ntiles) issue(tn);                             // next tile's loads fly beside this tile's gathers and stores
        for (int u = wave; u < 8 * TXT; u += 4) {
            const int r = u / TXT, xt = u % TXT;
            const int tz = D3 ? r / G::TY : 0, ty = D3 ? r % G::TY : r;
            const T* hp = halo + (tz * G::HY + ty) * G::HX + xt * 16 + px;
            unsigned short v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = hp[toff[j]];
            const uint4 xf = make_uint4((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                                        (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16));
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma8(acc, wf, xf);
            const int z = z0 + tz, y = y0 + ty, x = x0 + xt * 16 + px;
            const bool valid = z < P.D && y < P.H && x < P.W;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = acc[j] + bj[j];
                const float d = valid ? o[j] - cj[j] : 0.f;
                ssum[j] += d; ssq[j] += d * d;
            }
            if (valid) st4((T*)P.out + ((((long)n * P.D + z) * P.H + y) * P.W + x) * 16 + 4 * g, o);
        }
    }
    if (do_stats) {      // registers -> 16-lane DPP reduce -> LDS row of this wave -> the four rows in a fixed order -> this block's partial slot
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = row16_sum(ssum[j]), q = row16_sum(ssq[j]);
            if (px == 0) { bstat[(wave * 2 + 0) * 16 + 4 * g + j] = s; bstat[(wave * 2 + 1) * 16 + 4 * g + j] = q; }
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) *(int*)P.stats = (int)gridDim.x;      // header: slots in use
        float* st = P.stats + CHAP_STATS_HDR + (long)blockIdx.x * 2 * 16;
        if (threadIdx.x < 32) {
            const int which = threadIdx.x >> 4, c = threadIdx.x & 15;
            st[which * 16 + c] = (bstat[(0 * 2 + which) * 16 + c] + bstat[(1 * 2 + which) * 16 + c]) + (bstat[(2 * 2 + which) * 16 + c] + bstat[(3 * 2 + which) * 16 + c]);
        }
    }
}
