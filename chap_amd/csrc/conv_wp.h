// 3x3 convolution of the 2D FULL-RESOLUTION layers (16 or 32 input channels in one K-chunk) with WAVE-PRIVATE pipelines, bf16 (round 4).
//
// conv_fwd_kernel runs a block as ONE pipeline: the four waves share a tile (prefetch -> commit -> barrier -> MFMA -> epilogue), three blocks per CU:
// ~17 KB of halo in flight per CU.  These layers are the HBM-streaming part of the network (0.31-0.43 of the roof: the 16->16 layer at 256 x 256 moves
// 50 MB in 16-20 us) and, like the weight gradient before wgrad_wp.h, they are limited by what a CU keeps in flight, not by arithmetic.  Here every
// WAVE is its own pipeline: its own sequence of 4 x 16-pixel tiles, its own LDS halo region (single-buffered: a wave's LDS operations execute in
// order) and its own register prefetch of the next tile; it computes all output channels of its tile from the block's weights, which sit in LDS once
// per block (5-36 KB).  No block barrier in the tile loop; the four waves meet at the start (weights, scale / shift cache) and at the end (BatchNorm
// statistics: the four waves' registers -> one partial slot per block, same layout and order as conv_fwd_kernel).
// Same operands, packed-weight layout, lazy-activation prologue (BN affine, LeakyReLU, keep masks, channel multipliers, two concatenated sources
// inside the one chunk), bias, shifted statistics, channel-last output incl. out2; per output element the same MFMA sequence as conv_fwd_kernel
// (K order: steps of the one chunk), so outputs are bit-identical to it; the statistics partials differ in how the pixels are dealt to slots.
#pragma once
#include "conv_kernel.h"
#include "wgrad_wp.h"          // make_units_wave

constexpr int CWP_AFFC = 64;           // scale | shift halves of 32 channels per source

template <int KC, int NT>
__host__ __device__ constexpr size_t conv_wp_lds_bytes() {
    typedef conv_geom<3, 1, false, 1, false> G;
    constexpr int GPT = KC / 8, NP = G::NTAPS * GPT, STEPS = (NP + 3) / 4;
    return (size_t)STEPS * NT * 512 * sizeof(bf16_t)                                            // the block's weights, fragment order
           + 4 * ((size_t)G::HP * pix_stride<bf16_t, KC>() + HALO_DUMMY) * sizeof(bf16_t)       // one halo region per wave
           + 4 * 2 * 16 * NT * sizeof(float) + 2 * CWP_AFFC * sizeof(float) + 2 * 16 * NT * sizeof(float);
}

// runs behind chap_grouped<chap_conv_params, .., 256, MINW>; grid (blocks, ceil(Cout / (16 NT)))
template <int KC, int NT, bool LANESEL>
__device__ __forceinline__ void conv_wp_kernel(const chap_conv_params& P) {
    typedef bf16_t T;
    typedef conv_geom<3, 1, false, 1, false> G;                 // wave tile = 4 x 16 pixels, halo 6 x 18
    typedef typename frag<T>::type F;
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>();
    constexpr int NP = G::NTAPS * GPT, STEPS = (NP + 3) / 4;
    constexpr int UNITS = (G::HP * GPT + 63) / 64;
    constexpr int ROWS = G::TH;
    constexpr size_t HB = (size_t)G::HP * PS + HALO_DUMMY;
    static_assert((HB * sizeof(T)) % 16 == 0, "the float4 reads of bc need a 16-byte aligned tail");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* wlds = (T*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T* halo = wlds + (size_t)STEPS * NT * 512 + wave * HB;
    float* bstat = (float*)(wlds + (size_t)STEPS * NT * 512 + 4 * HB);
    float* aff = bstat + 4 * 2 * 16 * NT;
    float* bc = aff + 2 * CWP_AFFC;                             // bias | statistics shift of this block's 16 NT output channels (0 where absent)
    const int px = lane & 15, g = lane >> 4;
    const int tiles_x = (P.W + G::TW - 1) / G::TW, tiles_y = (P.H + G::TH - 1) / G::TH;
    const long ntiles = (long)P.N * tiles_y * tiles_x;
    const int ntiles_total = (P.Cout + 15) >> 4;
    const int nt0 = blockIdx.y * NT;
    const bool one = P.nsrc == 1;
    const bool plain = P.src[0].scale == nullptr && !P.src[0].act && P.src[0].keep == nullptr && P.src[0].chan_mul == nullptr &&
                       (one || (P.src[1].scale == nullptr && !P.src[1].act && P.src[1].keep == nullptr && P.src[1].chan_mul == nullptr));
    const src_scalars s0 = make_scalars(P.src[0]);
    const src_scalars s1 = make_scalars(one ? P.src[0] : P.src[1]);
    const bool do_stats = P.stats != nullptr;

    // XCD-aware tile assignment, waves as the unit (as wgrad_wp.h): blocks b and b + 8 share an XCD, which owns a contiguous eighth of the tiles;
    // its waves take them round-robin, so the tiles in flight on an XCD are neighbours (halo rows hit in its L2)
    const int nb = gridDim.x;
    const int NG = nb < 8 ? nb : 8;
    const int xcd = blockIdx.x % NG;
    const int sj = ((int)blockIdx.x / NG) * 4 + wave;
    const int bpx = ((nb + NG - 1 - xcd) / NG) * 4;
    const long per = (ntiles + NG - 1) / NG;
    const long t_lo = per * xcd, t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
    const long my_tiles = (bpx > 0 && t_lo + sj < t_hi) ? (t_hi - t_lo - sj + bpx - 1) / bpx : 0;
    const long last_tile = t_lo + sj + (my_tiles > 0 ? my_tiles - 1 : 0) * bpx;
    auto tile_of = [&](long k) __attribute__((always_inline)) { const long t = t_lo + sj + k * bpx; return t < last_tile ? t : last_tile; };

    unit_desc<UNITS> U;
    make_units_wave<G, GPT, PS, UNITS>(U, P.IH, P.IW, lane);
    halo_regs<T, UNITS, false, true> R;
    int n = 0, y0 = 0, x0 = 0, nn = 0, ny0 = 0, nx0 = 0;
    auto issue = [&](long tile) __attribute__((always_inline)) {
        int z0_;
        tile_coords<G::TH, G::TW, 1>(tile, tiles_x, tiles_y, 1, nn, z0_, ny0, nx0);
        halo_issue_impl<T, G, false, 1, KC, false, UNITS, LANESEL, false>(R, U, s0, s1, 1, P.IH, P.IW, nn, 0, ny0, nx0, 0);
    };
    // ---- prologue: every global load first (first halo, scale / shift, weights, bias), then the dependent LDS stores
    if (my_tiles > 0) issue(tile_of(0));
    float asc[2], ash[2];                                       // thread t < 64: source t / 32, channel t % 32
    {
        const int s_ = threadIdx.x >> 5, c = threadIdx.x & 31;
        const chap_src_t& S = P.src[(s_ < P.nsrc && s_ < 2) ? s_ : 0];
        const int cc = c < S.C ? c : 0;
        asc[0] = S.scale ? S.scale[cc] : 1.f; ash[0] = S.scale ? S.shift[cc] : 0.f;
        asc[1] = 0.f; ash[1] = 0.f;
    }
    const long wstep = (long)ntiles_total * 64 * 8;
    constexpr int WPT = (STEPS * NT * 64 + 255) / 256;          // weight fragments per thread
    F wreg[WPT];
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        const unsigned i = min((unsigned)(threadIdx.x + 256 * k), (unsigned)(STEPS * NT * 64 - 1));
        const unsigned ln = i & 63u, r = i >> 6, t = r % (unsigned)NT, step = r / (unsigned)NT;
        const unsigned tt = nt0 + (int)t < ntiles_total ? t : 0u;
        wreg[k] = frag<T>::load((const T*)P.wpacked + (long)step * wstep + ((long)(nt0 + tt) * 64 + ln) * 8);
    }
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[t][j] = 0.f; ssq[t][j] = 0.f; }
    // bias and statistics shift live in LDS and are re-read per tile (two ds_read_b128): held in registers they pushed the 128-VGPR variant into
    // scratch (32 B per lane = 8 MB of HBM writes per launch of the 16->16 layer, PMC: profiles/r04_pmc_traffic_dominant_2d.jsonl)
    float bcv = 0.f;
    if (threadIdx.x < 2 * 16 * NT) {
        const int which = threadIdx.x / (16 * NT), nl = nt0 * 16 + (int)threadIdx.x % (16 * NT);
        const float* srcp = which == 0 ? P.bias : (do_stats ? P.stats_shift : nullptr);
        if (srcp != nullptr && nl < P.Cout) bcv = srcp[nl];
    }
    if (threadIdx.x < 64) {
        const int s_ = threadIdx.x >> 5, c = threadIdx.x & 31;
        aff[s_ * CWP_AFFC + c] = asc[0]; aff[s_ * CWP_AFFC + CWP_AFFC / 2 + c] = ash[0];
    }
    if (threadIdx.x < 2 * 16 * NT) bc[threadIdx.x] = bcv;
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        const unsigned i = threadIdx.x + 256 * k;
        if (i < (unsigned)(STEPS * NT * 64)) {
            const unsigned t = (i >> 6) % (unsigned)NT;
            frag<T>::store(wlds + (size_t)i * 8, (nt0 + (int)t < ntiles_total) ? wreg[k] : frag<T>::zero());
        }
    }
    int xoff[STEPS];                                            // LDS element offset of this lane's pixel fragment (row 0), -1 = zero fragment
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int p = step * 4 + g;
        const int tap = p / GPT, cgl = p % GPT;
        const int dx = tap % 3, dy = tap / 3;
        xoff[step] = p < NP ? ((dy * G::HW) + px + dx) * PS + cgl * 8 : -1;
    }
    int ooff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nl = (nt0 + t) * 16 + 4 * g;
        ooff[t] = px * P.out_ld + P.out_coff + ((P.out2 && nl >= P.out2_from) ? nl - P.out2_from : nl);
    }
    const int orow = P.W * P.out_ld;
    __syncthreads();                                            // weights and the scale / shift cache are visible: the only block barrier before the end
    if (my_tiles > 0) {
        halo_commit_impl<T, KC, false, UNITS, LANESEL, true, false, CWP_AFFC>(R, U, halo, s0, s1, aff, plain, nn, 0);
        for (long k = 0;; ++k) {
            n = nn; y0 = ny0; x0 = nx0;
            issue(tile_of(k + 1));                              // (the tail re-requests the wave's last tile: every prefetch register is consumed on every path)
            // ---- MFMA over the taps: pixel fragments of the 4 rows from this wave's halo, weight fragments from the block's LDS copy
            f32x4 acc[ROWS][NT];
#pragma unroll
            for (int m = 0; m < ROWS; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int step = 0; step < STEPS; ++step) {
                F wf[NT], xf[ROWS];
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[t] = frag<T>::load(wlds + ((size_t)(step * NT + t) * 64 + lane) * 8);
                const int xo = xoff[step];
#pragma unroll
                for (int m = 0; m < ROWS; ++m) {
                    if (step * 4 + 3 < NP) xf[m] = frag<T>::load(halo + xo + m * (G::HW * PS));
                    else xf[m] = xo >= 0 ? frag<T>::load(halo + xo + m * (G::HW * PS)) : frag<T>::zero();
                }
#pragma unroll
                for (int m = 0; m < ROWS; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) mma8(acc[m][t], wf[t], xf[m]);
            }
            // ---- epilogue: bias, statistics, channel-last vector stores (lane: 4 consecutive output channels of pixel (row m, x = px))
            {
                const bool xok = x0 + px < P.W;
                const long o0 = (((long)n * P.H + y0) * P.W + x0) * P.out_ld;
                char* ob = (char*)P.out + o0 * (long)sizeof(T);
                char* ob2 = P.out2 ? (char*)P.out2 + o0 * (long)sizeof(T) : ob;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bool cok = (nt0 + t) * 16 + 4 * g < P.Cout;
                    char* obt = (P.out2 && (nt0 + t) * 16 >= P.out2_from) ? ob2 : ob;
                    const float4 b4 = *(const float4*)(bc + t * 16 + 4 * g), c4 = *(const float4*)(bc + 16 * NT + t * 16 + 4 * g);
                    const float bj[4] = {b4.x, b4.y, b4.z, b4.w}, cj[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                    for (int m = 0; m < ROWS; ++m) {
                        const bool valid = xok && cok && (y0 + m < P.H);
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[m][t][j] + bj[j];
                        if (do_stats) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) { const float d = valid ? v[j] - cj[j] : 0.f; ssum[t][j] += d; ssq[t][j] += d * d; }
                        }
                        if (valid) st4((T*)(obt + (unsigned)(m * orow + ooff[t]) * (unsigned)sizeof(T)), v);
                    }
                }
            }
            if (k + 1 >= my_tiles) break;                       // (leave BEFORE the commit: see conv_fwd_kernel)
            halo_commit_impl<T, KC, false, UNITS, LANESEL, true, false, CWP_AFFC>(R, U, halo, s0, s1, aff, plain, nn, 0);
        }
    }
    // ---- BatchNorm statistics: registers -> 16-lane DPP reduce -> LDS row of this wave -> the four rows in a fixed order -> this block's partial slot
    if (do_stats) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float s = row16_sum(ssum[t][j]), q = row16_sum(ssq[t][j]);
                if (px == 0) {
                    bstat[(wave * 2 + 0) * 16 * NT + t * 16 + 4 * g + j] = s;
                    bstat[(wave * 2 + 1) * 16 * NT + t * 16 + 4 * g + j] = q;
                }
            }
        }
        __syncthreads();
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *(int*)P.stats = (int)gridDim.x;      // header: slots in use
        float* st = P.stats + CHAP_STATS_HDR + (long)blockIdx.x * 2 * P.Cout;
        for (int i = threadIdx.x; i < 2 * 16 * NT; i += 256) {
            const int which = i / (16 * NT), kk = i % (16 * NT);
            const int nl = nt0 * 16 + kk;
            const float v = (bstat[(0 * 2 + which) * 16 * NT + kk] + bstat[(1 * 2 + which) * 16 * NT + kk]) + (bstat[(2 * 2 + which) * 16 * NT + kk] + bstat[(3 * 2 + which) * 16 * NT + kk]);
            if (nl < P.Cout) st[which * P.Cout + nl] = v;
        }
    }
}
