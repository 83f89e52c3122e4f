// Weight gradient of the 2D 3x3 full-resolution layers with WAVE-PRIVATE pipelines (round 4).
//
// wgrad_kernel (wgrad_kernel.h) runs ONE tile per block through "prefetch -> commit -> barrier -> MFMA": the four waves share the tile
// (each owns a few (tap, 16-row) pairs), so a block has one ~10 KB tile in flight and pays a block barrier per tile; with two blocks per CU
// that is ~20 KB in flight per CU -- Little's law at 8 TB/s x ~2 us asks for ~64 KB -- and the 16-channel layers sat at 0.17 of the HBM roof
// (16->16 at 256 x 256, N = 12: 29 us kernel + 6.5 us slab reduction for 50 MB of operands; rounds 2-3 fixed its LDS pipeline and its over-fetch
// with no change in time: a per-tile round-trip chain).
//
// Here every WAVE is its own pipeline: it owns a sequence of TH x 16 tiles, its own LDS region (A halo + B tile, single-buffered: a wave's
// LDS operations execute in order, so "all reads of tile k, then the writes of tile k + 1" needs no barrier) and its own register prefetch of
// the next tile; it computes all 9 taps of its tile.  A block is four such pipelines that meet once, at the end, to sum their dW registers in a
// fixed order into the block's partial slab -- the slab count (and the slab reduction) stays what it was.  In flight per CU: 8-12 tiles instead of
// 2, no barrier inside the tile loop.  Same per-element arithmetic as wgrad_kernel (bf16 MFMA 16x16x32, fp32 accumulate); the SUMMATION ORDER over
// pixels differs (tiles are dealt to waves, not to blocks), so results agree to fp32 rounding, not bitwise, and are bitwise reproducible run to run.
#pragma once
#include "wgrad_kernel.h"

constexpr int WP_AFFC = 512;           // scale | shift halves of 256 channels per source

template <int KC, int MR, int BN>
__host__ __device__ constexpr size_t wgrad_wp_lds_bytes() {
    typedef conv_geom<3, 1, false, MR, false> G;
    return 4 * ((size_t)G::HP * pix_stride<bf16_t, KC>() + HALO_DUMMY + (size_t)G::TH * G::TW * wg_psb<bf16_t, BN>() + WG_BDUMMY) * sizeof(bf16_t)
           + 3 * WP_AFFC * sizeof(float) + 4 * BN * sizeof(float);
}

template <typename G, int GPT, int PS, int UNITS>
__device__ __forceinline__ void make_units_wave(unit_desc<UNITS>& U, int IH, int IW, int lane) {
    U.c8 = (lane % GPT) * 8;
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        const int u = lane + 64 * j;
        U.hzyx[j] = UNIT_UNUSED; U.rel[j] = 0; U.lds[j] = G::HP * PS;
        if (u < G::HP * GPT) {
            const int pix = u / GPT, cgl = u % GPT;
            const int hx = pix % G::HW, hy = pix / G::HW;
            U.hzyx[j] = (hy << 10) | hx;
            U.rel[j] = hy * IW + hx;
            U.lds[j] = pix * PS + cgl * 8;
        }
    }
}

// runs behind chap_grouped_z<wgrad_args, .., 256, MINW>; grid (nsplit, Ca / KC, ceil(Cb / BN))
template <int KC, int MR, int BN>
__device__ __forceinline__ void wgrad_wp_kernel(const wgrad_args& A, int nb) {
    typedef bf16_t T;
    const chap_wgrad_params& P = A.P;
    float* __restrict__ ws = A.ws;
    float* __restrict__ ws_db = A.ws_db;
    const int nsplit = A.nsplit, Ca = A.Ca, Cb = A.Cb;
    typedef conv_geom<3, 1, false, MR, false> G;
    constexpr int TPX = G::TH * G::TW;
    typedef typename frag<T>::type F;
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>(), PSB = wg_psb<T, BN>();
    constexpr int KCT = KC / 16, NTB = BN / 16, CG = BN / 8;
    constexpr int PAIRS = G::NTAPS * KCT;                 // every wave runs all (tap, 16-row) pairs of its own tile
    constexpr int NKCH = TPX / 32;
    constexpr int UNITS = (G::HP * GPT + 63) / 64;
    constexpr int BUNITS = (TPX * CG + 63) / 64;
    constexpr size_t WBYTES = ((size_t)G::HP * PS + HALO_DUMMY + (size_t)TPX * PSB + WG_BDUMMY) * sizeof(T);
    constexpr int RP = (int)(WBYTES / (256 * sizeof(float))) < PAIRS ? (int)(WBYTES / (256 * sizeof(float))) : PAIRS;      // pairs per round of the end-of-block reduction (it reuses the tile regions)
    static_assert(RP >= 1, "tile region too small for the end-of-block reduction");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T* halo = (T*)(smem + wave * WBYTES);
    T* bt = halo + (size_t)G::HP * PS + HALO_DUMMY;
    float* aff = (float*)(smem + 4 * WBYTES);             // [A0 | A1 | B] x [scale | shift], shared by the block
    float* dbred = aff + 3 * WP_AFFC;
    const int g = lane >> 4;
    const int split = blockIdx.x, chunk = blockIdx.y;
    const int tiles_x = (P.W + G::TW - 1) / G::TW, tiles_y = (P.H + G::TH - 1) / G::TH;
    const long ntiles = (long)P.N * tiles_y * tiles_x;
    // XCD-aware tile assignment as in wgrad_kernel, one level finer: the "splits" are waves.  Blocks s and s + 8 share an XCD; the XCD owns a
    // contiguous eighth of the tiles and its waves take them round-robin (the tiles in flight on an XCD are neighbours: halo rows hit in its L2).
    const int NG = nsplit < 8 ? nsplit : 8;
    const int xcd = split % NG;
    const int sj = (split / NG) * 4 + wave;
    const int bpx = ((nsplit + NG - 1 - xcd) / NG) * 4;
    const long per = (ntiles + NG - 1) / NG;
    const long t_lo = per * xcd, t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
    const long my_tiles = (bpx > 0 && t_lo + sj < t_hi) ? (t_hi - t_lo - sj + bpx - 1) / bpx : 0;

    const src_scalars s0 = make_scalars(P.a[0]);
    const src_scalars s1 = make_scalars(P.na > 1 ? P.a[1] : P.a[0]);
    const src_scalars sb = make_scalars(P.b);
    const bool plainA = P.a[0].scale == nullptr && !P.a[0].act && P.a[0].keep == nullptr && P.a[0].chan_mul == nullptr &&
                        (P.na < 2 || (P.a[1].scale == nullptr && !P.a[1].act && P.a[1].keep == nullptr && P.a[1].chan_mul == nullptr));
    const bool plainB = P.b.scale == nullptr && !P.b.act && P.b.keep == nullptr && P.b.chan_mul == nullptr;

    unit_desc<UNITS> U;
    make_units_wave<G, GPT, PS, UNITS>(U, P.IH, P.IW, lane);
    int b_yx[BUNITS], b_rel[BUNITS], b_lds[BUNITS];
    const int bc8 = (lane & (CG - 1)) * 8;
    const int cbB = nb * BN + bc8;
    const bool bchan_ok = cbB < Cb;
    const int cbB_safe = bchan_ok ? cbB : 0;
#pragma unroll
    for (int j = 0; j < BUNITS; ++j) {
        const int u = lane + 64 * j;
        const int pix = u / CG;
        b_yx[j] = -1; b_rel[j] = 0; b_lds[j] = TPX * PSB;
        if (pix < TPX) {
            const int row = pix / G::TW, col = pix % G::TW;
            b_yx[j] = (row << 8) | col;
            b_rel[j] = row * P.W + col;
            b_lds[j] = pix * PSB + bc8;
        }
    }
    for (int s = 0; s < 3; ++s) {
        const chap_src_t& src = s == 2 ? P.b : P.a[s < P.na ? s : 0];
        const bool has = src.scale != nullptr;
        for (int c = threadIdx.x; c < src.C && c < WP_AFFC / 2; c += 256) {
            aff[s * WP_AFFC + c] = has ? src.scale[c] : 1.f;
            aff[s * WP_AFFC + WP_AFFC / 2 + c] = has ? src.shift[c] : 0.f;
        }
    }

    f32x4 acc[PAIRS][NTB];
#pragma unroll
    for (int i = 0; i < PAIRS; ++i)
#pragma unroll
        for (int t = 0; t < NTB; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dbsum[j] = 0.f;
    const bool want_db = ws_db != nullptr && chunk == 0;

    struct pre_t { halo_regs<T, UNITS, false, true> R; F braw[BUNITS]; uint2 bkeep[BUNITS]; unsigned bok; int n; };
    pre_t S;
    f32x2 ba[4], bb2[4];

    auto issue = [&](pre_t& Q, long tile) __attribute__((always_inline)) {
        int n, z0, y0, x0;
        tile_coords<G::TH, G::TW, 1>(tile, tiles_x, tiles_y, 1, n, z0, y0, x0);
        Q.n = n;
        halo_issue_impl<T, G, false, 1, KC, false, UNITS, false, false>(Q.R, U, s0, s1, 1, P.IH, P.IW, n, 0, y0, x0, chunk);
        const long gp0 = ((long)n * P.H + y0) * P.W + x0;
        const char* bb = (const char*)sb.ptr + (gp0 * sb.ld + sb.coff + cbB_safe) * (long)sizeof(T);
        const unsigned ldb = sb.ld * sizeof(T);
        unsigned r[BUNITS];
        Q.bok = 0;
#pragma unroll
        for (int j = 0; j < BUNITS; ++j) {
            const int d = b_yx[j];
            const bool ok = d >= 0 && bchan_ok && y0 + (d >> 8) < P.H && x0 + (d & 255) < P.W;
            Q.bok |= ok ? (1u << j) : 0u;
            r[j] = ok ? (unsigned)b_rel[j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < BUNITS; ++j) Q.braw[j] = frag<T>::load((const T*)(bb + __umul24(r[j], ldb)));
        if (sb.has_keep) {
            const uint8_t* kb = sb.keep + gp0 * sb.C + cbB_safe;
#pragma unroll
            for (int j = 0; j < BUNITS; ++j) Q.bkeep[j] = *(const uint2*)(kb + __umul24(r[j], (unsigned)sb.C));
        }
    };
    auto commit = [&](const pre_t& Q) __attribute__((always_inline)) {
        halo_commit_impl<T, KC, false, UNITS, false, true, false, WP_AFFC>(Q.R, U, halo, s0, s1, aff, plainA, Q.n, chunk);
        f32x2 a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = ba[k]; b[k] = bb2[k]; }
        if (sb.has_cm) {
            float bcm[8];
            ld8(sb.chan_mul + (long)Q.n * sb.C + cbB_safe, bcm);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const f32x2 m = {bcm[2 * k], bcm[2 * k + 1]}; a[k] *= m; b[k] *= m; }
        }
#pragma unroll
        for (int j = 0; j < BUNITS; ++j) {
            T* dst = bt + b_lds[j];
            const bool ok = (Q.bok >> j) & 1u;
            if (plainB) {
                if (want_db) {
                    float v[8];
                    frag<T>::unpack(Q.braw[j], v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) dbsum[k] += ok ? v[k] : 0.f;
                }
                frag<T>::store(dst, frag<T>::select(ok, Q.braw[j]));
                continue;
            }
            float v[8];
            frag<T>::unpack(Q.braw[j], v);
            affine_act8(v, a, b, sb.slope_eff);
            if (sb.has_keep) keep8(v, Q.bkeep[j], sb.keep_scale);
            if (want_db) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dbsum[k] += ok ? v[k] : 0.f;
            }
            frag<T>::store(dst, frag<T>::select(ok, frag<T>::pack(v)));
        }
    };

    // MFMA fragment offsets (elements): lane part; the (tap, 16-channel group) part of pair i is a compile-time constant
    const int q = lane & 15, qq = q >> 2, pp = q & 3;
    const int rowl = g >> 1, xb = 8 * (g & 1);
    const int a_lane = (rowl * G::HW + xb + qq) * PS + 4 * pp;
    const int b_off = (rowl * G::TW + xb + qq) * PSB + 4 * pp;

    const long last_tile = t_lo + sj + (my_tiles > 0 ? my_tiles - 1 : 0) * bpx;
    auto tile_of = [&](long k) __attribute__((always_inline)) { const long t = t_lo + sj + k * bpx; return t < last_tile ? t : last_tile; };
    if (my_tiles > 0) issue(S, tile_of(0));
    __syncthreads();                                            // affine caches visible (the only block barrier before the end)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ba[k] = *(const f32x2*)(aff + 2 * WP_AFFC + cbB_safe + 2 * k);
        bb2[k] = *(const f32x2*)(aff + 2 * WP_AFFC + WP_AFFC / 2 + cbB_safe + 2 * k);
    }
    if (my_tiles > 0) commit(S);

    auto compute = [&]() __attribute__((always_inline)) {
        constexpr int NSTEP = NKCH * PAIRS, DEPTH = NSTEP < WG_DEPTH ? NSTEP : WG_DEPTH;
        auto load_a = [&](int st) __attribute__((always_inline)) -> uint4 {
            const int kc = st / PAIRS, i = st % PAIRS;
            const int tap = i / KCT, kct = i % KCT;
            const int dx = tap % 3, dy = tap / 3;
            const bf16_t* p = (const bf16_t*)halo + a_lane + (kc * 2 + dy) * (G::HW * PS) + dx * PS + kct * 16;
            const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * PS);
            return __builtin_bit_cast(uint4, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
        };
        auto load_b = [&](int kc, uint4 (&bf)[NTB]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                const bf16_t* bp = (const bf16_t*)bt + b_off + kc * (2 * G::TW * PSB) + t * 16;
                const s16x4 lo = lds_tr16(bp), hi = lds_tr16(bp + 4 * PSB);
                bf[t] = __builtin_bit_cast(uint4, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            }
        };
        uint4 bf[2][NTB], af[DEPTH];
        load_b(0, bf[0]);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) af[d] = load_a(d);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int kc = st / PAIRS, i = st % PAIRS;
            if (i == 0 && kc + 1 < NKCH) load_b(kc + 1, bf[(kc + 1) & 1]);
            const uint4 a = af[st % DEPTH];
#pragma unroll
            for (int t = 0; t < NTB; ++t) mma8(acc[i][t], a, bf[kc & 1][t]);
            if (st + DEPTH < NSTEP) af[st % DEPTH] = load_a(st + DEPTH);
        }
    };
    // tile k is in this wave's LDS region: request tile k + 1 (registers), compute tile k, then land tile k + 1 over it
    if (my_tiles > 0) {
        for (long k = 0;; ++k) {
            issue(S, tile_of(k + 1));
            compute();
            if (k + 1 >= my_tiles) break;                       // (leave BEFORE the commit: see conv_fwd_kernel)
            commit(S);
        }
    }
    // ---- the four pipelines meet: dW registers -> LDS (each wave into its own tile region) -> summed in a fixed order -> the block's slab
    float* slab = ws + (long)split * G::NTAPS * Ca * Cb;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
#pragma unroll
        for (int i0 = 0; i0 < PAIRS; i0 += RP) {
            __syncthreads();                                    // every wave is done with its tile region (first round) / with the previous round's sums
            float* red = (float*)(smem + wave * WBYTES);
#pragma unroll
            for (int i = i0; i < i0 + RP && i < PAIRS; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[((i - i0) * 4 + j) * 64 + lane] = acc[i][t][j];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < RP; ++m) {
                const int E = threadIdx.x + 256 * m;
                const int ln = E & 63, rest = E >> 6;
                const int j = rest & 3, i = i0 + (rest >> 2);
                if (i < PAIRS) {
                    const float v = (((const float*)(smem + 0 * WBYTES))[E] + ((const float*)(smem + 1 * WBYTES))[E]) + (((const float*)(smem + 2 * WBYTES))[E] + ((const float*)(smem + 3 * WBYTES))[E]);
                    const int tap = i / KCT, kct = i % KCT;
                    const int kn = nb * BN + t * 16 + (ln & 15);
                    const int kcg = chunk * KC + kct * 16 + 4 * (ln >> 4) + j;
                    if (kn < Cb) slab[((long)tap * Ca + kcg) * Cb + kn] = v;
                }
            }
        }
    }
    if (want_db) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = dbsum[j];
            for (int o = CG; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane < CG) dbred[wave * BN + lane * 8 + j] = v;
        }
        __syncthreads();
        if (threadIdx.x < BN && nb * BN + threadIdx.x < Cb)
            ws_db[(long)split * Cb + nb * BN + threadIdx.x] = (dbred[threadIdx.x] + dbred[BN + threadIdx.x]) + (dbred[2 * BN + threadIdx.x] + dbred[3 * BN + threadIdx.x]);
    }
}
