// 2D deep-layer convolution, K-chunks side by side, round-4 form (see conv_kpar2d_kernel).  Included by conv_kpar_bf16.hip.
#pragma once
#include "conv_kpar.h"

// The halos of the CPAR chunks of one round with ONE bounds test / offset per unit (round 4): the chunks of a round are the same halo pixels at
// channel offsets chunk * KC, so the per-unit VALU work of halo_issue_impl (packed bounds test, select, 24-bit multiply) is paid once and each
// further chunk is a load at another wave-uniform base.  Sources: one, or two concatenated ones with the chunk wholly inside one of them.
template <typename T, typename G, bool D3, int ST, int KC, int UNITS, int CPAR, bool ONE, bool KEEPM>
__device__ __forceinline__ void halo_issue_round(halo_regs<T, UNITS, false, KEEPM> (&R)[CPAR], const unit_desc<UNITS>& U, const src_scalars& s0, const src_scalars& s1,
                                                 int ID, int IH, int IW, int n, int z0, int y0, int x0, int chunk0) {
    const int gz0 = z0 * G::STD - (D3 ? G::PAD : 0), gy0 = y0 * ST - G::PAD, gx0 = x0 * ST - G::PAD;
    const long gp0 = (((long)n * ID + gz0) * IH + gy0) * IW + gx0;
    const int loz = max(0, -gz0), hiz = min(G::HD, ID - gz0), loy = max(0, -gy0), hiy = min(G::HH, IH - gy0), lox = max(0, -gx0), hix = min(G::HW, IW - gx0);
    const unsigned PA = ((unsigned)(512 - loz) << 20) | ((unsigned)(512 - loy) << 10) | (unsigned)(512 - lox);
    const unsigned PB = ((unsigned)(512 - hiz) << 20) | ((unsigned)(512 - hiy) << 10) | (unsigned)(512 - hix);
    const unsigned rsafe = ((D3 ? G::PAD : 0) * IH + G::PAD) * IW + G::PAD;
    unsigned okm = 0, r[UNITS];
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        const unsigned d = (unsigned)U.hzyx[j];
        const bool ok = (((d + PA) ^ (d + PB)) & UNIT_GUARD) == UNIT_GUARD;
        okm |= ok ? (1u << j) : 0u;
        r[j] = ok ? (unsigned)U.rel[j] : rsafe;
    }
    const unsigned lane_c = U.c8 * sizeof(T);
    unsigned off0[UNITS], off1[ONE ? 1 : UNITS];
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        off0[j] = __umul24(r[j], (unsigned)(s0.ld * sizeof(T))) + lane_c;
        if (!ONE) off1[j] = __umul24(r[j], (unsigned)(s1.ld * sizeof(T))) + lane_c;
    }
#pragma unroll
    for (int c = 0; c < CPAR; ++c) {
        const int cb = (chunk0 + c) * KC;
        const bool second = ONE ? false : (cb >= s0.C);
        const int cs = second ? cb - s0.C : cb;
        const char* base = (const char*)(second ? s1.ptr : s0.ptr) + (gp0 * (second ? s1.ld : s0.ld) + (second ? s1.coff : s0.coff) + cs) * (long)sizeof(T);
        R[c].ok = okm;
#pragma unroll
        for (int j = 0; j < UNITS; ++j) R[c].raw[j] = frag<T>::load((const T*)(base + ((!ONE && second) ? off1[ONE ? 0 : j] : off0[j])));
        if constexpr (KEEPM) {
            const bool hk = second ? s1.has_keep : s0.has_keep;
            if (!hk) {
#pragma unroll
                for (int j = 0; j < UNITS; ++j) R[c].keep[j] = make_uint2(0u, 0u);
            } else {
                const int kC = second ? s1.C : s0.C;
                const uint8_t* kb = (second ? s1.keep : s0.keep) + (gp0 * kC + cs);
#pragma unroll
                for (int j = 0; j < UNITS; ++j) R[c].keep[j] = *(const uint2*)(kb + (__umul24(r[j], (unsigned)kC) + (unsigned)U.c8));
            }
        }
    }
}

// 2D deep layers (64..256 input channels at 64 x 64 and below), bf16, 32-channel K-chunks: conv_kpar_kernel's scheme -- the K-chunks of a tile side
// by side, wave w runs chunk (w % CPAR) -- rebuilt around what its phase stamps showed (tools/lab/conv_lab.hip p, round 4; 128->128 at 32 x 32, N = 12:
// 9.4 us of block life = setup 1.45 + issue 1.36 + commit 1.5 + taps 2.27 + reduction 0.5 + stores 1.25 + statistics 0.9):
//   * prologue: every global load of the block's start (scale / shift, the round's halos, ALL of the wave's weight fragments) is issued before the
//     first value is used -- the old form spent 1.45 us in dependent small loads in front of the halo loads and asked for the weight fragments only
//     behind the second barrier (they come from MALL / HBM inside the iteration: the packed weights are rewritten every step);
//   * issue: one bounds test / offset per halo unit for all chunks of the round (halo_issue_round) instead of one per chunk;
//   * taps: WALL -- the 18 weight fragments of the chunk sit in registers (no memory latency in the loop); XREUSE -- the pixel fragment of tap (dy, dx) for
//     output row m is halo row m + dy at column dx, the same for every (m, dy) with equal m + dy: walking the taps dx-major a wave reads RW + 2 row
//     fragments per dx for 3 * RW MFMA rows -- 30 LDS reads per chunk instead of 72 (the loop was LDS-read bound at twice its MFMA time).
// SINGLE (one round: <= 128 input channels): the accumulators are not live while the halos are staged, which is what leaves room for the weight
// fragments beside them; layers with two rounds (256 channels) request each round's weight fragments behind its staging.
// Same operands, statistics slots and output conventions as conv_kpar_kernel; conv_api.hip routes (2D, k3 s1, 32-channel chunks, plain output).
template <int NT, int CPAR, bool ONE, bool KEEPM, bool SINGLE>
__device__ __forceinline__ void conv_kpar2d_kernel(const chap_conv_params& P) {      // runs behind chap_grouped<.., 256, 2> (launch.h)
    typedef bf16_t T;
    constexpr int KS = 3, ST = 1, KC = 32;
    constexpr bool D3 = false;
    typedef conv_geom<KS, ST, false, 2> G;                      // tile = 8 x 16 pixels
    typedef typename frag<T>::type F;
    constexpr int ROWS = G::TH, RW = ROWS * CPAR / 4, FR = ROWS / 4;
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>();
    constexpr int STEPS = G::NTAPS;                             // GPT == 4: one MFMA step = one tap, lane group g = 8-channel group
    constexpr int UNITS = (G::HP * GPT + 255) / 256;
    constexpr size_t HB = (size_t)G::HP * PS + HALO_DUMMY;
    static_assert((CPAR == 2 || CPAR == 4) && STEPS * NT <= 18, "conv_kpar2d: 2 or 4 chunks side by side, all weight fragments of a chunk in registers");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CHAP_KSTAMP(0);
    T* halo = (T*)smem;
    f32x4* red = (f32x4*)smem;
    constexpr size_t HALOS = CPAR * HB * sizeof(T), RED = (size_t)4 * RW * NT * 64 * sizeof(f32x4);
    float* bstat = (float*)(smem + (HALOS > RED ? HALOS : RED));
    float* aff = bstat + 4 * 2 * 16 * NT;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const int tiles_x = (P.W + G::TW - 1) / G::TW, tiles_y = (P.H + G::TH - 1) / G::TH;
    const long ntiles = (long)P.N * P.D * tiles_y * tiles_x;
    const int ntiles_total = (P.Cout + 15) >> 4;
    const int nt0 = blockIdx.y * NT;
    const int Ctot = P.src[0].C + ((!ONE && P.nsrc > 1) ? P.src[1].C : 0);
    const int nrounds = SINGLE ? 1 : Ctot / (KC * CPAR);
    const bool plain = P.src[0].scale == nullptr && !P.src[0].act && P.src[0].keep == nullptr && P.src[0].chan_mul == nullptr &&
                       (ONE || P.nsrc < 2 || (P.src[1].scale == nullptr && !P.src[1].act && P.src[1].keep == nullptr && P.src[1].chan_mul == nullptr));
    const src_scalars s0 = make_scalars(P.src[0]);
    const src_scalars s1 = make_scalars((!ONE && P.nsrc > 1) ? P.src[1] : P.src[0]);
    const bool do_stats = P.stats != nullptr;
    const int ck = wave % CPAR, rgrp = wave / CPAR;
    // ---- the prologue's small loads first (they are the oldest in the queue when their values are needed)
    float asc[2][2], ash[2][2];
    if (!plain) {
#pragma unroll
        for (int s_ = 0; s_ < (ONE ? 1 : 2); ++s_) {
            const chap_src_t& S = P.src[s_ < P.nsrc ? s_ : 0];
            const bool has = S.scale != nullptr;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = min((int)threadIdx.x + 256 * k, S.C - 1);
                asc[s_][k] = 1.f; ash[s_][k] = 0.f;
                if (has) { asc[s_][k] = S.scale[c]; ash[s_][k] = S.shift[c]; }      // wave-uniform branch
            }
        }
    }
    unit_desc<UNITS> U;
    make_units<G, GPT, PS, UNITS>(U, P.IH, P.IW);
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[t][j] = 0.f; ssq[t][j] = 0.f; }
    const long wstep = (long)ntiles_total * 64 * 8;             // packed elements per (chunk, step)
    const int orow = P.W * P.out_ld;
    int ooff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ooff[t] = px * P.out_ld + P.out_coff + (nt0 + t) * 16 + 4 * g;

    CHAP_KSTAMP(1);
    bool first = true;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int n, z0, y0, x0;
        tile_coords<G::TH, G::TW, 1>(tile, tiles_x, tiles_y, P.D, n, z0, y0, x0);
        f32x4 acc[RW][NT];
        float bj[NT][4], cj[NT][4];                             // bias / statistics shift: requested behind the last round's staging, they land during its tap loop
        if constexpr (!SINGLE) {
#pragma unroll
            for (int m = 0; m < RW; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int r = 0; r < nrounds; ++r) {
            halo_regs<T, UNITS, false, KEEPM> R[CPAR];
            halo_issue_round<T, G, D3, ST, KC, UNITS, CPAR, ONE, KEEPM>(R, U, s0, s1, P.ID, P.IH, P.IW, n, z0, y0, x0, r * CPAR);
            const int chunk = r * CPAR + ck;
            const T* wc_g = (const T*)P.wpacked + (long)chunk * STEPS * wstep + ((long)nt0 * 64 + lane) * 8;
            F wall[STEPS][NT];
            // the fragments of tap column dx (taps dx, 3 + dx, 6 + dx): columns 0 and 1 are requested next to the halos, column 2 behind the staging
            // (it lands while columns 0 and 1 are computed) -- 48 instead of 72 VGPRs live across the commit phase, which spilled with all three
            auto load_wcol = [&](int dx) __attribute__((always_inline)) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int t = 0; t < NT; ++t) wall[dy * 3 + dx][t] = frag<T>::load(wc_g + (long)(dy * 3 + dx) * wstep + (nt0 + t < ntiles_total ? t : 0) * 512);   // clamped: those channels are never stored
            };
            if constexpr (SINGLE) { load_wcol(0); load_wcol(1); }
            if (first) {                                        // (block-uniform) the scale / shift loads of the prologue land in the LDS cache
                first = false;
                if (!plain) {
#pragma unroll
                    for (int s_ = 0; s_ < (ONE ? 1 : 2); ++s_) {
                        if (s_ < P.nsrc) {
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const int c = threadIdx.x + 256 * k;
                                if (c < P.src[s_].C) { aff[s_ * CONV_MAX_AFFINE_C + c] = asc[s_][k]; aff[s_ * CONV_MAX_AFFINE_C + CONV_MAX_AFFINE_C / 2 + c] = ash[s_][k]; }
                            }
                        }
                    }
                }
            }
            CHAP_KSTAMP(2);
            __syncthreads();                                    // the LDS buffers are free (previous round / tile); first: the scale / shift cache is written
            CHAP_KSTAMP(3);
#pragma unroll
            for (int c = 0; c < CPAR; ++c)
                halo_commit<T, KC, false, UNITS, KEEPM, ONE>(R[c], U, halo + c * HB, s0, s1, aff, plain, n, r * CPAR + c, false);
            CHAP_KSTAMP(4);
            __syncthreads();
            CHAP_KSTAMP(5);
            __builtin_amdgcn_sched_barrier(0);                  // (keep the accumulators' zeroing and the tap loop's loads behind the staging: its registers are all taken)
            if (r == nrounds - 1) {
                const float* bsrc = P.bias ? P.bias : (const float*)P.wpacked;      // any valid address when there is no bias: no branch around the loads
                const bool has_shift = do_stats && P.stats_shift != nullptr;
                const float* csrc = has_shift ? P.stats_shift : (const float*)P.wpacked;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int nl = (nt0 + t) * 16 + 4 * g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bj[t][j] = bsrc[(P.bias && nl + j < P.Cout) ? nl + j : 0];
                        cj[t][j] = csrc[(has_shift && nl + j < P.Cout) ? nl + j : 0];
                    }
                }
            }
            if constexpr (SINGLE) {
                load_wcol(2);
#pragma unroll
                for (int m = 0; m < RW; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            } else {
                load_wcol(0); load_wcol(1); load_wcol(2);
            }
            // ---- tap loop of chunk r*CPAR + ck over this wave's RW rows, dx-major (XREUSE)
            const T* hx = halo + ck * HB + (size_t)(rgrp * RW) * (G::HW * PS) + px * PS + g * 8;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                F xr[RW + 2];
#pragma unroll
                for (int rr = 0; rr < RW + 2; ++rr) xr[rr] = frag<T>::load(hx + (rr * G::HW + dx) * PS);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int m = 0; m < RW; ++m)
#pragma unroll
                        for (int t = 0; t < NT; ++t) mma8(acc[m][t], wall[dy * 3 + dx][t], xr[m + dy]);
            }
        }
        // ---- the CPAR partial sums of every row meet in LDS; wave f finishes rows [f*FR, f*FR + FR)
        CHAP_KSTAMP(6);
        __syncthreads();                                        // every wave is done reading the halos
        CHAP_KSTAMP(7);
#pragma unroll
        for (int m = 0; m < RW; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) red[((wave * RW + m) * NT + t) * 64 + lane] = acc[m][t];
        __syncthreads();
        CHAP_KSTAMP(8);
        const bool xok = x0 + px < P.W;
        const long o0 = ((((long)n * P.D + z0) * P.H + y0) * P.W + x0) * P.out_ld;
        char* ob = (char*)P.out + o0 * (P.out_f32 ? 4 : (long)sizeof(T));
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            const int row = wave * FR + f;                      // tile row this wave finishes
            const int rg = row / RW, ml = row % RW;             // who computed it: waves rg*CPAR .. rg*CPAR + CPAR - 1, their local row ml
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 s = red[(((rg * CPAR + 0) * RW + ml) * NT + t) * 64 + lane];
#pragma unroll
                for (int c = 1; c < CPAR; ++c) {                 // fixed order: chunk 0 + chunk 1 (+ chunk 2 + chunk 3)
                    const f32x4 q = red[(((rg * CPAR + c) * RW + ml) * NT + t) * 64 + lane];
                    s[0] += q[0]; s[1] += q[1]; s[2] += q[2]; s[3] += q[3];
                }
                const bool valid = xok && (y0 + row < P.H) && (nt0 + t) * 16 + 4 * g < P.Cout;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = s[j] + ((P.bias && (nt0 + t) * 16 + 4 * g + j < P.Cout) ? bj[t][j] : 0.f);
                if (do_stats) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float vs = valid ? v[j] - (P.stats_shift ? cj[t][j] : 0.f) : 0.f; ssum[t][j] += vs; ssq[t][j] += vs * vs; }
                }
                const unsigned oi = (unsigned)(row * orow + ooff[t]);
                if (valid) {
                    if (P.out_f32) st4((float*)(ob + oi * 4u), v); else st4((T*)(ob + oi * (unsigned)sizeof(T)), v);
                }
            }
        }
    }
    CHAP_KSTAMP(9);
    // ---- BatchNorm statistics: this block's partial slot (conv_fwd_kernel's layout and order)
    if (do_stats) {
        __syncthreads();
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
            const int which = i / (16 * NT), k = i % (16 * NT);
            const int nl = nt0 * 16 + k;
            const float v = (bstat[(0 * 2 + which) * 16 * NT + k] + bstat[(1 * 2 + which) * 16 * NT + k]) + (bstat[(2 * 2 + which) * 16 * NT + k] + bstat[(3 * 2 + which) * 16 * NT + k]);
            if (nl < P.Cout) st[which * P.Cout + nl] = v;
        }
    }
    CHAP_KSTAMP(10);
}
