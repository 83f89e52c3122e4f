// "K-parallel" 3x3(x3) convolution for the DEEP, SMALL layers (64..512 input channels on 7^3 .. 64^2 grids), bf16.
//
// conv_fwd_kernel gives such a layer one wave of blocks, and a block walks the K-chunks of its only tile one after the
// other: prologue + nchunks x (issue, tap loop, commit, barrier) -- 12-22 us of block life for 1.5-3.5 GFLOP
// (tools/lab/conv_lab.hip d: 128->128 at 32x32 = 3.0 us prologue + 4 x 1.9 us + flush).  Here the K-chunks of a tile run SIDE BY
// SIDE: the halos of CPAR chunks are fetched in ONE burst and staged into CPAR LDS buffers, then wave w runs the tap loop of
// chunk (w % CPAR) for the rows of row group (w / CPAR) -- its weight fragments stream straight from L2 in fragment order, every
// wave a different chunk, so no wave loads what another one loads -- and the CPAR partial accumulators of a row meet in LDS, in
// a fixed order (deterministic), where each wave finishes a quarter of the tile's rows: bias, BatchNorm statistics, store.
// A tile costs one memory round trip + one tap loop per ROUND of CPAR chunks instead of one per chunk.
//
// Same operands, packed-weight layout, lazy-activation prologue (BN affine, LeakyReLU, Dropout keep masks, Dropout3d channel
// multipliers, two concatenated sources), statistics slots and bit-level output conventions as conv_fwd_kernel; built for the
// plain channel-last output (no depth-to-space, no planar logits, no skip add) -- conv_api.hip routes everything else to
// conv_fwd_kernel.
#pragma once
#include "conv_kernel.h"

#ifdef CHAP_CONV_TRACE         // lab only (tools/lab/conv_lab.hip p): phase stamps of every wave of four blocks
__device__ unsigned long long chap_trace_k[4][4][16];
#define CHAP_KSTAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 4) chap_trace_k[blockIdx.x / 97][threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CHAP_KSTAMP(k) do {} while (0)
#endif

template <typename T, bool D3, int KC, int CPAR>
__host__ __device__ constexpr size_t conv_kpar_lds_bytes(int NT) {
    typedef conv_geom<3, 1, D3, D3 ? 1 : 2> G;
    const size_t halos = CPAR * ((size_t)G::HP * pix_stride<T, KC>() + HALO_DUMMY) * sizeof(T);
    const size_t red = (size_t)4 * (G::TH * CPAR / 4) * NT * 64 * sizeof(f32x4);         // partial accumulators (aliases the halos)
    return (halos > red ? halos : red) + 4 * 2 * 16 * NT * sizeof(float) + 2 * CONV_MAX_AFFINE_C * sizeof(float);
}

template <typename T, bool D3, int KC, int NT, int CPAR, bool ONE>
__device__ __forceinline__ void conv_kpar_kernel(const chap_conv_params& P) {      // runs behind chap_grouped<.., 256, 2> (launch.h)
    constexpr int KS = 3, ST = 1, MRG = D3 ? 1 : 2;
    typedef conv_geom<KS, ST, D3, MRG> G;                        // tile = ROWS x 16 pixels of one plane (2D: 8 rows, 3D: 4)
    typedef typename frag<T>::type F;
    constexpr int ROWS = G::TH, RW = ROWS * CPAR / 4, FR = ROWS / 4;     // rows per wave in the tap loop / rows a wave finishes
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>();
    constexpr int NP = G::NTAPS * GPT, STEPS = (NP + 3) / 4;
    constexpr int UNITS = (G::HP * GPT + 255) / 256;
    constexpr size_t HB = (size_t)G::HP * PS + HALO_DUMMY;      // elements per halo buffer
    static_assert(sizeof(T) == 2 && (CPAR == 2 || CPAR == 4) && ROWS % 4 == 0, "conv_kpar: bf16, 2 or 4 chunks side by side");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CHAP_KSTAMP(0);
    T* halo = (T*)smem;
    f32x4* red = (f32x4*)smem;                                  // [wave][RW][NT][64 lanes], after the tap loops
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
    const int nrounds = Ctot / (KC * CPAR);
    const bool plain = P.src[0].scale == nullptr && !P.src[0].act && P.src[0].keep == nullptr && P.src[0].chan_mul == nullptr &&
                       (ONE || P.nsrc < 2 || (P.src[1].scale == nullptr && !P.src[1].act && P.src[1].keep == nullptr && P.src[1].chan_mul == nullptr));
    const src_scalars s0 = make_scalars(P.src[0]);
    const src_scalars s1 = make_scalars((!ONE && P.nsrc > 1) ? P.src[1] : P.src[0]);
    const bool do_stats = P.stats != nullptr;

    // ---- one-time per thread
    unit_desc<UNITS> U;
    make_units<G, GPT, PS, UNITS>(U, P.IH, P.IW);
    int xoff[STEPS];                                            // LDS element offset of this lane's pixel fragment (row 0), -1 = zero fragment
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int p = step * 4 + g;
        const int tap = p / GPT, cgl = p % GPT;
        const int dx = tap % KS, dy = (tap / KS) % KS, dz = tap / (KS * KS);
        xoff[step] = p < NP ? ((dz * G::HH + dy) * G::HW + px + dx) * PS + cgl * 8 : -1;
    }
    const int ck = wave % CPAR, rgrp = wave / CPAR;             // this wave's chunk of a round, its row group in the tap loop
    // scale/shift of both sources -> LDS cache (conv_fwd_kernel's layout)
    if (!plain) {
        for (int s = 0; s < (ONE ? 1 : 2); ++s) {
            if (s < P.nsrc) {
                const chap_src_t& S = P.src[s];
                const bool has = S.scale != nullptr;
                for (int c = threadIdx.x; c < S.C && c < CONV_MAX_AFFINE_C / 2; c += 256) {
                    aff[s * CONV_MAX_AFFINE_C + c] = has ? S.scale[c] : 1.f;
                    aff[s * CONV_MAX_AFFINE_C + CONV_MAX_AFFINE_C / 2 + c] = has ? S.shift[c] : 0.f;
                }
            }
        }
    }
    float ssum[NT][4], ssq[NT][4], bj[NT][4], cj[NT][4];
    {
        const bool has_shift = do_stats && P.stats_shift != nullptr;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nl = (nt0 + t) * 16 + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ssum[t][j] = 0.f; ssq[t][j] = 0.f;
                bj[t][j] = (P.bias && nl + j < P.Cout) ? P.bias[nl + j] : 0.f;
                cj[t][j] = (has_shift && nl + j < P.Cout) ? P.stats_shift[nl + j] : 0.f;
            }
        }
    }
    const long wstep = (long)ntiles_total * 64 * 8;             // packed elements per (chunk, step)
    const int orow = P.W * P.out_ld;
    int ooff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ooff[t] = px * P.out_ld + P.out_coff + (nt0 + t) * 16 + 4 * g;

    CHAP_KSTAMP(1);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int n, z0, y0, x0;
        tile_coords<G::TH, G::TW, 1>(tile, tiles_x, tiles_y, P.D, n, z0, y0, x0);
        f32x4 acc[RW][NT];
#pragma unroll
        for (int m = 0; m < RW; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < nrounds; ++r) {
            // ---- the halos of the round's CPAR chunks: every load first (in flight across the barrier), then the transforms
            halo_regs<T, UNITS, false, !D3> R[CPAR];
#pragma unroll
            for (int c = 0; c < CPAR; ++c)
                halo_issue<T, KS, ST, D3, KC, MRG, false, UNITS, false, ONE>(R[c], U, s0, s1, P.ID, P.IH, P.IW, n, z0, y0, x0, r * CPAR + c, false);
            const int chunk = r * CPAR + ck;
            const T* wc_g = (const T*)P.wpacked + (long)chunk * STEPS * wstep + ((long)nt0 * 64 + lane) * 8;
            auto load_w = [&](int step, F (&wf)[NT]) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[t] = frag<T>::load(wc_g + (long)step * wstep + (nt0 + t < ntiles_total ? t : 0) * 512);   // clamped: those channels are never stored
            };
            CHAP_KSTAMP(2);
            __syncthreads();                                    // the LDS buffers are free (previous round's tap loops / previous tile's reduction; first: the scale/shift cache is written)
            CHAP_KSTAMP(3);
#pragma unroll
            for (int c = 0; c < CPAR; ++c)
                halo_commit<T, KC, false, UNITS, !D3, ONE>(R[c], U, halo + c * HB, s0, s1, aff, plain, n, r * CPAR + c, false);
            CHAP_KSTAMP(4);
            __syncthreads();
            CHAP_KSTAMP(5);
            // ---- tap loop of chunk r*CPAR + ck over this wave's RW rows; weight fragments stream from L2 two steps ahead
            const T* hb = halo + ck * HB + (size_t)(rgrp * RW) * (ST * G::HW * PS);
            auto load_x = [&](int step, F (&xf)[RW]) __attribute__((always_inline)) {
                const int xo = xoff[step];
#pragma unroll
                for (int m = 0; m < RW; ++m) {
                    if (step * 4 + 3 < NP) xf[m] = frag<T>::load(hb + xo + m * (ST * G::HW * PS));
                    else xf[m] = xo >= 0 ? frag<T>::load(hb + xo + m * (ST * G::HW * PS)) : frag<T>::zero();
                }
            };
            // ring depth of the weight fragments: a step is RW*NT MFMAs (16 cycles each), an L2 round trip ~700 cycles
            constexpr int WD0 = 2 + 700 / (RW * NT * 16), WD = WD0 > 8 ? 8 : (WD0 > STEPS ? STEPS : WD0);
            F wq[WD][NT], xq[2][RW];
#pragma unroll
            for (int d = 0; d < WD - 1; ++d) load_w(d, wq[d]);
            load_x(0, xq[0]);
#pragma unroll
            for (int step = 0; step < STEPS; ++step) {
                if (step + WD - 1 < STEPS) load_w(step + WD - 1, wq[(step + WD - 1) % WD]);
                if (step + 1 < STEPS) load_x(step + 1, xq[(step + 1) & 1]);
#pragma unroll
                for (int m = 0; m < RW; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) mma8(acc[m][t], wq[step % WD][t], xq[step & 1][m]);
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
                for (int j = 0; j < 4; ++j) v[j] = s[j] + bj[t][j];
                if (do_stats) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float vs = valid ? v[j] - cj[t][j] : 0.f; ssum[t][j] += vs; ssq[t][j] += vs * vs; }
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
