// Weight gradient as an implicit GEMM with the PIXEL index as the reduction dimension:
//     dW[tap][kc][kn] = sum_p  A[s*p + tap - pad][kc] * B[p][kn]
// MFMA roles: rows = kc (16 input channels of the halo-tiled operand A), cols = kn (16 channels of
// B), k = pixels.  Both operands are channel-last, i.e. k runs along the SLOW memory axis, so
//   * bf16: fragments come from `ds_read_b64_tr_b16` (gfx950 transposing LDS read: a 4-pixel x
//           16-channel block is delivered channel-major, two reads = the 8 k-values of a lane);
//   * fp32: 16x16x4 MFMA takes one k-value per lane -> plain ds_read_b32 of [pixel 4s+g][channel].
// A block owns one (KC-channel chunk of A) x (32-channel chunk of B) pair and one pixel split; it
// walks its TH x 16 pixel tiles keeping dW in registers (each wave owns a subset of (tap, 16-row)
// pairs), then writes ONE fp32 partial slab.  A second kernel sums the slabs in fixed order
// (deterministic, no float atomics) and accumulates into the checkpoint-layout gradient.
#pragma once
#include "conv_kernel.h"

constexpr int WG_BN = 32;                       // B channels per block (16 for layers with <= 16 output channels)
constexpr int WG_BDUMMY = 8;                    // elements behind each B tile: store target of threads without a B unit
#ifndef CHAP_WGRAD_DEPTH
#define CHAP_WGRAD_DEPTH 6
#endif
constexpr int WG_DEPTH = CHAP_WGRAD_DEPTH;      // A fragments (transposing LDS reads) in flight ahead of the MFMAs
#ifndef CHAP_WGRAD_PSB16
#define CHAP_WGRAD_PSB16 24
#endif
// LDS pixel stride of the B tile (elements): 32 (16) channels + padding for the transposing reads
// Round 4 (with conv_min_waves, conv_kernel.h): the bf16 block-tile instances are bounded to 256 registers per lane (unbounded, the 3D bricks with a 32-wide B
// tile took 277-370 incl. accumulation registers: one wave per SIMD) and the 32-wide B tile is stored without row padding (stride 32 instead of 40 elements:
// 80 960 instead of 89 152 B of LDS per brick block) -- two blocks per CU where one was.  3D step 14.08 -> 13.96 ms (three pairs), 2D unchanged
// (profiles/r04_wgrad3d_two_blocks_ab.log, r04_wgrad_two_blocks_2d_ab.log); 512 instead of 256 blocks for these layers: slower (14.32).
#ifndef CHAP_WGRAD_PSB32
#define CHAP_WGRAD_PSB32 32        // row stride (elements) of the 32-wide bf16 B tile (40 = padded, rounds 1-3)
#endif
#ifndef CHAP_WGRAD_MINW
#define CHAP_WGRAD_MINW 2          // __launch_bounds__ second argument of the bf16 block-tile weight-gradient instances (1 = unbounded, rounds 1-3)
#endif
template <typename T, int BN = WG_BN> __host__ __device__ constexpr int wg_psb() { return sizeof(T) == 2 ? (BN == 16 ? CHAP_WGRAD_PSB16 : CHAP_WGRAD_PSB32) : 36; }
// scale/shift cache per source: the brick kernels run up to 256 channels per source and keep it small (LDS for a second block per CU)
template <bool ZW> __host__ __device__ constexpr int wg_affc() { return ZW ? 512 : CONV_MAX_AFFINE_C; }

template <typename T, int KS, int ST, bool D3, int KC, int MR, int BN = WG_BN, bool ZW = false>
__host__ __device__ constexpr size_t wgrad_lds_bytes() {
    typedef conv_geom<KS, ST, D3, MR, ZW> G;
    return 2 * ((size_t)G::HP * pix_stride<T, KC>() + HALO_DUMMY + (size_t)G::TD * G::TH * G::TW * wg_psb<T, BN>() + WG_BDUMMY) * sizeof(T)    // double-buffered A halo + B tile
           + 3 * wg_affc<ZW>() * sizeof(float) + 4 * BN * sizeof(float);                                   // affine caches (A0, A1, B), db partials
}

__device__ __forceinline__ s16x4 lds_tr16(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

// Persistent blocks, one (A-chunk, B-chunk, split) each: the A halo and the B tile of tile i+1 are
// fetched into registers while tile i runs on the MFMA pipe out of LDS (same scheme as conv_fwd_kernel).
// PD = prefetch distance in tiles: the register sets of PD tiles are in flight while one tile is computed from LDS.  A tile
// of the 16-channel layers is a few KB, so one tile ahead (PD = 1) leaves the kernel latency-bound (0.9-1.5 TB/s); the ring
// costs 12 VGPRs per extra tile there.  Loads are unconditional (the tile index is clamped: the tail re-reads the block's
// last tile from L2) so that the compiler can count the waits instead of draining the queue (see halo_issue).
// ZW (3D, bf16): the tile is a TD x TH x 16 = 4 x 4 x 16 BRICK (conv_geom's z-per-wave geometry; here the waves split the
// (tap, 16-row) pairs as always, the brick only changes which pixels a tile holds): its halo is 6 x 6 x 18 = 2.5 A-pixels per
// output pixel where the 1 x 4 x 16 slab stages 3 x 6 x 18 = 5.1, and one barrier / prefetch round trip serves 256 pixels
// instead of 64.
// argument block of one launch (launch.h): base_z = the kernel's own grid.z (B-channel chunks); a grouped launch stacks groups along z
struct wgrad_args { chap_wgrad_params P; float* ws; float* ws_db; int nsplit, Ca, Cb, base_z; };

template <typename T, int KS, int ST, bool D3, int KC, int MR, bool ADD2, int BN = WG_BN, int PD = 1, bool ZW = false>
__device__ __forceinline__ void wgrad_kernel(const wgrad_args& A, int nb) {      // runs behind chap_grouped_z<.., 256, 1>
    const chap_wgrad_params& P = A.P;
    float* __restrict__ ws = A.ws;
    float* __restrict__ ws_db = A.ws_db;
    const int nsplit = A.nsplit, Ca = A.Ca, Cb = A.Cb;
    typedef conv_geom<KS, ST, D3, MR, ZW> G;
    constexpr int TPX = G::TD * G::TH * G::TW;           // pixels per tile
    static_assert(!ZW || (sizeof(T) == 2 && D3 && TPX % 32 == 0 && G::TH % 2 == 0), "brick tiles: 3D bf16 only");
    typedef typename frag<T>::type F;
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>(), PSB = wg_psb<T, BN>(), AFFC = wg_affc<ZW>();
    constexpr int KCT = KC / 16, NTB = BN / 16, CG = BN / 8;     // B: 16-channel MFMA tiles, 8-channel staging groups
    constexpr int PAIRS = G::NTAPS * KCT, MAXP = (PAIRS + 3) / 4;
    constexpr int NKCH = TPX / 32;                      // 32-pixel k-chunks per tile (two rows of 16)
    constexpr int UNITS = (G::HP * GPT + 255) / 256;
    constexpr int BUNITS = (TPX * CG + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo0 = (T*)smem;
    T* halo1 = halo0 + (size_t)G::HP * PS + HALO_DUMMY;
    T* bt0 = halo1 + (size_t)G::HP * PS + HALO_DUMMY;
    T* bt1 = bt0 + (size_t)TPX * PSB + WG_BDUMMY;
    float* aff = (float*)(bt1 + (size_t)TPX * PSB + WG_BDUMMY);       // [A0 | A1 | B] x [scale | shift]
    float* dbred = aff + 3 * AFFC;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int split = blockIdx.x, chunk = blockIdx.y;
    const int tiles_x = (P.W + G::TW - 1) / G::TW, tiles_y = (P.H + G::TH - 1) / G::TH, tiles_z = (P.D + G::TD - 1) / G::TD;
    const long ntiles = (long)P.N * tiles_z * tiles_y * tiles_x;
    // XCD-aware tile assignment (round 3): blocks (= splits) s and s + 8 run on the same XCD; that XCD owns a contiguous eighth of the tiles
    // and its blocks take them round-robin, so that the tiles in flight on one XCD are neighbours whose halo rows hit in its L2.  With the
    // plain `split + k * nsplit` of rounds 1-2 neighbouring tiles ran on different XCDs and every tile fetched its whole halo past the L2:
    // 16->16 at 256x256, N = 12: 86.8 MB of L2-miss reads for 50.3 MB of operands (profiles/r03_pmc_traffic_kernels.json).
    const int NG = nsplit < 8 ? nsplit : 8;                                // tile groups (one per XCD once there are >= 8 splits)
    const int xcd = split % NG, sj = split / NG;
    const int bpx = (nsplit + NG - 1 - xcd) / NG;                          // splits of this group
    const long per = (ntiles + NG - 1) / NG;
    const long t_lo = per * xcd, t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
    const long my_tiles = (bpx > 0 && t_lo + sj < t_hi) ? (t_hi - t_lo - sj + bpx - 1) / bpx : 0;

    const src_scalars s0 = make_scalars(P.a[0]);
    const src_scalars s1 = make_scalars(P.na > 1 ? P.a[1] : P.a[0]);
    const src_scalars sb = make_scalars(P.b);
    const bool plainA = !ADD2 && P.a[0].scale == nullptr && !P.a[0].act && P.a[0].keep == nullptr && P.a[0].chan_mul == nullptr &&
                        (P.na < 2 || (P.a[1].scale == nullptr && !P.a[1].act && P.a[1].keep == nullptr && P.a[1].chan_mul == nullptr));
    const bool plainB = P.b.scale == nullptr && !P.b.act && P.b.keep == nullptr && P.b.chan_mul == nullptr;

    // ---- one-time per thread: unit descriptors
    unit_desc<UNITS> U;
    make_units<G, GPT, PS, UNITS>(U, P.IH, P.IW);
    const bool lanesel = !ADD2 && P.na > 1 && (P.a[0].C % KC) != 0;
    int b_yx[BUNITS], b_rel[BUNITS], b_lds[BUNITS];     // B tile units: (plane << 16 | row << 8 | col), pixel offset, LDS offset; channel = nb*32 + (tid % 4) * 8
    const int bc8 = (threadIdx.x & (CG - 1)) * 8;
    const int cbB = nb * BN + bc8;
    const bool bchan_ok = cbB < Cb;
#pragma unroll
    for (int j = 0; j < BUNITS; ++j) {
        const int u = threadIdx.x + 256 * j;
        const int pix = u / CG;
        b_yx[j] = -1; b_rel[j] = 0; b_lds[j] = TPX * PSB;                // no such unit: loads the tile origin, stores to the dummy slot
        if (pix < TPX) {
            const int pz = pix / (G::TH * G::TW), row = (pix / G::TW) % G::TH, col = pix % G::TW;
            b_yx[j] = (pz << 16) | (row << 8) | col;
            b_rel[j] = (pz * P.H + row) * P.W + col;
            b_lds[j] = pix * PSB + bc8;
        }
    }
    // affine caches
#pragma unroll
    for (int s = 0; s < 3; ++s) {                                 // (unrolled: a runtime index into the argument block puts its pointers in scratch)
        const chap_src_t& src = s == 2 ? P.b : P.a[s < P.na ? s : 0];
        const bool has = src.scale != nullptr;
        for (int c = threadIdx.x; c < src.C && c < AFFC / 2; c += 256) {
            aff[s * AFFC + c] = has ? src.scale[c] : 1.f;
            aff[s * AFFC + AFFC / 2 + c] = has ? src.shift[c] : 0.f;
        }
    }

    f32x4 acc[MAXP][NTB];
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
#pragma unroll
        for (int t = 0; t < NTB; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dbsum[j] = 0.f;
    const bool want_db = ws_db != nullptr && chunk == 0;

    struct pre_t { halo_regs<T, UNITS, ADD2, !D3> R; F braw[BUNITS]; uint2 bkeep[BUNITS]; unsigned bok; int n; };
    pre_t S[PD];                                                // register sets of the tiles in flight (compile-time indexed)
    f32x2 ba[4], bb2[4];                                        // scale/shift of this thread's 8 B channels
    const int cbB_safe = bchan_ok ? cbB : 0;

    // all loads unconditional (out-of-range units read the tile origin and are zeroed at commit): see halo_issue
    auto issue = [&](pre_t& Q, long tile) __attribute__((always_inline)) {
        halo_regs<T, UNITS, ADD2, !D3>& R = Q.R;
        F (&braw)[BUNITS] = Q.braw; uint2 (&bkeep)[BUNITS] = Q.bkeep; unsigned& bok = Q.bok;
        int n, z0, y0, x0;
        tile_coords<G::TH, G::TW, G::TD>(tile, tiles_x, tiles_y, tiles_z, n, z0, y0, x0);
        Q.n = n;
        halo_issue<T, KS, ST, D3, KC, MR, ADD2, UNITS, ZW>(R, U, s0, s1, P.ID, P.IH, P.IW, n, z0, y0, x0, chunk, lanesel);
        const long gp0 = (((long)n * P.D + z0) * P.H + y0) * P.W + x0;
        const char* bb = (const char*)sb.ptr + (gp0 * sb.ld + sb.coff + cbB_safe) * (long)sizeof(T);
        const unsigned ldb = sb.ld * sizeof(T);
        unsigned r[BUNITS];
        bok = 0;
#pragma unroll
        for (int j = 0; j < BUNITS; ++j) {
            const int d = b_yx[j];
            const bool ok = d >= 0 && bchan_ok && (!ZW || z0 + (d >> 16) < P.D) && y0 + ((d >> 8) & 255) < P.H && x0 + (d & 255) < P.W;
            bok |= ok ? (1u << j) : 0u;
            r[j] = ok ? (unsigned)b_rel[j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < BUNITS; ++j) braw[j] = frag<T>::load((const T*)(bb + __umul24(r[j], ldb)));
        if (sb.has_keep) {
            const uint8_t* kb = sb.keep + gp0 * sb.C + cbB_safe;
#pragma unroll
            for (int j = 0; j < BUNITS; ++j) bkeep[j] = *(const uint2*)(kb + __umul24(r[j], (unsigned)sb.C));
        }
    };
    auto commit = [&](const pre_t& Q, T* halo, T* bt) __attribute__((always_inline)) {
        const halo_regs<T, UNITS, ADD2, !D3>& R = Q.R;
        const F (&braw)[BUNITS] = Q.braw; const uint2 (&bkeep)[BUNITS] = Q.bkeep; const unsigned bok = Q.bok; const int n = Q.n;
        halo_commit<T, KC, ADD2, UNITS, !D3, false, AFFC>(R, U, halo, s0, s1, aff, plainA, n, chunk, lanesel);
        f32x2 a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = ba[k]; b[k] = bb2[k]; }
        if (sb.has_cm) {
            float bcm[8];
            ld8(sb.chan_mul + (long)n * sb.C + cbB_safe, bcm);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const f32x2 m = {bcm[2 * k], bcm[2 * k + 1]}; a[k] *= m; b[k] *= m; }
        }
#pragma unroll
        for (int j = 0; j < BUNITS; ++j) {                      // threads without a unit stage zeros into the dummy slot
            T* dst = bt + b_lds[j];
            const bool ok = (bok >> j) & 1u;
            if (plainB) {
                if (want_db) {
                    float v[8];
                    frag<T>::unpack(braw[j], v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) dbsum[k] += ok ? v[k] : 0.f;
                }
                frag<T>::store(dst, frag<T>::select(ok, braw[j]));
                continue;
            }
            float v[8];
            frag<T>::unpack(braw[j], v);
            affine_act8(v, a, b, sb.slope_eff);
            if (sb.has_keep) keep8(v, bkeep[j], sb.keep_scale);
            if (want_db) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dbsum[k] += ok ? v[k] : 0.f;
            }
            frag<T>::store(dst, frag<T>::select(ok, frag<T>::pack(v)));
        }
    };

    // bf16 MFMA fragment offsets (elements), once per thread: lane part + (tap, 16-channel group) part of each pair
    int a_off[MAXP], b_off = 0;
    {
        const int q = lane & 15, qq = q >> 2, pp = q & 3;
        const int rowl = g >> 1, xb = 8 * (g & 1);
        b_off = (rowl * G::TW + xb + qq) * PSB + 4 * pp;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int pair = wave + 4 * i < PAIRS ? wave + 4 * i : 0;
            const int tap = pair / KCT, kct = pair % KCT;
            const int dx = tap % KS, dy = (tap / KS) % KS, dz = tap / (KS * KS);
            a_off[i] = ((dz * G::HH + rowl * ST + dy) * G::HW + (xb + qq) * ST + dx) * PS + kct * 16 + 4 * pp;
        }
    }
    const long last_tile = t_lo + sj + (my_tiles > 0 ? my_tiles - 1 : 0) * bpx;     // clamp target of the unconditional prefetch
    auto tile_of = [&](long k) __attribute__((always_inline)) { const long t = t_lo + sj + k * bpx; return t < last_tile ? t : last_tile; };
    if (my_tiles > 0) {
#pragma unroll
        for (int d = 0; d < PD; ++d) issue(S[d], tile_of(d));
    }
    __syncthreads();                                            // affine caches visible
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ba[k] = *(const f32x2*)(aff + 2 * AFFC + cbB_safe + 2 * k);
        bb2[k] = *(const f32x2*)(aff + 2 * AFFC + AFFC / 2 + cbB_safe + 2 * k);
    }
    if (my_tiles > 0) commit(S[0], halo0, bt0);
    __syncthreads();

    auto compute = [&](const T* halo, const T* btile) __attribute__((always_inline)) {
        if constexpr (sizeof(T) == 2) {
            // The (32-pixel k-chunk, (tap, 16-row) pair) steps of a tile as ONE software pipeline: the A fragments of the next
            // WG_DEPTH steps and the B fragments of the next k-chunk are in flight while a step's MFMAs issue.  A pair is one or two
            // MFMAs (16-32 cycles) against ~130 cycles of transposing-LDS-read latency, and the large-volume kernels run one or two
            // waves per SIMD: one step ahead left the loop waiting on LDS (3D 16->16 at 112x112x80: 4 us per 256-pixel brick).
            // lane group g covers pixels 8g..8g+7 of the 32-pixel chunk: row = 2*kc + (g>>1), x = 8*(g&1) + 0..7;
            // this lane supplies pixel +qq, channels 4*pp.. (offsets precomputed once per thread: a_off / b_off)
            constexpr int NSTEP = NKCH * MAXP, DEPTH = NSTEP < WG_DEPTH ? NSTEP : WG_DEPTH;
            auto load_a = [&](int st) __attribute__((always_inline)) -> uint4 {
                const int kc = st / MAXP, i = st % MAXP;
                // rows 2kc, 2kc + 1 of the tile: a slab's rows are consecutive halo rows, a brick has TH rows per plane
                const bf16_t* p = (const bf16_t*)halo + (ZW ? (((2 * kc) / G::TH) * G::STDg * G::HH + ((2 * kc) % G::TH) * ST) * (G::HW * PS) : kc * (2 * ST * G::HW * PS)) + a_off[i];
                const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * ST * PS);
                return __builtin_bit_cast(uint4, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            };
            auto load_b = [&](int kc, uint4 (&bf)[NTB]) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < NTB; ++t) {
                    const bf16_t* bp = (const bf16_t*)btile + b_off + kc * (2 * G::TW * PSB) + t * 16;
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
                const int kc = st / MAXP, i = st % MAXP;
                if (i == 0 && kc + 1 < NKCH) load_b(kc + 1, bf[(kc + 1) & 1]);
                const uint4 a = af[st % DEPTH];
                if (wave + 4 * i < PAIRS) {                            // wave-uniform (only the last pair of a wave can be missing)
#pragma unroll
                    for (int t = 0; t < NTB; ++t) mma8(acc[i][t], a, bf[kc & 1][t]);
                }
                if (st + DEPTH < NSTEP) af[st % DEPTH] = load_a(st + DEPTH);
            }
        } else {
#pragma unroll 1
            for (int kc = 0; kc < NKCH; ++kc) {
                // fp32: 8 MFMAs of 4 pixels each; lane (channel l15, pixel 4*sub + g)
#pragma unroll 2
                for (int sub = 0; sub < 8; ++sub) {
                    const int q = kc * 32 + sub * 4 + g;
                    const int row = q / G::TW, xx = q % G::TW;
                    float bfr[NTB];
#pragma unroll
                    for (int t = 0; t < NTB; ++t) bfr[t] = ((const float*)btile)[q * PSB + t * 16 + l15];
#pragma unroll
                    for (int i = 0; i < MAXP; ++i) {
                        const int pair = wave + 4 * i;
                        if (pair < PAIRS) {
                            const int tap = pair / KCT, kct = pair % KCT;
                            const int dx = tap % KS, dy = (tap / KS) % KS, dz = tap / (KS * KS);
                            const float a = ((const float*)halo)[((dz * G::HH + row * ST + dy) * G::HW + xx * ST + dx) * PS + kct * 16 + l15];
#pragma unroll
                            for (int t = 0; t < NTB; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfr[t], acc[i][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };
    // one tile: set `s` (tile k, already in LDS) is free -> prefetch tile k + PD into it; compute tile k; land tile k + 1
    // (set s + 1, requested PD - 1 tiles ago) in the other LDS buffers.  Returns false after the last tile.
    long k = 0;
    auto body = [&](auto s_const) __attribute__((always_inline)) -> bool {
        constexpr int s = decltype(s_const)::value;
        issue(S[s], tile_of(k + PD));
        compute((k & 1) ? halo1 : halo0, (k & 1) ? bt1 : bt0);
        if (k + 1 >= my_tiles) return false;                    // see conv_fwd_kernel: leave BEFORE the commit
        commit(S[(s + 1) % PD], (k & 1) ? halo0 : halo1, (k & 1) ? bt0 : bt1);
        __syncthreads();
        ++k;
        return true;
    };
    if (my_tiles > 0) {
        for (;;) {
            if (!body(std::integral_constant<int, 0>{})) break;
            if constexpr (PD > 1) { if (!body(std::integral_constant<int, 1 % PD>{})) break; }
            if constexpr (PD > 2) { if (!body(std::integral_constant<int, 2 % PD>{})) break; }
            if constexpr (PD > 3) { if (!body(std::integral_constant<int, 3 % PD>{})) break; }
        }
    }
    // ---- partial slab: ws[split][tap][kc_global][kn_global]; lane holds rows 4g+j (kc), col l15 (kn) ----
    float* slab = ws + (long)split * G::NTAPS * Ca * Cb;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int pair = wave + 4 * i;
        if (pair < PAIRS) {
            const int tap = pair / KCT, kct = pair % KCT;
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                const int kn = nb * BN + t * 16 + l15;
                if (kn < Cb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kcg = chunk * KC + kct * 16 + 4 * g + j;
                        slab[((long)tap * Ca + kcg) * Cb + kn] = acc[i][t][j];
                    }
                }
            }
        }
    }
    if (want_db) {
        // threads with equal (tid % CG) hold the same 8 channels: shuffle over lanes CG, 2CG, .. apart, then across waves via LDS
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = dbsum[j];
            for (int o = CG; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane < CG) dbred[wave * BN + lane * 8 + j] = v;
        }
        __syncthreads();
        if (threadIdx.x < BN && nb * BN + threadIdx.x < Cb)
            ws_db[(long)split * Cb + nb * BN + threadIdx.x] = dbred[threadIdx.x] + dbred[BN + threadIdx.x] + dbred[2 * BN + threadIdx.x] + dbred[3 * BN + threadIdx.x];
    }
}
