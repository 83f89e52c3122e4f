// Implicit-GEMM convolution for gfx950 (CDNA4, wave64): LDS-staged halo tile ("im2col in LDS"),
// MFMA 16x16x32 bf16 / 16x16x4 f32, fp32 accumulate, fused bias + BatchNorm-statistics epilogue.
//
// GEMM view:  D[cout][pixel] = sum_k  W[cout][k] * X[k][pixel],   k = (tap, channel)
//   "A" operand = packed weights (row = cout), fragment order, resident in LDS when they fit,
//   "B" operand = pixels (col = pixel), read from the LDS halo tile.
// With that role assignment a lane ends up holding 4 consecutive output channels of ONE pixel, so the
// channel-last store is a vector store and the per-channel statistics are a 16-lane shuffle reduce.
//
// PERSISTENT blocks (256 threads = 4 waves): a block walks a list of work items (tile, K-chunk).
// The halo of item i+1 is fetched into registers (global loads in flight) while item i is computed
// from LDS, then transformed (lazy activation: BN affine + LeakyReLU + dropout) and written to the
// other LDS buffer -- one barrier per item.  Output tile = TH x 16 pixels of one z-plane (TH = 4*MR,
// wave w owns rows [w*MR, w*MR+MR)) x 16*NT output channels.  BatchNorm statistics are kept in
// registers across tiles and flushed with ONE float atomic per channel per block.
#pragma once
#include "common.h"

#ifndef CHAP_ABLATE
#define CHAP_ABLATE 0          // tools/lab/conv_lab.hip builds ablated variants; the library never does
#endif

template <int KC> struct lds_pix_stride {};        // LDS pixel stride in elements, conflict-free for
template <> struct lds_pix_stride<32> { static constexpr int bf16 = 48, f32 = 36; };  // ds_read_b128
template <> struct lds_pix_stride<16> { static constexpr int bf16 = 16, f32 = 20; };

template <typename T, int KC> __host__ __device__ constexpr int pix_stride() {
    return sizeof(T) == 2 ? lds_pix_stride<KC>::bf16 : lds_pix_stride<KC>::f32;
}

template <int KS, int ST, bool D3, int MR> struct conv_geom {
    static constexpr int TW = 16, TH = 4 * MR;
    static constexpr int KD = D3 ? KS : 1;
    static constexpr int HW = (TW - 1) * ST + KS, HH = (TH - 1) * ST + KS, HD = KD;
    static constexpr int HP = HD * HH * HW;
    static constexpr int NTAPS = KD * KS * KS;
    static constexpr int PAD = (KS - ST) / 2;
    static constexpr int STD = D3 ? ST : 1;
};

constexpr int CONV_MAX_AFFINE_C = 1024;            // channels (both sources) whose scale/shift are cached in LDS

template <typename T, int KS, int ST, bool D3, int KC, int MR>
__host__ __device__ constexpr size_t conv_lds_fixed_bytes(int NT) {
    return 2 * (size_t)conv_geom<KS, ST, D3, MR>::HP * pix_stride<T, KC>() * sizeof(T)   // two halo buffers
           + 2 * 16 * NT * sizeof(float);                                                // block statistics
}

// ---- MFMA wrappers: acc += W(8 k-values of one cout) x X(8 k-values of one pixel) ---------------
__device__ __forceinline__ void mma8(f32x4& acc, const uint4& w, const uint4& x) {       // bf16
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, w), __builtin_bit_cast(s16x8, x), acc, 0, 0, 0);
}
struct f32x8 { float4 lo, hi; };
__device__ __forceinline__ void mma8(f32x4& acc, const f32x8& w, const f32x8& x) {       // exact f32
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.x, x.lo.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.y, x.lo.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.z, x.lo.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.w, x.lo.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.x, x.hi.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.y, x.hi.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.z, x.hi.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.w, x.hi.w, acc, 0, 0, 0);
}
template <typename T> struct frag;
template <> struct frag<bf16_t> {
    typedef uint4 type;
    static __device__ __forceinline__ uint4 load(const bf16_t* p) { return *(const uint4*)p; }
    static __device__ __forceinline__ uint4 zero() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ void unpack(const uint4& a, float v[8]) {
        v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
        v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
        v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
        v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const uint4& a) { *(uint4*)p = a; }
};
template <> struct frag<float> {
    typedef f32x8 type;
    static __device__ __forceinline__ f32x8 load(const float* p) { f32x8 r; r.lo = *(const float4*)p; r.hi = *(const float4*)(p + 4); return r; }
    static __device__ __forceinline__ f32x8 zero() { f32x8 r; r.lo = make_float4(0, 0, 0, 0); r.hi = r.lo; return r; }
    static __device__ __forceinline__ void unpack(const f32x8& a, float v[8]) {
        v[0] = a.lo.x; v[1] = a.lo.y; v[2] = a.lo.z; v[3] = a.lo.w; v[4] = a.hi.x; v[5] = a.hi.y; v[6] = a.hi.z; v[7] = a.hi.w;
    }
    static __device__ __forceinline__ void store(float* p, const f32x8& a) { *(float4*)p = a.lo; *(float4*)(p + 4) = a.hi; }
};

// Synchronous halo staging (used by the weight-gradient kernel): global -> transform -> LDS.
template <typename T, int KS, int ST, bool D3, int KC, int MR>
__device__ __forceinline__ void stage_halo(T* halo, const chap_src_t& s0, const chap_src_t& s1, int nsrc, int combine,
                                           int n, int z0, int y0, int x0, int ID, int IH, int IW, int chunk) {
    typedef conv_geom<KS, ST, D3, MR> G;
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>();
    for (int u = threadIdx.x; u < G::HP * GPT; u += 256) {
        const int pix = u / GPT, cgl = u % GPT;
        const int hx = pix % G::HW, hy = (pix / G::HW) % G::HH, hz = pix / (G::HW * G::HH);
        const int gz = z0 * G::STD + hz - (D3 ? G::PAD : 0), gy = y0 * ST + hy - G::PAD, gx = x0 * ST + hx - G::PAD;
        float v[8];
        if ((unsigned)gz < (unsigned)ID && (unsigned)gy < (unsigned)IH && (unsigned)gx < (unsigned)IW) {
            const long gp = (((long)n * ID + gz) * IH + gy) * IW + gx;
            const int c = chunk * KC + cgl * 8;
            if (combine == 0) {
                if (c < s0.C) src_load8<T>(s0, n, gp, c, v); else src_load8<T>(s1, n, gp, c - s0.C, v);
            } else {
                src_load8<T>(s0, n, gp, c, v);
                if (nsrc > 1) {
                    float w[8];
                    src_load8<T>(s1, n, gp, c, w);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += w[j];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        st8(halo + pix * PS + cgl * 8, v);
    }
}

// ---- register-staged prefetch of one halo (item = tile x K-chunk) ---------------------------------
template <typename T, int UNITS, bool ADD2> struct halo_regs {
    typename frag<T>::type raw[UNITS];
    typename frag<T>::type raw2[ADD2 ? UNITS : 1];
    uint2 keep[UNITS];
    long gpix[UNITS];          // global pixel index, -1 = out of bounds (zero padding)
};

template <typename T, int KS, int ST, bool D3, int KC, int MR, bool ADD2, int UNITS>
__device__ __forceinline__ void halo_issue(halo_regs<T, UNITS, ADD2>& R, const chap_conv_params& P, int n, int z0, int y0, int x0, int chunk) {
    typedef conv_geom<KS, ST, D3, MR> G;
    constexpr int GPT = KC / 8;
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        const int u = threadIdx.x + 256 * j;
        R.gpix[j] = -1;
        if (u < G::HP * GPT) {
            const int pix = u / GPT, cgl = u % GPT;
            const int hx = pix % G::HW, hy = (pix / G::HW) % G::HH, hz = pix / (G::HW * G::HH);
            const int gz = z0 * G::STD + hz - (D3 ? G::PAD : 0), gy = y0 * ST + hy - G::PAD, gx = x0 * ST + hx - G::PAD;
            if (!(CHAP_ABLATE & 16) && (unsigned)gz < (unsigned)P.ID && (unsigned)gy < (unsigned)P.IH && (unsigned)gx < (unsigned)P.IW) {
                const long gp = (((long)n * P.ID + gz) * P.IH + gy) * P.IW + gx;
                R.gpix[j] = gp;
                int c = chunk * KC + cgl * 8;
                if (ADD2) {
                    R.raw[j] = frag<T>::load((const T*)P.src[0].ptr + gp * P.src[0].ld + P.src[0].coff + c);
                    R.raw2[j] = frag<T>::load((const T*)P.src[1].ptr + gp * P.src[1].ld + P.src[1].coff + c);
                    if (P.src[0].keep) R.keep[j] = *(const uint2*)(P.src[0].keep + gp * P.src[0].C + c);
                } else {
                    const bool second = c >= P.src[0].C;
                    const chap_src_t& s = second ? P.src[1] : P.src[0];
                    if (second) c -= P.src[0].C;
                    R.raw[j] = frag<T>::load((const T*)s.ptr + gp * s.ld + s.coff + c);
                    if (s.keep) R.keep[j] = *(const uint2*)(s.keep + gp * s.C + c);
                }
            }
        }
    }
}

// affine cache layout in LDS: [src][2][Cmax]: scale then shift.
template <typename T>
__device__ __forceinline__ void lazy_transform(float v[8], const chap_src_t& s, const float* aff, int c, uint2 keep, int n) {
    if (s.scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], aff[c + j], aff[CONV_MAX_AFFINE_C / 2 + c + j]);
    }
    if (s.act) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * s.slope;
    }
    if (s.keep) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t w = j < 4 ? keep.x : keep.y;
            v[j] = ((w >> (8 * (j & 3))) & 0xff) ? v[j] * s.keep_scale : 0.f;
        }
    }
    if (s.chan_mul) {
        float a[8];
        ld8(s.chan_mul + (long)n * s.C + c, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= a[j];
    }
}

template <typename T, int KS, int ST, bool D3, int KC, int MR, bool ADD2, int UNITS>
__device__ __forceinline__ void halo_commit(const halo_regs<T, UNITS, ADD2>& R, T* halo, const chap_conv_params& P, const float* aff,
                                            bool plain, int n, int chunk) {
    typedef conv_geom<KS, ST, D3, MR> G;
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>();
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        const int u = threadIdx.x + 256 * j;
        if (u < G::HP * GPT) {
            const int pix = u / GPT, cgl = u % GPT;
            T* dst = halo + pix * PS + cgl * 8;
            if (R.gpix[j] < 0) { frag<T>::store(dst, frag<T>::zero()); continue; }
            if (plain) { frag<T>::store(dst, R.raw[j]); continue; }
            float v[8];
            frag<T>::unpack(R.raw[j], v);
            int c = chunk * KC + cgl * 8;
            if (ADD2) {
                lazy_transform<T>(v, P.src[0], aff, c, R.keep[j], n);
                float w[8];
                frag<T>::unpack(R.raw2[j], w);
                lazy_transform<T>(w, P.src[1], aff + CONV_MAX_AFFINE_C, c, make_uint2(0, 0), n);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += w[k];
            } else {
                const bool second = c >= P.src[0].C;
                if (second) lazy_transform<T>(v, P.src[1], aff + CONV_MAX_AFFINE_C, c - P.src[0].C, R.keep[j], n);
                else lazy_transform<T>(v, P.src[0], aff, c, R.keep[j], n);
            }
            st8(dst, v);
        }
    }
}

// tile index -> (n, z0, y0, x0)
template <int TH, int TW>
__device__ __forceinline__ void tile_coords(long tile, int tiles_x, int tiles_y, int D, int& n, int& z0, int& y0, int& x0) {
    const int tx = (int)(tile % tiles_x); tile /= tiles_x;
    const int ty = (int)(tile % tiles_y); tile /= tiles_y;
    z0 = (int)(tile % D); n = (int)(tile / D);
    x0 = tx * TW; y0 = ty * TH;
}

template <typename T, int KS, int ST, bool D3, int KC, int NT, int MR, bool ADD2, bool WLDS>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const chap_conv_params P) {
    typedef conv_geom<KS, ST, D3, MR> G;
    typedef typename frag<T>::type F;
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>();
    constexpr int NP = G::NTAPS * GPT, STEPS = (NP + 3) / 4;
    constexpr int UNITS = (G::HP * GPT + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo0 = (T*)smem;
    T* halo1 = halo0 + (size_t)G::HP * PS;
    float* bstat = (float*)(halo1 + (size_t)G::HP * PS);
    float* aff = bstat + 2 * 16 * NT;                       // [2 sources][scale | shift][CONV_MAX_AFFINE_C/2]
    T* wlds = (T*)(aff + 2 * CONV_MAX_AFFINE_C);            // WLDS only

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const int tiles_x = (P.W + G::TW - 1) / G::TW, tiles_y = (P.H + G::TH - 1) / G::TH;
    const long ntiles = (long)P.N * P.D * tiles_y * tiles_x;
    const int ntiles_total = (P.Cout + 15) >> 4;
    const int nt0 = blockIdx.y * NT;
    const int Ctot = P.combine == 0 ? (P.src[0].C + (P.nsrc > 1 ? P.src[1].C : 0)) : P.src[0].C;
    const int nchunks = Ctot / KC;
    const bool plain = (CHAP_ABLATE & 4) ||
                       (!ADD2 && P.src[0].scale == nullptr && !P.src[0].act && P.src[0].keep == nullptr && P.src[0].chan_mul == nullptr &&
                        (P.nsrc < 2 || (P.src[1].scale == nullptr && !P.src[1].act && P.src[1].keep == nullptr && P.src[1].chan_mul == nullptr)));

    // XCD-aware tile walk: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous range
    // of tiles so that the halos re-read by neighbouring tiles are L2 hits.
    const int nb = gridDim.x;
    const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
    const int bpx = (nb + 7 - xcd) >> 3;                        // blocks on this XCD group
    const long per = (ntiles + 7) / 8;
    const long t_begin = per * xcd, t_end = (t_begin + per < ntiles) ? t_begin + per : ntiles;
    const long my_tiles = (t_begin + bix < t_end) ? (t_end - t_begin - bix + bpx - 1) / bpx : 0;
    const long nitems = my_tiles * nchunks;

    // ---- one-time per block: affine cache (only when a source carries one), resident weights ----
    if (!plain) {
        for (int s = 0; s < P.nsrc; ++s) {
            const bool has = P.src[s].scale != nullptr;
            for (int c = threadIdx.x; c < P.src[s].C; c += 256) {
                aff[s * CONV_MAX_AFFINE_C + c] = has ? P.src[s].scale[c] : 1.f;
                aff[s * CONV_MAX_AFFINE_C + CONV_MAX_AFFINE_C / 2 + c] = has ? P.src[s].shift[c] : 0.f;
            }
        }
    }
    const long wstep = (long)ntiles_total * 64 * 8;             // packed elements per (chunk, step)
    if (WLDS) {
        // resident copy: [chunk][step][t < NT][64 lanes][8]
        const long tot = (long)nchunks * STEPS * NT * 64;
        for (long i = threadIdx.x; i < tot; i += 256) {
            const int ln = (int)(i & 63); long r = i >> 6;
            const int t = (int)(r % NT); r /= NT;                // r = chunk*STEPS + step
            F f = (nt0 + t < ntiles_total) ? frag<T>::load((const T*)P.wpacked + r * wstep + ((long)(nt0 + t) * 64 + ln) * 8) : frag<T>::zero();
            frag<T>::store(wlds + i * 8, f);
        }
    }

    f32x4 acc[MR][NT];
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[t][j] = 0.f; ssq[t][j] = 0.f; }
    const bool do_stats = P.stats != nullptr;
    const int SD2 = (P.dims == 3) ? 2 : 1;

    halo_regs<T, UNITS, ADD2> R;
    int n = 0, z0 = 0, y0 = 0, x0 = 0;
    int nn = 0, nz0 = 0, ny0 = 0, nx0 = 0;                      // coordinates of the item being prefetched
    if (nitems > 0) {
        tile_coords<G::TH, G::TW>(t_begin + bix, tiles_x, tiles_y, P.D, nn, nz0, ny0, nx0);
        halo_issue<T, KS, ST, D3, KC, MR, ADD2, UNITS>(R, P, nn, nz0, ny0, nx0, 0);
    }
    __syncthreads();                                            // affine cache (+ resident weights) visible
    if (nitems > 0) halo_commit<T, KS, ST, D3, KC, MR, ADD2, UNITS>(R, halo0, P, aff, plain, nn, 0);
    __syncthreads();

    long tile_k = 0;                                            // index into this block's tile list
    int chunk = 0;
    for (long it = 0; it < nitems; ++it) {
        T* cur = (it & 1) ? halo1 : halo0;
        T* nxt = (it & 1) ? halo0 : halo1;
        if (chunk == 0) {
            n = nn; z0 = nz0; y0 = ny0; x0 = nx0;
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // ---- prefetch the next item's halo (global loads stay in flight during the MFMA loop)
        const bool has_next = it + 1 < nitems;
        int nchunk = chunk + 1;
        if (has_next) {
            if (nchunk == nchunks) {
                nchunk = 0;
                tile_coords<G::TH, G::TW>(t_begin + bix + (tile_k + 1) * bpx, tiles_x, tiles_y, P.D, nn, nz0, ny0, nx0);
            }
            halo_issue<T, KS, ST, D3, KC, MR, ADD2, UNITS>(R, P, nn, nz0, ny0, nx0, nchunk);
        }
        // ---- MFMA over the taps of this chunk
        const T* wc_g = (const T*)P.wpacked + (long)chunk * STEPS * wstep + ((long)nt0 * 64 + lane) * 8;
        const T* wc_l = wlds + ((long)chunk * STEPS * NT * 64 + lane) * 8;
#pragma unroll 3
        for (int step = 0; step < ((CHAP_ABLATE & 2) ? 0 : STEPS); ++step) {
            F wf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (WLDS) wf[t] = frag<T>::load(wc_l + ((long)step * NT + t) * 512);
                else wf[t] = (nt0 + t < ntiles_total) ? frag<T>::load(wc_g + (long)step * wstep + t * 512) : frag<T>::zero();
            }
            const int p = step * 4 + g;
            const int tap = p / GPT, cgl = p % GPT;
            const int dx = tap % KS, dy = (tap / KS) % KS, dz = tap / (KS * KS);
            const bool pv = p < NP;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const int row = wave * MR + m;
                const int off = ((dz * G::HH + row * ST + dy) * G::HW + px * ST + dx) * PS + cgl * 8;
                F xf = pv ? frag<T>::load(cur + off) : frag<T>::zero();
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (CHAP_ABLATE & 1) { float kx[8], kw[8]; frag<T>::unpack(xf, kx); frag<T>::unpack(wf[t], kw); asm volatile("" :: "v"(kx[0]), "v"(kx[7]), "v"(kw[0]), "v"(kw[7])); } else mma8(acc[m][t], wf[t], xf);
                }
            }
        }
        // ---- epilogue after the last K-chunk of a tile: lane holds D[cout = 16*t + 4*g + j][pixel (row m, x = px)]
        if (chunk == nchunks - 1) {
            const int gx = x0 + px;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int nl = (nt0 + t) * 16 + 4 * g;            // logical output channel of j = 0
                if (nl >= P.Cout) continue;
                float bj[4] = {0.f, 0.f, 0.f, 0.f};
                const int cb = P.out_mode == 1 ? (nl % P.out_Cn) : nl;
                if (P.bias) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (nl + j < P.Cout) bj[j] = P.bias[cb + j];
                }
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const int gy = y0 + wave * MR + m;
                    const bool valid = gy < P.H && gx < P.W;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[m][t][j] + bj[j];
                        if (valid) { ssum[t][j] += v[j]; ssq[t][j] += v[j] * v[j]; }
                    }
                    if (!valid) continue;
                    if (CHAP_ABLATE & 8) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }
                    if (P.out_planar) {
                        float* o = (float*)P.out;
                        const long plane = (long)P.D * P.H * P.W;
                        const long pp = ((long)z0 * P.H + gy) * P.W + gx;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (nl + j < P.Cout) o[((long)n * P.Cout + nl + j) * plane + pp] = v[j];
                    } else {
                        long opix; int oc;
                        if (P.out_mode == 1) {
                            const int sub = nl / P.out_Cn; oc = nl % P.out_Cn;
                            const int sx = sub & 1, sy = (sub >> 1) & 1, sz = (sub >> 2) & 1;
                            opix = (((long)n * (P.D * SD2) + z0 * SD2 + sz) * (2 * P.H) + 2 * gy + sy) * (2 * P.W) + 2 * gx + sx;
                        } else {
                            oc = nl;
                            opix = (((long)n * P.D + z0) * P.H + gy) * P.W + gx;
                        }
                        const long oi = opix * P.out_ld + P.out_coff + oc;
                        if (nl + 3 < P.Cout) {
                            if (P.out_f32) st4((float*)P.out + oi, v); else st4((T*)P.out + oi, v);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) if (nl + j < P.Cout) {
                                if (P.out_f32) ((float*)P.out)[oi + j] = v[j]; else ((T*)P.out)[oi + j] = elem<T>::put(v[j]);
                            }
                        }
                    }
                }
            }
            ++tile_k;
        }
        // ---- land the prefetched halo in the other buffer; one barrier per item
        if (has_next) halo_commit<T, KS, ST, D3, KC, MR, ADD2, UNITS>(R, nxt, P, aff, plain, nn, nchunk);
        chunk = nchunk == nchunks ? 0 : nchunk;
        __syncthreads();
    }

    // ---- BatchNorm statistics: registers -> 16-lane shuffle -> LDS -> one atomic per channel per block
    if (do_stats) {
        for (int i = threadIdx.x; i < 2 * 16 * NT; i += 256) bstat[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = ssum[t][j], q = ssq[t][j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
                if (px == 0) {
                    atomicAdd(&bstat[t * 16 + 4 * g + j], s);
                    atomicAdd(&bstat[16 * NT + t * 16 + 4 * g + j], q);
                }
            }
        }
        __syncthreads();
        const int rep = P.stats_reps > 1 ? (blockIdx.x % P.stats_reps) : 0;
        const int Cs = P.out_mode == 1 ? P.out_Cn : P.Cout;       // statistics are per REAL channel
        float* st = P.stats + (long)rep * 2 * Cs;
        for (int i = threadIdx.x; i < 2 * 16 * NT; i += 256) {
            const int which = i / (16 * NT), k = i % (16 * NT);
            const int nl = nt0 * 16 + k;
            if (nl < P.Cout) atomicAdd(&st[which * Cs + (P.out_mode == 1 ? nl % P.out_Cn : nl)], bstat[i]);
        }
    }
}
