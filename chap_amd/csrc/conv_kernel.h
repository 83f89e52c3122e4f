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

#ifndef CHAP_CONV_MINWAVES
#define CHAP_CONV_MINWAVES 1      // __launch_bounds__ 2nd argument (waves per SIMD) -- lab knob
#endif
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
// Everything that does not depend on the tile is computed ONCE per thread (unit -> halo coordinates,
// LDS offset, MFMA fragment offsets, output offsets); per item only a few adds/compares remain and all
// global addresses are (wave-uniform 64-bit base) + (32-bit lane offset).
template <typename T, int UNITS, bool ADD2> struct halo_regs {
    typename frag<T>::type raw[UNITS];
    typename frag<T>::type raw2[ADD2 ? UNITS : 1];
    uint2 keep[UNITS];
    unsigned ok;               // bit j: unit j is inside the input (else zero padding)
};

template <int UNITS> struct unit_desc {
    int hzyx[UNITS];           // hz << 20 | hy << 10 | hx   (-1: unit not used by this thread)
    int rel[UNITS];            // (hz*IH + hy)*IW + hx : input pixel offset from the halo origin
    int lds[UNITS];            // pix*PS + cgl*8 (elements)
    int c8[UNITS];             // cgl*8
};

struct src_scalars {           // wave-uniform per-source values hoisted out of the loops
    const void* ptr; const uint8_t* keep; const float* chan_mul;
    int C, ld, coff; float slope_eff, keep_scale; bool has_keep, has_cm;
};
__device__ __forceinline__ src_scalars make_scalars(const chap_src_t& s) {
    src_scalars r;
    r.ptr = s.ptr; r.keep = s.keep; r.chan_mul = s.chan_mul; r.C = s.C; r.ld = s.ld; r.coff = s.coff;
    r.slope_eff = s.act ? s.slope : 1.f; r.keep_scale = s.keep_scale; r.has_keep = s.keep != nullptr; r.has_cm = s.chan_mul != nullptr;
    return r;
}

template <typename T, int KS, int ST, bool D3, int KC, int MR, bool ADD2, int UNITS>
__device__ __forceinline__ void halo_issue(halo_regs<T, UNITS, ADD2>& R, const unit_desc<UNITS>& U, const src_scalars& s0, const src_scalars& s1,
                                           int ID, int IH, int IW, int n, int z0, int y0, int x0, int chunk) {
    typedef conv_geom<KS, ST, D3, MR> G;
    const int gz0 = z0 * G::STD - (D3 ? G::PAD : 0), gy0 = y0 * ST - G::PAD, gx0 = x0 * ST - G::PAD;
    const long gp0 = (((long)n * ID + gz0) * IH + gy0) * IW + gx0;           // halo origin (may lie outside: only offsets that pass the bounds test are used)
    const T* b0 = (const T*)s0.ptr + gp0 * s0.ld + s0.coff;
    const T* b1 = (const T*)s1.ptr + gp0 * s1.ld + s1.coff;
    const uint8_t* k0 = s0.keep + gp0 * s0.C;
    const uint8_t* k1 = s1.keep + gp0 * s1.C;
    const int cbase = chunk * KC;
    R.ok = 0;
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        const int d = U.hzyx[j];
        if (CHAP_ABLATE & 16) continue;
        const int hz = d >> 20, hy = (d >> 10) & 1023, hx = d & 1023;
        const bool ok = d >= 0 && (unsigned)(gz0 + hz) < (unsigned)ID && (unsigned)(gy0 + hy) < (unsigned)IH && (unsigned)(gx0 + hx) < (unsigned)IW;
        if (ok) {
            R.ok |= 1u << j;
            int c = cbase + U.c8[j];
            if (ADD2) {
                R.raw[j] = frag<T>::load(b0 + U.rel[j] * s0.ld + c);
                R.raw2[j] = frag<T>::load(b1 + U.rel[j] * s1.ld + c);
                if (s0.has_keep) R.keep[j] = *(const uint2*)(k0 + U.rel[j] * s0.C + c);
            } else {
                const bool second = c >= s0.C;
                c = second ? c - s0.C : c;
                const T* b = second ? b1 : b0;
                const int ld = second ? s1.ld : s0.ld;
                R.raw[j] = frag<T>::load(b + U.rel[j] * ld + c);
                const bool hk = second ? s1.has_keep : s0.has_keep;
                if (hk) {
                    const uint8_t* kb = second ? k1 : k0;
                    const int kc = second ? s1.C : s0.C;
                    R.keep[j] = *(const uint2*)(kb + U.rel[j] * kc + c);
                }
            }
        }
    }
}

// affine cache layout in LDS: [src][scale | shift][CONV_MAX_AFFINE_C/2] (identity when a source has none).
// Packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32, 2 elements per VALU instruction): the commit phase is
// VALU-issue-bound.  leaky(z) = max(z, slope*z) for 0 <= slope <= 1 (slope_eff = 1: no activation).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lazy_transform(float v[8], const float* aff, int c, float slope_eff, bool has_keep, uint2 keep, float keep_scale,
                                               const float* cm) {
    const f32x2 sl = {slope_eff, slope_eff};
    const f32x2* a2 = (const f32x2*)(aff + c);
    const f32x2* b2 = (const f32x2*)(aff + CONV_MAX_AFFINE_C / 2 + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 z = {v[2 * j], v[2 * j + 1]};
        z = z * a2[j] + b2[j];
        const f32x2 m = z * sl;
        v[2 * j] = fmaxf(z.x, m.x); v[2 * j + 1] = fmaxf(z.y, m.y);
    }
    if (has_keep) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t w = j < 4 ? keep.x : keep.y;
            v[j] = ((w >> (8 * (j & 3))) & 0xff) ? v[j] * keep_scale : 0.f;
        }
    }
    if (cm) {
        float a[8];
        ld8(cm + c, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= a[j];
    }
}

template <typename T, int KS, int ST, bool D3, int KC, int MR, bool ADD2, int UNITS>
__device__ __forceinline__ void halo_commit(const halo_regs<T, UNITS, ADD2>& R, const unit_desc<UNITS>& U, T* halo, const src_scalars& s0, const src_scalars& s1,
                                            const float* aff, bool plain, int n, int chunk) {
    const int cbase = chunk * KC;
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        if (U.hzyx[j] < 0) continue;
        T* dst = halo + U.lds[j];
        if (!((R.ok >> j) & 1u)) { frag<T>::store(dst, frag<T>::zero()); continue; }
        if (plain) { frag<T>::store(dst, R.raw[j]); continue; }
        float v[8];
        frag<T>::unpack(R.raw[j], v);
        int c = cbase + U.c8[j];
        if (ADD2) {
            lazy_transform(v, aff, c, s0.slope_eff, s0.has_keep, R.keep[j], s0.keep_scale, s0.has_cm ? s0.chan_mul + (long)n * s0.C : nullptr);
            float w[8];
            frag<T>::unpack(R.raw2[j], w);
            lazy_transform(w, aff + CONV_MAX_AFFINE_C, c, s1.slope_eff, false, make_uint2(0, 0), 1.f, s1.has_cm ? s1.chan_mul + (long)n * s1.C : nullptr);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += w[k];
        } else {
            const bool second = c >= s0.C;
            c = second ? c - s0.C : c;
            const float* af = second ? aff + CONV_MAX_AFFINE_C : aff;
            const float se = second ? s1.slope_eff : s0.slope_eff;
            const bool hk = second ? s1.has_keep : s0.has_keep;
            const float ks = second ? s1.keep_scale : s0.keep_scale;
            const float* cm = second ? (s1.has_cm ? s1.chan_mul + (long)n * s1.C : nullptr) : (s0.has_cm ? s0.chan_mul + (long)n * s0.C : nullptr);
            lazy_transform(v, af, c, se, hk, R.keep[j], ks, cm);
        }
        st8(dst, v);
    }
}

// tile index -> (n, z0, y0, x0)
template <int TH, int TW>
__device__ __forceinline__ void tile_coords(long tile, int tiles_x, int tiles_y, int D, int& n, int& z0, int& y0, int& x0) {
    const unsigned t = (unsigned)tile;
    const unsigned tx = t % (unsigned)tiles_x; unsigned r = t / (unsigned)tiles_x;
    const unsigned ty = r % (unsigned)tiles_y; r /= (unsigned)tiles_y;
    z0 = (int)(r % (unsigned)D); n = (int)(r / (unsigned)D);
    x0 = (int)tx * TW; y0 = (int)ty * TH;
}

template <typename T, int KS, int ST, bool D3, int KC, int NT, int MR, bool ADD2, bool WLDS>
__global__ __launch_bounds__(256, (CHAP_CONV_MINWAVES > 1 ? CHAP_CONV_MINWAVES : (KC == 16 && !D3 ? 4 : 1))) void conv_fwd_kernel(const chap_conv_params P) {
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
    const src_scalars s0 = make_scalars(P.src[0]);
    const src_scalars s1 = make_scalars(P.nsrc > 1 ? P.src[1] : P.src[0]);

    // XCD-aware tile walk: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous range
    // of tiles so that the halos re-read by neighbouring tiles are L2 hits.
    const int nb = gridDim.x;
    const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
    const int bpx = (nb + 7 - xcd) >> 3;                        // blocks on this XCD group
    const long per = (ntiles + 7) / 8;
    const long t_begin = per * xcd, t_end = (t_begin + per < ntiles) ? t_begin + per : ntiles;
    const long my_tiles = (t_begin + bix < t_end) ? (t_end - t_begin - bix + bpx - 1) / bpx : 0;
    const long nitems = my_tiles * nchunks;

    // ---- one-time per thread: unit descriptors, MFMA fragment offsets ----
    unit_desc<UNITS> U;
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        const int u = threadIdx.x + 256 * j;
        U.hzyx[j] = -1; U.rel[j] = 0; U.lds[j] = 0; U.c8[j] = 0;
        if (u < G::HP * GPT) {
            const int pix = u / GPT, cgl = u % GPT;
            const int hx = pix % G::HW, hy = (pix / G::HW) % G::HH, hz = pix / (G::HW * G::HH);
            U.hzyx[j] = (hz << 20) | (hy << 10) | hx;
            U.rel[j] = (hz * P.IH + hy) * P.IW + hx;
            U.lds[j] = pix * PS + cgl * 8;
            U.c8[j] = cgl * 8;
        }
    }
    int xoff[STEPS];                                            // LDS element offset of this lane's B fragment (row m = 0), -1 = zero fragment
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int p = step * 4 + g;
        const int tap = p / GPT, cgl = p % GPT;
        const int dx = tap % KS, dy = (tap / KS) % KS, dz = tap / (KS * KS);
        xoff[step] = p < NP ? ((dz * G::HH + (wave * MR) * ST + dy) * G::HW + px * ST + dx) * PS + cgl * 8 : -1;
    }

    // ---- one-time per block: affine cache, resident weights ----
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
        // resident copy: [chunk][step][t < NT][64 lanes][8]; 8 loads in flight per thread (a load->store->load
        // chain would serialise ~1 us of L2 latency per 16 bytes)
        const long tot = (long)nchunks * STEPS * NT * 64;
        for (long i0 = threadIdx.x; i0 < tot; i0 += 256 * 8) {
            F f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long i = i0 + 256 * k;
                f[k] = frag<T>::zero();
                if (i < tot) {
                    const int ln = (int)(i & 63); long r = i >> 6;
                    const int t = (int)(r % NT); r /= NT;            // r = chunk*STEPS + step
                    if (nt0 + t < ntiles_total) f[k] = frag<T>::load((const T*)P.wpacked + r * wstep + ((long)(nt0 + t) * 64 + ln) * 8);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long i = i0 + 256 * k;
                if (i < tot) frag<T>::store(wlds + i * 8, f[k]);
            }
        }
    }

    f32x4 acc[MR][NT];
    float ssum[NT][4], ssq[NT][4], bj[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nl = (nt0 + t) * 16 + 4 * g;
        const int cb = P.out_mode == 1 ? (nl % P.out_Cn) : nl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ssum[t][j] = 0.f; ssq[t][j] = 0.f;
            bj[t][j] = (P.bias && nl + j < P.Cout) ? P.bias[cb + j] : 0.f;
        }
    }
    const bool do_stats = P.stats != nullptr;
    const int SD2 = (P.dims == 3) ? 2 : 1;
    // per-lane output offsets relative to the tile origin (elements), one per (t): row term added per m
    int ooff[NT], orow;                                         // orow = element stride of one tile row
    {
        const int OW = P.out_mode == 1 ? 2 * P.W : P.W, OH = P.out_mode == 1 ? 2 * P.H : P.H;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nl = (nt0 + t) * 16 + 4 * g;
            if (P.out_planar) {
                ooff[t] = px;                                   // + channel plane handled at store time
            } else if (P.out_mode == 1) {
                const int sub = nl / P.out_Cn, oc = nl % P.out_Cn;
                const int sx = sub & 1, sy = (sub >> 1) & 1, sz = (sub >> 2) & 1;
                ooff[t] = (((sz * OH) + sy) * OW + 2 * px + sx) * P.out_ld + P.out_coff + oc;
            } else {
                ooff[t] = px * P.out_ld + P.out_coff + nl;
            }
        }
        orow = P.out_planar ? P.W : (P.out_mode == 1 ? 2 * OW * P.out_ld : P.W * P.out_ld);
    }

    halo_regs<T, UNITS, ADD2> R;
    int n = 0, z0 = 0, y0 = 0, x0 = 0;
    int nn = 0, nz0 = 0, ny0 = 0, nx0 = 0;                      // coordinates of the item being prefetched
    if (nitems > 0) {
        tile_coords<G::TH, G::TW>(t_begin + bix, tiles_x, tiles_y, P.D, nn, nz0, ny0, nx0);
        halo_issue<T, KS, ST, D3, KC, MR, ADD2, UNITS>(R, U, s0, s1, P.ID, P.IH, P.IW, nn, nz0, ny0, nx0, 0);
    }
    __syncthreads();                                            // affine cache (+ resident weights) visible
    if (nitems > 0) halo_commit<T, KS, ST, D3, KC, MR, ADD2, UNITS>(R, U, halo0, s0, s1, aff, plain, nn, 0);
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
            halo_issue<T, KS, ST, D3, KC, MR, ADD2, UNITS>(R, U, s0, s1, P.ID, P.IH, P.IW, nn, nz0, ny0, nx0, nchunk);
        }
        // ---- MFMA over the taps of this chunk
        const T* wc_g = (const T*)P.wpacked + (long)chunk * STEPS * wstep + ((long)nt0 * 64 + lane) * 8;
        const T* wc_l = wlds + ((long)chunk * STEPS * NT * 64 + lane) * 8;
#pragma unroll
        for (int step = 0; step < ((CHAP_ABLATE & 2) ? 0 : STEPS); ++step) {
            F wf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (WLDS) wf[t] = frag<T>::load(wc_l + ((long)step * NT + t) * 512);
                else wf[t] = (nt0 + t < ntiles_total) ? frag<T>::load(wc_g + (long)step * wstep + t * 512) : frag<T>::zero();
            }
            const int xo = xoff[step];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                F xf = xo >= 0 ? frag<T>::load(cur + xo + m * (ST * G::HW * PS)) : frag<T>::zero();
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (CHAP_ABLATE & 1) { float kx[8], kw[8]; frag<T>::unpack(xf, kx); frag<T>::unpack(wf[t], kw); asm volatile("" :: "v"(kx[0]), "v"(kx[7]), "v"(kw[0]), "v"(kw[7])); } else mma8(acc[m][t], wf[t], xf);
                }
            }
        }
        // ---- epilogue after the last K-chunk of a tile: lane holds D[cout = 16*t + 4*g + j][pixel (row m, x = px)]
        if (chunk == nchunks - 1) {
            const bool xok = x0 + px < P.W;
            // wave-uniform 64-bit origin of this tile in the output, 32-bit lane offsets on top
            long o0;
            if (P.out_planar) o0 = ((long)n * P.Cout * P.D + z0) * P.H * P.W + (long)y0 * P.W + x0;     // + c*plane
            else if (P.out_mode == 1) o0 = ((((long)n * (P.D * SD2) + z0 * SD2) * (2 * P.H) + 2 * y0) * (2 * P.W) + 2 * x0) * P.out_ld;
            else o0 = ((((long)n * P.D + z0) * P.H + y0) * P.W + x0) * P.out_ld;
            const long plane = (long)P.D * P.H * P.W;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int nl = (nt0 + t) * 16 + 4 * g;            // logical output channel of j = 0
                if (nl >= P.Cout) continue;
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const int row = wave * MR + m;
                    const bool valid = xok && (y0 + row < P.H);
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[m][t][j] + bj[t][j];
                    if (!valid) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ssum[t][j] += v[j]; ssq[t][j] += v[j] * v[j]; }
                    if (CHAP_ABLATE & 8) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }
                    if (P.out_planar) {
                        float* o = (float*)P.out + o0 + row * orow + ooff[t];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (nl + j < P.Cout) o[(nl + j) * plane] = v[j];
                    } else {
                        const long oi = o0 + row * orow + ooff[t];
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
        if (has_next) halo_commit<T, KS, ST, D3, KC, MR, ADD2, UNITS>(R, U, nxt, s0, s1, aff, plain, nn, nchunk);
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
