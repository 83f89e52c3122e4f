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
// registers across tiles and flushed once per block into that block's partial slot (no atomics).
#pragma once
#include "common.h"
#include "launch.h"
#include <type_traits>

#ifndef CHAP_CONV_MINWAVES
#define CHAP_CONV_MINWAVES 1      // __launch_bounds__ 2nd argument (waves per SIMD) -- lab knob
#endif
#ifndef CHAP_ZW_INTERLEAVE
#define CHAP_ZW_INTERLEAVE 1   // round-robin tile walk of the 3D brick kernels (see conv_fwd_kernel); 0 = contiguous runs (rounds 1-2)
#endif
#ifndef CHAP_ABLATE
#define CHAP_ABLATE 0          // tools/lab/conv_lab.hip builds ablated variants; the library never does
#endif
#ifdef CHAP_CONV_TRACE         // lab only: per-phase s_memtime stamps of one wave per traced block
__device__ unsigned long long chap_trace[4][64][8];
#define CHAP_STAMP(k) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x % 257) == 0 && blockIdx.x / 257 < 4 && it < 8) chap_trace[blockIdx.x / 257][it * 4 + (threadIdx.x >> 6)][k] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long chap_trace_p[4][4][8];
#define CHAP_STAMP_P(k) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x % 257) == 0 && blockIdx.x / 257 < 4) chap_trace_p[blockIdx.x / 257][threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CHAP_STAMP(k) do {} while (0)
#define CHAP_STAMP_P(k) do {} while (0)
#endif

template <int KC> struct lds_pix_stride {};        // LDS pixel stride in elements, conflict-free for
template <> struct lds_pix_stride<32> { static constexpr int bf16 = 48, f32 = 36; };  // ds_read_b128
template <> struct lds_pix_stride<16> { static constexpr int bf16 = 16, f32 = 20; };

template <typename T, int KC> __host__ __device__ constexpr int pix_stride() {
    return sizeof(T) == 2 ? lds_pix_stride<KC>::bf16 : lds_pix_stride<KC>::f32;
}

// ZW ("z per wave", 3D only): the 4 waves own 4 consecutive z-planes (MR rows of 16 pixels each) instead of
// 4*MR rows of one plane: a 4 x MR x 16 output brick has a much smaller halo per output than a slab.
template <int KS, int ST, bool D3, int MR, bool ZW = false> struct conv_geom {
    static constexpr int TW = 16, TH = ZW ? MR : 4 * MR, TD = ZW ? 4 : 1;
    static constexpr int KD = D3 ? KS : 1;
    static constexpr int STDg = D3 ? ST : 1;
    static constexpr int HW = (TW - 1) * ST + KS, HH = (TH - 1) * ST + KS, HD = (TD - 1) * STDg + KD;
    static constexpr int HP = HD * HH * HW;
    static constexpr int NTAPS = KD * KS * KS;
    static constexpr int PAD = (KS - ST) / 2;
    static constexpr int STD = D3 ? ST : 1;
};

constexpr int HALO_DUMMY = 8;                      // elements behind each halo buffer: target of the stores of a thread's unused last unit
constexpr int CONV_MAX_AFFINE_C = 1024;            // channels (both sources) whose scale/shift are cached in LDS

template <typename T, int KS, int ST, bool D3, int KC, int MR, bool ZW = false>
__host__ __device__ constexpr size_t conv_lds_fixed_bytes(int NT) {
    return 2 * ((size_t)conv_geom<KS, ST, D3, MR, ZW>::HP * pix_stride<T, KC>() + HALO_DUMMY) * sizeof(T)   // two halo buffers
           + 4 * 2 * 16 * NT * sizeof(float);                                            // block statistics: [wave][S | Q][16*NT]
}

// Weights that do not fit LDS whole are staged per item (one K-chunk: STEPS x NT fragment blocks) through
// registers into an LDS buffer -- when that takes <= 14 fragments per thread and fits (the predicate still counts two
// buffers, as rounds 1-3 had: the set of staged instances is unchanged).
template <typename T, int KS, int ST, bool D3, int KC, int NT, int MR, bool ZW>
__host__ __device__ constexpr bool conv_wstaged() {
    typedef conv_geom<KS, ST, D3, MR, ZW> G;
    constexpr int STEPS = (G::NTAPS * (KC / 8) + 3) / 4;
    return (STEPS * NT + 3) / 4 <= 14 &&
           conv_lds_fixed_bytes<T, KS, ST, D3, KC, MR, ZW>(NT) + 2 * CONV_MAX_AFFINE_C * sizeof(float) + 2 * (size_t)STEPS * NT * 512 * sizeof(T) <= 158 * 1024;
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
    static __device__ __forceinline__ uint4 pack(const float v[8]) { return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])); }
    static __device__ __forceinline__ uint4 select(bool keep, const uint4& a) { return make_uint4(keep ? a.x : 0u, keep ? a.y : 0u, keep ? a.z : 0u, keep ? a.w : 0u); }
};
template <> struct frag<float> {
    typedef f32x8 type;
    static __device__ __forceinline__ f32x8 load(const float* p) { f32x8 r; r.lo = *(const float4*)p; r.hi = *(const float4*)(p + 4); return r; }
    static __device__ __forceinline__ f32x8 zero() { f32x8 r; r.lo = make_float4(0, 0, 0, 0); r.hi = r.lo; return r; }
    static __device__ __forceinline__ void unpack(const f32x8& a, float v[8]) {
        v[0] = a.lo.x; v[1] = a.lo.y; v[2] = a.lo.z; v[3] = a.lo.w; v[4] = a.hi.x; v[5] = a.hi.y; v[6] = a.hi.z; v[7] = a.hi.w;
    }
    static __device__ __forceinline__ void store(float* p, const f32x8& a) { *(float4*)p = a.lo; *(float4*)(p + 4) = a.hi; }
    static __device__ __forceinline__ f32x8 pack(const float v[8]) { f32x8 r; r.lo = make_float4(v[0], v[1], v[2], v[3]); r.hi = make_float4(v[4], v[5], v[6], v[7]); return r; }
    static __device__ __forceinline__ f32x8 select(bool keep, const f32x8& a) {
        f32x8 r;
        r.lo = make_float4(keep ? a.lo.x : 0.f, keep ? a.lo.y : 0.f, keep ? a.lo.z : 0.f, keep ? a.lo.w : 0.f);
        r.hi = make_float4(keep ? a.hi.x : 0.f, keep ? a.hi.y : 0.f, keep ? a.hi.z : 0.f, keep ? a.hi.w : 0.f);
        return r;
    }
};

// ---- register-staged prefetch of one halo (item = tile x K-chunk) ---------------------------------
// Everything that does not depend on the tile is computed ONCE per thread (unit -> halo coordinates,
// LDS offset); per item a unit costs a handful of VALU instructions:
//   * bounds: the halo coordinates are packed (hz<<20 | hy<<10 | hx, 9 bits + 1 guard bit each); adding the
//     packed (512 - lo) and (512 - hi) of the wave-uniform valid range sets the guard bit of a field iff
//     f >= lo resp. f >= hi, so "inside" is ((d+PA) ^ (d+PB)) & GUARD == GUARD;
//   * address: wave-uniform 64-bit base (SGPR pair) + 32-bit lane offset -> global_load saddr form;
//   * NO branch around a load: out-of-range units read the tile's first output pixel (always inside the
//     input) and are zeroed at commit time.  A load inside a divergent `if` makes the compiler put an
//     s_waitcnt vmcnt(0) in front of every one of them, which serialises the whole prefetch.
// KEEPM: element keep masks (nn.Dropout) are supported -- 2D only: no 3D network of the path has element dropout
// (the V-Net uses Dropout3d = channel multipliers), and their 2 registers per unit are what the 3D bricks lack.
template <typename T, int UNITS, bool ADD2, bool KEEPM = true> struct halo_regs {
    typename frag<T>::type raw[UNITS];
    typename frag<T>::type raw2[ADD2 ? UNITS : 1];
    uint2 keep[KEEPM ? UNITS : 1];
    unsigned ok;               // bit j: unit j is inside the input (else zero padding)
};

constexpr int UNIT_UNUSED = 511 << 20;                          // z field beyond any halo: never inside
constexpr unsigned UNIT_GUARD = (1u << 29) | (1u << 19) | (1u << 9);

template <int UNITS> struct unit_desc {
    int hzyx[UNITS];           // hz << 20 | hy << 10 | hx   (UNIT_UNUSED: this thread has no such unit; only the last can be)
    int rel[UNITS];            // (hz*IH + hy)*IW + hx : input pixel offset from the halo origin
    int lds[UNITS];            // pix*PS + cgl*8 (elements)
    int c8;                    // first of this thread's 8 channels inside a K-chunk (256 % GPT == 0: same for all its units)
};

template <typename G, int GPT, int PS, int UNITS>
__device__ __forceinline__ void make_units(unit_desc<UNITS>& U, int IH, int IW) {
    U.c8 = (threadIdx.x % GPT) * 8;
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        const int u = threadIdx.x + 256 * j;
        U.hzyx[j] = UNIT_UNUSED; U.rel[j] = 0; U.lds[j] = G::HP * PS;    // dummy slot: every unit is staged and stored unconditionally
        if (u < G::HP * GPT) {
            const int pix = u / GPT, cgl = u % GPT;
            const int hx = pix % G::HW, hy = (pix / G::HW) % G::HH, hz = pix / (G::HW * G::HH);
            U.hzyx[j] = (hz << 20) | (hy << 10) | hx;
            U.rel[j] = (hz * IH + hy) * IW + hx;
            U.lds[j] = pix * PS + cgl * 8;
        }
    }
}

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

// LANESEL: the K-chunk straddles the two concatenated sources (s0.C % KC != 0), so the source is a per-thread
// choice (64-bit lane addresses); otherwise it is wave-uniform per item and everything below stays scalar.
// ONE: the launch has exactly one source (most layers): every "which source" select and all of s1 disappear.
template <typename T, typename G, bool D3, int ST, int KC, bool ADD2, int UNITS, bool LANESEL, bool ONE>
__device__ __forceinline__ void halo_issue_impl(halo_regs<T, UNITS, ADD2, !D3>& R, const unit_desc<UNITS>& U, const src_scalars& s0, const src_scalars& s1,
                                                int ID, int IH, int IW, int n, int z0, int y0, int x0, int chunk) {
    const int gz0 = z0 * G::STD - (D3 ? G::PAD : 0), gy0 = y0 * ST - G::PAD, gx0 = x0 * ST - G::PAD;
    const long gp0 = (((long)n * ID + gz0) * IH + gy0) * IW + gx0;           // halo origin (may lie outside: only offsets that pass the bounds test are used)
    const int loz = max(0, -gz0), hiz = min(G::HD, ID - gz0), loy = max(0, -gy0), hiy = min(G::HH, IH - gy0), lox = max(0, -gx0), hix = min(G::HW, IW - gx0);
    const unsigned PA = ((unsigned)(512 - loz) << 20) | ((unsigned)(512 - loy) << 10) | (unsigned)(512 - lox);
    const unsigned PB = ((unsigned)(512 - hiz) << 20) | ((unsigned)(512 - hiy) << 10) | (unsigned)(512 - hix);
    const unsigned rsafe = ((D3 ? G::PAD : 0) * IH + G::PAD) * IW + G::PAD;
    const int cb = chunk * KC;
    const bool second = (ADD2 || ONE) ? false : (LANESEL ? (cb + U.c8 >= s0.C) : (cb >= s0.C));
    const int cs = second ? cb - s0.C : cb;                      // first channel of the chunk inside the source (can be < 0 for LANESEL lanes of s1: + c8 >= 0)
    const int ld = second ? s1.ld : s0.ld;
    // (s0.ptr + (second ? s1.ptr - s0.ptr : 0), not second ? s1.ptr : s0.ptr: the compiler turns a select of the two loaded pointers into a load
    //  from a selected address, which pins both in scratch -- 24 B per lane and two scratch loads per tile in the 3D weight-gradient kernels)
    const long dptr = (const char*)s1.ptr - (const char*)s0.ptr;
    const char* base = (const char*)s0.ptr + (second ? dptr : 0l) + (gp0 * ld + (second ? s1.coff : s0.coff) + cs) * (long)sizeof(T);
    const unsigned ldb = ld * sizeof(T), lane_c = U.c8 * sizeof(T);
    unsigned okm = 0, r[UNITS];
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        const unsigned d = (unsigned)U.hzyx[j];
        const bool ok = (((d + PA) ^ (d + PB)) & UNIT_GUARD) == UNIT_GUARD;
        okm |= ok ? (1u << j) : 0u;
        r[j] = ok ? (unsigned)U.rel[j] : rsafe;
    }
    R.ok = okm;
    if (CHAP_ABLATE & 16) return;
#pragma unroll
    for (int j = 0; j < UNITS; ++j) R.raw[j] = frag<T>::load((const T*)(base + (__umul24(r[j], ldb) + lane_c)));
    if (ADD2) {
        const char* base2 = (const char*)s1.ptr + (gp0 * s1.ld + s1.coff + cb) * (long)sizeof(T);
        const unsigned ldb2 = s1.ld * sizeof(T);
#pragma unroll
        for (int j = 0; j < UNITS; ++j) R.raw2[j] = frag<T>::load((const T*)(base2 + (__umul24(r[j], ldb2) + lane_c)));
    }
    // element keep masks (Dropout): u8 [pixel][C] of the source.  The registers are (re)defined on BOTH paths: a
    // value carried over from the previous item would come back as register copies behind the loads, each with
    // its own s_waitcnt -- i.e. a synchronous prefetch.
    const bool hk = second ? s1.has_keep : s0.has_keep;
    if constexpr (!D3) {
    if (!(LANESEL ? (s0.has_keep || s1.has_keep) : hk)) {
#pragma unroll
        for (int j = 0; j < UNITS; ++j) R.keep[j] = make_uint2(0u, 0u);
    } else {                                                     // wave-uniform
        const int kC = second ? s1.C : s0.C;
        const uint8_t* kb = (second ? s1.keep : s0.keep) + (gp0 * kC + cs);
#pragma unroll
        for (int j = 0; j < UNITS; ++j) {
            const uint8_t* kp = kb + (__umul24(r[j], (unsigned)kC) + (unsigned)U.c8);
            // LANESEL lanes whose source has no mask read 8 bytes of their own data instead (ignored at commit)
            if (LANESEL) kp = hk ? kp : (const uint8_t*)(base + (__umul24(r[j], ldb) + lane_c));
            R.keep[j] = *(const uint2*)kp;
        }
    }
    }
}

template <typename T, int KS, int ST, bool D3, int KC, int MR, bool ADD2, int UNITS, bool ZW = false, bool ONE = false>
__device__ __forceinline__ void halo_issue(halo_regs<T, UNITS, ADD2, !D3>& R, const unit_desc<UNITS>& U, const src_scalars& s0, const src_scalars& s1,
                                           int ID, int IH, int IW, int n, int z0, int y0, int x0, int chunk, bool lanesel) {
    typedef conv_geom<KS, ST, D3, MR, ZW> G;
    if (ONE) halo_issue_impl<T, G, D3, ST, KC, ADD2, UNITS, false, true>(R, U, s0, s0, ID, IH, IW, n, z0, y0, x0, chunk);
    else if (!ADD2 && lanesel) halo_issue_impl<T, G, D3, ST, KC, ADD2, UNITS, true, false>(R, U, s0, s1, ID, IH, IW, n, z0, y0, x0, chunk);
    else halo_issue_impl<T, G, D3, ST, KC, ADD2, UNITS, false, false>(R, U, s0, s1, ID, IH, IW, n, z0, y0, x0, chunk);
}

// affine cache layout in LDS: [src][scale | shift][CONV_MAX_AFFINE_C/2] (identity when a source has none).
// Packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32, 2 elements per VALU instruction): the commit phase is
// VALU-issue-bound.  leaky(z) = max(z, slope*z) for 0 <= slope <= 1 (slope_eff = 1: no activation).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// scale/shift of the 8 channels a thread stages, held in registers (per K-chunk; they do not depend on the tile)
template <bool ADD2> struct aff_regs { f32x2 a[4], b[4], a2[ADD2 ? 4 : 1], b2[ADD2 ? 4 : 1]; };

// AFFC: capacity of the cache per source (scale | shift halves of AFFC / 2 channels each); the weight-gradient bricks use a small one
template <int KC, bool ADD2, bool ONE = false, int AFFC = CONV_MAX_AFFINE_C>
__device__ __forceinline__ void load_aff(aff_regs<ADD2>& A, const float* aff, int c8, int C0, int chunk, bool lanesel) {
    const int cb = chunk * KC;
    const bool second = (ADD2 || ONE) ? false : (lanesel ? (cb + c8 >= C0) : (cb >= C0));
    const float* af = aff + (second ? AFFC + cb - C0 : cb) + c8;
#pragma unroll
    for (int k = 0; k < 4; ++k) { A.a[k] = *(const f32x2*)(af + 2 * k); A.b[k] = *(const f32x2*)(af + AFFC / 2 + 2 * k); }
    if (ADD2) {
        const float* af2 = aff + AFFC + cb + c8;
#pragma unroll
        for (int k = 0; k < 4; ++k) { A.a2[k] = *(const f32x2*)(af2 + 2 * k); A.b2[k] = *(const f32x2*)(af2 + AFFC / 2 + 2 * k); }
    }
}

// v <- leaky(v * a + b) with slope_eff folded as max(z, slope*z)
__device__ __forceinline__ void affine_act8(float v[8], const f32x2 a[4], const f32x2 b[4], float slope_eff) {
    const f32x2 sl = {slope_eff, slope_eff};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x2 z = {v[2 * k], v[2 * k + 1]};
        z = z * a[k] + b[k];
        const f32x2 m = z * sl;
        v[2 * k] = fmaxf(z.x, m.x); v[2 * k + 1] = fmaxf(z.y, m.y);
    }
}
__device__ __forceinline__ void keep8(float v[8], uint2 keep, float keep_scale) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t w = j < 4 ? keep.x : keep.y;
        v[j] = ((w >> (8 * (j & 3))) & 0xff) ? v[j] * keep_scale : 0.f;
    }
}

template <typename T, int KC, bool ADD2, int UNITS, bool LANESEL, bool KEEPM, bool ONE, int AFFC = CONV_MAX_AFFINE_C>
__device__ __forceinline__ void halo_commit_impl(const halo_regs<T, UNITS, ADD2, KEEPM>& R, const unit_desc<UNITS>& U, T* halo, const src_scalars& s0, const src_scalars& s1,
                                                 const float* aff, bool plain, int n, int chunk) {
    // no per-unit branches: a path that skips a unit would leave its prefetch load un-waited as far as the compiler
    // can tell, and the next item's issue phase would then wait for every outstanding memory operation
    if (plain) {
#pragma unroll
        for (int j = 0; j < UNITS; ++j) frag<T>::store(halo + U.lds[j], frag<T>::select((R.ok >> j) & 1u, R.raw[j]));
        return;
    }
    const int cb = chunk * KC;
    const bool second = (ADD2 || ONE) ? false : (LANESEL ? (cb + U.c8 >= s0.C) : (cb >= s0.C));
    const float se = second ? s1.slope_eff : s0.slope_eff;
    const bool hk = second ? s1.has_keep : s0.has_keep;
    const float ks = second ? s1.keep_scale : s0.keep_scale;
    const bool hcm = second ? s1.has_cm : s0.has_cm;
    // scale/shift of this thread's 8 channels: 4 LDS reads per item are cheaper than 16 registers held across the MFMA loop
    aff_regs<ADD2> A;
    load_aff<KC, ADD2, ONE, AFFC>(A, aff, U.c8, s0.C, chunk, LANESEL);
    f32x2 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = A.a[k]; b[k] = A.b[k]; }
    // per-sample channel multipliers (Dropout3d / channel masks; rare, so loaded here rather than carried in registers).
    // They are >= 0: cm * leaky(a x + b) = leaky((cm a) x + cm b)
    if (LANESEL ? (s0.has_cm || s1.has_cm) : hcm) {
        // (the compiler hoists chan_mul + c8 out of the tile loop as a 64-bit per-lane pointer: two VGPRs held across the loop for a path that is
        //  rarely taken -- conv_wp_kernel<16,1> sits exactly on its 128-VGPR cap and any extra live value there goes to scratch)
        const int cs = second ? cb - s0.C : cb;
        const float* cpu = second ? s1.chan_mul + (long)n * s1.C + cs : s0.chan_mul + (long)n * s0.C + cs;
        unsigned c8o = (unsigned)U.c8;
        asm volatile("" : "+v"(c8o));                            // defined HERE as far as the compiler can tell: the sum below stays on this path
        const float* cp = cpu + c8o;
        if (LANESEL) cp = hcm ? cp : (s0.has_cm ? s0.chan_mul : s1.chan_mul);
        float cm[8];
        ld8(cp, cm);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x2 m = {cm[2 * k], cm[2 * k + 1]};
            if (LANESEL) { m.x = hcm ? m.x : 1.f; m.y = hcm ? m.y : 1.f; }
            a[k] *= m; b[k] *= m;
        }
    }
    const bool anyk = KEEPM && (LANESEL ? (s0.has_keep || s1.has_keep) : hk);
#pragma unroll
    for (int j = 0; j < UNITS; ++j) {
        float v[8];
        frag<T>::unpack(R.raw[j], v);
        affine_act8(v, a, b, se);
        if (anyk) {
            uint2 kp = R.keep[KEEPM ? j : 0];
            if (LANESEL) { kp.x = hk ? kp.x : 0x01010101u; kp.y = hk ? kp.y : 0x01010101u; }
            keep8(v, kp, (LANESEL && !hk) ? 1.f : ks);
        }
        if (ADD2) {
            float w[8];
            frag<T>::unpack(R.raw2[j], w);
            affine_act8(w, A.a2, A.b2, s1.slope_eff);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += w[k];
        }
        frag<T>::store(halo + U.lds[j], frag<T>::select((R.ok >> j) & 1u, frag<T>::pack(v)));
    }
}

template <typename T, int KC, bool ADD2, int UNITS, bool KEEPM, bool ONE = false, int AFFC = CONV_MAX_AFFINE_C>
__device__ __forceinline__ void halo_commit(const halo_regs<T, UNITS, ADD2, KEEPM>& R, const unit_desc<UNITS>& U, T* halo, const src_scalars& s0, const src_scalars& s1,
                                            const float* aff, bool plain, int n, int chunk, bool lanesel) {
    if (ONE) halo_commit_impl<T, KC, ADD2, UNITS, false, KEEPM, true, AFFC>(R, U, halo, s0, s0, aff, plain, n, chunk);
    else if (!ADD2 && lanesel) halo_commit_impl<T, KC, ADD2, UNITS, true, KEEPM, false, AFFC>(R, U, halo, s0, s1, aff, plain, n, chunk);
    else halo_commit_impl<T, KC, ADD2, UNITS, false, KEEPM, false, AFFC>(R, U, halo, s0, s1, aff, plain, n, chunk);
}

// tile index -> (n, z0, y0, x0); tiles_z = number of tile layers along D (D itself when the tile is one plane)
template <int TH, int TW, int TD = 1>
__device__ __forceinline__ void tile_coords(long tile, int tiles_x, int tiles_y, int tiles_z, int& n, int& z0, int& y0, int& x0) {
    const unsigned t = (unsigned)tile;
    const unsigned tx = t % (unsigned)tiles_x; unsigned r = t / (unsigned)tiles_x;
    const unsigned ty = r % (unsigned)tiles_y; r /= (unsigned)tiles_y;
    z0 = (int)(r % (unsigned)tiles_z) * TD; n = (int)(r / (unsigned)tiles_z);
    x0 = (int)tx * TW; y0 = (int)ty * TH;
}

// __launch_bounds__ second argument of the trampoline (launch.h) that runs this body
// (round 4: 2 instead of 1 for the 3D and the 32-channel-chunk 2D instances.  Unbounded they took 268-350 registers per lane (VGPRs + accumulation registers)
//  and ONE block fitted a CU whatever its LDS; at 256 two do.  The staged 32-output-channel 3D instances pay 160-330 B per lane of scratch for it.)
//  The fp32 instances (the parity mode) keep the build of rounds 1-3: no bound, two staged weight buffers.  Both changes leave every conv OUTPUT bit for bit
//  (profiles/r04_wbuf1_bits.log), but one weight buffer changes a launch's LDS size, hence its persistent grid, hence how the BatchNorm statistics are dealt to
//  partial slots (5e-9 .. 1e-8 relative in fp32) -- enough to re-roll the rounding noise of the fp32 iteration's loss (config 0: 2.2e-6 -> 1.2e-5 from the fp32
//  oracle, tests/test_iteration_conditioning_gpu.py), whose bounds are pinned to the grids of round 3.)
template <int KC, bool D3, typename T = bf16_t> constexpr int conv_min_waves() { return CHAP_CONV_MINWAVES > 1 ? CHAP_CONV_MINWAVES : (KC == 16 && !D3 ? 3 : (sizeof(T) == 2 ? 2 : 1)); }
template <typename T> constexpr bool conv_one_wbuf() { return sizeof(T) == 2; }      // staged weights: ONE LDS buffer (bf16) / two (fp32)

template <typename T, int KS, int ST, bool D3, int KC, int NT, int MR, bool ADD2, bool WLDS, bool ZW = false, bool ONE = false>
__device__ __forceinline__ void conv_fwd_kernel(const chap_conv_params& P) {
    typedef conv_geom<KS, ST, D3, MR, ZW> G;
    typedef typename frag<T>::type F;
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>();
    constexpr int NP = G::NTAPS * GPT, STEPS = (NP + 3) / 4;
    constexpr int UNITS = (G::HP * GPT + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CHAP_STAMP_P(0);
    T* halo0 = (T*)smem;
    T* halo1 = halo0 + (size_t)G::HP * PS + HALO_DUMMY;
    float* bstat = (float*)(halo1 + (size_t)G::HP * PS + HALO_DUMMY);
    float* aff = bstat + 4 * 2 * 16 * NT;                   // [2 sources][scale | shift][CONV_MAX_AFFINE_C/2]
    T* wlds = (T*)(aff + 2 * CONV_MAX_AFFINE_C);            // WLDS: all weights; staged mode: one buffer of one K-chunk's weights

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const int tiles_x = (P.W + G::TW - 1) / G::TW, tiles_y = (P.H + G::TH - 1) / G::TH, tiles_z = (P.D + G::TD - 1) / G::TD;
    const long ntiles = (long)P.N * tiles_z * tiles_y * tiles_x;
    const int ntiles_total = (P.Cout + 15) >> 4;
    const int nt0 = blockIdx.y * NT;
    const int Ctot = P.combine == 0 ? (P.src[0].C + (P.nsrc > 1 ? P.src[1].C : 0)) : P.src[0].C;
    const int nchunks = Ctot / KC;
    const bool plain = (CHAP_ABLATE & 4) ||
                       (!ADD2 && P.src[0].scale == nullptr && !P.src[0].act && P.src[0].keep == nullptr && P.src[0].chan_mul == nullptr &&
                        (P.nsrc < 2 || (P.src[1].scale == nullptr && !P.src[1].act && P.src[1].keep == nullptr && P.src[1].chan_mul == nullptr)));
    const src_scalars s0 = make_scalars(P.src[0]);
    const src_scalars s1 = make_scalars((!ONE && P.nsrc > 1) ? P.src[1] : P.src[0]);

    // XCD-aware tile walk: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous range
    // of tiles so that the halos re-read by neighbouring tiles are L2 hits.
    const int nb = gridDim.x;
    const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
    const int bpx = (nb + 7 - xcd) >> 3;                        // blocks on this XCD group
    // each block owns a CONTIGUOUS run of this XCD's tiles (consecutive tiles share halo rows; the walk is +1 with carries)
    const unsigned per = ((unsigned)ntiles + 7u) / 8u;
    const unsigned t_lo = per * (unsigned)xcd, t_hi = min(t_lo + per, (unsigned)ntiles);
    const unsigned range = t_hi > t_lo ? t_hi - t_lo : 0u;
    const unsigned tq = range / (unsigned)bpx, trem = range - tq * (unsigned)bpx;
    // ILV (CHAP_ZW_INTERLEAVE, 3D bricks, round 3): the XCD's blocks take its tiles ROUND-ROBIN (block b: t_lo + b, t_lo + b + bpx, ...) instead of
    // one contiguous run each, so that at any time they work on ~bpx CONSECUTIVE bricks -- a compact region whose halos overlap -- rather
    // than on bpx runs spread over the whole 8 MB range of the XCD (more than its 4 MB L2).  Measured on the 16->16 layer at 112x112x80, N = 2
    // (profiles/r03_pmc_traffic_walk_*.jsonl): L2-miss reads 106.6 -> 72.0 MB per launch (1.66x -> 1.12x the algorithmic 64.2 MB; reads +
    // writes 1.33x -> 1.06x), kernel time unchanged (65.0 -> 65.6 us: the kernel is not traffic-bound), 3D step 14.97 -> 14.92 ms
    constexpr bool ILV = (CHAP_ZW_INTERLEAVE != 0) && ZW;
    const unsigned t_first = ILV ? t_lo + (unsigned)bix : t_lo + (unsigned)bix * tq + min((unsigned)bix, trem);
    const long my_tiles = ILV ? (range > (unsigned)bix ? (long)((range - (unsigned)bix + (unsigned)bpx - 1u) / (unsigned)bpx) : 0l)
                              : (long)(tq + ((unsigned)bix < trem ? 1u : 0u));
    const long nitems = my_tiles * nchunks;
    // digits of the walk's step (1, or bpx when interleaved) in the mixed radix (z, x, y | x, y, z) of the tile coordinates
    int stz = 0, stx = 0, sty = 0, stn = 0;
    if (ILV) { unsigned q = (unsigned)bpx; stz = q % (unsigned)tiles_z; q /= (unsigned)tiles_z; stx = q % (unsigned)tiles_x; q /= (unsigned)tiles_x; sty = q % (unsigned)tiles_y; stn = q / (unsigned)tiles_y; }

    CHAP_STAMP_P(6);
    // ---- one-time per thread: unit descriptors, MFMA fragment offsets ----
    unit_desc<UNITS> U;
    make_units<G, GPT, PS, UNITS>(U, P.IH, P.IW);
    const bool lanesel = !ADD2 && !ONE && P.nsrc > 1 && (P.src[0].C % KC) != 0;   // a K-chunk straddles the two concatenated sources
    int xoff[STEPS];                                            // LDS element offset of this lane's B fragment (row m = 0), -1 = zero fragment
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int p = step * 4 + g;
        const int tap = p / GPT, cgl = p % GPT;
        const int dx = tap % KS, dy = (tap / KS) % KS, dz = tap / (KS * KS);
        xoff[step] = p < NP ? (ZW ? (((dz + wave * G::STDg) * G::HH + dy) * G::HW + px * ST + dx) * PS + cgl * 8
                                  : ((dz * G::HH + (wave * MR) * ST + dy) * G::HW + px * ST + dx) * PS + cgl * 8) : -1;
    }

    CHAP_STAMP_P(7);
    // ---- one-time per block.  Every global load of the prologue (first halo, scale/shift, resident weights,
    // bias) is issued before the first dependent LDS store, so that the block pays ONE memory round trip before
    // its first tile instead of four in a row (a block only owns a handful of tiles: this is a large part of
    // its life).  No load sits inside a divergent branch (see halo_issue).
    halo_regs<T, UNITS, ADD2, !D3> R;
    // tile walk: tiles t_first, t_first + 1, ...: the (x, y, z, n) tile coordinates advance by +1 with carries
    int ntx = 0, nty = 0, ntz = 0, nn = 0;                      // tile coordinates of the item being prefetched
    {
        // 3D: z runs fastest -- consecutive bricks of a run share 2 of their 6 halo planes (x neighbours: 2 of 18 columns),
        // and the x / y neighbours are only tiles_z bricks away, i.e. still in this XCD's L2 (measured on the 16->16 layer at
        // 112x112x80: FETCH_SIZE 1.95x the algorithmic read with x fastest)
        unsigned t = t_first;
        if (D3) { ntz = t % (unsigned)tiles_z; t /= (unsigned)tiles_z; }
        ntx = t % (unsigned)tiles_x; t /= (unsigned)tiles_x;
        nty = t % (unsigned)tiles_y; t /= (unsigned)tiles_y;
        if (D3) nn = t; else { ntz = t % (unsigned)tiles_z; nn = t / (unsigned)tiles_z; }
    }
    if (nitems > 0)
        halo_issue<T, KS, ST, D3, KC, MR, ADD2, UNITS, ZW, ONE>(R, U, s0, s1, P.ID, P.IH, P.IW, nn, ntz * G::TD, nty * G::TH, ntx * G::TW, 0, lanesel);

    // scale/shift of both sources -> registers (CONV_MAX_AFFINE_C/2 = 512 channels per source: 2 per thread)
    float asc[2][2], ash[2][2];
    if (!plain) {
#pragma unroll
        for (int s = 0; s < (ONE ? 1 : 2); ++s) {
            const chap_src_t& S = P.src[s < P.nsrc ? s : 0];
            const bool has = S.scale != nullptr;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = min((int)threadIdx.x + 256 * k, S.C - 1);
                asc[s][k] = 1.f; ash[s][k] = 0.f;
                if (has) { asc[s][k] = S.scale[c]; ash[s][k] = S.shift[c]; }      // wave-uniform branch
            }
        }
    }
    const long wstep = (long)ntiles_total * 64 * 8;             // packed elements per (chunk, step)
    // Staged weights (layers whose weights do not fit LDS whole): the STEPS x NT fragment blocks (already in lane
    // order) of ONE K-chunk are prefetched into registers next to the halo and written to the LDS buffer the
    // NEXT item reads.  Streaming them from L2 inside the tap loop exposes a load latency per step.
    constexpr bool WST = !WLDS && conv_wstaged<T, KS, ST, D3, KC, NT, MR, ZW>();
    constexpr int WFR = WST ? (STEPS * NT + 3) / 4 : 1;          // fragments per thread (wave w takes blocks w, w+4, ..)
    F wreg[WFR];
    auto wstage_issue = [&](int chunk_) {
        if constexpr (WST) {
            const T* src = (const T*)P.wpacked + (long)chunk_ * STEPS * wstep + lane * 8;
#pragma unroll
            for (int k = 0; k < WFR; ++k) {
                const int i = min(4 * k + wave, STEPS * NT - 1);  // clamp (wave-uniform): no branch around the load
                const int step = i / NT, t = i % NT;
                const int tt = nt0 + t < ntiles_total ? nt0 + t : nt0;          // clamp: those channels are never stored
                wreg[k] = frag<T>::load(src + (long)step * wstep + (long)tt * 512);
            }
        }
    };
    auto wstage_commit = [&](int buf) {
        if constexpr (WST) {
            T* dst = wlds + (size_t)buf * (STEPS * NT * 512) + lane * 8;
#pragma unroll
            for (int k = 0; k < WFR; ++k) {
                const int i = min(4 * k + wave, STEPS * NT - 1);  // clamped waves rewrite the last block with identical data: every prefetch register is consumed on every path
                frag<T>::store(dst + i * 512, wreg[k]);
            }
        }
    };
    if (WST && nitems > 0) wstage_issue(0);
    // resident weights: [chunk][step][t < NT][64 lanes][8]; first batch of 8 fragments per thread in flight now
    const long wtot = WLDS ? (long)nchunks * STEPS * NT * 64 : 0;
    const unsigned uwtot = (unsigned)wtot;
    auto wload = [&](unsigned i) -> F {                          // 32-bit index math (the packed weights are far below 2 GB)
        const unsigned ic = i < uwtot ? i : 0u;                  // clamp, no branch around the load
        const unsigned ln = ic & 63u; unsigned r = ic >> 6;
        unsigned t = r % (unsigned)NT; r /= (unsigned)NT;        // r = chunk*STEPS + step
        t = nt0 + (int)t < ntiles_total ? t : 0u;
        return frag<T>::load((const T*)((const char*)P.wpacked + (size_t)((r * (unsigned)wstep + ((unsigned)nt0 + t) * 512u + ln * 8u) * (unsigned)sizeof(T))));
    };
    auto wkeep = [&](unsigned i) -> bool { return i < uwtot && nt0 + (int)((i >> 6) % (unsigned)NT) < ntiles_total; };
    F wf0[8];
    if (WLDS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) wf0[k] = wload(threadIdx.x + 256u * k);
    }

    f32x4 acc[MR][NT];
    float ssum[NT][4], ssq[NT][4], bj[NT][4], cj[NT][4];        // cj: shift of the statistics (moments of v - c)
    // bias: all loads first (index clamped, no branch, no use in between), the selects afterwards -- a use right
    // behind each load makes the compiler wait for it (and for every prefetch issued above) NT*4 times in a row
    {
        const float* bsrc = P.bias ? P.bias : (const float*)P.wpacked;          // wave-uniform; any valid address when there is no bias
        const bool has_shift = P.stats != nullptr && P.stats_shift != nullptr;
        const float* csrc = has_shift ? P.stats_shift : (const float*)P.wpacked;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nl = (nt0 + t) * 16 + 4 * g;
            const int cb = P.out_mode == 1 ? (nl % P.out_Cn) : nl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ssum[t][j] = 0.f; ssq[t][j] = 0.f;
                bj[t][j] = bsrc[(P.bias && nl + j < P.Cout) ? cb + j : 0];
                cj[t][j] = csrc[(has_shift && nl + j < P.Cout) ? cb + j : 0];
            }
        }
    }
    // ---- now the dependent LDS stores
    if (!plain) {
#pragma unroll
        for (int s = 0; s < (ONE ? 1 : 2); ++s) {
            if (s < P.nsrc) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int c = threadIdx.x + 256 * k;
                    if (c < P.src[s].C) { aff[s * CONV_MAX_AFFINE_C + c] = asc[s][k]; aff[s * CONV_MAX_AFFINE_C + CONV_MAX_AFFINE_C / 2 + c] = ash[s][k]; }
                }
            }
        }
    }
    if (WLDS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned i = threadIdx.x + 256u * k;
            if (i < uwtot) frag<T>::store(wlds + i * 8u, wkeep(i) ? wf0[k] : frag<T>::zero());
        }
        for (unsigned i0 = threadIdx.x + 256u * 8u; i0 < uwtot; i0 += 256u * 8u) {     // larger layers: further batches of 8
            F f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = wload(i0 + 256u * k);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned i = i0 + 256u * k;
                if (i < uwtot) frag<T>::store(wlds + i * 8u, wkeep(i) ? f[k] : frag<T>::zero());
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nl = (nt0 + t) * 16 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bj[t][j] = (P.bias && nl + j < P.Cout) ? bj[t][j] : 0.f;
            cj[t][j] = (P.stats != nullptr && P.stats_shift != nullptr && nl + j < P.Cout) ? cj[t][j] : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(bj[t][j])); asm volatile("" : "+v"(cj[t][j])); }   // land the bias loads here, not behind the first prefetch of the item loop
    const bool do_stats = P.stats != nullptr;
    const int SD2 = (P.dims == 3) ? 2 : 1;
    // per-lane output offsets relative to the tile origin (elements), one per (t): row term added per m
    int ooff[NT], orow, oplane;                                 // orow / oplane = element stride of one tile row / z-plane
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
                ooff[t] = px * P.out_ld + P.out_coff + ((P.out2 && nl >= P.out2_from) ? nl - P.out2_from : nl);      // (out2: the upper channel tiles land in a second dense tensor)
            }
        }
        orow = P.out_planar ? P.W : (P.out_mode == 1 ? 2 * OW * P.out_ld : P.W * P.out_ld);
        oplane = P.out_planar ? P.H * P.W : (P.out_mode == 1 ? 2 * OH * OW * P.out_ld : P.H * P.W * P.out_ld);   // one z-plane (ZW tiles)
    }

    int n = 0, z0 = 0, y0 = 0, x0 = 0;
    CHAP_STAMP_P(1);
    __syncthreads();                                            // affine cache (+ resident weights) visible
    CHAP_STAMP_P(2);
    if (nitems > 0) { halo_commit<T, KC, ADD2, UNITS, !D3, ONE>(R, U, halo0, s0, s1, aff, plain, nn, 0, lanesel); wstage_commit(0); }
    __syncthreads();
    CHAP_STAMP_P(3);

    int chunk = 0;
    for (long it = 0; it < nitems; ++it) {
        T* cur = (it & 1) ? halo1 : halo0;
        T* nxt = (it & 1) ? halo0 : halo1;
        CHAP_STAMP(0);
        if (chunk == 0) {
            n = nn; z0 = ntz * G::TD; y0 = nty * G::TH; x0 = ntx * G::TW;
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
                int c = 1;
                if (ILV) {
                    ntz += stz; c = ntz >= tiles_z; ntz -= c ? tiles_z : 0;
                    ntx += stx + c; c = ntx >= tiles_x; ntx -= c ? tiles_x : 0;
                    nty += sty + c; c = nty >= tiles_y; nty -= c ? tiles_y : 0;
                    nn += stn + c;
                } else {
                if (D3) { ntz += 1; c = ntz >= tiles_z; ntz -= c ? tiles_z : 0; }
                ntx += c; c = ntx >= tiles_x; ntx -= c ? tiles_x : 0;
                nty += c; c = nty >= tiles_y; nty -= c ? tiles_y : 0;
                if (D3) nn += c; else { ntz += c; c = ntz >= tiles_z; ntz -= c ? tiles_z : 0; nn += c; }
                }
            }
            halo_issue<T, KS, ST, D3, KC, MR, ADD2, UNITS, ZW, ONE>(R, U, s0, s1, P.ID, P.IH, P.IW, nn, ntz * G::TD, nty * G::TH, ntx * G::TW, nchunk, lanesel);
            wstage_issue(nchunk);
        }
        CHAP_STAMP(1);
        // ---- MFMA over the taps of this chunk
        const T* wc_g = (const T*)P.wpacked + (long)chunk * STEPS * wstep + ((long)nt0 * 64 + lane) * 8;
        const T* wc_l = wlds + ((long)(WLDS ? chunk : (conv_one_wbuf<T>() ? 0 : (int)(it & 1))) * STEPS * NT * 64 + lane) * 8;      // (staged mode, bf16: ONE buffer, see the commit at the end of the item)
        // software-pipelined: the fragments of step s+1 are requested before the MFMAs of step s are issued, so
        // the LDS (or L2) latency hides behind MR*NT MFMAs instead of preceding every one of them
        auto load_w = [&](int step, F (&wf)[NT]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (WLDS || WST) wf[t] = frag<T>::load(wc_l + ((long)step * NT + t) * 512);
                else wf[t] = frag<T>::load(wc_g + (long)step * wstep + (nt0 + t < ntiles_total ? t : 0) * 512);   // clamped: those channels are never stored
            }
        };
        auto load_x = [&](int step, F (&xf)[MR]) {
            const int xo = xoff[step];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (step * 4 + 3 < NP) xf[m] = frag<T>::load(cur + xo + m * (ST * G::HW * PS));        // every lane group has a tap
                else xf[m] = xo >= 0 ? frag<T>::load(cur + xo + m * (ST * G::HW * PS)) : frag<T>::zero();
            }
        };
        // pixel fragments (LDS) one step ahead; weight fragments one step ahead from LDS, TWO steps ahead when they
        // stream from L2 (one step of MR*NT MFMAs does not cover an L2 round trip)
        constexpr int WD = (WLDS || WST) ? 2 : 3;               // ring depth of the weight fragments
        F wq[WD][NT], xq[2][MR];
        constexpr int NSTEP = (CHAP_ABLATE & 2) ? 0 : STEPS;
        if (NSTEP > 0) { load_w(0, wq[0]); load_x(0, xq[0]); }
        if (WD == 3 && NSTEP > 1) load_w(1, wq[1]);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            if (step + WD - 1 < NSTEP) load_w(step + WD - 1, wq[(step + WD - 1) % WD]);
            if (step + 1 < NSTEP) load_x(step + 1, xq[(step + 1) & 1]);
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const F& w = wq[step % WD][t];
                    const F& x = xq[step & 1][m];
                    if (CHAP_ABLATE & 1) { float kx[8], kw[8]; frag<T>::unpack(x, kx); frag<T>::unpack(w, kw); asm volatile("" :: "v"(kx[0]), "v"(kx[7]), "v"(kw[0]), "v"(kw[7])); } else mma8(acc[m][t], w, x);
                }
        }
        CHAP_STAMP(2);
        // ---- epilogue after the last K-chunk of a tile: lane holds D[cout = 16*t + 4*g + j][pixel (row m, x = px)]
        if (chunk == nchunks - 1) {
            const bool xok = x0 + px < P.W;
            // wave-uniform 64-bit origin of this tile in the output, 32-bit lane offsets on top
            long o0;
            if (P.out_planar) o0 = ((long)n * P.Cout * P.D + z0) * P.H * P.W + (long)y0 * P.W + x0;     // + c*plane
            else if (P.out_mode == 1) o0 = ((((long)n * (P.D * SD2) + z0 * SD2) * (2 * P.H) + 2 * y0) * (2 * P.W) + 2 * x0) * P.out_ld;
            else o0 = ((((long)n * P.D + z0) * P.H + y0) * P.W + x0) * P.out_ld;
            const int zw_off = ZW ? wave * oplane : 0;
            const bool zok = !ZW || z0 + wave < P.D;
            if (P.out_planar || (P.Cout & 3)) {
                // heads (planar fp32 logits) and odd channel counts: element stores
                const long plane = (long)P.D * P.H * P.W;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int nl = (nt0 + t) * 16 + 4 * g;        // logical output channel of j = 0
#pragma unroll
                    for (int m = 0; m < MR; ++m) {
                        const int row = ZW ? m : wave * MR + m;
                        const bool valid = xok && zok && (y0 + row < P.H) && nl < P.Cout;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[m][t][j] + bj[t][j];
                        if (!valid) continue;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float vs = v[j] - cj[t][j]; ssum[t][j] += vs; ssq[t][j] += vs * vs; }
                        const long oi = o0 + row * orow + zw_off + ooff[t];
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (nl + j < P.Cout) {
                            if (P.out_planar) ((float*)P.out)[oi + (nl + j) * plane] = v[j];
                            else if (P.out_f32) ((float*)P.out)[oi + j] = v[j];
                            else ((T*)P.out)[oi + j] = elem<T>::put(v[j]);
                        }
                    }
                }
            } else {
                // channel-last, 4 channels per lane: vector stores at (uniform base) + (32-bit lane offset).  Interior
                // tiles with a full channel group (wave-uniform test; nearly all of them) skip the per-lane validity:
                // no selects in the statistics, no exec masking around the stores.
                char* ob = (char*)P.out + o0 * (P.out_f32 ? 4 : (long)sizeof(T));
                char* ob2 = P.out2 ? (char*)P.out2 + o0 * (long)sizeof(T) : ob;
                const bool full = (x0 + G::TW <= P.W) && (y0 + G::TH <= P.H) && (!ZW || z0 + G::TD <= P.D) && ((nt0 + NT) * 16 <= P.Cout);
                auto emit = [&](auto FULL) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const bool cok = (nt0 + t) * 16 + 4 * g < P.Cout;
                        char* obt = (P.out2 && (nt0 + t) * 16 >= P.out2_from) ? ob2 : ob;      // wave-uniform
#pragma unroll
                        for (int m = 0; m < MR; ++m) {
                            const int row = ZW ? m : wave * MR + m;
                            const bool valid = decltype(FULL)::value || (xok && zok && cok && (y0 + row < P.H));
                            float v[4];
                            // packed fp32 (v_pk_add_f32 / v_pk_fma_f32): bias, shifted moments -- the epilogue is VALU time of a VALU-bound kernel
                            const f32x2 v0 = (f32x2){acc[m][t][0], acc[m][t][1]} + (f32x2){bj[t][0], bj[t][1]};
                            const f32x2 v1 = (f32x2){acc[m][t][2], acc[m][t][3]} + (f32x2){bj[t][2], bj[t][3]};
                            v[0] = v0.x; v[1] = v0.y; v[2] = v1.x; v[3] = v1.y;
                            if (do_stats) {
                                f32x2 d0 = v0 - (f32x2){cj[t][0], cj[t][1]}, d1 = v1 - (f32x2){cj[t][2], cj[t][3]};
                                if (!decltype(FULL)::value) { d0.x = valid ? d0.x : 0.f; d0.y = valid ? d0.y : 0.f; d1.x = valid ? d1.x : 0.f; d1.y = valid ? d1.y : 0.f; }
                                const f32x2 s0 = (f32x2){ssum[t][0], ssum[t][1]} + d0, s1 = (f32x2){ssum[t][2], ssum[t][3]} + d1;
                                const f32x2 q0 = d0 * d0 + (f32x2){ssq[t][0], ssq[t][1]}, q1 = d1 * d1 + (f32x2){ssq[t][2], ssq[t][3]};
                                ssum[t][0] = s0.x; ssum[t][1] = s0.y; ssum[t][2] = s1.x; ssum[t][3] = s1.y;
                                ssq[t][0] = q0.x; ssq[t][1] = q0.y; ssq[t][2] = q1.x; ssq[t][3] = q1.y;
                            }
                            if (CHAP_ABLATE & 8) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }
                            const unsigned oi = (unsigned)(row * orow + zw_off + ooff[t]);
                            if (valid) {
                                if (P.out_f32) st4((float*)(ob + oi * 4u), v); else st4((T*)(obt + oi * (unsigned)sizeof(T)), v);
                            }
                        }
                    }
                };
                if (full) emit(std::true_type{}); else emit(std::false_type{});
            }
        }
        CHAP_STAMP(3);
        // ---- land the prefetched halo in the other buffer; one barrier per item
        // (leaving BEFORE the commit keeps "prefetch issued but never waited for" off the loop's back edge: with
        //  `if (has_next) commit` the compiler must assume the prefetch registers still have loads in flight at the
        //  top of the next item and waits for every outstanding store before it issues the new loads)
        if (!has_next) break;
        halo_commit<T, KC, ADD2, UNITS, !D3, ONE>(R, U, nxt, s0, s1, aff, plain, nn, nchunk, lanesel);
        // Staged weights have ONE LDS buffer in bf16 (round 4; rounds 1-3 and fp32: two, like the halo): every wave has to be through with it before it is overwritten -- a
        // second barrier per item -- but a block then needs 79 instead of 108 KB of LDS (3D bricks, 32 output channels), and together with the 256-register
        // bound (conv_min_waves) TWO blocks are resident per CU where one was: the 3D 64->64 layer at 28x28x20 (280 blocks on 256 CUs: two rounds of blocks at
        // one per CU) 32.0 -> 25.6 us when its weights are staged, the 3D step 14.52 -> 14.09 ms, the 2D step 6.09 -> 6.03 (profiles/r04_conv_wst1m2_steps.log;
        // either change alone: 14.6 / 14.47 ms)
        if (conv_one_wbuf<T>()) {
            if (WST) __syncthreads();
            wstage_commit(0);
        } else {
            wstage_commit((int)((it + 1) & 1));
        }
        chunk = nchunk == nchunks ? 0 : nchunk;
        CHAP_STAMP(4);
        __syncthreads();
        CHAP_STAMP(5);
    }

    CHAP_STAMP_P(4);
    // ---- BatchNorm statistics: registers -> 16-lane DPP reduce -> LDS row of this wave -> the four rows summed in a fixed
    // order -> this block's partial slot (plain stores: no atomics anywhere, chap_bn_finalize sums the slots in a fixed
    // order, so the statistics are bitwise reproducible for a given launch geometry)
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
        float* st = P.stats + CHAP_STATS_HDR + (long)blockIdx.x * 2 * P.Cout;      // slot rows are indexed by the LOGICAL channel (a transposed conv's sub-lattices are folded by the finalize)
        for (int i = threadIdx.x; i < 2 * 16 * NT; i += 256) {
            const int which = i / (16 * NT), k = i % (16 * NT);
            const int nl = nt0 * 16 + k;
            const float v = (bstat[(0 * 2 + which) * 16 * NT + k] + bstat[(1 * 2 + which) * 16 * NT + k]) + (bstat[(2 * 2 + which) * 16 * NT + k] + bstat[(3 * 2 + which) * 16 * NT + k]);
            if (nl < P.Cout) st[which * P.Cout + nl] = v;
        }
    }
    CHAP_STAMP_P(5);
}
