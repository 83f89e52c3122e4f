// Implicit-GEMM convolution for gfx950 (CDNA4, wave64): LDS-staged halo tile ("im2col in LDS"),
// MFMA 16x16x32 bf16 / 16x16x4 f32, fp32 accumulate, fused bias + BatchNorm-statistics epilogue.
//
// GEMM view:  D[cout][pixel] = sum_k  W[cout][k] * X[k][pixel],   k = (tap, channel)
//   "A" operand = packed weights (row = cout), read straight from global/L2 in fragment order,
//   "B" operand = pixels (col = pixel), read from the LDS halo tile.
// With that role assignment a lane ends up holding 4 consecutive output channels of ONE pixel, so the
// channel-last store is a vector store and the per-channel statistics are a 16-lane shuffle reduce.
//
// Block = 256 threads = 4 waves; output tile = TH x 16 pixels of one z-plane (TH = 4*MR, wave w owns
// rows [w*MR, w*MR+MR)) x 16*NT output channels.  K is walked in chunks of KC input channels; per
// chunk the (TH-1)*s+k by 15*s+k (by k planes in 3D) input halo is staged ONCE through registers
// (the lazy-activation transform is applied there) and then serves all k^2 / k^3 taps from LDS.
#pragma once
#include "common.h"

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

template <typename T, int KS, int ST, bool D3, int KC, int MR>
__host__ __device__ constexpr size_t conv_lds_bytes(int NT) {
    return (size_t)conv_geom<KS, ST, D3, MR>::HP * pix_stride<T, KC>() * sizeof(T) + 2 * 16 * NT * sizeof(float);
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
};
template <> struct frag<float> {
    typedef f32x8 type;
    static __device__ __forceinline__ f32x8 load(const float* p) { f32x8 r; r.lo = *(const float4*)p; r.hi = *(const float4*)(p + 4); return r; }
    static __device__ __forceinline__ f32x8 zero() { f32x8 r; r.lo = make_float4(0, 0, 0, 0); r.hi = r.lo; return r; }
};

// Stage one halo tile (chunk of KC channels) into LDS with the lazy-activation transform applied.
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

template <typename T, int KS, int ST, bool D3, int KC, int NT, int MR>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const chap_conv_params P) {
    typedef conv_geom<KS, ST, D3, MR> G;
    typedef typename frag<T>::type F;
    constexpr int GPT = KC / 8, PS = pix_stride<T, KC>();
    constexpr int NP = G::NTAPS * GPT, STEPS = (NP + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo = (T*)smem;
    float* bstat = (float*)(smem + (size_t)G::HP * PS * sizeof(T));

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const int tiles_x = (P.W + G::TW - 1) / G::TW, tiles_y = (P.H + G::TH - 1) / G::TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int z0 = b % P.D;     const int n = b / P.D;
    const int x0 = tx * G::TW, y0 = ty * G::TH;
    const int ntiles_total = (P.Cout + 15) >> 4;
    const int nt0 = blockIdx.y * NT;
    const int Ctot = P.combine == 0 ? (P.src[0].C + (P.nsrc > 1 ? P.src[1].C : 0)) : P.src[0].C;
    const int nchunks = Ctot / KC;

    f32x4 acc[MR][NT];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const T* wbase = (const T*)P.wpacked + ((long)nt0 * 64 + lane) * 8;
    const long wstep = (long)ntiles_total * 64 * 8;      // elements per (chunk, step)

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk) __syncthreads();
        stage_halo<T, KS, ST, D3, KC, MR>(halo, P.src[0], P.src[1], P.nsrc, P.combine, n, z0, y0, x0, P.ID, P.IH, P.IW, chunk);
        __syncthreads();
        const T* wc = wbase + (long)chunk * STEPS * wstep;
#pragma unroll 3
        for (int step = 0; step < STEPS; ++step) {
            F wf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wf[t] = (nt0 + t < ntiles_total) ? frag<T>::load(wc + (long)step * wstep + t * 512) : frag<T>::zero();
            const int p = step * 4 + g;
            const int tap = p / GPT, cgl = p % GPT;
            const int dx = tap % KS, dy = (tap / KS) % KS, dz = tap / (KS * KS);
            const bool pv = p < NP;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const int row = wave * MR + m;
                const int off = ((dz * G::HH + row * ST + dy) * G::HW + px * ST + dx) * PS + cgl * 8;
                F xf = pv ? frag<T>::load(halo + off) : frag<T>::zero();
#pragma unroll
                for (int t = 0; t < NT; ++t) mma8(acc[m][t], wf[t], xf);
            }
        }
    }

    // ---- epilogue: lane holds D[cout = 16*t + 4*g + j][pixel (row m, x = px)] ----
    const bool do_stats = P.stats != nullptr;
    if (do_stats) {
        for (int i = threadIdx.x; i < 2 * 16 * NT; i += 256) bstat[i] = 0.f;
        __syncthreads();
    }
    const int gx = x0 + px;
    const int SD2 = (P.dims == 3) ? 2 : 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nl = (nt0 + t) * 16 + 4 * g;            // logical output channel of j = 0
        if (nl >= P.Cout) continue;                        // whole 16-tile padding (or j-range beyond Cout below)
        float bj[4] = {0.f, 0.f, 0.f, 0.f};
        const int cb = P.out_mode == 1 ? (nl % P.out_Cn) : nl;
        if (P.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (nl + j < P.Cout) bj[j] = P.bias[cb + j];
        }
        float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int gy = y0 + wave * MR + m;
            const bool valid = gy < P.H && gx < P.W;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[m][t][j] + bj[j];
                if (valid) { s[j] += v[j]; q[j] += v[j] * v[j]; }
            }
            if (!valid) continue;
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
        if (do_stats) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s[j] += __shfl_xor(s[j], o, 64); q[j] += __shfl_xor(q[j], o, 64); }
            }
            if (px == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    atomicAdd(&bstat[t * 16 + 4 * g + j], s[j]);
                    atomicAdd(&bstat[16 * NT + t * 16 + 4 * g + j], q[j]);
                }
            }
        }
    }
    if (do_stats) {
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
