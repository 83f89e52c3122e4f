#include <cstdlib>
// Element-wise / reduction kernels around the convolutions: first-layer (Cin = 1) direct conv,
// BatchNorm finalize, lazy-activation max-pool, align_corners 2x upsample (+ adjoint), the
// BN/activation backward pair, layout converters.  All HBM-bound: 16-byte vector accesses along
// the channel axis, wave64 shuffle reductions, one float atomic per block and channel.
#include "common.h"
#include "launch.h"
#include "actbwd_math.h"
#include "conv_c1_mfma.h"

// Index decoding uses 32-bit unsigned arithmetic (64-bit integer division costs ~100 VALU instructions on
// gfx950 and turned these streaming kernels VALU-bound); entry points reject tensors with >= 2^32 elements.
typedef unsigned int u32;

// Fixed-order wave reduction in fp64 (xor butterfly: the same pairing on every run).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// =========================================================================================
// First conv, Cin = 1, k3 s1 "same" (unet.py:50 with in_chns=1; vnet.py:19 with n_channels=1).
// One thread per output pixel computes all Cout (<= 32) channels; weights + bias live in LDS.
template <typename T, bool D3, int CO>
__device__ __forceinline__ void conv_c1_fwd_kernel(const chap_conv_c1_params& P) {
    constexpr int KD = D3 ? 3 : 1, TAPS = KD * 9;
    __shared__ float ws[CO * TAPS + 2 * CO];
    __shared__ float bstat[4][2 * CO];
    for (int i = threadIdx.x; i < CO * TAPS; i += 256) ws[i] = P.w[i];
    for (int i = threadIdx.x; i < CO; i += 256) { ws[CO * TAPS + i] = P.bias ? P.bias[i] : 0.f; ws[CO * TAPS + CO + i] = (P.stats && P.stats_shift) ? P.stats_shift[i] : 0.f; }
    __syncthreads();
    const long npix = (long)P.N * P.D * P.H * P.W;
    float ssum[CO], ssq[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long)gridDim.x * 256) {
        const u32 up = (u32)pix;
        const int x = (int)(up % (u32)P.W); u32 r = up / (u32)P.W;
        const int y = (int)(r % (u32)P.H); r /= (u32)P.H;
        const int z = (int)(r % (u32)P.D); const int n = (int)(r / (u32)P.D);
        float in[TAPS];
#pragma unroll
        for (int dz = 0; dz < KD; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int zz = z + dz - (D3 ? 1 : 0), yy = y + dy - 1, xx = x + dx - 1;
                    const bool ib = (unsigned)zz < (unsigned)P.D && (unsigned)yy < (unsigned)P.H && (unsigned)xx < (unsigned)P.W;
                    in[(dz * 3 + dy) * 3 + dx] = ib ? P.x[(((long)n * P.D + zz) * P.H + yy) * P.W + xx] : 0.f;
                }
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float a = ws[CO * TAPS + c];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) a = fmaf(in[t], ws[c * TAPS + t], a);
            acc[c] = a;
            const float as = a - ws[CO * TAPS + CO + c];        // shifted moments (chap_hip.h, "stats")
            ssum[c] += as; ssq[c] += as * as;
        }
        T* o = (T*)P.out + pix * CO;
#pragma unroll
        for (int c = 0; c < CO; c += 8) st8(o + c, acc + c);
    }
    if (P.stats) {          // this block's partial slot: fixed-order wave butterfly, the four waves summed in a fixed order
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float s = wave_sum(ssum[c]), q = wave_sum(ssq[c]);
            if ((threadIdx.x & 63) == 0) { bstat[threadIdx.x >> 6][c] = s; bstat[threadIdx.x >> 6][CO + c] = q; }
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) *(int*)P.stats = (int)gridDim.x;
        for (int i = threadIdx.x; i < 2 * CO; i += 256)
            P.stats[CHAP_STATS_HDR + (long)blockIdx.x * 2 * CO + i] = (bstat[0][i] + bstat[1][i]) + (bstat[2][i] + bstat[3][i]);
    }
}

extern "C" int chap_conv_c1_fwd(const chap_conv_c1_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->x && p->w && p->out, "chap_conv_c1_fwd: null argument");
    CHAP_CHECK_ARG(p->Cout == 16, "chap_conv_c1_fwd: Cout=%d (only 16 built)", p->Cout);
    CHAP_CHECK_ARG(p->dims == 2 || p->dims == 3, "chap_conv_c1_fwd: dims=%d", p->dims);
    const long npix = (long)p->N * p->D * p->H * p->W;
    dim3 grid((unsigned)(cdiv(npix, 256) < CHAP_STATS_MAX_SLOTS ? cdiv(npix, 256) : CHAP_STATS_MAX_SLOTS));   // one statistics slot per block
    hipStream_t s = (hipStream_t)stream;
    const bool d3 = p->dims == 3, bf = p->dtype == CHAP_BF16;
    if (bf) {
        // bf16: the taps as the K dimension of one MFMA per 16 pixels (conv_c1_mfma.h).  CHAP_C1_MFMA=0 (lab knob): the scalar kernel below.
        static int mf = -1;
        if (mf < 0) { const char* e = getenv("CHAP_C1_MFMA"); mf = (e && atoi(e) == 0) ? 0 : 1; }
        if (mf) {
            CHAP_CHECK_ARG(npix < (1l << 31), "chap_conv_c1_fwd: %ld pixels", npix);
            auto waste = [&](int txt) { const int tx = 16 * txt; return cdiv(p->W, tx) * tx - p->W; };
            const int txt = waste(5) < waste(4) ? 5 : 4;         // 16-pixel tiles per block row: W = 80 (LA patches) takes 5, powers of two 4
            const int ty = d3 ? 4 : 8, tz = d3 ? 2 : 1;
            const long ntiles = (long)p->N * cdiv(p->D, tz) * cdiv(p->H, ty) * cdiv(p->W, 16 * txt);
            dim3 g((unsigned)(ntiles < CHAP_STATS_MAX_SLOTS ? ntiles : CHAP_STATS_MAX_SLOTS));      // persistent blocks, one statistics slot each
            if (d3 && txt == 5) return chap_launch<chap_conv_c1_params, conv_c1_mfma_kernel<true, 5>, 256>(g, dim3(256), 0, s, *p, "chap_conv_c1_fwd(mfma)");
            if (d3) return chap_launch<chap_conv_c1_params, conv_c1_mfma_kernel<true, 4>, 256>(g, dim3(256), 0, s, *p, "chap_conv_c1_fwd(mfma)");
            if (txt == 5) return chap_launch<chap_conv_c1_params, conv_c1_mfma_kernel<false, 5>, 256>(g, dim3(256), 0, s, *p, "chap_conv_c1_fwd(mfma)");
            return chap_launch<chap_conv_c1_params, conv_c1_mfma_kernel<false, 4>, 256>(g, dim3(256), 0, s, *p, "chap_conv_c1_fwd(mfma)");
        }
    }
    if (bf && d3) return chap_launch<chap_conv_c1_params, conv_c1_fwd_kernel<bf16_t, true, 16>, 256>(grid, dim3(256), 0, s, *p, "chap_conv_c1_fwd");
    if (bf) return chap_launch<chap_conv_c1_params, conv_c1_fwd_kernel<bf16_t, false, 16>, 256>(grid, dim3(256), 0, s, *p, "chap_conv_c1_fwd");
    if (d3) return chap_launch<chap_conv_c1_params, conv_c1_fwd_kernel<float, true, 16>, 256>(grid, dim3(256), 0, s, *p, "chap_conv_c1_fwd");
    return chap_launch<chap_conv_c1_params, conv_c1_fwd_kernel<float, false, 16>, 256>(grid, dim3(256), 0, s, *p, "chap_conv_c1_fwd");
}

// Backward of the first conv: dx (VAT needs dL/dx), dw, db.
//   dx[p]      = sum_{tap,c} g[p - tap + pad][c] * w[c][tap]
//   dw[c][tap] = sum_p x[p + tap - pad] * g[p][c]      (block partials -> ws, then one reduce pass)
struct c1_bwd_args { chap_conv_c1_bwd_params P; int nblocks; };
template <typename T, bool D3, int CO>
__device__ __forceinline__ void conv_c1_bwd_kernel(const c1_bwd_args& A) {
    const chap_conv_c1_bwd_params& P = A.P;
    const int nblocks = A.nblocks;
    constexpr int KD = D3 ? 3 : 1, TAPS = KD * 9;
    __shared__ float ws[CO * TAPS];
    __shared__ float part[4][CO * TAPS + CO];
    for (int i = threadIdx.x; i < CO * TAPS; i += 256) ws[i] = P.w[i];
    __syncthreads();
    const long npix = (long)P.N * P.D * P.H * P.W;
    const T* g = (const T*)P.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- dx ----
    if (P.dx) {
        for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long)nblocks * 256) {
            const u32 up = (u32)pix;
            const int x = (int)(up % (u32)P.W); u32 r = up / (u32)P.W;
            const int y = (int)(r % (u32)P.H); r /= (u32)P.H;
            const int z = (int)(r % (u32)P.D); const int n = (int)(r / (u32)P.D);
            float a = 0.f;
#pragma unroll
            for (int dz = 0; dz < KD; ++dz)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int zz = z - dz + (D3 ? 1 : 0), yy = y - dy + 1, xx = x - dx + 1;
                        if ((unsigned)zz < (unsigned)P.D && (unsigned)yy < (unsigned)P.H && (unsigned)xx < (unsigned)P.W) {
                            const long q = (((long)n * P.D + zz) * P.H + yy) * P.W + xx;
                            const int tap = (dz * 3 + dy) * 3 + dx;
#pragma unroll
                            for (int c = 0; c < CO; c += 8) {
                                float v[8];
                                ld8(g + q * CO + c, v);
#pragma unroll
                                for (int j = 0; j < 8; ++j) a = fmaf(v[j], ws[(c + j) * TAPS + tap], a);
                            }
                        }
                    }
            P.dx[pix] = a;
        }
    }
    // ---- dw / db partials: thread t handles channel c = t % CO and pixel-lane q = t / CO ----
    if (P.dw || P.db) {
        constexpr int PL = 256 / CO;          // pixels processed in parallel per block step
        const int c = threadIdx.x % CO, q = threadIdx.x / CO;
        float a[TAPS], ab = 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) a[t] = 0.f;
        for (long pix = (long)blockIdx.x * PL + q; pix < npix; pix += (long)nblocks * PL) {
            const u32 up = (u32)pix;
            const int x = (int)(up % (u32)P.W); u32 r = up / (u32)P.W;
            const int y = (int)(r % (u32)P.H); r /= (u32)P.H;
            const int z = (int)(r % (u32)P.D); const int n = (int)(r / (u32)P.D);
            const float gv = elem<T>::get(g[pix * CO + c]);
            ab += gv;
#pragma unroll
            for (int dz = 0; dz < KD; ++dz)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int zz = z + dz - (D3 ? 1 : 0), yy = y + dy - 1, xx = x + dx - 1;
                        const bool ib = (unsigned)zz < (unsigned)P.D && (unsigned)yy < (unsigned)P.H && (unsigned)xx < (unsigned)P.W;
                        const float xv = ib ? P.x[(((long)n * P.D + zz) * P.H + yy) * P.W + xx] : 0.f;
                        a[(dz * 3 + dy) * 3 + dx] = fmaf(xv, gv, a[(dz * 3 + dy) * 3 + dx]);
                    }
        }
        // reduce over q (threads with equal c): lanes c, c+CO, c+2CO, ... inside a wave, then across waves via LDS
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            float v = a[t];
            for (int o = CO; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane < CO) part[wave][c * TAPS + t] = v;
        }
        {
            float v = ab;
            for (int o = CO; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane < CO) part[wave][CO * TAPS + c] = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CO * TAPS + CO; i += 256)
            P.ws[(long)blockIdx.x * (CO * TAPS + CO) + i] = part[0][i] + part[1][i] + part[2][i] + part[3][i];
    }
}

struct c1_reduce_args { const float* ws; int nblocks, taps; float* dw; float* db; };
template <int CO>
__device__ __forceinline__ void conv_c1_reduce_kernel(const c1_reduce_args& A) {
    const float* ws = A.ws; const int nblocks = A.nblocks, taps = A.taps; float* dw = A.dw; float* db = A.db;
    __shared__ float red[4];
    const int i = blockIdx.x;                    // one block per output element, threads stride the block partials
    const int tot = CO * taps + CO;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += ws[(long)b * tot + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0] + red[1]) + (red[2] + red[3]);
        if (i < CO * taps) { if (dw) dw[i] += s; } else if (db) db[i - CO * taps] += s;
    }
}

static int c1_bwd_blocks(const chap_conv_c1_bwd_params* p) {
    const long npix = (long)p->N * p->D * p->H * p->W;
    long b = (npix + 255) / 256;
    return (int)(b < 1024 ? b : 1024);
}
extern "C" size_t chap_conv_c1_bwd_ws(const chap_conv_c1_bwd_params* p) {
    const int taps = p->dims == 3 ? 27 : 9;
    return (size_t)c1_bwd_blocks(p) * (p->Cout * taps + p->Cout) * sizeof(float);
}
extern "C" int chap_conv_c1_bwd(const chap_conv_c1_bwd_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->g && p->w, "chap_conv_c1_bwd: null argument");
    CHAP_CHECK_ARG(p->Cout == 16, "chap_conv_c1_bwd: Cout=%d (only 16 built)", p->Cout);
    CHAP_CHECK_ARG(!(p->dw || p->db) || (p->ws && p->x), "chap_conv_c1_bwd: dw/db need x and ws");
    const int nb = c1_bwd_blocks(p);
    hipStream_t s = (hipStream_t)stream;
    const bool d3 = p->dims == 3, bf = p->dtype == CHAP_BF16;
    const c1_bwd_args a = {*p, nb};
    int r;
    if (bf && d3) r = chap_launch<c1_bwd_args, conv_c1_bwd_kernel<bf16_t, true, 16>, 256>(dim3(nb), dim3(256), 0, s, a, "chap_conv_c1_bwd");
    else if (bf) r = chap_launch<c1_bwd_args, conv_c1_bwd_kernel<bf16_t, false, 16>, 256>(dim3(nb), dim3(256), 0, s, a, "chap_conv_c1_bwd");
    else if (d3) r = chap_launch<c1_bwd_args, conv_c1_bwd_kernel<float, true, 16>, 256>(dim3(nb), dim3(256), 0, s, a, "chap_conv_c1_bwd");
    else r = chap_launch<c1_bwd_args, conv_c1_bwd_kernel<float, false, 16>, 256>(dim3(nb), dim3(256), 0, s, a, "chap_conv_c1_bwd");
    if (r) return r;
    if (p->dw || p->db) {
        const int taps = d3 ? 27 : 9, tot = 16 * taps + 16;
        const c1_reduce_args ra = {(const float*)p->ws, nb, taps, p->dw, p->db};
        return chap_launch<c1_reduce_args, conv_c1_reduce_kernel<16>, 256>(dim3(tot), dim3(256), 0, s, ra, "chap_conv_c1_bwd(reduce)");
    }
    return CHAP_OK;
}

// =========================================================================================
// BatchNorm finalize: batch statistics -> (scale, shift) of the lazy activation; running stats
// follow F.batch_norm(training=True): running = (1-m)*running + m*batch, unbiased variance.
// One wave per channel: lane l sums slots l, l+64, ... (and the sub-lattice rows of a transposed conv) in fp64, then a
// fixed xor butterfly -- the same order on every run.  mean = c + S/n, var = Q/n - (S/n)^2 with the moments taken about
// the conv's shift c (no cancellation once c tracks the mean), evaluated in fp64.
__device__ __forceinline__ void bn_finalize_kernel(const chap_bn_finalize_params& P) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= P.C) return;
    // slots in use: the header word the producing conv wrote (a host-side note of the conv's grid was tried in round 3 to save this
    // load: no measurable difference, and a hidden coupling between two calls -- removed)
    const int nslots = *(const int*)P.stats;
    // the per-channel parameters are requested NOW, next to the header and the slots: behind the reduction they were a third
    // dependent memory round trip of a kernel that is nothing but latency
    const bool upd = P.momentum > 0.f && P.running_mean;
    const float gam = P.gamma[c], bet = P.beta[c], sft = P.stats_shift ? P.stats_shift[c] : 0.f;
    const float rm0 = upd ? P.running_mean[c] : 0.f, rv0 = upd ? P.running_var[c] : 0.f;
    const int nsub = P.Clog / P.C;
    const float* st = P.stats + CHAP_STATS_HDR;
    double s = 0.0, q = 0.0;
    // a lane's slots are lane, lane + 64, ...: at most CHAP_STATS_MAX_SLOTS / 64 = 16 of them.  All loads of a batch are issued
    // before the first add (the kernel is pure latency: one dependent load per iteration made it 6.7 us), the adds keep the
    // fixed order b = lane, lane + 64, ...
    constexpr int NB = CHAP_STATS_MAX_SLOTS / 64;
    for (int k = 0; k < nsub; ++k) {
        float vs[NB], vq[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int b = lane + 64 * i;
            const float* row = st + (long)(b < nslots ? b : 0) * 2 * P.Clog + k * P.C + c;
            vs[i] = row[0]; vq[i] = row[P.Clog];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const bool ok = lane + 64 * i < nslots;
            s += ok ? (double)vs[i] : 0.0; q += ok ? (double)vq[i] : 0.0;
        }
    }
    s = wave_sum_f64(s); q = wave_sum_f64(q);
    if (lane != 0) return;
    if (c == 0 && P.num_batches_tracked && P.momentum > 0.f) *P.num_batches_tracked += 1;
    const double cnt = (double)P.count;
    const double ms = s / cnt;                                   // mean of (x - shift)
    double var = q / cnt - ms * ms;
    var = var > 0.0 ? var : 0.0;
    const float mean = (float)((double)sft + ms);
    const float invstd = (float)(1.0 / sqrt(var + (double)P.eps));
    const float sc = gam * invstd;
    P.scale[c] = sc;
    P.shift[c] = bet - mean * sc;
    if (P.mean) { P.mean[c] = mean; P.invstd[c] = invstd; }
    if (upd) {
        const float unb = (float)(cnt > 1.0 ? var * cnt / (cnt - 1.0) : var);
        P.running_mean[c] = (1.f - P.momentum) * rm0 + P.momentum * mean;
        P.running_var[c] = (1.f - P.momentum) * rv0 + P.momentum * unb;
    }
}
extern "C" int chap_bn_finalize(const chap_bn_finalize_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->stats && p->gamma && p->beta && p->scale && p->shift && p->C > 0 && p->count > 0, "chap_bn_finalize: bad argument");
    CHAP_CHECK_ARG(p->Clog >= p->C && p->Clog % p->C == 0, "chap_bn_finalize: Clog=%d must be a multiple of C=%d", p->Clog, p->C);
#ifdef CHAP_LAB      // lab builds only (tools/lab, loaded through CHAP_LIBPATH): timing bound, wrong numerics
    { static int skip = -1; if (skip < 0) skip = getenv("CHAP_LAB_SKIP_BNFIN") ? 1 : 0; if (skip) return CHAP_OK; }
#endif
    return chap_launch<chap_bn_finalize_params, bn_finalize_kernel, 256>(dim3(cdiv(p->C, 4)), dim3(256), 0, (hipStream_t)stream, *p, "chap_bn_finalize");
}

__device__ __forceinline__ void bn_eval_kernel(const chap_bn_eval_params& P) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.C) return;
    const float sc = P.gamma[c] * rsqrtf(P.running_var[c] + P.eps);
    P.scale[c] = sc;
    P.shift[c] = P.beta[c] - P.running_mean[c] * sc;
}
extern "C" int chap_bn_eval_affine(const chap_bn_eval_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->gamma && p->beta && p->running_mean && p->running_var && p->scale && p->shift && p->C > 0, "chap_bn_eval_affine: bad argument");
    return chap_launch<chap_bn_eval_params, bn_eval_kernel, 64>(dim3(cdiv(p->C, 64)), dim3(64), 0, (hipStream_t)stream, *p, "chap_bn_eval_affine");
}

// =========================================================================================
// 2x2 max-pool of a lazy activation. One thread per (pooled pixel, 8 channels).
template <typename T>
__device__ __forceinline__ void act_pool2_kernel(const chap_pool_params& P) {
    const int D = P.D > 1 ? P.D : 1, KZ = P.D > 1 ? 2 : 1;
    const int C8 = P.r.C / 8, OD = D / KZ, OH = P.H / 2, OW = P.W / 2;
    const long total = (long)P.N * OD * OH * OW * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const u32 ui = (u32)i;
        const int c8 = (int)(ui % (u32)C8) * 8; u32 r = ui / (u32)C8;
        const int ox = (int)(r % (u32)OW); r /= (u32)OW;
        const int oy = (int)(r % (u32)OH); r /= (u32)OH;
        const int oz = (int)(r % (u32)OD); const int n = (int)(r / (u32)OD);
        float best[8]; uint32_t bi[8];
        for (int k = 0; k < 4 * KZ; ++k) {
            const long pix = (((long)n * D + KZ * oz + (k >> 2)) * P.H + 2 * oy + ((k >> 1) & 1)) * P.W + 2 * ox + (k & 1);
            float v[8];
            src_load8<T>(P.r, n, pix, c8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k == 0 || v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = k; }
        }
        const long op = (((long)n * OD + oz) * OH + oy) * OW + ox;
        st8((T*)P.out + op * P.r.C + c8, best);
        if (P.idx) {
            uint2 m;
            m.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
            m.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
            *(uint2*)(P.idx + op * P.r.C + c8) = m;
        }
    }
}
extern "C" int chap_act_pool2(const chap_pool_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->r.ptr && p->out, "chap_act_pool2: null argument");
    CHAP_CHECK_ARG(p->r.C % 8 == 0 && p->H % 2 == 0 && p->W % 2 == 0 && (p->D <= 1 || p->D % 2 == 0), "chap_act_pool2: C%%8, even dims required");
#ifdef CHAP_LAB      // lab builds only: timing bound (wrong numerics) of fusing this launch into its consumer
    { static int skip = -1; if (skip < 0) skip = getenv("CHAP_LAB_SKIP_POOL") ? 1 : 0; if (skip) return CHAP_OK; }
#endif
    const long total = (long)p->N * (p->D > 1 ? p->D / 2 : 1) * (p->H / 2) * (p->W / 2) * (p->r.C / 8);
    const int blocks = chap_blocks(total, 4096);
    if (p->dtype == CHAP_BF16) return chap_launch<chap_pool_params, act_pool2_kernel<bf16_t>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_act_pool2");
    return chap_launch<chap_pool_params, act_pool2_kernel<float>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_act_pool2");
}

// =========================================================================================
// 2x upsample, align_corners=True: src = dst * (in-1)/(out-1)  (F.interpolate bilinear/trilinear).
__device__ __forceinline__ void ac_coord(int o, int in, int out, int& i0, int& i1, float& w1, bool half_pixel = false) {
    const float sc = half_pixel ? (float)in / (float)out : (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f);
    float f = half_pixel ? sc * ((float)o + 0.5f) - 0.5f : sc * (float)o;     // area_pixel_compute_source_index
    if (f < 0.f) f = 0.f;
    i0 = (int)f;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + 1 < in ? i0 + 1 : in - 1;
    w1 = f - (float)i0;
}

template <typename T>
__device__ __forceinline__ void upsample2x_kernel(const chap_upsample_params& P) {
    const int C8 = P.r.C / 8;
    const int OD = P.dims == 3 ? 2 * P.D : P.D, OH = 2 * P.H, OW = 2 * P.W;
    const long total = (long)P.N * OD * OH * OW * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const u32 ui = (u32)i;
        const int c8 = (int)(ui % (u32)C8) * 8; u32 r = ui / (u32)C8;
        const int ox = (int)(r % (u32)OW); r /= (u32)OW;
        const int oy = (int)(r % (u32)OH); r /= (u32)OH;
        const int oz = (int)(r % (u32)OD); const int n = (int)(r / (u32)OD);
        int x0, x1, y0, y1, z0, z1; float wx, wy, wz;
        const bool hp = P.half_pixel != 0;
        ac_coord(ox, P.W, OW, x0, x1, wx, hp);
        ac_coord(oy, P.H, OH, y0, y1, wy, hp);
        if (P.dims == 3) ac_coord(oz, P.D, OD, z0, z1, wz, hp); else { z0 = z1 = oz; wz = 0.f; }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        const int nz = P.dims == 3 ? 2 : 1;
        for (int kz = 0; kz < nz; ++kz)
            for (int ky = 0; ky < 2; ++ky)
                for (int kx = 0; kx < 2; ++kx) {
                    const float w = (kz ? wz : 1.f - wz) * (ky ? wy : 1.f - wy) * (kx ? wx : 1.f - wx);
                    const long pix = (((long)n * P.D + (kz ? z1 : z0)) * P.H + (ky ? y1 : y0)) * P.W + (kx ? x1 : x0);
                    float v[8];
                    src_load8<T>(P.r, n, pix, c8, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, v[j], acc[j]);
                }
        const long op = (((long)n * OD + oz) * OH + oy) * OW + ox;
        st8((T*)P.out + op * P.out_ld + P.out_coff + c8, acc);
    }
}
// Cell form (used whenever every interpolated axis has >= 2 input samples): one thread owns one input CELL
// (the interval [i, i+1] on each interpolated axis) and 8 channels, loads and transforms (lazy activation) its
// 4 / 8 corner samples ONCE and writes every output sample whose source coordinate falls into the cell (2 per
// axis on average for x2).  The per-output form above re-loads and re-transforms the corners for each output:
// 8x the loads and VALU work in 3D.  Same weights, same summation order as the per-output form.
__device__ __forceinline__ int up_axis_outputs(int i, int in, int out, bool hp, int o_[5], float w_[5]) {
    int n = 0;
    for (int o = max(0, 2 * i - 1); o <= min(out - 1, 2 * i + 3); ++o) {      // 2i+3: only the clamped last output of the last cell
        int i0, i1; float w1;
        ac_coord(o, in, out, i0, i1, w1, hp);
        if (i0 > in - 2) { i0 = in - 2; w1 = 1.f; }             // source clamped to the last sample: corner 1 with weight 1
        if (i0 == i) { o_[n] = o; w_[n] = w1; ++n; }
    }
    return n;
}
template <typename T, bool D3>
__device__ __forceinline__ void upsample2x_cell_kernel(const chap_upsample_params& P) {
    const int C8 = P.r.C / 8;
    const int OD = D3 ? 2 * P.D : P.D, OH = 2 * P.H, OW = 2 * P.W;
    const int CD = D3 ? P.D - 1 : P.D, CH = P.H - 1, CW = P.W - 1;       // cells per axis
    const long total = (long)P.N * CD * CH * CW * C8;
    const bool hp = P.half_pixel != 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const u32 ui = (u32)i;
        const int c8 = (int)(ui % (u32)C8) * 8; u32 r = ui / (u32)C8;
        const int cx = (int)(r % (u32)CW); r /= (u32)CW;
        const int cy = (int)(r % (u32)CH); r /= (u32)CH;
        const int cz = (int)(r % (u32)CD); const int n = (int)(r / (u32)CD);
        float v[D3 ? 2 : 1][2][2][8];
#pragma unroll
        for (int kz = 0; kz < (D3 ? 2 : 1); ++kz)
#pragma unroll
            for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                for (int kx = 0; kx < 2; ++kx)
                    src_load8<T>(P.r, n, (((long)n * P.D + cz + kz) * P.H + cy + ky) * P.W + cx + kx, c8, v[kz][ky][kx]);
        int oz_[5], oy_[5], ox_[5]; float wz_[5], wy_[5], wx_[5];
        int nz = 1; oz_[0] = cz; wz_[0] = 0.f;
        if (D3) nz = up_axis_outputs(cz, P.D, OD, hp, oz_, wz_);
        const int ny = up_axis_outputs(cy, P.H, OH, hp, oy_, wy_);
        const int nx = up_axis_outputs(cx, P.W, OW, hp, ox_, wx_);
        for (int a = 0; a < nz; ++a)
            for (int b = 0; b < ny; ++b)
                for (int c = 0; c < nx; ++c) {
                    const float wz = wz_[a], wy = wy_[b], wx = wx_[c];
                    float acc[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
                    for (int kz = 0; kz < (D3 ? 2 : 1); ++kz)
#pragma unroll
                        for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 2; ++kx) {
                                const float w = (kz ? wz : 1.f - wz) * (ky ? wy : 1.f - wy) * (kx ? wx : 1.f - wx);
#pragma unroll
                                for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, v[kz][ky][kx][j], acc[j]);
                            }
                    const long op = (((long)n * OD + oz_[a]) * OH + oy_[b]) * OW + ox_[c];
                    st8((T*)P.out + op * P.out_ld + P.out_coff + c8, acc);
                }
    }
}
extern "C" int chap_upsample2x(const chap_upsample_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->r.ptr && p->out && p->r.C % 8 == 0, "chap_upsample2x: bad argument");
    CHAP_CHECK_ARG(p->out_ld % 8 == 0 && p->out_coff % 8 == 0, "chap_upsample2x: out_ld/out_coff must be multiples of 8");
#ifdef CHAP_LAB      // lab builds only: timing bound (wrong numerics) of fusing this launch into its consumer
    { static int skip = -1; if (skip < 0) skip = getenv("CHAP_LAB_SKIP_UPSAMPLE") ? 1 : 0; if (skip) return CHAP_OK; }
#endif
    const bool d3 = p->dims == 3;
    if (p->H >= 2 && p->W >= 2 && (!d3 || p->D >= 2)) {
        const long cells = (long)p->N * (d3 ? p->D - 1 : p->D) * (p->H - 1) * (p->W - 1) * (p->r.C / 8);
        const int blocks = chap_blocks(cells, 16384);
        hipStream_t s = (hipStream_t)stream;
        if (p->dtype == CHAP_BF16) return d3 ? chap_launch<chap_upsample_params, upsample2x_cell_kernel<bf16_t, true>, 256>(dim3(blocks), dim3(256), 0, s, *p, "chap_upsample2x")
                                             : chap_launch<chap_upsample_params, upsample2x_cell_kernel<bf16_t, false>, 256>(dim3(blocks), dim3(256), 0, s, *p, "chap_upsample2x");
        return d3 ? chap_launch<chap_upsample_params, upsample2x_cell_kernel<float, true>, 256>(dim3(blocks), dim3(256), 0, s, *p, "chap_upsample2x")
                  : chap_launch<chap_upsample_params, upsample2x_cell_kernel<float, false>, 256>(dim3(blocks), dim3(256), 0, s, *p, "chap_upsample2x");
    }
    const long total = (long)p->N * (p->dims == 3 ? 2 * p->D : p->D) * 2 * p->H * 2 * p->W * (p->r.C / 8);
    const int blocks = chap_blocks(total, 8192);
    if (p->dtype == CHAP_BF16) return chap_launch<chap_upsample_params, upsample2x_kernel<bf16_t>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_upsample2x");
    return chap_launch<chap_upsample_params, upsample2x_kernel<float>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_upsample2x");
}

// Adjoint: each coarse pixel gathers from the fine pixels whose stencil touches it (no atomics).
// Along one axis, fine index o touches coarse i iff i0(o) == i or i1(o) == i; with scale
// (in-1)/(out-1) < 1/2 only o in [2i-2, 2i+2] can qualify, so a 5-wide window per axis suffices.
// weight with which fine sample o feeds coarse sample i along one axis (0 when its stencil does not touch i)
__device__ __forceinline__ int up_axis_adjoint(int i, int in, int out, int o_[5], float w_[5]) {
    int n = 0;
    for (int o = max(0, 2 * i - 2); o <= min(out - 1, 2 * i + 2); ++o) {
        int a0, a1; float w1;
        ac_coord(o, in, out, a0, a1, w1);
        const float w = (a0 == i ? 1.f - w1 : 0.f) + (a1 == i ? w1 : 0.f);
        if (w != 0.f) { o_[n] = o; w_[n] = w; ++n; }
    }
    return n;
}
template <typename T>
__device__ __forceinline__ void upsample2x_bwd_kernel(const chap_upsample_bwd_params& P) {
    const int C8 = P.C / 8;
    const int OD = P.dims == 3 ? 2 * P.D : P.D, OH = 2 * P.H, OW = 2 * P.W;
    const long total = (long)P.N * P.D * P.H * P.W * C8;
    const T* g = (const T*)P.g;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const u32 ui = (u32)i;
        const int c8 = (int)(ui % (u32)C8) * 8; u32 r = ui / (u32)C8;
        const int x = (int)(r % (u32)P.W); r /= (u32)P.W;
        const int y = (int)(r % (u32)P.H); r /= (u32)P.H;
        const int z = (int)(r % (u32)P.D); const int n = (int)(r / (u32)P.D);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        // per-axis contributor lists first (the nested form evaluated the source coordinate 5*5*5 times per thread)
        int oz_[5], oy_[5]; float wz_[5], wy_[5];
        int nz = 1; oz_[0] = z; wz_[0] = 1.f;
        if (P.dims == 3) nz = up_axis_adjoint(z, P.D, OD, oz_, wz_);
        const int ny = up_axis_adjoint(y, P.H, OH, oy_, wy_);
        // x: the FIXED window 2x-2 .. 2x+2, clamped, weight 0 where a sample does not touch x (or lies outside): the five loads of a
        // (z, y) row are issued together instead of one dependent load per trip of a runtime-bounded loop (the kernel was load-latency
        // bound: ~70 serial 16-byte loads per thread).  Contributions are added in the same ascending order as before, zero-weight
        // ones skipped: bit-identical results.
        int ox_[5]; float wx_[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int o = 2 * x - 2 + k, oc = min(max(o, 0), OW - 1);
            int a0, a1; float w1;
            ac_coord(oc, P.W, OW, a0, a1, w1);
            const float w = (a0 == x ? 1.f - w1 : 0.f) + (a1 == x ? w1 : 0.f);
            ox_[k] = oc; wx_[k] = (o == oc) ? w : 0.f;
        }
        for (int a = 0; a < nz; ++a)
            for (int b = 0; b < ny; ++b) {
                const long row = (((long)n * OD + oz_[a]) * OH + oy_[b]) * OW;
                const float wzy = wz_[a] * wy_[b];
                float v[5][8];
#pragma unroll
                for (int c = 0; c < 5; ++c) ld8(g + (row + ox_[c]) * P.g_ld + P.g_coff + c8, v[c]);
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const float w = wzy * wx_[c];
                    if (wx_[c] != 0.f) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, v[c][j], acc[j]);
                    }
                }
            }
        const long cp = (((long)n * P.D + z) * P.H + y) * P.W + x;
        st8((T*)P.out + cp * P.C + c8, acc);
    }
}
extern "C" int chap_upsample2x_bwd(const chap_upsample_bwd_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->g && p->out && p->C % 8 == 0 && p->g_ld % 8 == 0 && p->g_coff % 8 == 0, "chap_upsample2x_bwd: bad argument");
#ifdef CHAP_LAB      // lab builds only: timing bound (wrong numerics) of fusing this launch into its consumer
    { static int skip = -1; if (skip < 0) skip = getenv("CHAP_LAB_SKIP_UPSAMPLE_BWD") ? 1 : 0; if (skip) return CHAP_OK; }
#endif
    const long total = (long)p->N * p->D * p->H * p->W * (p->C / 8);
    const int blocks = chap_blocks(total, 8192);
    if (p->dtype == CHAP_BF16) return chap_launch<chap_upsample_bwd_params, upsample2x_bwd_kernel<bf16_t>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_upsample2x_bwd");
    return chap_launch<chap_upsample_bwd_params, upsample2x_bwd_kernel<float>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_upsample2x_bwd");
}

// =========================================================================================
// Backward through the lazy activation (+ pooled consumer) and training-mode BatchNorm.
// Thread = (pixel, 8 channels); a block covers 256/C8 pixels per step, grid-stride; per-channel
// partial sums are reduced over the block in a fixed order and stored as this block's partial row
// (sums layout: [1 + CHAP_ACT_BWD_SLOTS][2][C], row 0 = totals written by act_bwd_sum_kernel).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ void act_bwd_dz(const chap_act_bwd_params& P, const float sa[8], const float sb[8], int n, long pix, int y, int x, int c8,
                                           float raw[8], float dz[8]) {
    const chap_src_t& s = P.r;
    const int C = s.C;
    ld8((const T*)s.ptr + pix * s.ld + s.coff + c8, raw);
    float gsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gsum[j] = 0.f;
    for (int k = 0; k < P.ng; ++k) {
        float v[8];
        ld8((const T*)P.g[k] + pix * P.g_ld[k] + P.g_coff[k] + c8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) gsum[j] += v[j];
    }
    if (P.g_pool) {
        const int OH = P.H / 2, OW = P.W / 2;
        const long pp = ((long)n * OH + (y >> 1)) * OW + (x >> 1);
        const uint32_t me = ((y & 1) << 1) | (x & 1);
        float v[8];
        ld8((const T*)P.g_pool + pp * C + c8, v);
        const uint2 m = *(const uint2*)(P.pool_idx + pp * C + c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t w = j < 4 ? m.x : m.y;
            if (((w >> (8 * (j & 3))) & 0xff) == me) gsum[j] += v[j];
        }
    }
    // a = keep*ks*cm*leaky(z), z = scale*raw+shift  ->  da/dz = keep*ks*cm*(z>0 ? 1 : slope)      (actbwd_math.h)
    actbwd_deriv8(raw, gsum, sa, sb, s.act != 0, s.slope, dz);
    if (s.keep) actbwd_keep8(dz, *(const uint2*)(s.keep + pix * C + c8), s.keep_scale);
    if (s.chan_mul) {
        float cm[8];
        ld8(s.chan_mul + (long)n * C + c8, cm);
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] *= cm[j];
    }
}

template <typename T, bool APPLY>
__device__ __forceinline__ void act_bwd_kernel(const chap_act_bwd_params& P) {
    extern __shared__ float red[];            // reduce phase: [4 waves][2][C] partials; apply phase: [2][C] totals
    const int C = P.r.C, C8 = C / 8;
    const long npix = (long)P.N * P.D * P.H * P.W;
    const int c8 = (threadIdx.x % C8) * 8;
    const int prow = threadIdx.x / C8, PPB = 256 / C8;      // C8 in {2,4,8,...,32} divides 256
    float s0[8], s1[8], mean[8], istd[8], k0[8], sa[8], sb[8], cB[8], cC[8], cM[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; sa[j] = 1.f; sb[j] = 0.f; }
    if (P.r.scale) { ld8(P.r.scale + c8, sa); ld8(P.r.shift + c8, sb); }      // per-channel constants: loaded once
#pragma unroll
    for (int j = 0; j < 8; ++j) { k0[j] = 1.f; mean[j] = 0.f; istd[j] = 1.f; }
    if (P.bn) { ld8(P.mean + c8, mean); ld8(P.invstd + c8, istd); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { cB[j] = 0.f; cC[j] = 0.f; cM[j] = -mean[j] * istd[j]; }      // xhat = raw*istd + cM
    if (APPLY) {
        if (P.bn == 1) {                               // training-mode BatchNorm backward
            for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = P.sums[i];      // totals in row 0 (written by act_bwd_sum_kernel together with dgamma / dbeta)
            __syncthreads();
            float gm[8], a0[8], a1[8];
            ld8(P.gamma + c8, gm);
#pragma unroll
            for (int j = 0; j < 8; ++j) { a0[j] = red[c8 + j]; a1[j] = red[C + c8 + j]; }
            // g = k0*(dz - k1 - xhat*k2), xhat = raw*istd - mean*istd  ==  dz*k0 + raw*cB + cC  (two FMAs per element; actbwd_math.h)
            actbwd_consts8(a0, a1, gm, mean, istd, P.count, k0, cB, cC);
        } else if (P.r.scale) {                        // fixed affine (eval-mode BN): dz/draw = scale
            ld8(P.r.scale + c8, k0);
        }
    }
    const bool need_coords = P.g_pool != nullptr || P.r.chan_mul != nullptr;
    if (prow < PPB) {
        for (long pix = (long)blockIdx.x * PPB + prow; pix < npix; pix += (long)gridDim.x * PPB) {
            int x = 0, y = 0, n = 0;
            if (need_coords) {                     // wave-uniform: only the pooled gradient / Dropout3d path needs (n, y, x)
                const u32 up = (u32)pix;
                x = (int)(up % (u32)P.W); u32 r = up / (u32)P.W;
                y = (int)(r % (u32)P.H); r /= (u32)P.H;
                n = (int)(r / (u32)P.D);
            }
            float raw[8], dz[8];
            act_bwd_dz<T>(P, sa, sb, n, pix, y, x, c8, raw, dz);
            if (!APPLY) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 d = {dz[2 * j], dz[2 * j + 1]}, r2 = {raw[2 * j], raw[2 * j + 1]};
                    const f32x2 i2 = {istd[2 * j], istd[2 * j + 1]}, m2 = {cM[2 * j], cM[2 * j + 1]};
                    const f32x2 xh = r2 * i2 + m2;
                    f32x2 a = {s0[2 * j], s0[2 * j + 1]}, b = {s1[2 * j], s1[2 * j + 1]};
                    a = a + d; b = d * xh + b;
                    s0[2 * j] = a.x; s0[2 * j + 1] = a.y; s1[2 * j] = b.x; s1[2 * j + 1] = b.y;
                }
            } else {
                float o[8];
                actbwd_apply8(dz, raw, k0, cB, cC, o);         // bn != 1: cB = cC = 0 -> dz*k0
                st8((T*)P.gout + pix * C + c8, o);
            }
        }
    }
    if (!APPLY) {
        for (int i = threadIdx.x; i < 4 * 2 * C; i += 256) red[i] = 0.f;
        __syncthreads();
        // lanes l, l+C8, l+2*C8, ... of a wave hold the same 8 channels: shuffle-reduce them (fixed pairing), one row of
        // partials per wave in LDS, the four rows summed in a fixed order -> this block's partial row.  No atomics.
        const int wave = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = s0[j], b = s1[j];
            // inside a 16-lane row: DPP rotations (C8 is wave-uniform); across rows: two shuffles
            if (C8 == 2) { a = row16_stride_sum<2>(a); b = row16_stride_sum<2>(b); }
            else if (C8 == 4) { a = row16_stride_sum<4>(a); b = row16_stride_sum<4>(b); }
            else if (C8 == 8) { a = row16_stride_sum<8>(a); b = row16_stride_sum<8>(b); }
            for (int o = (C8 > 16 ? C8 : 16); o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if ((threadIdx.x & 63) < C8 && prow < PPB) { red[(wave * 2 + 0) * C + c8 + j] = a; red[(wave * 2 + 1) * C + c8 + j] = b; }
        }
        __syncthreads();
        float* dst = P.sums + (long)(1 + blockIdx.x) * 2 * C;
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            const int which = i / C, c = i % C;
            dst[i] = (red[(0 * 2 + which) * C + c] + red[(1 * 2 + which) * C + c]) + (red[(2 * 2 + which) * C + c] + red[(3 * 2 + which) * C + c]);
        }
    }
}

// Fixed-order total of the per-block partial rows (one wave per value, fp64) into row 0, read by the apply phase;
// accumulates the BatchNorm parameter gradients.
struct act_bwd_sum_args { float* sums; int nblocks; float* dgamma; float* dbeta; int C; };
__device__ __forceinline__ void act_bwd_sum_kernel(const act_bwd_sum_args& A) {
    float* sums = A.sums; const int nblocks = A.nblocks; float* dgamma = A.dgamma; float* dbeta = A.dbeta; const int C = A.C;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= 2 * C) return;
    double t = 0.0;
    constexpr int NB = CHAP_ACT_BWD_SLOTS / 64;                  // all of a lane's loads in flight at once, adds in the fixed order
    float vv[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int b = lane + 64 * k;
        vv[k] = sums[(long)(1 + (b < nblocks ? b : 0)) * 2 * C + i];
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) t += (lane + 64 * k < nblocks) ? (double)vv[k] : 0.0;
    t = wave_sum_f64(t);
    if (lane != 0) return;
    const float v = (float)t;
    sums[i] = v;
    if (i < C) { if (dbeta) dbeta[i] += v; } else if (dgamma) dgamma[i - C] += v;
}

static int act_bwd_check(const chap_act_bwd_params* p) {
    CHAP_CHECK_ARG(p && p->r.ptr, "chap_act_bwd: null argument");
    CHAP_CHECK_ARG(p->r.C % 8 == 0 && 256 % (p->r.C / 8) == 0 && p->r.C <= 1024, "chap_act_bwd: C=%d unsupported", p->r.C);
    CHAP_CHECK_ARG(p->ng >= 0 && p->ng <= 3, "chap_act_bwd: ng=%d", p->ng);
    CHAP_CHECK_ARG(p->bn >= 0 && p->bn <= 2, "chap_act_bwd: bn=%d", p->bn);
    CHAP_CHECK_ARG(p->bn != 1 || (p->mean && p->invstd && p->gamma && p->sums), "chap_act_bwd: bn=1 needs mean/invstd/gamma/sums");
    CHAP_CHECK_ARG(!p->g_pool || (p->pool_idx && p->D == 1 && p->H % 2 == 0 && p->W % 2 == 0), "chap_act_bwd: pooled gradient needs idx and even 2D dims");
    return CHAP_OK;
}
// (Round 3 measured the totals folded into the apply phase -- every apply block re-reading the partial rows instead of the totals kernel:
// 2D 6.77 -> 6.95 ms, 3D 14.91 -> 15.04 ms per step, slower; removed in round 4, the record is in DESIGN.md section 5.)
static int act_bwd_blocks(const chap_act_bwd_params* p) {
    const long npix = (long)p->N * p->D * p->H * p->W;
    const int ppb = 256 / (p->r.C / 8);
    long b = (npix + ppb - 1) / ppb;
    // 2 blocks per CU, swept on the whole iteration (round 2, final tree: 256 / 384 / 512 / 640 / 768 / 1024 blocks -> 7.36 / 7.28 / 7.28 / 7.31 /
    // 7.38 / 7.41 ms per 2D step, 3D 16.80-16.87 for all): more blocks add partial rows to total and take CUs from the kernels of the
    // other streams
    static long cap = 0;
    if (!cap) { const char* e = getenv("CHAP_ACTBWD_BLOCKS"); cap = (e && atol(e) > 0 && atol(e) <= CHAP_ACT_BWD_SLOTS) ? atol(e) : 512; }      // lab knob
    return (int)(b < cap ? b : cap);          // one partial row per block
}
extern "C" int chap_act_bwd_reduce(const chap_act_bwd_params* p, void* stream) {
    int r = act_bwd_check(p); if (r) return r;
    CHAP_CHECK_ARG(p->bn && p->mean && p->invstd && p->sums, "chap_act_bwd_reduce: needs bn, mean, invstd, sums");
    const size_t lds = 4 * 2 * p->r.C * sizeof(float);
    const int nb = act_bwd_blocks(p);              // one partial row per block
    if (p->dtype == CHAP_BF16) r = chap_launch<chap_act_bwd_params, act_bwd_kernel<bf16_t, false>, 256>(dim3(nb), dim3(256), lds, (hipStream_t)stream, *p, "chap_act_bwd_reduce");
    else r = chap_launch<chap_act_bwd_params, act_bwd_kernel<float, false>, 256>(dim3(nb), dim3(256), lds, (hipStream_t)stream, *p, "chap_act_bwd_reduce");
    if (r) return r;
#ifdef CHAP_LAB      // lab builds only: timing bound, wrong numerics
    { static int skip = -1; if (skip < 0) skip = getenv("CHAP_LAB_SKIP_ACTSUM") ? 1 : 0; if (skip) return CHAP_OK; }
#endif
    const act_bwd_sum_args sa = {p->sums, nb, p->dgamma, p->dbeta, p->r.C};
    return chap_launch<act_bwd_sum_args, act_bwd_sum_kernel, 256>(dim3(cdiv(2 * p->r.C, 4)), dim3(256), 0, (hipStream_t)stream, sa, "chap_act_bwd_reduce(sum)");
}
extern "C" int chap_act_bwd_apply(const chap_act_bwd_params* p, void* stream) {
    int r = act_bwd_check(p); if (r) return r;
    CHAP_CHECK_ARG(p->gout, "chap_act_bwd_apply: null gout");
#ifdef CHAP_LAB      // lab builds only: timing bound (wrong numerics) for folding this pass into the loads of its consumers
    { static int skip = -1; if (skip < 0) skip = getenv("CHAP_LAB_SKIP_ACTAPPLY") ? 1 : 0; if (skip && p->bn == 1) return CHAP_OK; }
#endif
    const size_t lds = 2 * p->r.C * sizeof(float);
    if (p->dtype == CHAP_BF16) return chap_launch<chap_act_bwd_params, act_bwd_kernel<bf16_t, true>, 256>(dim3(act_bwd_blocks(p)), dim3(256), lds, (hipStream_t)stream, *p, "chap_act_bwd_apply");
    return chap_launch<chap_act_bwd_params, act_bwd_kernel<float, true>, 256>(dim3(act_bwd_blocks(p)), dim3(256), lds, (hipStream_t)stream, *p, "chap_act_bwd_apply");
}

// =========================================================================================
// Layout converters at the module boundary.
template <typename T>
__device__ __forceinline__ void planar_to_cl_kernel(const chap_planar_to_cl_params& P) {
    const int Cp = P.Cpad > P.C ? P.Cpad : P.C;       // channels [C, Cpad) are written as zeros
    const long total = (long)P.N * P.P * Cp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const u32 ui = (u32)i;
        const int c = (int)(ui % (u32)Cp); const u32 r = ui / (u32)Cp;
        const long pp = r % (u32)P.P; const int n = (int)(r / (u32)P.P);
        const float v = c < P.C ? P.in[((long)n * P.C + c) * P.P + pp] : 0.f;
        ((T*)P.out)[(n * (long)P.P + pp) * P.out_ld + P.out_coff + c] = elem<T>::put(v);
    }
}
// vector form: thread = (pixel, 8 channels), one 16-/32-byte store (the scalar form above writes 2 bytes per thread)
template <typename T>
__device__ __forceinline__ void planar_to_cl8_kernel(const chap_planar_to_cl_params& P) {
    const int Cp = P.Cpad > P.C ? P.Cpad : P.C, C8 = Cp / 8;
    const long total = (long)P.N * P.P * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const u32 ui = (u32)i;
        const int c8 = (int)(ui % (u32)C8) * 8; const u32 r = ui / (u32)C8;
        const long pp = r % (u32)P.P; const int n = (int)(r / (u32)P.P);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = c8 + j < P.C ? P.in[((long)n * P.C + c8 + j) * P.P + pp] : 0.f;
        st8((T*)P.out + (n * (long)P.P + pp) * P.out_ld + P.out_coff + c8, v);
    }
}
extern "C" int chap_planar_to_cl(const chap_planar_to_cl_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->in && p->out, "chap_planar_to_cl: null argument");
    const int Cp = p->Cpad > p->C ? p->Cpad : p->C;
    if (Cp % 8 == 0 && p->out_ld % 8 == 0 && p->out_coff % 8 == 0) {
        const long total = (long)p->N * p->P * (Cp / 8);
        const int blocks = chap_blocks(total, 16384);
        if (p->dtype == CHAP_BF16) return chap_launch<chap_planar_to_cl_params, planar_to_cl8_kernel<bf16_t>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_planar_to_cl");
        return chap_launch<chap_planar_to_cl_params, planar_to_cl8_kernel<float>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_planar_to_cl");
    }
    const long total = (long)p->N * p->P * Cp;
    const int blocks = chap_blocks(total, 4096);
    if (p->dtype == CHAP_BF16) return chap_launch<chap_planar_to_cl_params, planar_to_cl_kernel<bf16_t>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_planar_to_cl");
    return chap_launch<chap_planar_to_cl_params, planar_to_cl_kernel<float>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_planar_to_cl");
}

template <typename T>
__device__ __forceinline__ void cl_to_planar_kernel(const chap_cl_to_planar_params& P) {
    const int C = P.r.C;
    const long total = (long)P.N * P.P * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const u32 ui = (u32)i;
        const long pp = ui % (u32)P.P; const u32 r = ui / (u32)P.P;
        const int c = (int)(r % (u32)C); const int n = (int)(r / (u32)C);
        P.out[i] = src_load1<T>(P.r, n, (long)n * P.P + pp, c);
    }
}
extern "C" int chap_cl_to_planar(const chap_cl_to_planar_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->r.ptr && p->out, "chap_cl_to_planar: null argument");
    const long total = (long)p->N * p->P * p->r.C;
    const int blocks = chap_blocks(total, 4096);
    if (p->dtype == CHAP_BF16) return chap_launch<chap_cl_to_planar_params, cl_to_planar_kernel<bf16_t>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_cl_to_planar");
    return chap_launch<chap_cl_to_planar_params, cl_to_planar_kernel<float>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_cl_to_planar");
}

// =========================================================================================
// out[c] += sum over pixels of a (lazy) channel-last tensor: bias gradient of layers whose output
// gradient is not the B operand of chap_wgrad (transposed conv).
template <typename T>
__device__ __forceinline__ void channel_sum_kernel(const chap_chansum_params& P) {
    extern __shared__ float red[];             // [4 waves][C]
    const int C = P.r.C, C8 = C / 8;
    const int c8 = (threadIdx.x % C8) * 8, prow = threadIdx.x / C8, PPB = 256 / C8;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    for (long pix = (long)blockIdx.x * PPB + prow; pix < P.npix; pix += (long)gridDim.x * PPB) {
        float v[8];
        src_load8<T>(P.r, (int)((u32)pix / (u32)P.pix_per_sample), pix, c8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
    for (int i = threadIdx.x; i < 4 * C; i += 256) red[i] = 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = s[j];
        for (int o = C8; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
        if ((threadIdx.x & 63) < C8) red[wave * C + c8 + j] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) P.ws[(long)blockIdx.x * C + i] = (red[i] + red[C + i]) + (red[2 * C + i] + red[3 * C + i]);
}
// out[c] += fixed-order total of the block partials (one wave per channel, fp64)
struct chansum_final_args { const float* ws; int nblocks, C; float* out; };
__device__ __forceinline__ void channel_sum_final_kernel(const chansum_final_args& A) {
    const float* ws = A.ws; const int nblocks = A.nblocks, C = A.C; float* out = A.out;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double t = 0.0;
    for (int b = lane; b < nblocks; b += 64) t += (double)ws[(long)b * C + c];
    t = wave_sum_f64(t);
    if (lane == 0) out[c] += (float)t;
}
extern "C" int chap_channel_sum(const chap_chansum_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->r.ptr && p->out && p->ws && p->r.C % 8 == 0 && 256 % (p->r.C / 8) == 0 && p->r.C <= 512, "chap_channel_sum: bad argument");
    const int ppb = 256 / (p->r.C / 8);
    long b = (p->npix + ppb - 1) / ppb;
    const int nb = (int)(b < CHAP_CHANSUM_SLOTS ? b : CHAP_CHANSUM_SLOTS);
    const size_t lds = 4 * p->r.C * sizeof(float);
    int r;
    if (p->dtype == CHAP_BF16) r = chap_launch<chap_chansum_params, channel_sum_kernel<bf16_t>, 256>(dim3(nb), dim3(256), lds, (hipStream_t)stream, *p, "chap_channel_sum");
    else r = chap_launch<chap_chansum_params, channel_sum_kernel<float>, 256>(dim3(nb), dim3(256), lds, (hipStream_t)stream, *p, "chap_channel_sum");
    if (r) return r;
    const chansum_final_args fa = {(const float*)p->ws, nb, p->r.C, p->out};
    return chap_launch<chansum_final_args, channel_sum_final_kernel, 256>(dim3(cdiv(p->r.C, 4)), dim3(256), 0, (hipStream_t)stream, fa, "chap_channel_sum(final)");
}
