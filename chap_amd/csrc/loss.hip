// Segmentation-loss kernels on fp32 planar logits [N][C][P] (C <= 8): one thread per pixel keeps
// the C logits in registers, softmax in registers, wave64 shuffle reductions, one partial row per block and a
// fixed-order fp64 total (no atomics: bitwise reproducible).  HBM-bound: each logit is read once per pass.
#include "common.h"

// Fixed-order wave reduction in fp64 (xor butterfly: the same pairing on every run).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Row 0 of a [1 + nrows][rl] partial workspace <- fixed-order column sums of rows 1..nrows (one wave per column, fp64).
// Called by all 256 threads of ONE block; `tot` (shared, rl floats) receives the totals too.  Ends with a barrier.
__device__ __forceinline__ void sum_partial_rows(float* ws, int nrows, int rl, float* tot) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (rl <= 32 && nrows <= CHAP_LOSS_SLOTS) {
        // every load of the block first (a wave's columns wave, wave + 4, ..: at most 8, a lane's rows lane, lane + 64, ..: at most 8),
        // then the sums in the same fixed order as the loop below -- the kernel is one dependent round trip instead of eight
        constexpr int NR = 8, NB = CHAP_LOSS_SLOTS / 64;
        float v[NR][NB];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = wave + 4 * r;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int b = lane + 64 * k;
                v[r][k] = ws[(long)(1 + (b < nrows ? b : 0)) * rl + (i < rl ? i : 0)];
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = wave + 4 * r;
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < NB; ++k) t += (lane + 64 * k < nrows) ? (double)v[r][k] : 0.0;
            t = wave_sum_f64(t);
            if (lane == 0 && i < rl) { tot[i] = (float)t; ws[i] = (float)t; }
        }
        __syncthreads();
        return;
    }
    for (int i = wave; i < rl; i += 4) {
        double t = 0.0;
        for (int b = lane; b < nrows; b += 64) t += (double)ws[(long)(1 + b) * rl + i];
        t = wave_sum_f64(t);
        if (lane == 0) { tot[i] = (float)t; ws[i] = (float)t; }
    }
    __syncthreads();
}


template <int C>
__device__ __forceinline__ void softmax_px(const float* __restrict__ lg, long base, long P, float z[C], float p[C], float& lse) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) { z[c] = lg[base + c * P]; m = fmaxf(m, z[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = expf(z[c] - m); s += p[c]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
    lse = m + logf(s);
}

// acc layout per part k in {a, b}: [0] ce_sum, [1..C] I_c, [1+C..2C] Z_c, [1+2C..3C] Y_c, [1+3C] msum
template <int C>
__global__ __launch_bounds__(256) void mix_loss_acc_kernel(const chap_mix_loss_params P_) {
    constexpr int NA = 2 + 3 * C;
    __shared__ float red[4][2 * NA];
    float a[2][NA];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < NA; ++i) a[k][i] = 0.f;
    const long P = P_.P, total = (long)P_.N * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = (unsigned)i / (unsigned)P, pp = (unsigned)i % (unsigned)P;
        float z[C], p[C], lse;
        softmax_px<C>(P_.logits, n * C * P + pp, P, z, p, lse);
        const float m = P_.mask ? (float)P_.mask[i] : 1.f;
        const int t[2] = {(int)P_.target_a[i], (int)(P_.target_b ? P_.target_b[i] : P_.target_a[i])};
        const float mk[2] = {m, 1.f - m};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float zt = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float tc = (t[k] == c) ? 1.f : 0.f;
                zt += tc * z[c];
                a[k][1 + c] += p[c] * tc * mk[k];
                a[k][1 + C + c] += p[c] * p[c] * mk[k];
                a[k][1 + 2 * C + c] += tc * mk[k];
            }
            a[k][0] += (lse - zt) * mk[k];
            a[k][1 + 3 * C] += mk[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float v = wave_sum(a[k][i]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k * NA + i] = v;
        }
    __syncthreads();
    // this block's partial row (fixed order over the four waves; no atomics): acc[1 + block][2*NA]
    for (int i = threadIdx.x; i < 2 * NA; i += 256) P_.acc[(long)(1 + blockIdx.x) * 2 * NA + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// Totals (row 0 of acc, fixed-order sum of the block partials) and the mix_loss return triple (train_ours_2D.py:205-216).
__global__ __launch_bounds__(256) void mix_loss_final_kernel(float* acc, int nblocks, float* loss, int C, float w_a, float w_b, float smooth, float k_dice, float k_ce) {
    __shared__ float tot[2 * (2 + 3 * 8)];
    const int NA = 2 + 3 * C;
    sum_partial_rows(acc, nblocks, 2 * NA, tot);
    if (threadIdx.x != 0) return;
    float part[2];
    const float w[2] = {w_a, w_b};
    for (int k = 0; k < 2; ++k) {
        const float* a = tot + k * NA;
        float dice = 0.f;
        for (int c = 0; c < C; ++c) dice += 1.f - (2.f * a[1 + c] + smooth) / (a[1 + C + c] + a[1 + 2 * C + c] + smooth);
        dice = dice / C * w[k];
        const float ce = w[k] * a[0] / (a[1 + 3 * C] + 1e-16f);
        part[k] = k_dice * dice + k_ce * ce;
    }
    loss[0] = part[0]; loss[1] = part[1]; loss[2] = part[0] + part[1];
}

template <int C>
__global__ __launch_bounds__(256) void mix_loss_bwd_kernel(const chap_mix_loss_params P_) {
    constexpr int NA = 2 + 3 * C;
    __shared__ float sa[2 * NA];
    for (int i = threadIdx.x; i < 2 * NA; i += 256) sa[i] = P_.acc[i];
    __syncthreads();
    const long P = P_.P, total = (long)P_.N * P;
    const float w[2] = {P_.w_a, P_.w_b};
    const float gsc = P_.gscale * (P_.gscale_dev ? *P_.gscale_dev : 1.f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = (unsigned)i / (unsigned)P, pp = (unsigned)i % (unsigned)P;
        float z[C], p[C], lse;
        softmax_px<C>(P_.logits, n * C * P + pp, P, z, p, lse);
        const float m = P_.mask ? (float)P_.mask[i] : 1.f;
        const int t[2] = {(int)P_.target_a[i], (int)(P_.target_b ? P_.target_b[i] : P_.target_a[i])};
        const float mk[2] = {m, 1.f - m};
        const bool dflt = P_.k_dice == 0.f && P_.k_ce == 0.f;
        const float kd = dflt ? 0.5f : P_.k_dice, kc = dflt ? 0.5f : P_.k_ce;
        float dz[C], dp[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { dz[c] = 0.f; dp[c] = 0.f; }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float* a = sa + k * NA;
            const float kce = w[k] * mk[k] / (a[1 + 3 * C] + 1e-16f);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float tc = (t[k] == c) ? 1.f : 0.f;
                dz[c] += kce * (p[c] - tc);
                const float den = a[1 + C + c] + a[1 + 2 * C + c] + P_.smooth;
                dp[c] += (w[k] / C) * mk[k] * (-2.f * tc / den + (2.f * a[1 + c] + P_.smooth) * 2.f * p[c] / (den * den));
            }
        }
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) dot += dp[c] * p[c];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float gz = gsc * (kc * dz[c] + kd * p[c] * (dp[c] - dot));
            float* o = P_.dlogits + n * C * P + c * P + pp;
            *o = P_.accumulate ? *o + gz : gz;
        }
    }
}

static int loss_blocks(long total) { long b = (total + 255) / 256; return (int)(b < 2048 ? b : 2048); }
static int loss_slots(long total) { long b = (total + 255) / 256; return (int)(b < CHAP_LOSS_SLOTS ? b : CHAP_LOSS_SLOTS); }   // kernels that leave one partial row per block

extern "C" int chap_mix_loss_fwd(const chap_mix_loss_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->logits && p->target_a && p->acc && p->loss, "chap_mix_loss_fwd: null argument");
    CHAP_CHECK_ARG(p->C == 4 || p->C == 2, "chap_mix_loss: C=%d (2 or 4 built)", p->C);
    CHAP_CHECK_ARG((long)p->N * p->P < (1L << 32), "chap_mix_loss: N*P=%ld exceeds the 32-bit pixel index", (long)p->N * p->P);
    const int nb = loss_slots((long)p->N * p->P);
    hipStream_t s = (hipStream_t)stream;
    if (p->C == 4) hipLaunchKernelGGL(mix_loss_acc_kernel<4>, dim3(nb), dim3(256), 0, s, *p);
    else hipLaunchKernelGGL(mix_loss_acc_kernel<2>, dim3(nb), dim3(256), 0, s, *p);
    CHAP_LAUNCH_CHECK("chap_mix_loss_fwd");
    const bool dflt = p->k_dice == 0.f && p->k_ce == 0.f;
    hipLaunchKernelGGL(mix_loss_final_kernel, dim3(1), dim3(256), 0, s, p->acc, nb, p->loss, p->C, p->w_a, p->w_b, p->smooth,
                       dflt ? 0.5f : p->k_dice, dflt ? 0.5f : p->k_ce);
    CHAP_LAUNCH_CHECK("chap_mix_loss_fwd(final)");
    return CHAP_OK;
}
extern "C" int chap_mix_loss_bwd(const chap_mix_loss_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->logits && p->target_a && p->acc && p->dlogits, "chap_mix_loss_bwd: null argument");
    CHAP_CHECK_ARG(p->C == 4 || p->C == 2, "chap_mix_loss: C=%d (2 or 4 built)", p->C);
    CHAP_CHECK_ARG((long)p->N * p->P < (1L << 32), "chap_mix_loss: N*P=%ld exceeds the 32-bit pixel index", (long)p->N * p->P);
    const int nb = loss_blocks((long)p->N * p->P);
    if (p->C == 4) hipLaunchKernelGGL(mix_loss_bwd_kernel<4>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(mix_loss_bwd_kernel<2>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_mix_loss_bwd");
    return CHAP_OK;
}

// pass-A pseudo-label block (train_ours_2D.py:319-325)
template <int C>
__global__ __launch_bounds__(256) void pseudo_kernel(const chap_pseudo_params P_) {
    const long P = P_.P, total = (long)P_.N * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = (unsigned)i / (unsigned)P, pp = (unsigned)i % (unsigned)P, base = n * C * P + pp;
        float z1[C], p1[C], l1, z2[C], p2[C], l2;
        softmax_px<C>(P_.logits1, base, P, z1, p1, l1);
        softmax_px<C>(P_.logits2, base, P, z2, p2, l2);
        int a1 = 0, a2 = 0;
#pragma unroll
        for (int c = 1; c < C; ++c) { if (p1[c] > p1[a1]) a1 = c; if (p2[c] > p2[a2]) a2 = c; }   // first max wins (torch.argmax)
        if (P_.soft1) {
#pragma unroll
            for (int c = 0; c < C; ++c) { P_.soft1[base + c * P] = p1[c]; P_.soft2[base + c * P] = p2[c]; }
        }
        if (P_.arg1) { P_.arg1[i] = a1; P_.arg2[i] = a2; }
        if (P_.knowledge) {
            float za = 0.f, zb = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) { za += (c == a2) ? z1[c] : 0.f; zb += (c == a1) ? z2[c] : 0.f; }
            P_.knowledge[i] = (l1 - za) + (l2 - zb);
        }
    }
}
extern "C" int chap_pseudo_block(const chap_pseudo_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->logits1 && p->logits2, "chap_pseudo_block: null argument");
    CHAP_CHECK_ARG(p->C == 4 || p->C == 2, "chap_pseudo_block: C=%d (2 or 4 built)", p->C);
    CHAP_CHECK_ARG((p->soft1 == nullptr) == (p->soft2 == nullptr) && (p->arg1 == nullptr) == (p->arg2 == nullptr), "chap_pseudo_block: outputs come in pairs");
    CHAP_CHECK_ARG((long)p->N * p->P < (1L << 32), "chap_pseudo_block: N*P=%ld exceeds the 32-bit pixel index", (long)p->N * p->P);
    const int nb = loss_blocks((long)p->N * p->P);
    if (p->C == 4) hipLaunchKernelGGL(pseudo_kernel<4>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(pseudo_kernel<2>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_pseudo_block");
    return CHAP_OK;
}

// VAT distances between the two heads' logits and their targets, with the gradient wrt the logits.
//   mode 0 (kl):   sum_heads mean_{n,p} KL(target || softmax(logits)) -- one pass (the gradient is local);
//   mode 1 (dice): sum_heads mean_c [1 - (2 I_c + s) / (Z_c + Y_c + s)],  I = sum p t, Z = sum p^2, Y = sum t^2 over (n, p)
//                  (the soft Dice of the VAT reference implementations this script family uses for adv_losstype='dice'):
//                  pass 1 accumulates (I, Z, Y), pass 2 writes the gradient through the softmax.
// ws rows: [1 + CHAP_LOSS_SLOTS][2 heads][3*C + 1]; the KL partial loss of a block sits in column 0.
template <int C>
__global__ __launch_bounds__(256) void kl_kernel(const chap_kl_params P_) {
    __shared__ float red[4];
    const long P = P_.P, total = (long)P_.N * P;
    const float inv = 1.f / (float)total;
    const float gs = P_.gscale * (P_.gscale_dev ? *P_.gscale_dev : 1.f) * inv;
    float acc = 0.f;
    const bool quads = (P & 3) == 0 && (((uintptr_t)P_.logits[0] | (uintptr_t)P_.logits[1] | (uintptr_t)P_.target[0] | (uintptr_t)P_.target[1] |
                                         (uintptr_t)P_.dlogits[0] | (uintptr_t)P_.dlogits[1]) & 15) == 0;
    if (quads) {
        // four consecutive pixels per thread: 16-byte loads / stores of every class plane (the kernel sits between the VAT forward and
        // backward passes, alone on the GPU; one pixel per thread left it at 2 TB/s)
        const long totalq = total >> 2;
        for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < totalq; q += (long)gridDim.x * 256) {
            const long i = q << 2;
            const long n = (unsigned)i / (unsigned)P, pp = (unsigned)i % (unsigned)P, base = n * C * P + pp;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float z[C][4], t[C][4], g[C][4];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float4 a = *(const float4*)(P_.logits[h] + base + c * P), b = *(const float4*)(P_.target[h] + base + c * P);
                    z[c][0] = a.x; z[c][1] = a.y; z[c][2] = a.z; z[c][3] = a.w; t[c][0] = b.x; t[c][1] = b.y; t[c][2] = b.z; t[c][3] = b.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float m = -INFINITY, pr[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) m = fmaxf(m, z[c][e]);
                    float sum = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) { pr[c] = expf(z[c][e] - m); sum += pr[c]; }
                    const float inv_s = 1.f / sum, lse = m + logf(sum);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float tt = t[c][e];
                        if (tt > 0.f) acc += tt * (logf(tt) - (z[c][e] - lse));
                        g[c][e] = gs * (pr[c] * inv_s - tt);
                    }
                }
                if (P_.dlogits[h]) {
#pragma unroll
                    for (int c = 0; c < C; ++c) *(float4*)(P_.dlogits[h] + base + c * P) = make_float4(g[c][0], g[c][1], g[c][2], g[c][3]);
                }
            }
        }
    } else
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = (unsigned)i / (unsigned)P, pp = (unsigned)i % (unsigned)P, base = n * C * P + pp;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float z[C], p[C], lse;
            softmax_px<C>(P_.logits[h], base, P, z, p, lse);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float t = P_.target[h][base + c * P];
                if (t > 0.f) acc += t * (logf(t) - (z[c] - lse));
                if (P_.dlogits[h]) P_.dlogits[h][base + c * P] = gs * (p[c] - t);
            }
        }
    }
    if (!P_.ws) return;
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) P_.ws[(long)(1 + blockIdx.x) * 2 * (3 * C + 1)] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int C>
__global__ __launch_bounds__(256) void dice_dist_acc_kernel(const chap_kl_params P_) {
    constexpr int RH = 3 * C + 1, RL = 2 * RH;
    __shared__ float red[4][RL];
    const long P = P_.P, total = (long)P_.N * P;
    float a[2][3 * C];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 3 * C; ++i) a[h][i] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = (unsigned)i / (unsigned)P, pp = (unsigned)i % (unsigned)P, base = n * C * P + pp;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float z[C], p[C], lse;
            softmax_px<C>(P_.logits[h], base, P, z, p, lse);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float t = P_.target[h][base + c * P];
                a[h][c] += p[c] * t; a[h][C + c] += p[c] * p[c]; a[h][2 * C + c] += t * t;
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 3 * C; ++i) {
            const float v = wave_sum(a[h][i]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][h * RH + i] = v;
        }
    if (threadIdx.x < 4) { red[threadIdx.x][3 * C] = 0.f; red[threadIdx.x][RH + 3 * C] = 0.f; }
    __syncthreads();
    for (int i = threadIdx.x; i < RL; i += 256) P_.ws[(long)(1 + blockIdx.x) * RL + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

constexpr float DICE_DIST_SMOOTH = 1e-10f;

// totals -> row 0; loss += the distance (one thread: deterministic)
__global__ __launch_bounds__(256) void dist_final_kernel(float* ws, int nblocks, int C, int mode, float inv_total, float* loss) {
    __shared__ float tot[2 * (3 * 8 + 1)];
    const int RH = 3 * C + 1;
    sum_partial_rows(ws, nblocks, 2 * RH, tot);
    if (threadIdx.x != 0 || !loss) return;
    if (mode == 0) { *loss += tot[0] * inv_total; return; }
    float d = 0.f;
    for (int h = 0; h < 2; ++h)
        for (int c = 0; c < C; ++c) d += (1.f - (2.f * tot[h * RH + c] + DICE_DIST_SMOOTH) / (tot[h * RH + C + c] + tot[h * RH + 2 * C + c] + DICE_DIST_SMOOTH)) / (float)C;
    *loss += d;
}

template <int C>
__global__ __launch_bounds__(256) void dice_dist_grad_kernel(const chap_kl_params P_) {
    constexpr int RH = 3 * C + 1;
    __shared__ float sa[2 * RH];
    for (int i = threadIdx.x; i < 2 * RH; i += 256) sa[i] = P_.ws[i];
    __syncthreads();
    const long P = P_.P, total = (long)P_.N * P;
    const float gs = P_.gscale * (P_.gscale_dev ? *P_.gscale_dev : 1.f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = (unsigned)i / (unsigned)P, pp = (unsigned)i % (unsigned)P, base = n * C * P + pp;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!P_.dlogits[h]) continue;
            float z[C], p[C], lse, dp[C], dot = 0.f;
            softmax_px<C>(P_.logits[h], base, P, z, p, lse);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float t = P_.target[h][base + c * P];
                const float den = sa[h * RH + C + c] + sa[h * RH + 2 * C + c] + DICE_DIST_SMOOTH;
                dp[c] = (-2.f * t / den + (2.f * sa[h * RH + c] + DICE_DIST_SMOOTH) * 2.f * p[c] / (den * den)) / (float)C;
                dot += dp[c] * p[c];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) P_.dlogits[h][base + c * P] = gs * p[c] * (dp[c] - dot);
        }
    }
}

extern "C" int chap_kl_fwd_bwd(const chap_kl_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->logits[0] && p->logits[1] && p->target[0] && p->target[1], "chap_kl_fwd_bwd: null argument");
    CHAP_CHECK_ARG(p->C == 4 || p->C == 2, "chap_kl_fwd_bwd: C=%d (2 or 4 built)", p->C);
    CHAP_CHECK_ARG(p->mode == 0 || p->mode == 1, "chap_kl_fwd_bwd: mode=%d (0 kl, 1 dice)", p->mode);
    CHAP_CHECK_ARG(p->ws || (p->mode == 0 && !p->loss), "chap_kl_fwd_bwd: ws is required for the loss value and for the dice distance");
    const long total = (long)p->N * p->P;
    CHAP_CHECK_ARG(total < (1L << 32), "chap_kl_fwd_bwd: N*P=%ld exceeds the 32-bit pixel index", total);
    hipStream_t s = (hipStream_t)stream;
    const int nb = loss_slots(total);
    if (p->mode == 0) {
        if (p->C == 4) hipLaunchKernelGGL(kl_kernel<4>, dim3(nb), dim3(256), 0, s, *p);
        else hipLaunchKernelGGL(kl_kernel<2>, dim3(nb), dim3(256), 0, s, *p);
        CHAP_LAUNCH_CHECK("chap_kl_fwd_bwd");
        if (p->loss) {
            hipLaunchKernelGGL(dist_final_kernel, dim3(1), dim3(256), 0, s, p->ws, nb, p->C, 0, 1.f / (float)total, p->loss);
            CHAP_LAUNCH_CHECK("chap_kl_fwd_bwd(final)");
        }
        return CHAP_OK;
    }
    if (p->C == 4) hipLaunchKernelGGL(dice_dist_acc_kernel<4>, dim3(nb), dim3(256), 0, s, *p);
    else hipLaunchKernelGGL(dice_dist_acc_kernel<2>, dim3(nb), dim3(256), 0, s, *p);
    hipLaunchKernelGGL(dist_final_kernel, dim3(1), dim3(256), 0, s, p->ws, nb, p->C, 1, 1.f / (float)total, p->loss);
    if (p->dlogits[0] || p->dlogits[1]) {
        const int nbg = loss_blocks(total);
        if (p->C == 4) hipLaunchKernelGGL(dice_dist_grad_kernel<4>, dim3(nbg), dim3(256), 0, s, *p);
        else hipLaunchKernelGGL(dice_dist_grad_kernel<2>, dim3(nbg), dim3(256), 0, s, *p);
    }
    CHAP_LAUNCH_CHECK("chap_kl_fwd_bwd(dice)");
    return CHAP_OK;
}


// =========================================================================================
// Inference callers (val_2D.py:54-97, test_3D_util.py:14-79): ensemble + softmax + argmax, sliding-window scores.
constexpr int INFER_MAXC = 8;

__device__ __forceinline__ void softmax_c(float* v, int C) {
    float m = v[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, v[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) { v[c] = __expf(v[c] - m); s += v[c]; }
    const float inv = 1.f / s;
    for (int c = 0; c < C; ++c) v[c] *= inv;
}

__global__ __launch_bounds__(256) void ensemble_argmax_kernel(const chap_ensemble_params P) {
    const long total = (long)P.N * P.P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / P.P, pp = i % P.P;
        const long base = n * P.C * P.P + pp;
        float a[INFER_MAXC], b[INFER_MAXC];
        for (int c = 0; c < P.C; ++c) {
            a[c] = (P.mode != 1) ? P.logits1[base + c * P.P] : P.logits2[base + c * P.P];
            b[c] = (P.mode >= 2) ? P.logits2[base + c * P.P] : 0.f;
        }
        if (P.mode == 2) { for (int c = 0; c < P.C; ++c) a[c] = (a[c] + b[c]) / 2.0f; }
        softmax_c(a, P.C);
        if (P.mode == 3) { softmax_c(b, P.C); for (int c = 0; c < P.C; ++c) a[c] = (a[c] + b[c]) / 2.0f; }
        int best = 0;
        for (int c = 1; c < P.C; ++c) if (a[c] > a[best]) best = c;
        P.label[i] = (uint8_t)best;
        if (P.prob) for (int c = 0; c < P.C; ++c) P.prob[base + c * P.P] = a[c];
    }
}
extern "C" int chap_ensemble_argmax(const chap_ensemble_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->label && p->N > 0 && p->P > 0 && p->C >= 1 && p->C <= INFER_MAXC && p->mode >= 0 && p->mode <= 3, "chap_ensemble_argmax: bad argument");
    CHAP_CHECK_ARG((p->mode == 1 || p->logits1) && (p->mode == 0 || p->logits2), "chap_ensemble_argmax: mode %d needs the other head", p->mode);
    hipLaunchKernelGGL(ensemble_argmax_kernel, dim3(loss_blocks((long)p->N * p->P)), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_ensemble_argmax");
    return CHAP_OK;
}

// Gather form, fixed order: one thread per voxel of the volume adds the soft-max scores of the patches that cover it in PATCH
// ORDER k = 0, 1, ... -- the order of the reference's loop (test_3D_util.py:62-69) -- with plain loads and stores: no float atomics,
// the score map does not depend on scheduling (round 2 scattered the patches with atomicAdd: last-bit differences between runs).
__global__ __launch_bounds__(256) void window_accumulate_kernel(const chap_window_acc_params P) {
    const long pvox = (long)P.pw * P.ph * P.pd;
    const long vol = (long)P.W * P.H * P.D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < vol; i += (long)gridDim.x * 256) {
        const int z = (int)(i % P.D); const long r = i / P.D;
        const int y = (int)(r % P.H); const int x = (int)(r / P.H);
        float acc[INFER_MAXC], cn = 0.f;
        bool touched = false;
        for (int k = 0; k < P.npatch; ++k) {
            const int lx = x - P.origins[3 * k], ly = y - P.origins[3 * k + 1], lz = z - P.origins[3 * k + 2];
            if ((unsigned)lx >= (unsigned)P.pw || (unsigned)ly >= (unsigned)P.ph || (unsigned)lz >= (unsigned)P.pd) continue;
            if (!touched) {
                for (int c = 0; c < P.C; ++c) acc[c] = P.score[c * vol + i];
                cn = P.cnt[i];
                touched = true;
            }
            const float* lg = P.logits + (long)k * P.C * pvox + ((long)lx * P.ph + ly) * P.pd + lz;
            float v[INFER_MAXC];
            for (int c = 0; c < P.C; ++c) v[c] = lg[c * pvox];
            softmax_c(v, P.C);
            for (int c = 0; c < P.C; ++c) acc[c] += v[c];
            cn += 1.f;
        }
        if (touched) {
            for (int c = 0; c < P.C; ++c) P.score[c * vol + i] = acc[c];
            P.cnt[i] = cn;
        }
    }
}
extern "C" int chap_window_accumulate(const chap_window_acc_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->logits && p->origins && p->score && p->cnt && p->npatch > 0 && p->C >= 1 && p->C <= INFER_MAXC, "chap_window_accumulate: bad argument");
    CHAP_CHECK_ARG(p->pw > 0 && p->ph > 0 && p->pd > 0 && p->pw <= p->W && p->ph <= p->H && p->pd <= p->D, "chap_window_accumulate: patch larger than the volume");
    hipLaunchKernelGGL(window_accumulate_kernel, dim3(loss_blocks((long)p->W * p->H * p->D)), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_window_accumulate");
    return CHAP_OK;
}

__global__ __launch_bounds__(256) void window_finalize_kernel(const chap_window_fin_params P) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < P.P; i += (long)gridDim.x * 256) {
        const float cn = P.cnt[i];
        int best = 0; float bv = 0.f;
        for (int c = 0; c < P.C; ++c) {
            const float v = P.score[c * P.P + i] / cn;           // 0/0 = NaN where no patch landed, as in the reference
            P.score[c * P.P + i] = v;
            if (c == 0 || v > bv) { best = c; bv = v; }
        }
        P.label[i] = (uint8_t)best;
    }
}
extern "C" int chap_window_finalize(const chap_window_fin_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->score && p->cnt && p->label && p->C >= 1 && p->C <= INFER_MAXC && p->P > 0, "chap_window_finalize: bad argument");
    hipLaunchKernelGGL(window_finalize_kernel, dim3(loss_blocks(p->P)), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_window_finalize");
    return CHAP_OK;
}
