// Small training-loop kernels: VAT perturbation helpers, counter-based RNG, BCP box mixing,
// spatial perturbation mask (avg-pool + per-sample top-k by radix select), fused SGD.
#include "common.h"
#include "launch.h"

// ---- per-sample L2 normalise: (sample, slice) blocks write their sum of squares to ss[n][slice] (no atomics); every block of
// the scale pass re-derives the sample's total from those partials in the same fixed order -> bitwise reproducible
__global__ __launch_bounds__(256) void l2norm_sumsq_kernel(const float* __restrict__ in, float* __restrict__ ss, int P) {
    __shared__ float red[4];
    const float* x = in + (long)blockIdx.y * P;
    float s = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) { const float v = x[i]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ss[(long)blockIdx.y * CHAP_L2NORM_SLOTS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void l2norm_scale_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ ss, int P, float eps) {
    __shared__ float tot;
    if (threadIdx.x < 64) {
        double t = 0.0;
        for (int b = threadIdx.x; b < (int)gridDim.x; b += 64) t += (double)ss[(long)blockIdx.y * CHAP_L2NORM_SLOTS + b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (threadIdx.x == 0) tot = (float)t;
    }
    __syncthreads();
    const float inv = 1.f / (sqrtf(tot) + eps);
    const float* x = in + (long)blockIdx.y * P;
    float* o = out + (long)blockIdx.y * P;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) o[i] = x[i] * inv;
}
extern "C" int chap_l2_normalize(const chap_l2norm_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->in && p->out && p->ws && p->N > 0 && p->P > 0, "chap_l2_normalize: bad argument (ws = N * CHAP_L2NORM_SLOTS floats)");
    hipStream_t s = (hipStream_t)stream;
    int bx = (p->P + 256 * 8 - 1) / (256 * 8);
    if (bx > CHAP_L2NORM_SLOTS) bx = CHAP_L2NORM_SLOTS;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(l2norm_sumsq_kernel, dim3(bx, p->N), dim3(256), 0, s, p->in, p->ws, p->P);
    hipLaunchKernelGGL(l2norm_scale_kernel, dim3(bx, p->N), dim3(256), 0, s, p->in, p->out, (const float*)p->ws, p->P, p->eps);
    CHAP_LAUNCH_CHECK("chap_l2_normalize");
    return CHAP_OK;
}

__global__ void perturb_kernel(const chap_axpy_params P) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += (long)gridDim.x * blockDim.x) {
        float d = P.d[i];
        if (P.sign) d = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float m = P.mask ? P.mask[i] : 1.f;
        P.out[i] = P.x[i] + P.alpha * m * d;
    }
}
extern "C" int chap_perturb(const chap_axpy_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->x && p->d && p->out, "chap_perturb: null argument");
    const int nb = chap_blocks(p->n, 2048);
    hipLaunchKernelGGL(perturb_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_perturb");
    return CHAP_OK;
}

// ---- counter-based RNG: u = hash(seed, index) in [0,1) ----------------------------------------
__device__ __forceinline__ float u01(uint64_t seed, uint64_t i) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__global__ void rand_kernel(const chap_rand_params P) {
    const uint64_t seed = P.seed + (P.seed_dev ? *P.seed_dev * 0xD1342543DE82EF95ull : 0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += (long)gridDim.x * blockDim.x)
        P.out[i] = P.lo + (P.hi - P.lo) * u01(seed, i);
}
extern "C" int chap_rand_uniform(const chap_rand_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->out, "chap_rand_uniform: null argument");
    const int nb = chap_blocks(p->n, 2048);
    hipLaunchKernelGGL(rand_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_rand_uniform");
    return CHAP_OK;
}
// 16 mask bytes per thread and store (one byte per thread was 12.6 M byte stores for the first encoder level); the value of byte i
// depends on (seed, i) only, so any launch geometry gives the same mask.
__device__ __forceinline__ void keepmask_kernel(const chap_keepmask_params& P) {
    const uint64_t seed = P.seed + (P.seed_dev ? *P.seed_dev * 0xD1342543DE82EF95ull : 0);
    const long n16 = (P.n + 15) / 16;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n16; q += (long)gridDim.x * blockDim.x) {
        const long i0 = q * 16;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j >> 2] |= (u01(seed, i0 + j) >= P.p ? 1u : 0u) << (8 * (j & 3));
        if (i0 + 16 <= P.n && (((uintptr_t)P.keep) & 15) == 0) *(uint4*)(P.keep + i0) = make_uint4(w[0], w[1], w[2], w[3]);
        else for (int j = 0; j < 16 && i0 + j < P.n; ++j) P.keep[i0 + j] = (uint8_t)((w[j >> 2] >> (8 * (j & 3))) & 0xff);
    }
}
extern "C" int chap_keep_mask(const chap_keepmask_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->keep, "chap_keep_mask: null argument");
    // one launch geometry for every mask of at least 1 MB (grid-stride): the masks of the five encoder levels of a pass are then the lanes
    // of ONE grouped launch (chap_group_*)
    const long n16 = (p->n + 15) / 16;
    const int nb = n16 >= 256 * 256 ? 256 : (int)((n16 + 255) / 256 > 0 ? (n16 + 255) / 256 : 1);
    return chap_launch<chap_keepmask_params, keepmask_kernel, 256>(dim3(nb), dim3(256), 0, (hipStream_t)stream, *p, "chap_keep_mask");
}
__device__ __forceinline__ void chanmask_kernel(const chap_chanmask_params& P) {
    const uint64_t seed = P.seed + (P.seed_dev ? *P.seed_dev * 0xD1342543DE82EF95ull : 0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += (long)gridDim.x * blockDim.x)
        P.mul[i] = u01(seed, i) >= P.p ? 1.f / (1.f - P.p) : 0.f;
}
extern "C" int chap_chan_mask(const chap_chanmask_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->mul, "chap_chan_mask: null argument");
    return chap_launch<chap_chanmask_params, chanmask_kernel, 256>(dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, "chap_chan_mask");
}

// ---- BCP box mixing (2D: box = {y0, x0, bh, bw}; 3D cuboid: box = {z0, y0, x0, bd, bh, bw}) ----------
__device__ __forceinline__ bool in_box(const int32_t* box, int D, int z, int y, int x) {
    if (D > 1) return z >= box[0] && z < box[0] + box[3] && y >= box[1] && y < box[1] + box[4] && x >= box[2] && x < box[2] + box[5];
    return y >= box[0] && y < box[0] + box[2] && x >= box[1] && x < box[1] + box[3];
}
template <typename T>
__global__ void boxmix_kernel(const T* a, const T* b, T* out, const int32_t* box, int N, int D, int H, int W) {
    const long total = (long)N * D * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const unsigned ui = (unsigned)i;
        const int x = (int)(ui % (unsigned)W), y = (int)((ui / (unsigned)W) % (unsigned)H), z = (int)((ui / (unsigned)(W * H)) % (unsigned)D);
        out[i] = in_box(box, D, z, y, x) ? b[i] : a[i];
    }
}
extern "C" int chap_box_mix(const chap_boxmix_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->a && p->b && p->out && p->box, "chap_box_mix: null argument");
    const int D = p->D > 1 ? p->D : 1;
    const long total = (long)p->N * D * p->H * p->W;
    const int nb = chap_blocks(total, 2048);
    if (p->is_i64) hipLaunchKernelGGL(boxmix_kernel<int64_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const int64_t*)p->a, (const int64_t*)p->b, (int64_t*)p->out, p->box, p->N, D, p->H, p->W);
    else hipLaunchKernelGGL(boxmix_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)p->a, (const float*)p->b, (float*)p->out, p->box, p->N, D, p->H, p->W);
    CHAP_LAUNCH_CHECK("chap_box_mix");
    return CHAP_OK;
}
__global__ void boxmask_kernel(const chap_boxmask_params P, int D) {
    const long total = (long)P.N * D * P.H * P.W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const unsigned ui = (unsigned)i;
        const int x = (int)(ui % (unsigned)P.W), y = (int)((ui / (unsigned)P.W) % (unsigned)P.H), z = (int)((ui / (unsigned)(P.W * P.H)) % (unsigned)D);
        P.mask[i] = in_box(P.box, D, z, y, x) ? 0 : 1;
    }
}
extern "C" int chap_box_mask(const chap_boxmask_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->mask && p->box, "chap_box_mask: null argument");
    const int D = p->D > 1 ? p->D : 1;
    const long total = (long)p->N * D * p->H * p->W;
    const int nb = chap_blocks(total, 2048);
    hipLaunchKernelGGL(boxmask_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *p, D);
    CHAP_LAUNCH_CHECK("chap_box_mask");
    return CHAP_OK;
}

// ---- spatial perturbation mask: avg-pool(scale) -> per-sample k-th largest -> mask --------------
__global__ void diffmask_pool_kernel(const chap_diffmask_params P) {
    const int PH = P.H / P.scale, PW = P.W / P.scale;
    const long total = (long)P.N * PH * PW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % PW), py = (int)((i / PW) % PH), n = (int)(i / ((long)PW * PH));
        float s = 0.f;
        for (int dy = 0; dy < P.scale; ++dy)
            for (int dx = 0; dx < P.scale; ++dx) s += P.knowledge[((long)n * P.H + py * P.scale + dy) * P.W + px * P.scale + dx];
        P.pooled_ws[i] = s / (float)(P.scale * P.scale);
    }
}
// one block per sample: radix-select (8 bits per pass, 256-bin LDS histogram) the k-th largest pooled value
// (non-negative floats order like their bit patterns), then a second kernel writes the mask.
// Round 4: one 256-bin histogram PER WAVE (16 KB of LDS), summed once per pass -- with one shared histogram the 1024 threads' LDS atomics of the
// later passes serialised on the bins (3D, 62 720 pooled values per sample, 2 samples = 2 blocks: 88 us on the iteration's critical chain).
__global__ __launch_bounds__(1024) void diffmask_select_kernel(const chap_diffmask_params P, float* thr_out) {
    __shared__ unsigned whist[16][256];
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix;
    __shared__ int s_rem;
    const int PH = P.H / P.scale, PW = P.W / P.scale, M = PH * PW;
    const int n = blockIdx.x;
    const float* v = P.pooled_ws + (long)n * M;
    int k = (int)(P.topk * (float)M);
    if (k < 1) k = 1;
    if (threadIdx.x == 0) { s_prefix = 0; s_rem = k; }
    unsigned* myh = whist[threadIdx.x >> 6];
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = shift == 24 ? 0u : ~((1u << (shift + 8)) - 1u);
        const int Mpad = (M + 63) & ~63;                       // whole waves go through the loop (ballots)
        // four values per thread and trip, their loads issued together: one dependent load per trip left the two blocks of a 3D batch
        // (62 720 pooled values per sample) at ~90 us, all of it load latency
        for (int i0 = threadIdx.x; i0 < Mpad; i0 += 4 * 1024) {
            float fv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int i = i0 + q * 1024; fv[q] = i < M ? v[i] : 0.f; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * 1024;
                if (i - (int)threadIdx.x >= Mpad) break;            // wave-uniform: the trip's later quarters may lie beyond the padded end
                float f = fv[q]; if (f < 0.f) f = 0.f;
                const unsigned u = __float_as_uint(f);
                const bool hit = i < M && (u & himask) == prefix;
                const unsigned bin = (u >> shift) & 255u;
                if (shift == 24) {
                    // first pass: sign + exponent bits -- nearly every value falls into one or two bins, i.e. one LDS word takes every
                    // increment of the block.  One increment per distinct bin of a wave instead (a leader adds the group's count).
                    unsigned long long todo = __ballot(hit);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const unsigned lb = __shfl(bin, leader, 64);
                        const unsigned long long same = __ballot(hit && bin == lb) & todo;
                        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&myh[lb], (unsigned)__popcll(same));
                        todo &= ~same;
                    }
                } else if (hit) atomicAdd(&myh[bin], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            unsigned t = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) t += whist[w][threadIdx.x];
            hist[threadIdx.x] = t;
        }
        __syncthreads();
        // the k-th largest lies in the bin t with  (values in bins above t) < rem <= (values in bins above t) + hist[t]: exactly one
        // of the 256 threads finds it (a single thread walking the bins cost 7 us per pass)
        const int rem0 = s_rem;
        __syncthreads();
        if (threadIdx.x < 256) {
            const int t = threadIdx.x;
            int above = 0;
            for (int b = t + 1; b < 256; ++b) above += (int)hist[b];
            if (above < rem0 && rem0 <= above + (int)hist[t]) { s_rem = rem0 - above; s_prefix = prefix | ((unsigned)t << shift); }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) thr_out[n] = __uint_as_float(s_prefix);
}
__global__ void diffmask_write_kernel(const chap_diffmask_params P, const float* thr) {
    const int PW = P.W / P.scale, M = (P.H / P.scale) * PW;
    const long total = (long)P.N * P.H * P.W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const unsigned ui = (unsigned)i;
        const int x = (int)(ui % (unsigned)P.W), y = (int)((ui / (unsigned)P.W) % (unsigned)P.H), n = (int)(ui / (unsigned)(P.W * P.H));
        float f = P.pooled_ws[(long)n * M + (y / P.scale) * PW + x / P.scale]; if (f < 0.f) f = 0.f;
        P.out[i] = (f >= thr[n] || P.p1[i] != P.p2[i]) ? 1.f : 0.f;
    }
}
extern "C" int chap_diff_mask(const chap_diffmask_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->p1 && p->p2 && p->knowledge && p->out && p->pooled_ws, "chap_diff_mask: null argument");
    CHAP_CHECK_ARG(p->scale > 0 && p->H % p->scale == 0 && p->W % p->scale == 0, "chap_diff_mask: H,W must be multiples of scale");
    const long total = (long)p->N * (p->H / p->scale) * (p->W / p->scale);
    hipLaunchKernelGGL(diffmask_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_diff_mask(pool)");
    float* thr = p->pooled_ws + total;          // N floats after the pooled map
    hipLaunchKernelGGL(diffmask_select_kernel, dim3(p->N), dim3(1024), 0, (hipStream_t)stream, *p, thr);
    const long npx = (long)p->N * p->H * p->W;
    hipLaunchKernelGGL(diffmask_write_kernel, dim3((unsigned)((npx + 255) / 256 < 2048 ? (npx + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, *p, (const float*)thr);
    CHAP_LAUNCH_CHECK("chap_diff_mask(select)");
    return CHAP_OK;
}

// ---- fused SGD over the flat parameter buffer ---------------------------------------------------
__global__ void sgd_kernel(const chap_sgd_params P) {
    const float lr = *P.lr;
    const long n4 = P.n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 p = ((float4*)P.param)[i], g = ((float4*)P.grad)[i], m = ((float4*)P.mom)[i];
        if (P.grad2) { const float4 h = ((float4*)P.grad2)[i]; g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
#define CHAP_SGD1(f) { const float gg = g.f * P.grad_scale + P.weight_decay * p.f; m.f = P.momentum * m.f + gg; p.f -= lr * m.f; }
        CHAP_SGD1(x) CHAP_SGD1(y) CHAP_SGD1(z) CHAP_SGD1(w)
        ((float4*)P.param)[i] = p; ((float4*)P.mom)[i] = m;
        if (P.zero_grad) {
            ((float4*)P.grad)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (P.grad2) ((float4*)P.grad2)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (blockIdx.x == 0) {
        for (long i = n4 * 4 + threadIdx.x; i < P.n; i += blockDim.x) {
            const float gr = P.grad[i] + (P.grad2 ? P.grad2[i] : 0.f);
            const float gg = gr * P.grad_scale + P.weight_decay * P.param[i];
            const float mm = P.momentum * P.mom[i] + gg;
            P.mom[i] = mm; P.param[i] -= lr * mm;
            if (P.zero_grad) { P.grad[i] = 0.f; if (P.grad2) P.grad2[i] = 0.f; }
        }
    }
}
extern "C" int chap_sgd_step(const chap_sgd_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->param && p->grad && p->mom && p->lr && p->n > 0, "chap_sgd_step: bad argument");
    CHAP_CHECK_ARG(((uintptr_t)p->param | (uintptr_t)p->grad | (uintptr_t)p->mom) % 16 == 0, "chap_sgd_step: buffers must be 16-byte aligned");
    const long n4 = p->n / 4;
    const int nb = chap_blocks(n4, 2048);
    hipLaunchKernelGGL(sgd_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_sgd_step");
    return CHAP_OK;
}

// ---- GradSim: per output channel, cosine similarity of two gradients of one conv kernel (rows of K contiguous floats) ----
__global__ __launch_bounds__(256) void gradsim_kernel(const chap_gradsim_params P) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= P.C) return;
    const float* a = P.gl + (long)c * P.K;
    const float* b = P.gu + (long)c * P.K;
    double dot = 0.0, na = 0.0, nb = 0.0;
    for (int k = lane; k < P.K; k += 64) { const double x = a[k], y = b[k]; dot += x * y; na += x * x; nb += y * y; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dot += __shfl_xor(dot, o, 64); na += __shfl_xor(na, o, 64); nb += __shfl_xor(nb, o, 64); }
    if (lane == 0) {
        const float sim = (float)(dot / (sqrt(na) * sqrt(nb) + 1e-12));
        P.score[c] = P.ema * P.score[c] + (1.f - P.ema) * sim;
    }
}
extern "C" int chap_grad_sim(const chap_gradsim_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->gl && p->gu && p->score && p->C > 0 && p->K > 0, "chap_grad_sim: bad argument");
    hipLaunchKernelGGL(gradsim_kernel, dim3((p->C + 3) / 4), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_grad_sim");
    return CHAP_OK;
}

// ---- bandwidth calibration (tools/membw.py): float4 copy ------------------------------------------
__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
extern "C" int chap_debug_copy(const void* src, void* dst, int64_t bytes, int32_t blocks, void* stream) {
    hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, (long)(bytes / 16));
    CHAP_LAUNCH_CHECK("chap_debug_copy");
    return CHAP_OK;
}
