// Small training-loop kernels: VAT perturbation helpers, counter-based RNG, BCP box mixing,
// spatial perturbation mask (avg-pool + per-sample top-k by radix select), fused SGD.
#include "common.h"

// ---- per-sample L2 normalise: one 1024-thread block per sample --------------------------------
__global__ __launch_bounds__(1024) void l2norm_kernel(const chap_l2norm_params P) {
    __shared__ float red[16];
    const float* x = P.in + (long)blockIdx.x * P.P;
    float* o = P.out + (long)blockIdx.x * P.P;
    float s = 0.f;
    for (int i = threadIdx.x; i < P.P; i += 1024) { const float v = x[i]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    const float inv = 1.f / (sqrtf(t) + P.eps);
    for (int i = threadIdx.x; i < P.P; i += 1024) o[i] = x[i] * inv;
}
extern "C" int chap_l2_normalize(const chap_l2norm_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->in && p->out && p->N > 0 && p->P > 0, "chap_l2_normalize: bad argument");
    hipLaunchKernelGGL(l2norm_kernel, dim3(p->N), dim3(1024), 0, (hipStream_t)stream, *p);
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
    const int nb = (int)((p->n + 255) / 256 < 2048 ? (p->n + 255) / 256 : 2048);
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
    const int nb = (int)((p->n + 255) / 256 < 2048 ? (p->n + 255) / 256 : 2048);
    hipLaunchKernelGGL(rand_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_rand_uniform");
    return CHAP_OK;
}
__global__ void keepmask_kernel(const chap_keepmask_params P) {
    const uint64_t seed = P.seed + (P.seed_dev ? *P.seed_dev * 0xD1342543DE82EF95ull : 0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += (long)gridDim.x * blockDim.x)
        P.keep[i] = u01(seed, i) >= P.p ? 1 : 0;
}
extern "C" int chap_keep_mask(const chap_keepmask_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->keep, "chap_keep_mask: null argument");
    const int nb = (int)((p->n + 255) / 256 < 4096 ? (p->n + 255) / 256 : 4096);
    hipLaunchKernelGGL(keepmask_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_keep_mask");
    return CHAP_OK;
}
__global__ void chanmask_kernel(const chap_chanmask_params P) {
    const uint64_t seed = P.seed + (P.seed_dev ? *P.seed_dev * 0xD1342543DE82EF95ull : 0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += (long)gridDim.x * blockDim.x)
        P.mul[i] = u01(seed, i) >= P.p ? 1.f / (1.f - P.p) : 0.f;
}
extern "C" int chap_chan_mask(const chap_chanmask_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->mul, "chap_chan_mask: null argument");
    hipLaunchKernelGGL(chanmask_kernel, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_chan_mask");
    return CHAP_OK;
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
    const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
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
    const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
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
// one block per sample: radix-select the k-th largest pooled value (non-negative floats order like
// their bit patterns), then write the mask.
__global__ __launch_bounds__(1024) void diffmask_select_kernel(const chap_diffmask_params P) {
    __shared__ unsigned cnt, s_prefix;
    __shared__ int s_rem;
    const int PH = P.H / P.scale, PW = P.W / P.scale, M = PH * PW;
    const int n = blockIdx.x;
    const float* v = P.pooled_ws + (long)n * M;
    int k = (int)(P.topk * (float)M);
    if (k < 1) k = 1;
    if (threadIdx.x == 0) { s_prefix = 0; s_rem = k; }
    __syncthreads();
    for (int bit = 31; bit >= 0; --bit) {
        if (threadIdx.x == 0) cnt = 0;
        __syncthreads();
        const unsigned want = s_prefix | (1u << bit);
        const unsigned maskhi = ~((1u << bit) - 1u);
        unsigned c = 0;
        for (int i = threadIdx.x; i < M; i += 1024) {
            float f = v[i]; if (f < 0.f) f = 0.f;
            if ((__float_as_uint(f) & maskhi) == want) ++c;
        }
        if (c) atomicAdd(&cnt, c);
        __syncthreads();
        if (threadIdx.x == 0) { if ((int)cnt >= s_rem) s_prefix = want; else s_rem -= (int)cnt; }
        __syncthreads();
    }
    const unsigned prefix = s_prefix;
    const float thr = __uint_as_float(prefix);
    const long base = (long)n * P.H * P.W;
    for (int i = threadIdx.x; i < P.H * P.W; i += 1024) {
        const int y = i / P.W, x = i % P.W;
        float f = v[(y / P.scale) * PW + x / P.scale]; if (f < 0.f) f = 0.f;
        const bool sel = f >= thr;
        const bool diff = P.p1[base + i] != P.p2[base + i];
        P.out[base + i] = (sel || diff) ? 1.f : 0.f;
    }
}
extern "C" int chap_diff_mask(const chap_diffmask_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->p1 && p->p2 && p->knowledge && p->out && p->pooled_ws, "chap_diff_mask: null argument");
    CHAP_CHECK_ARG(p->scale > 0 && p->H % p->scale == 0 && p->W % p->scale == 0, "chap_diff_mask: H,W must be multiples of scale");
    const long total = (long)p->N * (p->H / p->scale) * (p->W / p->scale);
    hipLaunchKernelGGL(diffmask_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_diff_mask(pool)");
    hipLaunchKernelGGL(diffmask_select_kernel, dim3(p->N), dim3(1024), 0, (hipStream_t)stream, *p);
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
    const int nb = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(sgd_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_sgd_step");
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
