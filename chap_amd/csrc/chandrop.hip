// Channel-level perturbation masks: the device side of FilterDropout.perform_dropout
// (code/networks/FilterDropout.py:45-89), scores_dropoutV2 (:116-138) and drop_based_on_prob (:140-160).
//
//   chap_sample_channel_sum   per-sample spatial sums of a lazy activation (adaptive_avg_pool2d(unlab_feat, 1), :75),
//                             as fixed-order partial sums so that the masks do not depend on atomic ordering
//   chap_channel_drop         the two per-(sample, channel) multipliers of one encoder level, written as the
//                             chan_mul rows of the (B + U)-sample decoder batch  torch.cat((feat, perturb_feat))  (:86-87)
//   chap_fold_perturbed       the adjoint of that cat: gradient of the (B + U)-sample batch -> gradient of the B samples
#include "common.h"
#include "launch.h"

template <typename T>
__global__ __launch_bounds__(256) void sample_channel_sum_kernel(const chap_sample_chansum_params P) {
    extern __shared__ float red[];                       // [PPB][C]
    const int C = P.r.C, C8 = C / 8, PPB = 256 / C8;
    const int c8 = (threadIdx.x % C8) * 8, prow = threadIdx.x / C8;
    const int n = blockIdx.y;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (prow < PPB) {
        const long base = (long)n * P.pix_per_sample;
        for (long pp = (long)blockIdx.x * PPB + prow; pp < P.pix_per_sample; pp += (long)gridDim.x * PPB) {
            float v[8];
            src_load8<T>(P.r, n, base + pp, c8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[prow * C + c8 + j] = s[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int r = 0; r < PPB; ++r) a += red[r * C + c];
        P.partial[((long)n * gridDim.x + blockIdx.x) * C + c] = a;
    }
}

extern "C" int chap_sample_channel_sum(const chap_sample_chansum_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->r.ptr && p->partial && p->N > 0 && p->nchunk > 0 && p->pix_per_sample > 0, "chap_sample_channel_sum: bad argument");
    CHAP_CHECK_ARG(p->r.C % 8 == 0 && p->r.C <= 2048 && 256 % (p->r.C / 8) == 0, "chap_sample_channel_sum: C = %d must be 8 * (a power of two <= 256)", p->r.C);
    CHAP_CHECK_ARG(p->N <= 65535, "chap_sample_channel_sum: too many samples");
    const size_t lds = (size_t)(256 / (p->r.C / 8)) * p->r.C * sizeof(float);    // = 8 KiB
    if (p->dtype == CHAP_BF16) hipLaunchKernelGGL(sample_channel_sum_kernel<bf16_t>, dim3(p->nchunk, p->N), dim3(256), lds, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(sample_channel_sum_kernel<float>, dim3(p->nchunk, p->N), dim3(256), lds, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_sample_channel_sum");
    return CHAP_OK;
}

// One block: U * C <= 16384 scores live in LDS.
__device__ __forceinline__ float block_sum(float v, float* scratch) {       // all 256 threads; result in every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__global__ __launch_bounds__(256) void channel_drop_kernel(const chap_channel_drop_params P) {
    extern __shared__ float sc[];                        // [U][C] scores -> drop probabilities
    __shared__ float scratch[4];
    const int U = P.U, C = P.C, UC = U * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int mode = P.mode;
    if (mode == 2) {
        // torch.all(scores[idx].eq(0)) -> Dropout2d pair (FilterDropout.py:71-73), decided on the device
        float nz = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) nz += (P.grad_sim[c] != 0.f) ? 1.f : 0.f;
        if (block_sum(nz, scratch) == 0.f) mode = 0;
    }
    if (mode == 2) {
        for (int i = threadIdx.x; i < UC; i += 256) {
            const int u = i / C, c = i - u * C;
            float a = 0.f;
            for (int k = 0; k < P.nchunk; ++k) a += P.pool_partial[((long)u * P.nchunk + k) * C + c];
            sc[i] = P.grad_sim[c] * (a * P.inv_npix);    // grad_sim.unsqueeze(0) * activation  (:122)
        }
        __syncthreads();
        for (int u = wave; u < U; u += 4) {              // one wave per sample row: mean, unbiased std (:124-125)
            float s = 0.f;
            for (int c = lane; c < C; c += 64) s += sc[u * C + c];
            const float mean = wave_sum(s) / (float)C;
            float q = 0.f;
            for (int c = lane; c < C; c += 64) { const float d = sc[u * C + c] - mean; q += d * d; }
            const float sigma = sqrtf(wave_sum(q) / (float)(C - 1));
            for (int c = lane; c < C; c += 64) {
                const float v = sc[u * C + c];
                float pr;
                if (P.prob_kind == 1) {                  // 'gauss' (:126-130)
                    const float z = (v - mean) / (sigma * 2.0f + 1e-8f);
                    pr = fminf(fmaxf(0.5f * (1.f + erff(z * 0.70710678118654752f)), 0.f), 1.f);
                } else {                                 // 'sigmoid', temp = 2 (:131-134)
                    const float z = (v - mean) / (sigma + 1e-8f);
                    pr = 1.f / (1.f + expf(2.0f * z));
                }
                sc[u * C + c] = pr;
                if (P.probs_out) P.probs_out[u * C + c] = pr;
            }
        }
        __syncthreads();
    }
    // ---- the two masks; bernoulli(q) == (u < q) on the supplied uniforms
    float n1 = 0.f, n2 = 0.f;
    for (int i = threadIdx.x; i < UC; i += 256) {
        const float u1 = P.u1[i], u2 = P.u2[i];
        float m1, m2;
        if (mode == 0) {                                 // two independent nn.Dropout2d(0.5)  (:67-69)
            m1 = u1 < 0.5f ? 2.f : 0.f; m2 = u2 < 0.5f ? 2.f : 0.f;
        } else if (mode == 1) {                          // complementary Binomial(0.5) * 2  (:58-63)
            m1 = u1 < 0.5f ? 2.f : 0.f; m2 = 2.f - m1;
        } else {                                         // drop_based_on_prob (:140-153)
            const float pd = sc[i], pk = 1.f - pd;
            const float q1 = (P.comp && P.branch == 1) ? pd : pk;
            const float q2 = (P.comp && P.branch == 0) ? pd : pk;
            m1 = u1 < q1 ? 1.f : 0.f; m2 = u2 < q2 ? 1.f : 0.f;
            n1 += m1; n2 += m2;
        }
        sc[i] = m1;                                      // m1 staged in LDS, m2 in the output (rescaled below)
        P.mul2[(long)(P.B + i / C) * C + (i % C)] = m2;
    }
    n1 = block_sum(n1, scratch);
    n2 = block_sum(n2, scratch);
    __syncthreads();
    for (int i = threadIdx.x; i < UC; i += 256) {
        const long o = (long)(P.B + i / C) * C + (i % C);
        float m1 = sc[i], m2 = P.mul2[o];
        if (mode == 2) {                                 // mask * numel / sum  (:157-158); an empty mask is NaN there too
            m1 = (m1 * (float)UC) / n1;
            m2 = (m2 * (float)UC) / n2;
        }
        P.mul1[o] = m1; P.mul2[o] = m2;
    }
    for (int i = threadIdx.x; i < P.B * C; i += 256) { P.mul1[i] = 1.f; P.mul2[i] = 1.f; }   // the unperturbed rows of the cat
}

extern "C" int chap_channel_drop(const chap_channel_drop_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->u1 && p->u2 && p->mul1 && p->mul2 && p->B >= 0 && p->U > 0 && p->C > 1, "chap_channel_drop: bad argument");
    CHAP_CHECK_ARG(p->mode >= 0 && p->mode <= 2, "chap_channel_drop: mode %d", p->mode);
    CHAP_CHECK_ARG((long)p->U * p->C <= 16384, "chap_channel_drop: U * C = %ld exceeds the 16384 scores one block holds", (long)p->U * p->C);
    CHAP_CHECK_ARG(p->mode != 2 || (p->grad_sim && p->pool_partial && p->nchunk > 0), "chap_channel_drop: score-driven mode needs grad_sim and the pooled sums");
    const size_t lds = (size_t)p->U * p->C * sizeof(float);
    hipLaunchKernelGGL(channel_drop_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, *p);
    CHAP_LAUNCH_CHECK("chap_channel_drop");
    return CHAP_OK;
}

// out[n] = g[n] (n < B);  out[B - U + u] += mul[B + u] * g[B + u]   -- channel slice [coff, coff + C) of g (row length ld)
template <typename T>
__device__ __forceinline__ void fold_perturbed_kernel(const chap_fold_params& P) {
    const int C8 = P.C / 8;
    const long per_sample = P.pix_per_sample * C8;
    const long total = (long)P.B * per_sample;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / per_sample);
        const long r = i - (long)n * per_sample;
        const long pp = r / C8;
        const int c8 = (int)(r - pp * C8) * 8;
        float v[8];
        ld8((const T*)P.g + ((long)n * P.pix_per_sample + pp) * P.ld + P.coff + c8, v);
        const int u = n - (P.B - P.U);
        if (u >= 0) {
            float w[8];
            ld8((const T*)P.g + ((long)(P.B + u) * P.pix_per_sample + pp) * P.ld + P.coff + c8, w);
            if (P.mul) {
                float m[8];
                ld8(P.mul + (long)(P.B + u) * P.C + c8, m);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(w[j], m[j], v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += w[j];
            }
        }
        st8((T*)P.out + ((long)n * P.pix_per_sample + pp) * P.C + c8, v);
    }
}

extern "C" int chap_fold_perturbed(const chap_fold_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->g && p->out && p->B > 0 && p->U >= 0 && p->U <= p->B && p->pix_per_sample > 0, "chap_fold_perturbed: bad argument");
    CHAP_CHECK_ARG(p->C % 8 == 0 && p->ld % 8 == 0 && p->coff % 8 == 0 && p->coff + p->C <= p->ld, "chap_fold_perturbed: C=%d ld=%d coff=%d must be multiples of 8", p->C, p->ld, p->coff);
    const long total = (long)p->B * p->pix_per_sample * (p->C / 8);
    const int blocks = chap_blocks(total, 2048);
    if (p->dtype == CHAP_BF16) return chap_launch<chap_fold_params, fold_perturbed_kernel<bf16_t>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_fold_perturbed");
    return chap_launch<chap_fold_params, fold_perturbed_kernel<float>, 256>(dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, "chap_fold_perturbed");
}
