// The arithmetic of the BatchNorm / activation backward, shared by chap_act_bwd_* (pointwise.hip) and by the weight gradient's "lazy
// gradient" path (wgrad_kernel.h, chap_bgrad_t): ONE definition, explicit fused multiply-adds, so that both produce the same bits.
//   a = keep*ks*cm*leaky(z), z = scale*raw + shift        ->  dz = dy * (z > 0 ? 1 : slope) [* keep*ks] [* cm]
//   g = gamma*invstd*(dz - S0/cnt - xhat*S1/cnt), xhat = (raw - mean)*invstd   ==   dz*k0 + raw*cB + cC
#pragma once
#include "common.h"

typedef float actbwd_f32x2 __attribute__((ext_vector_type(2)));

// per-channel constants of 8 channels from the totals S0 = sum dz, S1 = sum dz*xhat
__device__ __forceinline__ void actbwd_consts8(const float a0[8], const float a1[8], const float gm[8], const float mean[8], const float istd[8], float count,
                                               float k0[8], float cB[8], float cC[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        k0[j] = gm[j] * istd[j];
        const float k1 = a0[j] / count, k2 = a1[j] / count;
        cB[j] = -istd[j] * k0[j] * k2;
        cC[j] = __builtin_fmaf(mean[j] * istd[j] * k0[j], k2, -k0[j] * k1);
    }
}

// dz = dy * d(activation)/dz of 8 channels (packed fp32: v_pk_fma_f32 / v_pk_mul_f32)
__device__ __forceinline__ void actbwd_deriv8(const float raw[8], const float dy[8], const float sa[8], const float sb[8], bool act, float slope, float dz[8]) {
    if (act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const actbwd_f32x2 r2 = {raw[2 * j], raw[2 * j + 1]}, a2 = {sa[2 * j], sa[2 * j + 1]}, b2 = {sb[2 * j], sb[2 * j + 1]};
            const actbwd_f32x2 z = __builtin_elementwise_fma(r2, a2, b2);
            const actbwd_f32x2 f = {z.x > 0.f ? 1.f : slope, z.y > 0.f ? 1.f : slope};
            const actbwd_f32x2 g2 = {dy[2 * j], dy[2 * j + 1]};
            const actbwd_f32x2 d = g2 * f;
            dz[2 * j] = d.x; dz[2 * j + 1] = d.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] = dy[j];
    }
}
__device__ __forceinline__ void actbwd_keep8(float dz[8], uint2 m, float keep_scale) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t w = j < 4 ? m.x : m.y;
        dz[j] = ((w >> (8 * (j & 3))) & 0xff) ? dz[j] * keep_scale : 0.f;
    }
}

// g = dz*k0 + (raw*cB + cC)
__device__ __forceinline__ void actbwd_apply8(const float dz[8], const float raw[8], const float k0[8], const float cB[8], const float cC[8], float o[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const actbwd_f32x2 d = {dz[2 * j], dz[2 * j + 1]}, r2 = {raw[2 * j], raw[2 * j + 1]};
        const actbwd_f32x2 A = {k0[2 * j], k0[2 * j + 1]}, B = {cB[2 * j], cB[2 * j + 1]}, Cc = {cC[2 * j], cC[2 * j + 1]};
        const actbwd_f32x2 r = __builtin_elementwise_fma(d, A, __builtin_elementwise_fma(r2, B, Cc));
        o[2 * j] = r.x; o[2 * j + 1] = r.y;
    }
}
