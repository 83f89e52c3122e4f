// Shared device helpers for libchap_hip.so (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "chap_hip.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short s16x8;   // 8 x bf16 MFMA operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void chap_set_error(const char* fmt, ...);
#define CHAP_CHECK_ARG(cond, ...) do { if (!(cond)) { chap_set_error(__VA_ARGS__); return CHAP_EINVAL; } } while (0)
bool chap_group_recording();      // launch.h: true between chap_group_begin() and chap_group_end()
// after a DIRECT launch (kernels that are not behind launch.h's trampoline): inside a group region it would have overtaken the
// recorded launches, which is an error of the caller
#ifdef CHAP_TIMELINE      // lab builds only (launch.h): a one-thread marker kernel behind every direct launch stamps the time its stream got past it
void chap_timeline_mark(const char* name, void* stream);
#define CHAP_TL_MARK(name) chap_timeline_mark(name, (void*)stream)
#else
#define CHAP_TL_MARK(name) do {} while (0)
#endif
#define CHAP_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    chap_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); return CHAP_ELAUNCH; } \
    if (chap_group_recording()) { chap_set_error("%s: not allowed between chap_group_begin() and chap_group_end()", name); return CHAP_EUNSUPPORTED; } \
    CHAP_TL_MARK(name); } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(bf16_t, b);
}

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), result in every lane: four v_add_f32 with a DPP source
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) -- the same pairings, hence the same rounding, as an xor
// butterfly 1-2-4-8, without the ds_bpermute + address arithmetic of __shfl_xor.
#define CHAP_DPP_ADD(v, ctrl) (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float row16_sum(float v) {
    CHAP_DPP_ADD(v, 0xB1);      // quad_perm [1,0,3,2]
    CHAP_DPP_ADD(v, 0x4E);      // quad_perm [2,3,0,1]
    CHAP_DPP_ADD(v, 0x141);     // row_half_mirror
    CHAP_DPP_ADD(v, 0x140);     // row_mirror
    return v;
}

// Sum over the lanes l, l+S, l+2S, ... of a DPP row (S = 2, 4 or 8; S >= 16: no-op) by cyclic row rotations.
template <int S> __device__ __forceinline__ float row16_stride_sum(float v) {
    if (S <= 2) CHAP_DPP_ADD(v, 0x122);     // row_ror:2
    if (S <= 4) CHAP_DPP_ADD(v, 0x124);     // row_ror:4
    if (S <= 8) CHAP_DPP_ADD(v, 0x128);     // row_ror:8
    return v;
}

template <typename T> struct elem;
template <> struct elem<float> {
    static __device__ __forceinline__ float get(float v) { return v; }
    static __device__ __forceinline__ float put(float v) { return v; }
};
template <> struct elem<bf16_t> {
    static __device__ __forceinline__ float get(bf16_t v) { return bf2f(v); }
    static __device__ __forceinline__ bf16_t put(float v) { return f2bf(v); }
};

// 8 consecutive elements <-> 8 floats (16-B aligned for bf16, 32-B for fp32)
__device__ __forceinline__ void ld8(const float* p, float v[8]) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float v[8]) {
    uint4 a = *(const uint4*)p;
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void st8(float* p, const float v[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
typedef float f32x2_native __attribute__((ext_vector_type(2)));
// two floats -> one dword of bf16 with ONE v_cvt_pk_bf16_f32 (RNE); converting them one by one costs a cvt each plus
// and/shift/or to merge, and the staging / store paths are VALU-issue bound
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2_native v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_native));
}
__device__ __forceinline__ void st8(bf16_t* p, const float v[8]) {
    *(uint4*)p = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
__device__ __forceinline__ void st4(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void st4(bf16_t* p, const float v[4]) { *(uint2*)p = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3])); }
__device__ __forceinline__ void ld4(const float* p, float v[4]) { float4 a = *(const float4*)p; v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
__device__ __forceinline__ void ld4(const bf16_t* p, float v[4]) {
    uint2 a = *(const uint2*)p;
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}

// Load 8 channels [c8, c8+8) of pixel `pix` (global pixel index, sample `n`) of a lazy activation.
template <typename T>
__device__ __forceinline__ void src_load8(const chap_src_t& s, int n, long pix, int c8, float v[8]) {
    ld8((const T*)s.ptr + pix * s.ld + s.coff + c8, v);
    if (s.scale) {
        float a[8], b[8];
        ld8(s.scale + c8, a); ld8(s.shift + c8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a[j], b[j]);
    }
    if (s.act) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * s.slope;
    }
    if (s.keep) {
        uint2 m = *(const uint2*)(s.keep + pix * s.C + c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t w = j < 4 ? m.x : m.y;
            v[j] = ((w >> (8 * (j & 3))) & 0xff) ? v[j] * s.keep_scale : 0.f;
        }
    }
    if (s.chan_mul) {
        float a[8];
        ld8(s.chan_mul + (long)n * s.C + c8, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= a[j];
    }
}

// Scalar version (one channel) for element-wise kernels.
template <typename T>
__device__ __forceinline__ float src_load1(const chap_src_t& s, int n, long pix, int c) {
    float v = elem<T>::get(((const T*)s.ptr)[pix * s.ld + s.coff + c]);
    if (s.scale) v = fmaf(v, s.scale[c], s.shift[c]);
    if (s.act) v = v > 0.f ? v : v * s.slope;
    if (s.keep) v = s.keep[pix * s.C + c] ? v * s.keep_scale : 0.f;
    if (s.chan_mul) v *= s.chan_mul[(long)n * s.C + c];
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Grid of a grid-stride streaming kernel: ceil(units / 256) blocks, capped.  CHAP_GRID_SCALE (lab knob, per cent) scales every cap.
static inline int chap_blocks(long units, long cap) {
    static int scale = -1;
    if (scale < 0) { const char* e = getenv("CHAP_GRID_SCALE"); scale = (e && atoi(e) > 0) ? atoi(e) : 100; }
    long c = cap * scale / 100;
    if (c < 64) c = 64;
    const long b = (units + 255) / 256;
    return (int)(b < c ? (b > 0 ? b : 1) : c);
}
