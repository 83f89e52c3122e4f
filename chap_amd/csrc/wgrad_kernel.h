// Weight gradient as an implicit GEMM with the PIXEL index as the reduction dimension:
//     dW[tap][kc][kn] = sum_p  A[s*p + tap - pad][kc] * B[p][kn]
// MFMA roles: rows = kc (16 input channels of the halo-tiled operand A), cols = kn (16 channels of
// B), k = pixels.  Both operands are channel-last, i.e. k runs along the SLOW memory axis, so
//   * bf16: fragments come from `ds_read_b64_tr_b16` (gfx950 transposing LDS read: a 4-pixel x
//           16-channel block is delivered channel-major, two reads = the 8 k-values of a lane);
//   * fp32: 16x16x4 MFMA takes one k-value per lane -> plain ds_read_b32 of [pixel 4s+g][channel].
// A block owns one (KC-channel chunk of A) x (32-channel chunk of B) pair and one pixel split; it
// walks its TH x 16 pixel tiles keeping dW in registers (each wave owns a subset of (tap, 16-row)
// pairs), then writes ONE fp32 partial slab.  A second kernel sums the slabs in fixed order
// (deterministic, no float atomics) and accumulates into the checkpoint-layout gradient.
#pragma once
#include "conv_kernel.h"

constexpr int WG_BN = 32;                       // B channels per block
template <typename T> __host__ __device__ constexpr int wg_psb() { return sizeof(T) == 2 ? 40 : 36; }

template <typename T, int KS, int ST, bool D3, int KC, int MR>
__host__ __device__ constexpr size_t wgrad_lds_bytes() {
    typedef conv_geom<KS, ST, D3, MR> G;
    return ((size_t)G::HP * pix_stride<T, KC>() + (size_t)G::TH * G::TW * wg_psb<T>()) * sizeof(T);
}

__device__ __forceinline__ s16x4 lds_tr16(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

template <typename T, int KS, int ST, bool D3, int KC, int MR>
__global__ __launch_bounds__(256) void wgrad_kernel(const chap_wgrad_params P, float* __restrict__ ws, float* __restrict__ ws_db,
                                                    int nsplit, int Ca, int Cb) {
    typedef conv_geom<KS, ST, D3, MR> G;
    constexpr int PS = pix_stride<T, KC>(), PSB = wg_psb<T>();
    constexpr int KCT = KC / 16, NTB = WG_BN / 16;
    constexpr int PAIRS = G::NTAPS * KCT, MAXP = (PAIRS + 3) / 4;
    constexpr int NKCH = G::TH * G::TW / 32;            // 32-pixel k-chunks per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo = (T*)smem;
    T* btile = halo + (size_t)G::HP * PS;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int split = blockIdx.x, chunk = blockIdx.y, nb = blockIdx.z;
    const int tiles_x = (P.W + G::TW - 1) / G::TW, tiles_y = (P.H + G::TH - 1) / G::TH;
    const long ntiles = (long)P.N * P.D * tiles_y * tiles_x;

    f32x4 acc[MAXP][NTB];
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
#pragma unroll
        for (int t = 0; t < NTB; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbacc = 0.f;

    for (long tile = split; tile < ntiles; tile += nsplit) {
        long b = tile;
        const int tx = (int)(b % tiles_x); b /= tiles_x;
        const int ty = (int)(b % tiles_y); b /= tiles_y;
        const int z0 = (int)(b % P.D);     const int n = (int)(b / P.D);
        const int x0 = tx * G::TW, y0 = ty * G::TH;
        __syncthreads();                                   // previous tile's LDS reads are done
        stage_halo<T, KS, ST, D3, KC, MR>(halo, P.a[0], P.a[1], P.na, P.combine, n, z0, y0, x0, P.ID, P.IH, P.IW, chunk);
        for (int u = threadIdx.x; u < G::TH * G::TW * (WG_BN / 8); u += 256) {
            const int pix = u / (WG_BN / 8), c8 = (u % (WG_BN / 8)) * 8;
            const int yy = y0 + pix / G::TW, xx = x0 + pix % G::TW;
            float v[8];
            const int cb = nb * WG_BN + c8;
            if (yy < P.H && xx < P.W && cb < Cb) {
                const long gp = (((long)n * P.D + z0) * P.H + yy) * P.W + xx;
                src_load8<T>(P.b, n, gp, cb, v);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            st8(btile + pix * PSB + c8, v);
        }
        __syncthreads();
        if (ws_db != nullptr && chunk == 0 && threadIdx.x < WG_BN) {
            float s = 0.f;
            for (int pix = 0; pix < G::TH * G::TW; ++pix) s += elem<T>::get(btile[pix * PSB + threadIdx.x]);
            dbacc += s;
        }
#pragma unroll 1
        for (int kc = 0; kc < NKCH; ++kc) {
            if constexpr (sizeof(T) == 2) {
                // lane group g covers pixels 8g..8g+7 of the 32-pixel chunk: row = 2*kc + (g>>1), x = 8*(g&1) + 0..7
                const int q = lane & 15, qq = q >> 2, pp = q & 3;      // this lane supplies pixel +qq, channels 4*pp..
                const int row = 2 * kc + (g >> 1), xb = 8 * (g & 1);
                uint4 bf[NTB];
#pragma unroll
                for (int t = 0; t < NTB; ++t) {
                    const bf16_t* bp = (const bf16_t*)btile + (row * G::TW + xb + qq) * PSB + t * 16 + 4 * pp;
                    const s16x4 lo = lds_tr16(bp), hi = lds_tr16(bp + 4 * PSB);
                    bf[t] = __builtin_bit_cast(uint4, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
                }
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    const int pair = wave + 4 * i;
                    if (pair < PAIRS) {
                        const int tap = pair / KCT, kct = pair % KCT;
                        const int dx = tap % KS, dy = (tap / KS) % KS, dz = tap / (KS * KS);
                        const bf16_t* ap = (const bf16_t*)halo + ((dz * G::HH + row * ST + dy) * G::HW + (xb + qq) * ST + dx) * PS + kct * 16 + 4 * pp;
                        const s16x4 lo = lds_tr16(ap), hi = lds_tr16(ap + 4 * ST * PS);
                        const uint4 af = __builtin_bit_cast(uint4, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
#pragma unroll
                        for (int t = 0; t < NTB; ++t) mma8(acc[i][t], af, bf[t]);
                    }
                }
            } else {
                // fp32: 8 MFMAs of 4 pixels each; lane (channel l15, pixel 4*sub + g)
#pragma unroll 2
                for (int sub = 0; sub < 8; ++sub) {
                    const int q = kc * 32 + sub * 4 + g;
                    const int row = q / G::TW, xx = q % G::TW;
                    float bfr[NTB];
#pragma unroll
                    for (int t = 0; t < NTB; ++t) bfr[t] = ((const float*)btile)[q * PSB + t * 16 + l15];
#pragma unroll
                    for (int i = 0; i < MAXP; ++i) {
                        const int pair = wave + 4 * i;
                        if (pair < PAIRS) {
                            const int tap = pair / KCT, kct = pair % KCT;
                            const int dx = tap % KS, dy = (tap / KS) % KS, dz = tap / (KS * KS);
                            const float a = ((const float*)halo)[((dz * G::HH + row * ST + dy) * G::HW + xx * ST + dx) * PS + kct * 16 + l15];
#pragma unroll
                            for (int t = 0; t < NTB; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfr[t], acc[i][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- partial slab: ws[split][tap][kc_global][kn_global]; lane holds rows 4g+j (kc), col l15 (kn) ----
    float* slab = ws + (long)split * G::NTAPS * Ca * Cb;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int pair = wave + 4 * i;
        if (pair < PAIRS) {
            const int tap = pair / KCT, kct = pair % KCT;
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                const int kn = nb * WG_BN + t * 16 + l15;
                if (kn < Cb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kcg = chunk * KC + kct * 16 + 4 * g + j;
                        slab[((long)tap * Ca + kcg) * Cb + kn] = acc[i][t][j];
                    }
                }
            }
        }
    }
    if (ws_db != nullptr && chunk == 0 && threadIdx.x < WG_BN && nb * WG_BN + threadIdx.x < Cb)
        ws_db[(long)split * Cb + nb * WG_BN + threadIdx.x] = dbacc;
}
