// chap_wgrad: split selection, workspace sizing, launch, deterministic slab reduction.
#include "common.h"

int chap_wgrad_launch_bf16(const chap_wgrad_params* p, int KC, float* ws, float* ws_db, int nsplit, int Ca, int Cb, hipStream_t s);
int chap_wgrad_launch_f32(const chap_wgrad_params* p, int KC, float* ws, float* ws_db, int nsplit, int Ca, int Cb, hipStream_t s);

struct wg_plan { int Ca, Cb, KC, taps, nsplit; long ntiles; size_t slab, bytes; };

static int wg_make_plan(const chap_wgrad_params* p, wg_plan* q) {
    CHAP_CHECK_ARG(p->na == 1 || p->na == 2, "chap_wgrad: na=%d", p->na);
    q->Ca = p->combine == 0 ? p->a[0].C + (p->na > 1 ? p->a[1].C : 0) : p->a[0].C;
    q->Cb = p->b.C;
    CHAP_CHECK_ARG(q->Ca % 16 == 0 && q->Cb % 8 == 0, "chap_wgrad: Ca=%d must be a multiple of 16, Cb=%d of 8", q->Ca, q->Cb);
    q->KC = q->Ca >= 32 ? 32 : 16;
    CHAP_CHECK_ARG(q->Ca % q->KC == 0, "chap_wgrad: Ca=%d not a multiple of %d", q->Ca, q->KC);
    q->taps = p->ksize * p->ksize * (p->dims == 3 ? p->ksize : 1);
    const bool small_tile = (p->ksize == 2 && p->dims == 3);
    const int TH = small_tile ? 4 : 8;
    q->ntiles = (long)p->N * p->D * cdiv(p->H, TH) * cdiv(p->W, 16);
    const long pairs = (long)(q->Ca / q->KC) * cdiv(q->Cb, 32);
    long ns = 1024 / pairs;
    if (ns < 1) ns = 1;
    if (ns > q->ntiles) ns = q->ntiles;
    q->slab = (size_t)q->taps * q->Ca * q->Cb * sizeof(float);
    const size_t cap = (size_t)96 << 20;
    while (ns > 1 && (size_t)ns * q->slab > cap) ns /= 2;
    q->nsplit = (int)ns;
    q->bytes = (size_t)q->nsplit * (q->slab + (size_t)q->Cb * sizeof(float));
    return CHAP_OK;
}

extern "C" size_t chap_wgrad_ws(const chap_wgrad_params* p) {
    wg_plan q;
    if (!p || wg_make_plan(p, &q)) return 0;
    return q.bytes;
}

// Deterministic slab reduction: a block owns 16 consecutive dW elements and splits the slabs over
// 16 thread groups (fixed order inside a group, fixed tree across groups), so the result does not
// depend on scheduling.  Loads are 64-byte segments of each slab; 16*ceil(total/16) threads.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ ws_db, int nsplit, int taps, int Ca, int Cb,
                                                           float* dw, long s_tap, long s_kc, long s_kn, int kc_valid, int kn_valid, float* db) {
    __shared__ float red[16][17];
    const long total = (long)taps * Ca * Cb;
    const int e = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + e;
    float s = 0.f;
    if (i < total)
        for (int k = sg; k < nsplit; k += 16) s += ws[(long)k * total + i];
    red[sg][e] = s;
    __syncthreads();
    if (sg == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][e];
        const int kn = (int)(i % Cb); const long r = i / Cb;
        const int kc = (int)(r % Ca); const int tap = (int)(r / Ca);
        if (kc < kc_valid && kn < kn_valid) dw[tap * s_tap + kc * s_kc + kn * s_kn] += t;
    }
    if (db != nullptr && blockIdx.x == 0) {
        for (int c = threadIdx.x; c < Cb && c < kn_valid; c += 256) {
            float t = 0.f;
            for (int k = 0; k < nsplit; ++k) t += ws_db[(long)k * Cb + c];
            db[c] += t;
        }
    }
}

extern "C" int chap_wgrad(const chap_wgrad_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->dw && p->ws && p->b.ptr && p->a[0].ptr, "chap_wgrad: null argument");
    wg_plan q;
    int r = wg_make_plan(p, &q);
    if (r) return r;
    CHAP_CHECK_ARG(p->ws_bytes >= q.bytes, "chap_wgrad: workspace %zu < %zu bytes", p->ws_bytes, q.bytes);
    const int sd = p->dims == 3 ? p->stride : 1;
    CHAP_CHECK_ARG(p->ID == (p->stride == 1 ? p->D : p->D * sd) && p->IH == p->H * p->stride && p->IW == p->W * p->stride,
                   "chap_wgrad: A dims (%d,%d,%d) do not match grid (%d,%d,%d) stride %d", p->ID, p->IH, p->IW, p->D, p->H, p->W, p->stride);
    float* ws = (float*)p->ws;
    float* ws_db = p->db ? ws + (size_t)q.nsplit * (q.slab / sizeof(float)) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (p->dtype == CHAP_BF16) r = chap_wgrad_launch_bf16(p, q.KC, ws, ws_db, q.nsplit, q.Ca, q.Cb, s);
    else if (p->dtype == CHAP_F32) r = chap_wgrad_launch_f32(p, q.KC, ws, ws_db, q.nsplit, q.Ca, q.Cb, s);
    else { chap_set_error("chap_wgrad: dtype=%d", p->dtype); return CHAP_EINVAL; }
    if (r) return r;
    const long total = (long)q.taps * q.Ca * q.Cb;
    const int kcv = p->kc_valid > 0 ? p->kc_valid : q.Ca, knv = p->kn_valid > 0 ? p->kn_valid : q.Cb;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 16)), dim3(256), 0, s, (const float*)ws, (const float*)ws_db, q.nsplit, q.taps, q.Ca, q.Cb,
                       p->dw, (long)p->s_tap, (long)p->s_kc, (long)p->s_kn, kcv, knv, p->db);
    CHAP_LAUNCH_CHECK("chap_wgrad(reduce)");
    return CHAP_OK;
}
