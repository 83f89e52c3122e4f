// chap_wgrad: split selection, workspace sizing, launch, deterministic slab reduction.
#include <cstdlib>
#include <cstdio>
#include "common.h"
#include "launch.h"

int chap_wgrad_launch_bf16(const chap_wgrad_params* p, int KC, int brick, float* ws, float* ws_db, int nsplit, int Ca, int Cb, hipStream_t s);
int chap_wgrad_launch_f32(const chap_wgrad_params* p, int KC, int brick, float* ws, float* ws_db, int nsplit, int Ca, int Cb, hipStream_t s);

struct wg_plan { int Ca, Cb, KC, taps, nsplit, brick; long ntiles; size_t slab, bytes; };

static int wg_make_plan(const chap_wgrad_params* p, wg_plan* q) {
    CHAP_CHECK_ARG(p->na == 1 || p->na == 2, "chap_wgrad: na=%d", p->na);
    q->Ca = p->combine == 0 ? p->a[0].C + (p->na > 1 ? p->a[1].C : 0) : p->a[0].C;
    q->Cb = p->b.C;
    CHAP_CHECK_ARG(q->Ca % 16 == 0 && q->Cb % 8 == 0, "chap_wgrad: Ca=%d must be a multiple of 16, Cb=%d of 8", q->Ca, q->Cb);
    q->KC = (q->Ca >= 32 && q->Ca % 32 == 0) ? 32 : 16;
    if (p->dtype == CHAP_F32 && p->ksize == 2) q->KC = 16;      // fp32 k2 s2 halos: two buffers of 32 channels exceed the 160 KiB LDS
    // 3D 3x3x3, bf16: 4 x 4 x 16 bricks with 16-channel A chunks (the 1 x 4 x 16 slab stages 5.1 A-pixels per output pixel and pays a
    // barrier + a prefetch round trip per 64 pixels; the brick 2.5 and one per 256).  Threshold swept on the whole 3D iteration
    // (CHAP_WGRAD_BRICK = 2048 / 512 / 128 / 16 / 4 bricks: 17.87 / 17.65 / 17.33 / 17.17 / 17.17 ms per step): everything but the 7x7x5 level
    // (8 bricks, all the same) gains.
    q->brick = 0;
    {
        const char* eb = getenv("CHAP_WGRAD_BRICK");        // lab knob: 0 = slabs everywhere, N = bricks from N bricks up
        const long min_bricks = eb ? atol(eb) : 16;
        const long bricks = (long)p->N * cdiv(p->D, 4) * cdiv(p->H, 4) * cdiv(p->W, 16);
        if (p->dims == 3 && p->ksize == 3 && p->stride == 1 && p->dtype == CHAP_BF16 && min_bricks > 0 && bricks >= min_bricks &&
            p->a[0].C <= 256 && (p->na < 2 || p->a[1].C <= 256) && p->b.C <= 256) { q->brick = 1; q->KC = 16; }      // (<= 256 channels per source: the brick kernels' scale/shift cache)
    }
    // 2D 3x3 layers, bf16, <= 256 channels per source: wave-private pipelines (wgrad_wp.h; brick == 2), 16-channel A chunks.  CHAP_WGRAD_WP (lab knob):
    // 0 = never, N = from N tiles (8 x 16 pixels) up.  Default 1 = every eligible layer: the large images gain most (16->16 at 256 x 256 36.8 -> 25.2 us),
    // the deep layers 10-18 % (128->128 at 32 x 32 29.2 -> 25.9 us, 64+64->64 at 64 x 64 41.0 -> 33.8); whole 2D iteration, three A/B pairs per setting:
    // threshold 1024 / 256 / 64 / 1 -> 6.513 / 6.513 / 6.506 / 6.476 ms (profiles/r04_wgrad_wp_ab.log).
    {
        const char* e = getenv("CHAP_WGRAD_WP");        // (read per call: the tests force the kernel onto small ragged grids)
        const long wp_min = e ? atol(e) : 1;
        const long t8 = (long)p->N * cdiv(p->H, 8) * cdiv(p->W, 16);
        if (p->dims == 2 && p->ksize == 3 && p->stride == 1 && p->combine == 0 && p->dtype == CHAP_BF16 && wp_min > 0 && t8 >= wp_min &&
            p->a[0].C <= 256 && p->a[0].C % 16 == 0 && (p->na < 2 || (p->a[1].C <= 256 && p->a[1].C % 16 == 0)) && p->b.C <= 256) { q->brick = 2; q->KC = 16; }
    }
    // (tried: 16 x 16 tiles for the 2D 16-channel levels -- 16->16 at 256x256 35.8 -> 32.9 us with 512 blocks, 16+16->16 unchanged: not kept)
    CHAP_CHECK_ARG(q->Ca % q->KC == 0, "chap_wgrad: Ca=%d not a multiple of %d", q->Ca, q->KC);
    q->taps = p->ksize * p->ksize * (p->dims == 3 ? p->ksize : 1);
    const bool small_tile = (p->dims == 3 && p->ksize >= 2);   // 3D geometries use 4 x 16 tiles (MR = 1)
    const int TH = small_tile ? 4 : 8;
    q->ntiles = (long)p->N * (q->brick == 1 ? cdiv(p->D, 4) : p->D) * cdiv(p->H, TH) * cdiv(p->W, 16);
    static int bn16_maxc = -1;
    if (bn16_maxc < 0) { const char* e = getenv("CHAP_WGRAD_BN16_MAXC"); bn16_maxc = (e && atoi(e) > 0) ? atoi(e) : 16; }      // lab knob (wgrad_dispatch.inc reads it alike)
    const int bn = (q->Cb <= 16 || (q->brick == 1 && q->Cb <= bn16_maxc)) ? 16 : 32;
    const long pairs = (long)(q->Ca / q->KC) * cdiv(q->Cb, bn);
    // Persistent, pipelined blocks.  What bounds the small-channel layers (most of the bytes) is memory-level parallelism --
    // a block keeps a few KB in flight -- so they take as many blocks as stay resident (LDS: 3 per CU with 16-channel chunks,
    // 2 with 32); their slabs are tiny (9-37 KB).  The wide layers keep about one block per CU: more splits only add slab
    // bytes to reduce.  Targets measured per layer shape (tools/shape_table.py --only wgrad with CHAP_WGRAD_BLOCKS = 256 / 512 /
    // 768 / 1024 / 2048, gpurun_out/wgrad_sweep*.log): e.g. 16->16 at 256x256 58 / 41 / 37 / 43 / 49 us, 16+16->16 80 / 55 / 65 / 60 /
    // 72, 32+32->32 at 128x128 47 / 35 / 44 / 44 / 58, k2 s2 layers 17-23 / 20-25 / 24-28; 3D 16->16 at 112x112x80 209 / 129 / - / 134 / 145,
    // 3D 32->32 at 56x56x40 67 / 79 / - / 80 / 79.
    long target = 256;
    const bool d3 = p->dims == 3;
    const char* env = getenv("CHAP_WGRAD_BLOCKS");          // lab knob for those sweeps
    if (env && atol(env) > 0) target = atol(env);
    else if (q->brick == 2) { static long wpb = -1; if (wpb < 0) { const char* e = getenv("CHAP_WGRAD_WP_BLOCKS"); wpb = (e && atol(e) > 0) ? atol(e) : 512; } target = wpb; }
    else if (p->ksize == 2) target = 256;
    else if (q->brick) { const char* eb2 = getenv("CHAP_WGRAD_BRICK_BLOCKS"); target = bn == 16 ? 512 : (eb2 && atol(eb2) > 0 ? atol(eb2) : 256); }      // 16-wide B tiles leave LDS for two bricks per CU (3D 16->16 at 112x112x80: 88 / 66 / 85 us with 256 / 512 / 768 blocks)
    else if (q->KC == 16) target = d3 ? 512 : 768;
    // (the stand-alone timings above also favoured 512 blocks for the 2D two-source / 32-channel layers; on the whole iteration the default
    //  256 is better -- CHAP_WGRAD_TARGETS sweep, final tree: 768,512,256 -> 7.25 ms, 768,256,256 -> 7.16 ms per 2D step)
    {   // lab knob: CHAP_WGRAD_TARGETS="a,b,c" = split targets of (2D 16-channel chunks, 2D two-source / <= 32 channels, everything else k3/k1)
        const char* et = getenv("CHAP_WGRAD_TARGETS");
        long ta = 0, tb = 0, tc = 0;
        if (et && !(env && atol(env) > 0) && sscanf(et, "%ld,%ld,%ld", &ta, &tb, &tc) == 3 && p->ksize != 2 && !q->brick) {      // (brick: 3D bricks = 1, wave-private 2D = 2)
            if (q->KC == 16) { if (!d3 && ta > 0) target = ta; }
            else if (!d3 && (p->na == 2 || q->Ca <= 32)) { if (tb > 0) target = tb; }      // (the class that used to have its own default)
            else if (tc > 0) target = tc;
        }
    }
    long ns = target / pairs;
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

// Deterministic slab reduction.  A block owns E4*4 consecutive elements (one float4 per thread column) and
// splits the slabs over G = 256/E4 thread groups: group g sums slabs g, g+G, ... with 4 loads in flight, the G
// partials are combined in a fixed order through LDS.  E4 = 64 (1 KB contiguous per wave-load, G = 4) for the
// few-slab / large-weight layers, E4 = 8 (G = 32) when there are many slabs of a small weight.
// Fixed summation order -> bitwise reproducible.  total % 4 == 0 (Cb is a multiple of 16).
// Blocks [0, nb_dw) reduce dW, blocks [nb_dw, ...) reduce the bias-gradient partials the same way.
struct wgrad_reduce_args { const float* ws; const float* ws_db; int nsplit, taps, Ca, Cb; float* dw; long s_tap, s_kc, s_kn; int kc_valid, kn_valid; float* db; int nb_dw; };
template <int E4>
__device__ __forceinline__ void wgrad_reduce_kernel(const wgrad_reduce_args& A) {
    const int bid = (int)blockIdx.x;
    const float* __restrict__ ws = A.ws; const float* __restrict__ ws_db = A.ws_db;
    const int nsplit = A.nsplit, taps = A.taps, Ca = A.Ca, Cb = A.Cb, kc_valid = A.kc_valid, kn_valid = A.kn_valid, nb_dw = A.nb_dw;
    float* dw = A.dw; float* db = A.db;
    const long s_tap = A.s_tap, s_kc = A.s_kc, s_kn = A.s_kn;
    constexpr int G = 256 / E4;
    __shared__ float4 red[G][E4];
    const int col = threadIdx.x % E4, g = threadIdx.x / E4;
    const bool is_db = bid >= nb_dw;                             // bias gradient: Cb values x nsplit partials, same scheme
    const long total = is_db ? (long)Cb : (long)taps * Ca * Cb;
    const long i = ((long)(is_db ? bid - nb_dw : bid) * E4 + col) * 4;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (i < total) {
        const float* src = (is_db ? ws_db : ws) + i;
        int k = g;
        for (; k + 3 * G < nsplit; k += 4 * G) {
            const float4 a = *(const float4*)(src + (long)k * total), b = *(const float4*)(src + (long)(k + G) * total);
            const float4 c = *(const float4*)(src + (long)(k + 2 * G) * total), d = *(const float4*)(src + (long)(k + 3 * G) * total);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;  s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;  s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        for (; k < nsplit; k += G) { const float4 a = *(const float4*)(src + (long)k * total); s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
    }
    red[g][col] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (g == 0 && i < total) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < G; ++k) { const float4 a = red[k][col]; t[0] += a.x; t[1] += a.y; t[2] += a.z; t[3] += a.w; }
        const int kn0 = (int)(i % Cb); const long r = i / Cb;            // 4 consecutive kn of one (tap, kc): Cb % 4 == 0
        const int kc = (int)(r % Ca); const int tap = (int)(r / Ca);
        if (is_db) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (kn0 + j < kn_valid) db[kn0 + j] += t[j];
        } else if (kc < kc_valid) {
            float* o = dw + tap * s_tap + kc * s_kc;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (kn0 + j < kn_valid) o[(kn0 + j) * s_kn] += t[j];
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
    if (p->dims == 3 && (p->a[0].keep || (p->na > 1 && p->a[1].keep))) { chap_set_error("chap_wgrad: element keep masks on the A operand are built for 2D only"); return CHAP_EUNSUPPORTED; }
    float* ws = (float*)p->ws;
    float* ws_db = p->db ? ws + (size_t)q.nsplit * (q.slab / sizeof(float)) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (p->dtype == CHAP_BF16) r = chap_wgrad_launch_bf16(p, q.KC, q.brick, ws, ws_db, q.nsplit, q.Ca, q.Cb, s);
    else if (p->dtype == CHAP_F32) r = chap_wgrad_launch_f32(p, q.KC, q.brick, ws, ws_db, q.nsplit, q.Ca, q.Cb, s);
    else { chap_set_error("chap_wgrad: dtype=%d", p->dtype); return CHAP_EINVAL; }
    if (r) return r;
    const long total = (long)q.taps * q.Ca * q.Cb;
    const int kcv = p->kc_valid > 0 ? p->kc_valid : q.Ca, knv = p->kn_valid > 0 ? p->kn_valid : q.Cb;
    wgrad_reduce_args ra = {(const float*)ws, (const float*)ws_db, q.nsplit, q.taps, q.Ca, q.Cb, p->dw, (long)p->s_tap, (long)p->s_kc, (long)p->s_kn, kcv, knv, p->db, 0};
    const int e4 = q.nsplit >= 64 ? 8 : 64;      // (8 elements per block / 128 slab groups for the 768-split layers measured 25 us against 7 us: too few loads in flight per thread)
    const int nb_db = p->db ? cdiv(q.Cb, 4 * e4) : 0;
    ra.nb_dw = cdiv(total, 4 * e4);
    if (e4 == 8) return chap_launch<wgrad_reduce_args, wgrad_reduce_kernel<8>, 256>(dim3(ra.nb_dw + nb_db), dim3(256), 0, s, ra, "chap_wgrad(reduce)");
    return chap_launch<wgrad_reduce_args, wgrad_reduce_kernel<64>, 256>(dim3(ra.nb_dw + nb_db), dim3(256), 0, s, ra, "chap_wgrad(reduce)");
}
