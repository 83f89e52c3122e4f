// chap_conv_fwd / chap_pack_weights: argument checks, blocking choice, weight packing kernel.
#include <cstdlib>
#include <atomic>
#include "common.h"
#include "launch.h"

int chap_conv_launch_kpar_bf16(const chap_conv_params* p, int KC, int cpar, hipStream_t s);
int chap_conv_launch_wp_bf16(const chap_conv_params* p, int KC, hipStream_t s);
#define DECL_GEOM(dt, g) int chap_conv_launch_##dt##_g##g(const chap_conv_params* p, int KC, int NT, int MR, hipStream_t s);
DECL_GEOM(bf16, 1) DECL_GEOM(bf16, 2) DECL_GEOM(bf16, 3) DECL_GEOM(bf16, 4) DECL_GEOM(bf16, 5)
DECL_GEOM(f32, 1) DECL_GEOM(f32, 2) DECL_GEOM(f32, 3) DECL_GEOM(f32, 4) DECL_GEOM(f32, 5)
typedef int (*conv_launch_fn)(const chap_conv_params*, int, int, int, hipStream_t);

struct conv_blocking { int KC, GPT, NP, STEPS, nchunks, ntiles; };

// K-side geometry shared by the packer and the kernel: Ck = GEMM-K channels, taps = kernel taps.
static conv_blocking blocking_for(int Ck, int taps, int Cout_logical, int dtype) {
    conv_blocking b;
    b.KC = (Ck >= 32 && Ck % 32 == 0) ? 32 : 16;      // e.g. 16 + 32 concatenated channels (unet_3D) walk in chunks of 16
    // (tried in round 2: 64-channel chunks for the deep 2D layers, to halve the chain of dependent chunk round trips of a tile --
    //  weights then stream from L2 to keep two blocks per CU; single layers -0..6 %, the whole 2D iteration 7.83 -> 8.26 ms: dropped)
    (void)dtype;
    // 3D 3x3x3 with <= 32 input channels (the large-volume levels): chunks of 16 keep the 6x6x18 halo brick at 41 KB
    // (two buffers), which leaves LDS for the staged weights and registers for a pipelined tap loop: 32->16 at
    // 80x112x112 runs 1.9x faster than with KC = 32 (weights streamed from L2 inside the tap loop)
    {
        // 3D 3x3x3: 16-channel chunks for EVERY layer (round 1: only for 32 input channels, from stand-alone timings; on the whole iteration:
        // CHAP_CONV_KC16_MAXC = 32 / 64 / 128 / 256 -> 16.18 / 16.00 / 15.49 / 15.37 ms per 3D step -- half the halo LDS per block, more
        // blocks per CU beside the kernels of the other streams).  Lab knobs; the pack and the conv read them alike.
        static int kc16_maxc = -1, kc16_maxc2 = -1;
        if (kc16_maxc < 0) { const char* e = getenv("CHAP_CONV_KC16_MAXC"); kc16_maxc = (e && atoi(e) > 0) ? atoi(e) : 1 << 20; }
        if (kc16_maxc2 < 0) { const char* e = getenv("CHAP_CONV_KC16_MAXC2D"); kc16_maxc2 = (e && atoi(e) > 0) ? atoi(e) : 0; }
        if (taps == 27 && Ck >= 32 && Ck <= kc16_maxc) b.KC = 16;
        if (taps == 9 && Ck >= 32 && Ck <= kc16_maxc2) b.KC = 16;
    }
    b.GPT = b.KC / 8;
    b.NP = taps * b.GPT;
    b.STEPS = (b.NP + 3) / 4;
    b.nchunks = Ck / b.KC;
    b.ntiles = (Cout_logical + 15) / 16;
    return b;
}

static int check_src(const chap_src_t& s, const char* what) {
    CHAP_CHECK_ARG(s.ptr != nullptr, "%s: null tensor", what);
    CHAP_CHECK_ARG(s.C > 0 && s.C % 8 == 0, "%s: C=%d must be a positive multiple of 8", what, s.C);
    CHAP_CHECK_ARG(s.ld >= s.coff + s.C && s.ld % 8 == 0 && s.coff % 8 == 0, "%s: ld=%d coff=%d C=%d not 8-aligned / too small", what, s.ld, s.coff, s.C);
    CHAP_CHECK_ARG((s.scale == nullptr) == (s.shift == nullptr), "%s: scale and shift must come together", what);
    return CHAP_OK;
}

// The V-Net heads: 1x1x1 conv of a 16-channel lazy activation to <= 8 classes, fp32 planar logits (vnet.py:189 out_conv).  On the MFMA
// kernel this is a 16-wide tile with 2 live columns and element stores per class (43 us at 112x112x80, N = 2); it is a stream of 32 B
// in, 8 B out per voxel -- one thread per voxel, weights in LDS, the activation rounded to bf16 like an MFMA operand.
template <typename T>
__device__ __forceinline__ void conv_head1x1_kernel(const chap_conv_params& P) {
    __shared__ float w[8][16], bs[8];
    const T* wp = (const T*)P.wpacked;
    for (int i = threadIdx.x; i < 8 * 16; i += 256) {
        const int nl = i >> 4, c = i & 15;
        w[nl][c] = nl < P.Cout ? elem<T>::get(wp[(((c >> 3) * 16 + nl) << 3) + (c & 7)]) : 0.f;      // fragment order of chap_pack_weights: lane = (c/8)*16 + cout
    }
    if (threadIdx.x < 8) bs[threadIdx.x] = (P.bias && (int)threadIdx.x < P.Cout) ? P.bias[threadIdx.x] : 0.f;
    __syncthreads();
    const long npx = (long)P.D * P.H * P.W, total = (long)P.N * npx;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / npx);
        float v[16];
        src_load8<T>(P.src[0], n, i, 0, v);
        src_load8<T>(P.src[0], n, i, 8, v + 8);
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = elem<T>::get(elem<T>::put(v[c]));
        float* o = (float*)P.out + (long)n * P.Cout * npx + (i - (long)n * npx);
        for (int nl = 0; nl < P.Cout; ++nl) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) a = fmaf(w[nl][c], v[c], a);
            o[(long)nl * npx] = a + bs[nl];
        }
    }
}

extern "C" int chap_conv_fwd(const chap_conv_params* p, void* stream) {
    CHAP_CHECK_ARG(p != nullptr, "chap_conv_fwd: null params");
    CHAP_CHECK_ARG(p->nsrc == 1 || p->nsrc == 2, "chap_conv_fwd: nsrc=%d", p->nsrc);
    for (int i = 0; i < p->nsrc; ++i) { int r = check_src(p->src[i], "chap_conv_fwd src"); if (r) return r; }
    CHAP_CHECK_ARG(p->combine == 0 || (p->nsrc == 1 || p->src[0].C == p->src[1].C), "chap_conv_fwd: add-combine needs equal C");
    CHAP_CHECK_ARG(p->N > 0 && p->D > 0 && p->H > 0 && p->W > 0, "chap_conv_fwd: empty grid");
    CHAP_CHECK_ARG(p->dims == 2 || p->dims == 3, "chap_conv_fwd: dims=%d", p->dims);
    CHAP_CHECK_ARG(p->wpacked && p->out && p->Cout > 0, "chap_conv_fwd: null weights/out");
    const int sd = p->dims == 3 ? p->stride : 1;
    CHAP_CHECK_ARG(p->ID == (p->stride == 1 ? p->D : p->D * sd) && p->IH == p->H * p->stride && p->IW == p->W * p->stride,
                   "chap_conv_fwd: input dims (%d,%d,%d) do not match grid (%d,%d,%d) stride %d", p->ID, p->IH, p->IW, p->D, p->H, p->W, p->stride);
    if (p->dims == 3 && (p->src[0].keep || (p->nsrc > 1 && p->src[1].keep))) { chap_set_error("chap_conv_fwd: element keep masks are built for 2D only (3D: channel multipliers)"); return CHAP_EUNSUPPORTED; }
    if (p->out_mode == 1) CHAP_CHECK_ARG(p->out_Cn > 0 && p->out_Cn % 16 == 0 && p->Cout % p->out_Cn == 0, "chap_conv_fwd: depth-to-space needs Cn%%16==0");
    if (!p->out_planar) CHAP_CHECK_ARG(p->out_ld % 4 == 0 && p->out_coff % 4 == 0, "chap_conv_fwd: out_ld/out_coff must be multiples of 4");
    if (p->out2) CHAP_CHECK_ARG(p->out_mode == 0 && !p->out_planar && !p->out_f32 && (p->Cout & 3) == 0 && p->out2_from > 0 && p->out2_from % 16 == 0 && p->out2_from < p->Cout &&
                                p->ksize == 3 && p->stride == 1, "chap_conv_fwd: out2 needs a channel-last k3 s1 output, out2_from %% 16 == 0 inside (0, Cout)");
    const int Ck = p->combine == 0 ? p->src[0].C + (p->nsrc > 1 ? p->src[1].C : 0) : p->src[0].C;
    const int taps = p->ksize * p->ksize * (p->dims == 3 ? p->ksize : 1);
    conv_blocking b = blocking_for(Ck, taps, p->Cout, p->dtype);
    CHAP_CHECK_ARG(Ck % b.KC == 0, "chap_conv_fwd: K channels %d not a multiple of %d", Ck, b.KC);
    // geometry family
    const bool d3 = p->dims == 3;
    int geom;
    if (p->ksize == 3 && p->stride == 1) geom = d3 ? 2 : 1;
    else if (p->ksize == 1 && p->stride == 1) geom = 3;
    else if (p->ksize == 2 && p->stride == 2) geom = d3 ? 5 : 4;
    else { chap_set_error("chap_conv_fwd: unsupported (ksize=%d, stride=%d)", p->ksize, p->stride); return CHAP_EUNSUPPORTED; }
    if (p->combine == 1 && p->nsrc == 2 && geom != 2 && geom != 1) { chap_set_error("chap_conv_fwd: add-combine is built for k3 s1 only"); return CHAP_EUNSUPPORTED; }
    // blocking: NT = 16-channel tiles per block, MR = 16-pixel rows per wave.  Large tiles (halo overhead,
    // weight reuse) when the layer has plenty of pixels; small tiles when it would not fill 256 CUs.
    int NT = b.ntiles >= 4 ? 4 : (b.ntiles >= 2 ? 2 : 1);
    int MR = (geom == 2 || geom == 5) ? 1 : 2;
    const bool bf = p->dtype == CHAP_BF16;
    if (geom == 2 && bf) {
        // 3D 3x3x3 (measured on the V-Net shapes, tools/lab/conv_lab.hip sweeps): z-per-wave bricks (MR = 4) once
        // the grid has >= 64 bricks, with the widest NT that still gives >= 128 blocks; the deep, tiny layers run
        // 1 x 4 x 16 slabs with NT = 2 (two blocks per CU, weights staged through LDS).  Round-2 sweep over (NT, MR)
        // on the real layers (tools/lab/sweep_conv.sh, gpurun_out/conv_sweep3d.log): at 14x14x10, N = 2 (32 bricks) the slabs
        // win -- 128->128 22.0 vs 30.2 us, 128+128->128 24.4 vs 34.7, 256->128 37.4 vs 48.7 -- at 28x28x20 (196 bricks) the bricks
        // do (28.1 vs 41.6).
        const long bricks = (long)p->N * cdiv(p->D, 4) * cdiv(p->H, 4) * cdiv(p->W, 16);
        if (bricks >= 64) {
            MR = 4;
            // at most 32 output channels per block: the stand-alone sweeps preferred 64 for the 64-channel level (28x28x20), the whole
            // 3D iteration does not (17.09 -> 16.79 ms per step with NT = 2: more, smaller blocks share the CUs with the other streams)
            if (NT > 2) NT = 2;
            while (NT > 1 && bricks * cdiv(b.ntiles, NT) < 128) NT >>= 1;
        } else if (b.KC == 32 && NT > 2) {
            NT = 2;
        }
    }
    if (geom == 1 || geom == 3) {
        auto blocks = [&](int mr, int nt) { return (long)p->N * p->D * cdiv(p->H, 4 * mr) * cdiv(p->W, 16) * cdiv(b.ntiles, nt); };
        if (geom == 1 && bf && b.KC == 32 && b.ntiles >= 4) {
            // deep 2D layers (Cout >= 64): 8 x 16 tiles x 32 channels -- two blocks per CU hide each other's
            // staging latency and the staged weights fit; best or within 5% of best for 64@64 .. 256@16, N = 12 / 24
            MR = 2; NT = 2;
        } else if (geom == 1 && b.KC == 16 && blocks(4, NT) >= 512) MR = 4;
        else if (blocks(2, NT) >= 384) MR = 2;
        else {
            MR = 1;
            while (NT > 1 && blocks(1, NT) < 384) NT >>= 1;
        }
    }
    {   // lab knobs (tools/shape_table.py sweeps): CHAP_CONV_NT / CHAP_CONV_MR override the blocking of the k3 s1 layers with
        // at least CHAP_CONV_MINC (default 64) input channels
        const char* ent = getenv("CHAP_CONV_NT"); const char* emr = getenv("CHAP_CONV_MR"); const char* emc = getenv("CHAP_CONV_MINC");
        const int minc = emc ? atoi(emc) : 64;
        if ((geom == 1 || geom == 2) && Ck >= minc) {
            if (ent && atoi(ent) > 0 && atoi(ent) <= b.ntiles) NT = atoi(ent);
            if (emr && atoi(emr) > 0) MR = atoi(emr);
        }
    }
    if (p->dtype == CHAP_BF16 && geom == 3 && p->out_planar && p->Cout <= 8 && p->nsrc == 1 && Ck == 16 && p->src[0].C == 16 && !p->stats && p->out_mode == 0) {
        const long total = (long)p->N * p->D * p->H * p->W;
        return chap_launch<chap_conv_params, conv_head1x1_kernel<bf16_t>, 256>(dim3(chap_blocks(total, 4096)), dim3(256), 0, (hipStream_t)stream, *p, "chap_conv_fwd(head)");
    }
    // ---- the 2D full-resolution layers (all input channels in ONE chunk of 16 or 32): wave-private pipelines (conv_wp.h), bf16.  CHAP_CONV_WP (lab knob):
    // 0 = never, N = from N tiles of 4 x 16 pixels up (default 1: every eligible layer).  Stand-alone (profiles/r04_conv_wp_ab.log): 16->32 at 256 x 256
    // 34.1 -> 29.1 us, 32->16 32.0 -> 25.1, 32->64 at 128 x 128 20.6 -> 17.9, 16->32 at 128 x 128 14.9 -> 13.0; the 16->16 layer with BatchNorm prologue and
    // statistics 20.8 -> 20.5 (it is VALU-bound in its staging, not short of loads in flight), 32->32 with statistics 14.7 -> 15.3.  Whole 2D iteration, three
    // A/B pairs: 6.451 -> 6.406 ms.  Outputs are bit-identical to conv_fwd_kernel's.
    if (p->dtype == CHAP_BF16 && geom == 1 && p->out_mode == 0 && !p->out_planar && !p->out_f32 && (p->Cout & 15) == 0 && p->Cout <= 64 && p->combine == 0 &&
        Ck == b.KC && (p->nsrc == 1 || (b.KC == 32 && p->src[0].C == 16 && p->src[1].C == 16))) {
        const char* ew = getenv("CHAP_CONV_WP");
        const long wp_min = ew ? atol(ew) : 1;
        const long t4 = (long)p->N * cdiv(p->H, 4) * cdiv(p->W, 16);
        if (wp_min > 0 && t4 >= wp_min) return chap_conv_launch_wp_bf16(p, b.KC, (hipStream_t)stream);
    }
    // ---- the deep, small 3x3(x3) layers: K-chunks side by side (conv_kpar.h) instead of one after the other
    if (p->dtype == CHAP_BF16 && (geom == 1 || geom == 2) && p->out_mode == 0 && !p->out_planar && (p->Cout & 3) == 0 && !p->out2 &&
        (p->nsrc == 1 || (p->combine == 0 && p->src[0].C % b.KC == 0)) && !(d3 && (p->src[0].keep || (p->nsrc > 1 && p->src[1].keep)))) {
        // Measured per layer shape (tools/shape_table.py with CHAP_CONV_KPAR = 0 / 1, gpurun_out/kpar*.log), kernel alone: it wins while the
        // launch is about one wave of blocks -- 3D 256->256 at 7x7x5 23.3 -> 14.4 us, 128->128 at 14x14x10 20.5 -> 17.0, 256->128 39.1 -> 29.6;
        // 2D 128->256 at 32x32 20.4 -> 14.7, 256->256 at 16x16 15.2 -> 10.8, 128+128->128 23.5 -> 19.2 -- and loses where a CU gets several
        // tiles, which conv_fwd_kernel's persistent blocks overlap (3D 64->64 at 28x28x20, 1120 blocks: 24.4 -> 30.8).  Inside the
        // iteration the 3D step gains 0.8 % (18.32 -> 18.17 ms, three A/B pairs); the 2D step does NOT (7.62 -> 7.69 ms: there the
        // weights are not L2-hot as in the stand-alone timing, and a wave's fragment ring covers an L2 hit, not a MALL round trip), so
        // by default only the 3D layers take this kernel.
        const char* ek = getenv("CHAP_CONV_KPAR");            // lab knob: 0 = never, 1 = whenever eligible; default: 3D, by the block count
        const int mode = ek ? atoi(ek) : 2;
        const int cpar = b.nchunks % 4 == 0 ? 4 : (b.nchunks % 2 == 0 ? 2 : 0);
        const long kblocks = (long)p->N * p->D * cdiv(p->H, d3 ? 4 : 8) * cdiv(p->W, 16) * cdiv(b.ntiles, 2);
        const char* ekm = getenv("CHAP_CONV_KPAR_MAX");       // lab knob: the block-count threshold
        const long kmax = ekm ? atol(ekm) : 800;
        if (cpar && Ck >= 64 && (mode == 1 || (mode == 2 && d3 && kblocks <= kmax))) return chap_conv_launch_kpar_bf16(p, b.KC, cpar, (hipStream_t)stream);
    }
    static const conv_launch_fn table[2][5] = {
        {chap_conv_launch_f32_g1, chap_conv_launch_f32_g2, chap_conv_launch_f32_g3, chap_conv_launch_f32_g4, chap_conv_launch_f32_g5},
        {chap_conv_launch_bf16_g1, chap_conv_launch_bf16_g2, chap_conv_launch_bf16_g3, chap_conv_launch_bf16_g4, chap_conv_launch_bf16_g5}};
    if (p->dtype != CHAP_BF16 && p->dtype != CHAP_F32) { chap_set_error("chap_conv_fwd: dtype=%d", p->dtype); return CHAP_EINVAL; }
    return table[p->dtype == CHAP_BF16][geom - 1](p, b.KC, NT, MR, (hipStream_t)stream);
}

// ---- weight packing ------------------------------------------------------------------------
struct pack_geom { int Ck, Ck_real, Cn_logical, ctaps; };   // Ck = K channels padded to 16 (tiny heads)
static int pack_geometry(const chap_pack_params* p, pack_geom* g) {
    switch (p->kind) {
        case CHAP_PACK_CONV_FWD:     g->Ck = p->Cin;  g->Cn_logical = p->Cout;           g->ctaps = p->taps; break;
        case CHAP_PACK_CONV_DGRAD:   g->Ck = p->Cout; g->Cn_logical = p->Cin;            g->ctaps = p->taps; break;
        case CHAP_PACK_DECONV_FWD:   g->Ck = p->Cin;  g->Cn_logical = p->taps * p->Cout; g->ctaps = 1;       break;
        case CHAP_PACK_DECONV_DGRAD: g->Ck = p->Cout; g->Cn_logical = p->Cin;            g->ctaps = p->taps; break;
        case CHAP_PACK_DOWN_DGRAD:   g->Ck = p->Cout; g->Cn_logical = p->taps * p->Cin;  g->ctaps = 1;       break;
        default: chap_set_error("chap_pack: kind=%d", p->kind); return CHAP_EINVAL;
    }
    g->Ck_real = g->Ck;
    g->Ck = (g->Ck + 15) / 16 * 16;
    return CHAP_OK;
}

template <typename T>
__device__ __forceinline__ void pack_body(const float* __restrict__ w, T* __restrict__ out, int kind, int Cin, int Cout, int taps,
                                          int KC, int GPT, int NP, int STEPS, int nchunks, int ntiles, int Cn_logical, int Ck_real,
                                          long first, long stride) {
    const long total = (long)nchunks * STEPS * ntiles * 64;
    // The units are enumerated with (tap, 8-channel group) fastest, NOT in output order: neighbouring threads then read neighbouring
    // taps of the same checkpoint rows (a load instruction touches a few cache lines instead of 64 -- the kernel is gather-bound) and
    // each writes its 16-byte unit to its place in the fragment layout.
    const int PPS = STEPS * 4;
    for (long q = first; q < total; q += stride) {
        const int pp = (int)(q % PPS);
        long r = q / PPS;
        const int n16 = (int)(r & 15); r >>= 4;
        const int nt = (int)(r % ntiles);
        const int chunk = (int)(r / ntiles);
        const int step = pp >> 2, g = pp & 3;
        const long i = ((long)(chunk * STEPS + step) * ntiles + nt) * 64 + g * 16 + n16;
        const int nl = nt * 16 + n16;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = 0.f;
            if (pp < NP && nl < Cn_logical) {
                const int tap = pp / GPT, c = chunk * KC + (pp % GPT) * 8 + j;
                long a = -1;
                if (c < Ck_real)
                switch (kind) {
                    case CHAP_PACK_CONV_FWD:     a = ((long)nl * Cin + c) * taps + tap; break;
                    case CHAP_PACK_CONV_DGRAD:   a = ((long)c * Cin + nl) * taps + (taps - 1 - tap); break;
                    case CHAP_PACK_DECONV_FWD:   a = ((long)c * Cout + (nl % Cout)) * taps + (nl / Cout); break;
                    case CHAP_PACK_DECONV_DGRAD: a = ((long)nl * Cout + c) * taps + tap; break;
                    default:                     a = ((long)c * Cin + (nl % Cin)) * taps + (nl / Cin); break;  // DOWN_DGRAD
                }
                if (a >= 0) x = w[a];
            }
            v[j] = x;
        }
        st8(out + i * 8, v);
    }
}

template <typename T>
__global__ void pack_kernel(const float* __restrict__ w, T* __restrict__ out, int kind, int Cin, int Cout, int taps,
                            int KC, int GPT, int NP, int STEPS, int nchunks, int ntiles, int Cn_logical, int Ck_real) {
    pack_body<T>(w, out, kind, Cin, Cout, taps, KC, GPT, NP, STEPS, nchunks, ntiles, Cn_logical, Ck_real,
                 (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

__global__ void pack_multi_kernel(const chap_pack_entry* __restrict__ E) {
    const chap_pack_entry e = E[blockIdx.y];
    const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    if (first >= e.total) return;
    if (e.dtype == CHAP_BF16) pack_body<bf16_t>(e.w, (bf16_t*)e.out, e.kind, e.Cin, e.Cout, e.taps, e.KC, e.GPT, e.NP, e.STEPS, e.nchunks, e.ntiles, e.Cn_logical, e.Ck_real, first, stride);
    else pack_body<float>(e.w, (float*)e.out, e.kind, e.Cin, e.Cout, e.taps, e.KC, e.GPT, e.NP, e.STEPS, e.nchunks, e.ntiles, e.Cn_logical, e.Ck_real, first, stride);
}

extern "C" int chap_pack_describe(const chap_pack_params* p, chap_pack_entry* e) {
    CHAP_CHECK_ARG(p && e && p->w && p->out, "chap_pack_describe: null argument");
    pack_geom g;
    int r = pack_geometry(p, &g);
    if (r) return r;
    conv_blocking b = blocking_for(g.Ck, g.ctaps, g.Cn_logical, p->dtype);
    CHAP_CHECK_ARG(g.Ck % b.KC == 0, "chap_pack_describe: K channels %d not a multiple of %d", g.Ck, b.KC);
    e->w = p->w; e->out = p->out; e->kind = p->kind; e->Cin = p->Cin; e->Cout = p->Cout; e->taps = p->taps; e->dtype = p->dtype;
    e->KC = b.KC; e->GPT = b.GPT; e->NP = b.NP; e->STEPS = b.STEPS; e->nchunks = b.nchunks; e->ntiles = b.ntiles;
    e->Cn_logical = g.Cn_logical; e->Ck_real = g.Ck_real;
    e->total = (int64_t)b.nchunks * b.STEPS * b.ntiles * 64;
    return CHAP_OK;
}

extern "C" int chap_pack_multi(const chap_pack_entry* entries_dev, int32_t n, int64_t max_total, void* stream) {
    CHAP_CHECK_ARG(entries_dev && n > 0 && max_total > 0, "chap_pack_multi: bad argument");
    CHAP_NOT_IN_GROUP("chap_pack_multi");
    // one fragment unit (8 gathered values) per thread and loop step: the largest layer sets the time (3D 256x256x27: 221k units), so it gets
    // up to 1024 blocks -- 64 made this launch, the first of every iteration, 33 us (2D) / 164 us (3D), now 24 / 103; the small entries'
    // blocks exit at once.  (Tried: one block per (K-chunk, 16-channel tile) reading its runs of the checkpoint tensor coalesced into
    // LDS: fewer, longer blocks -- 58 / 153 us.)
    long bx = (max_total + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, entries_dev);
    CHAP_LAUNCH_CHECK("chap_pack_multi");
    return CHAP_OK;
}

extern "C" size_t chap_pack_size(const chap_pack_params* p) {
    pack_geom g;
    if (!p || pack_geometry(p, &g)) return 0;
    conv_blocking b = blocking_for(g.Ck, g.ctaps, g.Cn_logical, p->dtype);
    return (size_t)b.nchunks * b.STEPS * b.ntiles * 64 * 8 * (p->dtype == CHAP_BF16 ? 2 : 4);
}

extern "C" int chap_pack_weights(const chap_pack_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->w && p->out, "chap_pack_weights: null argument");
    CHAP_NOT_IN_GROUP("chap_pack_weights");
    pack_geom g;
    int r = pack_geometry(p, &g);
    if (r) return r;
    CHAP_CHECK_ARG(g.Ck % 16 == 0, "chap_pack_weights: K channels %d must be a multiple of 16", g.Ck);
    conv_blocking b = blocking_for(g.Ck, g.ctaps, g.Cn_logical, p->dtype);
    CHAP_CHECK_ARG(g.Ck % b.KC == 0, "chap_pack_weights: K channels %d not a multiple of %d", g.Ck, b.KC);
    const long total = (long)b.nchunks * b.STEPS * b.ntiles * 64;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (p->dtype == CHAP_BF16)
        hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p->w, (bf16_t*)p->out, p->kind, p->Cin, p->Cout, p->taps,
                           b.KC, b.GPT, b.NP, b.STEPS, b.nchunks, b.ntiles, g.Cn_logical, g.Ck_real);
    else if (p->dtype == CHAP_F32)
        hipLaunchKernelGGL(pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p->w, (float*)p->out, p->kind, p->Cin, p->Cout, p->taps,
                           b.KC, b.GPT, b.NP, b.STEPS, b.nchunks, b.ntiles, g.Cn_logical, g.Ck_real);
    else { chap_set_error("chap_pack_weights: dtype=%d", p->dtype); return CHAP_EINVAL; }
    CHAP_LAUNCH_CHECK("chap_pack_weights");
    return CHAP_OK;
}
