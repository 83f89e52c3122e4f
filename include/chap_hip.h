/*
 * chap_hip.h -- C ABI of libchap_hip.so: hand-written HIP kernels (gfx950 / MI355X)
 * for the CHAP training hot path.
 *
 * The reference (gardnerzhou/CHAP) has no native layer: every op on its hot path is an
 * ATen/cuDNN call made from Python.  Each entry point below therefore replaces a group of
 * torch.nn calls; the reference lines are cited per function.  The reference-side binding
 * is the ctypes stub shown in INTEGRATION.md (chap_amd/_lib.py is that stub).
 *
 * Conventions
 *   - plain C: pointers + sizes in POD structs, no torch types;
 *   - the caller owns every buffer (activations, workspaces, outputs); nothing is allocated,
 *     freed or synchronised inside a call, so calls are legal under HIP stream capture;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it asynchronously;
 *   - return 0 on success, a negative CHAP_E* code otherwise; chap_last_error() gives the
 *     text (thread-local);
 *   - activations are channel-last: [N][D][H][W][C] (2D: D = 1), element type `dtype`
 *     (CHAP_F32 or CHAP_BF16); per-channel vectors, statistics, gradients of parameters and
 *     logits are fp32.
 *
 * "Lazy activation".  A BatchNorm'd conv output is stored RAW (pre-BN).  Consumers apply
 *     a = keep * keep_scale * chan_mul[n][c] * leaky(scale[c] * raw + shift[c], slope)
 * while loading (chap_src_t), so the normalised/activated tensor is never written to HBM.
 *
 * ABI versions (chap_abi_version(), checked by the binding when the library is loaded)
 *   1  the training iteration: convolutions, weight gradients, BatchNorm, pooling / up-sampling, losses, VAT and BCP
 *      helpers, largest connected component, fused SGD
 *   2  chap_mix_loss_*: k_dice / k_ce weights, optional mask and target_b; chap_ensemble_argmax,
 *      chap_window_accumulate, chap_window_finalize (the inference callers)
 *   3  chap_sample_channel_sum, chap_channel_drop, chap_fold_perturbed (channel-level perturbation); add-combine
 *      (combine = 1) also for 2D k3 s1 in chap_conv_fwd / chap_wgrad
 *   4  run-to-run deterministic reductions (the reference sets cudnn.deterministic, train_ours_2D.py:542-547): no float
 *      atomics anywhere on the training path.  BatchNorm statistics are per-block partial slots (stats layout below) of
 *      SHIFTED moments sum(x - c), sum((x - c)^2) reduced in fixed order in fp64 by chap_bn_finalize; the BN-backward
 *      sums, the loss accumulators (chap_mix_loss_*: acc is a workspace of partial rows), chap_kl_fwd_bwd (ws),
 *      chap_channel_sum (ws), chap_l2_normalize (ws = N * CHAP_L2NORM_SLOTS floats) likewise.  chap_kl_fwd_bwd: `mode`
 *      (KL or Dice distance).  chap_grad_sim.
 *   5  chap_group_begin / chap_group_next_lane / chap_group_end: grouped launches of same-shaped layers (the two decoders of a
 *      DualDecoder, two passes of one network) -- one grid instead of 2-4, bit-identical results.
 *   6  (round 3, withdrawn in 7) chap_capture_mark / _goto / _join, chap_bgrad_t, chap_wgrad_reduce_multi
 *   7  the round-3 experiments that lost their whole-iteration A/B are gone from the ABI: capture points (graph branches / leaves on one
 *      stream: +15-22 % step time), the "lazy gradient" chap_bgrad_t of chap_wgrad (+0.8 / +2.7 %), the deferred multi-layer slab reduction
 *      (+-0 / +1 %); measurements in DESIGN.md section 5.  chap_conv_params.out2 / out2_from (a concat layer's input gradient as two dense
 *      tensors).
 */
#ifndef CHAP_HIP_H
#define CHAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHAP_ABI_VERSION 7
#define CHAP_STATS_MAX_SLOTS 1024  /* per-block partial slots of the BatchNorm statistics (one per persistent conv block) */
#define CHAP_STATS_HDR 4           /* floats in front of the slots; word 0 = number of slots in use (int32)                */
#define CHAP_ACT_BWD_SLOTS 1024    /* per-block partial slots of the BN-backward sums                                      */
#define CHAP_LOSS_SLOTS 512        /* per-block partial rows of the loss accumulators                                      */
#define CHAP_CHANSUM_SLOTS 512
#define CHAP_L2NORM_SLOTS 256

enum { CHAP_F32 = 0, CHAP_BF16 = 1 };
enum { CHAP_OK = 0, CHAP_EINVAL = -1, CHAP_EUNSUPPORTED = -2, CHAP_ELAUNCH = -3 };

/* One input of a fused op: a channel-last tensor plus the transform applied on load. */
typedef struct {
    const void*    ptr;        /* element type = op dtype; NULL = absent                          */
    const float*   scale;      /* [C] or NULL (identity affine)                                   */
    const float*   shift;      /* [C]                                                             */
    const uint8_t* keep;       /* element-wise keep mask [N][D][H][W][C] (dense) or NULL          */
    const float*   chan_mul;   /* [N][C] per-sample channel multiplier (Dropout3d) or NULL        */
    int32_t        C;          /* channels taken from this tensor                                 */
    int32_t        ld;         /* pixel stride in elements (>= coff + C)                          */
    int32_t        coff;       /* first channel inside the pixel                                  */
    int32_t        act;        /* 0: none, 1: leaky-relu(slope) after the affine                  */
    float          slope;      /* 0.01 LeakyReLU (unet.py:52), 0.0 ReLU (vnet.py:28)              */
    float          keep_scale; /* 1/(1-p) for the keep mask                                       */
} chap_src_t;

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution, forward.  Replaces nn.Conv2d/Conv3d (unet.py:50,54,86,168;
 * vnet.py:19,76,106,189), nn.ConvTranspose2d/3d (unet.py:90; vnet.py:103) and -- with packed
 * weights of kind *_DGRAD -- their input-gradient.  Up to two inputs, concatenated
 * (torch.cat, unet.py:97) or added (skip add, vnet.py:202).
 *   geometry      kernel k^3 (2D: 1 x k x k), stride s, pad (k-s)/2 ... supported (k,s):
 *                 (3,1) "same", (1,1), (2,2) down-sampling.
 *   out_mode 1    depth-to-space: logical output channel n' = sub*Cn + c is stored at fine pixel
 *                 2*p + sub (transposed conv k2 s2 == 1x1 conv + depth-to-space).
 *   stats         optional per-channel moments of the fp32 result v (BatchNorm batch statistics, F.batch_norm
 *                 training=True): every persistent block b writes ITS sums  S = sum(v - c), Q = sum((v - c)^2)  over the
 *                 pixels it owned to  stats[CHAP_STATS_HDR + (b*2 + {0: S, 1: Q}) * Cout + n]  (plain stores, no
 *                 atomics, nothing to zero) and block 0 writes the number of blocks into word 0 of the header;
 *                 c = stats_shift[channel] (any per-channel constant near the mean removes the cancellation in
 *                 E[x^2] - E[x]^2; NULL = 0).  Buffer: CHAP_STATS_HDR + CHAP_STATS_MAX_SLOTS * 2 * Cout floats.
 *                 chap_bn_finalize sums the slots in a fixed order: bitwise reproducible for a given launch geometry.
 */
typedef struct {
    chap_src_t  src[2];
    int32_t     nsrc;          /* 1 or 2                                                          */
    int32_t     combine;       /* 0 concat channels, 1 add                                        */
    int32_t     N, D, H, W;    /* OUTPUT grid of the GEMM (pixels = N*D*H*W)                      */
    int32_t     ID, IH, IW;    /* input spatial dims                                              */
    int32_t     ksize;         /* 1, 2 or 3                                                       */
    int32_t     stride;        /* 1 or 2                                                          */
    int32_t     dims;          /* 2 or 3 (2: kernel and stride do not extend over D)              */
    const void* wpacked;       /* from chap_pack_weights                                          */
    const float* bias;         /* [Cout_logical] or NULL                                          */
    void*       out;
    int32_t     Cout;          /* logical GEMM N (for out_mode 1: nsub*Cn)                        */
    int32_t     out_ld;        /* pixel stride of out in elements                                 */
    int32_t     out_coff;
    int32_t     out_mode;      /* 0 same grid, 1 depth-to-space (fine grid = 2x in H,W and D if dims==3) */
    int32_t     out_Cn;        /* channels per sub-position for out_mode 1                        */
    int32_t     out_planar;    /* 1: write fp32 [N][Cout][D][H][W] (logits, NCHW) instead         */
    int32_t     out_f32;       /* 1: out element type is fp32 regardless of dtype                 */
    float*      stats;         /* partial slots (layout above) or NULL                            */
    const float* stats_shift;  /* [real channels] or NULL: the c of the shifted moments           */
    int32_t     dtype;
    int32_t     out2_from;     /* ABI 7, with out2: output channels [out2_from, Cout) go to out2 (channel c - out2_from), [0, out2_from) to out.  The    */
    void*       out2;          /* input gradient of a layer whose input was torch.cat((a, b), 1) (unet.py:98) as two DENSE tensors: the consumers of a  */
                               /* half (BatchNorm backward) read whole sectors instead of a 16-channel slice of 32-channel rows.  Same out_ld / out_coff; */
                               /* out_mode 0, channel-last, out2_from % 16 == 0; NULL = off                                                             */
} chap_conv_params;

int chap_conv_fwd(const chap_conv_params* p, void* stream);

/* Weight packing for chap_conv_fwd.  Reads the checkpoint-layout fp32 parameter
 * (OIHW / OIDHW for conv, [Cin][Cout][k..] for transposed conv) and writes the MFMA fragment
 * order consumed by the kernel. */
enum {
    CHAP_PACK_CONV_FWD     = 0,  /* conv weight  [Co][Ci][taps]      -> forward                   */
    CHAP_PACK_CONV_DGRAD   = 1,  /* conv weight, k3 s1: flipped taps, Ci<->Co -> input gradient   */
    CHAP_PACK_DECONV_FWD   = 2,  /* deconv weight [Ci][Co][sub]      -> 1x1 conv + depth-to-space */
    CHAP_PACK_DECONV_DGRAD = 3,  /* deconv weight -> k2 s2 conv of the output gradient            */
    CHAP_PACK_DOWN_DGRAD   = 4   /* k2 s2 conv weight [Co][Ci][sub]  -> 1x1 conv + depth-to-space */
};
typedef struct {
    const float* w;            /* parameter, checkpoint layout                                    */
    void*        out;          /* packed, element type dtype                                      */
    int32_t      kind;
    int32_t      Cin, Cout;    /* of the PARAMETER (nn.Module meaning)                            */
    int32_t      taps;         /* k^2 or k^3 (sub-positions for k2 s2)                            */
    int32_t      dtype;
} chap_pack_params;
size_t chap_pack_size(const chap_pack_params* p);   /* bytes of `out` */
int    chap_pack_weights(const chap_pack_params* p, void* stream);

/* All weights of a network in ONE launch (they change every optimizer step): the host fills one
 * chap_pack_entry per (weight, kind) with chap_pack_describe, uploads the array once (pointers are stable:
 * flat parameter buffer, persistent packed buffers) and calls chap_pack_multi each step. */
typedef struct {
    const float* w; void* out;
    int32_t kind, Cin, Cout, taps, dtype;
    int32_t KC, GPT, NP, STEPS, nchunks, ntiles, Cn_logical, Ck_real;
    int64_t total;             /* 64-lane fragments to write */
} chap_pack_entry;
int chap_pack_describe(const chap_pack_params* p, chap_pack_entry* e);
int chap_pack_multi(const chap_pack_entry* entries_dev, int32_t n, int64_t max_total, void* stream);

/* First layer, Cin == 1 (encoder.in_conv / block_one first conv): direct VALU conv k3 s1.
 * x is fp32 [N][D][H][W] (C=1: NCHW == NHWC).  Also its input gradient (VAT needs dL/dx)
 * and weight gradient. */
typedef struct {
    const float* x;  const float* w;  const float* bias;  /* w: [Cout][1][taps] fp32 */
    void* out;  float* stats;  const float* stats_shift;   /* as in chap_conv_params */
    int32_t N, D, H, W, dims, Cout, dtype;
} chap_conv_c1_params;
int chap_conv_c1_fwd(const chap_conv_c1_params* p, void* stream);

typedef struct {
    const void* g;             /* gradient wrt the raw conv output [..][Cout], dtype              */
    const float* w;            /* [Cout][1][taps]                                                 */
    const float* x;            /* layer input (for wgrad)                                         */
    float* dx;                 /* [N][D][H][W] fp32 or NULL                                       */
    float* dw;                 /* [Cout][taps] fp32, accumulated (+=) or NULL                     */
    float* db;                 /* [Cout] accumulated or NULL                                      */
    float* ws;                 /* workspace, chap_conv_c1_bwd_ws() bytes                          */
    int32_t N, D, H, W, dims, Cout, dtype;
} chap_conv_c1_bwd_params;
size_t chap_conv_c1_bwd_ws(const chap_conv_c1_bwd_params* p);
int    chap_conv_c1_bwd(const chap_conv_c1_bwd_params* p, void* stream);

/* Weight gradient of chap_conv_fwd geometries:
 *   dW[tap][kc][kn] = sum_p A[s*p + tap - pad][kc] * B[p][kn]
 * A = "strided" operand (halo-tiled; the layer input for conv, the fine-grid output gradient
 * for transposed conv), B = the operand on the GEMM grid.  Both may be lazy activations.
 * Partials are written per pixel-split to `ws` and reduced deterministically (no atomics) into
 * dw (+=) using element strides so the result lands in checkpoint layout.  Also db (+= sum_p B)
 * when requested (valid when B is the output gradient). */
typedef struct {
    chap_src_t  a[2];          /* strided operand, up to two concatenated/added sources           */
    int32_t     na;  int32_t combine;
    chap_src_t  b;
    int32_t     N, D, H, W;    /* GEMM grid (B's pixels)                                          */
    int32_t     ID, IH, IW;    /* A's spatial dims                                                */
    int32_t     ksize, stride, dims;
    float*      dw;            /* accumulated: dw[tap*s_tap + kc*s_kc + kn*s_kn] += ...           */
    int64_t     s_tap, s_kc, s_kn;
    int32_t     kc_valid, kn_valid; /* write only kc < kc_valid, kn < kn_valid (0 = all): padded heads */
    float*      db;            /* [Cb] += or NULL                                                 */
    void*       ws;  size_t ws_bytes;
    int32_t     dtype;
} chap_wgrad_params;
size_t chap_wgrad_ws(const chap_wgrad_params* p);
int    chap_wgrad(const chap_wgrad_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm pieces (nn.BatchNorm2d/3d, unet.py:51,55; vnet.py:21,80,110).                    */
typedef struct {
    const float* stats;                        /* partial slots from chap_conv_fwd (header + slots) */
    const float* stats_shift;                  /* the c the conv was given (NULL = 0); read BEFORE running_mean is updated */
    int32_t Clog;                              /* floats per slot row = the conv's logical Cout (C * nsub for a transposed conv) */
    const float* gamma;  const float* beta;    /* BN weight / bias                                 */
    float* running_mean; float* running_var;   /* updated in place when momentum > 0               */
    int64_t* num_batches_tracked;              /* +1 when momentum > 0 (may be NULL)               */
    float* scale; float* shift;                /* out: affine for the lazy activation              */
    float* mean;  float* invstd;               /* out: saved for backward                          */
    int32_t C;  float count;  float eps;  float momentum;
} chap_bn_finalize_params;
int chap_bn_finalize(const chap_bn_finalize_params* p, void* stream);

typedef struct {                                /* eval mode: affine from running statistics        */
    const float* gamma; const float* beta; const float* running_mean; const float* running_var;
    float* scale; float* shift; int32_t C; float eps;
} chap_bn_eval_params;
int chap_bn_eval_affine(const chap_bn_eval_params* p, void* stream);

/* Backward through  a = keep*ks*cm*leaky(scale*r+shift)  [followed by optional 2x2 max-pool
 * routing] and training-mode BatchNorm, for one stored raw tensor r.
 *   phase 1 (reduce): dz = (sum of incoming dact grads) * da/dz;  sums[0][c] = sum dz,
 *                     sums[1][c] = sum dz * rhat              (rhat = (r-mean)*invstd): per-block partial rows
 *                     sums[1 + b][2][C] (plain stores), then a fixed-order fp64 sum into row 0 and
 *                     dgamma += sums1, dbeta += sums0 -- no atomics, bitwise reproducible
 *   phase 2 (apply):  g = gamma*invstd*(dz - sums0/cnt - rhat*sums1/cnt)  -> gout (dtype)
 * Incoming gradients: up to 3 same-grid tensors (ptr, ld, coff) and one half-resolution pooled
 * gradient routed through the saved arg-max index. With bn == 0 (no BatchNorm after the conv)
 * phase 2 writes g = dz and no sums are needed. */
typedef struct {
    const void* g[3];  int32_t g_ld[3];  int32_t g_coff[3];  int32_t ng;
    const void* g_pool; const uint8_t* pool_idx;     /* [N][H/2][W/2][C] each, or NULL           */
    chap_src_t  r;                                   /* the raw tensor + its forward transform    */
    const float* mean; const float* invstd; const float* gamma;
    float* sums;            /* [1 + CHAP_ACT_BWD_SLOTS][2][C] workspace (row 0 = totals); nothing to zero */
    void*  gout;            /* [pixels][C] dtype                                                  */
    float* dgamma; float* dbeta;                     /* accumulated (+=)                          */
    int32_t N, D, H, W;  int32_t bn;  /* 0 none, 1 training-mode BN, 2 fixed affine (eval BN): g = dz*scale */
    float count;  int32_t dtype;
} chap_act_bwd_params;
int chap_act_bwd_reduce(const chap_act_bwd_params* p, void* stream);
int chap_act_bwd_apply(const chap_act_bwd_params* p, void* stream);

/* nn.MaxPool2d(2) of a lazy activation (unet.py:69): out [N][H/2][W/2][C] + arg-max idx. */
typedef struct { chap_src_t r; void* out; uint8_t* idx; int32_t N, H, W, dtype; int32_t D; /* D > 1: MaxPool3d(2) (utils.py / unet_3D.py:33) */ } chap_pool_params;
int chap_act_pool2(const chap_pool_params* p, void* stream);

/* nn.Upsample(scale_factor=2, bilinear/trilinear, align_corners=True) of a (lazy) tensor
 * (unet.py:87, vnet.py:105) and its adjoint. dims==2: D untouched. */
typedef struct {
    chap_src_t r;  void* out; int32_t out_ld, out_coff;
    int32_t N, D, H, W;  /* INPUT dims */  int32_t dims, dtype;
    int32_t half_pixel;  /* 1: align_corners=False (nn.Upsample default, networks/utils.py:264) */
} chap_upsample_params;
int chap_upsample2x(const chap_upsample_params* p, void* stream);
typedef struct {
    const void* g; int32_t g_ld, g_coff;   /* fine-grid gradient                                  */
    void* out;                               /* coarse [..][C] dtype                                */
    int32_t N, D, H, W, C, dims, dtype;     /* INPUT (coarse) dims                                 */
} chap_upsample_bwd_params;
int chap_upsample2x_bwd(const chap_upsample_bwd_params* p, void* stream);

/* Layout / dtype helpers at the module boundary. */
typedef struct { const float* in; void* out; int32_t N, C, P, out_ld, out_coff, Cpad, dtype; } chap_planar_to_cl_params;  /* channels [C,Cpad) zero-filled */
int chap_planar_to_cl(const chap_planar_to_cl_params* p, void* stream);  /* fp32 [N][C][P] -> dtype [N][P][C] */
typedef struct { chap_src_t r; float* out; int32_t N, P; int32_t dtype; } chap_cl_to_planar_params;
int chap_cl_to_planar(const chap_cl_to_planar_params* p, void* stream);  /* (lazy) [N][P][C] -> fp32 [N][C][P] */

typedef struct { chap_src_t r; float* out; int64_t npix; int64_t pix_per_sample; int32_t dtype; float* ws; /* CHAP_CHANSUM_SLOTS * C floats */ } chap_chansum_params;
int chap_channel_sum(const chap_chansum_params* p, void* stream);       /* out[c] += sum_pixels a[pixel][c] (fp32, fixed-order partials) */

/* ------------------------------------------------------------------------------------------
 * Segmentation losses on fp32 planar logits [N][C][P]  (train_ours_2D.py:198-216, 319-325). */
typedef struct {
    const float* logits;        /* [N][C][P]                                                       */
    const int64_t* target_a;    /* [N][P] labels under mask                                        */
    const int64_t* target_b;    /* [N][P] labels under (1-mask)                                    */
    const int64_t* mask;        /* [N][P] in {0,1}                                                 */
    float w_a, w_b;             /* image_weight, patch_weight                                      */
    float* acc;                 /* [1 + CHAP_LOSS_SLOTS][2][1 + 3*C + 1] fp32 workspace: row 0 = totals (ce, per-class i/p2/t2, msum) that
                                 * chap_mix_loss_fwd leaves for chap_mix_loss_bwd, rows 1.. = per-block partials; nothing to zero */
    float* loss;                /* [3]: loss_a, loss_b, total   (mix_loss return triple)           */
    float* dlogits;             /* [N][C][P] or NULL: d(total*gscale)/dlogits, accumulated (+=) if accumulate */
    float gscale; int32_t accumulate;
    int32_t N, C, P; float smooth;
    /* Generalisation for the second caller (train_ablation_2D.py:171-176,216-217): loss_k = w_k * (k_dice*Dice_k +
     * k_ce*CE_k); k_dice = k_ce = 0 selects mix_loss's 0.5 / 0.5.  mask == NULL: all ones; target_b == NULL: target_a. */
    float k_dice, k_ce;
    const float* gscale_dev;    /* optional device scalar multiplied into gscale (the consistency weight of a captured iteration) */
} chap_mix_loss_params;
int chap_mix_loss_fwd(const chap_mix_loss_params* p, void* stream);
int chap_mix_loss_bwd(const chap_mix_loss_params* p, void* stream);

typedef struct {               /* pass-A block: softmax, argmax, cross CE "knowledge"             */
    const float* logits1; const float* logits2;   /* [N][C][P]                                     */
    float* soft1; float* soft2;                    /* [N][C][P] or NULL                            */
    int64_t* arg1; int64_t* arg2;                  /* [N][P]                                       */
    float* knowledge;                              /* [N][P]                                       */
    int32_t N, C, P;
} chap_pseudo_params;
int chap_pseudo_block(const chap_pseudo_params* p, void* stream);

typedef struct {               /* VAT distance between the two heads' logits and their targets (soft outputs of pass A):  */
    const float* logits[2]; const float* target[2];
    float* loss;  float* dlogits[2];  /* loss += ; dlogits may be NULL                            */
    float gscale; const float* gscale_dev;   /* gradient scale = gscale * (*gscale_dev if given) */
    int32_t N, C, P;
    int32_t mode;              /* 0 'kl':   sum_heads mean_{n,p} KL(target || softmax(logits))                                */
                               /* 1 'dice': sum_heads mean_c [1 - (2 sum p t + s) / (sum p^2 + sum t^2 + s)], sums over (n, p), s = 1e-10
                                *           (--adv_losstype dice, train_ours_2D.py:515); needs ws                             */
    float* ws;                 /* [1 + CHAP_LOSS_SLOTS][2 heads][3*C + 1] floats: per-block partials (mode 0: one loss value per row) */
} chap_kl_params;
int chap_kl_fwd_bwd(const chap_kl_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * VAT perturbation helpers (losses.VAT2d is ABSENT from the reference; semantics defined in
 * DESIGN.md "P1"): per-sample L2 normalise, masked axpy / sign step, counter-based RNG.
 * `seed_dev` (optional, device uint64) is added to `seed` so a captured graph draws new numbers
 * on every replay once the host bumps that word. */
typedef struct { const float* in; float* out; int32_t N, P; float eps; float* ws; /* N * CHAP_L2NORM_SLOTS floats */ } chap_l2norm_params;
int chap_l2_normalize(const chap_l2norm_params* p, void* stream);   /* out[n] = in[n]/(||in[n]||+eps) */
typedef struct { const float* x; const float* d; const float* mask; float* out; float alpha; int32_t sign; int64_t n; } chap_axpy_params;
int chap_perturb(const chap_axpy_params* p, void* stream);          /* out = x + alpha*mask*(sign? sgn(d): d) */
typedef struct { float* out; uint64_t seed; const uint64_t* seed_dev; int64_t n; float lo, hi; } chap_rand_params;
int chap_rand_uniform(const chap_rand_params* p, void* stream);
typedef struct { uint8_t* keep; uint64_t seed; const uint64_t* seed_dev; int64_t n; float p; } chap_keepmask_params;
int chap_keep_mask(const chap_keepmask_params* p, void* stream);     /* keep[i] = u(i) >= p */
typedef struct { float* mul; uint64_t seed; const uint64_t* seed_dev; int64_t n; float p; } chap_chanmask_params;
int chap_chan_mask(const chap_chanmask_params* p, void* stream);     /* mul[i] = u(i) >= p ? 1/(1-p) : 0  (Dropout3d) */

/* BCP copy-paste mixing (train_ours_2D.py:91-101, 331-338). box = device int32[4] {y0, x0, bh, bw}:
 * out = inside box ? b : a  (mask is 0 inside the box: a*mask + b*(1-mask)). */
typedef struct { const void* a; const void* b; void* out; const int32_t* box; int32_t N, H, W; int32_t is_i64; int32_t D; } chap_boxmix_params;  /* D > 1: 3D cuboid, box = {z0,y0,x0,bd,bh,bw} */
int chap_box_mix(const chap_boxmix_params* p, void* stream);
typedef struct { int64_t* mask; const int32_t* box; int32_t N, H, W, D; } chap_boxmask_params;
int chap_box_mask(const chap_boxmask_params* p, void* stream);       /* loss_mask: 0 inside the box, 1 outside */

/* Largest connected component per (image, class>0), 8-connectivity (skimage.measure.label default),
 * get_ACDC_2DLargestCC (train_ours_2D.py:123-144) without the 72 device->host round trips.
 * labels/out: int64 [N][H][W]; ws: chap_lcc_ws() bytes. Ties: the component met first in raster order. */
typedef struct { const int64_t* labels; int64_t* out; void* ws; int32_t N, H, W, num_classes; int32_t D; /* D > 1: volumes, 26-connectivity */ } chap_lcc_params;
size_t chap_lcc_ws(const chap_lcc_params* p);
int    chap_largest_cc(const chap_lcc_params* p, void* stream);

/* patch.create_maskV1 (ABSENT from the reference; DESIGN.md "P2"): mask = (p1 != p2) OR
 * nearest-upsample(top-k fraction of avg_pool(knowledge, scale)), per sample. out fp32 [N][H][W]. */
typedef struct { const int64_t* p1; const int64_t* p2; const float* knowledge; float* out; float* pooled_ws; /* N*(H/s)*(W/s) + N floats */
                 int32_t N, H, W, scale; float topk; } chap_diffmask_params;
int chap_diff_mask(const chap_diffmask_params* p, void* stream);

/* Fused SGD(momentum, weight decay) over a flat fp32 parameter buffer (train_ours_2D.py:278,383):
 * g = grad*grad_scale + wd*p; m = mu*m + g; p -= lr*m; optionally grad = 0.
 * lr is read from device memory so a captured graph can be replayed with a new value. */
typedef struct { float* param; float* grad; float* grad2 /* optional second bucket, summed */; float* mom; const float* lr;
                 float momentum, weight_decay, grad_scale; int64_t n; int32_t zero_grad; } chap_sgd_params;
int chap_sgd_step(const chap_sgd_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Inference callers: the device-side parts of test_single_volume (val_2D.py:54-97) and of the
 * sliding-window test_single_case (test_3D_util.py:14-79).
 *
 * chap_ensemble_argmax: prob = softmax of one head (mode 0: logits1, 1: logits2), of the mean logits
 * (2: "logit_ensemble", val_2D.py:72-75) or the mean of the two softmaxes (3: "prob_ensemble", :76-80);
 * label = argmax_c prob (first maximum, as torch.argmax).  logits fp32 planar [N][C][P].                 */
typedef struct { const float* logits1; const float* logits2; float* prob /* [N][C][P] or NULL */; uint8_t* label /* [N][P] */;
                 int32_t N, C; int64_t P; int32_t mode; } chap_ensemble_params;
int chap_ensemble_argmax(const chap_ensemble_params* p, void* stream);

/* chap_window_accumulate: score[:, xs:xs+pw, ys:ys+ph, zs:zs+pd] += softmax(logits), cnt[same] += 1
 * (test_3D_util.py:62-69) for `npatch` patches at once: logits fp32 [npatch][C][pw][ph][pd], origins int32
 * [npatch][3]; score fp32 [C][W][H][D], cnt fp32 [W][H][D].  Gather form: every voxel adds the patches that cover it in patch
 * order k = 0, 1, ... (the order of the reference's loop), plain loads and stores -- no float atomics, bitwise reproducible.
 * chap_window_finalize: label = argmax_c score/cnt (:70-71), uint8 [W][H][D]; score is normalised in place. */
typedef struct { const float* logits; const int32_t* origins; float* score; float* cnt; int32_t npatch, C; int32_t pw, ph, pd; int32_t W, H, D; } chap_window_acc_params;
int chap_window_accumulate(const chap_window_acc_params* p, void* stream);
typedef struct { float* score; const float* cnt; uint8_t* label; int32_t C; int64_t P; } chap_window_fin_params;
int chap_window_finalize(const chap_window_fin_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Channel-level perturbation: FilterDropout.perform_dropout (FilterDropout.py:45-89) as two kernels.
 *
 * chap_sample_channel_sum: partial[n][k][c] = sum over the k-th pixel stripe of sample n of the lazy activation
 * (adaptive_avg_pool2d(unlab_feat, 1), :75); fixed-order partial sums, no atomics.  r.ptr = the first pooled sample.
 *
 * chap_channel_drop: the two multipliers of one encoder level for the decoder batch torch.cat((feat, perturb_feat))
 * (:86-87): rows [0, B) of mul1/mul2 are 1, rows [B, B+U) are the masks of the U unlabeled samples.
 *   mode 0  two independent nn.Dropout2d(0.5) (:67-69):           m = u < 0.5 ? 2 : 0
 *   mode 1  complementary Binomial(0.5) * 2 (:58-63):             m1 = u1 < 0.5 ? 2 : 0, m2 = 2 - m1
 *   mode 2  scores_dropoutV2 (:116-138) + drop_based_on_prob (:140-160): s = grad_sim[c] * mean activation,
 *           z over the channels of each sample (unbiased std), p_drop = sigmoid(-2 z) (prob_kind 0) or the 'gauss'
 *           CDF with sigma * 2 (prob_kind 1); m = bernoulli(1 - p_drop), the complementary pair when comp
 *           (branch = the reference's random.randint(0, 1)); m * numel / sum(m).  An all-zero grad_sim falls back
 *           to mode 0 (:71-73) on the device.
 * bernoulli(q) is (u < q) on the caller's uniforms u1, u2 [U][C] (chap_rand_uniform, or the test's).            */
typedef struct { chap_src_t r; float* partial /* [N][nchunk][C] */; int32_t N, nchunk; int64_t pix_per_sample; int32_t dtype; } chap_sample_chansum_params;
int chap_sample_channel_sum(const chap_sample_chansum_params* p, void* stream);
typedef struct {
    const float* pool_partial;  /* [U][nchunk][C] from chap_sample_channel_sum, or NULL (modes 0, 1)  */
    const float* grad_sim;      /* [C] or NULL (modes 0, 1)                                           */
    const float* u1; const float* u2;   /* [U][C] uniforms in [0, 1)                                   */
    float* mul1; float* mul2;   /* [B + U][C]                                                         */
    float* probs_out;           /* optional [U][C]: p_drop of mode 2                                  */
    float inv_npix;             /* 1 / pixels per sample                                              */
    int32_t nchunk, B, U, C, mode, comp, branch, prob_kind;
} chap_channel_drop_params;
int chap_channel_drop(const chap_channel_drop_params* p, void* stream);
/* Adjoint of torch.cat((feat, mul * feat[B-U:])) for the backward pass of the perturbed decoders: g is the gradient
 * w.r.t. the (B + U)-sample batch (channel-last, row length ld, channels [coff, coff + C)), out [B][pix][C] the gradient
 * w.r.t. feat: out[n] = g[n], and for the unlabeled rows out[B-U+u] += mul[B+u][c] * g[B+u] (mul NULL: plain sum).     */
typedef struct { const void* g; const float* mul; void* out; int32_t B, U, C, ld, coff; int64_t pix_per_sample; int32_t dtype; } chap_fold_params;
int chap_fold_perturbed(const chap_fold_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * grad.GradSim (ABSENT from the reference; call sites train_ours_2D.py:288,297,360,365; DESIGN.md "N1"): per OUTPUT
 * channel c of a conv kernel, the cosine similarity of the labeled-loss and the unlabeled-loss gradients
 *     sim[c] = <gl[c,:], gu[c,:]> / (||gl[c,:]|| * ||gu[c,:]|| + 1e-12),   rows of K = Cin * taps contiguous floats
 * (checkpoint layout [Cout][Cin][k][k]); score[c] = ema * score[c] + (1 - ema) * sim[c] (ema = 0: the last iteration's
 * similarity).  One wave per channel, fixed-order fp64 reduction. */
typedef struct { const float* gl; const float* gu; float* score; int32_t C, K; float ema; } chap_gradsim_params;
int chap_grad_sim(const chap_gradsim_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped launches.  An iteration of the hot path is a chain of ~600 dependent launches of kernels that do 2-5 us of work
 * in 8-20 us (fixed cost: launch gap, block prologue, one cold memory round trip), and it contains the SAME layer several
 * times on different tensors: the two decoders of DualDecoder / DualDecoder3d (unet.py:277-292: decoder1 and decoder2 share every
 * ConvBlock shape; vnet.py:234-238), and independent passes of one network (pass A on the unlabeled half, train_ours_2D.py:314,
 * and the first VAT forward, :372, meet only in the distance kernel).
 *
 *     chap_group_begin(stream);   ...calls of lane 0...   chap_group_next_lane();   ...calls of lane 1...   chap_group_end();
 *
 * Between begin and end the entry points of the networks' forward / backward path (chap_conv_fwd, chap_conv_c1_*, chap_wgrad,
 * chap_bn_finalize, chap_bn_eval_affine, chap_act_bwd_*, chap_act_pool2, chap_upsample2x*, chap_planar_to_cl, chap_cl_to_planar,
 * chap_channel_sum, chap_keep_mask, chap_chan_mask, chap_fold_perturbed) check their arguments and RECORD their launches instead
 * of issuing them; chap_group_end issues, for j = 0, 1, ..., the j-th recorded launch of every lane -- as ONE grid (gridDim.z = lanes,
 * up to 4) when they resolved to the same kernel instance and launch geometry, else one after the other in lane order.  Every block
 * does exactly the work it would do in a launch of its own (same tiles, same reduction slots): results are bit-identical to the
 * ungrouped calls.  The lanes must be independent of each other (no lane reads what another lane of the same region writes);
 * within a lane the order of the calls is kept.  All calls of a region go to the stream given to chap_group_begin; any other
 * entry point of this library fails with CHAP_EUNSUPPORTED inside a region.  State is thread-local; regions do not nest.
 * chap_group_end returns the number of grids launched (>= 0) or a negative CHAP_E* code. */
int chap_group_begin(void* stream);
int chap_group_next_lane(void);
int chap_group_end(void);
int chap_group_cancel(void);   /* leave the region without issuing what was recorded (error paths of the caller) */

/* Bandwidth calibration helper (tools/membw.py): grid-stride float4 copy with `blocks` blocks of 256. */
int chap_debug_copy(const void* src, void* dst, int64_t bytes, int32_t blocks, void* stream);

const char* chap_last_error(void);
int chap_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
