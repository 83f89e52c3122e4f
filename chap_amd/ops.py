"""Thin tensor-level wrappers over the C ABI (one Python function per entry point).

Activations are torch tensors shaped [N, D, H, W, C] (channel-last; D == 1 for 2D nets) of dtype
float32 or bfloat16.  torch is used for memory, streams and dtype bookkeeping only: every kernel
launched here is hand-written HIP from libchap_hip.so, enqueued on torch's current stream.
"""
import torch

from . import _lib as L


def _stream():
    return torch.cuda.current_stream().cuda_stream


def dt(t):
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise TypeError("unsupported activation dtype %s" % t.dtype)


def _p(t):
    return None if t is None else t.data_ptr()


class Lazy:
    """A stored raw tensor + the transform consumers apply while loading it
    (a = keep * keep_scale * chan_mul * leaky(scale*raw + shift))."""

    __slots__ = ("raw", "C", "coff", "scale", "shift", "act", "slope", "keep", "keep_scale", "chan_mul")

    def __init__(self, raw, scale=None, shift=None, act=False, slope=0.0, keep=None, keep_scale=1.0,
                 chan_mul=None, C=None, coff=0):
        self.raw, self.scale, self.shift = raw, scale, shift
        self.act, self.slope = bool(act), float(slope)
        self.keep, self.keep_scale, self.chan_mul = keep, float(keep_scale), chan_mul
        self.C = raw.shape[-1] if C is None else C
        self.coff = coff

    @property
    def ld(self):
        return self.raw.shape[-1]

    def fill(self, s):
        s.ptr, s.scale, s.shift = _p(self.raw), _p(self.scale), _p(self.shift)
        s.keep, s.chan_mul = _p(self.keep), _p(self.chan_mul)
        s.C, s.ld, s.coff = self.C, self.ld, self.coff
        s.act, s.slope, s.keep_scale = int(self.act), self.slope, self.keep_scale
        return s

    def src(self):
        return self.fill(L.Src())


def pack_weights(w, kind, dtype, cin, cout, taps):
    """w: fp32 parameter in checkpoint layout -> packed tensor (uint8 storage) in `dtype`."""
    p = L.PackParams()
    p.w, p.kind, p.Cin, p.Cout, p.taps = w.data_ptr(), kind, cin, cout, taps
    p.dtype = L.F32 if dtype == torch.float32 else L.BF16
    nbytes = L.size_of("chap_pack_size", p)
    out = L.hold_empty(nbytes, dtype=torch.uint8, device=w.device)
    p.out = out.data_ptr()
    L.call("chap_pack_weights", p, _stream())
    return out


def conv_fwd(srcs, wpacked, bias, cout, out, *, grid, in_dims, ksize, stride, dims, combine=0,
             out_ld=None, out_coff=0, out_mode=0, out_cn=0, out_planar=False, out_f32=False,
             stats=None, stats_shift=None, out2=None):
    """stats: fp32 buffer from `stats_buffer(cout)` (per-block partial slots of the shifted moments, chap_hip.h);
    stats_shift: fp32 [real channels] or None.  out2: a second output tensor taking the channels [out.shape[-1], cout) (chap_conv_params.out2)."""
    p = L.ConvParams()
    for i, s in enumerate(srcs):
        s.fill(p.src[i])
    p.nsrc, p.combine = len(srcs), combine
    p.N, p.D, p.H, p.W = grid
    p.ID, p.IH, p.IW = in_dims
    p.ksize, p.stride, p.dims = ksize, stride, dims
    p.wpacked, p.bias, p.out = wpacked.data_ptr(), _p(bias), out.data_ptr()
    p.Cout = cout
    p.out_ld = out.shape[-1] if out_ld is None else out_ld
    p.out_coff, p.out_mode, p.out_Cn = out_coff, out_mode, out_cn
    p.out_planar, p.out_f32 = int(out_planar), int(out_f32)
    p.stats, p.stats_shift = _p(stats), _p(stats_shift)
    p.dtype = dt(srcs[0].raw)
    if out2 is not None:
        assert out2.shape == out.shape and out2.dtype == out.dtype and cout == 2 * out.shape[-1]
        p.out2, p.out2_from = out2.data_ptr(), out.shape[-1]
    L.call("chap_conv_fwd", p, _stream())


def stats_size(clog):
    """floats of a BatchNorm-statistics buffer for a conv with `clog` logical output channels."""
    return L.STATS_HDR + L.STATS_MAX_SLOTS * 2 * clog


def stats_buffer(clog, device):
    return L.hold_empty(stats_size(clog), dtype=torch.float32, device=device)


def stats_totals(stats, clog, creal=None):
    """Host-side view of a statistics buffer (tests / debugging): fp64 [2, creal] = (sum(v - c), sum((v - c)^2)) over all
    slots in use, the sub-lattice rows of a transposed conv (clog = nsub * creal) folded.  Synchronises."""
    n = int(stats[:1].view(torch.int32).item())
    t = stats[L.STATS_HDR:L.STATS_HDR + n * 2 * clog].view(n, 2, clog).double().sum(0)
    creal = clog if creal is None else creal
    return t.view(2, clog // creal, creal).sum(1)


def stats_from_moments(s, q):
    """A one-slot statistics buffer holding the given per-channel moments (tests of chap_bn_finalize)."""
    c = s.numel()
    buf = torch.zeros(stats_size(c), dtype=torch.float32, device=s.device)
    buf[:1].view(torch.int32).fill_(1)
    buf[L.STATS_HDR:L.STATS_HDR + c] = s
    buf[L.STATS_HDR + c:L.STATS_HDR + 2 * c] = q
    return buf


def conv_c1_fwd(x, w, bias, out, *, dims, stats=None, stats_shift=None):
    """x: fp32 [N, D, H, W] (C == 1)."""
    p = L.ConvC1Params()
    p.x, p.w, p.bias, p.out = x.data_ptr(), w.data_ptr(), _p(bias), out.data_ptr()
    p.stats, p.stats_shift = _p(stats), _p(stats_shift)
    p.N, p.D, p.H, p.W = x.shape
    p.dims, p.Cout, p.dtype = dims, out.shape[-1], dt(out)
    L.call("chap_conv_c1_fwd", p, _stream())


def conv_c1_bwd(g, w, x, *, dims, dx=None, dw=None, db=None):
    p = L.ConvC1BwdParams()
    p.g, p.w, p.x = g.data_ptr(), w.data_ptr(), _p(x)
    p.dx, p.dw, p.db = _p(dx), _p(dw), _p(db)
    p.N, p.D, p.H, p.W = g.shape[:4]
    p.dims, p.Cout, p.dtype = dims, g.shape[-1], dt(g)
    ws = None
    if dw is not None or db is not None:
        ws = L.hold_empty(L.size_of("chap_conv_c1_bwd_ws", p), dtype=torch.uint8, device=g.device)
        p.ws = ws.data_ptr()
    L.call("chap_conv_c1_bwd", p, _stream())


def wgrad(a_srcs, b, dw, strides, *, grid, in_dims, ksize, stride, dims, combine=0, db=None, kc_valid=0, kn_valid=0):
    p = L.WgradParams()
    for i, s in enumerate(a_srcs):
        s.fill(p.a[i])
    p.na, p.combine = len(a_srcs), combine
    b.fill(p.b)
    p.N, p.D, p.H, p.W = grid
    p.ID, p.IH, p.IW = in_dims
    p.ksize, p.stride, p.dims = ksize, stride, dims
    p.dw = dw.data_ptr()
    p.s_tap, p.s_kc, p.s_kn = strides
    p.db = _p(db)
    p.kc_valid, p.kn_valid = kc_valid, kn_valid
    p.dtype = dt(b.raw)
    nbytes = L.size_of("chap_wgrad_ws", p)
    ws = L.hold_empty(max(nbytes, 16), dtype=torch.uint8, device=dw.device)
    p.ws, p.ws_bytes = ws.data_ptr(), nbytes
    L.call("chap_wgrad", p, _stream())


def bn_finalize(stats, gamma, beta, running_mean, running_var, nbt, count, eps, momentum,
                scale, shift, mean=None, invstd=None, stats_shift=None, clog=None):
    """stats: the conv's partial-slot buffer; clog: its logical channel count (C * sub-positions for a transposed conv);
    stats_shift: the shift the conv was given (read before running_mean is updated)."""
    p = L.BnFinalizeParams()
    p.stats, p.stats_shift, p.gamma, p.beta = stats.data_ptr(), _p(stats_shift), gamma.data_ptr(), beta.data_ptr()
    p.Clog = gamma.numel() if clog is None else clog
    p.running_mean, p.running_var, p.num_batches_tracked = _p(running_mean), _p(running_var), _p(nbt)
    p.scale, p.shift, p.mean, p.invstd = scale.data_ptr(), shift.data_ptr(), _p(mean), _p(invstd)
    p.C, p.count, p.eps, p.momentum = gamma.numel(), float(count), eps, momentum
    L.call("chap_bn_finalize", p, _stream())


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, scale, shift):
    p = L.BnEvalParams()
    p.gamma, p.beta, p.running_mean, p.running_var = gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr()
    p.scale, p.shift, p.C, p.eps = scale.data_ptr(), shift.data_ptr(), gamma.numel(), eps
    L.call("chap_bn_eval_affine", p, _stream())


def act_pool2(lazy, out, idx=None, dims=2):
    p = L.PoolParams()
    lazy.fill(p.r)
    p.out, p.idx = out.data_ptr(), _p(idx)
    p.N, p.H, p.W = lazy.raw.shape[0], lazy.raw.shape[2], lazy.raw.shape[3]
    p.D = lazy.raw.shape[1] if dims == 3 else 1
    p.dtype = dt(lazy.raw)
    L.call("chap_act_pool2", p, _stream())


def upsample2x(lazy, out, *, dims, out_coff=0, half_pixel=False):
    p = L.UpsampleParams()
    p.half_pixel = int(half_pixel)
    lazy.fill(p.r)
    p.out, p.out_ld, p.out_coff = out.data_ptr(), out.shape[-1], out_coff
    p.N, p.D, p.H, p.W = lazy.raw.shape[:4]
    p.dims, p.dtype = dims, dt(lazy.raw)
    L.call("chap_upsample2x", p, _stream())


def upsample2x_bwd(g, g_coff, C, out, *, dims):
    p = L.UpsampleBwdParams()
    p.g, p.g_ld, p.g_coff, p.out = g.data_ptr(), g.shape[-1], g_coff, out.data_ptr()
    p.N, p.D, p.H, p.W = out.shape[:4]
    p.C, p.dims, p.dtype = C, dims, dt(g)
    L.call("chap_upsample2x_bwd", p, _stream())


def _act_bwd_params(lazy, grads, g_pool, pool_idx, mean, invstd, gamma, sums, gout, dgamma, dbeta, count):
    p = L.ActBwdParams()
    for i, (g, coff) in enumerate(grads):
        p.g[i], p.g_ld[i], p.g_coff[i] = g.data_ptr(), g.shape[-1], coff
    p.ng = len(grads)
    p.g_pool, p.pool_idx = _p(g_pool), _p(pool_idx)
    lazy.fill(p.r)
    p.mean, p.invstd, p.gamma, p.sums = _p(mean), _p(invstd), _p(gamma), _p(sums)
    p.gout, p.dgamma, p.dbeta = _p(gout), _p(dgamma), _p(dbeta)
    p.N, p.D, p.H, p.W = lazy.raw.shape[:4]
    p.bn, p.count, p.dtype = int(mean is not None), float(count), dt(lazy.raw)
    return p


def act_bwd(lazy, grads, gout, *, g_pool=None, pool_idx=None, mean=None, invstd=None, gamma=None,
            dgamma=None, dbeta=None, count=1.0, bn_mode=None, sums=None):
    """grads: list of (tensor, channel offset).
    bn_mode 1 (default when mean is given): training-mode BN backward fused in;
    bn_mode 2: fixed affine (eval-mode BN), dgamma/dbeta from the same reduction when requested."""
    if bn_mode is None:
        bn_mode = 1 if mean is not None else 0
    need_reduce = bn_mode == 1 or (bn_mode == 2 and (dgamma is not None or dbeta is not None))
    if need_reduce and sums is None:
        sums = L.hold_empty(act_bwd_sums_size(lazy.C), dtype=torch.float32, device=gout.device)
    p = _act_bwd_params(lazy, grads, g_pool, pool_idx, mean, invstd, gamma, sums, gout, dgamma, dbeta, count)
    p.bn = bn_mode
    if need_reduce:
        L.call("chap_act_bwd_reduce", p, _stream())
    L.call("chap_act_bwd_apply", p, _stream())
    return sums


def act_bwd_sums_size(c):
    """floats of the BN-backward workspace: row 0 = totals, rows 1.. = per-block partials (chap_hip.h)."""
    return (1 + L.ACT_BWD_SLOTS) * 2 * c


def channel_sum(lazy, out):
    p = L.ChanSumParams()
    lazy.fill(p.r)
    ws = L.hold_empty(L.CHANSUM_SLOTS * lazy.C, dtype=torch.float32, device=out.device)
    p.ws = ws.data_ptr()
    p.out, p.npix, p.pix_per_sample, p.dtype = out.data_ptr(), lazy.raw[..., 0].numel(), lazy.raw[0, ..., 0].numel(), dt(lazy.raw)
    L.call("chap_channel_sum", p, _stream())


def planar_to_cl(x, out, out_coff=0, cpad=0):
    """fp32 [N, C, *spatial] -> out [N, D, H, W, ld] (dtype of out); channels [C, cpad) zero-filled."""
    p = L.PlanarToClParams()
    N, Cc = x.shape[0], x.shape[1]
    p.in_, p.out, p.N, p.C, p.P = x.data_ptr(), out.data_ptr(), N, Cc, x[0, 0].numel()
    p.out_ld, p.out_coff, p.Cpad, p.dtype = out.shape[-1], out_coff, cpad, dt(out)
    L.call("chap_planar_to_cl", p, _stream())


def cl_to_planar(lazy, out):
    p = L.ClToPlanarParams()
    lazy.fill(p.r)
    p.out, p.N, p.P, p.dtype = out.data_ptr(), lazy.raw.shape[0], lazy.raw[0, ..., 0].numel(), dt(lazy.raw)
    L.call("chap_cl_to_planar", p, _stream())


# ------------------------------------------------------------------------------------------
# losses / training-loop helpers (fp32 planar logits [N, C, *spatial])
def mix_loss_fwd(logits, target_a, target_b, mask, w_a, w_b, smooth=1e-10, k_dice=0.0, k_ce=0.0):
    """Returns (loss[3] = (loss_a, loss_b, total), acc) -- acc is needed by mix_loss_bwd.
    target_b / mask may be None (mask of ones); (k_dice, k_ce) = (0, 0) selects mix_loss's 0.5 / 0.5."""
    N, Cc = logits.shape[0], logits.shape[1]
    p = L.MixLossParams()
    acc = L.hold_empty((1 + L.LOSS_SLOTS) * 2 * (2 + 3 * Cc), dtype=torch.float32, device=logits.device)     # row 0 = totals (read by mix_loss_bwd)
    loss = L.hold_empty(3, dtype=torch.float32, device=logits.device)
    p.logits, p.target_a, p.target_b, p.mask = logits.data_ptr(), target_a.data_ptr(), _p(target_b), _p(mask)
    p.w_a, p.w_b, p.acc, p.loss = w_a, w_b, acc.data_ptr(), loss.data_ptr()
    p.N, p.C, p.P, p.smooth, p.k_dice, p.k_ce = N, Cc, logits[0, 0].numel(), smooth, k_dice, k_ce
    L.call("chap_mix_loss_fwd", p, _stream())
    return loss, acc


def mix_loss_bwd(logits, target_a, target_b, mask, w_a, w_b, acc, dlogits, gscale=1.0, accumulate=False, smooth=1e-10, k_dice=0.0, k_ce=0.0,
                 gscale_dev=None):
    N, Cc = logits.shape[0], logits.shape[1]
    p = L.MixLossParams()
    p.logits, p.target_a, p.target_b, p.mask = logits.data_ptr(), target_a.data_ptr(), _p(target_b), _p(mask)
    p.w_a, p.w_b, p.acc, p.dlogits = w_a, w_b, acc.data_ptr(), dlogits.data_ptr()
    p.gscale, p.accumulate, p.gscale_dev = gscale, int(accumulate), _p(gscale_dev)
    p.N, p.C, p.P, p.smooth, p.k_dice, p.k_ce = N, Cc, logits[0, 0].numel(), smooth, k_dice, k_ce
    L.call("chap_mix_loss_bwd", p, _stream())


def pseudo_block(logits1, logits2, want_soft=True):
    N, Cc = logits1.shape[0], logits1.shape[1]
    sp = logits1.shape[2:]
    dev = logits1.device
    soft1 = L.hold_empty_like(logits1) if want_soft else None
    soft2 = L.hold_empty_like(logits2) if want_soft else None
    arg1 = L.hold_empty((N,) + tuple(sp), dtype=torch.int64, device=dev)
    arg2 = L.hold_empty_like(arg1)
    know = L.hold_empty((N,) + tuple(sp), dtype=torch.float32, device=dev)
    p = L.PseudoParams()
    p.logits1, p.logits2, p.soft1, p.soft2 = logits1.data_ptr(), logits2.data_ptr(), _p(soft1), _p(soft2)
    p.arg1, p.arg2, p.knowledge = arg1.data_ptr(), arg2.data_ptr(), know.data_ptr()
    p.N, p.C, p.P = N, Cc, logits1[0, 0].numel()
    L.call("chap_pseudo_block", p, _stream())
    return soft1, soft2, arg1, arg2, know


DIST_MODES = {"kl": 0, "dice": 1}


def kl_fwd_bwd(logits, targets, loss, dlogits=(None, None), gscale=1.0, gscale_dev=None, mode="kl"):
    """loss (1-elem fp32 tensor) += the VAT distance between softmax(logits) and targets, summed over the two heads:
    'kl' = mean_{n,p} KL(target || softmax(logits)), 'dice' = mean_c soft Dice (chap_hip.h, chap_kl_params)."""
    p = L.KlParams()
    p.mode = DIST_MODES[mode]
    Cc = logits[0].shape[1]
    if loss is not None or p.mode == 1:
        ws = L.hold_empty((1 + L.LOSS_SLOTS) * 2 * (3 * Cc + 1), dtype=torch.float32, device=logits[0].device)
        p.ws = ws.data_ptr()
    for h in range(2):
        p.logits[h], p.target[h], p.dlogits[h] = logits[h].data_ptr(), targets[h].data_ptr(), _p(dlogits[h])
    p.loss, p.gscale, p.gscale_dev = _p(loss), gscale, _p(gscale_dev)
    p.N, p.C, p.P = logits[0].shape[0], logits[0].shape[1], logits[0][0, 0].numel()
    L.call("chap_kl_fwd_bwd", p, _stream())


def l2_normalize(x, out, eps=1e-8):
    p = L.L2NormParams()
    ws = L.hold_empty(x.shape[0] * L.L2NORM_SLOTS, dtype=torch.float32, device=x.device)
    p.in_, p.out, p.N, p.P, p.eps, p.ws = x.data_ptr(), out.data_ptr(), x.shape[0], x[0].numel(), eps, ws.data_ptr()
    L.call("chap_l2_normalize", p, _stream())


def perturb(x, d, out, alpha, mask=None, sign=False):
    p = L.AxpyParams()
    p.x, p.d, p.mask, p.out = x.data_ptr(), d.data_ptr(), _p(mask), out.data_ptr()
    p.alpha, p.sign, p.n = alpha, int(sign), x.numel()
    L.call("chap_perturb", p, _stream())


def rand_uniform(out, seed, lo=0.0, hi=1.0, seed_dev=None):
    p = L.RandParams()
    p.out, p.seed, p.seed_dev, p.n, p.lo, p.hi = out.data_ptr(), seed, _p(seed_dev), out.numel(), lo, hi
    L.call("chap_rand_uniform", p, _stream())


def keep_mask(keep, seed, prob, seed_dev=None):
    p = L.KeepMaskParams()
    p.keep, p.seed, p.seed_dev, p.n, p.p = keep.data_ptr(), seed, _p(seed_dev), keep.numel(), prob
    L.call("chap_keep_mask", p, _stream())


def chan_mask(mul, seed, prob, seed_dev=None):
    p = L.ChanMaskParams()
    p.mul, p.seed, p.seed_dev, p.n, p.p = mul.data_ptr(), seed, _p(seed_dev), mul.numel(), prob
    L.call("chap_chan_mask", p, _stream())


DROP_MODES = {"dropout2d": 0, "comp_binomial": 1, "scores": 2}


def sample_channel_sum(lazy, nchunk=32):
    """Per-sample spatial sums of a lazy activation [N, ..., C] as partial sums fp32 [N, nchunk, C] (fixed order)."""
    p = L.SampleChanSumParams()
    lazy.fill(p.r)
    N = lazy.raw.shape[0]
    partial = L.hold_empty(N, nchunk, lazy.C, dtype=torch.float32, device=lazy.raw.device)
    p.partial, p.N, p.nchunk, p.pix_per_sample, p.dtype = partial.data_ptr(), N, nchunk, lazy.raw[0, ..., 0].numel(), dt(lazy.raw)
    L.call("chap_sample_channel_sum", p, _stream())
    return partial


def channel_drop(mul1, mul2, u1, u2, B, mode, *, pool_partial=None, npix=1, grad_sim=None, comp=False, branch=0,
                 prob_kind="sigmoid", probs_out=None):
    """FilterDropout.perform_dropout's two channel masks of one level, as the chan_mul rows of the (B + U) decoder
    batch: mul1/mul2 fp32 [B + U, C]; u1/u2 fp32 [U, C] uniforms."""
    p = L.ChannelDropParams()
    U, Cc = u1.shape
    assert tuple(mul1.shape) == (B + U, Cc) and tuple(mul2.shape) == (B + U, Cc) and tuple(u2.shape) == (U, Cc)
    p.pool_partial, p.grad_sim, p.u1, p.u2 = _p(pool_partial), _p(grad_sim), u1.data_ptr(), u2.data_ptr()
    p.mul1, p.mul2, p.probs_out = mul1.data_ptr(), mul2.data_ptr(), _p(probs_out)
    p.inv_npix, p.nchunk = 1.0 / npix, (pool_partial.shape[1] if pool_partial is not None else 0)
    p.B, p.U, p.C, p.mode, p.comp, p.branch = B, U, Cc, DROP_MODES[mode], int(bool(comp)), int(branch)
    p.prob_kind = {"sigmoid": 0, "gauss": 1}[prob_kind]
    L.call("chap_channel_drop", p, _stream())


def fold_perturbed(g, coff, Cc, mul, B, U):
    """Adjoint of cat((feat, mul * feat[B-U:])): g [B + U, ..., ld] (channels [coff, coff + Cc)) -> [B, ..., Cc]."""
    p = L.FoldParams()
    out = L.hold_empty((B,) + tuple(g.shape[1:-1]) + (Cc,), dtype=g.dtype, device=g.device)
    p.g, p.mul, p.out = g.data_ptr(), _p(mul), out.data_ptr()
    p.B, p.U, p.C, p.ld, p.coff = B, U, Cc, g.shape[-1], coff
    p.pix_per_sample, p.dtype = g[0, ..., 0].numel(), dt(g)
    L.call("chap_fold_perturbed", p, _stream())
    return out


def box_mix(a, b, out, box):
    """out = inside box ? b : a ; a/b/out [N, (1,) H, W] float32 or int64; box: device int32[4]."""
    p = L.BoxMixParams()
    p.a, p.b, p.out, p.box = a.data_ptr(), b.data_ptr(), out.data_ptr(), box.data_ptr()
    p.N, p.H, p.W, p.is_i64 = a.shape[0], a.shape[-2], a.shape[-1], int(a.dtype == torch.int64)
    p.D = a.shape[-3] if box.numel() == 6 else 1
    L.call("chap_box_mix", p, _stream())


def box_mask(mask, box):
    p = L.BoxMaskParams()
    p.mask, p.box, p.N, p.H, p.W = mask.data_ptr(), box.data_ptr(), mask.shape[0], mask.shape[-2], mask.shape[-1]
    p.D = mask.shape[-3] if box.numel() == 6 else 1
    L.call("chap_box_mask", p, _stream())


def largest_cc(labels, num_classes):
    """labels int64 [N, H, W] (8-connectivity) or [N, D, H, W] (26-connectivity) -> same shape, keeping the
    largest connected component per (sample, class > 0)."""
    out = L.hold_empty_like(labels)
    p = L.LccParams()
    p.labels, p.out = labels.data_ptr(), out.data_ptr()
    p.N, p.H, p.W, p.num_classes = labels.shape[0], labels.shape[-2], labels.shape[-1], num_classes
    p.D = labels.shape[1] if labels.dim() == 4 else 1
    ws = L.hold_empty(L.size_of("chap_lcc_ws", p), dtype=torch.uint8, device=labels.device)
    p.ws = ws.data_ptr()
    L.call("chap_largest_cc", p, _stream())
    return out


def diff_mask(p1, p2, knowledge, scale, topk):
    """[N, H, W] maps (3D volumes are passed as [N, D*H, W]: in-plane scale x scale patches)."""
    if knowledge.dim() == 4:
        n, d, h, w = knowledge.shape
        return diff_mask(p1.reshape(n, d * h, w), p2.reshape(n, d * h, w), knowledge.reshape(n, d * h, w), scale, topk).reshape(n, d, h, w)
    N, H, W = knowledge.shape
    out = L.hold_empty(N, H, W, dtype=torch.float32, device=knowledge.device)
    ws = L.hold_empty(N * (H // scale) * (W // scale) + N, dtype=torch.float32, device=knowledge.device)
    p = L.DiffMaskParams()
    p.p1, p.p2, p.knowledge, p.out, p.pooled_ws = p1.data_ptr(), p2.data_ptr(), knowledge.data_ptr(), out.data_ptr(), ws.data_ptr()
    p.N, p.H, p.W, p.scale, p.topk = N, H, W, scale, topk
    L.call("chap_diff_mask", p, _stream())
    return out


def grad_sim(gl, gu, score, ema=0.0):
    """score[c] = ema * score[c] + (1 - ema) * cos(gl[c, :], gu[c, :]) for the rows (output channels) of a conv kernel's two
    gradients gl, gu [Cout, Cin, k, k] (contiguous fp32 views)."""
    p = L.GradSimParams()
    p.gl, p.gu, p.score = gl.data_ptr(), gu.data_ptr(), score.data_ptr()
    p.C, p.K, p.ema = gl.shape[0], gl[0].numel(), ema
    L.call("chap_grad_sim", p, _stream())


def sgd_step(param, grad, mom, lr_dev, momentum, weight_decay, grad_scale=1.0, zero_grad=True, grad2=None):
    p = L.SgdParams()
    p.param, p.grad, p.grad2, p.mom, p.lr = param.data_ptr(), grad.data_ptr(), _p(grad2), mom.data_ptr(), lr_dev.data_ptr()
    p.momentum, p.weight_decay, p.grad_scale, p.n, p.zero_grad = momentum, weight_decay, grad_scale, param.numel(), int(zero_grad)
    L.call("chap_sgd_step", p, _stream())


# ------------------------------------------------------------------------------------------
# inference callers (val_2D.py:54-97, test_3D_util.py:14-79)
ENSEMBLE_MODES = {"model1": 0, "model2": 1, "logit_ensemble": 2, "prob_ensemble": 3}


def ensemble_argmax(logits1, logits2, mode, want_prob=False):
    """logits*: fp32 [N, C, *spatial] (planar).  Returns (label uint8 [N, *spatial], prob or None)."""
    ref = logits1 if logits1 is not None else logits2
    N, Cc = ref.shape[0], ref.shape[1]
    p = L.EnsembleParams()
    label = L.hold_empty((N,) + tuple(ref.shape[2:]), dtype=torch.uint8, device=ref.device)
    prob = L.hold_empty_like(ref) if want_prob else None
    p.logits1, p.logits2, p.prob, p.label = _p(logits1), _p(logits2), _p(prob), label.data_ptr()
    p.N, p.C, p.P, p.mode = N, Cc, ref[0, 0].numel(), ENSEMBLE_MODES[mode] if isinstance(mode, str) else mode
    L.call("chap_ensemble_argmax", p, _stream())
    return label, prob


def window_accumulate(logits, origins, score, cnt):
    """logits fp32 [K, C, pw, ph, pd]; origins int32 [K, 3]; score fp32 [C, W, H, D]; cnt fp32 [W, H, D] (both += )."""
    p = L.WindowAccParams()
    p.logits, p.origins, p.score, p.cnt = logits.data_ptr(), origins.data_ptr(), score.data_ptr(), cnt.data_ptr()
    p.npatch, p.C = logits.shape[0], logits.shape[1]
    p.pw, p.ph, p.pd = logits.shape[2:]
    p.W, p.H, p.D = cnt.shape
    L.call("chap_window_accumulate", p, _stream())


def window_finalize(score, cnt):
    """score /= cnt in place; returns label uint8 [W, H, D] = argmax over classes."""
    p = L.WindowFinParams()
    label = L.hold_empty(cnt.shape, dtype=torch.uint8, device=cnt.device)
    p.score, p.cnt, p.label, p.C, p.P = score.data_ptr(), cnt.data_ptr(), label.data_ptr(), score.shape[0], cnt.numel()
    L.call("chap_window_finalize", p, _stream())
    return label
