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
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    p.out = out.data_ptr()
    L.call("chap_pack_weights", p, _stream())
    return out


def conv_fwd(srcs, wpacked, bias, cout, out, *, grid, in_dims, ksize, stride, dims, combine=0,
             out_ld=None, out_coff=0, out_mode=0, out_cn=0, out_planar=False, out_f32=False,
             stats=None, stats_reps=1):
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
    p.stats, p.stats_reps = _p(stats), stats_reps
    p.dtype = dt(srcs[0].raw)
    L.call("chap_conv_fwd", p, _stream())


def conv_c1_fwd(x, w, bias, out, *, dims, stats=None, stats_reps=1):
    """x: fp32 [N, D, H, W] (C == 1)."""
    p = L.ConvC1Params()
    p.x, p.w, p.bias, p.out = x.data_ptr(), w.data_ptr(), _p(bias), out.data_ptr()
    p.stats, p.stats_reps = _p(stats), stats_reps
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
        ws = torch.empty(L.size_of("chap_conv_c1_bwd_ws", p), dtype=torch.uint8, device=g.device)
        p.ws = ws.data_ptr()
    L.call("chap_conv_c1_bwd", p, _stream())


def wgrad(a_srcs, b, dw, strides, *, grid, in_dims, ksize, stride, dims, combine=0, db=None):
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
    p.dtype = dt(b.raw)
    nbytes = L.size_of("chap_wgrad_ws", p)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dw.device)
    p.ws, p.ws_bytes = ws.data_ptr(), nbytes
    L.call("chap_wgrad", p, _stream())


def bn_finalize(stats, stats_reps, gamma, beta, running_mean, running_var, nbt, count, eps, momentum,
                scale, shift, mean=None, invstd=None):
    p = L.BnFinalizeParams()
    p.stats, p.stats_reps, p.gamma, p.beta = stats.data_ptr(), stats_reps, gamma.data_ptr(), beta.data_ptr()
    p.running_mean, p.running_var, p.num_batches_tracked = _p(running_mean), _p(running_var), _p(nbt)
    p.scale, p.shift, p.mean, p.invstd = scale.data_ptr(), shift.data_ptr(), _p(mean), _p(invstd)
    p.C, p.count, p.eps, p.momentum = gamma.numel(), float(count), eps, momentum
    L.call("chap_bn_finalize", p, _stream())


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, scale, shift):
    p = L.BnEvalParams()
    p.gamma, p.beta, p.running_mean, p.running_var = gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr()
    p.scale, p.shift, p.C, p.eps = scale.data_ptr(), shift.data_ptr(), gamma.numel(), eps
    L.call("chap_bn_eval_affine", p, _stream())


def act_pool2(lazy, out, idx=None):
    p = L.PoolParams()
    lazy.fill(p.r)
    p.out, p.idx = out.data_ptr(), _p(idx)
    p.N, p.H, p.W = lazy.raw.shape[0], lazy.raw.shape[2], lazy.raw.shape[3]
    p.dtype = dt(lazy.raw)
    L.call("chap_act_pool2", p, _stream())


def upsample2x(lazy, out, *, dims, out_coff=0):
    p = L.UpsampleParams()
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
            dgamma=None, dbeta=None, count=1.0):
    """grads: list of (tensor, channel offset). With mean/invstd/gamma -> BN backward fused in."""
    sums = None
    if mean is not None:
        sums = torch.zeros(2 * lazy.C, dtype=torch.float32, device=gout.device)
    p = _act_bwd_params(lazy, grads, g_pool, pool_idx, mean, invstd, gamma, sums, gout, dgamma, dbeta, count)
    if mean is not None:
        L.call("chap_act_bwd_reduce", p, _stream())
    L.call("chap_act_bwd_apply", p, _stream())


def planar_to_cl(x, out, out_coff=0):
    """fp32 [N, C, *spatial] -> out [N, D, H, W, ld] (dtype of out)."""
    p = L.PlanarToClParams()
    N, Cc = x.shape[0], x.shape[1]
    p.in_, p.out, p.N, p.C, p.P = x.data_ptr(), out.data_ptr(), N, Cc, x[0, 0].numel()
    p.out_ld, p.out_coff, p.dtype = out.shape[-1], out_coff, dt(out)
    L.call("chap_planar_to_cl", p, _stream())


def cl_to_planar(lazy, out):
    p = L.ClToPlanarParams()
    lazy.fill(p.r)
    p.out, p.N, p.P, p.dtype = out.data_ptr(), lazy.raw.shape[0], lazy.raw[0, ..., 0].numel(), dt(lazy.raw)
    L.call("chap_cl_to_planar", p, _stream())
