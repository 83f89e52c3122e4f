"""ctypes binding of libchap_hip.so (the C ABI declared in include/chap_hip.h).

This file is the "reference-side binding a maintainer would add" (INTEGRATION.md): the
reference is pure Python, so its FFI is ctypes.  Structures mirror the header field by field.
There is NO fallback: if the library is missing, `lib()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHAP_LIBPATH") or os.path.join(_HERE, "libchap_hip.so")      # CHAP_LIBPATH: a lab build of the same sources (A/B of compile-time variants)

F32, BF16 = 0, 1
STATS_MAX_SLOTS, STATS_HDR = 1024, 4          # CHAP_STATS_MAX_SLOTS, CHAP_STATS_HDR
ACT_BWD_SLOTS = 1024                           # CHAP_ACT_BWD_SLOTS
LOSS_SLOTS, CHANSUM_SLOTS, L2NORM_SLOTS = 512, 512, 256
PACK_CONV_FWD, PACK_CONV_DGRAD, PACK_DECONV_FWD, PACK_DECONV_DGRAD, PACK_DOWN_DGRAD = range(5)

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class Src(C.Structure):
    _fields_ = [("ptr", _vp), ("scale", _vp), ("shift", _vp), ("keep", _vp), ("chan_mul", _vp),
                ("C", _i32), ("ld", _i32), ("coff", _i32), ("act", _i32), ("slope", _f32), ("keep_scale", _f32)]


class ConvParams(C.Structure):
    _fields_ = [("src", Src * 2), ("nsrc", _i32), ("combine", _i32),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("ID", _i32), ("IH", _i32), ("IW", _i32),
                ("ksize", _i32), ("stride", _i32), ("dims", _i32),
                ("wpacked", _vp), ("bias", _vp), ("out", _vp),
                ("Cout", _i32), ("out_ld", _i32), ("out_coff", _i32), ("out_mode", _i32), ("out_Cn", _i32),
                ("out_planar", _i32), ("out_f32", _i32),
                ("stats", _vp), ("stats_shift", _vp), ("dtype", _i32), ("out2_from", _i32), ("out2", _vp)]


class PackParams(C.Structure):
    _fields_ = [("w", _vp), ("out", _vp), ("kind", _i32), ("Cin", _i32), ("Cout", _i32), ("taps", _i32), ("dtype", _i32)]


class PackEntry(C.Structure):
    _fields_ = [("w", _vp), ("out", _vp), ("kind", _i32), ("Cin", _i32), ("Cout", _i32), ("taps", _i32), ("dtype", _i32),
                ("KC", _i32), ("GPT", _i32), ("NP", _i32), ("STEPS", _i32), ("nchunks", _i32), ("ntiles", _i32),
                ("Cn_logical", _i32), ("Ck_real", _i32), ("total", _i64)]


class ConvC1Params(C.Structure):
    _fields_ = [("x", _vp), ("w", _vp), ("bias", _vp), ("out", _vp), ("stats", _vp), ("stats_shift", _vp),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("dims", _i32), ("Cout", _i32), ("dtype", _i32)]


class ConvC1BwdParams(C.Structure):
    _fields_ = [("g", _vp), ("w", _vp), ("x", _vp), ("dx", _vp), ("dw", _vp), ("db", _vp), ("ws", _vp),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("dims", _i32), ("Cout", _i32), ("dtype", _i32)]


class WgradParams(C.Structure):
    _fields_ = [("a", Src * 2), ("na", _i32), ("combine", _i32), ("b", Src),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("ID", _i32), ("IH", _i32), ("IW", _i32),
                ("ksize", _i32), ("stride", _i32), ("dims", _i32),
                ("dw", _vp), ("s_tap", _i64), ("s_kc", _i64), ("s_kn", _i64), ("kc_valid", _i32), ("kn_valid", _i32),
                ("db", _vp), ("ws", _vp), ("ws_bytes", _sz), ("dtype", _i32)]


class BnFinalizeParams(C.Structure):
    _fields_ = [("stats", _vp), ("stats_shift", _vp), ("Clog", _i32), ("gamma", _vp), ("beta", _vp),
                ("running_mean", _vp), ("running_var", _vp), ("num_batches_tracked", _vp),
                ("scale", _vp), ("shift", _vp), ("mean", _vp), ("invstd", _vp),
                ("C", _i32), ("count", _f32), ("eps", _f32), ("momentum", _f32)]


class BnEvalParams(C.Structure):
    _fields_ = [("gamma", _vp), ("beta", _vp), ("running_mean", _vp), ("running_var", _vp),
                ("scale", _vp), ("shift", _vp), ("C", _i32), ("eps", _f32)]


class ActBwdParams(C.Structure):
    _fields_ = [("g", _vp * 3), ("g_ld", _i32 * 3), ("g_coff", _i32 * 3), ("ng", _i32),
                ("g_pool", _vp), ("pool_idx", _vp), ("r", Src),
                ("mean", _vp), ("invstd", _vp), ("gamma", _vp), ("sums", _vp), ("gout", _vp),
                ("dgamma", _vp), ("dbeta", _vp),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("bn", _i32), ("count", _f32), ("dtype", _i32)]


class PoolParams(C.Structure):
    _fields_ = [("r", Src), ("out", _vp), ("idx", _vp), ("N", _i32), ("H", _i32), ("W", _i32), ("dtype", _i32), ("D", _i32)]


class UpsampleParams(C.Structure):
    _fields_ = [("r", Src), ("out", _vp), ("out_ld", _i32), ("out_coff", _i32),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("dims", _i32), ("dtype", _i32), ("half_pixel", _i32)]


class UpsampleBwdParams(C.Structure):
    _fields_ = [("g", _vp), ("g_ld", _i32), ("g_coff", _i32), ("out", _vp),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("C", _i32), ("dims", _i32), ("dtype", _i32)]


class PlanarToClParams(C.Structure):
    _fields_ = [("in_", _vp), ("out", _vp), ("N", _i32), ("C", _i32), ("P", _i32), ("out_ld", _i32), ("out_coff", _i32), ("Cpad", _i32), ("dtype", _i32)]


class ChanSumParams(C.Structure):
    _fields_ = [("r", Src), ("out", _vp), ("npix", _i64), ("pix_per_sample", _i64), ("dtype", _i32), ("ws", _vp)]


class ClToPlanarParams(C.Structure):
    _fields_ = [("r", Src), ("out", _vp), ("N", _i32), ("P", _i32), ("dtype", _i32)]


class MixLossParams(C.Structure):
    _fields_ = [("logits", _vp), ("target_a", _vp), ("target_b", _vp), ("mask", _vp), ("w_a", _f32), ("w_b", _f32),
                ("acc", _vp), ("loss", _vp), ("dlogits", _vp), ("gscale", _f32), ("accumulate", _i32),
                ("N", _i32), ("C", _i32), ("P", _i32), ("smooth", _f32), ("k_dice", _f32), ("k_ce", _f32), ("gscale_dev", _vp)]


class PseudoParams(C.Structure):
    _fields_ = [("logits1", _vp), ("logits2", _vp), ("soft1", _vp), ("soft2", _vp), ("arg1", _vp), ("arg2", _vp),
                ("knowledge", _vp), ("N", _i32), ("C", _i32), ("P", _i32)]


class KlParams(C.Structure):
    _fields_ = [("logits", _vp * 2), ("target", _vp * 2), ("loss", _vp), ("dlogits", _vp * 2),
                ("gscale", _f32), ("gscale_dev", _vp), ("N", _i32), ("C", _i32), ("P", _i32), ("mode", _i32), ("ws", _vp)]


class EnsembleParams(C.Structure):
    _fields_ = [("logits1", _vp), ("logits2", _vp), ("prob", _vp), ("label", _vp), ("N", _i32), ("C", _i32), ("P", C.c_int64), ("mode", _i32)]


class WindowAccParams(C.Structure):
    _fields_ = [("logits", _vp), ("origins", _vp), ("score", _vp), ("cnt", _vp), ("npatch", _i32), ("C", _i32),
                ("pw", _i32), ("ph", _i32), ("pd", _i32), ("W", _i32), ("H", _i32), ("D", _i32)]


class WindowFinParams(C.Structure):
    _fields_ = [("score", _vp), ("cnt", _vp), ("label", _vp), ("C", _i32), ("P", C.c_int64)]


class L2NormParams(C.Structure):
    _fields_ = [("in_", _vp), ("out", _vp), ("N", _i32), ("P", _i32), ("eps", _f32), ("ws", _vp)]


class AxpyParams(C.Structure):
    _fields_ = [("x", _vp), ("d", _vp), ("mask", _vp), ("out", _vp), ("alpha", _f32), ("sign", _i32), ("n", _i64)]


class RandParams(C.Structure):
    _fields_ = [("out", _vp), ("seed", C.c_uint64), ("seed_dev", _vp), ("n", _i64), ("lo", _f32), ("hi", _f32)]


class KeepMaskParams(C.Structure):
    _fields_ = [("keep", _vp), ("seed", C.c_uint64), ("seed_dev", _vp), ("n", _i64), ("p", _f32)]


class ChanMaskParams(C.Structure):
    _fields_ = [("mul", _vp), ("seed", C.c_uint64), ("seed_dev", _vp), ("n", _i64), ("p", _f32)]


class SampleChanSumParams(C.Structure):
    _fields_ = [("r", Src), ("partial", _vp), ("N", _i32), ("nchunk", _i32), ("pix_per_sample", _i64), ("dtype", _i32)]


class ChannelDropParams(C.Structure):
    _fields_ = [("pool_partial", _vp), ("grad_sim", _vp), ("u1", _vp), ("u2", _vp), ("mul1", _vp), ("mul2", _vp),
                ("probs_out", _vp), ("inv_npix", _f32), ("nchunk", _i32), ("B", _i32), ("U", _i32), ("C", _i32),
                ("mode", _i32), ("comp", _i32), ("branch", _i32), ("prob_kind", _i32)]


class FoldParams(C.Structure):
    _fields_ = [("g", _vp), ("mul", _vp), ("out", _vp), ("B", _i32), ("U", _i32), ("C", _i32), ("ld", _i32), ("coff", _i32),
                ("pix_per_sample", _i64), ("dtype", _i32)]


class BoxMixParams(C.Structure):
    _fields_ = [("a", _vp), ("b", _vp), ("out", _vp), ("box", _vp), ("N", _i32), ("H", _i32), ("W", _i32), ("is_i64", _i32), ("D", _i32)]


class BoxMaskParams(C.Structure):
    _fields_ = [("mask", _vp), ("box", _vp), ("N", _i32), ("H", _i32), ("W", _i32), ("D", _i32)]


class LccParams(C.Structure):
    _fields_ = [("labels", _vp), ("out", _vp), ("ws", _vp), ("N", _i32), ("H", _i32), ("W", _i32), ("num_classes", _i32), ("D", _i32)]


class DiffMaskParams(C.Structure):
    _fields_ = [("p1", _vp), ("p2", _vp), ("knowledge", _vp), ("out", _vp), ("pooled_ws", _vp),
                ("N", _i32), ("H", _i32), ("W", _i32), ("scale", _i32), ("topk", _f32)]


class GradSimParams(C.Structure):
    _fields_ = [("gl", _vp), ("gu", _vp), ("score", _vp), ("C", _i32), ("K", _i32), ("ema", _f32)]


class SgdParams(C.Structure):
    _fields_ = [("param", _vp), ("grad", _vp), ("grad2", _vp), ("mom", _vp), ("lr", _vp), ("momentum", _f32), ("weight_decay", _f32),
                ("grad_scale", _f32), ("n", _i64), ("zero_grad", _i32)]


_SIGS = {  # name -> (restype, params struct or None)
    "chap_conv_fwd": ConvParams, "chap_pack_weights": PackParams, "chap_conv_c1_fwd": ConvC1Params,
    "chap_conv_c1_bwd": ConvC1BwdParams, "chap_wgrad": WgradParams, "chap_bn_finalize": BnFinalizeParams,
    "chap_bn_eval_affine": BnEvalParams, "chap_act_bwd_reduce": ActBwdParams, "chap_act_bwd_apply": ActBwdParams,
    "chap_act_pool2": PoolParams, "chap_upsample2x": UpsampleParams, "chap_upsample2x_bwd": UpsampleBwdParams,
    "chap_planar_to_cl": PlanarToClParams, "chap_channel_sum": ChanSumParams, "chap_cl_to_planar": ClToPlanarParams,
    "chap_ensemble_argmax": EnsembleParams, "chap_window_accumulate": WindowAccParams, "chap_window_finalize": WindowFinParams,
    "chap_mix_loss_fwd": MixLossParams, "chap_mix_loss_bwd": MixLossParams, "chap_pseudo_block": PseudoParams,
    "chap_kl_fwd_bwd": KlParams, "chap_l2_normalize": L2NormParams, "chap_perturb": AxpyParams,
    "chap_rand_uniform": RandParams, "chap_keep_mask": KeepMaskParams, "chap_chan_mask": ChanMaskParams,
    "chap_box_mix": BoxMixParams, "chap_box_mask": BoxMaskParams, "chap_largest_cc": LccParams,
    "chap_diff_mask": DiffMaskParams, "chap_sgd_step": SgdParams,
    "chap_sample_channel_sum": SampleChanSumParams, "chap_channel_drop": ChannelDropParams,
    "chap_fold_perturbed": FoldParams, "chap_grad_sim": GradSimParams,
}
_SIZE_FNS = {"chap_pack_size": PackParams, "chap_conv_c1_bwd_ws": ConvC1BwdParams, "chap_wgrad_ws": WgradParams,
             "chap_lcc_ws": LccParams}

_lib = None


ABI_VERSION = 7            # CHAP_ABI_VERSION of include/chap_hip.h this binding mirrors (checked when the library is loaded)


class ChapError(RuntimeError):
    pass


def lib():
    """Load libchap_hip.so once.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ChapError("libchap_hip.so not built (%s missing): run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C chap_amd/csrc`.  chap_amd has no CPU fallback." % LIB_PATH)
        # The library is linked against the system ROCm runtime while PyTorch ships its own copy: the HIP runtime
        # PyTorch uses has to be initialised BEFORE this library is loaded (observed on the MI355X box: loaded the
        # other way round, the first kernel launch fails with "no ROCm-capable device is detected").
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()               # an error here is the load-order problem itself: let it propagate
            torch.cuda.current_stream()
        L = C.CDLL(LIB_PATH)
        L.chap_last_error.restype = C.c_char_p
        L.chap_abi_version.restype = C.c_int
        got = L.chap_abi_version()
        if got != ABI_VERSION:
            raise ChapError("libchap_hip.so has ABI version %d, this binding mirrors include/chap_hip.h version %d: rebuild "
                            "(`make -C chap_amd/csrc`) -- a stale library would be driven through mismatching structs" % (got, ABI_VERSION))
        _lib = L
    return _lib


_bound = {}


def _fn(name):
    """Bind an entry point on first use (argtypes from the tables above)."""
    fn = _bound.get(name)
    if fn is None:
        L = lib()
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise ChapError("libchap_hip.so does not export %s (stale build?)" % name)
        if name in _SIGS:
            fn.restype, fn.argtypes = C.c_int, [C.POINTER(_SIGS[name]), _vp]
        else:
            fn.restype, fn.argtypes = _sz, [C.POINTER(_SIZE_FNS[name])]
        _bound[name] = fn
    return fn


def pack_describe(params):
    """chap_pack_describe: host-side, fills a PackEntry for chap_pack_multi."""
    L = lib()
    L.chap_pack_describe.restype = C.c_int
    L.chap_pack_describe.argtypes = [C.POINTER(PackParams), C.POINTER(PackEntry)]
    e = PackEntry()
    rc = L.chap_pack_describe(C.byref(params), C.byref(e))
    if rc != 0:
        raise ChapError("chap_pack_describe failed (%d): %s" % (rc, L.chap_last_error().decode()))
    return e


def pack_multi(entries_dev_ptr, n, max_total, stream):
    L = lib()
    L.chap_pack_multi.restype = C.c_int
    L.chap_pack_multi.argtypes = [_vp, _i32, _i64, _vp]
    rc = L.chap_pack_multi(_vp(entries_dev_ptr), n, max_total, _vp(stream))
    if rc != 0:
        raise ChapError("chap_pack_multi failed (%d): %s" % (rc, L.chap_last_error().decode()))


_holders = []       # the `held` lists of the open group regions, see hold()


def _drop_holder(h):
    for i, o in enumerate(_holders):
        if o is h:
            del _holders[i]
            return


class group:
    """`with group(stream) as g: ...lane 0...; g.next_lane(); ...lane 1...`: chap_group_begin / _next_lane / _end (chap_hip.h):
    the launches of the lanes are recorded and issued together, same-shaped ones as one grid.  `enabled=False` (or one lane
    only) makes it a no-op wrapper, which is how the executor switches grouping off (CHAP_GROUP=0)."""
    launched = 0            # grids issued by all regions so far (diagnostics / tests)
    held = None             # tensors allocated inside the open region (see hold())

    def __init__(self, stream, enabled=True):
        self.stream, self.enabled = stream, enabled

    def __enter__(self):
        if self.enabled:
            group.held = []
            _holders.append(group.held)
            L = lib()
            L.chap_group_begin.argtypes = [_vp]
            rc = L.chap_group_begin(_vp(self.stream))
            if rc != 0:
                raise ChapError("chap_group_begin failed (%d): %s" % (rc, L.chap_last_error().decode()))
        return self

    def next_lane(self):
        if self.enabled:
            rc = lib().chap_group_next_lane()
            if rc != 0:
                raise ChapError("chap_group_next_lane failed (%d): %s" % (rc, lib().chap_last_error().decode()))

    def __exit__(self, et, ev, tb):
        if self.enabled:
            if et is not None:                  # the body failed: drop what was recorded (its buffers are released with `held`), nothing is launched
                lib().chap_group_cancel()
                _drop_holder(group.held)
                group.held = None
                return False
            rc = lib().chap_group_end()
            _drop_holder(group.held)
            group.held = None
            if rc < 0:
                raise ChapError("chap_group_end failed (%d): %s" % (rc, lib().chap_last_error().decode()))
            group.launched += rc
        return False


def hold(t):
    """Inside a group region the lanes' kernels run CONCURRENTLY, later than the Python code that allocated their buffers: a
    temporary that Python drops (a workspace local to a wrapper, a gradient tensor consumed by the launches of its op) would go
    back to the caching allocator and could be handed to the NEXT lane of the same region -- two lanes of one grid writing the
    same memory.  Every tensor allocated on this path goes through hold(): kept alive until chap_group_end has issued the launches
    (stream order protects it from then on).  No-op outside a region."""
    for h in _holders:
        h.append(t)
    return t


def hold_empty(*a, **k):
    import torch
    return hold(torch.empty(*a, **k))


def hold_empty_like(*a, **k):
    import torch
    return hold(torch.empty_like(*a, **k))


def call(name, params, stream):
    rc = _fn(name)(C.byref(params), _vp(stream))
    if rc != 0:
        raise ChapError("%s failed (%d): %s" % (name, rc, lib().chap_last_error().decode()))


def size_of(name, params):
    return int(_fn(name)(C.byref(params)))
