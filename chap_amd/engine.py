"""Static-graph executor for the CHAP networks: hand-scheduled forward and backward passes made
only of libchap_hip.so kernel launches (no torch compute ops, no autograd inside).

A network is a `Program`: an ordered list of ops over named values.  A value is a *lazy
activation* (ops.Lazy): the raw conv output plus the BatchNorm affine / LeakyReLU / dropout that
its consumers apply while loading it.  The backward pass walks the program in reverse, keeping for
every value the list of gradient contributions w.r.t. its ACTIVATED form; `chap_act_bwd_*` turns
them into the gradient w.r.t. the raw tensor (BatchNorm backward fused), `chap_wgrad` and
`chap_conv_fwd` (with *_DGRAD packings) do the rest.  Parameter gradients are accumulated straight
into the model's flat gradient buffer.

Reference semantics followed: code/networks/unet.py:44-292, code/networks/vnet.py:8-238.
"""
import torch

from . import _lib as L
from . import ops
from .ops import Lazy

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class Op:
    """kind: 'c1' | 'conv' | 'down' | 'deconv' | 'pool' | 'up'."""

    def __init__(self, kind, out, srcs, **kw):
        self.kind, self.out, self.srcs = kind, out, list(srcs)
        self.w = kw.get("w")              # weight key (state dict)
        self.b = kw.get("b")              # bias key
        self.bn = kw.get("bn")            # BatchNorm prefix or None
        self.ksize = kw.get("ksize", 3)
        self.cin = kw.get("cin")
        self.cout = kw.get("cout")
        self.combine = kw.get("combine", 0)
        self.slope = kw.get("slope", 0.0)  # activation after BN (when bn is set)
        self.head = kw.get("head", False)  # planar fp32 logits
        self.drop = kw.get("drop")         # (site, p, 'elem' | 'chan') applied to the OUTPUT value
        self.branch = kw.get("branch", 0)            # 0 = shared trunk (encoder); k > 0 = k-th decoder: decoders run on
                                                    # parallel streams (they only meet again in the encoder's backward)
        self.inorm = kw.get("inorm", False)          # InstanceNorm (no affine) + ReLU after the conv: batch-of-one statistics
        self.half_pixel = kw.get("half_pixel", False)  # 'up': align_corners=False


class Program:
    def __init__(self, dims, ops_, heads, in_name="x"):
        self.dims, self.ops, self.heads, self.in_name = dims, ops_, heads, in_name
        self.consumers = {}
        for op in ops_:
            for s in op.srcs:
                self.consumers.setdefault(s, []).append(op)


def _taps(op, dims):
    if op.kind in ("down", "deconv"):
        return 2 ** dims
    return op.ksize ** dims


def _op_sig(op):
    """What two ops must share to resolve to the same kernel launches (same instance, grid, call sequence)."""
    return (op.kind, op.ksize, op.cin, op.cout, bool(op.bn), op.head, op.combine, op.inorm, op.half_pixel, len(op.srcs),
            None if not op.drop else (op.drop[1], op.drop[2]), op.slope)


def zip_branches(a_ops, b_ops):
    """Align the op lists of two decoders (program order): [(op_a, op_b)] for same-shaped ops -- they become the two lanes of
    one grouped launch region -- and [(op,)] for the ones without a partner (decoder 1: 1x1 conv + up-sampling, decoder 2:
    transposed conv).  Each list keeps its order, so every op still runs after the ops of its own branch it depends on."""
    out, i, j = [], 0, 0
    while i < len(a_ops) or j < len(b_ops):
        if i < len(a_ops) and j < len(b_ops) and _op_sig(a_ops[i]) == _op_sig(b_ops[j]):
            out.append((a_ops[i], b_ops[j]))
            i, j = i + 1, j + 1
            continue
        best = None
        for di in range(0, min(4, len(a_ops) - i) + 1):
            for dj in range(0, min(4, len(b_ops) - j) + 1):
                if (di or dj) and i + di < len(a_ops) and j + dj < len(b_ops) and _op_sig(a_ops[i + di]) == _op_sig(b_ops[j + dj]):
                    if best is None or di + dj < best[0] + best[1]:
                        best = (di, dj)
        if best is None:
            best = (len(a_ops) - i, len(b_ops) - j)
        out += [(o,) for o in a_ops[i:i + best[0]]] + [(o,) for o in b_ops[j:j + best[1]]]
        i, j = i + best[0], j + best[1]
    return out


def drive(gen, stream):
    """Run a step generator (Executor.forward_steps / backward_steps): the lanes of a step -- the same-shaped ops of the two decoders --
    are enqueued inside ONE grouped launch region (their kernels become one grid), a single lane directly.  Returns the generator's
    return value."""
    while True:
        try:
            lanes = next(gen)
        except StopIteration as e:
            return e.value
        if not lanes:
            continue
        if len(lanes) == 1:
            lanes[0]()
        else:
            with L.group(stream) as region:
                for k, f in enumerate(lanes):
                    if k:
                        region.next_lane()
                    f()


def side_branch():
    """CHAP_SIDE_DECODER (lab / A-B switch, default 1): which decoder of a forking pass runs on the forked stream (the other stays on the pass's
    own).  Round 4, two runs each: 2D 6.413 / 6.421 ms with decoder 2 on the fork, 6.385 / 6.407 with decoder 1; 3D 14.82 / 14.70 vs 14.66 / 14.62
    (profiles/r04_side_decoder_ab.log)."""
    import os
    return 2 if os.environ.get("CHAP_SIDE_DECODER", "1") == "2" else 1


def first_conv_direct():
    """CHAP_C1_DIRECT (lab / A-B switch, default 1): the first conv (one input channel) of a bf16 pass on its own kernel (csrc/conv_c1_mfma.h) instead
    of the generic conv over the image zero-padded to 16 channels."""
    import os
    return os.environ.get("CHAP_C1_DIRECT", "1") != "0"


def defer_decoder_wgrad():
    """CHAP_DEFER_WGRAD (lab / A-B switch, default 1): in a backward pass that forks its second decoder, the decoders' weight gradients are issued on the
    forked stream BEHIND both decoders' chains -- they are not on the path to the join with the trunk, and the forked stream used to idle during the trunk's
    backward.  Round 4, three pairs: 2D 6.24 / 6.22 / 6.26 -> 6.09 / 6.09 / 6.11 ms, 3D 14.68 / 14.61 / 14.69 -> 14.64 / 14.60 / 14.51 ms.  Tried on top and
    removed: the trunk's weight gradients on that stream too, one event per layer (6.6 / 15.3 ms); pass B (grouped decoders on one stream) borrowing the early
    VAT pass's idle stream for its decoders' weight gradients (7.65 / 17.3 ms: one more chain for the graph executor's queues); the forked decoder's own
    weight gradients right behind its chain instead of behind both (no difference) -- profiles/r04_defer_wgrad_ab.log."""
    import os
    return os.environ.get("CHAP_DEFER_WGRAD", "1") != "0"


def split_concat_gradient():
    """CHAP_SPLIT_CONCAT (lab / A-B switch, default 1): the input gradient of a concat layer as two dense tensors (chap_conv_params.out2)."""
    import os
    return os.environ.get("CHAP_SPLIT_CONCAT", "1") != "0"


def grouping_mode():
    """CHAP_GROUP (lab / A-B switch): 0 = never group (round 2: decoders back to back where a pass cannot fork a second stream), 1 (default) =
    group the two decoders' same-shaped layers in the passes that cannot fork one.  (Round 3 also measured grouping in EVERY pass, the
    decoders as parallel graph branches on one stream and the weight gradients as graph leaves: all slower, removed -- DESIGN.md section 5.)"""
    import os
    return int(os.environ.get("CHAP_GROUP", "1"))


class Saved:
    """What one forward pass leaves behind for its backward pass."""

    def __init__(self):
        self.vals = {}       # name -> Lazy
        self.dims = {}       # name -> (D, H, W)
        self.bnstat = {}     # bn prefix -> (mean, invstd, count)
        self.pool_idx = {}   # pooled value name -> idx tensor
        self.x = None
        self.xpad = None     # bf16 mode: the input zero-padded to 16 channels (channel-last)
        self.train = False
        self.tables = None   # perturbed pass: branch -> value table of that decoder (trunk values overlaid)
        self.fold = {}       # perturbed pass: branch -> {trunk value name: chan_mul [B + U, C] or None}
        self.n_dec = 0       # perturbed pass: the decoders' batch size B + U


class Executor:
    """Runs a Program for one nn.Module (which owns parameters, buffers and the flat grad buffer)."""

    def __init__(self, module, program):
        self.m, self.prog = module, program
        self._packed = {}          # dtype -> {bufs, table, ...}: packed weights + chap_pack_multi entry table
        self._ident = {}           # C -> (ones, zeros) for InstanceNorm (no affine)
        self._sides = {}           # parent stream -> forked stream for the second decoder
        self._capture_sides = {}   # same, for use inside a graph capture (registered by the owner of the capture)
        self.has_inorm = any(op.inorm for op in program.ops)

    # ---------------------------------------------------------------- parameters
    def _sd(self):
        return self.m._tensors()   # name -> tensor (params and buffers, fp32, on device)

    def _side_stream(self, parent):
        """Forked stream for the second decoder, one per parent stream (pass B on the main stream and the VAT
        branch on its own stream may both be inside this executor).  Eager mode only: a second level of
        fork/join inside a captured HIP graph crashes hipStreamEndCapture on ROCm 7.2, so under capture the
        decoders run back to back (the pass-B / VAT fork of ChapStep is the one level that is captured)."""
        key = parent.cuda_stream
        if torch.cuda.is_current_stream_capturing():
            # only a stream that the caller forked from the capture's ORIGIN stream beforehand (ChapStep does, flat)
            return self._capture_sides.get(key)
        st = self._sides.get(key)
        if st is None:
            if len(self._sides) >= 8:
                return None
            st = torch.cuda.Stream(device=parent.device)
            self._sides[key] = st
        return st

    def _static_dims(self, D, H, W):
        """value name -> (d, h, w) for an input of (D, H, W), without running anything (same rules as run_op)."""
        dims, three = {self.prog.in_name: (D, H, W)}, self.prog.dims == 3
        for op in self.prog.ops:
            d, h, w = dims[op.srcs[0]] if op.srcs else (D, H, W)
            if op.kind in ("pool", "down"):
                dims[op.out] = (d // 2 if three else d, h // 2, w // 2)
            elif op.kind in ("up", "deconv"):
                dims[op.out] = (2 * d if three else d, 2 * h, 2 * w)
            else:
                dims[op.out] = (d, h, w)
        return dims

    def _zipped(self):
        """The aligned schedule of the two decoder branches (zip_branches), built once per program."""
        z = getattr(self, "_zip", None)
        if z is None:
            b1 = [op for op in self.prog.ops if op.branch == 1]
            b2 = [op for op in self.prog.ops if op.branch == 2]
            z = self._zip = zip_branches(b1, b2)
        return z

    def _pack_kinds(self, op):
        if op.kind == "conv":
            return (L.PACK_CONV_FWD, L.PACK_CONV_DGRAD)
        if op.kind == "down":
            return (L.PACK_CONV_FWD, L.PACK_DOWN_DGRAD)
        if op.kind == "deconv":
            return (L.PACK_DECONV_FWD, L.PACK_DECONV_DGRAD)
        if op.kind == "c1":
            # dL/dx of the first layer (VAT) runs on the MFMA conv kernel (16 -> 1 channel); in bf16 mode the forward
            # does too, on the input zero-padded to 16 channels (the scalar 1-channel kernel is load-issue bound)
            return (L.PACK_CONV_DGRAD, L.PACK_CONV_FWD)
        return ()

    def _build_pack_table(self, dtype, sd):
        """Persistent packed buffers + the device-side entry table for chap_pack_multi (one launch per step)."""
        import ctypes as C
        entries, bufs, max_total = [], {}, 0
        dev = next(iter(sd.values())).device
        for op in self.prog.ops:
            for kind in self._pack_kinds(op):
                w = sd[op.w]
                cin, cout = (w.shape[0], w.shape[1]) if op.kind == "deconv" else (w.shape[1], w.shape[0])
                p = L.PackParams()
                p.w, p.kind, p.Cin, p.Cout, p.taps = w.data_ptr(), kind, cin, cout, _taps(op, self.prog.dims)
                p.dtype = L.F32 if dtype == torch.float32 else L.BF16
                buf = L.hold_empty(L.size_of("chap_pack_size", p), dtype=torch.uint8, device=dev)
                p.out = buf.data_ptr()
                e = L.pack_describe(p)
                entries.append(e)
                bufs[(op.w, kind)] = buf
                max_total = max(max_total, int(e.total))
        arr = (L.PackEntry * len(entries))(*entries)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        return dict(bufs=bufs, table=host.to(dev), n=len(entries), max_total=max_total, key=sd[self.prog.ops[0].w].data_ptr())

    def _ensure_packed(self, dtype, sd):
        tab = self._packed.get(dtype)
        if tab is None or tab["key"] != sd[self.prog.ops[0].w].data_ptr():
            tab = self._build_pack_table(dtype, sd)
            self._packed[dtype] = tab
            tab["version"] = None
        ver = self.m._params_version()
        if tab["version"] != ver:
            L.pack_multi(tab["table"].data_ptr(), tab["n"], tab["max_total"], torch.cuda.current_stream().cuda_stream)
            tab["version"] = ver
        return tab

    def _pack(self, op, kind, dtype, sd):
        return self._ensure_packed(dtype, sd)["bufs"][(op.w, kind)]

    # ---------------------------------------------------------------- forward
    def forward(self, x, **kw):
        """x: fp32 [N, in_chns, *spatial] contiguous. Returns (list of planar fp32 logits, Saved|None, extras):
        extras = the activated values named in `want`, materialised as planar fp32 [N, C, *spatial]."""
        return drive(self.forward_steps(x, **kw), torch.cuda.current_stream().cuda_stream)

    def forward_steps(self, x, *, train, dtype, save, update_stats=True, drop_masks=None, rng=None, want=(), perturb=None):
        """The forward pass as a generator of STEPS: each `yield` hands the driver (engine.drive) the lanes of one step -- a list
        of callables that enqueue the step's kernels, one per lane (one op; or the same-shaped ops of the two decoders, which the
        driver issues as ONE grouped launch region: chap_hip.h, chap_group_*) -- and the generator's return value is forward()'s."""
        prog, sd, dims = self.prog, self._sd(), self.prog.dims
        dev = x.device
        N = x.shape[0]
        sp = tuple(x.shape[2:])
        D, H, W = ((1,) + sp) if dims == 2 else sp
        if self.has_inorm and (save or N != 1):
            raise NotImplementedError("chap_amd: InstanceNorm networks (unet_3D) are inference-only and run one sample per pass")
        S = Saved()
        S.train, S.x = train, x
        vals, vdims = S.vals, S.dims
        vdims[prog.in_name] = (D, H, W)
        # one fp32 arena for all BN statistics (per-block partial slots: nothing to zero) / affines of this pass
        def clog(op):       # logical output channels of the conv = floats per statistics slot row
            return op.cout * (2 ** dims) if op.kind == "deconv" else op.cout
        nbn = sum(op.cout for op in prog.ops if op.bn or op.inorm)
        nstat = sum(ops.stats_size(clog(op)) for op in prog.ops if (op.bn and train) or op.inorm)
        arena = L.hold_empty(nstat + nbn * 4, dtype=torch.float32, device=dev)
        apos = 0
        # shift of the statistics' moments (sum(x - c), sum((x - c)^2)): a pass-private snapshot of the running means, so
        # that the conv and its finalize see the same c whatever another stream's pass does to the running statistics
        rm_flat, rm_off = self.m._running_mean_flat() if train else (None, None)
        shift_snap = None
        if rm_flat is not None:
            shift_snap = self.m._shift_hold                   # the iteration's snapshot (ChapNet.hold_stat_shift), if one is held
            if shift_snap is None:
                # sized by the module's running-mean buffer, not by the program's BN channel count (a module may own BatchNorm
                # layers that this program never runs)
                shift_snap = L.hold_empty_like(rm_flat)
                shift_snap.copy_(rm_flat)

        def take(n):
            nonlocal apos
            t = arena[apos:apos + n]
            apos += n
            return t

        outs = {}
        cur_stream = torch.cuda.current_stream()
        branches = sorted({op.branch for op in prog.ops})
        # two decoders: on a second stream where this pass may fork one (eager; under capture only from the capture's origin
        # stream), otherwise in lockstep with grouped launches
        side = self._side_stream(cur_stream) if len(branches) > 2 else None
        zipped = self._zipped() if (side is None and len(branches) == 3 and grouping_mode() != 0) else None

        # Dropout seeds are drawn HERE, in program order: the order in which the ops are ISSUED depends on whether this pass
        # may fork its second decoder (eager / captured, which stream), and a seed must not (round 2: the early VAT pass drew
        # other masks whenever its stream happened to be the capture's origin stream)
        drop_seed = {}
        if train and drop_masks is None:
            for op_ in prog.ops:
                if op_.drop:
                    drop_seed[id(op_)] = rng.next_seed()
        # The element keep masks of the trunk (nn.Dropout of the five encoder ConvBlocks) depend on nothing but their seeds: they are
        # generated up front, as the lanes of one step (same-sized ones share a grid), instead of one launch on the chain behind each conv
        pre_keep = {}
        if train:
            elem = [op_ for op_ in prog.ops if op_.drop and op_.drop[2] == "elem" and op_.branch == 0 and op_.kind in ("c1", "conv")] if drop_masks is None else []
            sdims = self._static_dims(D, H, W) if elem else None

            def make_keep(op_):
                keep = L.hold_empty((N,) + sdims[op_.out] + (op_.cout,), dtype=torch.uint8, device=dev)
                ops.keep_mask(keep, drop_seed[id(op_)], op_.drop[1], seed_dev=rng.seed_dev)
                pre_keep[id(op_)] = keep
            yield [lambda op_=op_: make_keep(op_) for op_ in elem]        # (a step of every training-mode pass, possibly without lanes: passes driven together stay aligned)

        def run_op(op, V=vals, n=N):      # V / n: value table and batch size (the decoders of a perturbed pass see their own)
            nonlocal apos
            k = op.kind
            if k == "pool":
                src = V[op.srcs[0]]
                d, h, w = vdims[op.srcs[0]]
                od = d // 2 if dims == 3 else 1
                out = L.hold_empty(n, od, h // 2, w // 2, src.C, dtype=dtype, device=dev)
                idx = L.hold_empty(n, od, h // 2, w // 2, src.C, dtype=torch.uint8, device=dev) if save else None
                ops.act_pool2(src, out, idx, dims=dims)
                V[op.out], vdims[op.out] = Lazy(out), (od, h // 2, w // 2)
                if save:
                    S.pool_idx[op.out] = idx
                return
            if k == "up":
                src = V[op.srcs[0]]
                d, h, w = vdims[op.srcs[0]]
                od = 2 * d if dims == 3 else d
                out = L.hold_empty(n, od, 2 * h, 2 * w, src.C, dtype=dtype, device=dev)
                ops.upsample2x(src, out, dims=dims, half_pixel=op.half_pixel)
                V[op.out], vdims[op.out] = Lazy(out), (od, 2 * h, 2 * w)
                return
            # ---- convolutions
            stats = take(ops.stats_size(clog(op))) if ((op.bn and train) or op.inorm) else None
            sshift = None
            if stats is not None and op.bn and shift_snap is not None:
                o = rm_off[op.bn]
                sshift = shift_snap[o:o + op.cout]
            bias = sd[op.b] if op.b else None
            if k == "c1":
                gd = (D, H, W)
                out = L.hold_empty(n, D, H, W, op.cout, dtype=dtype, device=dev)
                if dtype == torch.bfloat16 and x.shape[1] == 1 and op.cout == 16 and first_conv_direct():
                    # bf16, one input channel: the taps as the K dimension of one MFMA per 16 pixels (csrc/conv_c1_mfma.h) -- no zero-padded copy of the
                    # image; the backward pass builds it when (and only when) it needs the weight gradient
                    ops.conv_c1_fwd(x.view(n, D, H, W), sd[op.w], bias, out, dims=dims, stats=stats, stats_shift=sshift)
                elif dtype == torch.bfloat16 or x.shape[1] > 1:     # (in_chns > 1: the padded MFMA path in fp32 too; the scalar kernels are 1-channel)
                    xpad = L.hold_empty(n, D, H, W, 16, dtype=dtype, device=dev)
                    ops.planar_to_cl(x, xpad, cpad=16)
                    S.xpad = xpad
                    wp = self._pack(op, L.PACK_CONV_FWD, dtype, sd)
                    ops.conv_fwd([Lazy(xpad)], wp, bias, op.cout, out, grid=(n,) + gd, in_dims=gd, ksize=3, stride=1, dims=dims,
                                 stats=stats, stats_shift=sshift)
                else:
                    ops.conv_c1_fwd(x.view(n, D, H, W), sd[op.w], bias, out, dims=dims, stats=stats, stats_shift=sshift)
            else:
                srcs = [V[s] for s in op.srcs]
                sd_, sh_, sw_ = vdims[op.srcs[0]]
                if k == "conv":
                    gd, ind, ks, st, kind = (sd_, sh_, sw_), (sd_, sh_, sw_), op.ksize, 1, L.PACK_CONV_FWD
                elif k == "down":
                    gd = ((sd_ // 2) if dims == 3 else sd_, sh_ // 2, sw_ // 2)
                    ind, ks, st, kind = (sd_, sh_, sw_), 2, 2, L.PACK_CONV_FWD
                else:  # deconv == 1x1 conv + depth-to-space
                    gd, ind, ks, st, kind = (sd_, sh_, sw_), (sd_, sh_, sw_), 1, 1, L.PACK_DECONV_FWD
                wp = self._pack(op, kind, dtype, sd)
                if op.head:
                    out = L.hold_empty((n, op.cout) + ((gd[1], gd[2]) if dims == 2 else gd), dtype=torch.float32, device=dev)
                    ops.conv_fwd(srcs, wp, bias, op.cout, out, grid=(n,) + gd, in_dims=ind, ksize=ks, stride=st, dims=dims,
                                 combine=op.combine, out_planar=True, out_f32=True)
                    outs[op.out] = out
                    vdims[op.out] = gd
                    return
                if k == "deconv":
                    od = (2 * gd[0] if dims == 3 else gd[0], 2 * gd[1], 2 * gd[2])
                    out = L.hold_empty((n,) + od + (op.cout,), dtype=dtype, device=dev)
                    ops.conv_fwd(srcs, wp, bias, (2 ** dims) * op.cout, out, grid=(n,) + gd, in_dims=ind, ksize=1, stride=1, dims=dims,
                                 combine=op.combine, out_mode=1, out_cn=op.cout, stats=stats, stats_shift=sshift)
                    gd = od
                else:
                    out = L.hold_empty((n,) + gd + (op.cout,), dtype=dtype, device=dev)
                    ops.conv_fwd(srcs, wp, bias, op.cout, out, grid=(n,) + gd, in_dims=ind, ksize=ks, stride=st, dims=dims,
                                 combine=op.combine, stats=stats, stats_shift=sshift)
            vdims[op.out] = gd
            if op.inorm:            # InstanceNorm3d (affine=False) + ReLU: statistics of this single sample
                scale, shift = take(op.cout), take(op.cout)
                if op.cout not in self._ident:
                    self._ident[op.cout] = (torch.ones(op.cout, device=dev), torch.zeros(op.cout, device=dev))
                one, zero = self._ident[op.cout]
                ops.bn_finalize(stats, one, zero, None, None, None, gd[0] * gd[1] * gd[2], BN_EPS, 0.0, scale, shift)
                lz = Lazy(out, scale, shift, True, 0.0)
            elif op.bn:
                scale, shift = take(op.cout), take(op.cout)
                if train:
                    mean, invstd = take(op.cout), take(op.cout)
                    cnt = n * gd[0] * gd[1] * gd[2]
                    upd = update_stats
                    ops.bn_finalize(stats, sd[op.bn + ".weight"], sd[op.bn + ".bias"],
                                    sd[op.bn + ".running_mean"] if upd else None, sd[op.bn + ".running_var"] if upd else None,
                                    sd.get(op.bn + ".num_batches_tracked") if upd else None,
                                    cnt, BN_EPS, BN_MOMENTUM if upd else 0.0, scale, shift, mean, invstd,
                                    stats_shift=sshift, clog=clog(op))
                    S.bnstat[op.bn] = (mean, invstd, cnt)
                else:
                    ops.bn_eval_affine(sd[op.bn + ".weight"], sd[op.bn + ".bias"], sd[op.bn + ".running_mean"], sd[op.bn + ".running_var"],
                                       BN_EPS, scale, shift)
                lz = Lazy(out, scale, shift, True, op.slope)
            else:
                lz = Lazy(out)
            if op.drop and train:
                site, p, mode = op.drop
                if mode == "elem":
                    if drop_masks is not None:
                        keep = drop_masks.get(site)
                    elif id(op) in pre_keep:
                        keep = pre_keep[id(op)]
                        assert keep.shape == out.shape
                    else:
                        keep = L.hold_empty(out.shape, dtype=torch.uint8, device=dev)
                        ops.keep_mask(keep, drop_seed[id(op)], p, seed_dev=rng.seed_dev)
                    if keep is not None:
                        lz.keep, lz.keep_scale = keep, 1.0 / (1.0 - p)
                else:
                    if drop_masks is not None:
                        cm = drop_masks.get(site)
                    else:
                        cm = L.hold_empty(n, op.cout, dtype=torch.float32, device=dev)
                        ops.chan_mask(cm, drop_seed[id(op)], p, seed_dev=rng.seed_dev)
                    if cm is not None:
                        lz.chan_mul = cm
            V[op.out] = lz
        # ---- schedule: trunk, then the decoders side by side (second decoder on a forked stream)
        for op in prog.ops:
            if op.branch == 0:
                yield [lambda op=op: run_op(op)]
        if perturb is not None:
            # channel-level perturbation (FilterDropout.perform_dropout): every decoder gets its own version of the
            # trunk's values -- a larger batch with per-(sample, channel) multipliers -- and runs on that batch
            overlays, n_dec = perturb(vals, vdims)
            tables = {b: dict(vals, **overlays[b]) for b in branches if b != 0}
            if save:
                S.tables, S.n_dec = tables, n_dec
                S.fold = {b: {name: lz.chan_mul for name, lz in overlays[b].items()} for b in tables}
            run_dec = lambda op: run_op(op, tables[op.branch], n_dec)      # noqa: E731
        else:
            run_dec = run_op
        if zipped is not None:
            # the two decoders in lockstep: same-shaped layers (ConvBlock convs, their BatchNorm finalizes, the heads) are the two lanes
            # of ONE grouped launch (chap_hip.h, chap_group_*): half the launches of the decoder part, twice the tiles per launch,
            # and no second stream (which a captured pass on a forked stream could not have, see _side_stream)
            for pair in zipped:
                yield [lambda op=op: run_dec(op) for op in pair]
        elif side is not None:
            # two streams: the forked decoder's ops are issued first, then the other one's.  (Round 4 tried them alternately, op by op: neutral, 6.55 vs 6.5 ms;
            # the ISSUE order is the creation order of a captured graph's nodes, by which the ROCm 7.2 graph executor places the chains on its hardware
            # queues: DESIGN.md section 5, "Issue order" -- any change here has to be measured on the whole iteration.)
            side.wait_stream(cur_stream)
            sb = side_branch()                  # which decoder goes to the forked stream
            with torch.cuda.stream(side):
                for op in prog.ops:
                    if op.branch != 0 and (op.branch >= 2) == (sb == 2):
                        run_dec(op)
            for op in prog.ops:
                if op.branch != 0 and (op.branch >= 2) != (sb == 2):
                    run_dec(op)
            cur_stream.wait_stream(side)
        else:
            for op in prog.ops:
                if op.branch != 0:
                    yield [lambda op=op: run_dec(op)]
        logits = [outs[h] for h in prog.heads]
        extras = []
        for name in want:
            lz, (d_, h_, w_) = vals[name], vdims[name]
            o = L.hold_empty((N, lz.C) + ((h_, w_) if dims == 2 else (d_, h_, w_)), dtype=torch.float32, device=dev)
            ops.cl_to_planar(lz, o)
            extras.append(o)
        return logits, (S if save else None), extras

    # ---------------------------------------------------------------- backward
    def backward(self, S, dlogits, **kw):
        """dlogits: list (per head) of planar fp32 gradients or None. Accumulates parameter gradients
        into the module's flat grad views; returns dx (fp32, shape of x) or None."""
        return drive(self.backward_steps(S, dlogits, **kw), torch.cuda.current_stream().cuda_stream)

    def backward_steps(self, S, dlogits, *, dtype, need_wgrad, need_dx, grad_buffer=None):
        """The backward pass as a generator of steps (see forward_steps); returns dx."""
        prog, sd, dims = self.prog, self._sd(), self.prog.dims
        gr = None
        if need_wgrad:
            gr = self.m.grad_views_of(grad_buffer) if grad_buffer is not None else self.m._grad_views()
        dev = S.x.device
        N = S.x.shape[0]
        contrib = {}        # value name -> list of (tensor, coff, index of the op whose backward produced it)
        okey = {id(op): i for i, op in enumerate(prog.ops)}

        def incoming(name):
            """The gradient contributions of a value in PROGRAM order of their producers, whatever order the schedule (two
            streams, lockstep decoders) appended them in: the sum of up to three of them is not associative in floating point,
            and eager, captured and data-parallel runs must agree bit for bit."""
            c = contrib.get(name)
            return None if not c else [(t, o) for t, o, _ in sorted(c, key=lambda e: e[2])]

        pooled = {}         # value name -> (grad tensor, idx)
        head_g = dict(zip(prog.heads, dlogits))
        dx = None
        nsub = 2 ** dims
        # one arena for the BN-backward partial sums of every layer (per-block partial rows: nothing to zero)
        nsum = sum(ops.act_bwd_sums_size(op.cout) for op in prog.ops if op.bn)
        sums_arena = L.hold_empty(nsum, dtype=torch.float32, device=dev)
        spos = [0]

        def take_sums(c):
            n = ops.act_bwd_sums_size(c)
            t = sums_arena[spos[0]:spos[0] + n]
            spos[0] += n
            return t

        # the weight gradients of the two decoders are not on the path to the join with the trunk: deferred (defer_decoder_wgrad), they are issued on the
        # forked stream BEHIND both decoders' chains and run beside the trunk's backward, where that stream used to idle.  The closures stay in `deferred`
        # until this generator ends, i.e. until the streams have joined: they keep the gradient tensors alive (freed earlier, a tensor goes back to the
        # allocator of ITS stream and is handed out again while the forked stream still reads it)
        deferred = []
        defer_on = [False]

        def later(op, fn):
            if defer_on[0] and op.branch != 0:
                deferred.append(fn)
            else:
                fn()

        def scatter(op, srcs, dsrc):
            muls = S.fold.get(op.branch) if S.tables is not None else None
            if not muls:
                return self._scatter(contrib, op, srcs, dsrc, okey[id(op)])
            off = 0
            for name, s in zip(op.srcs, srcs):
                o = off if op.combine == 0 else 0
                if name in muls:        # a trunk value seen through cat((feat, mul * feat[B-U:])): fold the gradient back to B samples
                    contrib.setdefault(name, []).append((ops.fold_perturbed(dsrc, o, s.C, muls[name], N, S.n_dec - N), 0, okey[id(op)]))
                else:
                    contrib.setdefault(name, []).append((dsrc, o, okey[id(op)]))
                off += s.C

        def bwd_op(op):
            # a perturbed pass (FilterDropout): the decoders ran on their own value tables and batch of B + U samples
            V, n = (S.tables[op.branch], S.n_dec) if (S.tables is not None and op.branch != 0) else (S.vals, N)
            nonlocal dx
            k = op.kind
            if k == "pool":
                c = incoming(op.out)
                if c:
                    assert len(c) == 1 and c[0][1] == 0
                    pooled[op.srcs[0]] = (c[0][0], S.pool_idx[op.out])
                return
            if k == "up":
                c = incoming(op.out)
                if c:
                    assert len(c) == 1
                    src = V[op.srcs[0]]
                    d, h, w = S.dims[op.srcs[0]]
                    o = L.hold_empty(n, d, h, w, src.C, dtype=dtype, device=dev)
                    ops.upsample2x_bwd(c[0][0], c[0][1], src.C, o, dims=dims)
                    contrib.setdefault(op.srcs[0], []).append((o, 0, okey[id(op)]))
                return
            # ---- gradient w.r.t. the raw output of this conv
            v = None
            if op.head:
                dl = head_g.get(op.out)
                if dl is None:
                    return
                gd = S.dims[op.out]
                g16 = L.hold_empty((n,) + gd + (16,), dtype=dtype, device=dev)
                ops.planar_to_cl(dl, g16, cpad=16)
                g = Lazy(g16)
                kn_valid = op.cout
            else:
                c = incoming(op.out)
                pl = pooled.get(op.out)
                if not c and pl is None:
                    return
                v = V[op.out]
                gd = S.dims[op.out]
                kn_valid = 0
                plain = (v.scale is None and not v.act and v.keep is None and v.chan_mul is None)
                if plain and pl is None and len(c) == 1:
                    g = Lazy(c[0][0], C=v.C, coff=c[0][1])
                else:
                    gout = L.hold_empty((n,) + gd + (v.C,), dtype=dtype, device=dev)
                    kw = {}
                    if op.bn:
                        if S.train:
                            mean, invstd, cnt = S.bnstat[op.bn]
                            kw = dict(mean=mean, invstd=invstd, gamma=sd[op.bn + ".weight"], count=cnt, bn_mode=1)
                        else:
                            # eval-mode BN: fixed affine; BN parameter gradients from the same reduction
                            rm, rv = sd[op.bn + ".running_mean"], sd[op.bn + ".running_var"]
                            kw = dict(bn_mode=2)
                            if need_wgrad:
                                istd = L.hold(self.m._eval_invstd(op.bn, rv))      # a torch temporary: see _lib.hold
                                kw.update(mean=rm, invstd=istd, gamma=sd[op.bn + ".weight"])
                        if need_wgrad:
                            kw.update(dgamma=gr[op.bn + ".weight"], dbeta=gr[op.bn + ".bias"])
                    if op.bn:
                        kw["sums"] = take_sums(v.C)
                    ops.act_bwd(v, c or [], gout, g_pool=pl[0] if pl else None, pool_idx=pl[1] if pl else None, **kw)
                    g = Lazy(gout)
            # ---- this conv's own backward
            if k == "c1":
                D, H, W = S.dims[op.out]
                gt = g.raw if (g.coff == 0 and g.C == g.ld) else None
                assert gt is not None
                if need_wgrad:
                    if S.xpad is None and dtype == torch.bfloat16:      # the forward ran the direct first-layer kernel: pad the image now
                        S.xpad = L.hold_empty(n, D, H, W, 16, dtype=dtype, device=dev)
                        ops.planar_to_cl(S.x, S.xpad, cpad=16)
                    if S.xpad is not None:
                        taps = 3 ** dims
                        cin = S.x.shape[1]
                        ops.wgrad([Lazy(S.xpad)], g, gr[op.w], (1, taps, cin * taps), grid=(n, D, H, W), in_dims=(D, H, W),
                                  ksize=3, stride=1, dims=dims, db=gr[op.b] if op.b else None, kc_valid=cin)
                    else:
                        ops.conv_c1_bwd(gt, sd[op.w], S.x.view(n, D, H, W), dims=dims, dx=None,
                                        dw=gr[op.w], db=gr[op.b] if op.b else None)
                if need_dx:
                    dx = L.hold_empty_like(S.x)      # [n, in_chns, *spatial] fp32 == planar output
                    wp = self._pack(op, L.PACK_CONV_DGRAD, dtype, sd)
                    ops.conv_fwd([g], wp, None, S.x.shape[1], dx, grid=(n, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=dims,
                                 out_planar=True, out_f32=True)
                return
            srcs = [V[s] for s in op.srcs]
            sd_, sh_, sw_ = S.dims[op.srcs[0]]
            ctot = sum(s.C for s in srcs) if op.combine == 0 else srcs[0].C
            if k == "conv":
                taps = op.ksize ** dims
                if need_wgrad:
                    later(op, lambda: ops.wgrad(srcs, g, gr[op.w], (1, taps, ctot * taps), grid=(n, sd_, sh_, sw_), in_dims=(sd_, sh_, sw_),
                                                ksize=op.ksize, stride=1, dims=dims, combine=op.combine, db=gr[op.b] if op.b else None, kn_valid=kn_valid))
                wp = self._pack(op, L.PACK_CONV_DGRAD, dtype, sd)
                if (len(srcs) == 2 and op.combine == 0 and op.ksize == 3 and srcs[0].C == srcs[1].C and srcs[0].C % 16 == 0 and S.tables is None
                        and split_concat_gradient()):
                    # the input was torch.cat((skip, up), 1) (unet.py:98): its gradient as two DENSE tensors, one per source (chap_conv_params.out2) -- the
                    # BatchNorm backward of either source then reads whole 32-byte sectors instead of a 16-channel slice of a 32-channel row (round 3:
                    # FETCH_SIZE 1.25 x the algorithmic bytes of act_bwd at C = 16, 256 x 256)
                    d0 = L.hold_empty(n, sd_, sh_, sw_, srcs[0].C, dtype=dtype, device=dev)
                    d1 = L.hold_empty_like(d0)
                    ops.conv_fwd([g], wp, None, ctot, d0, grid=(n, sd_, sh_, sw_), in_dims=(sd_, sh_, sw_), ksize=op.ksize, stride=1, dims=dims, out2=d1)
                    for name, t in zip(op.srcs, (d0, d1)):
                        contrib.setdefault(name, []).append((t, 0, okey[id(op)]))
                else:
                    dsrc = L.hold_empty(n, sd_, sh_, sw_, ctot, dtype=dtype, device=dev)
                    ops.conv_fwd([g], wp, None, ctot, dsrc, grid=(n, sd_, sh_, sw_), in_dims=(sd_, sh_, sw_), ksize=op.ksize, stride=1, dims=dims)
                    scatter(op, srcs, dsrc)
            elif k == "down":
                gdd = S.dims[op.out]
                if need_wgrad:
                    later(op, lambda: ops.wgrad(srcs, g, gr[op.w], (1, nsub, ctot * nsub), grid=(n,) + gdd, in_dims=(sd_, sh_, sw_),
                                                ksize=2, stride=2, dims=dims, combine=op.combine, db=gr[op.b] if op.b else None))
                wp = self._pack(op, L.PACK_DOWN_DGRAD, dtype, sd)
                dsrc = L.hold_empty(n, sd_, sh_, sw_, ctot, dtype=dtype, device=dev)
                ops.conv_fwd([g], wp, None, nsub * ctot, dsrc, grid=(n,) + gdd, in_dims=gdd, ksize=1, stride=1, dims=dims,
                             out_mode=1, out_cn=ctot)
                scatter(op, srcs, dsrc)
            else:  # deconv: A = fine gradient (kc = co), B = coarse input (kn = ci)
                fine = S.dims[op.out]
                if need_wgrad:
                    assert len(srcs) == 1
                    def deconv_wgrad():
                        ops.wgrad([g], srcs[0], gr[op.w], (1, nsub, op.cout * nsub), grid=(n, sd_, sh_, sw_), in_dims=fine,
                                  ksize=2, stride=2, dims=dims)
                        if op.b:
                            ops.channel_sum(g, gr[op.b])
                    later(op, deconv_wgrad)
                wp = self._pack(op, L.PACK_DECONV_DGRAD, dtype, sd)
                dsrc = L.hold_empty(n, sd_, sh_, sw_, ctot, dtype=dtype, device=dev)
                ops.conv_fwd([g], wp, None, ctot, dsrc, grid=(n, sd_, sh_, sw_), in_dims=fine, ksize=2, stride=2, dims=dims)
                scatter(op, srcs, dsrc)
        # ---- schedule: the decoders' backward passes side by side, then the shared trunk
        cur_stream = torch.cuda.current_stream()
        rev = list(reversed(prog.ops))
        nbr = len({op.branch for op in prog.ops})
        side = self._side_stream(cur_stream) if nbr > 2 else None
        zipped = self._zipped() if (side is None and nbr == 3 and grouping_mode() != 0) else None
        if zipped is not None:
            for pair in reversed(zipped):
                yield [lambda op=op: bwd_op(op) for op in pair]
            for op in rev:
                if op.branch == 0:
                    yield [lambda op=op: bwd_op(op)]
        elif side is not None:
            side.wait_stream(cur_stream)
            sb = side_branch()
            defer_on[0] = need_wgrad and defer_decoder_wgrad()
            with torch.cuda.stream(side):
                for op in rev:
                    if op.branch != 0 and (op.branch >= 2) == (sb == 2):
                        bwd_op(op)
            for op in rev:
                if op.branch != 0 and (op.branch >= 2) != (sb == 2):
                    bwd_op(op)
            defer_on[0] = False
            if deferred:
                # join the CHAINS only; the deferred weight gradients follow on the forked stream (they read gradient tensors of both decoders) and are
                # joined at the end of the pass
                chain_done = torch.cuda.Event()
                chain_done.record(side)
                main_done = torch.cuda.Event()
                main_done.record(cur_stream)
                cur_stream.wait_event(chain_done)
                side.wait_event(main_done)
                with torch.cuda.stream(side):
                    for fn in deferred:
                        fn()
            else:
                cur_stream.wait_stream(side)
            for op in rev:
                if op.branch == 0:
                    bwd_op(op)
            if deferred:
                cur_stream.wait_stream(side)
        else:
            for op in rev:
                yield [lambda op=op: bwd_op(op)]
        return dx

    @staticmethod
    def _scatter(contrib, op, srcs, dsrc, key=0):
        if op.combine == 0:
            off = 0
            for name, s in zip(op.srcs, srcs):
                contrib.setdefault(name, []).append((dsrc, off, key))
                off += s.C
        else:
            for name in op.srcs:
                contrib.setdefault(name, []).append((dsrc, 0, key))


class Rng:
    """Seeds for the in-kernel counter RNG: a host counter (distinct stream per call site) plus an
    optional device word that the host bumps between replays of a captured graph."""

    def __init__(self, seed, device):
        self.base = int(seed) & 0xFFFFFFFF
        self.count = 0
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)

    def next_seed(self):
        self.count += 1
        return (self.base << 20) + self.count

    def reset_counter(self):
        self.count = 0
