"""Common machinery of the drop-in nn.Modules: flat parameter/gradient buffers, the executor,
and the single autograd node that wraps a whole hand-scheduled forward/backward."""
import contextlib

import torch
import torch.nn as nn

from .. import _lib as L
from ..engine import Executor, Rng


class _NetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward = Executor.forward, backward =
    Executor.backward.  Parameter gradients are accumulated directly into the module's flat
    gradient buffer (p.grad are views of it), so None is returned for them."""

    @staticmethod
    def forward(ctx, net, opts, x, *params):
        logits, S, extras = net._exec.forward(x, train=opts["train"], dtype=net.compute_dtype, save=opts["save"],
                                              update_stats=opts["update_stats"], drop_masks=opts["drop_masks"], rng=net._rng,
                                              want=opts.get("want", ()), perturb=opts.get("perturb"))
        ctx.net, ctx.S, ctx.nlogits = net, S, len(logits)
        ctx.grad_buffer = opts.get("grad_buffer")
        ctx.mark_non_differentiable(*extras)
        return tuple(logits) + tuple(extras)

    @staticmethod
    def backward(ctx, *dlogits):
        net = ctx.net
        if ctx.S is None:
            raise RuntimeError("chap_amd: backward through a forward pass that saved nothing")
        need_dx = ctx.needs_input_grad[2]
        need_w = any(ctx.needs_input_grad[3:])
        dl = [None if g is None else g.contiguous() for g in dlogits[:ctx.nlogits]]
        dx = net._exec.backward(ctx.S, dl, dtype=net.compute_dtype, need_wgrad=need_w, need_dx=need_dx, grad_buffer=ctx.grad_buffer)
        ctx.S = None
        return (None, None, dx) + (None,) * (len(ctx.needs_input_grad) - 3)


class ChapNet(nn.Module):
    """Base class: subclasses build the parameter containers (reference names) and a Program."""

    compute_dtype = torch.float32     # torch.bfloat16 = throughput mode (bench); fp32 = parity mode
    dims = 2
    in_chns = 1                       # input channels (1 on the CHAP hot path: ACDC slices, LA volumes); <= 16

    def _finish_init(self, program):
        self._exec = Executor(self, program)
        self._flat = None
        self._flat_grad = None
        self._rm_flat, self._rm_off = None, {}
        self._shift_buf, self._shift_hold = None, None
        self._manual_version = 0
        self._tensor_cache = None
        self._frozen = False
        self._rng = None
        self._eval_istd = {}

    # ------------------------------------------------------------------ options
    def set_compute_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        return self

    @contextlib.contextmanager
    def frozen(self):
        """Inside: forward passes do not require parameter gradients (VAT power iterations:
        only dL/dx is wanted, weight-gradient kernels are skipped)."""
        old, self._frozen = self._frozen, True
        try:
            yield self
        finally:
            self._frozen = old

    # ------------------------------------------------------------------ flat buffers
    def _param_list(self):
        return list(self.named_parameters())

    def _flat_ok(self):
        if self._flat is None:
            return False
        base = self._flat.data_ptr()
        for (name, p), off in zip(self._param_list(), self._offsets):
            if p.data_ptr() != base + 4 * off:
                return False
        if self._rm_flat is not None:
            rb = self._rm_flat.data_ptr()
            for n, o in self._rm_off.items():
                if self.get_buffer(n + ".running_mean").data_ptr() != rb + 4 * o:
                    return False
        return True

    def _ensure_flat(self):
        if self._flat_ok():
            return
        plist = self._param_list()
        dev = plist[0][1].device
        if dev.type != "cuda":
            raise L.ChapError("chap_amd networks run on an MI355X only (module is on %s); there is no CPU fallback" % dev)
        offs, n = [], 0
        for _, p in plist:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        grad = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for (_, p), o in zip(plist, offs):
                flat[o:o + p.numel()].copy_(p.detach().reshape(-1).float())
                p.data = flat[o:o + p.numel()].view(p.shape)
                p.grad = None
        self._flat, self._flat_grad, self._offsets = flat, grad, offs
        # BatchNorm running means in ONE flat buffer too (the modules' buffers become views): a forward pass snapshots
        # them with a single copy as the shift of its statistics' moments (engine.Executor.forward)
        rms = [(n, b) for n, b in self.named_buffers() if n.endswith(".running_mean")]
        self._rm_flat, self._rm_off = None, {}
        if rms:
            self._rm_flat = torch.empty(sum(b.numel() for _, b in rms), dtype=torch.float32, device=dev)
            o = 0
            with torch.no_grad():
                for n, b in rms:
                    v = self._rm_flat[o:o + b.numel()]
                    v.copy_(b.float())
                    self.get_submodule(n.rsplit(".", 1)[0])._buffers["running_mean"] = v
                    self._rm_off[n[:-len(".running_mean")]] = o
                    o += b.numel()
        self._tensor_cache = None
        self._manual_version += 1
        if self._rng is None or self._rng.seed_dev.device != dev:
            self._rng = Rng(torch.initial_seed(), dev)

    def _tensors(self):
        if self._tensor_cache is None:
            t = {n: p.data for n, p in self.named_parameters()}
            t.update({n: b for n, b in self.named_buffers()})
            self._tensor_cache = t
        return self._tensor_cache

    def _running_mean_flat(self):
        """(flat fp32 buffer of all BatchNorm running means, {bn prefix: offset}) or (None, None)."""
        self._ensure_flat()
        return (self._rm_flat, self._rm_off) if self._rm_flat is not None else (None, None)

    @contextlib.contextmanager
    def hold_stat_shift(self):
        """Inside: every forward pass takes the shift of its BatchNorm statistics (the c of sum(x - c), sum((x - c)^2)) from
        ONE snapshot of the running means made here, on the current stream.  An iteration that runs passes on several
        streams (ChapStep: pass B beside the VAT branch) needs this to be bitwise reproducible: a pass that snapshots the
        running means itself would see them before or after another stream's pass has updated them, depending on timing."""
        self._ensure_flat()
        if self._rm_flat is None or self._shift_hold is not None:
            yield self
            return
        if self._shift_buf is None or self._shift_buf.shape != self._rm_flat.shape or self._shift_buf.device != self._rm_flat.device:
            self._shift_buf = torch.empty_like(self._rm_flat)
        self._shift_buf.copy_(self._rm_flat)
        self._shift_hold = self._shift_buf
        try:
            yield self
        finally:
            self._shift_hold = None

    def _params_version(self):
        return self._manual_version + sum(p._version for p in self.parameters())

    def mark_params_dirty(self):
        self._manual_version += 1

    def _grad_views(self):
        """name -> fp32 view of the flat gradient buffer; (re)attaches p.grad, zeroing segments
        whose .grad had been set to None (optimizer.zero_grad(set_to_none=True))."""
        views = {}
        base = self._flat_grad.data_ptr()
        for (name, p), o in zip(self._param_list(), self._offsets):
            v = self._flat_grad[o:o + p.numel()].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                v.zero_()
                p.grad = v
            views[name] = v
        return views

    def grad_views_of(self, flat_grad):
        """name -> view of ANOTHER flat gradient buffer with this model's layout (second bucket: VAT branch /
        data-parallel overlap); p.grad is left alone."""
        assert flat_grad.shape == self._flat_grad.shape
        return {name: flat_grad[o:o + p.numel()].view(p.shape) for (name, p), o in zip(self._param_list(), self._offsets)}

    def flat_buffers(self):
        """(params, grads) flat fp32 tensors (for the fused optimizer and the DP all-reduce)."""
        self._ensure_flat()
        self._grad_views()
        return self._flat, self._flat_grad

    def swap_grad_buffer(self, new_flat_grad):
        """Point p.grad at another flat buffer of the same layout (DP bucket schedule); contents are
        left untouched."""
        assert new_flat_grad.shape == self._flat_grad.shape
        self._flat_grad = new_flat_grad
        for (name, p), o in zip(self._param_list(), self._offsets):
            p.grad = new_flat_grad[o:o + p.numel()].view(p.shape)

    def prepare_weights(self):
        """Re-pack the weights now, on the current stream, if an optimizer step has changed them.  A caller that is about to
        run forward passes of this model on SEVERAL streams at once must do this first: otherwise the first pass re-packs on
        its stream while the others already read the packed copies."""
        self._ensure_flat()
        self._exec._ensure_packed(self.compute_dtype, self._exec._sd())

    def backward_saved(self, out, dlogits, grad_buffer=None, need_wgrad=True, need_dx=False):
        """Run the backward pass of the forward that produced `out` (a logits tensor of this network) directly -- not through
        torch.autograd, on the CURRENT stream (autograd would run it on the forward's stream) -- accumulating the parameter
        gradients of sum_h <logits_h, dlogits_h> into `grad_buffer` (a flat buffer with this model's layout; None = the
        model's own) and / or returning the input gradient.  May be called several times on one forward (the saved state is
        kept until release_saved): the backward pass is linear in dlogits, so the gradients of the parts of a loss can be
        taken apart (GradSim: labeled vs unlabeled part of the BCP loss)."""
        ctx = out.grad_fn
        if ctx is None or getattr(ctx, "S", None) is None:
            raise RuntimeError("chap_amd: backward_saved needs the output of a forward pass that saved its state")
        dl = [None if g is None else g.contiguous() for g in dlogits]
        return self._exec.backward(ctx.S, dl, dtype=self.compute_dtype, need_wgrad=need_wgrad, need_dx=need_dx, grad_buffer=grad_buffer)

    def release_saved(self, out):
        ctx = out.grad_fn
        if ctx is not None:
            ctx.S = None

    def _eval_invstd(self, bn, rv):
        return (rv + 1e-5).rsqrt()

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._flat = None
        self._tensor_cache = None
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._manual_version = getattr(self, "_manual_version", 0) + 1
        return r

    # ------------------------------------------------------------------ running the program
    def _run(self, x, *, drop_masks=None, update_stats=True, want=(), grad_buffer=None, perturb=None):
        if x.dim() != self.dims + 2 or x.shape[1] != self.in_chns:
            raise ValueError("chap_amd: expected input [N, %d, %s], got %s" % (self.in_chns, ", ".join("*" * self.dims), tuple(x.shape)))
        self._ensure_flat()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        params = [p for _, p in self._param_list()]
        grad_on = torch.is_grad_enabled()
        if self._frozen or not grad_on:
            params = [p.detach() for p in params]
        save = grad_on and (x.requires_grad or any(p.requires_grad for p in params))
        opts = dict(train=self.training, save=save, update_stats=update_stats, drop_masks=drop_masks, want=tuple(want), grad_buffer=grad_buffer,
                    perturb=perturb)
        return _NetFn.apply(self, opts, x, *params)


def holder(**mods):
    """A parameter container whose children carry the reference's numeric child names
    (e.g. conv_conv.0 / .1 / .4 / .5): only parameterised layers are instantiated."""
    m = nn.Module()
    for k, v in mods.items():
        m.add_module(k.lstrip("_"), v)
    return m
