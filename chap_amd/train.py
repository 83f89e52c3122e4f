"""Host side of one CHAP training iteration, mirroring code/train_ours_2D.py:301-389 step by step
(same names where the reference has them).  Every tensor op is a libchap_hip.so kernel; torch
provides memory, streams, the autograd trampoline for the two network nodes and (optionally) HIP
graph capture of the whole iteration.

Pieces that are ABSENT from the reference (losses.VAT2d, patch.create_maskV1, losses.DiceLoss_bcp,
ramps.sigmoid_rampup) follow the definitions in DESIGN.md (P1-P4); the CPU oracle restates the same
definitions (oracle/train_step.py).
"""
import contextlib
import math
import os

import numpy as np
import torch

from . import ops


def sigmoid_rampup(current, rampup_length):
    if rampup_length == 0:
        return 1.0
    t = float(np.clip(current, 0.0, rampup_length)) / rampup_length
    return float(math.exp(-5.0 * (1.0 - t) ** 2))


def get_current_consistency_weight(epoch, args):            # train_ours_2D.py:34-36
    return args["consistency"] * sigmoid_rampup(epoch, args["consistency_rampup"])


DEFAULT_ARGS = dict(base_lr=0.01, batch_size=24, labeled_bs=12, max_iterations=30000, num_classes=4,
                    consistency=1.0, consistency_rampup=50.0, noise_mag=10.0, epi=6.0, topk1=0.1,
                    adv_noise=True, adv_losstype="kl", vat_iters=1, vat_sign=False, nms=1,
                    momentum=0.9, weight_decay=1e-4, dropout=False, comp_drop=False)


class FusedSGD:
    """optim.SGD(lr, momentum=0.9, weight_decay=1e-4) (train_ours_2D.py:278) as ONE kernel over the
    model's flat parameter buffer; lr lives in device memory (graph replays pick up new values)."""

    def __init__(self, model, lr, momentum=0.9, weight_decay=1e-4):
        self.model, self.momentum, self.weight_decay = model, momentum, weight_decay
        flat, grad = model.flat_buffers()
        self.mom = torch.zeros_like(flat)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=flat.device)
        self.param_groups = [{"lr": float(lr)}]

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = float(lr)
        self.lr_dev.fill_(float(lr))

    def zero_grad(self, set_to_none=False):
        pass                                   # the step kernel zeroes the gradient buffer

    def step(self, grad_scale=1.0, grad2=None):
        flat, grad = self.model.flat_buffers()
        ops.sgd_step(flat, grad, self.mom, self.lr_dev, self.momentum, self.weight_decay, grad_scale, True, grad2)
        self.model.mark_params_dirty()


def _distinct_streams(dev, n):
    """n streams with pairwise different handles, none of them the current or the default stream."""
    seen = {0, torch.cuda.current_stream(dev).cuda_stream}
    out = []
    for _ in range(256):
        s = torch.cuda.Stream(device=dev)
        if s.cuda_stream not in seen:
            seen.add(s.cuda_stream)
            out.append(s)
            if len(out) == n:
                return out
    raise RuntimeError("chap_amd: could not get %d distinct streams from the pool" % n)


def check_graph_environment(concurrent=True, environ=None):
    """Refuse a graph capture the runtime is known to die in.  The captured iteration forks four streams (capture origin, pass B, second decoder,
    early VAT pass); with GPU_MAX_HW_QUEUES below 3 the FIRST REPLAY of any such graph aborts inside the ROCm 7.2 runtime with no HIP error
    (profiles/r03_runtime_aborts.log: 2D and 3D, default schedule) -- a user-reachable environment, so capture() raises here instead.  A
    single-stream iteration (ChapStep(args={'concurrent': False})) captures as one chain and is not affected."""
    from ._lib import ChapError
    env = os.environ if environ is None else environ
    v = env.get("GPU_MAX_HW_QUEUES")
    if v is None or not concurrent:
        return
    try:
        n = int(v)
    except ValueError:
        return
    if n < 3:
        raise ChapError("chap_amd: GPU_MAX_HW_QUEUES=%d: the multi-stream HIP graph of the iteration needs at least 3 hardware queues (its first replay "
                        "aborts in the runtime below that).  Unset GPU_MAX_HW_QUEUES (default 4) or build the step with args={'concurrent': False}." % n)


class VAT2d:
    """adv_loss = VAT2d(xi, epi, num_classes); adv_loss(model, x, soft1, soft2, mask, losstype)
    (call sites train_ours_2D.py:290,372); losstype 'kl' or 'dice' (--adv_losstype, :515), `sign=True` = the FGSM-style
    sign step  r = epi / sqrt(P) * sign(d) * mask.  Returns a 1-element device tensor; the gradient of
    `weight * loss` w.r.t. the parameters is accumulated into the model's gradient buffer here
    (the network node is driven directly with d(loss)/d(logits) from the fused KL kernel)."""

    def __init__(self, xi=10.0, epi=6.0, num_classes=4, ip=1, sign=False):
        self.xi, self.epi, self.num_classes, self.ip, self.sign = xi, epi, num_classes, ip, sign

    def begin(self, model, x, U, inject=None):
        """The part of the VAT computation that does NOT depend on pass A: the initial direction d (U(-.5,.5), per-sample L2
        normalised) and the forward pass of the first power iteration on x + xi * d.  ChapStep runs it on a stream of its own
        BESIDE pass A (the two forwards only meet in the distance kernel), which takes one forward pass off the critical chain
        of the iteration.  Returns the state for finish()."""
        inject = inject or {}
        x = x[-U:].contiguous()                 # "perturb the last U samples" (SURVEY.md section 3.1 note)
        d = torch.empty_like(x)
        if inject.get("d0") is not None:
            ops.l2_normalize(inject["d0"], d)
        else:
            ops.rand_uniform(d, model._rng.next_seed(), -0.5, 0.5, seed_dev=model._rng.seed_dev)
            ops.l2_normalize(d, d)
        xh = torch.empty_like(x).requires_grad_(True)
        first = None
        if self.ip > 0:
            ops.perturb(x, d, xh, self.xi)
            with model.frozen():
                first = model(xh, update_stats=False, drop_masks=inject.get("drop_V0"))
        return dict(x=x, d=d, xh=xh, first=first, inject=inject)

    def finish(self, model, st, soft1, soft2, mask, losstype="kl", weight_dev=None, accumulate_grad=True, grad_buffer=None, weight=1.0, fork_ctl=None):
        gen = self.finish_steps(model, st, soft1, soft2, mask, losstype, weight_dev, accumulate_grad, grad_buffer, weight, fork_ctl)
        while True:
            try:
                next(gen)
            except StopIteration as e:
                return e.value

    def finish_steps(self, model, st, soft1, soft2, mask, losstype="kl", weight_dev=None, accumulate_grad=True, grad_buffer=None, weight=1.0, fork_ctl=None):
        """finish() as a generator that yields between its passes -- [power iteration k: distance gradient, backward to the input, normalise] ...,
        [final forward + distance], [final backward] -- so that the caller can ISSUE another chain's passes in between (ChapStep._iteration: the order
        in which the nodes of a captured graph were created is the order in which a replay feeds them to the GPU, see there).  Returns the loss."""
        if losstype not in ops.DIST_MODES:
            raise ValueError("chap_amd VAT2d: adv_losstype=%r (--adv_losstype {kl,dice}, train_ours_2D.py:515)" % (losstype,))
        x, d, xh, inject = st["x"], st["d"], st["xh"], st["inject"]
        fork_ctl = fork_ctl or (lambda bit: None)      # (ChapStep: which passes of the chain may fork their second decoder under capture, CHAP_FORK_MASK)
        for it in range(self.ip):
            if it == 0:
                l1, l2 = st["first"]
            else:
                ops.perturb(x, d, xh, self.xi)
                fork_ctl(4)
                with model.frozen():
                    l1, l2 = model(xh, update_stats=False, drop_masks=inject.get("drop_V%d" % it))
            g1, g2 = torch.empty_like(l1), torch.empty_like(l2)
            ops.kl_fwd_bwd((l1, l2), (soft1, soft2), None, (g1, g2), mode=losstype)
            fork_ctl(2)
            # d(distance)/d(x) only (the weights are frozen: no weight-gradient kernels), on the CURRENT stream whatever stream the
            # forward ran on (torch.autograd would run the node on the forward's stream)
            dx = model.backward_saved(l1, [g1, g2], need_wgrad=False, need_dx=True)
            model.release_saved(l1)
            ops.l2_normalize(dx, d)
            yield
        xa = torch.empty_like(x)
        m = None if mask is None else mask.reshape(x.shape)
        alpha = self.epi / math.sqrt(x[0].numel()) if self.sign else self.epi
        ops.perturb(x, d, xa, alpha, mask=m, sign=self.sign)
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
        if accumulate_grad:
            fork_ctl(4)
            l1, l2 = model(xa, update_stats=False, drop_masks=inject.get("drop_VF"), grad_buffer=grad_buffer)
            g1, g2 = torch.empty_like(l1), torch.empty_like(l2)
            ops.kl_fwd_bwd((l1, l2), (soft1, soft2), loss, (g1, g2), gscale=weight, gscale_dev=weight_dev, mode=losstype)
            yield
            fork_ctl(8)
            torch.autograd.backward([l1, l2], [g1, g2])
        else:
            with torch.no_grad():
                l1, l2 = model(xa, update_stats=False, drop_masks=inject.get("drop_VF"))
            ops.kl_fwd_bwd((l1, l2), (soft1, soft2), loss, mode=losstype)
        return loss

    def __call__(self, model, x, soft1, soft2, mask, losstype="kl", weight_dev=None, inject=None, accumulate_grad=True, grad_buffer=None,
                 weight=1.0):
        if losstype not in ops.DIST_MODES:
            raise ValueError("chap_amd VAT2d: adv_losstype=%r (--adv_losstype {kl,dice}, train_ours_2D.py:515)" % (losstype,))
        st = self.begin(model, x, soft1.shape[0], inject)
        return self.finish(model, st, soft1, soft2, mask, losstype, weight_dev, accumulate_grad, grad_buffer, weight)


class GradSim:
    """gradsim = grad.GradSim(device, num_classes, dir) (ABSENT upstream; call sites train_ours_2D.py:288,297,360,365): the
    channel scores of the channel-level perturbation.  Definition of this build (parity unpinned): for each of
    the five encoder levels, the conv kernel that produces the level's feature (the second conv of its ConvBlock) and, per
    output channel, the cosine similarity of the gradients of the LABELED part and of the UNLABELED part of the BCP loss
    (loss_l, loss_u at train_ours_2D.py:352-353) with respect to that kernel -- channels on which the two supervisions agree
    score high and are kept more often (scores_dropoutV2, FilterDropout.py:116-138).

        init_simsocre() -> [zeros(C_l)]          (:297; all-zero scores = the Dropout2d pair, FilterDropout.py:71-73)
        get_sim()       -> the current scores     (:360)
        get_grad_convkernel(grad_l, grad_u, model, optimizer=None, iter_num=0) -> the new scores   (:365)

    Difference to the call site: upstream hands over the two LOSSES (autograd); here the caller hands over the two flat
    gradient buffers (model layout) that its two backward passes filled."""

    KEYS_2D = ["encoder.in_conv.conv_conv.4.weight"] + ["encoder.down%d.maxpool_conv.1.conv_conv.4.weight" % i for i in range(1, 5)]

    def __init__(self, device, num_classes=4, dir=None, ema=0.0):
        self.device, self.num_classes, self.dir, self.ema = device, num_classes, dir, ema
        self.scores = None

    def init_simsocre(self, model=None):
        chans = (16, 32, 64, 128, 256) if model is None else [dict(model.named_parameters())[k].shape[0] for k in self.KEYS_2D]
        self.scores = [torch.zeros(c, dtype=torch.float32, device=self.device) for c in chans]
        return self.scores

    def get_sim(self):
        if self.scores is None:
            self.init_simsocre()
        return self.scores

    def get_grad_convkernel(self, grad_l, grad_u, model, optimizer=None, iter_num=0):
        vl, vu = model.grad_views_of(grad_l), model.grad_views_of(grad_u)
        for sc, k in zip(self.get_sim(), self.KEYS_2D):
            ops.grad_sim(vl[k], vu[k], sc, self.ema)
        return self.scores


class ChapStep:
    """One iteration of train() (train_ours_2D.py:301-389) for a DualDecoder on 2D slices, or -- the
    same loop restated on 5-D tensors, the reference has no 3D training script -- a DualDecoder3d on
    volumes (cuboid BCP box, 26-connected largest component, in-plane 4x4 patches for the VAT mask).

    step(volume_batch [B,1,*sp] fp32, label_batch [B,*sp] int64) -> dict of device scalars.
    The BCP box offsets, LR and consistency weight live in device memory so that the whole iteration
    can be captured once as a HIP graph (`capture()`) and replayed."""

    def __init__(self, model, args=None, optimizer=None, world_size=1):
        a = dict(DEFAULT_ARGS)
        a.update(args or {})
        self.args, self.model = a, model
        self.opt = optimizer or FusedSGD(model, a["base_lr"], a["momentum"], a["weight_decay"])
        self.adv_loss = VAT2d(xi=a["noise_mag"], epi=a["epi"], num_classes=a["num_classes"], ip=a["vat_iters"], sign=a["vat_sign"])
        dev = self.opt.lr_dev.device
        self.dims = getattr(model, "dims", 2)
        # host-scheduled values of an iteration in ONE device word block -> one host-to-device copy per step:
        # [BCP box: 4 / 6 int32 | consistency weight f32 | learning rate f32]
        nbox = 4 if self.dims == 2 else 6
        self._sched = torch.zeros(8, dtype=torch.int32, device=dev)
        self._sched_pin = [torch.empty(8, dtype=torch.int32).pin_memory() for _ in range(4)]      # host side of the schedule block (see _upload_sched)
        self._sched_ev, self._sched_k = [None] * 4, 0
        self.box = self._sched[:nbox]
        self.cw_dev = self._sched[6:7].view(torch.float32)
        if isinstance(self.opt, FusedSGD):                       # the optimizer reads its lr from the same block
            self._sched[7:8].view(torch.float32).copy_(self.opt.lr_dev)
            self.opt.lr_dev = self._sched[7:8].view(torch.float32)
        self.iter_num = 0
        self.world_size = world_size
        self.grad_sync = None                   # parallel.DataParallelSync (world_size > 1)
        self._graph, self._graph_opt, self._graphs_dp = None, None, None
        # --dropout: per-level channel scores [C_l] = gradsim.get_sim() (train_ours_2D.py:360), refreshed every iteration by
        # gradsim.get_grad_convkernel (:365); all-zero (the initial value) = the Dropout2d pair (FilterDropout.py:71-73).
        # `sim_score` (or inject['sim_score']) overrides them.
        self.sim_score = None
        self.gradsim = None
        if a["dropout"] and self.dims == 2:
            self.gradsim = GradSim(dev, a["num_classes"])
            self.gradsim.init_simsocre(model)
            n_ = model.flat_buffers()[1].numel()
            self._grad_lu = torch.zeros(2 * n_, dtype=torch.float32, device=dev)      # gradients of loss_l | loss_u (pass B)
        # second gradient bucket: the VAT branch accumulates here, so it can run on its own stream beside the
        # BCP branch (and, data-parallel, its all-reduce overlaps); the fused SGD sums both buckets
        n = model.flat_buffers()[1].numel()
        self.grad_both = torch.zeros(2 * n, dtype=torch.float32, device=dev)     # [bucket 0 | bucket 1], one all-reduce
        model.swap_grad_buffer(self.grad_both[:n])
        self.grad2 = self.grad_both[n:]
        self.concurrent = bool(a.get("concurrent", True))
        # Streams come from PyTorch's per-device pool (32 of them, handed out round-robin): two "new" streams of a long-lived
        # process can be the SAME stream.  The ones of an iteration must differ from each other and from the stream the graph is
        # captured on (pass B on the capture's origin stream would run behind the VAT chain, not beside it), so they are drawn
        # until they do and the capture gets a stream of its own.
        st = _distinct_streams(dev, 4)
        self._cap = st[3]                                          # origin stream of capture()
        self._side = st[0] if self.concurrent else None
        # Decoder-level concurrency inside a captured graph.  On ROCm 7.2 a captured stream may fork/join with the
        # capture's ORIGIN stream any number of times, but an event dependency between two forked streams crashes
        # hipStreamEndCapture.  So the long chain of the iteration (the VAT branch: K+1 forward/backward pairs)
        # stays on the origin stream, where the executor may fork its second decoder, and the short one (pass B)
        # goes to the side stream with its decoders back to back.
        self._d2 = st[1] if self.concurrent else None
        self._pre = st[2] if self.concurrent else None              # the VAT pre-pass beside pass A

    # ------------------------------------------------------------------ host-side schedule values
    def prepare(self, box_yx=None):
        """Host work of an iteration that must happen BEFORE the device work (and outside a captured
        graph): BCP box offsets (np.random.randint, train_ours_2D.py:97-98), consistency weight."""
        a = self.args
        sizes = [int(s * 2 / 3) for s in self._hw]                 # patch = 2/3 of every side (:96)
        if box_yx is None:
            box_yx = tuple(np.random.randint(0, s - p) for s, p in zip(self._hw, sizes))
        self._upload_sched(list(box_yx) + sizes, get_current_consistency_weight(self.iter_num // 150, a))

    def _upload_sched(self, box_vals, cw):
        import struct
        f2i = lambda v: struct.unpack("<i", struct.pack("<f", float(v)))[0]
        vals = list(box_vals) + [0] * (6 - len(box_vals)) + [f2i(cw), f2i(self.opt.param_groups[0]["lr"])]
        # Through a small ring of PINNED host blocks, asynchronously: a plain `copy_` from pageable memory synchronises the stream,
        # i.e. the host would wait for the previous iteration to drain before it even starts launching the next one (the GPU then
        # idles while the graph's first nodes are being submitted: ~0.5 ms of gaps at the head of every replay in the kernel trace).
        # A slot is reused only after the copy that last read it has run (event), which also bounds how far the host runs ahead.
        k = self._sched_k
        self._sched_k = (k + 1) % len(self._sched_pin)
        if self._sched_ev[k] is not None:
            self._sched_ev[k].synchronize()
        self._sched_pin[k].copy_(torch.tensor(vals, dtype=torch.int32))
        self._sched.copy_(self._sched_pin[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._sched_ev[k] = ev

    def finish(self):
        """poly LR applied AFTER the step (train_ours_2D.py:385-389)."""
        a = self.args
        self.iter_num += 1
        lr_ = a["base_lr"] * (1.0 - self.iter_num / a["max_iterations"]) ** 0.9
        if isinstance(self.opt, FusedSGD):
            self.opt.param_groups[0]["lr"] = float(lr_)          # reaches the device with the next step's schedule block
        else:
            self.opt.set_lr(lr_)
        return lr_

    # ------------------------------------------------------------------ checkpoint / resume (build extension)
    def state_dict(self):
        """Everything a long run needs to continue exactly where it stopped.  'model' is the reference's checkpoint
        (what train_ours_2D.py:428-435 saves with torch.save(model.state_dict())); the rest is absent upstream:
        momentum buffer, iteration counter (drives poly LR and the consistency ramp), numpy RNG state (BCP box
        offsets) and the dropout / VAT noise RNG epoch."""
        rng = self.model._rng
        return {"model": {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                "momentum": self.opt.mom.detach().clone(), "iter_num": int(self.iter_num),
                "lr": float(self.opt.param_groups[0]["lr"]), "numpy_rng": np.random.get_state(),
                "rng": {"base": rng.base, "count": rng.count, "seed_dev": int(rng.seed_dev.item())},
                "gradsim": None if self.gradsim is None else [sc.detach().clone() for sc in self.gradsim.get_sim()]}

    def load_state_dict(self, st):
        self.model.load_state_dict(st["model"], strict=True)
        self.opt.mom.copy_(st["momentum"])
        self.iter_num = int(st["iter_num"])
        self.opt.set_lr(st["lr"])
        np.random.set_state(st["numpy_rng"])
        rng = self.model._rng
        rng.base, rng.count = int(st["rng"]["base"]), int(st["rng"]["count"])
        rng.seed_dev.fill_(int(st["rng"]["seed_dev"]))
        if self.gradsim is not None and st.get("gradsim") is not None:
            for dst, src in zip(self.gradsim.get_sim(), st["gradsim"]):
                dst.copy_(src)
        self.grad_both.zero_()
        return self

    # ------------------------------------------------------------------ the device work
    def exchange_and_update(self):
        """(data-parallel) all-reduce of both gradient buckets, then optimizer.step() (:381-383): the fused SGD
        sums the buckets, scales by 1/world and zeroes them."""
        if self.grad_sync is not None:
            self.grad_sync.start()
            self.grad_sync.wait()
        self.opt.step(grad_scale=1.0 / self.world_size, grad2=self.grad2)

    def device_step(self, volume_batch, label_batch, inject=None, update=True):
        """The device work of one iteration on the current stream (+ the side streams): phase A (pass A, pseudo labels,
        perturbation mask), then phase B (largest-CC filter, BCP mixing, pass B forward / backward -> gradient bucket 0) on a
        side stream BESIDE phase V (the VAT chain -> bucket 1) on this one, then the gradient exchange and the optimizer."""
        with self.model.hold_stat_shift():          # one snapshot of the running means for all passes of the iteration (determinism)
            return self._iteration(volume_batch, label_batch, inject, update)

    def _iteration(self, volume_batch, label_batch, inject=None, update=True):
        main = torch.cuda.current_stream()
        with self._decoder_fork(main):
            ctx = self._phase_a(volume_batch, label_batch, inject)
            if self.concurrent and self.args["adv_noise"]:
                # Phase B (side stream) and phase V (this stream, the iteration's critical chain): all of phase B is issued first, then phase V.  Round 4 tried
                # issuing them pass by pass in alternation (same launches, same streams, bit-identical results): 8.05 ms per 2D step instead of 6.5 -- the
                # ROCm 7.2 graph executor places the chains of a captured graph on its hardware queues by the creation order of the nodes, and in that
                # order pass B lands on the VAT chain's queue and runs entirely after it ("Issue order").
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    losses = self._phase_b(ctx)
                    if self.grad_sync is not None and update and not torch.cuda.is_current_stream_capturing():
                        self.grad_sync.start_first()        # bucket 0 is final: its all-reduce runs beside the VAT chain
                vat_loss = self._phase_v(ctx)
                main.wait_stream(self._side)
            else:
                losses = self._phase_b(ctx)
                vat_loss = self._phase_v(ctx)
        out = {"mix_losses": losses, "vat_loss": vat_loss}
        if self.args["dropout"]:
            out["fp_losses"] = self._fp_branch(ctx["uimg_ab"], ctx["pseudo_outputs1"], ctx["pseudo_outputs2"], ctx["inject"], None)
        if update:
            self.exchange_and_update()
        return out

    def _fork_ctl(self, bit):
        """CHAP_FORK_MASK (lab / A-B switch, default 14): which passes of the capture's origin stream fork their second decoder onto a stream of its own
        (1 pass A, 2 the power iteration's backward, 4 the VAT forward passes after the first, 8 the final backward); the others run their decoders with
        grouped launches.  Every fork is one more chain for the graph executor to place on its few hardware queues ("Issue order").
        Round 4, 12 masks on the whole iteration (profiles/r04_issue_order_ab.log): 14 -- pass A, which already shares the GPU with the early VAT pass,
        keeps its decoders on one stream -- 6.469 ms against 6.513 for 15 (three pairs; 3D 14.79 vs 14.81), every other mask slower (0: 6.83 / 15.8)."""
        cs = getattr(self, "_cs_active", None)
        if cs is None:
            return
        mask = int(os.environ.get("CHAP_FORK_MASK", "14"))
        self.model._exec._capture_sides = cs if (mask & bit) else {}

    @contextlib.contextmanager
    def _decoder_fork(self, origin):
        """Under capture: let the executor run the second decoder of the passes on `origin` on a stream forked from it (a
        captured stream may fork / join with the capture's ORIGIN stream only, see __init__)."""
        capturing = self.concurrent and torch.cuda.is_current_stream_capturing()
        if capturing:
            self._d2.wait_stream(origin)
            self._cs_active = {origin.cuda_stream: self._d2}
            self.model._exec._capture_sides = self._cs_active
        try:
            yield
        finally:
            if capturing:
                origin.wait_stream(self._d2)
                self.model._exec._capture_sides = {}
                self._cs_active = None

    def _phase_a(self, volume_batch, label_batch, inject):
        a, model = self.args, self.model
        inject = inject or {}
        lbs = a["labeled_bs"]
        B = volume_batch.shape[0]
        lsub, usub = lbs // 2, (B - lbs) // 2
        ctx = dict(inject=inject, volume_batch=volume_batch, lsub=lsub, usub=usub,
                   img_a=volume_batch[:lsub], img_b=volume_batch[lsub:lbs], uimg_a=volume_batch[lbs:lbs + usub], uimg_b=volume_batch[lbs + usub:],
                   lab_a=label_batch[:lsub], lab_b=label_batch[lsub:lbs], uimg_ab=volume_batch[lbs:])
        # ---- the first VAT power-iteration forward (x + xi * d) needs nothing from pass A: it runs beside it on its own stream
        main = torch.cuda.current_stream()
        # (`vat_early`, default on since the end of round 2: 2D 7.60 -> 7.41 ms, 3D 18.29 -> 17.75 ms per step.  It had been off because
        #  two identical runs were not always bitwise equal with it; the cause was not this overlap but dropout seeds drawn in ISSUE
        #  order -- see Executor.forward -- together with stream handles repeating in PyTorch's pool, see ChapStep.__init__.)
        pre = self._pre if (self.concurrent and a["adv_noise"] and a.get("vat_early", True)) else None
        model.prepare_weights()                  # before the streams fork: every pass of the iteration reads the same packed copies
        if a["adv_noise"]:
            if pre is not None:
                # The early pass runs on its own stream beside pass A and is issued in FRONT of it (round 4 tried issuing it between pass A's encoder and
                # its decoders: 6.95 vs 6.5 ms -- it then shares a queue with pass A)
                pre.wait_stream(main)
                with torch.enable_grad(), torch.cuda.stream(pre):
                    ctx["vat_state"] = self.adv_loss.begin(model, volume_batch, B - lbs, inject)
            else:
                ctx["vat_state"] = self.adv_loss.begin(model, volume_batch, B - lbs, inject)
        # ---- pass A: pseudo labels from both decoders (no grad), train_ours_2D.py:314-330
        self._fork_ctl(1)
        with torch.no_grad():
            pre_ab1, pre_ab2 = model(ctx["uimg_ab"], drop_masks=inject.get("drop_A"))
            soft1, soft2, pseudo1, pseudo2, knowledge = ops.pseudo_block(pre_ab1, pre_ab2)
        if pre is not None:
            main.wait_stream(pre)
        ctx.update(outputs_soft1=soft1, outputs_soft2=soft2, pseudo_outputs1=pseudo1, pseudo_outputs2=pseudo2, knowledge=knowledge)
        # ---- the VAT branch (:368-375) depends only on pass A: it and pass B run side by side on two streams and
        #      accumulate their parameter gradients into separate buckets (VAT: the second one)
        if a["adv_noise"]:
            ctx["diff_mask"] = ops.diff_mask(pseudo1, pseudo2, knowledge, 4, a["topk1"])
        return ctx

    @staticmethod
    def _drain(gen):
        while True:
            try:
                next(gen)
            except StopIteration as e:
                return e.value

    def _phase_b(self, ctx):
        return self._drain(self._phase_b_steps(ctx))

    def _phase_v(self, ctx):
        return self._drain(self._phase_v_steps(ctx))

    def _phase_b_steps(self, ctx):
        """Largest-CC filter, loss mask and BCP mixing (:326-338) feed pass B only: they run at the head of this branch, off
        the critical path (the VAT branch needs the soft / arg-max outputs, not these); then pass B and the four mix_loss
        terms (:339-351), backward into gradient bucket 0.  A generator: yields between its forward and its backward part (see _iteration)."""
        a, model, inject = self.args, self.model, ctx["inject"]
        nc = a["num_classes"]
        lsub, usub, volume_batch = ctx["lsub"], ctx["usub"], ctx["volume_batch"]
        with torch.no_grad():
            if a["nms"]:
                plab1 = ops.largest_cc(ctx["pseudo_outputs1"], nc)
                plab2 = ops.largest_cc(ctx["pseudo_outputs2"], nc)
            else:
                plab1, plab2 = ctx["pseudo_outputs1"], ctx["pseudo_outputs2"]
            loss_mask = torch.empty(lsub, *volume_batch.shape[2:], dtype=torch.int64, device=volume_batch.device)
            ops.box_mask(loss_mask, self.box)
            # BCP mixing (:335-338): net_input_mix = cat(net_input_l, net_input_unl)
            net_input_mix = torch.empty((lsub + usub,) + tuple(volume_batch.shape[1:]), dtype=torch.float32, device=volume_batch.device)
            ops.box_mix(ctx["img_b"], ctx["uimg_b"], net_input_mix[:lsub], self.box)       # img_b*mask + uimg_b*(1-mask)
            ops.box_mix(ctx["uimg_a"], ctx["img_a"], net_input_mix[lsub:], self.box)       # uimg_a*mask + img_a*(1-mask)
        lab_a, lab_b = ctx["lab_a"], ctx["lab_b"]
        plab_a1, plab_b1 = plab1[:usub], plab1[usub:]
        plab_a2, plab_b2 = plab2[:usub], plab2[usub:]
        out_mix1, out_mix2 = model(net_input_mix, drop_masks=inject.get("drop_B"))
        d1, d2 = torch.empty_like(out_mix1), torch.empty_like(out_mix2)
        terms = (  # (logits, dlogits, img_l, patch_l, unlab)
            (out_mix1[lsub:], d1[lsub:], plab_a2, lab_a, True),      # mix_loss1: out_unl1
            (out_mix2[lsub:], d2[lsub:], plab_a1, lab_a, True),      # mix_loss2: out_unl2
            (out_mix1[:lsub], d1[:lsub], lab_b, plab_b2, False),     # mix_loss3: out_l1
            (out_mix2[:lsub], d2[:lsub], lab_b, plab_b1, False),     # mix_loss4: out_l2
        )
        losses = []
        split = self.gradsim is not None and inject.get("sim_score") is None and self.sim_score is None
        if split:       # loss_l / loss_u (:352-353) are taken apart: e* holds the unlabeled-supervised parts' gradient
            e1, e2 = torch.empty_like(out_mix1), torch.empty_like(out_mix2)
            eterms = (e1[lsub:], e2[lsub:], e1[:lsub], e2[:lsub])
        for ti, (lg, dl, img_l, patch_l, unlab) in enumerate(terms):
            iw, pw = (0.5, 1.0) if unlab else (1.0, 0.5)            # l_weight=1.0, u_weight=0.5 (:198-203)
            loss3, acc = ops.mix_loss_fwd(lg, img_l, patch_l, loss_mask, iw, pw)
            if split:   # (loss_image, loss_patch) = (loss_u_out, loss_l_in) for the unlabeled rows, (loss_l_out, loss_u_in) for the labeled ones (:345-349)
                wl, wu = ((0.0, pw), (iw, 0.0)) if unlab else ((iw, 0.0), (0.0, pw))
                ops.mix_loss_bwd(lg, img_l, patch_l, loss_mask, wl[0], wl[1], acc, dl)
                ops.mix_loss_bwd(lg, img_l, patch_l, loss_mask, wu[0], wu[1], acc, eterms[ti])
            else:
                ops.mix_loss_bwd(lg, img_l, patch_l, loss_mask, iw, pw, acc, dl)
            losses.append(loss3)
        yield
        if not split:
            torch.autograd.backward([out_mix1, out_mix2], [d1, d2])
            return losses
        # two backward passes over the saved forward (linear in dlogits): gradients of loss_l and of loss_u into their own
        # buffers -> the channel scores of the NEXT iteration (gradsim.get_grad_convkernel, :365); their sum is the BCP
        # gradient the single pass would have produced
        n_ = self._grad_lu.numel() // 2
        g_l, g_u = self._grad_lu[:n_], self._grad_lu[n_:]
        self._grad_lu.zero_()
        model.backward_saved(out_mix1, [d1, d2], g_l)
        model.backward_saved(out_mix1, [e1, e2], g_u)
        model.release_saved(out_mix1)
        self._new_scores_from = (g_l, g_u)
        b0 = self.grad_both[:n_]
        ops.perturb(b0, g_l, b0, 1.0)
        ops.perturb(b0, g_u, b0, 1.0)
        return losses

    def _phase_v_steps(self, ctx):
        a = self.args
        if not a["adv_noise"]:
            return torch.zeros(1, dtype=torch.float32, device=ctx["volume_batch"].device)
        return (yield from self.adv_loss.finish_steps(self.model, ctx["vat_state"], ctx["outputs_soft1"], ctx["outputs_soft2"], ctx["diff_mask"], a["adv_losstype"],
                                                      weight_dev=self.cw_dev, grad_buffer=self.grad2, fork_ctl=self._fork_ctl))

    def _fp_branch(self, uimg_ab, pseudo1, pseudo2, inject, capturing):
        """"2) fp" of the loop (train_ours_2D.py:359-365, default off): both decoders on the channel-perturbed features
        (DualDecoder.forward(dropout=True)), cross-entropy against the other head's pseudo labels, weighted with the
        consistency weight like the VAT term (:378).  Upstream compares the 1.5 U logits with U labels (a shape error) and
        reads the scores from the absent grad.GradSim: here every output row is paired with its own sample's pseudo label,
        cat(pseudo, pseudo[U/2:]), and the scores are `self.sim_score`.  Runs after the VAT branch on the main stream
        and accumulates into the second gradient bucket; the consistency weight is read from device memory (`cw_dev`), so
        the branch is part of a captured iteration like everything else."""
        a, model = self.args, self.model
        if self.dims != 2:
            raise NotImplementedError("chap_amd: the dropout (fp_loss) branch exists for the 2D DualDecoder only, as upstream")
        U = uimg_ab.shape[0]
        scores = inject.get("sim_score", self.sim_score)
        produced = scores is None and self.gradsim is not None
        if produced:
            scores = self.gradsim.get_sim()                     # of the previous iteration (:360)
        o1, o2 = model(uimg_ab, False, True, [0, 1, 2, 3, 4], scores, a["comp_drop"],
                       drop_masks=inject.get("drop_FP"), drop_uniforms=inject.get("fp_uniforms"),
                       drop_branches=inject.get("fp_branches"), grad_buffer=self.grad2)
        t1, t2 = torch.cat((pseudo1, pseudo1[U // 2:])), torch.cat((pseudo2, pseudo2[U // 2:]))
        losses, ds = [], []
        for o, t in ((o1, t2), (o2, t1)):
            l3, acc = ops.mix_loss_fwd(o, t, None, None, 1.0, 0.0, k_dice=0.0, k_ce=1.0)        # mean cross-entropy
            d = torch.empty_like(o)
            ops.mix_loss_bwd(o, t, None, None, 1.0, 0.0, acc, d, gscale_dev=self.cw_dev, k_dice=0.0, k_ce=1.0)
            losses.append(l3[0:1])
            ds.append(d)
        torch.autograd.backward([o1, o2], ds)
        if produced:                                            # :365, after the perturbed pass has read the old scores
            self.gradsim.get_grad_convkernel(self._new_scores_from[0], self._new_scores_from[1], model, self.opt, self.iter_num)
        return losses

    def step(self, volume_batch, label_batch, box_yx=None, inject=None):
        self._hw = tuple(volume_batch.shape[2:])
        self.prepare(box_yx)
        out = self.device_step(volume_batch, label_batch, inject)
        self.finish()
        return out

    # ------------------------------------------------------------------ HIP graph capture
    def capture(self, volume_batch, label_batch, warmup=3, inject=None, restore=True):
        """Capture device_step() into one HIP graph over static input buffers; afterwards call
        replay(volume_batch, label_batch).  The `warmup` eager iterations (allocator, packed-weight tables, lazily set
        kernel attributes: nothing of that may happen for the first time under capture) run on the given batch; with
        `restore` (default) parameters, momentum, BatchNorm statistics, iter_num / LR and the RNG state are put back
        afterwards, so capture() does not train -- without it they count as `warmup` real iterations.  The captured pass
        itself is not executed: iter_num == number of applied updates at all times.  `inject` (tests): static tensors
        (dropout masks, VAT noise) the captured iteration reads instead of drawing its own."""
        check_graph_environment(self.concurrent)
        self._hw = tuple(volume_batch.shape[2:])
        self._static_v = volume_batch.clone()
        self._static_l = label_batch.clone()
        snap = self.state_dict() if (restore and warmup > 0) else None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.prepare()
                self.device_step(self._static_v, self._static_l, inject)
                self.finish()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if snap is not None:
            self.load_state_dict(snap)
            torch.cuda.synchronize()
        self.model._rng.reset_counter()
        np_state = np.random.get_state()
        self.prepare()                          # the schedule block must hold valid values while the graph is being captured ...
        if restore:
            np.random.set_state(np_state)       # ... but its BCP-box draw is not an iteration's: capture() leaves the numpy stream untouched
        # the four-graph form only when bucket 0 is to be all-reduced beside the VAT chain; the fold schedule exchanges once, at the
        # end: [compute graph] -> all-reduce -> [optimizer graph] keeps pass B and the VAT chain as forked branches of ONE graph
        if (self.grad_sync is not None and getattr(self.grad_sync, "overlap", False) and type(self)._iteration is ChapStep._iteration
                and self.concurrent and self.args["adv_noise"]):
            return self._capture_dp(inject)
        g = torch.cuda.CUDAGraph()
        dp = self.grad_sync is not None
        # thread_local: the RCCL watchdog thread polls events while we capture (global mode would abort on that)
        with torch.cuda.graph(g, stream=self._cap, capture_error_mode="thread_local"):
            self.model._rng.seed_dev.add_(1)
            self._static_out = self.device_step(self._static_v, self._static_l, inject, update=not dp)
        self._graph, self._graph_opt, self._graphs_dp = g, None, None
        if dp:          # data-parallel without the two-branch schedule: [compute graph] -> RCCL all-reduce (eager) -> [optimizer graph]
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g.pool(), stream=self._cap, capture_error_mode="thread_local"):
                self.opt.step(grad_scale=1.0 / self.world_size, grad2=self.grad2)
            self._graph_opt = g2
        return g

    def _capture_dp(self, inject):
        """Data-parallel capture: the iteration as FOUR graphs so that RCCL (never captured) can run between them and the
        all-reduce of bucket 0 overlaps the VAT chain (north_star):

            main:  [A: pass A, pseudo labels, mask] -> [V: VAT chain -> bucket 1] -> all-reduce(bucket 1) ----+-> [optimizer]
            side:                       wait A -> [B: LCC, BCP mix, pass B -> bucket 0] -> all-reduce(bucket 0) --^

        B and V are captured into SEPARATE memory pools (they run concurrently at replay: sharing a pool would let one
        reuse memory the other has freed during capture); what crosses graph boundaries (pass A's outputs) is kept alive
        by `self._dp_ctx`.  Each capture's origin stream forks its own second-decoder stream."""
        kw = dict(capture_error_mode="thread_local", stream=self._cap)
        stack = contextlib.ExitStack()
        gA, gB, gV, gO = (torch.cuda.CUDAGraph() for _ in range(4))
        with torch.cuda.graph(gA, **kw):
            self.model._rng.seed_dev.add_(1)
            stack.enter_context(self.model.hold_stat_shift())       # the snapshot copy is a node of graph A; held until V is captured
            with self._decoder_fork(torch.cuda.current_stream()):
                ctx = self._phase_a(self._static_v, self._static_l, inject)
        with torch.cuda.graph(gB, **kw):
            with self._decoder_fork(torch.cuda.current_stream()):
                losses = self._phase_b(ctx)
        with torch.cuda.graph(gV, **kw):
            with self._decoder_fork(torch.cuda.current_stream()):
                vat_loss = self._phase_v(ctx)
            out = {"mix_losses": losses, "vat_loss": vat_loss}
            if self.args["dropout"]:
                out["fp_losses"] = self._fp_branch(ctx["uimg_ab"], ctx["pseudo_outputs1"], ctx["pseudo_outputs2"], ctx["inject"], None)
        stack.close()
        with torch.cuda.graph(gO, **kw):
            self.opt.step(grad_scale=1.0 / self.world_size, grad2=self.grad2)
        self._dp_ctx, self._static_out = ctx, out
        self._graph, self._graph_opt, self._graphs_dp = None, None, (gA, gB, gV, gO)
        return gA

    def stage(self, volume_batch, label_batch):
        """Start the host-to-device copy of the NEXT batch on a copy stream, beside the iteration that is running: the loader of
        train_ours_2D.py:301-304 yields CPU tensors and `.cuda()`s them on the compute stream (19 MB per 2D iteration, 0.4 ms of PCIe time in front
        of every step; 3D 1.0 ms).  replay() without a batch then takes the staged one (a device-to-device copy of microseconds).  Pinned source
        tensors make the copy asynchronous to the host as well."""
        if getattr(self, "_stage_v", None) is None:
            self._stage_v, self._stage_l = torch.empty_like(self._static_v), torch.empty_like(self._static_l)
            self._copy_stream = torch.cuda.Stream(device=self._static_v.device)
            self._staged_evt, self._taken_evt = torch.cuda.Event(), None
        with torch.cuda.stream(self._copy_stream):
            if self._taken_evt is not None:
                self._copy_stream.wait_event(self._taken_evt)      # the previous staged batch has been moved into the static buffers
            self._stage_v.copy_(volume_batch, non_blocking=True)
            self._stage_l.copy_(label_batch, non_blocking=True)
            self._staged_evt.record(self._copy_stream)
        self._staged = True

    def replay(self, volume_batch=None, label_batch=None, box_yx=None):
        if volume_batch is None:
            if not getattr(self, "_staged", False):
                raise RuntimeError("chap_amd: replay() without a batch needs a stage()d one")
            main = torch.cuda.current_stream()
            main.wait_event(self._staged_evt)
            self._static_v.copy_(self._stage_v, non_blocking=True)
            self._static_l.copy_(self._stage_l, non_blocking=True)
            self._taken_evt = torch.cuda.Event()
            self._taken_evt.record(main)
            self._staged = False
        else:
            self._static_v.copy_(volume_batch, non_blocking=True)
            self._static_l.copy_(label_batch, non_blocking=True)
        self.prepare(box_yx)
        if self._graphs_dp is not None:
            gA, gB, gV, gO = self._graphs_dp
            main = torch.cuda.current_stream()
            gA.replay()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                gB.replay()
                self.grad_sync.start_first()        # all-reduce of bucket 0 on the side stream, beside graph V
            gV.replay()
            self.grad_sync.start()
            main.wait_stream(self._side)
            self.grad_sync.wait()
            gO.replay()
        else:
            self._graph.replay()
            if self._graph_opt is not None:
                self.grad_sync.start()
                self.grad_sync.wait()
                self._graph_opt.replay()
        # the optimizer inside the graph changed the weights on the device: the packed copies an EAGER forward
        # (validation, inference, step()) would otherwise reuse are stale -- the graph itself re-packs on every replay
        self.model.mark_params_dirty()
        self.finish()
        return self._static_out


class AblationStep(ChapStep):
    """One iteration of the ablation loop (train_ablation_2D.py:159-246): ONE full-batch forward, supervised
    0.5*(CE + Dice) of both heads on the labeled half (:171-176), cross pseudo supervision -- CE of each head against
    the other head's arg-max -- on the unlabeled half (:203-207,216-217), the same create_maskV1 + VAT2d pair as the main
    loop (:228-230) and  loss = model1_loss + model2_loss + cw * (w_adv * vat_loss + w_drop * fp_loss)  (:236), then
    SGD + poly LR.  `losses.DiceLoss` is absent upstream: the SSL4MIS definition (1 - (2 sum p t + s)/(sum p^2 + sum t +
    s), s = 1e-5, mean over classes) is used.  The shipped VAT call passes the full-batch soft outputs next to an
    unlabeled-half mask (shape-inconsistent); here, as in the main loop, VAT acts on the unlabeled half.  fp_loss (the
    `dropout` branch, :209-213) is 0 as upstream.  The consistency weight lives in device memory: capture()/replay() work."""

    def _iteration(self, volume_batch, label_batch, inject=None, update=True):
        a, model = self.args, self.model
        inject = inject or {}
        lbs = a["labeled_bs"]
        out1, out2 = model(volume_batch, drop_masks=inject.get("drop_F"))
        with torch.no_grad():
            soft1, soft2, arg1, arg2, knowledge = ops.pseudo_block(out1[lbs:].contiguous(), out2[lbs:].contiguous())
        d1, d2 = torch.empty_like(out1), torch.empty_like(out2)
        lab = label_batch[:lbs].contiguous()
        sup, cps = [], []
        for o, d, other in ((out1, d1, arg2), (out2, d2, arg1)):
            ol, ou = o[:lbs].contiguous(), o[lbs:].contiguous()
            dl, du = torch.empty_like(ol), torch.empty_like(ou)
            l3, acc = ops.mix_loss_fwd(ol, lab, None, None, 1.0, 0.0, smooth=1e-5)            # 0.5*(CE + Dice)
            ops.mix_loss_bwd(ol, lab, None, None, 1.0, 0.0, acc, dl, smooth=1e-5)
            c3, acc2 = ops.mix_loss_fwd(ou, other, None, None, 1.0, 0.0, k_dice=0.0, k_ce=1.0)  # mean CE vs the other head
            ops.mix_loss_bwd(ou, other, None, None, 1.0, 0.0, acc2, du, gscale_dev=self.cw_dev, k_dice=0.0, k_ce=1.0)
            d[:lbs].copy_(dl); d[lbs:].copy_(du)
            sup.append(l3[0:1]); cps.append(c3[0:1])
        torch.autograd.backward([out1, out2], [d1, d2])
        vat_loss = torch.zeros(1, dtype=torch.float32, device=volume_batch.device)
        if a["adv_noise"]:
            diff_mask = ops.diff_mask(arg1, arg2, knowledge, 4, a["topk1"])
            vat_loss = self.adv_loss(model, volume_batch, soft1, soft2, diff_mask, a["adv_losstype"], weight_dev=self.cw_dev,
                                     inject=inject, grad_buffer=self.grad2, weight=a.get("w_adv", 1.0))
        if update:
            self.exchange_and_update()
        return {"sup_losses": sup, "cps_losses": cps, "vat_loss": vat_loss, "consistency_weight": self._cw_host}

    def prepare(self, box_yx=None):
        self._cw_host = get_current_consistency_weight(self.iter_num // 150, self.args)      # no BCP box in this loop
        self._upload_sched([], self._cw_host)

    def replay(self, volume_batch=None, label_batch=None, box_yx=None):
        out = super().replay(volume_batch, label_batch, box_yx)
        out["consistency_weight"] = self._cw_host        # the one host-side entry of the (static) output dict
        return out
