"""One full CHAP iteration (pass A, BCP mix, four mix_loss terms, VAT power iteration + final pass,
SGD) on the GPU in fp32 mode against the CPU oracle with identical injected randomness."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from chap_amd.networks import DualDecoder
from chap_amd.train import ChapStep
from oracle import init as oinit
from oracle import train_step as ots

DEV = "cuda"


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def cl_masks(masks):
    return {k: v.permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(DEV) for k, v in masks.items()}


def update_agreement(sd, state, after):
    """How well one SGD update agrees with the oracle's, as a whole: (relative L2 error over all parameters, smallest
    per-tensor cosine, its key).  Conv biases in front of a train-mode BatchNorm are left out of the cosine: their true
    gradient is exactly zero, what both sides hold there is rounding noise."""
    num = den = 0.0
    cos_min, cos_key = 1.0, None
    for k, v in sd.items():
        if not v.is_floating_point() or k.endswith(("running_mean", "running_var")):
            continue
        ug = (after[k].cpu().double() - state[k].double()).flatten()
        uo = (v.detach().double() - state[k].double()).flatten()
        num += float(((ug - uo) ** 2).sum())
        den += float((uo ** 2).sum())
        if k.endswith(("conv_conv.0.bias", "conv_conv.4.bias")) or float(uo.norm()) == 0:
            continue
        c = float((ug * uo).sum() / (ug.norm() * uo.norm() + 1e-300))
        if c < cos_min:
            cos_min, cos_key = c, k
    return (num / den) ** 0.5, cos_min, cos_key


def test_iteration_matches_oracle():
    B, lbs, H, W = 8, 4, 64, 64
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1)
    state = oinit.dual_decoder_2d_state(301)
    vol, lab = ots.synthetic_batch(1337, lbs, U, H, W)
    inj_cpu = {"drop_A": oinit.drop_masks_2d(1, U, H, W), "drop_B": oinit.drop_masks_2d(2, lbs // 2 + U // 2, H, W),
               "drop_V0": oinit.drop_masks_2d(3, U, H, W), "drop_VF": oinit.drop_masks_2d(4, U, H, W),
               "d0": torch.rand(U, 1, H, W, generator=torch.Generator().manual_seed(5)) - 0.5}
    box = (7, 11)
    # ---- oracle
    sd = {k: v.clone() for k, v in state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
    ref = ots.iteration(sd, moms, vol, lab, box, iter_num=0, lr=0.01, args=args, inject=inj_cpu)
    # ---- HIP path
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m.load_state_dict(state, strict=True)
    step = ChapStep(m, args)
    inj = {k: (cl_masks(v) if k.startswith("drop") else v.to(DEV)) for k, v in inj_cpu.items()}
    out = step.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=inj)
    torch.cuda.synchronize()
    for got, want in zip(out["mix_losses"], ref["losses"]):
        assert relerr(got.cpu(), torch.stack([w.detach() for w in want])) < 2e-4
    assert relerr(out["vat_loss"].cpu(), ref["vat_loss"].reshape(1)) < 5e-3
    # parameters after the SGD step and BN running statistics (updated by passes A and B only)
    after = m.state_dict()
    worst, worst_key = 0.0, None
    for k, v in sd.items():
        if not v.is_floating_point():
            assert int(after[k]) == int(v), k
            continue
        d = (after[k].cpu().double() - v.detach().double()).abs().max().item()
        upd = (v.detach().double() - state[k].double()).abs().max().item()
        # compare the UPDATE (what the step changed) -- tolerance relative to the size of the update
        floor = 3e-7 * v.detach().abs().max().item()       # a few fp32 ulps (weight-decay-only updates)
        if upd > 0 and max(d - floor, 0.0) / upd > worst:
            worst, worst_key = max(d - floor, 0.0) / upd, k
    assert worst < 0.05, (worst, worst_key)          # measured 0.0085 (consistency weight 0.0067 at iteration 0: VAT barely counts)
    rel_l2, cos_min, cos_key = update_agreement(sd, state, after)
    assert rel_l2 < 0.01, rel_l2                     # measured 0.0014
    assert cos_min > 0.999, (cos_min, cos_key)       # measured 0.99997
    assert step.iter_num == 1 and abs(step.opt.param_groups[0]["lr"] - 0.01 * (1 - 1 / 30000) ** 0.9) < 1e-12


def test_graph_capture_replay_runs():
    torch.manual_seed(0)
    B, lbs, H, W = 8, 4, 64, 64
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train().set_compute_dtype(torch.bfloat16)
    step = ChapStep(m, dict(labeled_bs=lbs, batch_size=B))
    vol, lab = ots.synthetic_batch(7, lbs, B - lbs, H, W)
    vol, lab = vol.to(DEV), lab.to(DEV)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    step.capture(vol, lab)
    assert step.iter_num == 0 and all(torch.equal(v, m.state_dict()[k]) for k, v in sd0.items())     # capture() does not train
    before = m.flat_buffers()[0].clone()
    for _ in range(3):
        out = step.replay(vol, lab)
    torch.cuda.synchronize()
    after = m.flat_buffers()[0]
    assert step.iter_num == 3                              # == number of applied updates
    assert torch.isfinite(after).all() and (after - before).abs().max() > 0
    assert all(torch.isfinite(l).all() for l in out["mix_losses"]) and torch.isfinite(out["vat_loss"]).all()
    # an EAGER forward after graph replays must see the replayed weights (packed copies re-made): eval logits equal those of
    # a fresh model loaded from the checkpoint
    x = vol[:2]
    m.eval()
    with torch.no_grad():
        a1, a2 = m(x)
    fresh = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).eval().set_compute_dtype(torch.bfloat16)
    fresh.load_state_dict(m.state_dict(), strict=True)
    with torch.no_grad():
        b1, b2 = fresh(x)
    assert torch.equal(a1, b1) and torch.equal(a2, b2)


def test_iteration_3d_matches_oracle():
    """The same loop on 5-D tensors (DualDecoder3d, 2 classes, cuboid BCP box, 26-connected LCC)."""
    from chap_amd.networks import DualDecoder3d
    from oracle import nets as onets
    B, lbs, D, H, W = 4, 2, 16, 32, 16
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1, num_classes=2)
    state = oinit.dual_decoder_3d_state(401)
    vol, lab = ots.synthetic_batch_3d(1337, lbs, U, D, H, W)
    dm = lambda seed, n: oinit.drop_masks_3d(seed, n)
    inj_cpu = {"drop_A": dm(1, U), "drop_B": dm(2, lbs // 2 + U // 2), "drop_V0": dm(3, U), "drop_VF": dm(4, U),
               "d0": torch.rand(U, 1, D, H, W, generator=torch.Generator().manual_seed(5)) - 0.5}
    box = (2, 5, 3)
    sd = {k: v.clone() for k, v in state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
    ref = ots.iteration(sd, moms, vol, lab, box, iter_num=0, lr=0.01, args=args, inject=inj_cpu, net=onets.dual_decoder_3d)
    m = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(DEV).train()
    m.load_state_dict(state, strict=True)
    step = ChapStep(m, args)
    inj = {k: ({kk: (vv.float() * 2.0).to(DEV) for kk, vv in v.items()} if k.startswith("drop") else v.to(DEV)) for k, v in inj_cpu.items()}
    out = step.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=inj)
    torch.cuda.synchronize()
    for got, want in zip(out["mix_losses"], ref["losses"]):
        assert relerr(got.cpu(), torch.stack([w.detach() for w in want])) < 5e-4
    # VAT loss and the SGD update (tiny volumes: BatchNorm over 32-256 elements per channel at the deep levels makes fp32
    # training-mode gradients noisy on BOTH sides) are judged against an fp64 oracle, relative to the fp32 oracle's own distance
    # to it: tests/test_iteration_conditioning_gpu.py::test_small_3d_iteration_is_as_close_to_fp64_as_the_fp32_oracle[1]
    # (round 2 asserted `worst < 0.5` here)


def test_ablation_iteration_matches_oracle():
    """The second caller of the same kernels (train_ablation_2D.py:159-246): full-batch forward, supervised CE+Dice,
    cross pseudo supervision, create_maskV1 + VAT, SGD -- fp32 against the CPU oracle with injected randomness."""
    from chap_amd.train import AblationStep
    B, lbs, H, W = 8, 4, 64, 64
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1, w_adv=0.7)
    state = oinit.dual_decoder_2d_state(404)
    vol, lab = ots.synthetic_batch(4321, lbs, U, H, W)
    inj_cpu = {"drop_F": oinit.drop_masks_2d(11, B, H, W), "drop_V0": oinit.drop_masks_2d(13, U, H, W), "drop_VF": oinit.drop_masks_2d(14, U, H, W),
               "d0": torch.rand(U, 1, H, W, generator=torch.Generator().manual_seed(15)) - 0.5}
    it0 = 3000                                                   # consistency weight exp(-5 * 0.6^2) = 0.165: the CPS / VAT terms matter
    sd = {k: v.clone() for k, v in state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
    ref = ots.ablation_iteration(sd, moms, vol, lab, iter_num=it0, lr=0.01, args=args, inject=inj_cpu)
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m.load_state_dict(state, strict=True)
    step = AblationStep(m, args)
    step.iter_num = it0
    inj = {k: (cl_masks(v) if k.startswith("drop") else v.to(DEV)) for k, v in inj_cpu.items()}
    out = step.step(vol.to(DEV), lab.to(DEV), inject=inj)
    torch.cuda.synchronize()
    assert abs(out["consistency_weight"] - ref["consistency_weight"]) < 1e-12
    for got, want in zip(out["sup_losses"] + out["cps_losses"], ref["sup"] + ref["cps"]):
        assert relerr(got.cpu(), want.reshape(1)) < 2e-4
    assert relerr(out["vat_loss"].cpu(), ref["vat_loss"].reshape(1)) < 5e-3
    after = m.state_dict()
    worst, worst_key = 0.0, None
    for k, v in sd.items():
        if not v.is_floating_point():
            continue
        d = (after[k].cpu().double() - v.detach().double()).abs().max().item()
        upd = (v.detach().double() - state[k].double()).abs().max().item()
        floor = 3e-7 * v.detach().abs().max().item()
        if upd > 0 and max(d - floor, 0.0) / upd > worst:
            worst, worst_key = max(d - floor, 0.0) / upd, k
    # Every reduction on the device is fixed-order (ABI 4), so this number is the same on every run; it is what the
    # different summation ORDER of the CPU oracle costs on this small fixture, where rounding flips a few discrete decisions
    # (max-pool routing, arg-max targets, the top-k patch threshold): single ELEMENTS of an update move by a few per cent of
    # the tensor's largest update while the update as a whole does not (relative L2, per-tensor cosine below).
    assert worst < 0.5, (worst, worst_key)
    rel_l2, cos_min, cos_key = update_agreement(sd, state, after)
    assert rel_l2 < 0.03, rel_l2
    assert cos_min > 0.995, (cos_min, cos_key)
    assert step.iter_num == it0 + 1
    # the same iteration as a captured HIP graph (the consistency weight lives in device memory): bitwise the eager result
    m2 = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m2.load_state_dict(state, strict=True)
    step2 = AblationStep(m2, args)
    step2.iter_num = it0
    step2.capture(vol.to(DEV), lab.to(DEV), warmup=1, inject=inj)
    out2 = step2.replay(vol.to(DEV), lab.to(DEV))
    torch.cuda.synchronize()
    assert step2.iter_num == it0 + 1 and abs(out2["consistency_weight"] - ref["consistency_weight"]) < 1e-12
    after2 = m2.state_dict()
    assert all(torch.equal(after[k], after2[k]) for k in after), [k for k in after if not torch.equal(after[k], after2[k])][:5]


def test_iteration_with_channel_dropout_matches_oracle():
    """args['dropout'] (train_ours_2D.py:359-365, default off): the fp_loss term on the channel-perturbed features -- forward,
    cross-entropy against cat(pseudo, pseudo[U/2:]), backward into the second gradient bucket, one SGD step -- against the
    oracle with the same scripted draws.  VAT is switched off here so that the fp term is what the update shows."""
    from oracle import filter_dropout as ofd
    B, lbs, H, W = 8, 4, 64, 64
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1, dropout=True, adv_noise=False)
    state = oinit.dual_decoder_2d_state(505)
    vol, lab = ots.synthetic_batch(2468, lbs, U, H, W)
    _, scores, uniforms = ofd.fd_inputs(B=U)
    inj_cpu = {"drop_A": oinit.drop_masks_2d(1, U, H, W), "drop_B": oinit.drop_masks_2d(2, lbs // 2 + U // 2, H, W),
               "drop_FP": oinit.drop_masks_2d(6, U, H, W), "fp_uniforms": uniforms, "sim_score": scores}
    box = (5, 9)
    it0 = 3000                                                   # consistency weight 0.165: the fp term matters
    sd = {k: v.clone() for k, v in state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
    ref = ots.iteration(sd, moms, vol, lab, box, iter_num=it0, lr=0.01, args=args, inject=inj_cpu)
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m.load_state_dict(state, strict=True)
    step = ChapStep(m, args)
    step.iter_num = it0
    inj = {k: (cl_masks(v) if k.startswith("drop") else v) for k, v in inj_cpu.items()}
    out = step.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=inj)
    torch.cuda.synchronize()
    for got, want in zip(out["fp_losses"], ref["fp_losses"]):
        assert relerr(got.cpu(), want.reshape(1)) < 2e-4
    for got, want in zip(out["mix_losses"], ref["losses"]):
        assert relerr(got.cpu(), torch.stack([w.detach() for w in want])) < 2e-4
    after = m.state_dict()
    rel_l2, cos_min, cos_key = update_agreement(sd, state, after)
    assert rel_l2 < 0.03, rel_l2
    assert cos_min > 0.995, (cos_min, cos_key)
    # the update must really contain the fp term: without it the same step lands somewhere else
    m0 = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m0.load_state_dict(state, strict=True)
    step0 = ChapStep(m0, dict(args, dropout=False))
    step0.iter_num = it0
    step0.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=inj)
    k = "decoder1.up4.conv.conv_conv.4.weight"
    d_fp = (after[k] - m0.state_dict()[k]).abs().max().item()
    upd = (after[k].cpu() - state[k]).abs().max().item()
    assert d_fp > 0.02 * upd, (d_fp, upd)
    # BN running statistics also saw the perturbed pass (a plain train-mode forward upstream)
    for k in ("encoder.in_conv.conv_conv.1.running_mean", "decoder2.up4.conv.conv_conv.5.running_var"):
        assert relerr(after[k].cpu(), sd[k]) < 1e-3, k
    # captured (the fp term reads the consistency weight from device memory): bitwise the eager result
    m2 = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m2.load_state_dict(state, strict=True)
    step2 = ChapStep(m2, args)
    step2.iter_num = it0
    inj2 = dict(inj, fp_uniforms=[(a.to(DEV), b.to(DEV)) for a, b in uniforms], sim_score=[sc.to(DEV) for sc in scores])   # no host copies under capture
    step2.capture(vol.to(DEV), lab.to(DEV), warmup=1, inject=inj2)
    step2.replay(vol.to(DEV), lab.to(DEV), box_yx=box)
    torch.cuda.synchronize()
    after2 = m2.state_dict()
    assert all(torch.equal(after[k], after2[k]) for k in after), [k for k in after if not torch.equal(after[k], after2[k])][:5]


def test_dropout_masks_do_not_depend_on_the_issue_order():
    """Regression (round 2): the executor issues the second decoder's ops first when it may fork a stream for it and in program
    order when it may not (captured pass on a non-origin stream), and the dropout seeds used to be drawn in ISSUE order -- the
    same pass drew other masks depending on the stream it ran on.  Seeds are drawn in program order now: a train-mode forward
    with the device RNG gives bit-identical logits with and without the fork, for the 2D and the 3D network."""
    from chap_amd.networks import DualDecoder3d
    for make, shape in ((lambda: DualDecoder(1, 4, {"decoder_type": "mcnet"}), (4, 1, 64, 64)),
                        (lambda: DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True), (2, 1, 32, 32, 16))):
        torch.manual_seed(5)
        m = make().to(DEV).train()
        x = torch.randn(*shape, device=DEV)
        with torch.no_grad():
            m(x, update_stats=False)                            # builds the executor and its RNG
        outs = []
        for fork in (True, False):
            m._rng.reset_counter()
            if not fork:
                orig = m._exec._side_stream
                m._exec._side_stream = lambda parent: None
            with torch.no_grad():
                o = m(x, update_stats=False)
            if not fork:
                m._exec._side_stream = orig
            torch.cuda.synchronize()
            outs.append([t.clone() for t in o])
        assert all(torch.equal(a, b) for a, b in zip(*outs))
        assert not torch.equal(outs[0][0], outs[0][1])          # the two decoders do differ (dropout, mcnet up-sampling)


def test_chapstep_streams_are_distinct():
    """PyTorch hands out the streams of its per-device pool round-robin, so two 'new' streams of a long-lived process can be
    the same stream; the iteration's streams and the capture's origin stream are drawn until they differ."""
    keep = [torch.cuda.Stream() for _ in range(70)]             # walk the pool past its size
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    for _ in range(3):
        s = ChapStep(m, dict(labeled_bs=4, batch_size=8))
        handles = [t.cuda_stream for t in (s._side, s._d2, s._pre, s._cap)]
        assert len(set(handles)) == 4 and 0 not in handles and torch.cuda.current_stream().cuda_stream not in handles
    del keep


@pytest.mark.parametrize("dims", [2, 3])
def test_grouped_decoder_launches_equal_separate_launches(dims, monkeypatch):
    """Grouped launches (chap_hip.h, chap_group_*): in a captured iteration the passes that cannot fork a second stream (pass B, the early VAT
    pass) run the same-shaped layers of their two decoders as ONE grid (CHAP_GROUP=1, the default) instead of back to back (CHAP_GROUP=0).  A
    grouped grid runs the same blocks on the same data as the separate launches, so a replay of either graph and the eager step (decoders on two
    streams) agree BIT FOR BIT: losses, parameters, BatchNorm buffers, momentum."""
    from chap_amd import _lib as L
    from chap_amd.networks import DualDecoder3d
    from tests.iteration_parity import inject_2d, inject_3d, to_dev
    if dims == 2:
        B, lbs, sp = 8, 4, (64, 64)
        state = oinit.dual_decoder_2d_state(301)
        vol, lab = ots.synthetic_batch(1337, lbs, B - lbs, *sp)
        inj = to_dev(inject_2d(B - lbs, lbs // 2 + (B - lbs) // 2, sp[0], sp[1], 2), 2)
        mk, box, extra = (lambda: DualDecoder(1, 4, {"decoder_type": "mcnet"})), (7, 11), {}
    else:
        B, lbs, sp = 4, 2, (16, 32, 16)
        state = oinit.dual_decoder_3d_state(401)
        vol, lab = ots.synthetic_batch_3d(1337, lbs, B - lbs, *sp)
        inj = to_dev(inject_3d(B - lbs, lbs // 2 + (B - lbs) // 2, sp, 2), 3)
        mk, box, extra = (lambda: DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True)), (2, 5, 3), {"num_classes": 2}
    res = {}
    for mode in ("eager", "graph_grouped", "graph_separate"):
        monkeypatch.setenv("CHAP_GROUP", "0" if mode == "graph_separate" else "1")
        m = mk().to(DEV).train()
        m.load_state_dict(state, strict=True)
        step = ChapStep(m, dict(dict(labeled_bs=lbs, batch_size=B, vat_iters=2), **extra))
        step.iter_num = 4500
        n0 = L.group.launched
        if mode == "eager":
            out = step.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=inj)
        else:
            step.capture(vol.to(DEV), lab.to(DEV), warmup=1, inject=inj)
            n0 = L.group.launched                           # (count the regions of the capture itself, not of its eager warm-up)
            out = step.replay(vol.to(DEV), lab.to(DEV), box_yx=box)
        torch.cuda.synchronize()
        res[mode] = (out, {k: v.clone() for k, v in m.state_dict().items()}, step.opt.mom.clone(), L.group.launched - n0)
    for mode in ("graph_grouped", "graph_separate"):
        (oa, sa, ma, _), (ob, sb, mb, _) = res["eager"], res[mode]
        for x, y in zip(oa["mix_losses"] + [oa["vat_loss"]], ob["mix_losses"] + [ob["vat_loss"]]):
            assert torch.equal(x, y), (mode, x, y)
        assert [k for k in sa if not torch.equal(sa[k], sb[k])] == [], mode
        assert torch.equal(ma, mb), mode


def test_staged_host_batches_equal_direct_replay():
    """ChapStep.stage() + replay() (the next batch's host-to-device copy on a copy stream beside the running iteration, what train() does) against
    replay(volume, label) with device tensors: three iterations on three different batches, bit for bit -- the hand-over between the copy stream,
    the staging buffers and the graph's static inputs loses or reorders nothing."""
    B, lbs, sp = 8, 4, (64, 64)
    state = oinit.dual_decoder_2d_state(301)
    batches = [ots.synthetic_batch(1337 + k, lbs, B - lbs, *sp) for k in range(3)]
    res = {}
    for mode in ("direct", "staged"):
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
        m.load_state_dict(state, strict=True)
        step = ChapStep(m, dict(labeled_bs=lbs, batch_size=B, vat_iters=1))
        step.capture(batches[0][0].to(DEV), batches[0][1].to(DEV), warmup=1)
        outs = []
        if mode == "staged":
            pinned = [(v.pin_memory(), l.pin_memory()) for v, l in batches]
            step.stage(*pinned[0])
            for k in range(3):
                out = step.replay(box_yx=(7, 11))
                if k + 1 < 3:
                    step.stage(*pinned[k + 1])          # enqueued while iteration k runs
                outs.append([x.clone() for x in out["mix_losses"]] + [out["vat_loss"].clone()])
            with pytest.raises(RuntimeError, match="stage"):
                step.replay()                           # nothing staged
        else:
            for k in range(3):
                out = step.replay(batches[k][0].to(DEV), batches[k][1].to(DEV), box_yx=(7, 11))
                outs.append([x.clone() for x in out["mix_losses"]] + [out["vat_loss"].clone()])
        torch.cuda.synchronize()
        res[mode] = (outs, {k: v.clone() for k, v in m.state_dict().items()})
    for a, b in zip(res["direct"][0], res["staged"][0]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert [k for k in res["direct"][1] if not torch.equal(res["direct"][1][k], res["staged"][1][k])] == []
