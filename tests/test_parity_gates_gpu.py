"""Parity gates on the path bench.py times (VERDICT r1 "next round" item 1):

  (a) one captured HIP-graph replay == one eager step() (bitwise: every reduction on the device is fixed-order) == the
      CPU oracle (same bounds as test_iteration_matches_oracle), fp32, injected masks / VAT noise as static buffers;
  (b) K = 2 perturbation power iterations (BASELINE config 4) against the oracle, 2D and 3D, and the VAT variants the
      reference's flags offer (--adv_losstype dice, train_ours_2D.py:515) / north_star names (FGSM-style sign step) at
      iteration level;
  (c) full-size property tests THROUGH the graph path at the BASELINE sizes (2D B=24 256x256; 3D B=4 112x112x80): bf16 and
      fp32 from the same state -- losses finite, the bf16 losses within a stated bound of the fp32 run, BatchNorm running
      statistics within 5e-2, run-to-run bitwise reproducibility, largest-CC / BCP box kernels equal to the CPU oracle;
  (d) a Dice gate for the bf16 throughput mode: N iterations in bf16 and in fp32 from one seed, Dice of the two
      checkpoints on held-out slices within a stated, measured bound (replaces the vacuous 2.0-on-probabilities bound).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from chap_amd import ops
from chap_amd.networks import DualDecoder, DualDecoder3d
from chap_amd.train import ChapStep
from oracle import init as oinit
from oracle import nets as onets
from oracle import train_step as ots
from tests.test_train_step_gpu import cl_masks, relerr, update_agreement

DEV = "cuda"


def _oracle_state(state):
    sd = {k: v.clone() for k, v in state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
    return sd, moms


def _inject_2d(U, lbs, H, W, K):
    inj = {"drop_A": oinit.drop_masks_2d(1, U, H, W), "drop_B": oinit.drop_masks_2d(2, lbs // 2 + U // 2, H, W),
           "drop_VF": oinit.drop_masks_2d(4, U, H, W), "d0": torch.rand(U, 1, H, W, generator=torch.Generator().manual_seed(5)) - 0.5}
    for k in range(K):
        inj["drop_V%d" % k] = oinit.drop_masks_2d(30 + k, U, H, W)
    return inj


def _to_dev_2d(inj):
    return {k: (cl_masks(v) if k.startswith("drop") else v.to(DEV)) for k, v in inj.items()}


def _equal_states(a, b):
    return [k for k in a if not torch.equal(a[k], b[k])]


# ------------------------------------------------------------------------------------------------ (a)
def test_replay_equals_eager_equals_oracle():
    B, lbs, H, W = 8, 4, 64, 64
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1)
    state = oinit.dual_decoder_2d_state(301)
    vol, lab = ots.synthetic_batch(1337, lbs, U, H, W)
    inj_cpu = _inject_2d(U, lbs, H, W, 1)
    box = (7, 11)
    it0 = 3000                                       # consistency weight 0.165: the VAT term is part of the update
    sd, moms = _oracle_state(state)
    ref = ots.iteration(sd, moms, vol, lab, box, iter_num=it0, lr=0.01, args=args, inject=inj_cpu)
    inj = _to_dev_2d(inj_cpu)

    def fresh():
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
        m.load_state_dict(state, strict=True)
        st = ChapStep(m, args)
        st.iter_num = it0
        return m, st

    m_e, s_e = fresh()
    out_e = s_e.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=inj)
    m_g, s_g = fresh()
    s_g.capture(vol.to(DEV), lab.to(DEV), warmup=2, inject=inj)
    assert s_g.iter_num == it0 and not _equal_states(m_g.state_dict(), {k: v.to(DEV) for k, v in state.items()})      # capture() does not train
    out_g = s_g.replay(vol.to(DEV), lab.to(DEV), box_yx=box)
    torch.cuda.synchronize()
    # replay == eager, bit for bit (losses, parameters, BatchNorm buffers, momentum)
    for a, b in zip(out_e["mix_losses"] + [out_e["vat_loss"]], out_g["mix_losses"] + [out_g["vat_loss"]]):
        assert torch.equal(a, b), (a, b)
    assert not _equal_states(m_e.state_dict(), m_g.state_dict())
    assert torch.equal(s_e.opt.mom, s_g.opt.mom) and s_e.iter_num == s_g.iter_num == it0 + 1
    # ... == oracle
    for got, want in zip(out_g["mix_losses"], ref["losses"]):
        assert relerr(got.cpu(), torch.stack([w.detach() for w in want])) < 2e-4
    assert relerr(out_g["vat_loss"].cpu(), ref["vat_loss"].reshape(1)) < 5e-3
    rel_l2, cos_min, cos_key = update_agreement(sd, state, m_g.state_dict())
    assert rel_l2 < 0.02, rel_l2
    assert cos_min > 0.998, (cos_min, cos_key)
    for k in ("encoder.in_conv.conv_conv.1.running_mean", "decoder2.up4.conv.conv_conv.5.running_var", "encoder.down4.maxpool_conv.1.conv_conv.5.running_mean"):
        assert relerr(m_g.state_dict()[k].cpu(), sd[k]) < 1e-4, k
    # a second replay continues the run exactly like a second eager step
    vol2, lab2 = ots.synthetic_batch(99, lbs, U, H, W)
    s_e.step(vol2.to(DEV), lab2.to(DEV), box_yx=(3, 4), inject=inj)
    s_g.replay(vol2.to(DEV), lab2.to(DEV), box_yx=(3, 4))
    torch.cuda.synchronize()
    assert not _equal_states(m_e.state_dict(), m_g.state_dict())


# ------------------------------------------------------------------------------------------------ (b)
@pytest.mark.parametrize("variant", ["k2_dice_sign"])      # (k2, dice, sign: tests/test_iteration_conditioning_gpu.py, against fp64, several seeds)
def test_vat_variants_iteration_matches_oracle(variant):
    B, lbs, H, W = 8, 4, 64, 64
    U = B - lbs
    K = 2 if "k2" in variant else 1
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=K, adv_losstype="dice" if "dice" in variant else "kl", vat_sign="sign" in variant)
    state = oinit.dual_decoder_2d_state(611)
    vol, lab = ots.synthetic_batch(1441, lbs, U, H, W)
    inj_cpu = _inject_2d(U, lbs, H, W, K)
    box, it0 = (9, 4), 4500                         # consistency weight 0.449
    sd, moms = _oracle_state(state)
    ref = ots.iteration(sd, moms, vol, lab, box, iter_num=it0, lr=0.01, args=args, inject=inj_cpu)
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m.load_state_dict(state, strict=True)
    step = ChapStep(m, args)
    step.iter_num = it0
    out = step.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=_to_dev_2d(inj_cpu))
    torch.cuda.synchronize()
    for got, want in zip(out["mix_losses"], ref["losses"]):
        assert relerr(got.cpu(), torch.stack([w.detach() for w in want])) < 2e-4
    # The VAT loss and the SGD update of these variants are ill-conditioned on this tiny batch (K power iterations through
    # training-mode BatchNorm over 4 x 64 x 64 values, sign(d) discontinuous where d ~ 0): they are judged against an fp64 oracle,
    # relative to the fp32 oracle's own distance to it, in tests/test_iteration_conditioning_gpu.py (same state, same variants) --
    # no absolute bound is asserted here any more (round 2 had widened them to 5e-2 / 0.15 / 0.97 after red runs).
    # the variant really is a different computation from the default iteration
    sd0, moms0 = _oracle_state(state)
    base_args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1)
    inj0 = dict(inj_cpu)
    ref0 = ots.iteration(sd0, moms0, vol, lab, box, iter_num=it0, lr=0.01, args=base_args, inject=inj0)
    assert abs(float(ref0["vat_loss"]) - float(ref["vat_loss"])) > 1e-3 * abs(float(ref0["vat_loss"]))


def test_k2_iteration_3d_matches_oracle():
    """BASELINE config 4's inner loop (2 perturbation power iterations) on the 3D net, fp32, against the oracle."""
    B, lbs, D, H, W = 4, 2, 16, 32, 16
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=2, num_classes=2)
    state = oinit.dual_decoder_3d_state(402)
    vol, lab = ots.synthetic_batch_3d(1338, lbs, U, D, H, W)
    dm = lambda seed, n: oinit.drop_masks_3d(seed, n)     # noqa: E731
    inj_cpu = {"drop_A": dm(1, U), "drop_B": dm(2, lbs // 2 + U // 2), "drop_V0": dm(3, U), "drop_V1": dm(6, U), "drop_VF": dm(4, U),
               "d0": torch.rand(U, 1, D, H, W, generator=torch.Generator().manual_seed(5)) - 0.5}
    box, it0 = (2, 5, 3), 4500
    sd, moms = _oracle_state(state)
    ref = ots.iteration(sd, moms, vol, lab, box, iter_num=it0, lr=0.01, args=args, inject=inj_cpu, net=onets.dual_decoder_3d)
    m = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(DEV).train()
    m.load_state_dict(state, strict=True)
    step = ChapStep(m, args)
    step.iter_num = it0
    inj = {k: ({kk: (vv.float() * 2.0).to(DEV) for kk, vv in v.items()} if k.startswith("drop") else v.to(DEV)) for k, v in inj_cpu.items()}
    out = step.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=inj)
    torch.cuda.synchronize()
    for got, want in zip(out["mix_losses"], ref["losses"]):
        assert relerr(got.cpu(), torch.stack([w.detach() for w in want])) < 5e-4
    # VAT loss and update: tests/test_iteration_conditioning_gpu.py::test_small_3d_iteration_is_as_close_to_fp64_as_the_fp32_oracle[2]
    # (distance to an fp64 oracle relative to the fp32 oracle's own)


# ------------------------------------------------------------------------------------------------ (c)
def _full_size_run(cfg, dtype, steps=3):
    torch.manual_seed(1337)
    np.random.seed(1337)
    if cfg == "2d":
        B, sp = 24, (256, 256)
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"})
        args = dict(batch_size=B, labeled_bs=B // 2, vat_iters=1)
        vol, lab = ots.synthetic_batch(1337, B // 2, B // 2, *sp)
    else:
        B, sp = 4, (112, 112, 80)
        m = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True)
        args = dict(batch_size=B, labeled_bs=B // 2, vat_iters=1, num_classes=2)
        vol, lab = ots.synthetic_batch_3d(1337, B // 2, B // 2, *sp)
    m = m.to(DEV).train().set_compute_dtype(dtype)
    step = ChapStep(m, args)
    step.iter_num = 4500                              # consistency weight 0.449: the VAT term counts
    vol, lab = vol.to(DEV), lab.to(DEV)
    step.capture(vol, lab, warmup=2)
    losses, stats1 = [], None
    for i in range(steps):
        out = step.replay(vol, lab)
        losses.append(torch.cat([l[2:3] for l in out["mix_losses"]] + [out["vat_loss"]]).clone())
        if i == 0:      # BatchNorm running statistics after ONE iteration from identical weights: arithmetic differences only
            stats1 = {k: v.clone() for k, v in m.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    torch.cuda.synchronize()
    return m, step, torch.stack(losses).cpu(), (vol, lab), stats1


@pytest.mark.parametrize("cfg", ["2d", "3d"])
def test_full_size_graph_path_properties(cfg):
    """The configuration bench.py times (graph replay at the BASELINE sizes), bf16 AND fp32 from the same initial state."""
    m32, s32, l32, (vol, lab), st32 = _full_size_run(cfg, torch.float32)
    m16, s16, l16, _, st16 = _full_size_run(cfg, torch.bfloat16)
    m16b, s16b, l16b, _, _ = _full_size_run(cfg, torch.bfloat16)
    assert torch.isfinite(l32).all() and torch.isfinite(l16).all()
    # run-to-run: bitwise (no atomics anywhere; the two streams of an iteration write disjoint buffers)
    assert torch.equal(l16, l16b) and not _equal_states(m16.state_dict(), m16b.state_dict())
    # bf16 against fp32 on the same data, same initial weights, same RNG epochs: the four mix_loss totals and the VAT loss
    bcp32, bcp16 = l32[:, :4].sum(1), l16[:, :4].sum(1)
    assert ((bcp16 - bcp32).abs() / bcp32.abs()).max() < 0.03, (bcp16, bcp32)       # measured: see DESIGN.md section 8
    assert ((l16[:, 4] - l32[:, 4]).abs() / l32[:, 4].abs().clamp_min(1e-6)).max() < 0.25, (l16[:, 4], l32[:, 4])
    # BatchNorm running statistics after the FIRST iteration (identical weights on both sides, so what differs is the bf16
    # storage of the activations; later iterations also carry the divergence of two training trajectories through arg-max
    # pseudo labels, which is not an arithmetic property).  One EMA update with momentum 0.1 per pass that tracks statistics.
    for k in st32:
        if k.endswith("running_mean"):           # in units of the channel's standard deviation
            scale = st32[k.replace("running_mean", "running_var")].sqrt()
            assert ((st16[k] - st32[k]).abs() / scale).max() < 5e-2, k
        else:
            assert ((st16[k] - st32[k]).abs() / st32[k].abs().clamp_min(1e-3)).max() < 5e-2, k
    # discrete kernels of the path at full size against the CPU oracle: arg-max pseudo labels of the trained fp32 model ->
    # largest connected component (scipy, full connectivity), BCP box mask / mixing
    with torch.no_grad():
        pre1, pre2 = m32(vol[vol.shape[0] // 2:], update_stats=False)
        _, _, a1, a2, know = ops.pseudo_block(pre1, pre2)
    nc = pre1.shape[1]
    got = ops.largest_cc(a1, nc)
    assert torch.equal(got.cpu(), ots.largest_cc(a1.cpu(), nc))
    assert torch.equal(ops.largest_cc(got, nc), got)                                  # idempotent
    box = torch.tensor([5, 9, 170, 170] if cfg == "2d" else [3, 5, 9, 74, 74, 53], dtype=torch.int32, device=DEV)
    lsub = vol.shape[0] // 4
    mask = torch.empty((lsub,) + tuple(vol.shape[2:]), dtype=torch.int64, device=DEV)
    ops.box_mask(mask, box)
    ref_mask = (ots.box_masks(lsub, *vol.shape[2:], 5, 9)[1] if cfg == "2d" else ots.box_masks_3d(lsub, *vol.shape[2:], 3, 5, 9)[1])
    assert torch.equal(mask.cpu(), ref_mask.long())
    mixed = torch.empty_like(vol[:lsub])
    ops.box_mix(vol[:lsub], vol[lsub:2 * lsub], mixed, box)
    mm = ref_mask.unsqueeze(1).to(DEV)
    assert torch.equal(mixed, vol[:lsub] * mm + vol[lsub:2 * lsub] * (1 - mm))
    # the perturbation mask at full size: the selected fraction is top-k of the pooled map OR disagreement
    dmask = ops.diff_mask(a1, a2, know, 4, 0.1)
    assert torch.equal(dmask.cpu(), ots.create_mask_v1(a1.cpu(), a2.cpu(), know.cpu(), 4, 0.1))


# ------------------------------------------------------------------------------------------------ (d)
def _dice(pred, gt, n_classes=4):
    out = []
    for c in range(1, n_classes):
        p, g = pred == c, gt == c
        den = p.sum() + g.sum()
        out.append(2.0 * float((p & g).sum()) / float(den) if den > 0 else 1.0)
    return np.array(out)


def _train_and_dice(dtype, rng_seed, B, H, W, iters, lr=0.05):
    """The same schedule from the same initial weights (manual_seed(1337) default init), data pool and BCP boxes; `rng_seed`
    seeds the device RNG of the dropout masks and the VAT noise.  Returns per-class Dice of the logit-ensemble prediction
    (test_2D_fully.py:69-75) on 24 held-out slices."""
    lbs = B // 2
    pool = [ots.synthetic_batch(2000 + i, lbs, B - lbs, H, W) for i in range(16)]
    pool = [(v.to(DEV), l.to(DEV)) for v, l in pool]
    val, gt = ots.synthetic_batch(4242, 24, 0, H, W)
    torch.manual_seed(1337)
    np.random.seed(1337)
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train().set_compute_dtype(dtype)
    torch.manual_seed(rng_seed)                       # the model's RNG (dropout, VAT noise) is created from torch's seed at first use
    step = ChapStep(m, dict(labeled_bs=lbs, batch_size=B, base_lr=lr, max_iterations=iters))     # poly LR runs down to 0: a converged checkpoint
    step.capture(*pool[0], warmup=1)
    for it in range(iters):
        step.replay(*pool[it % len(pool)])
    m.eval()
    with torch.no_grad():
        o1, o2 = m(val.to(DEV))
    pred = torch.argmax(torch.softmax((o1 + o2) / 2.0, dim=1), dim=1).cpu().numpy()
    return _dice(pred, gt.numpy())


def _log_dice_gate(rec):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_dice_gate.jsonl")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    print(json.dumps(rec))


# The bf16-vs-fp32 statement is a STATISTIC (round 4, tools/dice_pairs.py -> profiles/r04_dice_pairs.json; 6 seeds, each seed trained in fp32 and in
# bf16 from the same initial weights, data order, BCP boxes and device-RNG seed, 1 500 iterations at 256 x 256, B = 24).  Measured TWICE in round 4:
#     start of the round   fp32 0.7466 +- 0.0150 (s.d. 0.037)   bf16 0.7775 +- 0.0023 (s.d. 0.0056)   paired bf16 - fp32 +0.031 +- 0.014, t = 2.23
#     last tree            fp32 0.7466 +- 0.0150 (unchanged:    bf16 0.7541 +- 0.0165 (s.d. 0.040)    paired bf16 - fp32 +0.0075 +- 0.0066, t = 1.14
#                          the fp32 kernels were not touched)
# (t < 2.57: no bias detectable at the 5 % level, both times).  The bf16 kernels changed in between (wave-private kernels, the first conv's own kernel: other
# summation orders) and with them every bf16 run: the first run's tiny bf16 spread (0.0056) was a property of that build's draws, not of bf16 -- the second
# has the same seed-to-seed spread as fp32.  A single (fp32, bf16) pair is one draw from a distribution with s.d. 0.016-0.034 and cannot carry a 1e-3 claim
# (round 3 moved a one-pair bound 0.02 -> 0.037 -> 0.085 behind its measurements).  What a default-suite run CAN detect is a broken bf16 path (Dice 0-0.3):
# the gate is "the bf16 run learns", 0.62 = the measured bf16 mean minus 3.3 of its seed-to-seed standard deviations; the paired statistic itself is
# re-measured by tools/dice_pairs.py (CHAP_DICE_PAIRS=1 runs it here with the bounds fixed beforehand: |mean paired difference| <= 3 s.e.m. + 0.01 of the
# first recorded statistic, i.e. 0.052).
DICE_GATE = {"64": dict(B=8, H=64, W=64, iters=1500), "256": dict(B=24, H=256, W=256, iters=1500, bf16_mean=0.7541, bf16_sd=0.0404)}


@pytest.mark.parametrize("size", ["64", "256"])
def test_bf16_training_dice_gate(size):
    """bf16 is the throughput mode bench.py times: train the schedule in bf16 (at 64 x 64 also in fp32, from the same seed: graph replay, device RNG --
    identical dropout masks, VAT noise and BCP boxes) until the model segments the synthetic slices, and evaluate the reference's inference recipe
    (test_2D_fully.py:69-75) on held-out slices.  north_star's 1e-3 Dice is a statement about one checkpoint evaluated on both sides
    (tests/test_training_parity_gpu.py holds it there); between two training runs it is below the seed-to-seed spread of fp32 itself (see above)."""
    c = DICE_GATE[size]
    d16 = _train_and_dice(torch.bfloat16, 1337, c["B"], c["H"], c["W"], c["iters"])
    rec = {"size": "%dx%d" % (c["H"], c["W"]), "batch": c["B"], "iterations": c["iters"], "dice_bf16": d16.round(5).tolist(), "mean_bf16": round(float(d16.mean()), 5)}
    if size == "64":
        d32 = _train_and_dice(torch.float32, 1337, c["B"], c["H"], c["W"], c["iters"])
        rec.update(dice_fp32=d32.round(5).tolist(), mean_fp32=round(float(d32.mean()), 5), abs_delta_bf16_vs_fp32=round(abs(float(d32.mean() - d16.mean())), 5))
        _log_dice_gate(rec)
        assert d32.mean() > 0.6 and d16.mean() > 0.6, (d32, d16)          # both runs learned to segment
        assert d16.mean() > d32.mean() - 0.05, rec                         # bf16 not materially BELOW fp32 (one pair: no tighter claim, see above)
    else:
        _log_dice_gate(rec)
        assert d16.mean() > c["bf16_mean"] - 3.3 * c["bf16_sd"], rec      # 0.62: 3.3 s.d. of the bf16 seed spread below its measured mean (this seed: 0.771 on the last tree)


@pytest.mark.skipif(os.environ.get("CHAP_DICE_PAIRS") != "1", reason="the paired multi-seed statistic (about 5 min): CHAP_DICE_PAIRS=1, or python tools/dice_pairs.py")
def test_bf16_dice_pairs_statistic():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out", "r04_dice_pairs_test.json")
    subprocess.run([sys.executable, os.path.join(root, "tools", "dice_pairs.py"), "--seeds", "6", "--out", out], check=True, timeout=1200)
    d = json.load(open(out))
    assert d["bf16"]["mean"] > 0.7 and d["fp32"]["mean"] > 0.7
    assert abs(d["paired_diff_bf16_minus_fp32"]["mean"]) <= 0.052, d["paired_diff_bf16_minus_fp32"]       # 3 s.e.m. + 0.01 of profiles/r04_dice_pairs.json, fixed before this run
    assert d["bf16"]["mean"] > d["fp32"]["mean"] - 0.03


# ------------------------------------------------------------------------------------------------ N1: GradSim
def test_gradsim_scores_produced_inside_the_iteration():
    """ChapStep(dropout=True) with the channel scores PRODUCED (gradsim.get_sim() / get_grad_convkernel, train_ours_2D.py:360,365),
    not injected: iteration 0 runs on the initial all-zero scores (the Dropout2d pair) and leaves the per-channel cosine
    similarity of the labeled-loss and unlabeled-loss gradients; iteration 1 perturbs with those.  Against the oracle's
    restatement of the same definition (parity unpinned: grad.GradSim is absent upstream)."""
    from oracle import filter_dropout as ofd
    B, lbs, H, W = 8, 4, 64, 64
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1, dropout=True, adv_noise=False)
    state = oinit.dual_decoder_2d_state(515)
    _, _, uniforms = ofd.fd_inputs(B=U)
    sd, moms = _oracle_state(state)
    gs_oracle = [torch.zeros(c) for c in (16, 32, 64, 128, 256)]
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m.load_state_dict(state, strict=True)
    step = ChapStep(m, args)
    it0, lr = 3000, 0.01
    step.iter_num = it0
    step.opt.set_lr(lr)
    for it in range(2):
        vol, lab = ots.synthetic_batch(2468 + it, lbs, U, H, W)
        inj_cpu = {"drop_A": oinit.drop_masks_2d(10 * it + 1, U, H, W), "drop_B": oinit.drop_masks_2d(10 * it + 2, lbs // 2 + U // 2, H, W),
                   "drop_FP": oinit.drop_masks_2d(10 * it + 6, U, H, W), "fp_uniforms": uniforms}
        box = (5 + it, 9 - it)
        ref = ots.iteration(sd, moms, vol, lab, box, iter_num=it0 + it, lr=lr, args=args, inject=dict(inj_cpu, gradsim=gs_oracle))
        inj = {k: (cl_masks(v) if k.startswith("drop") else v) for k, v in inj_cpu.items()}
        out = step.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=inj)
        torch.cuda.synchronize()
        lr = ots.poly_lr(0.01, it0 + it + 1, 30000)
        for got, want in zip(out["fp_losses"], ref["fp_losses"]):
            assert relerr(got.cpu(), want.reshape(1)) < 1e-3, it
        # the scores the iteration leaves behind (cosines in [-1, 1]); rows whose gradient is tiny on both sides are noise
        for lvl, (got, want) in enumerate(zip(step.gradsim.get_sim(), gs_oracle)):
            assert got.abs().max() <= 1.0 + 1e-5 and float(got.abs().max()) > 0
            err = (got.cpu() - want).abs()
            assert err.mean() < 6e-2 and err.max() < 0.3, (it, lvl, float(err.mean()), float(err.max()))     # measured <= 0.037 / 0.15 (16 rows of 144 elements at level 0)
    rel_l2, cos_min, cos_key = update_agreement(sd, state, m.state_dict())
    assert rel_l2 < 0.05, rel_l2
    assert cos_min > 0.99, (cos_min, cos_key)
    # resume carries the scores
    ck = step.state_dict()
    assert all(torch.equal(a, b) for a, b in zip(ck["gradsim"], step.gradsim.get_sim()))


def test_train_entry_point_writes_the_reference_outputs(tmp_path):
    """train(args, snapshot_path) (code/train_ours_2D.py:219): latest.pth / {model}_best_model.pth are plain state dicts with
    the reference's keys (loadable by test_2D_fully.py:115-117), val.csv and log.txt appear."""
    from chap_amd.train_ours_2D import train
    snap = str(tmp_path / "run")
    model = train(dict(model="dualdecoder", decoder_type="mcnet", num_classes=4, batch_size=8, labeled_bs=4, image_size=[64, 64],
                       max_iterations=60, val_interval=30, base_lr=0.05, gpu=0, seed=7), snap)
    import os
    files = sorted(os.listdir(snap))
    assert "latest.pth" in files and "log.txt" in files
    assert ("dualdecoder_best_model.pth" in files) == ("val.csv" in files)       # written together, when the validation Dice improved (:431-449)
    ck = torch.load(os.path.join(snap, "latest.pth"), map_location="cpu")
    assert list(ck.keys()) == list(oinit.dual_decoder_2d_state(1).keys())
    fresh = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).eval()
    fresh.load_state_dict(ck, strict=True)
    x = torch.rand(2, 1, 64, 64, device=DEV)
    model.eval()
    with torch.no_grad():
        a, b = model(x), fresh(x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert "model1_mean_dice" in open(os.path.join(snap, "log.txt")).read()
