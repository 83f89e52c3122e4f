"""Multi-iteration training parity: the same K iterations (identical injected dropout masks, VAT noise and
BCP boxes) on the GPU (chap_amd) and on the CPU oracle, then Dice of the ensemble prediction on held-out
synthetic slices (north_star: |dDice| <= 1e-3 in fp32 mode).  Also the reference's inference recipe
(test_2D_fully.py:54-95: per slice -> net -> (o1+o2)/2 -> softmax -> argmax) on both sides."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from chap_amd.networks import DualDecoder
from chap_amd.train import ChapStep
from oracle import init as oinit
from oracle import nets as onets
from oracle import train_step as ots

DEV = "cuda"


def dice_per_class(pred, gt, n_classes=4):
    out = []
    for c in range(1, n_classes):
        p, g = (pred == c), (gt == c)
        den = p.sum() + g.sum()
        out.append(2.0 * (p & g).sum() / den if den > 0 else 1.0)
    return np.array(out, dtype=np.float64)


def cl_masks(masks):
    return {k: v.permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(DEV) for k, v in masks.items()}


def predict_oracle(sd, vol):
    with torch.no_grad():
        o1, o2 = onets.dual_decoder_2d(sd, vol, train=False)
        return torch.argmax(torch.softmax((o1 + o2) / 2.0, dim=1), dim=1).numpy()


def predict_hip(m, vol):
    m.eval()
    preds = []
    with torch.no_grad():
        for i in range(vol.shape[0]):                       # slice by slice, batch 1, like test_single_volume
            o = m(vol[i:i + 1].to(DEV))
            preds.append(torch.argmax(torch.softmax((o[0] + o[1]) / 2.0, dim=1), dim=1).cpu())
    m.train()
    return torch.cat(preds).numpy()


def soft_pred_oracle(sd, vol):
    with torch.no_grad():
        o1, o2 = onets.dual_decoder_2d(sd, vol, train=False)
        return torch.softmax((o1 + o2) / 2.0, dim=1)


@pytest.mark.parametrize("dtype,tol_mean,tol_max", [(torch.float32, 1e-2, 1.5e-1), (torch.bfloat16, 6e-2, 8e-1)])
def test_k_iterations_match_oracle(dtype, tol_mean, tol_max):
    """K full iterations with identical injected randomness: the ensemble probabilities of the two trained
    models (eval mode) agree; the loss trajectories agree.  Tolerances: the iteration is discontinuous in the
    weights (arg-max pseudo labels, largest-CC filter) and the 64x64 fixture has tiny-batch BatchNorm, so two
    fp32 implementations with different summation orders drift by ~1e-3 per iteration.  Bounds = 2.5x the measured deviations
    (round 2, deterministic reductions: fp32 mean 4.2e-3 / max 5.9e-2 of a probability, bf16 2.7e-2 / 0.41 -- single pixels flip class;
    what bf16 training is worth is gated by the Dice test in test_parity_gates_gpu.py)."""
    K, B, lbs, H, W = 5, 8, 4, 64, 64
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1, base_lr=0.01)
    state = oinit.dual_decoder_2d_state(777)
    sd = {k: v.clone() for k, v in state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train().set_compute_dtype(dtype)
    m.load_state_dict(state, strict=True)
    step = ChapStep(m, args)
    lr = 0.01
    for it in range(K):
        vol, lab = ots.synthetic_batch(1000 + it, lbs, U, H, W)
        inj = {"drop_A": oinit.drop_masks_2d(10 * it + 1, U, H, W), "drop_B": oinit.drop_masks_2d(10 * it + 2, lbs // 2 + U // 2, H, W),
               "drop_V0": oinit.drop_masks_2d(10 * it + 3, U, H, W), "drop_VF": oinit.drop_masks_2d(10 * it + 4, U, H, W),
               "d0": torch.rand(U, 1, H, W, generator=torch.Generator().manual_seed(10 * it + 5)) - 0.5}
        box = (3 + it, 5 + 2 * it)
        ref = ots.iteration(sd, moms, vol, lab, box, iter_num=it, lr=lr, args=args, inject=inj)
        inj_d = {k: (cl_masks(v) if k.startswith("drop") else v.to(DEV)) for k, v in inj.items()}
        out = step.step(vol.to(DEV), lab.to(DEV), box_yx=box, inject=inj_d)
        bcp = sum(float(l[2]) for l in out["mix_losses"])
        assert abs(bcp - float(ref["bcp_loss"])) / float(ref["bcp_loss"]) < (2e-2 if dtype == torch.float32 else 1e-1), (it, bcp, float(ref["bcp_loss"]))
        lr = ots.poly_lr(0.01, it + 1, 30000)
        assert abs(step.opt.param_groups[0]["lr"] - lr) < 1e-12
    val, _ = ots.synthetic_batch(4242, 4, 0, H, W)
    p_ref = soft_pred_oracle({k: v.detach() for k, v in sd.items()}, val)
    m.eval()
    with torch.no_grad():
        o = m(val.to(DEV))
        p_hip = torch.softmax((o[0] + o[1]) / 2.0, dim=1).cpu()
    diff = (p_ref - p_hip).abs()
    assert diff.mean().item() < tol_mean and diff.max().item() < tol_max, (diff.mean().item(), diff.max().item())


def test_trained_model_dice_matches_oracle_inference():
    """Train on the GPU until predictions are non-trivial, move the checkpoint to the CPU oracle
    (state_dict interchange) and run the reference's inference recipe on both: Dice within 1e-3."""
    torch.manual_seed(1337)
    np.random.seed(1337)                                     # BCP box offsets (train_ours_2D.py:97-98 draws them from numpy)
    B, lbs, H, W = 8, 4, 64, 64
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    step = ChapStep(m, dict(labeled_bs=lbs, batch_size=B, base_lr=0.05, adv_noise=True))
    pool = [ots.synthetic_batch(2000 + i, lbs, B - lbs, H, W) for i in range(8)]
    pool = [(v.to(DEV), l.to(DEV)) for v, l in pool]
    first = last = None
    for it in range(240):
        v, l = pool[it % len(pool)]
        out = step.step(v, l)
        if it % 40 == 0 or it == 239:
            tot = sum(float(x[2]) for x in out["mix_losses"])
            first = tot if first is None else first
            last = tot
    assert last < 0.6 * first, (first, last)
    val, gt = ots.synthetic_batch(4242, 6, 0, H, W)
    p_hip = predict_hip(m, val)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    p_ref = predict_oracle(sd, val)
    d_ref, d_hip = dice_per_class(p_ref, gt.numpy()), dice_per_class(p_hip, gt.numpy())
    assert d_ref.mean() > 0.2, d_ref                        # the trained model really segments something (the trajectory is chaotic:
                                                            # rounding x arg-max pseudo labels; 0.28 .. 0.45 over seeds / arithmetic)
    assert np.abs(d_ref - d_hip).max() <= 1e-3, (d_ref, d_hip)
    assert (p_ref == p_hip).mean() > 0.999


def test_resume_continues_the_run():
    """Checkpoint / resume (build extension): 2 iterations, state_dict(), a FRESH model + ChapStep loaded from it,
    2 more iterations == 4 uninterrupted iterations (same injected randomness; BCP boxes come from numpy's RNG,
    which the checkpoint carries).  Bitwise: every reduction on the device has a fixed order (round 2)."""
    B, lbs, H, W = 8, 4, 64, 64
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1, base_lr=0.01)
    state = oinit.dual_decoder_2d_state(321)

    def inj(it):
        d = {"drop_A": oinit.drop_masks_2d(10 * it + 1, U, H, W), "drop_B": oinit.drop_masks_2d(10 * it + 2, lbs // 2 + U // 2, H, W),
             "drop_V0": oinit.drop_masks_2d(10 * it + 3, U, H, W), "drop_VF": oinit.drop_masks_2d(10 * it + 4, U, H, W),
             "d0": torch.rand(U, 1, H, W, generator=torch.Generator().manual_seed(10 * it + 5)) - 0.5}
        return {k: (cl_masks(v) if k.startswith("drop") else v.to(DEV)) for k, v in d.items()}

    def fresh():
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
        m.load_state_dict(state, strict=True)
        return m, ChapStep(m, args)

    def run(step, its):
        for it in its:
            vol, lab = ots.synthetic_batch(2000 + it, lbs, U, H, W)
            step.step(vol.to(DEV), lab.to(DEV), inject=inj(it))       # box offsets: np.random (train_ours_2D.py:97-98)

    def dist(ma, mb):
        sa, sb = ma.state_dict(), mb.state_dict()
        return max(((sa[k].float() - sb[k].float()).abs().max() / sa[k].float().abs().max().clamp_min(1.0)).item() for k in sa)

    np.random.seed(99)
    m_a, s_a = fresh()
    run(s_a, range(4))
    np.random.seed(99)
    m_a2, s_a2 = fresh()                                               # the same run again: bit-identical (no float atomics, round 2)
    run(s_a2, range(4))
    np.random.seed(99)
    m_b, s_b = fresh()
    run(s_b, range(2))
    ckpt = s_b.state_dict()
    assert set(ckpt["model"]) == set(state) and ckpt["iter_num"] == 2
    np.random.seed(12345)                                              # the checkpoint must restore the box RNG
    m_c = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    s_c = ChapStep(m_c, args).load_state_dict(ckpt)
    assert s_c.iter_num == 2 and abs(s_c.opt.param_groups[0]["lr"] - ots.poly_lr(0.01, 2, 30000)) < 1e-12
    assert torch.equal(s_c.opt.mom, s_b.opt.mom) and all(torch.equal(v, m_c.state_dict()[k]) for k, v in m_b.state_dict().items())
    run(s_c, range(2, 4))
    noise, resumed = dist(m_a, m_a2), dist(m_a, m_c)
    assert noise == 0.0 and resumed == 0.0, (noise, resumed)           # a resumed run continues BIT FOR BIT (parameters, momentum, BN buffers, RNG epochs, box RNG)
    wrong = dist(m_a, m_b)                                             # a run that stopped after 2 iterations is far away
    assert wrong > 3 * resumed, (wrong, resumed)
