"""Iteration-level parity with conditioning evidence.

(1) The small-batch cases whose bounds had been widened to fit (K = 2, Dice distance, sign step, the 3D loop): the HIP fp32
    iteration is judged by its distance to the fp64 oracle relative to the fp32 oracle's own distance to fp64
    (tests/iteration_parity.py) -- no absolute rel-L2 0.15 / cosine 0.97 / worst < 0.5 bounds any more.
(2) The BASELINE sizes: one fp32 iteration with injected masks against oracle.train_step.iteration at config 0 (B = 8 = 4 + 4,
    256 x 256), config 1 (B = 24, 256 x 256, through the captured graph = the benched path) and config 3 (B = 4, 112 x 112 x 80,
    graph): losses, BatchNorm running statistics and the SGD update against the fp32 oracle, absolute bounds = 2 x the measured
    values; with CHAP_FULL_FP64=1 also the relative-to-fp64 criterion (that run's figures: profiles/r03_iteration_parity.jsonl).
    Reference lines: code/train_ours_2D.py:301-389.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import init as oinit
from oracle import train_step as ots
from tests.iteration_parity import assert_as_close_to_fp64_as_the_fp32_oracle, gm_over, inject_2d, inject_3d, three_way

SEEDS = int(os.environ.get("CHAP_CONDITIONING_SEEDS", "3"))        # realisations per small case (data, dropout masks, VAT noise); profiles/r03_iteration_parity.jsonl: 4 -- geometric means over them, tests/iteration_parity.py:gm_over


# ------------------------------------------------------------------------------------------------ (1) small, ill-conditioned
# default run (round 4): the default loop, K = 2, the Dice distance, the sign step, and 3D with K = 1 and K = 2 -- every variant that failed round 3's
# criterion on one code state is back in the default suite (the schedule experiments that used to take its time are gone);
# CHAP_CONDITIONING_ALL=1 adds the combination k2_dice_sign.
ALL = os.environ.get("CHAP_CONDITIONING_ALL") == "1"
VARIANTS = ["base", "k2", "dice", "sign", "k2_dice_sign"] if ALL else ["base", "k2", "dice", "sign"]


@pytest.mark.parametrize("variant", VARIANTS)
def test_small_2d_iteration_is_as_close_to_fp64_as_the_fp32_oracle(variant):
    B, lbs, H, W = 8, 4, 64, 64
    U = B - lbs
    K = 2 if "k2" in variant else 1
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=K, adv_losstype="dice" if "dice" in variant else "kl", vat_sign="sign" in variant)
    state = oinit.dual_decoder_2d_state(611)
    runs = []
    for s in range(SEEDS):
        vol, lab = ots.synthetic_batch(1441 + s, lbs, U, H, W)
        runs.append(three_way("2d_64_%s_s%d" % (variant, s), 2, state, vol, lab, (9 + s, 4 + 2 * s), 4500, args, inject_2d(U, lbs // 2 + U // 2, H, W, K, seed=50 * s)))
        assert runs[-1]["hip_o32"]["loss"] < 2e-4           # absolute: the losses are well conditioned whatever the variant
    assert_as_close_to_fp64_as_the_fp32_oracle(gm_over(runs))


@pytest.mark.parametrize("K", [1, 2])
def test_small_3d_iteration_is_as_close_to_fp64_as_the_fp32_oracle(K):
    B, lbs, sp = 4, 2, (16, 32, 16)
    U = B - lbs
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=K, num_classes=2)
    state = oinit.dual_decoder_3d_state(402)
    runs = []
    for s in range(SEEDS):
        vol, lab = ots.synthetic_batch_3d(1338 + s, lbs, U, *sp)
        runs.append(three_way("3d_16x32x16_k%d_s%d" % (K, s), 3, state, vol, lab, (2, 5 - s, 3), 4500, args, inject_3d(U, lbs // 2 + U // 2, sp, K, seed=50 * s)))
        assert runs[-1]["hip_o32"]["loss"] < 5e-4
    assert_as_close_to_fp64_as_the_fp32_oracle(gm_over(runs))


# ------------------------------------------------------------------------------------------------ (2) the BASELINE sizes
FULL = {  # name: (dims, B, spatial, box, graph replay, K = VAT power iterations); absolute bounds = 2 x measured (profiles/r03_iteration_parity.jsonl)
    "config0_2d_b8_256": (2, 8, (256, 256), (31, 57), False, 1),
    "config1_2d_b24_256": (2, 24, (256, 256), (31, 57), True, 1),
    "config3_3d_b4_112x112x80": (3, 4, (112, 112, 80), (11, 20, 9), True, 1),
    "config4_3d_b4_112x112x80_k2": (3, 4, (112, 112, 80), (11, 20, 9), True, 2),       # BASELINE config 4's per-GPU iteration: K = 2 power iterations
}
# 2 x the values measured on MI355X, HIP fp32 against the fp32 oracle (profiles/r03_iteration_parity.jsonl; the same file holds
# the fp64 legs: the fp32 oracle's own distance to fp64 is 0.0023 / 0.0019 / 0.0445 relative L2 of the update, the HIP path's
# 0.0030 / 0.0021 / 0.0354 -- at config 3 the HIP path is CLOSER to fp64 than the fp32 oracle is)
# (Round 4: the `loss` distance is rounding noise of the HIP loss sums (fp32 over 0.5 M pixels; the fp32 oracle is 4.5e-8 from fp64 there) and is re-rolled by
#  anything that changes the last bits of the logits -- e.g. another dealing of the BatchNorm statistics to partial slots when a conv's persistent grid changes:
#  config 0 moved from 2.2e-6 to 1.16e-5 when the fp32 conv instances were given one staged weight buffer, with every conv output unchanged bit for bit
#  (profiles/r04_wbuf1_bits.log).  The fp32 instances therefore keep their round-3 build, and these bounds their round-3 values.)
FULL_BOUNDS = {
    "config0_2d_b8_256": dict(loss=5e-6, vat=8.2e-5, upd_rel_l2=6.7e-3, one_minus_cos=6.9e-4, bn_stats=7.7e-7),           # measured 2.2e-6, 4.1e-5, 3.3e-3, 3.4e-4, 3.8e-7
    "config1_2d_b24_256": dict(loss=3.1e-6, vat=9.4e-5, upd_rel_l2=4.3e-3, one_minus_cos=4.3e-4, bn_stats=4.7e-7),         # measured 1.5e-6, 4.7e-5, 2.1e-3, 2.1e-4, 2.3e-7
    "config3_3d_b4_112x112x80": dict(loss=2e-6, vat=2.8e-4, upd_rel_l2=9.1e-2, one_minus_cos=5.4e-3, bn_stats=1.4e-6),     # measured 3.9e-7, 1.4e-4, 4.5e-2, 2.7e-3, 6.7e-7
    # K = 2 at this size is ILL-CONDITIONED (the second power iteration normalises a gradient of a gradient): the fp32 ORACLE is 0.303 relative L2 /
    # cos 0.76 from its own fp64 run, the HIP path 0.269 / 0.74 from fp64 and 0.289 / 0.78 from the fp32 oracle (profiles/r03_iteration_parity.jsonl,
    # CHAP_FULL_FP64=1 run) -- as close to the truth as the reference arithmetic is; the bounds below are 2 x those measured values
    "config4_3d_b4_112x112x80_k2": dict(loss=8e-7, vat=1.9e-2, upd_rel_l2=0.58, one_minus_cos=0.44, bn_stats=1.4e-6),        # measured 3.9e-7, 9.3e-3, 0.289, 0.220, 6.7e-7
}
FULL_FP64 = os.environ.get("CHAP_FULL_FP64") == "1"      # the fp64 oracle at these sizes takes 24 / 42 / 139 / 172 s of host time: evidence run, not the default


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_iteration_matches_oracle(name):
    dims, B, sp, box, graph, K = FULL[name]
    lbs = B // 2
    U = B - lbs
    if dims == 2:
        args = dict(labeled_bs=lbs, batch_size=B, vat_iters=K)
        state = oinit.dual_decoder_2d_state(1337)
        vol, lab = ots.synthetic_batch(1337, lbs, U, *sp)
        inj = inject_2d(U, lbs // 2 + U // 2, sp[0], sp[1], K, seed=100)
    else:
        args = dict(labeled_bs=lbs, batch_size=B, vat_iters=K, num_classes=2)
        state = oinit.dual_decoder_3d_state(1337)
        vol, lab = ots.synthetic_batch_3d(1337, lbs, U, *sp)
        inj = inject_3d(U, lbs // 2 + U // 2, sp, K, seed=100)
    res = three_way(name, dims, state, vol, lab, box, 4500, args, inj, graph=graph, fp64=FULL_FP64)
    if FULL_FP64:
        assert_as_close_to_fp64_as_the_fp32_oracle(res)
    b, h = FULL_BOUNDS[name], res["hip_o32"]
    assert h["loss"] < b["loss"] and h["vat"] < b["vat"] and h["bn_stats"] < b["bn_stats"], h
    assert h["upd_rel_l2"] < b["upd_rel_l2"] and 1.0 - h["cos_min"] < b["one_minus_cos"], h
