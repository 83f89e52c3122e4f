"""Run-to-run spread of the update-relative error that test_ablation_iteration_matches_oracle bounds (float atomics in
the BN statistics, amplified by small-batch BN): prints the worst tensor of several repetitions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import init as oinit
from oracle import train_step as ots
from chap_amd.networks import DualDecoder
from chap_amd.train import AblationStep

DEV = "cuda"
cl = lambda masks: {k: v.permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(DEV) for k, v in masks.items()}
B, lbs, H, W = 8, 4, 64, 64
U = B - lbs
args = dict(labeled_bs=lbs, batch_size=B, vat_iters=1, w_adv=0.7)
state = oinit.dual_decoder_2d_state(404)
vol, lab = ots.synthetic_batch(4321, lbs, U, H, W)
inj_cpu = {"drop_F": oinit.drop_masks_2d(11, B, H, W), "drop_V0": oinit.drop_masks_2d(13, U, H, W), "drop_VF": oinit.drop_masks_2d(14, U, H, W),
           "d0": torch.rand(U, 1, H, W, generator=torch.Generator().manual_seed(15)) - 0.5}
sd = {k: v.clone() for k, v in state.items()}
for k, v in sd.items():
    if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
        v.requires_grad_(True)
moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
ots.ablation_iteration(sd, moms, vol, lab, iter_num=3000, lr=0.01, args=args, inject=inj_cpu)
def poison():
    """Fill the caching allocator's free blocks with NaN so that a read of uninitialised memory shows up."""
    junk = [torch.full((1 << k,), float("nan"), device=DEV) for k in range(7, 26) for _ in range(3)]
    torch.cuda.synchronize()
    del junk


for rep in range(8):
    if os.environ.get("CHAP_POISON") == "1":
        poison()
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m.load_state_dict(state, strict=True)
    step = AblationStep(m, args)
    step.iter_num = 3000
    inj = {k: (cl(v) if k.startswith("drop") else v.to(DEV)) for k, v in inj_cpu.items()}
    step.step(vol.to(DEV), lab.to(DEV), inject=inj)
    torch.cuda.synchronize()
    after = m.state_dict()
    res = []
    for k, v in sd.items():
        if not v.is_floating_point():
            continue
        d = (after[k].cpu().double() - v.detach().double()).abs().max().item()
        upd = (v.detach().double() - state[k].double()).abs().max().item()
        floor = 3e-7 * v.detach().abs().max().item()
        if upd > 0:
            res.append((max(d - floor, 0.0) / upd, k))
    res.sort(reverse=True)
    num = den = 0.0
    cos_min = 1.0
    for k, v in sd.items():
        if not v.is_floating_point() or k.endswith(("running_mean", "running_var")):
            continue
        ug = (after[k].cpu().double() - state[k].double()).flatten()
        uo = (v.detach().double() - state[k].double()).flatten()
        num += float(((ug - uo) ** 2).sum()); den += float((uo ** 2).sum())
        if float(uo.norm()) > 0:
            cos_min = min(cos_min, float((ug * uo).sum() / (ug.norm() * uo.norm() + 1e-300)))
    print("   global rel-L2 of the update %.4f, min per-tensor cosine %.5f" % ((num / den) ** 0.5, cos_min), flush=True)
    print(rep, " ".join("%.4f:%s" % (e, k.replace("conv_conv.", "").replace("decoder", "d")) for e, k in res[:3]), flush=True)
