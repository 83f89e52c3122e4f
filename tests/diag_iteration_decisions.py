"""Where does the HIP fp32 iteration leave the fp32 oracle?  (VERDICT r3 "next round" 2a.)

For the small ill-conditioned 2D case of tests/test_iteration_conditioning_gpu.py (8 x 64 x 64, state 611, data seeds 1441 + s) this script
runs ONE iteration three ways -- HIP fp32, oracle fp32, oracle fp64 -- and records, per data seed,

  * every DISCRETE decision of the iteration (train_ours_2D.py:314-372) on all three sides and the number of elements on which two sides
    differ: the arg-max pseudo labels of the two heads (:318-319), the largest-CC filtered labels (:326-327), the perturbation mask
    (create_maskV1, :371), the max-pool routes of every pass that is differentiated (pass B, the VAT passes), and sign(d) of the VAT
    direction;
  * the two halves of the parameter gradient separately -- the BCP part (pass B, gradient bucket 0) and the VAT part (bucket 1) -- as relative
    L2 distances between the three sides, plus the cosine of the final VAT perturbation r_adv.

Output: gpurun_out/r04_seed_diagnosis.json (copied to profiles/).  Test infrastructure: imports oracle/ (runs on the GPU box, not part of the
pytest suite: `python tests/diag_iteration_decisions.py`)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from chap_amd import ops                                        # noqa: E402
from chap_amd.networks import DualDecoder                       # noqa: E402
from chap_amd.train import ChapStep                             # noqa: E402
from oracle import init as oinit                                # noqa: E402
from oracle import nets as onets                                # noqa: E402
from oracle import train_step as ots                            # noqa: E402
from tests.iteration_parity import inject_2d, to_dev            # noqa: E402

DEV = "cuda"


def run_oracle(state, vol, lab, box, it0, args, inj, dtype):
    """Oracle iteration with every discrete decision recorded.  Returns (decisions, grads_total, grads_bcp, r_adv)."""
    import torch.nn.functional as F
    rec = {"pool": {}}
    tag = {"cur": None}
    names = {id(v): k for k, v in inj.items() if k.startswith("drop")}
    orig = dict(pool=F.max_pool2d, pseudo=ots.pseudo_block, lcc=ots.largest_cc, mask=ots.create_mask_v1, vat=ots.vat2d)

    def pool(x, k):
        out, idx = orig["pool"](x, k, return_indices=True)
        W = x.shape[-1]
        route = ((idx // W) % 2) * 2 + (idx % W) % 2             # position inside the 2 x 2 window, row-major (the HIP kernel's code)
        rec["pool"].setdefault(tag["cur"], []).append(route.permute(0, 2, 3, 1).contiguous().to(torch.uint8))
        return out

    def net(sd, xx, train=False, drop=None, update_stats=True):
        tag["cur"] = names.get(id(drop), "?")
        return onets.dual_decoder_2d(sd, xx, train=train, drop=drop, update_stats=update_stats)

    def pseudo(p1, p2):
        r = orig["pseudo"](p1, p2)
        rec["pseudo1"], rec["pseudo2"] = r[2].clone(), r[3].clone()
        return r

    def lcc(seg, nc):
        r = orig["lcc"](seg, nc)
        rec.setdefault("lcc", []).append(r.clone())
        return r

    def mask(*a, **k):
        r = orig["mask"](*a, **k)
        rec["diff_mask"] = r.clone()
        return r

    def vat(*a, **k):
        loss, r = orig["vat"](*a, **k)
        rec["r_adv"] = r.detach().clone()
        return loss, r

    def once(a2):
        sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
        i2 = dict(inj)
        i2["d0"] = inj["d0"].to(dtype)
        return ots.iteration(sd, moms, vol.to(dtype), lab, box, iter_num=it0, lr=0.01, args=a2, inject=i2, net=net)

    F.max_pool2d, ots.pseudo_block, ots.largest_cc, ots.create_mask_v1, ots.vat2d = pool, pseudo, lcc, mask, vat
    try:
        full = once(args)
        keep = {k: ({t: list(l) for t, l in v.items()} if k == "pool" else (list(v) if isinstance(v, list) else v)) for k, v in rec.items()}
        bcp = once(dict(args, adv_noise=False))             # (records into `rec` again; `keep` holds the full run's decisions)
    finally:
        F.max_pool2d, ots.pseudo_block, ots.largest_cc, ots.create_mask_v1, ots.vat2d = (orig[k] for k in ("pool", "pseudo", "lcc", "mask", "vat"))
    return keep, {k: v.double() for k, v in full["grads"].items()}, {k: v.double() for k, v in bcp["grads"].items()}


def run_hip(state, vol, lab, box, it0, args, inj):
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m.load_state_dict(state, strict=True)
    step = ChapStep(m, args)
    step.iter_num = it0
    vd, ld, injd = vol.to(DEV), lab.to(DEV), to_dev(inj, 2)
    names = {id(v): k for k, v in injd.items() if k.startswith("drop")}
    rec = {"pool": {}}
    tag = {"cur": None}
    orig = dict(pool=ops.act_pool2, pseudo=ops.pseudo_block, lcc=ops.largest_cc, mask=ops.diff_mask, perturb=ops.perturb, fwd=m.forward)

    def pool(src, out, idx=None, dims=2):
        orig["pool"](src, out, idx, dims=dims)
        if idx is not None:
            rec["pool"].setdefault(tag["cur"], []).append(idx)

    def fwd(x, *a, **k):
        tag["cur"] = names.get(id(k.get("drop_masks")), "?")
        return orig["fwd"](x, *a, **k)

    def pseudo(*a, **k):
        r = orig["pseudo"](*a, **k)
        rec["pseudo1"], rec["pseudo2"] = r[2], r[3]
        return r

    def lcc(lab_, nc):
        r = orig["lcc"](lab_, nc)
        rec.setdefault("lcc", []).append(r)
        return r

    def mask(*a, **k):
        r = orig["mask"](*a, **k)
        rec["diff_mask"] = r
        return r

    def perturb(x, d, out, alpha, mask=None, sign=False):
        orig["perturb"](x, d, out, alpha, mask=mask, sign=sign)
        if mask is not None:
            rec["r_adv"] = (out - x).clone()

    ops.act_pool2, ops.pseudo_block, ops.largest_cc, ops.diff_mask, ops.perturb, m.forward = pool, pseudo, lcc, mask, perturb, fwd
    try:
        step._hw = tuple(vd.shape[2:])
        step.prepare(box)
        step.device_step(vd, ld, injd, update=False)
        torch.cuda.synchronize()
    finally:
        ops.act_pool2, ops.pseudo_block, ops.largest_cc, ops.diff_mask, ops.perturb = (orig[k] for k in ("pool", "pseudo", "lcc", "mask", "perturb"))
        m.forward = orig["fwd"]
    n = step.grad2.numel()
    g0 = {k: v.detach().double().cpu().clone() for k, v in m.grad_views_of(step.grad_both[:n]).items()}
    g1 = {k: v.detach().double().cpu().clone() for k, v in m.grad_views_of(step.grad2).items()}
    out = {"pool": {t: [i.squeeze(1).cpu() for i in lst] for t, lst in rec["pool"].items()}}
    for k in ("pseudo1", "pseudo2", "diff_mask", "r_adv"):
        out[k] = rec[k].cpu()
    out["lcc"] = [t.cpu() for t in rec["lcc"]]
    return out, g0, g1


def rel_l2(a, b, keys):
    num = sum(float(((a[k] - b[k]) ** 2).sum()) for k in keys)
    den = sum(float((b[k] ** 2).sum()) for k in keys)
    return (num / max(den, 1e-300)) ** 0.5


def decisions_diff(a, b):
    """Number of elements on which two sides decided differently, per decision."""
    out = {}
    for k in ("pseudo1", "pseudo2", "diff_mask"):
        out[k] = int((a[k].double().reshape(-1) != b[k].double().reshape(-1)).sum())
    out["lcc"] = [int((x != y).sum()) for x, y in zip(a["lcc"], b["lcc"])]
    out["sign_r_adv"] = int((torch.sign(a["r_adv"].double().reshape(-1)) != torch.sign(b["r_adv"].double().reshape(-1))).sum())
    ra, rb = a["r_adv"].double().reshape(-1), b["r_adv"].double().reshape(-1)
    out["one_minus_cos_r_adv"] = float(1.0 - (ra * rb).sum() / (ra.norm() * rb.norm() + 1e-300))
    pools = {}
    for t in sorted(a["pool"]):
        if t in b["pool"]:
            pools[t] = [int((x != y).sum()) for x, y in zip(a["pool"][t], b["pool"][t])]
    out["pool_routes"] = pools
    return out


def stage_errors(state, vol, inj, U, xi=10.0):
    """The first power-iteration pass (forward on x + xi d, distance to FIXED targets, backward to the input) stage by stage: relative L2 distance
    to the fp64 oracle of (a) the five encoder features, (b) the logits, (c) d(distance)/d(logits), (d) the input gradient, for the HIP fp32 path and
    for the fp32 oracle -- which stage makes the HIP path's VAT direction noisier than the CPU fp32 arithmetic, if any.  The soft targets are
    the fp64 oracle's on every side (they only set the scale of the cancellation q - p)."""
    x = vol[-U:].contiguous()
    d0 = inj["d0"]

    def oracle(dtype):
        sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}
        with torch.no_grad():
            t1, t2 = onets.dual_decoder_2d(sd, x.to(dtype), train=True, drop=inj["drop_A"], update_stats=False)
            soft = (torch.softmax(t1, 1), torch.softmax(t2, 1))
        d = ots.l2_normalize(d0.to(dtype))
        xh = (x.to(dtype) + xi * d).requires_grad_(True)
        l1, l2, feats = onets.dual_decoder_2d(sd, xh, train=True, drop=inj["drop_V0"], update_stats=False, with_feat=True)
        return sd, soft, xh, (l1, l2), feats

    sd64, soft64, xh64, lg64, f64 = oracle(torch.float64)
    out = {}
    ref = None
    for tag, dtype in (("o64", torch.float64), ("o32", torch.float32)):
        sd, _, xh, (l1, l2), feats = (sd64, soft64, xh64, lg64, f64) if tag == "o64" else oracle(dtype)
        soft = tuple(t.to(dtype) for t in soft64)
        dist = ots.kl_two_heads((l1, l2), soft)
        g1, g2, dx = torch.autograd.grad(dist, [l1, l2, xh])
        out[tag] = dict(feats=[f.detach().double() for f in feats], logits=[l1.detach().double(), l2.detach().double()], dlogits=[g1.double(), g2.double()], dx=dx.double())
    # HIP
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    m.load_state_dict(state, strict=True)
    injd = to_dev(inj, 2)
    xd = x.to(DEV)
    d = torch.empty_like(xd)
    ops.l2_normalize(injd["d0"], d)
    xh = torch.empty_like(xd).requires_grad_(True)
    ops.perturb(xd, d, xh, xi)
    with m.frozen():
        l1, l2, feats = m(xh, True, update_stats=False, drop_masks=injd["drop_V0"])
    g1, g2 = torch.empty_like(l1), torch.empty_like(l2)
    ops.kl_fwd_bwd((l1, l2), tuple(t.float().to(DEV) for t in soft64), None, (g1, g2))
    dx = m.backward_saved(l1, [g1, g2], need_wgrad=False, need_dx=True)
    torch.cuda.synchronize()
    out["hip"] = dict(feats=[f.detach().double().cpu() for f in feats], logits=[l1.detach().double().cpu(), l2.detach().double().cpu()],
                      dlogits=[g1.double().cpu(), g2.double().cpu()], dx=dx.double().cpu())

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-300))

    res = {}
    for side in ("hip", "o32"):
        a, b = out[side], out["o64"]
        r = {"feat%d" % i: rel(x_, y_) for i, (x_, y_) in enumerate(zip(a["feats"], b["feats"]))}
        r.update(logits1=rel(a["logits"][0], b["logits"][0]), logits2=rel(a["logits"][1], b["logits"][1]),
                 dlogits1=rel(a["dlogits"][0], b["dlogits"][0]), dlogits2=rel(a["dlogits"][1], b["dlogits"][1]), dx=rel(a["dx"], b["dx"]))
        # the same backward arithmetic from EXACT dlogits would need another pass; the ratio dx / dlogits shows what the backward adds
        res[side] = r
    return res


def main():
    B, lbs, H, W = 8, 4, 64, 64
    U = B - lbs
    variant = os.environ.get("CHAP_DIAG_VARIANT", "base")
    K = 2 if "k2" in variant else 1
    args = dict(labeled_bs=lbs, batch_size=B, vat_iters=K, adv_losstype="dice" if "dice" in variant else "kl", vat_sign="sign" in variant)
    state = oinit.dual_decoder_2d_state(611)
    results = []
    for s in range(int(os.environ.get("CHAP_DIAG_SEEDS", "4"))):
        vol, lab = ots.synthetic_batch(1441 + s, lbs, U, H, W)
        box = (9 + s, 4 + 2 * s)
        inj = inject_2d(U, lbs // 2 + U // 2, H, W, K, seed=50 * s)
        hip, hg0, hg1 = run_hip(state, vol, lab, box, 4500, args, inj)
        o32, g32, b32 = run_oracle(state, vol, lab, box, 4500, args, inj, torch.float32)
        o64, g64, b64 = run_oracle(state, vol, lab, box, 4500, args, inj, torch.float64)
        keys = [k for k in g64 if k in hg0]
        v32 = {k: g32[k] - b32[k] for k in keys}                 # cw * VAT gradient = total - BCP-only
        v64 = {k: g64[k] - b64[k] for k in keys}
        res = {"case": "2d_64_%s_s%d" % (variant, s), "elements": {"pseudo": int(o64["pseudo1"].numel()), "r_adv": int(o64["r_adv"].numel())},
               "decisions_hip_vs_o32": decisions_diff(hip, o32), "decisions_hip_vs_o64": decisions_diff(hip, o64),
               "decisions_o32_vs_o64": decisions_diff(o32, o64),
               "grad_bcp_rel_l2": {"hip_o32": rel_l2(hg0, b32, keys), "hip_o64": rel_l2(hg0, b64, keys), "o32_o64": rel_l2(b32, b64, keys)},
               "grad_vat_rel_l2": {"hip_o32": rel_l2(hg1, v32, keys), "hip_o64": rel_l2(hg1, v64, keys), "o32_o64": rel_l2(v32, v64, keys)},
               "grad_norms": {"bcp": sum(float((b64[k] ** 2).sum()) for k in keys) ** 0.5, "vat": sum(float((v64[k] ** 2).sum()) for k in keys) ** 0.5}}
        # the parameter tensors that carry the HIP path's distance: top 5 by squared error of the total gradient against fp64
        tot_h = {k: hg0[k] + hg1[k] for k in keys}
        err = sorted(((float(((tot_h[k] - g64[k]) ** 2).sum()), k) for k in keys), reverse=True)[:5]
        den = sum(float((g64[k] ** 2).sum()) for k in keys)
        res["top_error_tensors_hip_o64"] = [{"key": k, "share_of_sq_error": e / max(sum(x for x, _ in err), 1e-300), "rel_to_total_norm": (e / den) ** 0.5} for e, k in err]
        err32 = sorted(((float(((g32[k] - g64[k]) ** 2).sum()), k) for k in keys), reverse=True)[:5]
        res["top_error_tensors_o32_o64"] = [{"key": k, "rel_to_total_norm": (e / den) ** 0.5} for e, k in err32]
        if variant == "base":
            res["v0_pass_stage_errors_vs_fp64"] = stage_errors(state, vol, inj, U)
        print(json.dumps(res))
        results.append(res)
    path = os.path.join(ROOT, "gpurun_out", "r04_seed_diagnosis_%s.json" % variant)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
