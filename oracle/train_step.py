"""CPU restatement of one iteration of code/train_ours_2D.py:301-389 (TEST INFRASTRUCTURE ONLY).

Present in the reference and restated from the cited lines (formula-level pin, the script itself
cannot be imported -- absent in-repo modules):
    mix_loss                 train_ours_2D.py:198-216
    generate_mask            train_ours_2D.py:91-101   (box offsets are an INPUT here)
    get_ACDC_masks / LCC     train_ours_2D.py:103-108, 123-144  (scipy.ndimage.label with a full 3x3
                             structure == skimage.measure.label default connectivity)
    pseudo-label block       train_ours_2D.py:314-325
    BCP mixing               train_ours_2D.py:331-338
    SGD + poly LR            train_ours_2D.py:278, 381-389
ABSENT from the reference ("parity unpinned"; defined by this build):
    dice_loss_bcp   (losses.DiceLoss_bcp)   masked multi-class Dice, smooth 1e-10
    vat2d           (losses.VAT2d)          VAT power iteration on the unlabeled half; distance 'kl' or 'dice' (--adv_losstype)
    create_mask_v1  (patch.create_maskV1)   disagreement OR top-k of the 4x-pooled knowledge map
    sigmoid_rampup  (ramps.sigmoid_rampup)  exp(-5 (1 - t)^2)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DICE_SMOOTH = 1e-10


def sigmoid_rampup(current, rampup_length):
    """Laine & Aila ramp-up (cited at train_ours_2D.py:35)."""
    if rampup_length == 0:
        return 1.0
    t = float(np.clip(current, 0.0, rampup_length)) / rampup_length
    return float(math.exp(-5.0 * (1.0 - t) ** 2))


def consistency_weight(iter_num, consistency=1.0, rampup=50.0):
    return consistency * sigmoid_rampup(iter_num // 150, rampup)     # train_ours_2D.py:34-36,356


def poly_lr(base_lr, iter_num, max_iterations):
    return base_lr * (1.0 - iter_num / max_iterations) ** 0.9        # train_ours_2D.py:387


def dice_loss_bcp(soft, target, mask, n_classes):
    """soft [N,C,...], target [N,...] int, mask [N,...] -> mean_c 1 - (2 I_c + s)/(Z_c + Y_c + s)."""
    loss = 0.0
    m = mask.to(soft.dtype)
    for c in range(n_classes):
        t = (target == c).to(soft.dtype)
        p = soft[:, c]
        inter = (p * t * m).sum()
        z = (p * p * m).sum()
        y = (t * t * m).sum()
        loss = loss + (1.0 - (2.0 * inter + DICE_SMOOTH) / (z + y + DICE_SMOOTH))
    return loss / n_classes


def mix_loss(output, img_l, patch_l, mask, l_weight=1.0, u_weight=0.5, unlab=False):
    """-> (loss_image, loss_patch, total) exactly as train_ours_2D.py:198-216."""
    n_classes = output.shape[1]
    img_l, patch_l = img_l.long(), patch_l.long()
    soft = F.softmax(output, dim=1)
    iw, pw = (u_weight, l_weight) if unlab else (l_weight, u_weight)
    m = mask.to(output.dtype)
    pm = 1.0 - m
    d1 = dice_loss_bcp(soft, img_l, m, n_classes) * iw
    d2 = dice_loss_bcp(soft, patch_l, pm, n_classes) * pw
    ce1 = iw * (F.cross_entropy(output, img_l, reduction="none") * m).sum() / (m.sum() + 1e-16)
    ce2 = pw * (F.cross_entropy(output, patch_l, reduction="none") * pm).sum() / (pm.sum() + 1e-16)
    return (d1 + ce1) / 2.0, (d2 + ce2) / 2.0, (d1 + d2 + ce1 + ce2) / 2.0


def box_masks(n, h, w, y0, x0, dtype=torch.float32):
    """generate_mask with explicit offsets: zero box of int(2H/3) x int(2W/3) at (y0, x0)."""
    ph, pw = int(h * 2 / 3), int(w * 2 / 3)
    mask = torch.ones(h, w, dtype=dtype)
    mask[y0:y0 + ph, x0:x0 + pw] = 0
    return mask, mask.unsqueeze(0).repeat(n, 1, 1)


def box_masks_3d(n, d, h, w, z0, y0, x0, dtype=torch.float32):
    """3D restatement of generate_mask (the reference has no 3D loop): zero cuboid of 2/3 of each side."""
    pd, ph, pw = int(d * 2 / 3), int(h * 2 / 3), int(w * 2 / 3)
    mask = torch.ones(d, h, w, dtype=dtype)
    mask[z0:z0 + pd, y0:y0 + ph, x0:x0 + pw] = 0
    return mask, mask.unsqueeze(0).repeat(n, 1, 1, 1)


def largest_cc(seg, n_classes):
    """seg int [N,H,W] / [N,D,H,W] -> keep, per sample and class 1..C-1, the largest component
    (full connectivity: 8 in 2D, 26 in 3D -- skimage.measure.label's default)."""
    from scipy import ndimage
    seg = seg.cpu().numpy()
    out = np.zeros_like(seg)
    full = np.ones((3,) * (seg.ndim - 1), dtype=bool)
    for i in range(seg.shape[0]):
        for c in range(1, n_classes):
            lab, n = ndimage.label(seg[i] == c, structure=full)
            if n == 0:
                continue
            best = np.argmax(np.bincount(lab.flat)[1:]) + 1
            out[i][lab == best] = c
    return torch.from_numpy(out)


def pseudo_block(pre1, pre2):
    soft1, soft2 = F.softmax(pre1, 1), F.softmax(pre2, 1)
    arg1, arg2 = soft1.argmax(1), soft2.argmax(1)
    know = F.cross_entropy(pre1, arg2, reduction="none") + F.cross_entropy(pre2, arg1, reduction="none")
    return soft1, soft2, arg1, arg2, know


def create_mask_v1(p1, p2, knowledge, scale_factor=4, topk=0.1):
    """(p1 != p2) OR nearest-upsample(top-k fraction of avg_pool(knowledge, scale)) -> float [N,H,W].
    Volumes [N,D,H,W] are treated as [N, D*H, W] (in-plane scale x scale patches, top-k per sample)."""
    if knowledge.dim() == 4:
        n, d, h, w = knowledge.shape
        return create_mask_v1(p1.reshape(n, d * h, w), p2.reshape(n, d * h, w), knowledge.reshape(n, d * h, w), scale_factor, topk).reshape(n, d, h, w)
    n, h, w = knowledge.shape
    pooled = F.avg_pool2d(knowledge.unsqueeze(1), scale_factor).squeeze(1).clamp_min(0)
    m = pooled[0].numel()
    k = max(int(topk * m), 1)
    thr = pooled.reshape(n, -1).topk(k, dim=1).values[:, -1]
    sel = (pooled >= thr.view(n, 1, 1)).float()
    sel = F.interpolate(sel.unsqueeze(1), scale_factor=scale_factor, mode="nearest").squeeze(1)
    return ((sel > 0) | (p1 != p2)).float()


def l2_normalize(d, eps=1e-8):
    n = d.reshape(d.shape[0], -1).norm(dim=1).view(-1, *([1] * (d.dim() - 1)))
    return d / (n + eps)


def kl_two_heads(logits, targets):
    """sum over heads of mean_{n,pixel} KL(target || softmax(logits))."""
    tot = 0.0
    for lg, t in zip(logits, targets):
        logp = F.log_softmax(lg, 1)
        kl = torch.where(t > 0, t * (torch.log(t.clamp_min(1e-38)) - logp), torch.zeros_like(t)).sum(1)
        tot = tot + kl.mean()
    return tot


def dice_two_heads(logits, targets, smooth=1e-10):
    """adv_losstype='dice' (--adv_losstype, train_ours_2D.py:515; losses.VAT2d is ABSENT, PARITY UNPINNED): the soft Dice
    of the VAT implementations of this script family (softDiceLoss: per class 1 - (2 sum p t + s)/(sum p^2 + sum t^2 + s)
    with the sums over batch and pixels, mean over classes), summed over the two heads."""
    tot = 0.0
    for lg, t in zip(logits, targets):
        p = F.softmax(lg, 1)
        red = [0] + list(range(2, p.dim()))
        inter, z, y = (p * t).sum(red), (p * p).sum(red), (t * t).sum(red)
        tot = tot + (1.0 - (2.0 * inter + smooth) / (z + y + smooth)).mean()
    return tot


DISTANCES = {"kl": kl_two_heads, "dice": dice_two_heads}


def vat2d(model_fn, x, soft1, soft2, mask, d0, xi=10.0, eps=6.0, k=1, sign=False, losstype="kl"):
    """model_fn(x) -> (logits1, logits2) in train mode WITHOUT running-stat updates.
    d0: initial noise in [-0.5, 0.5) (injected). mask [N,1,...] or None. Returns (loss, r_adv)."""
    dist_fn = DISTANCES[losstype]
    d = l2_normalize(d0)
    for _ in range(k):
        d = d.detach().requires_grad_(True)
        dist = dist_fn(model_fn(x + xi * d), (soft1, soft2))
        g, = torch.autograd.grad(dist, d)
        d = l2_normalize(g)
    d = d.detach()
    if sign:
        r = eps / math.sqrt(d[0].numel()) * torch.sign(d)
    else:
        r = eps * d
    if mask is not None:
        r = r * mask
    return dist_fn(model_fn(x + r), (soft1, soft2)), r


GRADSIM_KEYS_2D = ["encoder.in_conv.conv_conv.4.weight"] + ["encoder.down%d.maxpool_conv.1.conv_conv.4.weight" % i for i in range(1, 5)]


def grad_sim_scores(loss_l, loss_u, sd, keys=GRADSIM_KEYS_2D):
    """grad.GradSim.get_grad_convkernel (ABSENT upstream, call site train_ours_2D.py:365; PARITY UNPINNED -- the build's
    definition): per output channel of the conv kernel that produces each encoder feature, the cosine similarity
    of d loss_l / dW and d loss_u / dW."""
    ws = [sd[k] for k in keys]
    gl = torch.autograd.grad(loss_l, ws, retain_graph=True)
    gu = torch.autograd.grad(loss_u, ws, retain_graph=True)
    out = []
    for a, b in zip(gl, gu):
        a, b = a.reshape(a.shape[0], -1).double(), b.reshape(b.shape[0], -1).double()
        out.append(((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1) + 1e-12)).float())
    return out


def sgd_step(params, grads, moms, lr, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD(momentum, weight_decay), dampening 0, no nesterov; moms start at zero."""
    with torch.no_grad():
        for p, g, m in zip(params, grads, moms):
            gg = g + weight_decay * p
            m.mul_(momentum).add_(gg)
            p.sub_(lr * m)


# --------------------------------------------------------------------------- the whole iteration
ORACLE_ARGS = dict(base_lr=0.01, labeled_bs=12, max_iterations=30000, num_classes=4, consistency=1.0,
                   consistency_rampup=50.0, noise_mag=10.0, epi=6.0, topk1=0.1, adv_noise=True, vat_iters=1,
                   vat_sign=False, adv_losstype="kl", nms=1, momentum=0.9, weight_decay=1e-4)


def iteration(sd, moms, volume_batch, label_batch, box_yx, iter_num, lr, args=None, inject=None, net=None):
    """One iteration of train() (train_ours_2D.py:301-389) on CPU with the functional oracle nets.

    sd: state dict whose float parameters require grad (updated in place); moms: name -> momentum
    buffer; inject: {'drop_A','drop_B','drop_V0'..,'drop_VF': site->keep mask dicts, 'd0': VAT noise}.
    Returns dict(losses=[4 x (loss_image, loss_patch, total)], vat_loss, bcp_loss, loss)."""
    from . import nets
    a = dict(ORACLE_ARGS)
    a.update(args or {})
    inject = inject or {}
    net = net or nets.dual_decoder_2d
    nc, lbs = a["num_classes"], a["labeled_bs"]
    B = volume_batch.shape[0]
    lsub, usub = lbs // 2, (B - lbs) // 2
    img_a, img_b = volume_batch[:lsub], volume_batch[lsub:lbs]
    uimg_a, uimg_b = volume_batch[lbs:lbs + usub], volume_batch[lbs + usub:]
    lab_a, lab_b = label_batch[:lsub], label_batch[lsub:lbs]
    uimg_ab = volume_batch[lbs:]
    sp = tuple(volume_batch.shape[2:])
    with torch.no_grad():
        pre1, pre2 = net(sd, uimg_ab, train=True, drop=inject.get("drop_A"))
        soft1, soft2, arg1, arg2, know = pseudo_block(pre1, pre2)
        plab1 = largest_cc(arg1, nc) if a["nms"] else arg1
        plab2 = largest_cc(arg2, nc) if a["nms"] else arg2
        img_mask, loss_mask = (box_masks(lsub, *sp, *box_yx) if len(sp) == 2 else box_masks_3d(lsub, *sp, *box_yx))
        img_mask, loss_mask = img_mask.to(volume_batch.device), loss_mask.to(volume_batch.device)
        net_input_unl = uimg_a * img_mask + img_a * (1 - img_mask)
        net_input_l = img_b * img_mask + uimg_b * (1 - img_mask)
        net_input_mix = torch.cat((net_input_l, net_input_unl))
    out1, out2 = net(sd, net_input_mix, train=True, drop=inject.get("drop_B"))
    out_l1, out_unl1 = out1[:lsub], out1[lsub:]
    out_l2, out_unl2 = out2[:lsub], out2[lsub:]
    m1 = mix_loss(out_unl1, plab2[:usub], lab_a, loss_mask, u_weight=0.5, unlab=True)
    m2 = mix_loss(out_unl2, plab1[:usub], lab_a, loss_mask, u_weight=0.5, unlab=True)
    m3 = mix_loss(out_l1, lab_b, plab2[usub:], loss_mask, u_weight=0.5)
    m4 = mix_loss(out_l2, lab_b, plab1[usub:], loss_mask, u_weight=0.5)
    bcp_loss = m1[2] + m2[2] + m3[2] + m4[2]
    cw = consistency_weight(iter_num, a["consistency"], a["consistency_rampup"])
    if a["adv_noise"]:
        diff = create_mask_v1(arg1, arg2, know, 4, a["topk1"]).unsqueeze(1)
        calls = {"k": 0}

        def model_fn(xx):
            k = calls["k"]
            calls["k"] += 1
            key = "drop_V%d" % k if k < a["vat_iters"] else "drop_VF"
            return net(sd, xx, train=True, drop=inject.get(key), update_stats=False)

        d0 = inject["d0"] if inject.get("d0") is not None else torch.rand(uimg_ab.shape, device=uimg_ab.device) - 0.5
        vat_loss, _ = vat2d(model_fn, uimg_ab, soft1, soft2, diff, d0, a["noise_mag"], a["epi"], a["vat_iters"], a["vat_sign"], a["adv_losstype"])
    else:
        vat_loss = torch.zeros((), device=volume_batch.device)
    fp_loss = torch.zeros((), device=volume_batch.device)
    fp_terms = None
    if a.get("dropout"):
        # "2) fp" (train_ours_2D.py:359-365): both decoders on cat(features, channel-perturbed features of the second half).
        # Upstream compares these 1.5 U logits with the U pseudo labels (a shape error) and takes the scores from the absent
        # grad.GradSim; here every output row is paired with its own sample's pseudo label -- cat(pseudo, pseudo[U/2:]) --
        # and the scores are either injected (`sim_score`; all-zero = the Dropout2d pair of FilterDropout.py:71-73) or the running
        # GradSim state (`gradsim`, restated below from the call sites alone).  PARITY UNPINNED.
        from . import filter_dropout as ofd
        U = uimg_ab.shape[0]
        ctx = nets.Ctx(True, inject.get("drop_FP"), True)
        feats = nets.encoder_2d(sd, uimg_ab, ctx)
        scores_in = inject.get("sim_score")
        if scores_in is None and inject.get("gradsim") is not None:          # gradsim.get_sim() (:360): the previous iteration's scores
            scores_in = [sc.clone() for sc in inject["gradsim"]]
        f1, f2 = ofd.perform_dropout(feats, [0, 1, 2, 3, 4], scores_in, a.get("comp_drop", False),
                                     inject["fp_uniforms"], inject.get("fp_branches"))
        o1fp, o2fp = nets.decoder_2d(sd, "decoder1", f1, ctx), nets.decoder_2d(sd, "decoder2", f2, ctx)
        t1, t2 = torch.cat((arg1, arg1[U // 2:])), torch.cat((arg2, arg2[U // 2:]))
        fp_terms = (F.cross_entropy(o1fp, t2), F.cross_entropy(o2fp, t1))
        fp_loss = fp_terms[0] + fp_terms[1]
        if inject.get("sim_score") is None and inject.get("gradsim") is not None:
            # gradsim.get_grad_convkernel(loss_l, loss_u, ...) (:352-353,365): mix_loss returns (loss_image, loss_patch, total) =
            # (loss_u_out, loss_l_in, .) for the unlabeled rows and (loss_l_out, loss_u_in, .) for the labeled ones (:345-349)
            loss_l = m1[1] + m2[1] + m3[0] + m4[0]
            loss_u = m1[0] + m2[0] + m3[1] + m4[1]
            for dst, new in zip(inject["gradsim"], grad_sim_scores(loss_l, loss_u, sd)):
                dst.copy_(new)
    loss = bcp_loss + cw * (fp_loss + vat_loss)
    names = [k for k, v in sd.items() if v.is_floating_point() and v.requires_grad]
    grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(sd[k]) for g, k in zip(grads, names)]
    sgd_step([sd[k] for k in names], grads, [moms[k] for k in names], lr, a["momentum"], a["weight_decay"])
    return dict(losses=[m1, m2, m3, m4], vat_loss=vat_loss.detach(), bcp_loss=bcp_loss.detach(), loss=loss.detach(),
                grads=dict(zip(names, grads)), fp_losses=None if fp_terms is None else [t.detach() for t in fp_terms])


def dice_loss_std(soft, target, n_classes, smooth=1e-5):
    """losses.DiceLoss (ABSENT upstream; SSL4MIS definition, what train_ablation_2D.py:144,172-176 calls with
    softmax inputs and label.unsqueeze(1)): per class 1 - (2 sum s t + eps)/(sum s^2 + sum t^2 + eps), mean over classes."""
    loss = 0.0
    for c in range(n_classes):
        s = soft[:, c]
        t = (target == c).to(soft.dtype)
        inter = (s * t).sum()
        loss = loss + (1.0 - (2.0 * inter + smooth) / ((s * s).sum() + (t * t).sum() + smooth))
    return loss / n_classes


def ablation_iteration(sd, moms, volume_batch, label_batch, iter_num, lr, args=None, inject=None, net=None):
    """One iteration of the ablation loop (train_ablation_2D.py:159-246) on CPU: full-batch forward, supervised
    0.5*(CE + Dice) per head on the labeled half (:171-176), cross pseudo supervision on the unlabeled half (:203-207,
    216-217), create_maskV1 + VAT2d (:228-230; on the unlabeled half, see chap_amd.train.AblationStep), loss (:236),
    SGD (:238-241).  inject: {'drop_F': masks of the full-batch forward, 'drop_V0'.., 'drop_VF', 'd0'}."""
    from . import nets
    a = dict(ORACLE_ARGS)
    a.update(dict(w_adv=1.0, w_drop=1.0))
    a.update(args or {})
    inject = inject or {}
    net = net or nets.dual_decoder_2d
    nc, lbs = a["num_classes"], a["labeled_bs"]
    out1, out2 = net(sd, volume_batch, train=True, drop=inject.get("drop_F"))
    soft1, soft2 = torch.softmax(out1, dim=1), torch.softmax(out2, dim=1)
    cw = consistency_weight(iter_num, a["consistency"], a["consistency_rampup"])
    lab = label_batch[:lbs].long()
    loss1 = 0.5 * (F.cross_entropy(out1[:lbs], lab) + dice_loss_std(soft1[:lbs], lab, nc))
    loss2 = 0.5 * (F.cross_entropy(out2[:lbs], lab) + dice_loss_std(soft2[:lbs], lab, nc))
    arg1 = torch.max(soft1[lbs:].detach(), dim=1)[1]
    arg2 = torch.max(soft2[lbs:].detach(), dim=1)[1]
    ps1 = F.cross_entropy(out1[lbs:], arg2.long(), reduction="none")
    ps2 = F.cross_entropy(out2[lbs:], arg1.long(), reduction="none")
    knowledge = (ps1 + ps2).detach()
    model1_loss = loss1 + cw * ps1.mean()
    model2_loss = loss2 + cw * ps2.mean()
    if a["adv_noise"]:
        diff = create_mask_v1(arg1, arg2, knowledge, 4, a["topk1"]).unsqueeze(1)
        calls = {"k": 0}

        def model_fn(xx):
            k = calls["k"]
            calls["k"] += 1
            key = "drop_V%d" % k if k < a["vat_iters"] else "drop_VF"
            return net(sd, xx, train=True, drop=inject.get(key), update_stats=False)

        x_u = volume_batch[lbs:]
        d0 = inject["d0"] if inject.get("d0") is not None else torch.rand(x_u.shape) - 0.5
        vat_loss, _ = vat2d(model_fn, x_u, soft1[lbs:].detach(), soft2[lbs:].detach(), diff, d0, a["noise_mag"], a["epi"], a["vat_iters"], a["vat_sign"],
                            a["adv_losstype"])
    else:
        vat_loss = torch.zeros(())
    fp_loss = torch.zeros(())
    loss = model1_loss + model2_loss + cw * (vat_loss * a["w_adv"] + fp_loss * a["w_drop"])
    names = [k for k, v in sd.items() if v.is_floating_point() and v.requires_grad]
    grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(sd[k]) for g, k in zip(grads, names)]
    sgd_step([sd[k] for k in names], grads, [moms[k] for k in names], lr, a["momentum"], a["weight_decay"])
    return dict(sup=[loss1.detach(), loss2.detach()], cps=[ps1.mean().detach(), ps2.mean().detach()], vat_loss=vat_loss.detach(),
                loss=loss.detach(), consistency_weight=cw)


from chap_amd.synthetic import synthetic_batch, synthetic_batch_3d  # noqa: E402,F401  (data generators live with the product: bench.py must not import oracle/ for data)
