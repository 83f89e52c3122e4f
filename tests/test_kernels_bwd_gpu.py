"""Backward / training-loop kernels against plain PyTorch fp32 autograd and the oracle's
train_step restatement.  Needs a GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from chap_amd import _lib as L
from chap_amd import ops
from oracle import train_step as ots
from tests.test_kernels_gpu import DEV, TOL, cl, lazy_ref, relerr, rq, uncl


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,hw", [(16, 16, (20, 24)), (32, 64, (16, 32)), (128, 32, (8, 16))])
def test_wgrad_conv3x3(dtype, cin, cout, hw):
    g = torch.Generator().manual_seed(11)
    N, (H, W) = 2, hw
    x = rq(torch.randn(N, cin, H, W, generator=g), dtype)
    gy = rq(torch.randn(N, cout, H, W, generator=g), dtype)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.2
    a = rq(lazy_ref(x, sc, sh, 0.01, None, 1.0), dtype)
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(a, w, None, padding=1).backward(gy)
    dw = torch.full((cout, cin, 3, 3), 0.5, device=DEV)       # accumulate on top of existing content
    db = torch.zeros(cout, device=DEV)
    lz = ops.Lazy(cl(x, dtype), sc.to(DEV), sh.to(DEV), True, 0.01)
    ops.wgrad([lz], ops.Lazy(cl(gy, dtype)), dw, (1, 9, cin * 9), grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2, db=db)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert relerr(dw - 0.5, w.grad) < tol
    assert relerr(db, gy.sum((0, 2, 3))) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_wgrad_concat_1x1_deconv_down(dtype):
    g = torch.Generator().manual_seed(12)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    N, H, W = 2, 12, 16
    # concat sources (decoder ConvBlock input)
    x0 = rq(torch.randn(N, 16, H, W, generator=g), dtype)
    x1 = rq(torch.randn(N, 16, H, W, generator=g), dtype)
    gy = rq(torch.randn(N, 16, H, W, generator=g), dtype)
    w = torch.zeros(16, 32, 3, 3, requires_grad=True)
    F.conv2d(torch.cat([x0, x1], 1), w, None, padding=1).backward(gy)
    dw = torch.zeros(16, 32, 3, 3, device=DEV)
    ops.wgrad([ops.Lazy(cl(x0, dtype)), ops.Lazy(cl(x1, dtype))], ops.Lazy(cl(gy, dtype)), dw, (1, 9, 32 * 9),
              grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2)
    assert relerr(dw, w.grad) < tol
    # 1x1 conv 64 -> 32
    x = rq(torch.randn(N, 64, H, W, generator=g), dtype)
    gy = rq(torch.randn(N, 32, H, W, generator=g), dtype)
    w = torch.zeros(32, 64, 1, 1, requires_grad=True)
    F.conv2d(x, w).backward(gy)
    dw = torch.zeros(32, 64, 1, 1, device=DEV)
    ops.wgrad([ops.Lazy(cl(x, dtype))], ops.Lazy(cl(gy, dtype)), dw, (1, 1, 64), grid=(N, 1, H, W), in_dims=(1, H, W), ksize=1, stride=1, dims=2)
    assert relerr(dw, w.grad) < tol
    # transposed conv k2 s2 2D: dW[ci][co][sub] = sum_p x[p][ci] * g[2p+sub][co]  (roles swapped)
    x = rq(torch.randn(N, 64, H, W, generator=g), dtype)
    gy = rq(torch.randn(N, 32, 2 * H, 2 * W, generator=g), dtype)
    w = torch.zeros(64, 32, 2, 2, requires_grad=True)
    b = torch.zeros(32, requires_grad=True)
    F.conv_transpose2d(x, w, b, stride=2).backward(gy)
    dw = torch.zeros(64, 32, 2, 2, device=DEV)
    # kernel: A = fine gradient (kc = co), B = coarse input (kn = ci): strides (tap, kc=co, kn=ci)
    ops.wgrad([ops.Lazy(cl(gy, dtype))], ops.Lazy(cl(x, dtype)), dw, (1, 4, 32 * 4), grid=(N, 1, H, W), in_dims=(1, 2 * H, 2 * W), ksize=2, stride=2, dims=2)
    assert relerr(dw, w.grad) < tol
    # 3D down conv k2 s2 16 -> 32
    x = rq(torch.randn(1, 16, 8, 8, 16, generator=g), dtype)
    gy = rq(torch.randn(1, 32, 4, 4, 8, generator=g), dtype)
    w = torch.zeros(32, 16, 2, 2, 2, requires_grad=True)
    F.conv3d(x, w, None, stride=2).backward(gy)
    dw = torch.zeros(32, 16, 2, 2, 2, device=DEV)
    ops.wgrad([ops.Lazy(cl(x, dtype))], ops.Lazy(cl(gy, dtype)), dw, (1, 8, 16 * 8), grid=(1, 4, 4, 8), in_dims=(8, 8, 16), ksize=2, stride=2, dims=3)
    assert relerr(dw, w.grad) < tol
    # 3D 3^3 conv 32 -> 16
    x = rq(torch.randn(1, 32, 5, 9, 16, generator=g), dtype)
    gy = rq(torch.randn(1, 16, 5, 9, 16, generator=g), dtype)
    w = torch.zeros(16, 32, 3, 3, 3, requires_grad=True)
    F.conv3d(x, w, None, padding=1).backward(gy)
    dw = torch.zeros(16, 32, 3, 3, 3, device=DEV)
    ops.wgrad([ops.Lazy(cl(x, dtype))], ops.Lazy(cl(gy, dtype)), dw, (1, 27, 32 * 27), grid=(1, 5, 9, 16), in_dims=(5, 9, 16), ksize=3, stride=1, dims=3)
    assert relerr(dw, w.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_act_bn_backward_with_pool(dtype):
    """d/d(raw) of  sum(g1 * a) + sum(gp * maxpool(a)),  a = dropout(leaky(BN_train(raw)))."""
    g = torch.Generator().manual_seed(13)
    N, C, H, W = 2, 32, 12, 16
    raw = rq(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3, dtype)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    keep = (torch.rand(N, C, H, W, generator=g) > 0.2).float()
    g1 = rq(torch.randn(N, C, H, W, generator=g), dtype)
    gp = rq(torch.randn(N, C, H // 2, W // 2, generator=g), dtype)
    r = raw.clone().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = F.leaky_relu(F.batch_norm(r, None, None, gm, bt, True, 0.1, 1e-5), 0.01) * keep / 0.8
    ((a * g1).sum() + (F.max_pool2d(a, 2) * gp).sum()).backward()
    # device: forward pieces
    cnt = N * H * W
    mean = raw.mean((0, 2, 3)); var = raw.var((0, 2, 3), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    scale = gamma * invstd; shift = beta - mean * scale
    lz = ops.Lazy(cl(raw, dtype), scale.to(DEV), shift.to(DEV), True, 0.01, keep=cl(keep, torch.uint8), keep_scale=1 / 0.8)
    pooled = torch.empty(N, 1, H // 2, W // 2, C, device=DEV, dtype=dtype)
    idx = torch.empty(N, 1, H // 2, W // 2, C, device=DEV, dtype=torch.uint8)
    ops.act_pool2(lz, pooled, idx)
    gout = torch.empty(N, 1, H, W, C, device=DEV, dtype=dtype)
    dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    # split g1 over two gradient sources to exercise the multi-source sum (one inside a wider buffer)
    ga = rq(g1 * 0.25, dtype); gb = g1 - ga
    wide = torch.zeros(N, 1, H, W, 2 * C, device=DEV, dtype=dtype); wide[..., C:] = cl(gb, dtype)
    ops.act_bwd(lz, [(cl(ga, dtype), 0), (wide, C)], gout, g_pool=cl(gp, dtype), pool_idx=idx,
                mean=mean.to(DEV), invstd=invstd.to(DEV), gamma=gamma.to(DEV), dgamma=dgamma, dbeta=dbeta, count=cnt)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert relerr(uncl(gout).squeeze(2), r.grad) < tol
    assert relerr(dgamma, gm.grad) < tol and relerr(dbeta, bt.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dims", [2, 3])
def test_first_conv_backward(dtype, dims):
    g = torch.Generator().manual_seed(14)
    shape = (2, 1, 20, 28) if dims == 2 else (1, 6, 10, 12)
    x = torch.randn(shape, generator=g)
    w = (torch.randn(16, 1, *([3] * dims), generator=g) / 3).requires_grad_(True)
    b = torch.zeros(16, requires_grad=True)
    xin = (x[:, 0].unsqueeze(1) if dims == 2 else x.unsqueeze(1)).clone().requires_grad_(True)
    y = (F.conv2d if dims == 2 else F.conv3d)(xin, w, b, padding=1)
    gy = rq(torch.randn(y.shape, generator=g), dtype)
    y.backward(gy)
    gd = cl(gy, dtype)
    dx = torch.empty(shape, device=DEV)
    dw = torch.zeros_like(w, device=DEV); db = torch.zeros(16, device=DEV)
    ops.conv_c1_bwd(gd, w.detach().to(DEV), x.to(DEV), dims=dims, dx=dx, dw=dw, db=db)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    assert relerr(dx, xin.grad.reshape(shape)) < tol
    assert relerr(dw, w.grad) < tol and relerr(db, b.grad) < tol


def test_losses_vs_oracle():
    g = torch.Generator().manual_seed(15)
    N, C, H, W = 3, 4, 32, 48
    logits = torch.randn(N, C, H, W, generator=g) * 2
    ta = torch.randint(0, C, (N, H, W), generator=g); tb = torch.randint(0, C, (N, H, W), generator=g)
    _, lm = ots.box_masks(N, H, W, 5, 9)
    lo = logits.clone().requires_grad_(True)
    li, lp, tot = ots.mix_loss(lo, ta, tb, lm, u_weight=0.5, unlab=True)
    tot.backward()
    ld = logits.to(DEV)
    loss, acc = ops.mix_loss_fwd(ld, ta.to(DEV), tb.to(DEV), lm.long().to(DEV), 0.5, 1.0)
    dl = torch.empty_like(ld)
    ops.mix_loss_bwd(ld, ta.to(DEV), tb.to(DEV), lm.long().to(DEV), 0.5, 1.0, acc, dl)
    assert relerr(loss.cpu(), torch.stack([li, lp, tot]).detach()) < 1e-5
    assert relerr(dl, lo.grad) < 1e-4
    # pseudo block
    l2 = torch.randn(N, C, H, W, generator=g) * 2
    s1, s2, a1, a2, kn = ots.pseudo_block(logits, l2)
    ds1, ds2, da1, da2, dkn = ops.pseudo_block(ld, l2.to(DEV))
    assert relerr(ds1, s1) < 1e-5 and relerr(ds2, s2) < 1e-5 and relerr(dkn, kn) < 1e-5
    assert (da1.cpu() == a1).all() and (da2.cpu() == a2).all()
    # KL two heads + gradient
    la, lb = logits.clone().requires_grad_(True), l2.clone().requires_grad_(True)
    t1, t2 = F.softmax(torch.randn(N, C, H, W, generator=g), 1), F.softmax(torch.randn(N, C, H, W, generator=g), 1)
    kl = ots.kl_two_heads((la, lb), (t1, t2)); (kl * 0.7).backward()
    lossd = torch.zeros(1, device=DEV); g1 = torch.empty_like(ld); g2 = torch.empty_like(ld)
    ops.kl_fwd_bwd((ld, l2.to(DEV)), (t1.to(DEV), t2.to(DEV)), lossd, (g1, g2), gscale=0.7)
    assert relerr(lossd.cpu(), kl.detach().reshape(1)) < 1e-5
    assert relerr(g1, la.grad) < 1e-4 and relerr(g2, lb.grad) < 1e-4
    # Dice distance of the two heads (--adv_losstype dice) + gradient; loss accumulates into `loss` (+=)
    la, lb = logits.clone().requires_grad_(True), l2.clone().requires_grad_(True)
    dd = ots.dice_two_heads((la, lb), (t1, t2)); (dd * 0.7).backward()
    lossd = torch.full((1,), 0.25, device=DEV)
    ops.kl_fwd_bwd((ld, l2.to(DEV)), (t1.to(DEV), t2.to(DEV)), lossd, (g1, g2), gscale=0.7, mode="dice")
    assert relerr(lossd.cpu() - 0.25, dd.detach().reshape(1)) < 1e-5
    assert relerr(g1, la.grad) < 1e-4 and relerr(g2, lb.grad) < 1e-4
    # every reduction on the way is fixed-order: bitwise identical on a second run
    loss_b, acc_b = ops.mix_loss_fwd(ld, ta.to(DEV), tb.to(DEV), lm.long().to(DEV), 0.5, 1.0)
    assert torch.equal(loss_b.view(torch.int32), loss.view(torch.int32))


def test_mix_loss_and_box_kernels_against_the_reference_functions(golden_dir):
    """chap_mix_loss_fwd/bwd, chap_box_mask against outputs of the reference's own mix_loss / generate_mask
    (tests/golden/train_plumbing.npz, see tests/test_oracle_golden.py)."""
    import os
    import numpy as np
    z = np.load(os.path.join(golden_dir, "train_plumbing.npz"))
    logits = torch.from_numpy(z["logits"]).to(DEV)
    img_l, patch_l = torch.from_numpy(z["img_l"]).to(DEV), torch.from_numpy(z["patch_l"]).to(DEV)
    N, _, H, W = logits.shape
    box = torch.tensor([int(v) for v in z["box"]], dtype=torch.int32, device=DEV)
    lm = torch.empty(N, H, W, dtype=torch.int64, device=DEV)
    ops.box_mask(lm, box)
    assert np.array_equal(lm.cpu().numpy(), z["loss_mask"])
    for tag, (wa, wb) in (("lab", (1.0, 0.5)), ("unlab", (0.5, 1.0)), ("w", (0.7, 0.3))):      # (image_weight, patch_weight), train_ours_2D.py:201-204
        loss, acc = ops.mix_loss_fwd(logits, img_l, patch_l, lm, wa, wb)
        dl = torch.empty_like(logits)
        ops.mix_loss_bwd(logits, img_l, patch_l, lm, wa, wb, acc, dl)
        assert relerr(loss.cpu(), torch.from_numpy(z["%s_losses" % tag]).float()) < 1e-5, tag
        assert relerr(dl, torch.from_numpy(z["%s_dlogits" % tag])) < 1e-4, tag


def test_largest_cc_tie_goes_to_the_component_met_first_in_raster_order():
    """get_ACDC_2DLargestCC (train_ours_2D.py:134-136) keeps np.argmax(np.bincount(labels.flat)[1:]) + 1: on equal sizes the
    FIRST label, and skimage / scipy number components in raster order of their first pixel."""
    lab = torch.zeros(3, 16, 24, dtype=torch.int64)
    lab[0, 2:5, 3:6] = 1; lab[0, 9:12, 15:18] = 1                  # two 3x3 squares of class 1: the upper-left one is met first
    lab[1, 10:12, 1:4] = 2; lab[1, 2:4, 18:21] = 2                 # two 2x3 blocks of class 2: the one whose first pixel comes first (row 2) wins
    lab[1, 6, 0:5] = 3; lab[1, 6, 8:13] = 3; lab[1, 7, 20:24] = 3  # class 3: two 5-pixel runs on one row (left wins) and a smaller one
    lab[2, 5:9, 5:9] = 1; lab[2, 0:4, 12:16] = 1; lab[2, 12, 2] = 1  # 4x4 squares: the one starting in row 0 wins although it lies to the right
    ref = ots.largest_cc(lab, 4)
    got = ops.largest_cc(lab.to(DEV), 4).cpu()
    assert torch.equal(got, ref)
    assert got[0, 3, 4] == 1 and got[0, 10, 16] == 0 and got[1, 2, 19] == 2 and got[1, 10, 2] == 0
    assert got[1, 6, 2] == 3 and got[1, 6, 10] == 0 and got[2, 1, 13] == 1 and got[2, 6, 6] == 0
    vol = torch.zeros(1, 6, 8, 8, dtype=torch.int64)                # 3D (26-connectivity): two 2x2x2 cubes, the one in the lower z wins
    vol[0, 0:2, 4:6, 4:6] = 1; vol[0, 3:5, 0:2, 0:2] = 1
    got3 = ops.largest_cc(vol.to(DEV), 2).cpu()
    assert torch.equal(got3, ots.largest_cc(vol, 2)) and got3[0, 0, 4, 4] == 1 and got3[0, 3, 0, 0] == 0


def test_lcc_diffmask_vat_helpers_sgd():
    g = torch.Generator().manual_seed(16)
    N, H, W = 5, 64, 96
    # blobby label maps: threshold smooth noise
    z = F.avg_pool2d(torch.randn(N, 4, H + 8, W + 8, generator=g), 9, stride=1)
    lab = z.argmax(1)
    lab[0] = 0                      # empty image: nothing to keep
    lab[1, :, :] = 2                # one full component
    ref = ots.largest_cc(lab, 4)
    got = ops.largest_cc(lab.to(DEV), 4)
    assert (got.cpu() == ref).all()
    # diff mask
    p1, p2 = lab, torch.roll(lab, 1, 2)
    kn = torch.rand(N, H, W, generator=g) * 3
    ref = ots.create_mask_v1(p1, p2, kn, 4, 0.1)
    got = ops.diff_mask(p1.to(DEV), p2.to(DEV), kn.to(DEV), 4, 0.1)
    assert (got.cpu() == ref).all()
    # l2 normalise / perturb
    d = torch.randn(N, 1, H, W, generator=g)
    out = torch.empty_like(d, device=DEV)
    ops.l2_normalize(d.to(DEV), out)
    assert relerr(out, ots.l2_normalize(d)) < 1e-5
    x = torch.rand(N, 1, H, W, generator=g); m = (torch.rand(N, 1, H, W, generator=g) > 0.5).float()
    o = torch.empty_like(x, device=DEV)
    ops.perturb(x.to(DEV), d.to(DEV), o, 6.0, mask=m.to(DEV))
    assert relerr(o, x + 6.0 * m * d) < 1e-6
    ops.perturb(x.to(DEV), d.to(DEV), o, 0.5, sign=True)
    assert relerr(o, x + 0.5 * torch.sign(d)) < 1e-6
    # rng: range, mean, reproducibility, seed_dev changes the stream
    r1 = torch.empty(1 << 16, device=DEV); r2 = torch.empty(1 << 16, device=DEV)
    sd = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.rand_uniform(r1, 123, -0.5, 0.5, seed_dev=sd); ops.rand_uniform(r2, 123, -0.5, 0.5, seed_dev=sd)
    assert (r1 == r2).all() and r1.min() >= -0.5 and r1.max() < 0.5 and abs(r1.mean().item()) < 5e-3
    sd += 1
    ops.rand_uniform(r2, 123, -0.5, 0.5, seed_dev=sd)
    assert (r1 != r2).float().mean() > 0.99
    km = torch.empty(1 << 16, dtype=torch.uint8, device=DEV)
    ops.keep_mask(km, 5, 0.3)
    assert abs(km.float().mean().item() - 0.7) < 1e-2
    # box mix + loss mask
    box = torch.tensor([5, 9, int(H * 2 / 3), int(W * 2 / 3)], dtype=torch.int32, device=DEV)
    a, b = torch.rand(N, 1, H, W, generator=g), torch.rand(N, 1, H, W, generator=g)
    mask, lm = ots.box_masks(N, H, W, 5, 9)
    o = torch.empty_like(a, device=DEV)
    ops.box_mix(a.to(DEV), b.to(DEV), o, box)
    assert relerr(o, a * mask + b * (1 - mask)) < 1e-7
    lmd = torch.empty(N, H, W, dtype=torch.int64, device=DEV)
    ops.box_mask(lmd, box)
    assert (lmd.cpu() == lm.long()).all()
    # SGD with momentum and weight decay, two steps, odd length
    n = 1003
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g)
    params, moms = [p.clone()], [torch.zeros(n)]
    pd, gd, md = p.to(DEV), gr.to(DEV), torch.zeros(n, device=DEV)
    lr = torch.tensor([0.01], device=DEV)
    for step in range(2):
        ots.sgd_step(params, [gr], moms, 0.01)
        ops.sgd_step(pd, gd, md, lr, 0.9, 1e-4, zero_grad=False)
    assert relerr(pd, params[0]) < 1e-6 and relerr(md, moms[0]) < 1e-6
    ops.sgd_step(pd, gd, md, lr, 0.9, 1e-4, zero_grad=True)
    assert gd.abs().max().item() == 0


def test_lcc_and_box_3d():
    g = torch.Generator().manual_seed(17)
    N, D, H, W = 3, 12, 20, 24
    z = F.avg_pool3d(torch.randn(N, 3, D + 4, H + 4, W + 4, generator=g), 5, stride=1)
    lab = z.argmax(1)
    lab[0] = 0
    ref = ots.largest_cc(lab, 3)
    got = ops.largest_cc(lab.to(DEV), 3)
    assert (got.cpu() == ref).all()
    box = torch.tensor([1, 3, 5, int(D * 2 / 3), int(H * 2 / 3), int(W * 2 / 3)], dtype=torch.int32, device=DEV)
    mask, lm = ots.box_masks_3d(N, D, H, W, 1, 3, 5)
    a, b = torch.rand(N, 1, D, H, W, generator=g), torch.rand(N, 1, D, H, W, generator=g)
    o = torch.empty_like(a, device=DEV)
    ops.box_mix(a.to(DEV), b.to(DEV), o, box)
    assert relerr(o, a * mask + b * (1 - mask)) < 1e-7
    lmd = torch.empty(N, D, H, W, dtype=torch.int64, device=DEV)
    ops.box_mask(lmd, box)
    assert (lmd.cpu() == lm.long()).all()
    kn = torch.rand(N, D, H, W, generator=g) * 3
    p1, p2 = lab, torch.roll(lab, 1, 3)
    assert (ops.diff_mask(p1.to(DEV), p2.to(DEV), kn.to(DEV), 4, 0.1).cpu() == ots.create_mask_v1(p1, p2, kn, 4, 0.1)).all()


@pytest.mark.parametrize("brick", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,cin,cout,add2", [((2, 10, 14, 14), 32, 32, False), ((1, 9, 13, 20), 16, 16, False), ((2, 5, 7, 7), 64, 128, False),
                                                 ((1, 10, 14, 14), 32, 16, True)])
def test_wgrad_conv3d_ragged(dtype, shape, cin, cout, add2, brick, monkeypatch):
    """3x3x3 weight gradient on ragged grids (the V-Net's 14x14x10 / 7x7x5 levels): lazy BN/ReLU A operand with
    Dropout3d multipliers, optional skip add, 16- and 32-wide B tiles, bias gradient, accumulation into existing dW.
    `brick`: the 4 x 4 x 16 brick tiling of the large-volume levels (bf16), forced onto these small ragged grids
    (D, H not multiples of 4, W not a multiple of 16) through the library's lab knob; off = the 1 x 4 x 16 slabs."""
    if brick and dtype != torch.bfloat16:
        pytest.skip("brick tiles are bf16 only")
    monkeypatch.setenv("CHAP_WGRAD_BRICK", "1" if brick else "0")
    g = torch.Generator().manual_seed(21)
    N, D, H, W = shape
    x = rq(torch.randn(N, cin, D, H, W, generator=g), dtype)
    gy = rq(torch.randn(N, cout, D, H, W, generator=g), dtype)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.2
    cm = (torch.rand(N, cin, generator=g) > 0.3).float() * 1.5
    a = lazy_ref(x, sc, sh, 0.0, None, 1.0) * cm.view(N, cin, 1, 1, 1)
    srcs = [ops.Lazy(cl(x, dtype), sc.to(DEV), sh.to(DEV), True, 0.0, chan_mul=cm.to(DEV))]
    if add2:
        x1 = rq(torch.randn(N, cin, D, H, W, generator=g), dtype)
        a = a + x1
        srcs.append(ops.Lazy(cl(x1, dtype)))
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(rq(a, dtype), w, None, padding=1).backward(gy)
    dw = torch.full((cout, cin, 3, 3, 3), 0.25, device=DEV)
    db = torch.zeros(cout, device=DEV)
    ops.wgrad(srcs, ops.Lazy(cl(gy, dtype)), dw, (1, 27, cin * 27), grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=3,
              combine=1 if add2 else 0, db=db)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert relerr(dw - 0.25, w.grad) < tol
    assert relerr(db, gy.sum((0, 2, 3, 4))) < tol


@pytest.mark.parametrize("mr", [2, 1])
@pytest.mark.parametrize("cins,cout,hw,keep", [([16], 16, (40, 52), True), ([16, 16], 16, (37, 50), False), ([32], 32, (24, 40), False), ([32, 32], 32, (19, 33), True),
                                               ([16], 4, (33, 47), False)])
def test_wgrad_wave_private_2d(cins, cout, hw, keep, mr, monkeypatch):
    """The wave-private weight-gradient kernel of the 2D full-resolution layers (csrc/wgrad_wp.h; bf16, round 4), forced onto small RAGGED grids
    (H not a multiple of the 4 / 8 tile rows, W not a multiple of 16) through the library's lab knobs: one and two concatenated lazy A sources
    (BatchNorm affine + LeakyReLU, element keep mask), 16- and 32-wide B tiles, a padded head (kn_valid), bias gradient, accumulation into existing
    dW -- against PyTorch's conv2d weight gradient, and against the block-tile kernel (CHAP_WGRAD_WP=0) to fp32 rounding."""
    monkeypatch.setenv("CHAP_WGRAD_WP_MR", str(mr))
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(31)
    N, (H, W) = 3, hw
    cin = sum(cins)
    xs = [rq(torch.randn(N, c, H, W, generator=g), dtype) for c in cins]
    aff = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2) for c in cins]
    km = (torch.rand(N, cins[0], H, W, generator=g) > 0.3) if keep else None
    acts = []
    for i, (x, (sc, sh)) in enumerate(zip(xs, aff)):
        acts.append(lazy_ref(x, sc, sh, 0.01, km if i == 0 else None, 1.25 if (i == 0 and keep) else 1.0))
    a = rq(torch.cat(acts, 1), dtype)
    cb = max(cout, 16)                                   # the head's gradient arrives padded to 16 channels
    gy = rq(torch.randn(N, cb, H, W, generator=g), dtype)
    if cb != cout:
        gy[:, cout:] = 0
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(a, w, None, padding=1).backward(gy[:, :cout].contiguous())
    srcs = []
    for i, (x, (sc, sh)) in enumerate(zip(xs, aff)):
        kw = dict(keep=km.permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(torch.uint8).to(DEV), keep_scale=1.25) if (i == 0 and keep) else {}
        srcs.append(ops.Lazy(cl(x, dtype), sc.to(DEV), sh.to(DEV), True, 0.01, **kw))
    outs = {}
    for wp in ("1", "0"):
        monkeypatch.setenv("CHAP_WGRAD_WP", wp)
        dw = torch.full((cout, cin, 3, 3), 0.5, device=DEV)
        db = torch.zeros(cout, device=DEV)
        ops.wgrad(srcs, ops.Lazy(cl(gy, dtype)), dw, (1, 9, cin * 9), grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2, db=db,
                  kn_valid=cout if cb != cout else 0)
        torch.cuda.synchronize()
        outs[wp] = (dw.cpu() - 0.5, db.cpu())
    assert relerr(outs["1"][0], w.grad) < 2e-2 and relerr(outs["1"][1], gy[:, :cout].sum((0, 2, 3))) < 2e-2
    assert relerr(outs["1"][0], outs["0"][0]) < 2e-5 and relerr(outs["1"][1], outs["0"][1]) < 2e-5      # same products, another summation order
    assert float(outs["1"][0].abs().max()) > 0
