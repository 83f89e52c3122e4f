"""Per-kernel parity: each C-ABI entry point against a plain PyTorch fp32 CPU computation of the
same op (floating-point kernels; tolerance stated per test).  Needs a GPU."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from chap_amd import _lib as L
from chap_amd import ops

DEV = "cuda"
TOL = {torch.float32: 2e-5, torch.bfloat16: 2.5e-2}


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def cl(x, dtype):
    """NC(D)HW cpu fp32 -> [N,D,H,W,C] device tensor of dtype."""
    if x.dim() == 4:
        x = x.unsqueeze(2)
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, dtype)


def uncl(t):
    """[N,D,H,W,C] device -> NCDHW cpu fp32."""
    return t.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def rq(x, dtype):
    """round-trip through the storage dtype so the reference sees the same inputs."""
    return x.to(dtype).float()


def lazy_ref(x, scale, shift, slope, keep, ks):
    """reference for the lazy-activation transform on an NC... cpu tensor."""
    sh = [1, -1] + [1] * (x.dim() - 2)
    y = x * scale.view(sh) + shift.view(sh)
    y = torch.where(y > 0, y, y * slope)
    if keep is not None:
        y = y * keep * ks
    return y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,hw", [(16, 16, (20, 24)), (32, 64, (16, 16)), (64, 32, (9, 17)), (128, 128, (8, 16))])
def test_conv3x3_2d_plain(dtype, cin, cout, hw):
    g = torch.Generator().manual_seed(1)
    N, (H, W) = 2, hw
    x = rq(torch.randn(N, cin, H, W, generator=g), dtype)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, rq(w, dtype), b, padding=1)
    xd = cl(x, dtype)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, cin, cout, 9)
    out = torch.empty(N, 1, H, W, cout, device=DEV, dtype=dtype)
    stats = ops.stats_buffer(cout, DEV)
    c0 = torch.randn(cout, generator=g)                      # shift of the moments: sum(v - c), sum((v - c)^2)
    ops.conv_fwd([ops.Lazy(xd)], wp, b.to(DEV), cout, out, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2,
                 stats=stats, stats_shift=c0.to(DEV))
    torch.cuda.synchronize()
    assert relerr(uncl(out).squeeze(2), ref) < TOL[dtype]
    s = ops.stats_totals(stats, cout).float().cpu()
    rc = ref - c0.view(1, -1, 1, 1)
    assert relerr(s[0], rc.sum((0, 2, 3))) < 1e-3 + TOL[dtype]
    assert relerr(s[1], (rc * rc).sum((0, 2, 3))) < 1e-3 + TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3x3_concat_lazy_sources(dtype):
    """two concatenated sources, each with BN-affine + LeakyReLU on load, one with a dropout keep mask."""
    g = torch.Generator().manual_seed(2)
    N, H, W, c0, c1, cout = 2, 12, 20, 16, 16, 32
    x0 = rq(torch.randn(N, c0, H, W, generator=g), dtype)
    x1 = rq(torch.randn(N, c1, H, W, generator=g), dtype)
    sc0, sh0 = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.2
    sc1, sh1 = torch.rand(c1, generator=g) + 0.5, torch.randn(c1, generator=g) * 0.2
    keep = (torch.rand(N, c0, H, W, generator=g) > 0.3).float()
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / ((c0 + c1) * 9) ** 0.5
    a0 = lazy_ref(x0, sc0, sh0, 0.01, keep, 1 / 0.7)
    a1 = lazy_ref(x1, sc1, sh1, 0.01, None, 1.0)
    ref = F.conv2d(torch.cat([rq(a0, dtype), rq(a1, dtype)], 1), rq(w, dtype), None, padding=1)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, c0 + c1, cout, 9)
    out = torch.empty(N, 1, H, W, cout, device=DEV, dtype=dtype)
    s0 = ops.Lazy(cl(x0, dtype), sc0.to(DEV), sh0.to(DEV), True, 0.01, keep=cl(keep, torch.uint8), keep_scale=1 / 0.7)
    s1 = ops.Lazy(cl(x1, dtype), sc1.to(DEV), sh1.to(DEV), True, 0.01)
    ops.conv_fwd([s0, s1], wp, None, cout, out, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2)
    torch.cuda.synchronize()
    assert relerr(uncl(out).squeeze(2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv1x1_and_planar_head(dtype):
    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 10, 18
    x = rq(torch.randn(N, 64, H, W, generator=g), dtype)
    w = torch.randn(32, 64, 1, 1, generator=g) / 8
    b = torch.randn(32, generator=g)
    ref = F.conv2d(x, rq(w, dtype), b)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, 64, 32, 1)
    out = torch.empty(N, 1, H, W, 32, device=DEV, dtype=dtype)
    ops.conv_fwd([ops.Lazy(cl(x, dtype))], wp, b.to(DEV), 32, out, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=1, stride=1, dims=2)
    assert relerr(uncl(out).squeeze(2), ref) < TOL[dtype]
    # 3x3 head 16 -> 4, fp32 planar (NCHW) logits
    x = rq(torch.randn(N, 16, H, W, generator=g), dtype)
    w = torch.randn(4, 16, 3, 3, generator=g) / 12
    b = torch.randn(4, generator=g)
    ref = F.conv2d(x, rq(w, dtype), b, padding=1)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, 16, 4, 9)
    out = torch.empty(N, 4, H, W, device=DEV, dtype=torch.float32)
    ops.conv_fwd([ops.Lazy(cl(x, dtype))], wp, b.to(DEV), 4, out, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2,
                 out_planar=True, out_f32=True)
    assert relerr(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deconv_k2s2_2d_and_dgrads(dtype):
    g = torch.Generator().manual_seed(4)
    N, H, W, cin, cout = 2, 6, 10, 64, 32
    x = rq(torch.randn(N, cin, H, W, generator=g), dtype)
    w = torch.randn(cin, cout, 2, 2, generator=g) / 8
    b = torch.randn(cout, generator=g)
    ref = F.conv_transpose2d(x, rq(w, dtype), b, stride=2)
    wp = ops.pack_weights(w.to(DEV), L.PACK_DECONV_FWD, dtype, cin, cout, 4)
    # write into the upper half of a concat buffer (ld = 2*cout) to exercise out_ld/out_coff
    out = torch.zeros(N, 1, 2 * H, 2 * W, 2 * cout, device=DEV, dtype=dtype)
    ops.conv_fwd([ops.Lazy(cl(x, dtype))], wp, b.to(DEV), 4 * cout, out, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=1, stride=1, dims=2,
                 out_mode=1, out_cn=cout, out_coff=cout)
    assert relerr(uncl(out[..., cout:]).squeeze(2), ref) < TOL[dtype]
    assert out[..., :cout].abs().max().item() == 0
    # deconv input-gradient: d_in = conv k2 s2 of the output gradient
    gy = rq(torch.randn(N, cout, 2 * H, 2 * W, generator=g), dtype)
    ref_dx = F.conv2d(gy, rq(w, dtype), None, stride=2)
    wpd = ops.pack_weights(w.to(DEV), L.PACK_DECONV_DGRAD, dtype, cin, cout, 4)
    dx = torch.empty(N, 1, H, W, cin, device=DEV, dtype=dtype)
    ops.conv_fwd([ops.Lazy(cl(gy, dtype))], wpd, None, cin, dx, grid=(N, 1, H, W), in_dims=(1, 2 * H, 2 * W), ksize=2, stride=2, dims=2)
    assert relerr(uncl(dx).squeeze(2), ref_dx) < TOL[dtype]
    # conv3x3 input-gradient via flipped/transposed packing
    wc = torch.randn(cout, cin, 3, 3, generator=g) / 24
    gy = rq(torch.randn(N, cout, H, W, generator=g), dtype)
    ref_dx = F.conv_transpose2d(gy, rq(wc, dtype), None, padding=1)
    wpd = ops.pack_weights(wc.to(DEV), L.PACK_CONV_DGRAD, dtype, cin, cout, 9)
    dx = torch.empty(N, 1, H, W, cin, device=DEV, dtype=dtype)
    ops.conv_fwd([ops.Lazy(cl(gy, dtype))], wpd, None, cin, dx, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2)
    assert relerr(uncl(dx).squeeze(2), ref_dx) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3d_family(dtype):
    g = torch.Generator().manual_seed(5)
    N, D, H, W = 1, 6, 10, 12
    # 3^3 conv with add-combined lazy sources (V-Net skip add, vnet.py:202)
    c = 32
    x0 = rq(torch.randn(N, c, D, H, W, generator=g), dtype)
    x1 = rq(torch.randn(N, c, D, H, W, generator=g), dtype)
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    cm = (torch.rand(N, c, generator=g) > 0.5).float() * 2.0
    w = torch.randn(16, c, 3, 3, 3, generator=g) / (c * 27) ** 0.5
    a = lazy_ref(x0, sc, sh, 0.0, None, 1.0) * cm.view(N, c, 1, 1, 1) + x1
    ref = F.conv3d(rq(a, dtype), rq(w, dtype), None, padding=1)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, c, 16, 27)
    out = torch.empty(N, D, H, W, 16, device=DEV, dtype=dtype)
    s0 = ops.Lazy(cl(x0, dtype), sc.to(DEV), sh.to(DEV), True, 0.0, chan_mul=cm.to(DEV))
    ops.conv_fwd([s0, ops.Lazy(cl(x1, dtype))], wp, None, 16, out, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=3, combine=1)
    assert relerr(uncl(out), ref) < 2 * TOL[dtype]
    # k2 s2 down conv 16 -> 32
    x = rq(torch.randn(N, 16, D, H, W, generator=g), dtype)
    w = torch.randn(32, 16, 2, 2, 2, generator=g) / 11
    b = torch.randn(32, generator=g)
    ref = F.conv3d(x, rq(w, dtype), b, stride=2)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, 16, 32, 8)
    out = torch.empty(N, D // 2, H // 2, W // 2, 32, device=DEV, dtype=dtype)
    ops.conv_fwd([ops.Lazy(cl(x, dtype))], wp, b.to(DEV), 32, out, grid=(N, D // 2, H // 2, W // 2), in_dims=(D, H, W), ksize=2, stride=2, dims=3)
    assert relerr(uncl(out), ref) < TOL[dtype]
    # its input gradient (1x1 conv + depth-to-space)
    gy = rq(torch.randn(N, 32, D // 2, H // 2, W // 2, generator=g), dtype)
    ref_dx = F.conv_transpose3d(gy, rq(w, dtype), None, stride=2)
    wpd = ops.pack_weights(w.to(DEV), L.PACK_DOWN_DGRAD, dtype, 16, 32, 8)
    dx = torch.empty(N, D, H, W, 16, device=DEV, dtype=dtype)
    ops.conv_fwd([ops.Lazy(cl(gy, dtype))], wpd, None, 8 * 16, dx, grid=(N, D // 2, H // 2, W // 2), in_dims=(D // 2, H // 2, W // 2),
                 ksize=1, stride=1, dims=3, out_mode=1, out_cn=16)
    assert relerr(uncl(dx), ref_dx) < TOL[dtype]
    # transposed conv 3D 64 -> 32 with BN statistics per real channel
    x = rq(torch.randn(N, 64, 3, 5, 6, generator=g), dtype)
    w = torch.randn(64, 32, 2, 2, 2, generator=g) / 8
    b = torch.randn(32, generator=g)
    ref = F.conv_transpose3d(x, rq(w, dtype), b, stride=2)
    wp = ops.pack_weights(w.to(DEV), L.PACK_DECONV_FWD, dtype, 64, 32, 8)
    out = torch.empty(N, 6, 10, 12, 32, device=DEV, dtype=dtype)
    stats = ops.stats_buffer(8 * 32, DEV)
    c0 = torch.randn(32, generator=g)
    ops.conv_fwd([ops.Lazy(cl(x, dtype))], wp, b.to(DEV), 8 * 32, out, grid=(N, 3, 5, 6), in_dims=(3, 5, 6), ksize=1, stride=1, dims=3,
                 out_mode=1, out_cn=32, stats=stats, stats_shift=c0.to(DEV))
    assert relerr(uncl(out), ref) < TOL[dtype]
    st = ops.stats_totals(stats, 8 * 32, 32).float().cpu()
    rc = ref - c0.view(1, -1, 1, 1, 1)
    assert relerr(st[0], rc.sum((0, 2, 3, 4))) < 1e-3 + TOL[dtype]
    assert relerr(st[1], (rc * rc).sum((0, 2, 3, 4))) < 1e-3 + TOL[dtype]
    # ... and chap_bn_finalize folds the 8 sub-lattice rows: batch statistics of the 32 real channels
    gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
    scale, shift = torch.empty(32, device=DEV), torch.empty(32, device=DEV)
    mean, invstd = torch.empty(32, device=DEV), torch.empty(32, device=DEV)
    ops.bn_finalize(stats, gamma.to(DEV), beta.to(DEV), None, None, None, ref[:, 0].numel(), 1e-5, 0.0, scale, shift, mean, invstd,
                    stats_shift=c0.to(DEV), clog=8 * 32)
    assert relerr(mean, ref.mean((0, 2, 3, 4))) < 1e-3 + TOL[dtype]
    assert relerr(invstd, (ref.var((0, 2, 3, 4), unbiased=False) + 1e-5).rsqrt()) < 1e-3 + TOL[dtype]
    # 1x1x1 head 16 -> 2, planar fp32
    x = rq(torch.randn(N, 16, D, H, W, generator=g), dtype)
    w = torch.randn(2, 16, 1, 1, 1, generator=g) / 4
    b = torch.randn(2, generator=g)
    ref = F.conv3d(x, rq(w, dtype), b)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, 16, 2, 1)
    out = torch.empty(N, 2, D, H, W, device=DEV)
    ops.conv_fwd([ops.Lazy(cl(x, dtype))], wp, b.to(DEV), 2, out, grid=(N, D, H, W), in_dims=(D, H, W), ksize=1, stride=1, dims=3,
                 out_planar=True, out_f32=True)
    assert relerr(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dims", [2, 3])
def test_first_conv_c1(dtype, dims):
    g = torch.Generator().manual_seed(6)
    shape = (2, 1, 20, 28) if dims == 2 else (1, 6, 10, 12)
    x = torch.randn(shape, generator=g)
    taps = 9 if dims == 2 else 27
    w = torch.randn(16, 1, *([3] * dims), generator=g) / taps ** 0.5
    b = torch.randn(16, generator=g)
    xin = x[:, 0].unsqueeze(1) if dims == 2 else x.unsqueeze(1)
    ref = (F.conv2d if dims == 2 else F.conv3d)(xin, w, b, padding=1)
    xd = x.to(DEV)
    out = torch.empty(*xd.shape, 16, device=DEV, dtype=dtype)
    stats = ops.stats_buffer(16, DEV)
    c0 = torch.randn(16, generator=g)
    ops.conv_c1_fwd(xd, w.to(DEV), b.to(DEV), out, dims=dims, stats=stats, stats_shift=c0.to(DEV))
    got = uncl(out)
    if dims == 2:
        got = got.squeeze(2)
    assert relerr(got, ref) < (1e-5 if dtype == torch.float32 else 1e-2)
    red = [0] + list(range(2, ref.dim()))
    rc = ref - c0.view([1, -1] + [1] * (ref.dim() - 2))
    st = ops.stats_totals(stats, 16).float().cpu()
    assert relerr(st[0], rc.sum(red)) < 1e-3
    assert relerr(st[1], (rc * rc).sum(red)) < 1e-3


@pytest.mark.parametrize("shape", [(12, 1, 256, 256), (3, 1, 37, 90), (2, 56, 112, 80), (1, 9, 21, 70), (1300, 1, 8, 16)])
def test_first_conv_direct_equals_padded_path(shape):
    """bf16 first conv on its own kernel (csrc/conv_c1_mfma.h: the taps are the K dimension of one MFMA per 16 pixels) against the path it replaces --
    the image zero-padded to 16 channels through the generic conv: the same bf16 products in another summation order.  Full-size 2D batch (blocks walk
    several tiles), ragged 2D, the LA patch size (W = 80: five 16-pixel tiles per row), ragged 3D, more tiles than statistics slots."""
    g = torch.Generator().manual_seed(61)
    N, D, H, W = shape
    dims = 3 if D > 1 else 2
    dtype = torch.bfloat16
    x = torch.randn(N, D, H, W, generator=g).to(DEV)
    taps = 3 ** dims
    w = (torch.randn(16, 1, *([3] * dims), generator=g) / taps ** 0.5).to(DEV)
    b, c0 = torch.randn(16, generator=g).to(DEV), torch.randn(16, generator=g).to(DEV)
    out = torch.full((N, D, H, W, 16), float("nan"), device=DEV, dtype=dtype)
    stats = ops.stats_buffer(16, DEV)
    ops.conv_c1_fwd(x, w, b, out, dims=dims, stats=stats, stats_shift=c0)
    xpad = torch.empty(N, D, H, W, 16, device=DEV, dtype=dtype)
    ops.planar_to_cl(x.view(N, 1, D, H, W) if dims == 3 else x.view(N, 1, H, W), xpad, cpad=16)
    wp = ops.pack_weights(torch.cat((w, torch.zeros(16, 15, *([3] * dims), device=DEV)), 1), L.PACK_CONV_FWD, dtype, 16, 16, taps)
    ref = torch.empty_like(out)
    stats_ref = ops.stats_buffer(16, DEV)
    ops.conv_fwd([ops.Lazy(xpad)], wp, b, 16, ref, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=dims, stats=stats_ref, stats_shift=c0)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    # equal except where the two fp32 sums (same products, another order) rounded to neighbouring bf16 values: rare, and one place apart
    diff, mag = (out.float() - ref.float()).abs(), ref.float().abs()
    assert (diff <= 2.0 ** -7 * mag + 1e-6).all(), (diff / (mag + 1e-6)).max()
    assert (diff > 0).float().mean().item() < 1e-3
    a, r = ops.stats_totals(stats, 16), ops.stats_totals(stats_ref, 16)
    assert relerr(a.float(), r.float()) < 1e-4
    again = torch.empty_like(out)
    ops.conv_c1_fwd(x, w, b, again, dims=dims, stats=ops.stats_buffer(16, DEV), stats_shift=c0)
    assert torch.equal(again, out)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool_upsample_bnfinalize(dtype):
    g = torch.Generator().manual_seed(7)
    N, C, H, W = 2, 32, 12, 16
    x = rq(torch.randn(N, C, H, W, generator=g), dtype)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    a = lazy_ref(x, sc, sh, 0.01, None, 1.0)
    ref = F.max_pool2d(a, 2)
    lz = ops.Lazy(cl(x, dtype), sc.to(DEV), sh.to(DEV), True, 0.01)
    out = torch.empty(N, 1, H // 2, W // 2, C, device=DEV, dtype=dtype)
    idx = torch.empty(N, 1, H // 2, W // 2, C, device=DEV, dtype=torch.uint8)
    ops.act_pool2(lz, out, idx)
    assert relerr(uncl(out).squeeze(2), ref) < (1e-6 if dtype == torch.float32 else 1e-2)
    # bilinear x2 align_corners=True of a plain tensor into a concat slot
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    up = torch.zeros(N, 1, 2 * H, 2 * W, 2 * C, device=DEV, dtype=dtype)
    ops.upsample2x(ops.Lazy(cl(x, dtype)), up, dims=2, out_coff=C)
    assert relerr(uncl(up[..., C:]).squeeze(2), ref) < (1e-5 if dtype == torch.float32 else 1e-2)
    # adjoint
    gy = rq(torch.randn(N, C, 2 * H, 2 * W, generator=g), dtype)
    xr = x.clone().requires_grad_(True)
    F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True).backward(gy)
    gcat = torch.zeros(N, 1, 2 * H, 2 * W, 2 * C, device=DEV, dtype=dtype)
    gcat[..., C:] = cl(gy, dtype)
    dxo = torch.empty(N, 1, H, W, C, device=DEV, dtype=dtype)
    ops.upsample2x_bwd(gcat, C, C, dxo, dims=2)
    assert relerr(uncl(dxo).squeeze(2), xr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)
    # trilinear of a lazy ReLU'd tensor
    x3 = rq(torch.randn(1, 16, 3, 5, 6, generator=g), dtype)
    sc3, sh3 = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.2
    ref = F.interpolate(lazy_ref(x3, sc3, sh3, 0.0, None, 1.0), scale_factor=2, mode="trilinear", align_corners=True)
    up = torch.empty(1, 6, 10, 12, 16, device=DEV, dtype=dtype)
    ops.upsample2x(ops.Lazy(cl(x3, dtype), sc3.to(DEV), sh3.to(DEV), True, 0.0), up, dims=3)
    assert relerr(uncl(up), ref) < (1e-5 if dtype == torch.float32 else 1e-2)
    gy = rq(torch.randn(1, 16, 6, 10, 12, generator=g), dtype)
    xr = x3.clone().requires_grad_(True)
    F.interpolate(xr, scale_factor=2, mode="trilinear", align_corners=True).backward(gy)
    dxo = torch.empty(1, 3, 5, 6, 16, device=DEV, dtype=dtype)
    ops.upsample2x_bwd(cl(gy, dtype), 0, 16, dxo, dims=3)
    assert relerr(uncl(dxo), xr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)
    # BN finalize vs F.batch_norm bookkeeping
    y = torch.randn(4, C, 6, 6, generator=g) * 2 + 1
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(y, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    c0 = rm.clone()                                          # moments about the (old) running mean, as the engine does
    yc = y - c0.view(1, C, 1, 1)
    stats = ops.stats_from_moments(yc.sum((0, 2, 3)).to(DEV), (yc * yc).sum((0, 2, 3)).to(DEV))
    scale, shift = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    mean, invstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rmd, rvd, nbt = rm.to(DEV), rv.to(DEV), torch.zeros((), dtype=torch.long, device=DEV)
    ops.bn_finalize(stats, gamma.to(DEV), beta.to(DEV), rmd, rvd, nbt, 4 * 36, 1e-5, 0.1, scale, shift, mean, invstd, stats_shift=c0.to(DEV))
    got = y * scale.cpu().view(1, C, 1, 1) + shift.cpu().view(1, C, 1, 1)
    assert relerr(got, ref) < 1e-5
    assert relerr(rmd, rm_ref) < 1e-5 and relerr(rvd, rv_ref) < 1e-5 and int(nbt) == 1
    # |mean| >> std: E[x^2] - E[x]^2 about 0 loses the variance in fp32, the shifted moments do not (F.batch_norm: Welford)
    y2 = torch.randn(4, C, 6, 6, generator=g) * 1e-2 + 300.0
    ref2 = F.batch_norm(y2, None, None, gamma, beta, True, 0.1, 1e-5)
    c2 = y2.mean((0, 2, 3)) + 0.05                           # a shift NEAR the mean (what a running mean is)
    yc = (y2 - c2.view(1, C, 1, 1)).to(DEV)                  # fp32 sums on the device, like the conv epilogue
    stats = ops.stats_from_moments(yc.sum((0, 2, 3)), (yc * yc).sum((0, 2, 3)))
    ops.bn_finalize(stats, gamma.to(DEV), beta.to(DEV), None, None, None, 4 * 36, 1e-5, 0.0, scale, shift, mean, invstd, stats_shift=c2.to(DEV))
    got2 = y2 * scale.cpu().view(1, C, 1, 1) + shift.cpu().view(1, C, 1, 1)
    assert (got2 - ref2).abs().max() < 5e-2 * ref2.abs().max()      # limited by fp32 (y * scale + shift) at |y| = 300, not by the variance
    assert relerr(invstd, (y2.double().var((0, 2, 3), unbiased=False) + 1e-5).rsqrt().float()) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,c,add2", [((3, 10, 14, 14), 128, False), ((2, 9, 13, 20), 32, False), ((2, 12, 18, 30), 16, False),
                                          ((3, 10, 14, 14), 64, True), ((2, 11, 15, 34), 32, True)])
def test_conv3d_bricks_ragged(dtype, shape, c, add2):
    """3x3x3 conv on grids that are NOT multiples of the 4 x 4 x 16 brick (the V-Net's 14x14x10 / 7x7x5 levels), large
    enough (>= 32 bricks) to take the z-per-wave brick kernels (bf16) -- lazy BN/ReLU source, Dropout3d channel
    multipliers, optional skip add, bias, BatchNorm statistics."""
    g = torch.Generator().manual_seed(11)
    N, D, H, W = shape
    x0 = rq(torch.randn(N, c, D, H, W, generator=g), dtype)
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    cm = (torch.rand(N, c, generator=g) > 0.3).float() * 1.5
    w = torch.randn(c, c, 3, 3, 3, generator=g) / (c * 27) ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    a = lazy_ref(x0, sc, sh, 0.0, None, 1.0) * cm.view(N, c, 1, 1, 1)
    srcs = [ops.Lazy(cl(x0, dtype), sc.to(DEV), sh.to(DEV), True, 0.0, chan_mul=cm.to(DEV))]
    if add2:
        x1 = rq(torch.randn(N, c, D, H, W, generator=g), dtype)
        a = a + x1
        srcs.append(ops.Lazy(cl(x1, dtype)))
    ref = F.conv3d(rq(a, dtype), rq(w, dtype), b, padding=1)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, c, c, 27)
    out = torch.empty(N, D, H, W, c, device=DEV, dtype=dtype)
    stats = ops.stats_buffer(c, DEV)
    ops.conv_fwd(srcs, wp, b.to(DEV), c, out, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=3,
                 combine=1 if add2 else 0, stats=stats)
    assert relerr(uncl(out), ref) < 2 * TOL[dtype]
    st = ops.stats_totals(stats, c).float().cpu()
    assert relerr(st[0], ref.sum((0, 2, 3, 4))) < 1e-3 + TOL[dtype]
    assert relerr(st[1], (ref * ref).sum((0, 2, 3, 4))) < 1e-3 + TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,hw,cin,cout", [(4, (250, 254), 16, 16), (12, (62, 66), 64, 64), (12, (30, 34), 128, 128), (6, (126, 130), 32, 16)])
def test_conv3x3_2d_bench_shapes_ragged(dtype, N, hw, cin, cout):
    """The blockings the bench actually runs (16x16-pixel tiles with resident weights, 8x16 tiles x 32 channels with
    staged weights, ...) on grids that are NOT multiples of the tile: lazy BN/LeakyReLU source with a Dropout keep mask,
    bias, BatchNorm statistics."""
    g = torch.Generator().manual_seed(31)
    H, W = hw
    x = rq(torch.randn(N, cin, H, W, generator=g), dtype)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.2
    keep = (torch.rand(N, cin, H, W, generator=g) > 0.2).float()
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    a = lazy_ref(x, sc, sh, 0.01, keep, 1.25)
    ref = F.conv2d(rq(a, dtype), rq(w, dtype), b, padding=1)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, cin, cout, 9)
    out = torch.empty(N, 1, H, W, cout, device=DEV, dtype=dtype)
    stats = ops.stats_buffer(cout, DEV)
    km = keep.permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(DEV, torch.uint8)
    src = ops.Lazy(cl(x, dtype), sc.to(DEV), sh.to(DEV), True, 0.01, keep=km, keep_scale=1.25)
    ops.conv_fwd([src], wp, b.to(DEV), cout, out, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2, stats=stats)
    assert relerr(uncl(out).squeeze(2), ref) < 2 * TOL[dtype]
    st = ops.stats_totals(stats, cout).float().cpu()
    # bitwise reproducible: no atomics on the way (a second launch gives the same slots)
    first = stats.clone()
    ops.conv_fwd([src], wp, b.to(DEV), cout, out, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2, stats=stats)
    nslots = int(first[:1].view(torch.int32).item())
    assert torch.equal(first[:L.STATS_HDR + nslots * 2 * cout].view(torch.int32), stats[:L.STATS_HDR + nslots * 2 * cout].view(torch.int32))
    assert relerr(st[0], ref.sum((0, 2, 3))) < 1e-3 + TOL[dtype]
    assert relerr(st[1], (ref * ref).sum((0, 2, 3))) < 1e-3 + TOL[dtype]


@pytest.mark.parametrize("dims,shape,c0,c1,cout", [
    (2, (3, 1, 30, 34), 128, 0, 128),      # 4 chunks side by side, ragged 2D grid
    (2, (2, 1, 16, 16), 256, 0, 256),      # two rounds of 4 chunks
    (2, (2, 1, 17, 40), 64, 64, 64),       # two concatenated sources (decoder), 4 chunks
    (2, (2, 1, 20, 24), 64, 0, 100),       # 2 chunks side by side, Cout not a multiple of 32
    (3, (2, 10, 14, 14), 128, 0, 128),     # 3D slabs, 4 chunks
    (3, (1, 5, 7, 7), 256, 0, 128),        # 3D, two rounds
    (3, (2, 9, 13, 20), 64, 0, 64),        # 3D, 2 chunks side by side
])
def test_conv3x3_k_parallel(dims, shape, c0, c1, cout, monkeypatch):
    """The deep, small layers' kernel (csrc/conv_kpar.h: the K-chunks of a tile side by side, partial sums joined in LDS in a
    fixed order), forced through the library's knob: lazy BN-affine + LeakyReLU sources, Dropout keep mask (2D) / Dropout3d
    channel multipliers (3D), bias, shifted statistics -- against torch, and against conv_fwd_kernel on the same inputs."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(7)
    N, D, H, W = shape
    sp = (H, W) if dims == 2 else (D, H, W)
    cin = c0 + c1
    xs, lazies, refs = [], [], []
    for ci, c in enumerate([c0, c1]):
        if c == 0:
            continue
        x = rq(torch.randn(N, c, *sp, generator=g), dtype)
        sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
        keep = cm = None
        if ci == 0 and dims == 2:
            keep = (torch.rand(N, c, *sp, generator=g) > 0.3).float()
        if ci == 0 and dims == 3:
            cm = (torch.rand(N, c, generator=g) > 0.3).float() * 1.5
        ref = lazy_ref(x, sc, sh, 0.01, keep, 1.0 / 0.7)
        if cm is not None:
            ref = ref * cm.view(N, c, 1, 1, 1)
        refs.append(rq(ref, dtype))
        kd = None if keep is None else cl(keep, torch.uint8)
        lazies.append(ops.Lazy(cl(x, dtype), sc.to(DEV), sh.to(DEV), True, 0.01, keep=kd, keep_scale=1.0 / 0.7,
                               chan_mul=None if cm is None else cm.to(DEV)))
    a = torch.cat(refs, 1)
    w = torch.randn(cout, cin, *([3] * dims), generator=g) / (cin * 3 ** dims) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = (F.conv2d if dims == 2 else F.conv3d)(a, rq(w, dtype), b, padding=1)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, cin, cout, 3 ** dims)
    cshift = torch.randn(cout, generator=g)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CHAP_CONV_KPAR", mode)
        out = torch.full((N, D, H, W, cout), float("nan"), device=DEV, dtype=dtype)
        stats = ops.stats_buffer(cout, DEV)
        ops.conv_fwd(lazies, wp, b.to(DEV), cout, out, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=dims,
                     stats=stats, stats_shift=cshift.to(DEV))
        torch.cuda.synchronize()
        got[mode] = (uncl(out), ops.stats_totals(stats, cout).float().cpu())
    o1, s1 = got["1"]
    refc = ref if dims == 3 else ref.unsqueeze(2)
    assert torch.isfinite(o1).all()
    assert relerr(o1, refc) < TOL[dtype]
    assert relerr(o1, got["0"][0]) < 1e-2                        # both round the same fp32 sums to bf16: a last-bit difference at most
    rc = refc - cshift.view(1, -1, 1, 1, 1)
    assert relerr(s1[0], rc.sum((0, 2, 3, 4))) < 1e-3 + TOL[dtype]
    assert relerr(s1[1], (rc * rc).sum((0, 2, 3, 4))) < 1e-3 + TOL[dtype]


def test_group_region_contract():
    """Grouped launches at the C-ABI level: two same-shaped convs + finalizes recorded in two lanes are issued as 2 grids (instead of 4) and give
    the bits of the separate launches; lanes of different shapes are issued one after the other; an entry point whose kernel is launched
    directly (chap_sgd_step) is refused inside a region."""
    from chap_amd import _lib as L
    from chap_amd import ops
    g = torch.Generator().manual_seed(9)
    st = torch.cuda.current_stream().cuda_stream

    def layer(C, H):
        x = torch.randn(2, 1, H, H, C, generator=g).to(DEV).to(torch.bfloat16)
        w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
        return x, ops.pack_weights(w, L.PACK_CONV_FWD, torch.bfloat16, C, C, 9), torch.ones(C, device=DEV), torch.zeros(C, device=DEV)

    def run(x, wp, gam, bet):
        C, H = x.shape[-1], x.shape[2]
        out, stats = torch.empty_like(x), ops.stats_buffer(C, DEV)
        scale, shift = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        ops.conv_fwd([ops.Lazy(x)], wp, None, C, out, grid=(2, 1, H, H), in_dims=(1, H, H), ksize=3, stride=1, dims=2, stats=stats)
        ops.bn_finalize(stats, gam, bet, None, None, None, 2 * H * H, 1e-5, 0.0, scale, shift)
        return out, scale, shift

    a, b, c = layer(16, 48), layer(16, 48), layer(32, 24)
    ref = [run(*a), run(*b), run(*c)]
    n0 = L.group.launched
    with L.group(st) as region:
        got_a = run(*a)
        region.next_lane()
        got_b = run(*b)
    assert L.group.launched - n0 == 2                       # conv x 2 lanes, finalize x 2 lanes
    n0 = L.group.launched
    with L.group(st) as region:
        got_a2 = run(*a)
        region.next_lane()
        got_c = run(*c)
    assert L.group.launched - n0 == 4                       # different shapes: nothing to merge
    torch.cuda.synchronize()
    for r, q in ((ref[0], got_a), (ref[1], got_b), (ref[0], got_a2), (ref[2], got_c)):
        assert all(torch.equal(u, v) for u, v in zip(r, q))
    p, m, lr = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV), torch.full((1,), 0.1, device=DEV)
    with pytest.raises(L.ChapError, match="chap_group"):
        with L.group(st):
            ops.sgd_step(p, torch.ones(64, device=DEV), m, lr, 0.9, 0.0)
    assert L.group.held is None                             # the region was left


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,half,hw", [(16, 16, (37, 52)), (64, 64, (20, 24)), (32, 32, (16, 40))])
def test_conv_two_dense_outputs(dtype, cin, half, hw):
    """chap_conv_params.out2 (ABI 7): the input gradient of a concat layer (torch.cat((skip, up), 1), unet.py:98) written as two dense tensors, one per
    source -- bit for bit the two channel halves of the single [.., 2 * half] output, also where one block's channel tiles straddle the split (16 + 16)."""
    g = torch.Generator().manual_seed(41)
    N, (H, W) = 2, hw
    cout = 2 * half
    x = cl(rq(torch.randn(N, cin, H, W, generator=g), dtype), dtype)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, cin, cout, 9)
    one = torch.full((N, 1, H, W, cout), float("nan"), device=DEV, dtype=dtype)
    ops.conv_fwd([ops.Lazy(x)], wp, None, cout, one, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2)
    d0 = torch.full((N, 1, H, W, half), float("nan"), device=DEV, dtype=dtype)
    d1 = torch.full((N, 1, H, W, half), float("nan"), device=DEV, dtype=dtype)
    ops.conv_fwd([ops.Lazy(x)], wp, None, cout, d0, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2, out2=d1)
    torch.cuda.synchronize()
    assert torch.isfinite(one.float()).all()
    assert torch.equal(d0, one[..., :half].contiguous()) and torch.equal(d1, one[..., half:].contiguous())


@pytest.mark.parametrize("cins,cout,hw,keep,split", [([16], 16, (37, 52), True, False), ([16], 32, (40, 33), False, True), ([32], 32, (24, 40), False, False),
                                                     ([16, 16], 16, (19, 50), True, False), ([32], 64, (21, 36), False, False), ([32], 16, (9, 70), False, False)])
def test_conv_wave_private_2d(cins, cout, hw, keep, split, monkeypatch):
    """The wave-private convolution kernel of the 2D full-resolution layers (csrc/conv_wp.h; bf16, all input channels in one chunk of 16 or 32), forced
    onto small RAGGED grids: lazy BatchNorm-affine + LeakyReLU sources (one, or two concatenated inside the chunk), element keep mask, bias, shifted
    statistics, the two-tensor output of a concat layer's input gradient -- outputs BIT-IDENTICAL to conv_fwd_kernel (the same MFMA sequence per output
    element), statistics totals equal to fp32 rounding (another dealing of pixels to slots), both against torch."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(53)
    N, (H, W) = 3, hw
    cin = sum(cins)
    lazies, refs = [], []
    for i, c in enumerate(cins):
        x = rq(torch.randn(N, c, H, W, generator=g), dtype)
        sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
        km = (torch.rand(N, c, H, W, generator=g) > 0.3).float() if (keep and i == 0) else None
        refs.append(rq(lazy_ref(x, sc, sh, 0.01, km, 1.25 if km is not None else 1.0), dtype))
        lazies.append(ops.Lazy(cl(x, dtype), sc.to(DEV), sh.to(DEV), True, 0.01, keep=None if km is None else cl(km, torch.uint8), keep_scale=1.25 if km is not None else 1.0))
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(torch.cat(refs, 1), rq(w, dtype), b, padding=1)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dtype, cin, cout, 9)
    cshift = torch.randn(cout, generator=g)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CHAP_CONV_WP", mode)
        stats = ops.stats_buffer(cout, DEV)
        if split:
            o0 = torch.full((N, 1, H, W, cout // 2), float("nan"), device=DEV, dtype=dtype)
            o1 = torch.full_like(o0, float("nan"))
            ops.conv_fwd(lazies, wp, b.to(DEV), cout, o0, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2, stats=stats, stats_shift=cshift.to(DEV), out2=o1)
            out = torch.cat((o0, o1), -1)
        else:
            out = torch.full((N, 1, H, W, cout), float("nan"), device=DEV, dtype=dtype)
            ops.conv_fwd(lazies, wp, b.to(DEV), cout, out, grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2, stats=stats, stats_shift=cshift.to(DEV))
        torch.cuda.synchronize()
        got[mode] = (out.clone(), ops.stats_totals(stats, cout).float().cpu())
    assert torch.isfinite(got["1"][0].float()).all()
    assert torch.equal(got["1"][0], got["0"][0])
    assert relerr(uncl(got["1"][0]), ref.unsqueeze(2)) < TOL[dtype]
    assert relerr(got["1"][1], got["0"][1]) < 1e-4
    rc = ref - cshift.view(1, -1, 1, 1)
    assert relerr(got["1"][1][0], rc.sum((0, 2, 3))) < 1e-3 + TOL[dtype]
