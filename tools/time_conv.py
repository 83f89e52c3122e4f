"""Time single kernels on real layer shapes (HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from chap_amd import _lib as L, ops

dev = "cuda"

def timeit(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

def conv_case(N, H, cin, cout, dtype, stats, prologue, ks=3):
    x = torch.randn(N, 1, H, H, cin, device=dev).to(dtype)
    out = torch.empty(N, 1, H, H, cout, device=dev, dtype=dtype)
    w = torch.randn(cout, cin, ks, ks, device=dev) / 12
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dtype, cin, cout, ks * ks)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    st = ops.stats_buffer(cout, dev) if stats else None
    src = ops.Lazy(x, sc, sh, True, 0.01) if prologue else ops.Lazy(x)
    us = timeit(lambda: ops.conv_fwd([src], wp, None, cout, out, grid=(N, 1, H, H), in_dims=(1, H, H), ksize=ks, stride=1, dims=2, stats=st))
    esz = 2 if dtype == torch.bfloat16 else 4
    by = N * H * H * (cin + cout) * esz
    fl = 2.0 * N * H * H * cin * cout * ks * ks
    print("conv%dx%d %3d->%3d @%3d N=%d %s stats=%d prologue=%d: %8.1f us  %7.1f GB/s  %6.1f TFLOP/s" % (ks, ks, cin, cout, H, N, "bf16" if esz == 2 else "f32", stats, prologue, us, by / us / 1e3, fl / us / 1e6))

def first_case(N, sp):
    """first conv, bf16: direct kernel (conv_c1_mfma.h) against planar_to_cl + generic conv over the zero-padded image"""
    D, H, W = sp
    dims = 3 if D > 1 else 2
    bf = torch.bfloat16
    x = torch.randn(N, D, H, W, device=dev)
    w = torch.randn(16, 1, *([3] * dims), device=dev) / 5
    b = torch.randn(16, device=dev)
    out = torch.empty(N, D, H, W, 16, device=dev, dtype=bf)
    st = ops.stats_buffer(16, dev)
    us_direct = timeit(lambda: ops.conv_c1_fwd(x, w, b, out, dims=dims, stats=st))
    xpad = torch.empty(N, D, H, W, 16, device=dev, dtype=bf)
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, bf, 1, 16, 3 ** dims)
    xv = x.view(N, 1, D, H, W) if dims == 3 else x.view(N, 1, H, W)
    us_pad = timeit(lambda: ops.planar_to_cl(xv, xpad, cpad=16))
    us_conv = timeit(lambda: ops.conv_fwd([ops.Lazy(xpad)], wp, b, 16, out, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=dims, stats=st))
    by = N * D * H * W * (4 + 32)
    print("first conv 1->16 @%s N=%d bf16: direct %6.1f us (%6.1f GB/s of its 36 B per pixel)   padded path: planar_to_cl %5.1f + conv %5.1f us" % (
        "x".join(map(str, sp)), N, us_direct, by / us_direct / 1e3, us_pad, us_conv))


if __name__ == "__main__":
    bf = torch.bfloat16
    import sys as _sys
    if len(_sys.argv) > 1 and _sys.argv[1] == "first":
        for N, sp in ((12, (1, 256, 256)), (24, (1, 256, 256)), (2, (112, 112, 80)), (4, (112, 112, 80))):
            first_case(N, sp)
        raise SystemExit(0)
    if len(_sys.argv) > 1 and _sys.argv[1] == "shallow":     # the full-resolution 2D layers (CHAP_CONV_WP=0 / 1: conv_fwd_kernel / conv_wp_kernel)
        for (H, ci, co, st, pro) in ((256, 16, 16, 1, 1), (256, 16, 16, 0, 0), (256, 16, 32, 0, 0), (256, 32, 16, 1, 1), (128, 32, 32, 1, 1), (128, 32, 32, 0, 0), (128, 32, 64, 0, 0), (128, 16, 32, 1, 0)):
            conv_case(12, H, ci, co, bf, st, pro)
        raise SystemExit(0)
    if len(_sys.argv) > 1 and _sys.argv[1] == "deep":        # the deep 2D layers (CHAP_CONV_KPAR=0 / 1: conv_fwd_kernel / conv_kpar2d_kernel)
        for (H, ci, co) in ((64, 64, 64), (64, 64, 128), (32, 128, 128), (32, 128, 256), (16, 256, 256), (32, 256, 128), (64, 128, 64)):
            conv_case(12, H, ci, co, bf, 1, 1)
        raise SystemExit(0)
    for stats in (0, 1):
        for pro in (0, 1):
            conv_case(12, 256, 16, 16, bf, stats, pro)
    conv_case(12, 128, 32, 32, bf, 1, 1)
    conv_case(12, 64, 64, 64, bf, 1, 1)
    conv_case(12, 32, 128, 128, bf, 1, 1)
    conv_case(12, 16, 256, 256, bf, 1, 1)
    conv_case(12, 256, 32, 16, bf, 1, 1)
    conv_case(12, 128, 64, 32, bf, 1, 1)
    conv_case(12, 16, 256, 128, bf, 0, 1, ks=1)
    conv_case(12, 256, 16, 16, torch.float32, 1, 1)
