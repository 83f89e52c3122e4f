"""Time the 3D 3x3x3 conv kernel on the V-Net layer shapes (HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from chap_amd import _lib as L, ops
from tools.time_conv import timeit

dev = "cuda"

def conv3d_case(N, D, H, W, cin, cout, dtype=torch.bfloat16, stats=1, prologue=1):
    x = torch.randn(N, D, H, W, cin, device=dev).to(dtype)
    out = torch.empty(N, D, H, W, cout, device=dev, dtype=dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) / 20
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dtype, cin, cout, 27)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    st = ops.stats_buffer(cout, dev) if stats else None
    src = ops.Lazy(x, sc, sh, True, 0.0) if prologue else ops.Lazy(x)
    us = timeit(lambda: ops.conv_fwd([src], wp, None, cout, out, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=3, stats=st))
    esz = 2 if dtype == torch.bfloat16 else 4
    by = N * D * H * W * (cin + cout) * esz
    fl = 2.0 * N * D * H * W * cin * cout * 27
    print("conv3x3x3 %3d->%3d @%dx%dx%d N=%d: %8.1f us  %7.1f GB/s  %6.1f TFLOP/s" % (cin, cout, D, H, W, N, us, by / us / 1e3, fl / us / 1e6), flush=True)

if __name__ == "__main__":
    for N in (2, 4):
        conv3d_case(N, 80, 112, 112, 16, 16)
        conv3d_case(N, 40, 56, 56, 32, 32)
        conv3d_case(N, 20, 28, 28, 64, 64)
        conv3d_case(N, 10, 14, 14, 128, 128)
        conv3d_case(N, 5, 7, 7, 256, 256)
