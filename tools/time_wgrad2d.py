"""Time chap_wgrad (+ slab reduction) on the 2D layer shapes of BASELINE config 1 (HIP events on the launch stream), one or two A sources.
   CHAP_WGRAD_WP=0 python tools/time_wgrad2d.py     # block-tile kernel
   python tools/time_wgrad2d.py                      # default rule (wave-private pipelines on the large images)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chap_amd import ops
from tools.time_conv import timeit

dev = "cuda"


def case(tag, N, H, W, cas, cb, keep=False):
    dtype = torch.bfloat16
    srcs = []
    for i, ca in enumerate(cas):
        a = torch.randn(N, 1, H, W, ca, device=dev).to(dtype)
        if i == 0:
            k = (torch.rand(N, 1, H, W, ca, device=dev) > 0.1).to(torch.uint8) if keep else None
            srcs.append(ops.Lazy(a, torch.rand(ca, device=dev) + 0.5, torch.randn(ca, device=dev) * 0.1, True, 0.01, keep=k, keep_scale=1.1))
        else:
            srcs.append(ops.Lazy(a))
    g = torch.randn(N, 1, H, W, cb, device=dev).to(dtype)
    ca = sum(cas)
    dw = torch.zeros(cb, ca, 3, 3, device=dev)
    db = torch.zeros(cb, device=dev)
    us = timeit(lambda: ops.wgrad(srcs, ops.Lazy(g), dw, (1, 9, ca * 9), grid=(N, 1, H, W), in_dims=(1, H, W), ksize=3, stride=1, dims=2, db=db), reps=30)
    px = N * H * W
    print("%-30s %8.1f us   %6.1f GB/s (A+B once)" % (tag, us, px * (ca + cb) * 2 / us / 1e3), flush=True)


if __name__ == "__main__":
    N = int(os.environ.get("N", "12"))
    case("16->16 @256", N, 256, 256, [16], 16)
    case("16->16 @256 keep", N, 256, 256, [16], 16, keep=True)
    case("16+16->16 @256", N, 256, 256, [16, 16], 16)
    case("32->32 @128", N, 128, 128, [32], 32)
    case("32+32->32 @128", N, 128, 128, [32, 32], 32)
    case("16->32 @128 (Cb 32)", N, 128, 128, [16], 32)
    case("64->64 @64", N, 64, 64, [64], 64)
    case("64+64->64 @64", N, 64, 64, [64, 64], 64)
    case("128->128 @32", N, 32, 32, [128], 128)
    case("128+128->128 @32", N, 32, 32, [128, 128], 128)
    case("256->256 @16", N, 16, 16, [256], 256)
    case("64->128 @64", N, 64, 64, [64], 128)
