"""Time chap_wgrad (+ its slab reduction) on the real layer shapes (HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chap_amd import ops
from tools.time_conv import timeit

dev = "cuda"


def case(tag, N, D, H, W, ca, cb, ks=3, dims=3, dtype=torch.bfloat16):
    a = torch.randn(N, D, H, W, ca, device=dev).to(dtype)
    g = torch.randn(N, D, H, W, cb, device=dev).to(dtype)
    sc, sh = torch.rand(ca, device=dev) + 0.5, torch.randn(ca, device=dev) * 0.1
    taps = ks ** dims
    dw = torch.zeros(cb, ca, *([ks] * dims), device=dev)
    db = torch.zeros(cb, device=dev)
    us = timeit(lambda: ops.wgrad([ops.Lazy(a, sc, sh, True, 0.0)], ops.Lazy(g), dw, (1, taps, ca * taps), grid=(N, D, H, W), in_dims=(D, H, W),
                                  ksize=ks, stride=1, dims=dims, db=db), reps=20)
    px = N * D * H * W
    print("%-34s %8.1f us   %6.1f GB/s (A+B once)   %6.1f TFLOP/s" % (tag, us, px * (ca + cb) * 2 / us / 1e3, 2.0 * px * ca * cb * taps / us / 1e6), flush=True)


if __name__ == "__main__":
    for N in (2, 4):
        case("3D 16->16 @80x112x112 N%d" % N, N, 80, 112, 112, 16, 16)
        case("3D 32->32 @40x56x56 N%d" % N, N, 40, 56, 56, 32, 32)
        case("3D 64->64 @20x28x28 N%d" % N, N, 20, 28, 28, 64, 64)
        case("3D 128->128 @10x14x14 N%d" % N, N, 10, 14, 14, 128, 128)
        case("3D 256->256 @5x7x7 N%d" % N, N, 5, 7, 7, 256, 256)
    for N in (12, 24):
        case("2D 16->16 @256 N%d" % N, N, 1, 256, 256, 16, 16, dims=2)
        case("2D 32->32 @128 N%d" % N, N, 1, 128, 128, 32, 32, dims=2)
        case("2D 64->64 @64 N%d" % N, N, 1, 64, 64, 64, 64, dims=2)
        case("2D 128->128 @32 N%d" % N, N, 1, 32, 32, 128, 128, dims=2)
        case("2D 256->256 @16 N%d" % N, N, 1, 16, 16, 256, 256, dims=2)
