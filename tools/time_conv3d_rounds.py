"""Does the 64->64 3x3x3 layer pay for a second ROUND of blocks?  Its z-brick kernel runs one block per CU (LDS); 28x28x20, N = 2 is 280 blocks on 256 CUs.
Times the same layer at depths that give 224 / 256 / 280 / 336 / 512 blocks.   python tools/time_conv3d_rounds.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from chap_amd import _lib as L, ops

dev = "cuda"


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def case(N, D, H, W, c):
    bf = torch.bfloat16
    x = torch.randn(N, D, H, W, c, device=dev).to(bf)
    out = torch.empty_like(x)
    w = torch.randn(c, c, 3, 3, 3, device=dev) / (27 * c) ** 0.5
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, bf, c, c, 27)
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    st = ops.stats_buffer(c, dev)
    us = timeit(lambda: ops.conv_fwd([ops.Lazy(x, sc, sh, True, 0.01)], wp, None, c, out, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=3, stats=st))
    bricks = N * ((D + 3) // 4) * ((H + 3) // 4) * ((W + 15) // 16)
    fl = 2.0 * N * D * H * W * 27 * c * c
    print("conv3d %d->%d @%dx%dx%d N=%d: %3d bricks  %6.1f us  %6.1f TFLOP/s  (%.2f us per brick-round of 256)" % (c, c, D, H, W, N, bricks, us, fl / us / 1e6, us / max(1, -(-bricks * (c // 32) // 256))))


if __name__ == "__main__":
    for D in (16, 18, 20, 24, 36):
        case(2, D, 28, 28, 64)
    for (D, H, W) in ((10, 14, 14), (12, 16, 16)):
        case(2, D, H, W, 128)
    for D in (28, 32, 40, 48):
        case(2, D, 56, 56, 32)
