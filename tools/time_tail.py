"""What the in-launch BatchNorm finalize (chap_conv_params.fin, csrc/tail.h) costs per launch against the two-launch form, on real layer shapes,
as captured graphs of REPS dependent repetitions (each conv reads the previous conv's output through the finalized affine: the chain of a network).
    python tools/time_tail.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from chap_amd import _lib as L, ops

dev = "cuda"
REPS = 20


def chain(N, sp, c, dims, fused):
    D, H, W = sp
    dtype = torch.bfloat16
    xs = [torch.randn(N, D, H, W, c, device=dev).to(dtype) for _ in range(2)]
    w = torch.randn(*([c, c] + [3] * dims), device=dev) / (c * 3 ** dims) ** 0.5
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dtype, c, c, 3 ** dims)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    affs = [torch.zeros(4, c, device=dev) for _ in range(2)]
    affs[1][0].fill_(1.0)
    stats = [ops.stats_buffer(c, dev) for _ in range(2)]
    tickets = torch.zeros((REPS + 4) * L.TAIL_TICKETS, dtype=torch.int32, device=dev)
    rows = torch.empty(ops.tail_rows_size(2 * c), dtype=torch.float64, device=dev)
    cnt = N * D * H * W

    def body(reps):
        for i in range(reps):
            a, b = i & 1, (i + 1) & 1
            src = ops.Lazy(xs[a], affs[b][0], affs[b][1], True, 0.01)           # the affine the previous repetition finalized
            fin = None
            if fused:
                fin = ops.BnFinalize(tickets[i * L.TAIL_TICKETS:(i + 1) * L.TAIL_TICKETS], rows, gamma, beta, None, None, None, cnt, 1e-5, 0.0, affs[a])
            ops.conv_fwd([src], wp, None, c, xs[b], grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=dims, stats=stats[a], fin=fin)
            if not fused:
                ops.bn_finalize(stats[a], gamma, beta, None, None, None, cnt, 1e-5, 0.0, affs[a][0], affs[a][1], affs[a][2], affs[a][3])

    body(2)
    torch.cuda.synchronize()
    tickets.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(REPS)
    best = 1e9
    for _ in range(5):
        tickets.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


if __name__ == "__main__":
    for (N, sp, c, dims) in ((12, (1, 256, 256), 16, 2), (12, (1, 128, 128), 32, 2), (12, (1, 64, 64), 64, 2), (12, (1, 32, 32), 128, 2), (12, (1, 16, 16), 256, 2),
                             (2, (80, 112, 112), 16, 3), (2, (20, 28, 28), 64, 3), (2, (10, 14, 14), 128, 3)):
        two, one = chain(N, sp, c, dims, False), chain(N, sp, c, dims, True)
        print("conv %d->%d @%s N=%d: conv + bn_finalize %.1f us per layer, conv with fin %.1f us" % (c, c, "x".join(map(str, sp)), N, two, one))
