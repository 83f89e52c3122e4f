"""Data-parallel gradient exchange for the CHAP iteration (one process per GPU, RCCL over xGMI).

The reference has no distributed code; this is the build's own design:
each rank runs the whole iteration on its own (labeled + unlabeled) shard -- BatchNorm statistics
and Dice sums stay per replica (DDP semantics) -- and the ONLY exchange is a sum of the gradients.

The two gradient buckets of an iteration (bucket 0: BCP / mix_loss backward, bucket 1: VAT final
backward; they are produced concurrently on two streams) are the two halves of ONE contiguous fp32
buffer.  Only their SUM matters to the optimizer, so the exchange first folds bucket 1 into bucket 0 (one
axpy kernel + a memset) and all-reduces that half alone: 10.3 MB (2D) / 49.4 MB (3D) per step instead of
twice that -- the ring all-reduce is per-link bound on the xGMI mesh, so bytes are what it costs.  One flat
collective instead of per-tensor ones for the same reason.  The fused SGD kernel then consumes
(bucket0 + bucket1) * (1/world) with bucket 1 all zero.  With HIP-graph replay the iteration is two graphs
(compute, optimizer) with the fold and the collective in between, so RCCL never has to be captured.
"""
import torch


class DataParallelSync:
    def __init__(self, both_buckets, dist, group=None):
        self.buf, self.dist, self.group = both_buckets, dist, group
        self.work = None

    def start(self):
        """All gradients have been enqueued on the current stream: fold bucket 1 into bucket 0 and start the
        (asynchronous) all-reduce of bucket 0."""
        n = self.buf.numel() // 2
        b0, b1 = self.buf[:n], self.buf[n:]
        if b0.is_cuda:
            from . import ops
            ops.perturb(b0, b1, b0, 1.0)            # b0 += b1 (chap_perturb: out = x + alpha * d)
        else:
            b0.add_(b1)                             # gloo rehearsal on CPU tensors (tests/test_parallel_cpu.py)
        b1.zero_()
        self.work = self.dist.all_reduce(b0, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
