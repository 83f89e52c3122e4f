"""Data-parallel gradient exchange for the CHAP iteration (one process per GPU, RCCL over xGMI).

The reference has no distributed code; this is the build's own design:
each rank runs the whole iteration on its own (labeled + unlabeled) shard -- BatchNorm statistics
and Dice sums stay per replica (DDP semantics) -- and the ONLY exchange is a sum of the gradients.

The two gradient buckets of an iteration (bucket 0: BCP / mix_loss backward, bucket 1: VAT final
backward; they are produced concurrently on two streams) are the two halves of ONE contiguous fp32
buffer, so the exchange is a single all-reduce of 2 x 10.3 MB (2D) / 2 x 49.4 MB (3D): latency / per-link
bound on the xGMI mesh, hence one flat collective instead of per-tensor ones.  The fused SGD kernel then
consumes (bucket0 + bucket1) * (1/world).  With HIP-graph replay the iteration is two graphs
(compute, optimizer) with the collective in between, so RCCL never has to be captured.
"""
import torch


class DataParallelSync:
    def __init__(self, both_buckets, dist, group=None):
        self.buf, self.dist, self.group = both_buckets, dist, group
        self.work = None

    def start(self):
        """All gradients have been enqueued on the current stream: start the (asynchronous) all-reduce."""
        self.work = self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
