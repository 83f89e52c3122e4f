"""Data-parallel gradient exchange for the CHAP iteration (one process per GPU, RCCL over xGMI).

The reference has no distributed code; this is the build's own design:
each rank runs the whole iteration on its own (labeled + unlabeled) shard -- BatchNorm statistics
and Dice sums stay per replica (DDP semantics) -- and the ONLY exchange is a sum of the flat fp32
gradient buffer, split in two buckets so that it overlaps with compute:

  bucket 0  gradients of the BCP (mix_loss) backward, all-reduced asynchronously while the VAT
            power-iteration and final passes run (they read the not-yet-updated weights);
  bucket 1  gradients of the VAT final backward, all-reduced right after it.

The fused SGD kernel then consumes (bucket0 + bucket1) * (1/world).  Payloads are 10.3 MB (2D) /
49.4 MB (3D) of fp32, so the exchange is latency/per-link bound: one flat buffer per bucket, no
per-tensor collectives.
"""
import torch


class DataParallelSync:
    """bucket 0 = the model's own flat gradient buffer (BCP backward), bucket 1 = the second buffer the
    VAT branch accumulates into (ChapStep.grad2)."""

    def __init__(self, bucket0, bucket1, dist, group=None):
        self.dist, self.group = dist, group
        self.bucket = [bucket0, bucket1]
        self.work = [None, None]

    def bucket_ready(self, i):
        """All gradients of bucket i have been enqueued on the current stream: start its all-reduce (async:
        it overlaps whatever is enqueued next)."""
        self.work[i] = self.dist.all_reduce(self.bucket[i], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        for i in (0, 1):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None
