"""Data-parallel gradient exchange for the CHAP iteration (one process per GPU, RCCL over xGMI).

The reference has no distributed code; this is the build's own design:
each rank runs the whole iteration on its own (labeled + unlabeled) shard -- BatchNorm statistics
and Dice sums stay per replica (DDP semantics) -- and the ONLY exchange is a sum of the gradients.

The two gradient buckets of an iteration (bucket 0: BCP / mix_loss backward = phase B, bucket 1: VAT final
backward (+ fp_loss) = phase V; they are produced concurrently on two streams) are the two halves of ONE
contiguous fp32 buffer, and only their SUM matters to the optimizer (the fused SGD consumes
(bucket0 + bucket1) * (1/world)).  Two schedules:

* overlap (`overlap=True`; what north_star names; EXPERIMENTAL until the two-rank GPU test has run on a >= 2 GPU machine --
  measured slower than fold on a 1-rank group): phase B finishes long before the VAT chain -- its bucket is all-reduced on
  phase B's stream as soon as it is final (`start_first`), BESIDE the VAT forward/backward passes; bucket 1 follows
  when the chain is done (`start`).  Exposed on the critical path: the all-reduce of bucket 1 only (10.3 MB 2D / 49.4 MB
  3D), with twice those bytes on the links in total.
* fold (the default, what bench.py and DESIGN.md use): bucket 1 is folded into bucket 0 (one axpy kernel + a memset) at the end and that half alone is
  all-reduced: the same exposed bytes, half the link traffic, nothing overlapped.

With HIP-graph replay RCCL is never captured: the iteration is [graph A] -> [graph B on the side stream | graph V] with
the eager collectives behind graph B / graph V, then [optimizer graph] (chap_amd.train.ChapStep.capture).
"""
import torch


BUCKET_BYTES = 16 << 20     # the 3D gradient buffer (49.4 MB) goes out in pieces of <= 16 MB; the 2D one (10.3 MB) is one piece


class DataParallelSync:
    """bucket_bytes: a gradient half is all-reduced in contiguous pieces of at most this many bytes, issued back to back on the collective's
    stream (a ring over xGMI is per-link bound: pieces of a few MB keep every link busy while the next piece is being set up, and a later
    schedule can start the first pieces before the last gradients are final).  The element-wise sum does not depend on the partition: any
    bucket size gives the same bits (tests/test_parallel_cpu.py, tests/test_parallel_gpu.py: 1-rank RCCL group == the plain step)."""

    def __init__(self, both_buckets, dist, group=None, overlap=False, bucket_bytes=BUCKET_BYTES):
        self.buf, self.dist, self.group, self.overlap = both_buckets, dist, group, overlap
        self.bucket_elems = max(1, int(bucket_bytes) // both_buckets.element_size())
        self.work, self.work0 = [], []

    def _halves(self):
        n = self.buf.numel() // 2
        return self.buf[:n], self.buf[n:]

    def pieces(self, t):
        """The contiguous <= bucket_bytes pieces of a flat gradient half, in order."""
        return [t[i:i + self.bucket_elems] for i in range(0, t.numel(), self.bucket_elems)]

    def _all_reduce(self, t):
        return [self.dist.all_reduce(p, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True) for p in self.pieces(t)]

    def start_first(self):
        """Bucket 0 is final on the CURRENT stream (phase B's): start its all-reduce now; it runs beside whatever the
        other streams still compute (the VAT chain).  No-op in the fold schedule."""
        if not self.overlap:
            return
        b0, _ = self._halves()
        self.work0 = self._all_reduce(b0)

    def start(self):
        """All gradients have been enqueued on the current stream: all-reduce what is still outstanding (bucket 1 when
        bucket 0 went ahead with start_first; otherwise fold bucket 1 into bucket 0 and all-reduce that half)."""
        b0, b1 = self._halves()
        if self.work0:
            self.work = self._all_reduce(b1)
            return
        if b0.is_cuda:
            from . import ops
            ops.perturb(b0, b1, b0, 1.0)            # b0 += b1 (chap_perturb: out = x + alpha * d)
        else:
            b0.add_(b1)                             # gloo rehearsal on CPU tensors (tests/test_parallel_cpu.py)
        b1.zero_()
        self.work = self._all_reduce(b0)

    def wait(self):
        """The current stream (the optimizer's) waits for the collectives."""
        for w in self.work0 + self.work:
            w.wait()
        self.work0, self.work = [], []


class HostStagedDist:
    """torch.distributed as DataParallelSync uses it (`all_reduce`, `ReduceOp`), with every tensor staged through host memory: lets
    several ranks share ONE GPU over a CPU backend (gloo) -- rehearsals of the multi-rank paths on a 1-GPU box (`bench.py --gpus N
    --rehearse-one-gpu`, tests/dp_worker.py --backend gloo).  Not a product path: no overlap, a host round trip per exchange."""

    def __init__(self, dist):
        self._dist, self.ReduceOp = dist, dist.ReduceOp

    class _Done:
        def wait(self):
            pass

    def all_reduce(self, t, op=None, group=None, async_op=False):
        c = t.detach().cpu()                      # (synchronises the current stream: the buffer is final)
        self._dist.all_reduce(c, op=self._dist.ReduceOp.SUM if op is None else op, group=group)
        t.copy_(c)
        return self._Done()
