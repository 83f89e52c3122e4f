"""World-size-2 gloo tests of the two-bucket gradient exchange (CPU, no GPU kernels involved), both schedules of
chap_amd.parallel.DataParallelSync: what the optimizer consumes, (bucket0 + bucket1) / world, equals the mean over ranks of
the per-shard gradients (DDP semantics)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out, overlap):
    sys.path.insert(0, ROOT)
    from chap_amd.parallel import DataParallelSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1003
    both = torch.zeros(2 * n)
    b0, b1 = both[:n], both[n:]         # bucket 0 = phase B (BCP) gradients, bucket 1 = phase V (VAT branch)
    sync = DataParallelSync(both, dist, overlap=overlap)
    g = torch.Generator().manual_seed(100 + rank)
    b0 += torch.randn(n, generator=g)   # "BCP backward" finishes first ...
    sync.start_first()                  # ... its all-reduce starts while the VAT branch still runs (overlap schedule)
    b1 += torch.randn(n, generator=g)   # "VAT backward" (concurrent branch)
    sync.start()
    sync.wait()
    if not overlap:
        assert float(b1.abs().max()) == 0.0   # bucket 1 was folded into bucket 0 before the exchange (half the bytes on the wire)
    total = (b0 + b1) / world           # what the fused SGD consumes: (grad + grad2) * grad_scale
    torch.save(total, os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_bucket_allreduce_gloo(tmp_path, overlap):
    world, port = 2, 29000 + (os.getpid() + int(overlap)) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path), overlap), nprocs=world, join=True)
    want = torch.zeros(1003)
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        want += torch.randn(1003, generator=g) + torch.randn(1003, generator=g)
    want /= world
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert torch.allclose(got, want, atol=1e-6)
