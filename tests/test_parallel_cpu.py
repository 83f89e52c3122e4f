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


def _worker(rank, world, port, out, overlap, bucket_bytes=None):
    sys.path.insert(0, ROOT)
    from chap_amd.parallel import DataParallelSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1003
    both = torch.zeros(2 * n)
    b0, b1 = both[:n], both[n:]         # bucket 0 = phase B (BCP) gradients, bucket 1 = phase V (VAT branch)
    sync = DataParallelSync(both, dist, overlap=overlap) if bucket_bytes is None else DataParallelSync(both, dist, overlap=overlap, bucket_bytes=bucket_bytes)
    if bucket_bytes is not None:        # 1003 floats in pieces of 250: 250 + 250 + 250 + 250 + 3, contiguous and in order
        ps = sync.pieces(b0)
        assert [p.numel() for p in ps] == [250, 250, 250, 250, 3] and all(p.data_ptr() == b0.data_ptr() + 1000 * i for i, p in enumerate(ps))
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


@pytest.mark.parametrize("overlap,bucket_bytes", [(True, None), (False, None), (False, 1000), (True, 1000)])
def test_two_bucket_allreduce_gloo(tmp_path, overlap, bucket_bytes):
    """bucket_bytes = 1000: each gradient half goes out in five pieces (<= 16 MB pieces of the 49.4 MB 3D buffer) -- same sums."""
    world, port = 2, 29000 + (os.getpid() + int(overlap) + (7 if bucket_bytes else 0)) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path), overlap, bucket_bytes), nprocs=world, join=True)
    want = torch.zeros(1003)
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        want += torch.randn(1003, generator=g) + torch.randn(1003, generator=g)
    want /= world
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert torch.allclose(got, want, atol=1e-6)


def _bench(*argv, timeout=120):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + [str(a) for a in argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_self_launch_plumbing_dry_run():
    """`python bench.py --gpus N` without a launcher (bench.py:self_launch): N fresh rank processes are started before anything
    touches a GPU, rendezvous on 127.0.0.1, barrier / timed region / barrier with the MAX over ranks, and rank 0's ONE JSON line is
    forwarded on stdout -- rehearsed on the CPU (gloo, --dry-run: no GPU, no model)."""
    import json
    port = 29200 + os.getpid() % 500
    r = _bench("--gpus", 2, "--dry-run", "--steps", 4, "--master-port", port)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["steps"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["ms_per_step"] >= 2.0          # the slowest rank (rank 1 sleeps 2 ms per step) sets the time: MAX over ranks


def test_bench_self_launch_propagates_a_failing_rank():
    """A rank other than 0 dying early (port in use, RCCL init error) must not leave rank 0 hanging in the rendezvous: the launcher
    watches all children, stops the others and exits with the failing rank's code."""
    import time
    port = 29700 + os.getpid() % 250
    t0 = time.time()
    r = _bench("--gpus", 3, "--dry-run", "--steps", 2, "--dry-run-fail-rank", 2, "--master-port", port, "--launch-timeout", 90)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 60
    assert "rank 2 exited with code 3" in r.stderr
    assert r.stdout.strip() == ""
