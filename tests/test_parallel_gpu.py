"""The RCCL side of the data-parallel iteration on real GPUs (each rank in a process of its own, tests/dp_worker.py).

* one rank (any GPU box): a 1-rank NCCL group drives exactly the code path of a multi-GPU run -- eager and the four-graph
  replay, overlap and fold schedules; the update must equal the non-DP step's bit for bit (a 1-rank sum is the identity) and
  the fused SGD must leave both gradient buckets zeroed;
* two ranks (skipped unless the box has >= 2 GPUs): both ranks end with IDENTICAL parameters, equal to the CPU oracle's
  "mean of the per-shard gradients" update (DDP semantics)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_worker.py")


def _run(args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.Popen([sys.executable, WORKER] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def _wait(procs, timeout=600):
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out.decode(errors="replace")[-4000:]


@pytest.mark.parametrize("mode,overlap", [("eager", 1), ("graph", 1), ("graph", 0), ("eager", 0)])
def test_one_rank_nccl_group_equals_the_plain_step(tmp_path, mode, overlap):
    port = 29700 + os.getpid() % 200 + 2 * overlap + (1 if mode == "graph" else 0)
    _wait([_run(["--out", tmp_path, "--mode", mode, "--overlap", overlap, "--no-dp"])])
    _wait([_run(["--out", tmp_path, "--mode", mode, "--overlap", overlap, "--port", port])])
    ref = torch.load(os.path.join(tmp_path, "rank0_%s_%d_nodp.pt" % (mode, overlap)))
    got = torch.load(os.path.join(tmp_path, "rank0_%s_%d.pt" % (mode, overlap)))
    assert float(got["buckets"].abs().max()) == 0.0                      # both buckets consumed and zeroed
    assert got["iter_num"] == ref["iter_num"]
    for a, b in zip(got["losses"], ref["losses"]):
        assert torch.equal(a, b)
    bad = [k for k in ref["model"] if not torch.equal(ref["model"][k], got["model"][k])]
    assert not bad, bad[:5]
    assert torch.equal(ref["mom"], got["mom"])


def test_one_rank_nccl_group_bucketed_3d_equals_the_plain_step(tmp_path):
    """the 3D gradient buffer is exchanged in pieces of <= 16 MB.  Here the DualDecoder3d iteration (K = 2) through a 1-rank RCCL
    group with the halves cut into 256 KB pieces (the small test net's 9.4 M-parameter half -> ~150 all-reduces), graph replay, fold schedule: bit
    for bit the plain step, both buckets zeroed."""
    port = 29650 + os.getpid() % 40
    common = ["--out", tmp_path, "--mode", "graph", "--overlap", 0, "--net", "3d", "--vat-iters", 2]
    _wait([_run(common + ["--no-dp"])])
    _wait([_run(common + ["--port", port, "--bucket-bytes", 256 << 10])])
    ref, got = torch.load(os.path.join(tmp_path, "rank0_graph_0_nodp_3d.pt")), torch.load(os.path.join(tmp_path, "rank0_graph_0_3d.pt"))
    assert float(got["buckets"].abs().max()) == 0.0
    for a, b in zip(got["losses"], ref["losses"]):
        assert torch.equal(a, b)
    assert not [k for k in ref["model"] if not torch.equal(ref["model"][k], got["model"][k])]
    assert torch.equal(ref["mom"], got["mom"])


def test_two_ranks_3d_k2_rehearsed_on_one_gpu(tmp_path):
    """BASELINE config 4's data-parallel logic (DualDecoder3d, code/networks/vnet.py:225-238, K = 2 power iterations) with world 2: two rank processes
    on cuda:0, gradient halves all-reduced through host memory (gloo) in 1 MB pieces, captured iteration.  Both ranks end with identical parameters
    (BatchNorm running statistics stay per replica), equal to the oracle's mean-of-shard-gradients update."""
    from oracle import init as oinit
    from oracle import nets as onets
    from oracle import train_step as ots
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker as W
    port = 29850 + os.getpid() % 40
    _wait([_run(["--rank", r, "--world", 2, "--out", tmp_path, "--mode", "graph", "--overlap", 0, "--port", port, "--backend", "gloo", "--net", "3d", "--vat-iters", 2,
                 "--bucket-bytes", 1 << 20]) for r in range(2)])
    r0, r1 = (torch.load(os.path.join(tmp_path, "rank%d_graph_0_3d.pt" % r)) for r in range(2))
    assert float(r0["buckets"].abs().max()) == 0.0 and float(r1["buckets"].abs().max()) == 0.0
    for k in r0["model"]:
        if not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert torch.equal(r0["model"][k], r1["model"][k]), k
    assert not torch.equal(r0["losses"][0], r1["losses"][0])          # the ranks did work on different shards
    state = oinit.dual_decoder_3d_state(402)
    grads = []
    for r in range(2):
        sd = {k: v.clone() for k, v in state.items()}
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
        vol, lab, inj, box = W.shard_inputs_3d(r, 2)
        grads.append(ots.iteration(sd, moms, vol, lab, box, iter_num=W.IT0, lr=0.0, args=W.args_of("3d", 2), inject=inj, net=onets.dual_decoder_3d)["grads"])
    num = den = 0.0
    for k, g0 in grads[0].items():
        g = (g0 + grads[1][k]) / 2.0
        want = state[k] - 0.01 * (g + 1e-4 * state[k])
        upd_o, upd_h = (want - state[k]).double(), (r0["model"][k] - state[k]).double()
        num += float(((upd_o - upd_h) ** 2).sum())
        den += float((upd_o ** 2).sum())
    # 3D, K = 2 at this size is ill-conditioned: the fp32 oracle is itself 0.16 relative L2 from fp64 (profiles/r03_iteration_parity.jsonl, 3d_16x32x16_k2);
    # the bound says "the mean of the two shard gradients", not "one shard's" (which is > 0.7 away)
    assert (num / den) ** 0.5 < 0.35, (num / den) ** 0.5


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_two_ranks_hold_identical_parameters_equal_to_the_mean_of_shard_gradients(tmp_path, mode, backend):
    """backend nccl: one rank per GPU over RCCL (needs two GPUs).  backend gloo: BOTH ranks on cuda:0 of a 1-GPU box, the gradient buckets
    all-reduced through host memory (tests/dp_worker.py) -- the world-2 logic of ChapStep / DataParallelSync (shards, fold or overlap
    schedule, 1/world scaling in the fused SGD, per-replica BatchNorm) on the real kernels, without RCCL."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from oracle import init as oinit
    from oracle import train_step as ots
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker as W
    port = 29900 + os.getpid() % 90 + (5 if backend == "gloo" else 0) + (1 if mode == "graph" else 0)
    overlap = 1 if backend == "nccl" else 0                     # (gloo staging has no streams to overlap: the default fold schedule)
    _wait([_run(["--rank", r, "--world", 2, "--out", tmp_path, "--mode", mode, "--overlap", overlap, "--port", port, "--backend", backend]) for r in range(2)])
    r0 = torch.load(os.path.join(tmp_path, "rank0_%s_%d.pt" % (mode, overlap)))
    r1 = torch.load(os.path.join(tmp_path, "rank1_%s_%d.pt" % (mode, overlap)))
    assert float(r0["buckets"].abs().max()) == 0.0 and float(r1["buckets"].abs().max()) == 0.0
    # parameters (not BatchNorm running statistics: those are per replica, DDP semantics) are identical on the two ranks
    for k in r0["model"]:
        if not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert torch.equal(r0["model"][k], r1["model"][k]), k
    # oracle: gradients of each shard from the same initial state, averaged, ONE SGD step
    state = oinit.dual_decoder_2d_state(301)
    grads = []
    for r in range(2):
        sd = {k: v.clone() for k, v in state.items()}
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
        vol, lab, inj, box = W.shard_inputs(r)
        grads.append(ots.iteration(sd, moms, vol, lab, box, iter_num=W.IT0, lr=0.0, args=W.ARGS, inject=inj)["grads"])
    num = den = 0.0
    for k, g0 in grads[0].items():
        g = (g0 + grads[1][k]) / 2.0
        want = state[k] - 0.01 * (g + 1e-4 * state[k])                  # first step: momentum buffer = g + wd * p
        upd_o, upd_h = (want - state[k]).double(), (r0["model"][k] - state[k]).double()
        num += float(((upd_o - upd_h) ** 2).sum())
        den += float((upd_o ** 2).sum())
    assert (num / den) ** 0.5 < 0.02, (num / den) ** 0.5


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` end to end on a 1-GPU box: the script starts its two rank processes itself (self_launch), both ranks run the
    captured iteration on cuda:0 (--rehearse-one-gpu: gradients all-reduced through host memory over gloo), barriers + MAX-over-ranks timing,
    and rank 0 prints the ONE JSON line with the whole-job value.  A rehearsal of the N > 1 path, not a measurement."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    port = 29400 + os.getpid() % 300
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--master-port", str(port), "--launch-timeout", "400"], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["config"]["global_batch"] == 48 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["losses_finite"] and d["value"] > 0 and "rehearsal" in d
    assert abs(d["value"] - 48 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3          # whole-job volumes/s = global batch / step time
