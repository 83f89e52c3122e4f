#!/usr/bin/env python3
"""bench.py -- CHAP training throughput on MI355X (BASELINE.json metric: training volumes/sec).

    python bench.py --gpus N --steps K --warmup W [--config 2d|3d] [--dtype bf16|fp32]

One "step" = one full iteration of train() (code/train_ours_2D.py:301-389): pass A on the unlabeled
half, BCP mix, pass B fwd/bwd with the four mix_loss terms, VAT (K power iterations + final pass),
SGD -- on a synthetic fixed-seed batch already resident in HBM.  N=1 workload = BASELINE config 1
(ACDC 2D DualDecoder, bs=24 = 12 lab + 12 unlab, 256x256, 1 perturbation step, bf16).
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

T_START = time.perf_counter()

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# ideal-fusion algorithmic bytes per training volume = (3.5 + K) * Bf
BF_2D = {"bf16": 66.7e6, "fp32": 133.4e6}
F_2D = 9.868e9                  # forward FLOPs per 256x256 slice
BF_3D = {"bf16": 611.6e6, "fp32": 1223.3e6}     # per 112x112x80 patch
F_3D = 175.66e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="2d", choices=["2d", "3d"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=0, help="default 24 (2d) / 4 (3d)")
    ap.add_argument("--size", type=int, default=256, help="2d: H = W")
    ap.add_argument("--size3d", type=int, nargs=3, default=[112, 112, 80])
    ap.add_argument("--vat-iters", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dp-overlap", action="store_true", help="data-parallel: all-reduce bucket 0 beside the VAT chain (four-graph replay) instead of "
                    "folding the buckets and all-reducing once at the end (default: measured faster, see DESIGN.md section 6)")
    ap.add_argument("--master-port", type=int, default=29531)
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="override a ChapStep argument (experiments), e.g. --set vat_early=0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)       # ~13 s of CPU work at 2D config 1 (9.4 vol/s on 16 cores)
    return ap.parse_args()


def log(msg):
    print("[bench %6.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def usable_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(args, B, sp):
    """The oracle (CPU restatement, 'port') timed on this box's host cores on the same workload."""
    from oracle import init as oinit
    from oracle import nets as onets
    from oracle import train_step as ots
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    log("cpu_baseline: %d threads" % ncores)
    d3 = len(sp) == 3
    state = oinit.dual_decoder_3d_state(1337) if d3 else oinit.dual_decoder_2d_state(1337)
    sd = {k: v.clone() for k, v in state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
    if d3:
        vol, lab = ots.synthetic_batch_3d(1337, B // 2, B - B // 2, *sp)
        a = dict(labeled_bs=B // 2, vat_iters=args.vat_iters, num_classes=2)
        net, box, iters = onets.dual_decoder_3d, (5, 6, 7), 2            # ~13 s (0.6 vol/s on 16 cores)
    else:
        vol, lab = ots.synthetic_batch(1337, B // 2, B - B // 2, *sp)
        a = dict(labeled_bs=B // 2, vat_iters=args.vat_iters)
        net, box, iters = onets.dual_decoder_2d, (10, 20), args.cpu_iters
    ots.iteration(sd, moms, vol, lab, box, 0, 0.01, args=a, net=net)          # warm-up
    log("cpu_baseline: warm-up iteration done")
    t0 = time.perf_counter()
    for i in range(iters):
        ots.iteration(sd, moms, vol, lab, box, i + 1, 0.01, args=a, net=net)
    dt = time.perf_counter() - t0
    return {"value": B * iters / dt, "unit": "volumes/s", "cores": ncores, "kind": "port",
            "sample": "%d iteration(s) of the same bs=%d %s iteration (oracle/train_step.py, fp32, torch CPU)" % (iters, B, "x".join(map(str, sp)))}


def dominant_kernel_roofline(model, dtype, N, H):
    """HIP-event timing of the dominant kernel (the 16->16 3x3 implicit-GEMM conv at full resolution,
    `conv_fwd_kernel<.., KC=16, NT=1>`) on the stream it is launched on, on the real layer shape."""
    from chap_amd import _lib as L
    from chap_amd import ops
    dev = torch.device("cuda")
    x = torch.randn(N, 1, H, H, 16, device=dev).to(dtype)
    out = torch.empty_like(x)
    w = torch.randn(16, 16, 3, 3, device=dev) / 12
    scale, shift = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.1
    stats = ops.stats_buffer(16, dev)
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dtype, 16, 16, 9)
    src = ops.Lazy(x, scale, shift, True, 0.01)

    def launch():
        ops.conv_fwd([src], wp, None, 16, out, grid=(N, 1, H, H), in_dims=(1, H, H), ksize=3, stride=1, dims=2, stats=stats)

    for _ in range(5):
        launch()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    esz = 2 if dtype == torch.bfloat16 else 4
    alg_bytes = 2.0 * N * H * H * 16 * esz                      # read the input once + write the output once
    ach = alg_bytes / (us * 1e-6) / 1e9
    kname = "conv_fwd_kernel<%s,3,1,2D,KC16,NT1> 16->16 @%dx%d N=%d" % ("bf16" if esz == 2 else "f32", H, H, N)
    traffic = None                 # HBM bytes per launch from the committed rocprofv3 --pmc passes (same kernel, same shape)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")))
        traffic = pmc[kname]["traffic_bytes_per_launch"]
    except Exception:
        pass
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "kernel": kname,
            "avg_launch_us": round(us, 2), "algorithmic_bytes_per_launch": alg_bytes}


MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # MI355X_MICROARCH.md: dense MFMA peaks


def top_kernel_roofline(dtype, N, cfg):
    """The kernel INSTANCE with the largest share of the iteration's kernel time (profiles/r02_bench*_kernel_stats.csv): the
    deep-layer conv `conv_fwd_kernel<bf16,3,1,KC32,NT2,MR2>` (2D, the 32..256-channel 3x3 layers) / the z-brick kernel of the
    64..128-channel 3x3x3 layers (3D), timed live on its most frequent layer shape -- 128->128 at 32x32 (2D), 64->64 at
    28x28x20 (3D) -- as a captured graph of back-to-back launches (the kernel is shorter than a Python launch).  MFMA-bound by
    arithmetic intensity (576 FLOP/B in bf16), so the fraction is against the dense MFMA peak."""
    from chap_amd import _lib as L
    from chap_amd import ops
    dev = torch.device("cuda")
    if cfg == "3d":
        C, (D, H, W), taps, dims = 64, (28, 28, 20), 27, 3
    else:
        C, (D, H, W), taps, dims = 128, (1, 32, 32), 9, 2
    x = torch.randn(N, D, H, W, C, device=dev).to(dtype)
    out = torch.empty_like(x)
    w = torch.randn(*([C, C] + [3] * dims), device=dev) / (C * taps) ** 0.5
    scale, shift = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    stats = ops.stats_buffer(C, dev)
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dtype, C, C, taps)
    src = ops.Lazy(x, scale, shift, True, 0.01)

    def launch():
        ops.conv_fwd([src], wp, None, C, out, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=dims, stats=stats)

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    reps = 20
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            launch()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2.0 * N * D * H * W * taps * C * C
    esz = 2 if dtype == torch.bfloat16 else 4
    tf = flops / us / 1e6
    peak = MFMA_PEAK_TFLOPS["bf16" if esz == 2 else "fp32"]
    return {"kernel": "conv_fwd_kernel<%s,3,1,%s,KC32> %d->%d @%s N=%d" % ("bf16" if esz == 2 else "f32", "3D z-brick" if dims == 3 else "2D,NT2,MR2", C, C,
                                                                      "x".join(map(str, (D, H, W) if dims == 3 else (H, W))), N),
            "bound": "mfma", "avg_launch_us": round(us, 2), "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
            "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": 2.0 * N * D * H * W * C * esz}


def dominant_kernel_roofline_3d(dtype, N, sp):
    """3D: the 16->16 3^3 conv at full resolution (block_nine / its dgrad), KC=16, NT=1."""
    from chap_amd import _lib as L
    from chap_amd import ops
    dev = torch.device("cuda")
    D, H, W = sp
    x = torch.randn(N, D, H, W, 16, device=dev).to(dtype)
    out = torch.empty_like(x)
    w = torch.randn(16, 16, 3, 3, 3, device=dev) / 20
    scale, shift = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.1
    stats = ops.stats_buffer(16, dev)
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dtype, 16, 16, 27)
    src = ops.Lazy(x, scale, shift, True, 0.0)

    def launch():
        ops.conv_fwd([src], wp, None, 16, out, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=3, stats=stats)

    for _ in range(3):
        launch()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    esz = 2 if dtype == torch.bfloat16 else 4
    alg_bytes = 2.0 * N * D * H * W * 16 * esz
    ach = alg_bytes / (us * 1e-6) / 1e9
    kname = "conv_fwd_kernel<%s,3,1,3D,KC16,NT1> 16->16 @%dx%dx%d N=%d" % ("bf16" if esz == 2 else "f32", D, H, W, N)
    traffic = None                 # HBM bytes per launch from the committed rocprofv3 --pmc passes (same kernel, same shape)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")))
        traffic = pmc[kname]["traffic_bytes_per_launch"]
    except Exception:
        pass
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "kernel": kname,
            "avg_launch_us": round(us, 2), "algorithmic_bytes_per_launch": alg_bytes}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU, RCCL rendezvous on 127.0.0.1)
    BEFORE anything in this process touches the GPU, forward rank 0's JSON line, exit with the worst child's code."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.master_port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, p.wait())
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # Libraries (RCCL prints a version banner) must not pollute stdout: the contract is ONE JSON line there.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node == --gpus (or run `python bench.py --gpus N` plainly: it starts the ranks itself)" % (args.gpus, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in chap_amd)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_dp = os.environ.get("CHAP_FORCE_DP") == "1"       # exercise the RCCL path with a 1-rank group (1-GPU box)
    if world > 1 or force_dp:
        import torch.distributed as dist
        if force_dp and "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"))
        dist.init_process_group("nccl", device_id=dev)

    from chap_amd.networks import DualDecoder, DualDecoder3d
    from chap_amd.train import ChapStep
    from chap_amd import synthetic as ots   # fixed-seed synthetic inputs; the oracle is only used by cpu_baseline()

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    d3 = args.config == "3d"
    B = args.batch or (4 if d3 else 24)
    H = args.size
    sp = tuple(args.size3d) if d3 else (H, H)
    torch.manual_seed(1337)
    extra = {}
    for kv in args.set:
        k, v = kv.split("=", 1)
        extra[k] = json.loads(v) if v[:1] in "0123456789-[{tfn\"" else v
    if d3:
        model = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(dev).train().set_compute_dtype(dtype)
        step = ChapStep(model, dict(dict(batch_size=B, labeled_bs=B // 2, vat_iters=args.vat_iters, num_classes=2), **extra), world_size=world)
    else:
        model = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train().set_compute_dtype(dtype)
        step = ChapStep(model, dict(dict(batch_size=B, labeled_bs=B // 2, vat_iters=args.vat_iters), **extra), world_size=world)
    if dist is not None:
        from chap_amd.parallel import DataParallelSync
        step.grad_sync = DataParallelSync(step.grad_both, dist, overlap=args.dp_overlap)
    if d3:
        vol, lab = ots.synthetic_batch_3d(1337 + rank, B // 2, B - B // 2, *sp)
    else:
        vol, lab = ots.synthetic_batch(1337 + rank, B // 2, B - B // 2, H, H)    # each rank: its own shard (weak scaling)
    vol, lab = vol.to(dev), lab.to(dev)
    use_graph = not args.no_graph
    log("model + data ready (B=%d, %s, %s)" % (B, "x".join(map(str, sp)), args.dtype))
    if use_graph:
        step.capture(vol, lab, warmup=2)
        log("graph captured")
        run = lambda: step.replay(vol, lab)                                    # noqa: E731
    else:
        run = lambda: step.step(vol, lab)                                      # noqa: E731
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    log("warm-up done")
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    log("timed region done: %.1f ms/step" % (dt / args.steps * 1e3))
    finite = bool(torch.isfinite(out["vat_loss"]).all()) and all(bool(torch.isfinite(l).all()) for l in out["mix_losses"])
    vps = B * world * args.steps / dt
    if rank == 0:
        K = args.vat_iters
        if d3:
            vox = sp[0] * sp[1] * sp[2] / (112.0 * 112 * 80)
            bytes_per_vol = (3.5 + K) * BF_3D[args.dtype] * vox
            flops_per_vol = (3.5 + K) * F_3D * vox
            roof = dominant_kernel_roofline_3d(dtype, B // 2, sp)
        else:
            bytes_per_vol = (3.5 + K) * BF_2D[args.dtype] * (H * H) / (256 * 256)
            flops_per_vol = (3.5 + K) * F_2D * (H * H) / (256 * 256)
            roof = dominant_kernel_roofline(model, dtype, B // 2, H)
        roof["top_kernel"] = top_kernel_roofline(dtype, B // 2, args.config)
        roof["iteration_hbm_frac_vs_ideal_fusion"] = round(vps / world * bytes_per_vol / 1e9 / HBM_PEAK_GBS, 4)
        roof["iteration_tflops"] = round(vps / world * flops_per_vol / 1e12, 2)
        wl = ("LA 3D DualDecoder3d (V-Net) bs=%d (%d lab + %d unlab) %s patches, %d perturb step(s), N_v=U" % (B, B // 2, B - B // 2, "x".join(map(str, sp)), K)) if d3 else \
             ("ACDC 2D DualDecoder bs=%d (%d lab + %d unlab) %dx%d, %d perturb step(s), N_v=U" % (B, B // 2, B - B // 2, H, H, K))
        line = {"metric": "training volumes/sec (%s)" % ("3D 112x112x80 bs4" if d3 else "2D 256^2 bs24"), "value": round(vps, 2), "unit": "volumes/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": wl,
                           "global_batch": B * world, "parallelism": "dp%d" % world, "hip_graph": use_graph, "losses_finite": finite,
                           "grad_exchange": None if dist is None else ("rccl all-reduce, bucket 0 overlapped with the VAT chain" if args.dp_overlap else "rccl all-reduce of the folded buckets")},
                "roofline": roof}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, B, sp)
        else:
            line["cpu_baseline"] = None
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
