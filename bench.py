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
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs (config 3 = 3D bf16, and the 2D fp32 parity mode) of the default run")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--stage", action="store_true", help="with --host-inputs: ChapStep.stage() the next batch beside the running iteration (what chap_amd's train() does)")
    ap.add_argument("--host-inputs", choices=["pinned", "pageable"], default=None, help="hand every step HOST tensors (the reference's loader yields CPU batches): the "
                    "PCIe-inclusive rate for DESIGN.md -- never the headline value, which is measured with the inputs resident in HBM")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched ranks (--gpus N without a launcher): overall limit in seconds")
    ap.add_argument("--rehearse-one-gpu", action="store_true", help="run the --gpus N ranks on cuda:0 together (1-GPU box), gradients all-reduced through host "
                    "memory over gloo: exercises the whole multi-rank path of this script end to end; the value is NOT a scaling measurement")
    ap.add_argument("--dry-run", action="store_true", help="rehearse the multi-rank plumbing on the CPU (gloo, no GPU, no model): rendezvous, barriers, "
                    "MAX-over-ranks timing, rank 0's JSON line -- what tests/test_parallel_cpu.py drives")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="(dry run) this rank exits with code 3 before the rendezvous")
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
    """HIP-event timing of the dominant kernel (the 16->16 3x3 implicit-GEMM conv at full resolution: since round 4 the
    wave-private `conv_wp_kernel<KC=16, NT=1>` in bf16, `conv_fwd_kernel<.., KC=16, NT=1>` in fp32 or with CHAP_CONV_WP=0) on the
    stream it is launched on, on the real layer shape."""
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
    wp = esz == 2 and os.environ.get("CHAP_CONV_WP", "1") != "0"          # the routing rule of csrc/conv_api.hip
    kname = ("conv_wp_kernel<bf16,KC16,NT1> 16->16 @%dx%d N=%d" % (H, H, N) if wp else
             "conv_fwd_kernel<%s,3,1,2D,KC16,NT1> 16->16 @%dx%d N=%d" % ("bf16" if esz == 2 else "f32", H, H, N))
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


def top_instance_from_profile(cfg):
    """Name + share of the kernel instance with the most kernel time in the committed rocprofv3 --stats summary of this bench
    command (profiles/rNN_bench{2d,3d}_kernel_stats.csv, newest round present)."""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench%s_kernel_stats.csv" % cfg)))
    if not files:
        return None
    try:
        rows = list(csv.DictReader(open(files[-1])))
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from kname import short_name
        name = top.get("Kernel") or short_name(top["Name"])
        return {"instance": name, "share_of_kernel_time": round(float(top["Percentage"]) / 100.0, 4), "avg_ns_in_iteration": round(float(top["AverageNs"])),
                "source": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def top_kernel_roofline(dtype, N, cfg, sp=None):
    """The conv family with the largest share of the iteration's kernel time: the deep-layer 3x3 conv (2D, the 32..256-channel
    layers, `conv_fwd_kernel<bf16,3,1,2D,KC32,NT2,MR2>`) / the z-brick kernel of the 64..128-channel 3x3x3 layers (3D,
    `conv_fwd_kernel<bf16,3,1,3D,KC16,NT2,MR4,ZW>`), timed live on its most frequent layer shape -- 128->128 at H/8 x W/8 (2D),
    64->64 at D/4 x H/4 x W/4 (3D) -- as a captured graph of back-to-back launches (the kernel is shorter than a Python launch).
    MFMA-bound by arithmetic intensity (576 FLOP/B in bf16), so the fraction is against the dense MFMA peak.  `profile` names the
    instance that tops the committed kernel statistics."""
    from chap_amd import _lib as L
    from chap_amd import ops
    dev = torch.device("cuda")
    if cfg == "3d":
        sp = sp or (112, 112, 80)
        C, (D, H, W), taps, dims = 64, tuple(s // 4 for s in sp), 27, 3
    else:
        sp = sp or (256, 256)
        C, (D, H, W), taps, dims = 128, (1, sp[0] // 8, sp[1] // 8), 9, 2
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
    # thread_local: with a live RCCL process group its watchdog thread polls events, which aborts a global-mode capture
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
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
    inst = ("3D,KC16,NT2,MR4 z-brick" if esz == 2 else "3D,KC16,NT2,MR1 slab") if dims == 3 else "2D,KC32,NT2,MR2"
    return {"kernel": "conv_fwd_kernel<%s,3,1,%s> %d->%d @%s N=%d" % ("bf16" if esz == 2 else "f32", inst, C, C,
                                                                      "x".join(map(str, (D, H, W) if dims == 3 else (H, W))), N),
            "bound": "mfma", "avg_launch_us": round(us, 2), "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
            "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": 2.0 * N * D * H * W * C * esz,
            "profile": top_instance_from_profile(cfg) if esz == 2 else None}      # (the committed kernel statistics are bf16 runs)


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
    BEFORE anything in this process touches the GPU, forward rank 0's JSON line, exit with the worst child's code.
    All children are watched: when one exits non-zero (port in use, RCCL init error, ...) the others -- which would hang in the
    rendezvous or in a collective -- are terminated and that code is returned; an overall limit (--launch-timeout) bounds the run.
    Only fresh child processes are started (never an exec from a process that has touched the GPU)."""
    import subprocess
    import tempfile
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.master_port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs, outs = [], []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        f = tempfile.TemporaryFile()          # a file, not a pipe: nobody has to drain it while we poll
        outs.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e, stdout=f))
    deadline = time.monotonic() + args.launch_timeout
    rc, why = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = bad[0][1], "rank %d exited with code %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            rc, why = 124, "no result within --launch-timeout %.0f s" % args.launch_timeout
            break
        time.sleep(0.1)
    if why is not None:
        log("self_launch: %s; stopping the other ranks" % why)
        for p in procs:
            if p.poll() is None:
                p.terminate()                 # the exact PIDs started above
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for r, f in enumerate(outs[1:], 1):   # what the other ranks printed is evidence when something went wrong
            f.seek(0)
            txt = f.read().decode(errors="replace").strip()
            if txt:
                log("rank %d stdout: %s" % (r, txt[-2000:]))
    outs[0].seek(0)
    sys.stdout.write(outs[0].read().decode())
    sys.stdout.flush()
    raise SystemExit(rc if rc >= 0 else 128 - rc)


def dry_run(args):
    """The multi-rank plumbing without a GPU (tests/test_parallel_cpu.py): gloo rendezvous on 127.0.0.1, the barrier / timed region /
    barrier bracket with the MAX over ranks, rank 0's ONE JSON line.  The 'step' is a sleep: nothing here is a measurement."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    if rank == args.dry_run_fail_rank:
        raise SystemExit(3)
    real_stdout = os.dup(1)
    os.dup2(2, 1)                           # library banners must not reach stdout (one JSON line there)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        os.write(real_stdout, (json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                           "ms_per_step": round(float(t.item()) / args.steps * 1e3, 3), "config": {"parallelism": "dp%d" % world}}) + "\n").encode())
    dist.destroy_process_group()


def build_step(cfg, dtype_name, B, sp, K, extra, world, dev):
    from chap_amd.networks import DualDecoder, DualDecoder3d
    from chap_amd.train import ChapStep
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    torch.manual_seed(1337)
    if cfg == "3d":
        model = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(dev).train().set_compute_dtype(dtype)
        step = ChapStep(model, dict(dict(batch_size=B, labeled_bs=B // 2, vat_iters=K, num_classes=2), **extra), world_size=world)
    else:
        model = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train().set_compute_dtype(dtype)
        step = ChapStep(model, dict(dict(batch_size=B, labeled_bs=B // 2, vat_iters=K), **extra), world_size=world)
    return model, step, dtype


def synthetic(cfg, seed, B, sp, dev):
    from chap_amd import synthetic as ots   # fixed-seed synthetic inputs; the oracle is only used by cpu_baseline()
    vol, lab = ots.synthetic_batch_3d(seed, B // 2, B - B // 2, *sp) if cfg == "3d" else ots.synthetic_batch(seed, B // 2, B - B // 2, *sp)
    return vol.to(dev), lab.to(dev)


def workload_name(cfg, B, sp, K):
    if cfg == "3d":
        return "LA 3D DualDecoder3d (V-Net) bs=%d (%d lab + %d unlab) %s patches, %d perturb step(s), N_v=U" % (B, B // 2, B - B // 2, "x".join(map(str, sp)), K)
    return "ACDC 2D DualDecoder bs=%d (%d lab + %d unlab) %dx%d, %d perturb step(s), N_v=U" % (B, B // 2, B - B // 2, sp[0], sp[1], K)


def work_per_volume(cfg, dtype_name, sp, K):
    """(3.5 + K) * Bf bytes and (3.5 + K) * F FLOPs per training volume, scaled by the volume size."""
    if cfg == "3d":
        vox = sp[0] * sp[1] * sp[2] / (112.0 * 112 * 80)
        return (3.5 + K) * BF_3D[dtype_name] * vox, (3.5 + K) * F_3D * vox
    px = sp[0] * sp[1] / (256.0 * 256)
    return (3.5 + K) * BF_2D[dtype_name] * px, (3.5 + K) * F_2D * px


def time_steps(run, steps, warmup, dist, dev):
    """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def extra_config(cfg, dtype_name, B, sp, K, steps, dev):
    """One more configuration on the same line (`extra_configs`): the same iteration, captured and timed the same way."""
    model, step, dtype = build_step(cfg, dtype_name, B, sp, K, {}, 1, dev)
    vol, lab = synthetic(cfg, 1337, B, sp, dev)
    step.capture(vol, lab, warmup=2)
    dt, out = time_steps(lambda: step.replay(vol, lab), steps, 3, None, dev)
    finite = bool(torch.isfinite(out["vat_loss"]).all()) and all(bool(torch.isfinite(l).all()) for l in out["mix_losses"])
    vps = B * steps / dt
    bpv, fpv = work_per_volume(cfg, dtype_name, sp, K)
    roof = dominant_kernel_roofline_3d(dtype, B // 2, sp) if cfg == "3d" else dominant_kernel_roofline(model, dtype, B // 2, sp[0])
    roof["top_kernel"] = top_kernel_roofline(dtype, B // 2, cfg, sp)
    roof["iteration_hbm_frac_vs_ideal_fusion"] = round(vps * bpv / 1e9 / HBM_PEAK_GBS, 4)
    roof["iteration_tflops"] = round(vps * fpv / 1e12, 2)
    res = {"workload": workload_name(cfg, B, sp, K), "dtype": dtype_name, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
           "value": round(vps, 2), "unit": "volumes/s", "losses_finite": finite, "hip_graph": True, "roofline": roof}
    del step, model
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    if args.dry_run:
        return dry_run(args)
    # Libraries (RCCL prints a version banner) must not pollute stdout: the contract is ONE JSON line there.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node == --gpus (or run `python bench.py --gpus N` plainly: it starts the ranks itself)" % (args.gpus, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in chap_amd)")
    if args.rehearse_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_dp = os.environ.get("CHAP_FORCE_DP") == "1"       # exercise the RCCL path with a 1-rank group (1-GPU box)
    if world > 1 or force_dp:
        import torch.distributed as dist
        if force_dp and "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"))
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    d3 = args.config == "3d"
    B = args.batch or (4 if d3 else 24)
    H = args.size
    sp = tuple(args.size3d) if d3 else (H, H)
    extra = {}
    for kv in args.set:
        k, v = kv.split("=", 1)
        extra[k] = json.loads(v) if v[:1] in "0123456789-[{tfn\"" else v
    model, step, dtype = build_step(args.config, args.dtype, B, sp, args.vat_iters, extra, world, dev)
    if dist is not None:
        from chap_amd.parallel import DataParallelSync
        if args.rehearse_one_gpu:
            from chap_amd.parallel import HostStagedDist
            step.grad_sync = DataParallelSync(step.grad_both, HostStagedDist(dist), overlap=False)
        else:
            step.grad_sync = DataParallelSync(step.grad_both, dist, overlap=args.dp_overlap)
    vol, lab = synthetic(args.config, 1337 + rank, B, sp, dev)      # each rank: its own shard (weak scaling)
    use_graph = not args.no_graph
    log("model + data ready (B=%d, %s, %s)" % (B, "x".join(map(str, sp)), args.dtype))
    if use_graph:
        step.capture(vol, lab, warmup=2)
        log("graph captured")
    if args.host_inputs:
        vol, lab = vol.cpu(), lab.cpu()
        if args.host_inputs == "pinned":
            vol, lab = vol.pin_memory(), lab.pin_memory()
    if use_graph and args.stage and args.host_inputs:
        step.stage(vol, lab)

        def run():
            out = step.replay()
            step.stage(vol, lab)            # the next batch travels while this iteration runs
            return out
    elif use_graph:
        run = lambda: step.replay(vol, lab)                                    # noqa: E731
    else:
        run = lambda: step.step(vol, lab)                                      # noqa: E731
    dt, out = time_steps(run, args.steps, args.warmup, dist, torch.device("cpu") if args.rehearse_one_gpu else dev)
    log("timed region done: %.2f ms/step" % (dt / args.steps * 1e3))
    finite = bool(torch.isfinite(out["vat_loss"]).all()) and all(bool(torch.isfinite(l).all()) for l in out["mix_losses"])
    vps = B * world * args.steps / dt
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()        # before the roofline probes: they capture graphs, and the RCCL watchdog polls events
    if rank == 0:
        K = args.vat_iters
        bytes_per_vol, flops_per_vol = work_per_volume(args.config, args.dtype, sp, K)
        roof = dominant_kernel_roofline_3d(dtype, B // 2, sp) if d3 else dominant_kernel_roofline(model, dtype, B // 2, H)
        roof["top_kernel"] = top_kernel_roofline(dtype, B // 2, args.config, sp)
        roof["iteration_hbm_frac_vs_ideal_fusion"] = round(vps / world * bytes_per_vol / 1e9 / HBM_PEAK_GBS, 4)
        roof["iteration_tflops"] = round(vps / world * flops_per_vol / 1e12, 2)
        line = {"metric": "training volumes/sec (%s)" % ("3D 112x112x80 bs4" if d3 else "2D 256^2 bs24"), "value": round(vps, 2), "unit": "volumes/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                "data": "synthetic" if not args.host_inputs else "synthetic, handed over as %s HOST tensors every step (PCIe-inclusive: not the headline value)" % args.host_inputs,
                "config": {"workload": workload_name(args.config, B, sp, K),
                           "global_batch": B * world, "parallelism": "dp%d" % world, "hip_graph": use_graph, "losses_finite": finite,
                           "grad_exchange": None if world == 1 and not force_dp else ("rccl all-reduce, bucket 0 overlapped with the VAT chain" if args.dp_overlap else "rccl all-reduce of the folded buckets")},
                "roofline": roof}
        if args.rehearse_one_gpu:
            line["rehearsal"] = "%d ranks on ONE GPU, gradients through host memory (gloo): exercises the multi-rank path, NOT a scaling measurement" % world
        # the other single-GPU configurations on the driver's line (default run only): BASELINE config 3 (3D, bf16) and the fp32 parity mode
        default_run = world == 1 and not force_dp and not d3 and args.dtype == "bf16" and not args.batch and H == 256 and args.vat_iters == 1 and use_graph and not extra
        if default_run and not args.no_extra:
            del step, model
            torch.cuda.empty_cache()
            line["extra_configs"] = []
            for cfg_, dt_, B_, sp_ in (("3d", "bf16", 4, (112, 112, 80)), ("2d", "fp32", 24, (256, 256))):
                log("extra config: %s %s" % (cfg_, dt_))
                line["extra_configs"].append(extra_config(cfg_, dt_, B_, sp_, 1, args.extra_steps, dev))
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, B, sp)
        else:
            line["cpu_baseline"] = None
        os.write(real_stdout, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
