"""Per-(kernel, layer shape) roofline table of one CHAP iteration.

    python tools/shape_table.py [--config 2d|3d] [--dtype bf16|fp32] [--out profiles/r02_conv_shapes_2d.csv]
                                [--only SUBSTR] [--reps 20] [--trace-plan gpurun_out/plan.json]

1. runs ONE eager iteration of ChapStep at the bench configuration with the `ops.*` wrappers instrumented:
   every convolution / weight-gradient / BatchNorm-backward / pointwise launch is recorded with its layer
   shape (the recorded closure keeps the real tensors alive);
2. replays every UNIQUE (op, shape) alone in a HIP-event timing loop on the launch stream;
3. writes one CSV row per (op, shape): launches per iteration, us per launch, algorithmic bytes (inputs read once +
   outputs written once at the activation width; fp32 slabs for wgrad), FLOPs, fraction of the 8 TB/s HBM roof and
   of the dense MFMA roof (2.5 PFLOP/s bf16, 157.3 TFLOP/s fp32).

With --trace-plan every timing loop sits between two `copy_kernel` marker launches (chap_debug_copy) so that
tools/shape_join.py can attach the kernel INSTANCE names of a `rocprofv3 --kernel-trace` of this same command.
"""
import argparse
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

HBM_PEAK = 8000.0e9
MFMA_PEAK = {"bf16": 2500.0e12, "fp32": 157.3e12}


def esize(t):
    return t.element_size()


def lazy_flags(lz):
    return "".join(f for f, on in (("a", lz.scale is not None), ("r", lz.act), ("k", lz.keep is not None), ("m", lz.chan_mul is not None)) if on) or "-"


class Recorder:
    def __init__(self):
        self.calls = collections.OrderedDict()   # key -> dict(count, fn, desc..)

    def add(self, key, fn, **info):
        e = self.calls.get(key)
        if e is None:
            e = self.calls[key] = dict(count=0, fn=fn, **info)
        e["count"] += 1


def instrument(ops, rec):
    orig = {}

    def wrap(name, describe):
        f = getattr(ops, name)
        orig[name] = f

        def g(*a, **k):
            info = describe(*a, **k)
            if info is not None:
                key = (name,) + info.pop("key")
                rec.add(key, lambda: f(*a, **k), op=name, **info)
            return f(*a, **k)
        setattr(ops, name, g)

    def d_conv(srcs, wpacked, bias, cout, out, *, grid, in_dims, ksize, stride, dims, combine=0, out_mode=0, out_cn=0, out_planar=False, out_f32=False,
               stats=None, **kw):
        n, d, h, w = grid
        cs = [s.C for s in srcs]
        ck = sum(cs) if combine == 0 else cs[0]
        px_out = n * d * h * w
        px_in = n * in_dims[0] * in_dims[1] * in_dims[2]
        e = esize(srcs[0].raw)
        taps = ksize ** dims
        rd = px_in * sum(cs) * e
        wr = px_out * cout * (4 if (out_planar or out_f32) else e)
        fl = 2.0 * px_out * taps * ck * cout
        shape = "%s k%d s%d %s->%d @%s N=%d%s%s%s%s src=%s" % ("%dD" % dims, ksize, stride, "+".join(map(str, cs)) if combine == 0 else "add".join(map(str, cs)), cout,
                                                            "x".join(map(str, (d, h, w) if dims == 3 else (h, w))), n,
                                                            " d2s" if out_mode else "", " planar" if out_planar else "", " stats" if stats is not None else "",
                                                            " f32out" if out_f32 and not out_planar else "", ",".join(lazy_flags(s) for s in srcs))
        return dict(key=(shape,), shape=shape, bytes=rd + wr, flops=fl)

    def d_wgrad(a_srcs, b, dw, strides, *, grid, in_dims, ksize, stride, dims, combine=0, db=None, **kw):
        n, d, h, w = grid
        cs = [s.C for s in a_srcs]
        ca = sum(cs) if combine == 0 else cs[0]
        e = esize(b.raw)
        px_b = n * d * h * w
        px_a = n * in_dims[0] * in_dims[1] * in_dims[2]
        taps = ksize ** dims
        by = px_a * sum(cs) * e + px_b * b.C * e + taps * ca * b.C * 4
        fl = 2.0 * px_b * taps * ca * b.C
        shape = "%dD k%d s%d A=%s B=%d @%s N=%d A=%s B=%s" % (dims, ksize, stride, "+".join(map(str, cs)), b.C, "x".join(map(str, (d, h, w) if dims == 3 else (h, w))), n,
                                                         ",".join(lazy_flags(s) for s in a_srcs), lazy_flags(b))
        return dict(key=(shape,), shape=shape, bytes=by, flops=fl)

    def d_actbwd(lazy, grads, gout, *, g_pool=None, bn_mode=None, mean=None, **kw):
        raw = lazy.raw
        px = raw[..., 0].numel()
        e = esize(raw)
        k = len(grads) + (0.25 if g_pool is not None else 0)
        bn = bn_mode if bn_mode is not None else (1 if mean is not None else 0)
        passes = 2 if bn == 1 else 1
        by = passes * (k + 1) * px * lazy.C * e + px * lazy.C * e
        shape = "C=%d @%s ng=%d%s bn=%d %s" % (lazy.C, "x".join(map(str, raw.shape[:4])), len(grads), " +pool" if g_pool is not None else "", bn, lazy_flags(lazy))
        return dict(key=(shape,), shape=shape, bytes=by, flops=0.0)

    def d_lazy1(tag, mult_in, mult_out):
        def d(lazy, out, *a, **k):
            raw = lazy.raw
            e = esize(raw)
            by = raw[..., 0].numel() * lazy.C * e * mult_in + out.numel() * out.element_size() * mult_out
            shape = "%s C=%d @%s %s" % (tag, lazy.C, "x".join(map(str, raw.shape[:4])), lazy_flags(lazy))
            return dict(key=(shape,), shape=shape, bytes=by, flops=0.0)
        return d

    def d_upbwd(g, g_coff, C, out, *, dims):
        by = g[..., 0].numel() * C * esize(g) + out.numel() * out.element_size()
        shape = "C=%d out@%s" % (C, "x".join(map(str, out.shape[:4])))
        return dict(key=(shape,), shape=shape, bytes=by, flops=0.0)

    def d_bnfin(stats, gamma, *a, **k):
        shape = "C=%d" % gamma.numel()
        return dict(key=(shape,), shape=shape, bytes=0, flops=0.0)

    wrap("conv_fwd", d_conv)
    wrap("wgrad", d_wgrad)
    wrap("act_bwd", d_actbwd)
    wrap("act_pool2", d_lazy1("pool", 1, 2))
    wrap("upsample2x", d_lazy1("up", 1, 1))
    wrap("upsample2x_bwd", d_upbwd)
    wrap("bn_finalize", d_bnfin)
    return orig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2d", choices=["2d", "3d"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="time only rows whose 'op shape' contains this substring")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--trace-plan", default=None)
    ap.add_argument("--vat-iters", type=int, default=1)
    ap.add_argument("--eager", action="store_true", help="time plain launch loops instead of a captured graph (rocprofv3 --pmc passes)")
    args = ap.parse_args()

    from chap_amd import _lib as L
    from chap_amd import ops
    from chap_amd import synthetic as syn
    from chap_amd.networks import DualDecoder, DualDecoder3d
    from chap_amd.train import ChapStep

    dev = torch.device("cuda", 0)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(1337)
    if args.config == "3d":
        B, sp = 4, (112, 112, 80)
        model = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(dev).train().set_compute_dtype(dtype)
        step = ChapStep(model, dict(batch_size=B, labeled_bs=B // 2, vat_iters=args.vat_iters, num_classes=2, concurrent=False))
        vol, lab = syn.synthetic_batch_3d(1337, B // 2, B // 2, *sp)
    else:
        B, sp = 24, (256, 256)
        model = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train().set_compute_dtype(dtype)
        step = ChapStep(model, dict(batch_size=B, labeled_bs=B // 2, vat_iters=args.vat_iters, concurrent=False))
        vol, lab = syn.synthetic_batch(1337, B // 2, B // 2, *sp)
    vol, lab = vol.to(dev), lab.to(dev)
    step.step(vol, lab)                       # warm-up (allocator, packed weights)
    torch.cuda.synchronize()
    rec = Recorder()
    instrument(ops, rec)
    step.step(vol, lab)
    torch.cuda.synchronize()

    marker_src = torch.zeros(4096, device=dev)
    marker_dst = torch.zeros(4096, device=dev)

    import ctypes as C
    L.lib().chap_debug_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]

    def marker():
        lib = L.lib()
        lib.chap_debug_copy(marker_src.data_ptr(), marker_dst.data_ptr(), 4096 * 4, 1, torch.cuda.current_stream().cuda_stream)

    rows, plan = [], []
    for key, e in rec.calls.items():
        name = "%s %s" % (e["op"], e["shape"])
        if args.only and args.only not in name:
            continue
        fn = e["fn"]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        # the timing loop is a captured HIP graph of `reps` back-to-back launches: the Python / ctypes launch cost
        # (20-40 us per call) would otherwise bound every kernel shorter than that
        class _Eager:
            def replay(self):
                for _ in range(args.reps):
                    fn()
        if args.eager:
            g = _Eager()
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(args.reps):
                    fn()
        g.replay()
        torch.cuda.synchronize()
        if args.trace_plan:
            marker()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        if args.trace_plan:
            marker()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        del g
        by, fl = float(e["bytes"]), float(e["flops"])
        rows.append(dict(op=e["op"], shape=e["shape"], launches_per_iter=e["count"], us=round(us, 2), us_per_iter=round(us * e["count"], 1),
                         alg_MB=round(by / 1e6, 3), GFLOP=round(fl / 1e9, 3), GBps=round(by / us / 1e3, 1), hbm_frac=round(by / (us * 1e-6) / HBM_PEAK, 4),
                         TFLOPs=round(fl / us / 1e6, 2), mfma_frac=round(fl / (us * 1e-6) / MFMA_PEAK[args.dtype], 4)))
        plan.append(dict(op=e["op"], shape=e["shape"], reps=args.reps))
        print("%-9s %-78s x%-3d %8.1f us  %7.1f GB/s (%.3f)  %7.1f TF (%.3f)" % (e["op"], e["shape"][:78], e["count"], us, by / us / 1e3, rows[-1]["hbm_frac"],
                                                                            fl / us / 1e6, rows[-1]["mfma_frac"]), flush=True)
    if args.trace_plan:
        json.dump(plan, open(args.trace_plan, "w"))
    rows.sort(key=lambda r: -r["us_per_iter"])
    tot = sum(r["us_per_iter"] for r in rows)
    print("sum of (us x launches) over the listed ops: %.2f ms per iteration" % (tot / 1e3))
    if args.out:
        with open(args.out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)


if __name__ == "__main__":
    main()
