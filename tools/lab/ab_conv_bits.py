"""Bitwise A/B of chap_conv_fwd between two builds (CHAP_LIBPATH): run with `save PATH` under each build, then `cmp A B`."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch

def cases():
    g = torch.Generator().manual_seed(5)
    for (N, sp, cin, cout, dims) in ((4, (1, 32, 32), 128, 128, 2), (4, (1, 16, 16), 256, 256, 2), (4, (1, 64, 64), 64, 64, 2), (4, (1, 32, 32), 128, 256, 2),
                                     (4, (1, 128, 128), 32, 32, 2), (1, (8, 14, 14), 128, 128, 3), (1, (20, 28, 28), 64, 64, 3)):
        x = torch.randn(N, *sp, cin, generator=g)
        w = torch.randn(*([cout, cin] + [3] * dims), generator=g) / (cin * 3 ** dims) ** 0.5
        sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
        yield N, sp, cin, cout, dims, x, w, sc, sh

if sys.argv[1] == "save":
    from chap_amd import _lib as L, ops
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        for (N, sp, cin, cout, dims, x, w, sc, sh) in cases():
            D, H, W = sp
            xd = x.cuda().to(dt)
            wp = ops.pack_weights(w.cuda(), L.PACK_CONV_FWD, dt, cin, cout, 3 ** dims)
            o = torch.zeros(N, D, H, W, cout, device="cuda", dtype=dt)
            st = ops.stats_buffer(cout, "cuda")
            for rep in range(3):
                ops.conv_fwd([ops.Lazy(xd, sc.cuda(), sh.cuda(), True, 0.01)], wp, None, cout, o, grid=(N, D, H, W), in_dims=(D, H, W), ksize=3, stride=1, dims=dims, stats=st)
                torch.cuda.synchronize()
                out["%s %d->%d @%s rep%d" % (str(dt)[6:], cin, cout, "x".join(map(str, sp)), rep)] = (o.float().cpu().clone(), ops.stats_totals(st, cout).cpu())
    torch.save(out, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        (oa, sa), (ob, sb) = a[k], b[k]
        nd = (oa != ob).sum().item()
        print("%-40s outputs differ in %d of %d elements (max abs %.3g), statistics rel diff %.3g" % (k, nd, oa.numel(), (oa - ob).abs().max().item(), ((sa - sb).abs().max() / sb.abs().max()).item()))
