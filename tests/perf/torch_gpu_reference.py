"""How fast is the reference-equivalent iteration with STOCK PyTorch-ROCm kernels (MIOpen/ATen) on
this GPU?  Runs oracle.train_step.iteration (the CPU restatement, device-agnostic torch code) with
all tensors on cuda:0 -- i.e. what the reference itself would execute on an MI355X: fp32, NCHW,
unfused BN/act/dropout, host-side connected components (scipy, as skimage in the reference).
Not part of bench.py's contract; printed for DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from oracle import init as oinit, nets as onets, train_step as ots

def run(cfg):
    dev = torch.device("cuda")
    if cfg == "2d":
        B, sp, state, net, box, a = 24, (256, 256), oinit.dual_decoder_2d_state(1337), onets.dual_decoder_2d, (10, 20), dict(labeled_bs=12)
        vol, lab = ots.synthetic_batch(1337, 12, 12, *sp)
    else:
        B, sp, state, net, box, a = 4, (112, 112, 80), oinit.dual_decoder_3d_state(1337), onets.dual_decoder_3d, (5, 6, 7), dict(labeled_bs=2, num_classes=2)
        vol, lab = ots.synthetic_batch_3d(1337, 2, 2, *sp)
    sd = {k: v.clone().to(dev) for k, v in state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    moms = {k: torch.zeros_like(v) for k, v in sd.items() if v.requires_grad}
    vol, lab = vol.to(dev), lab.to(dev)
    torch.manual_seed(0)
    orig = ots.largest_cc
    ots.largest_cc = lambda seg, n: orig(seg, n).to(dev)          # host round trip like the reference's .cpu().numpy()
    for _ in range(3):
        ots.iteration(sd, moms, vol, lab, box, 0, 0.01, args=a, net=net)
    torch.cuda.synchronize()
    n = 10 if cfg == "2d" else 5
    t0 = time.perf_counter()
    for i in range(n):
        ots.iteration(sd, moms, vol, lab, box, i, 0.01, args=a, net=net)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("stock PyTorch-ROCm (MIOpen, fp32) %s iteration: %.1f ms/step, %.1f volumes/s" % (cfg, dt * 1e3, B / dt), flush=True)

if __name__ == "__main__":
    for c in sys.argv[1:] or ["2d", "3d"]:
        run(c)
