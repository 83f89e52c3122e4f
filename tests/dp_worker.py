"""Helper process of tests/test_parallel_gpu.py (NOT a test module): one data-parallel rank of ChapStep on one GPU.

    python tests/dp_worker.py --rank R --world W --port P --out DIR --mode {eager,graph} --overlap {0,1}

Every rank builds the same fp32 DualDecoder (state recipe 301), takes ITS shard (synthetic batch seeded with the rank,
injected dropout masks / VAT noise seeded with the rank), runs ONE iteration through the RCCL gradient exchange and saves
its parameters, momentum and the two gradient buckets (which the fused SGD must leave zeroed).  Run in a process of its
own so that an RCCL problem cannot take the test session down."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def shard_inputs_3d(rank, K=2, B=4, lbs=2, sp=(16, 32, 16)):
    """Config 4's per-rank iteration in small: DualDecoder3d (code/networks/vnet.py:225-238), K power iterations, Dropout3d channel multipliers."""
    from oracle import init as oinit
    from oracle import train_step as ots
    U = B - lbs
    vol, lab = ots.synthetic_batch_3d(1338 + rank, lbs, U, *sp)
    inj = {"drop_A": oinit.drop_masks_3d(100 * rank + 1, U), "drop_B": oinit.drop_masks_3d(100 * rank + 2, lbs // 2 + U // 2), "drop_VF": oinit.drop_masks_3d(100 * rank + 4, U),
           "d0": torch.rand((U, 1) + tuple(sp), generator=torch.Generator().manual_seed(100 * rank + 5)) - 0.5}
    for k in range(K):
        inj["drop_V%d" % k] = oinit.drop_masks_3d(100 * rank + 30 + k, U)
    return vol, lab, inj, (2, 5 - rank, 3)


def args_of(net, K):
    return dict(ARGS, vat_iters=K) if net == "2d" else dict(labeled_bs=2, batch_size=4, vat_iters=K, num_classes=2)


def shard_inputs(rank, B=8, lbs=4, H=64, W=64):
    from oracle import init as oinit
    from oracle import train_step as ots
    U = B - lbs
    vol, lab = ots.synthetic_batch(1337 + rank, lbs, U, H, W)
    inj = {"drop_A": oinit.drop_masks_2d(100 * rank + 1, U, H, W), "drop_B": oinit.drop_masks_2d(100 * rank + 2, lbs // 2 + U // 2, H, W),
           "drop_V0": oinit.drop_masks_2d(100 * rank + 3, U, H, W), "drop_VF": oinit.drop_masks_2d(100 * rank + 4, U, H, W),
           "d0": torch.rand(U, 1, H, W, generator=torch.Generator().manual_seed(100 * rank + 5)) - 0.5}
    return vol, lab, inj, (7 + rank, 11 - rank)


ARGS = dict(labeled_bs=4, batch_size=8, vat_iters=1)
IT0 = 3000                                  # consistency weight 0.165: both buckets carry weight


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--port", type=int, default=29611)
    ap.add_argument("--out", required=True)
    ap.add_argument("--mode", default="eager", choices=["eager", "graph"])
    ap.add_argument("--overlap", type=int, default=1)
    ap.add_argument("--no-dp", action="store_true", help="the same iteration without a process group (reference for world 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: every rank on cuda:0 (a 1-GPU box), the gradient buckets "
                    "all-reduced through host memory -- the data-parallel LOGIC of a world > 1 on real kernels without a second GPU")
    ap.add_argument("--net", default="2d", choices=["2d", "3d"], help="3d: DualDecoder3d at 16 x 32 x 16 (BASELINE config 4's data-parallel logic)")
    ap.add_argument("--vat-iters", type=int, default=1)
    ap.add_argument("--bucket-bytes", type=int, default=0, help="> 0: all-reduce the gradient halves in pieces of at most this many bytes")
    a = ap.parse_args()
    from chap_amd.networks import DualDecoder, DualDecoder3d
    from chap_amd.parallel import DataParallelSync
    from chap_amd.train import ChapStep
    from oracle import init as oinit
    dev = torch.device("cuda", 0 if a.backend == "gloo" else a.rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    xdist = dist
    if not a.no_dp:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world))
        if a.backend == "gloo":
            dist.init_process_group("gloo", rank=a.rank, world_size=a.world)

            from chap_amd.parallel import HostStagedDist
            xdist = HostStagedDist(dist)
        else:
            dist.init_process_group("nccl", rank=a.rank, world_size=a.world, device_id=dev)
    if a.net == "3d":
        m = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(dev).train()
        m.load_state_dict(oinit.dual_decoder_3d_state(402), strict=True)
    else:
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train()
        m.load_state_dict(oinit.dual_decoder_2d_state(301), strict=True)
    step = ChapStep(m, args_of(a.net, a.vat_iters), world_size=1 if a.no_dp else a.world)
    step.iter_num = IT0
    if not a.no_dp:
        kw = dict(bucket_bytes=a.bucket_bytes) if a.bucket_bytes > 0 else {}
        step.grad_sync = DataParallelSync(step.grad_both, xdist, overlap=bool(a.overlap), **kw)
        if a.bucket_bytes > 0:
            assert len(step.grad_sync.pieces(step.grad2)) > 1, "the bucketed path needs more than one piece"
    if a.net == "3d":
        vol, lab, inj, box = shard_inputs_3d(a.rank, a.vat_iters)
        inj = {k: ({kk: (vv.float() * 2.0).to(dev) for kk, vv in v.items()} if k.startswith("drop") else v.to(dev)) for k, v in inj.items()}      # Dropout3d(0.5) channel multipliers
    else:
        vol, lab, inj, box = shard_inputs(a.rank)
        for k in range(1, a.vat_iters):
            inj["drop_V%d" % k] = oinit.drop_masks_2d(100 * a.rank + 30 + k, vol.shape[0] - 4, 64, 64)
        inj = {k: ({kk: vv.permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(dev) for kk, vv in v.items()} if k.startswith("drop") else v.to(dev)) for k, v in inj.items()}
    vol, lab = vol.to(dev), lab.to(dev)
    if a.mode == "graph":
        step.capture(vol, lab, warmup=1, inject=inj)
        out = step.replay(vol, lab, box_yx=box)
    else:
        out = step.step(vol, lab, box_yx=box, inject=inj)
    torch.cuda.synchronize()
    torch.save({"model": {k: v.cpu() for k, v in m.state_dict().items()}, "mom": step.opt.mom.cpu(), "buckets": step.grad_both.cpu(),
                "losses": [l.cpu() for l in out["mix_losses"]] + [out["vat_loss"].cpu()], "iter_num": step.iter_num},
               os.path.join(a.out, "rank%d_%s_%d%s%s.pt" % (a.rank, a.mode, a.overlap, "_nodp" if a.no_dp else "", "" if a.net == "2d" else "_3d")))
    if not a.no_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
